#!/bin/bash
# Run on the GPU box from the repository root: collects the round's profile
# artefacts into gpurun_out/prof_<tag>/ (copy the summaries into profiles/).
#   profiles/collect.sh TAG
set -e
TAG=${1:-run}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. per-kernel times of the bench command
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu > $OUT/bench_under_rocprof.log 2>&1
# 2. HBM traffic of the fit kernel: FETCH_SIZE and WRITE_SIZE in passes of their own
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --no-extra > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --no-extra > $OUT/pmc_write.log 2>&1
# 3. instruction mix and VALU activity of the fit kernel
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --no-extra > $OUT/pmc_sq.log 2>&1
# 4. hardware-counted FP64 operations of the Z^2_m kernel (one launch, 1e6 photons x 2^18 frequencies)
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/pmc_z2m -- python3 $R/profiles/z2m_once.py $OUT/z2m_once.json > $OUT/pmc_z2m.log 2>&1
# 5. the plain bench line (no profiler attached)
timeout -k 10 400 python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 $R/profiles/summarize.py $OUT > $OUT/summary.txt
tail -5 $OUT/summary.txt
