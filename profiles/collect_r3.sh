#!/bin/bash
# Run on the GPU box from the repository root: collects this round's profile artefacts into
# gpurun_out/prof_<tag>/ ; profiles/summarize_r3.py condenses them (copy its output into profiles/).
#   profiles/collect_r3.sh TAG [a|b|all]     a: the counter and trace passes (steps 1-6), b: the rest
TAG=${1:-run}
PART=${2:-all}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FIT="--steps 12 --warmup 4 --no-cpu --no-extra"
# the traced and counted launches have the GPU to themselves: one stream (the plain bench line of
# step 7 deals its timed steps over three, where launches overlap and each one alone takes longer)
export GODOT_BENCH_STREAMS=1
if [ "$PART" != b ]; then
# 1. per-kernel times of the bench command (C4 leg cut to 2^24 trial frequencies to bound the traced run)
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu --nfreq 16777216 > $OUT/bench_under_rocprof.log 2>&1 || exit 1
# 2. HBM traffic of the fit kernel on the rotating lists: FETCH_SIZE and WRITE_SIZE in passes of their own
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $FIT > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $FIT > $OUT/pmc_write.log 2>&1 || exit 1
# 3. instruction mix and VALU activity of the fit kernel
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $FIT > $OUT/pmc_sq.log 2>&1 || exit 1
# 4. hardware-counted FP64 operations of the Z^2_2 kernel as the bench launches it
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_z2m -- python3 $R/profiles/z2m_once.py $OUT/z2m_once.json > $OUT/pmc_z2m.log 2>&1 || exit 1
# 5. the FFT power spectrum at the C3 size: kernel times and HBM traffic of every kernel of the call
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fft_trace -- python3 $R/profiles/fft_once.py > $OUT/fft_trace.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fft_fetch -- python3 $R/profiles/fft_once.py > $OUT/fft_fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/fft_write -- python3 $R/profiles/fft_once.py > $OUT/fft_write.log 2>&1 || exit 1
# 6. the profile grid of the C2 cells: instruction counts, VALU activity, bytes written
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/grid_sq -- python3 $R/profiles/grid_once.py > $OUT/grid_sq.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/grid_write -- python3 $R/profiles/grid_once.py > $OUT/grid_write.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/grid_trace -- python3 $R/profiles/grid_once.py > $OUT/grid_trace.log 2>&1 || exit 1
fi
unset GODOT_BENCH_STREAMS
if [ "$PART" = a ]; then exit 0; fi
# 7. the timed steps as the bench deals them over three streams: kernel start and end stamps
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace3 -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu --no-extra > $OUT/bench3_under_rocprof.log 2>&1 || exit 1
# 8. the multi-source job at its full size: stage breakdown, and per-kernel times of 256 sources
PYTHONPATH=$R timeout -k 10 600 python3 $R/profiles/c5_full.py 1000 > $OUT/c5_full.txt 2>&1 || exit 1
PYTHONPATH=$R timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_trace -- python3 $R/profiles/c5_full.py 256 > $OUT/c5_trace.log 2>&1 || exit 1
# 9. the C2 light curve end to end
timeout -k 10 300 python3 $R/profiles/lightcurve_time.py > $OUT/lightcurve_time.txt 2>&1 || exit 1
# 10. the plain bench line (no profiler attached), streams as the bench chooses them, twice
T0=$SECONDS
timeout -k 10 600 python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
WALL=$((SECONDS - T0))
timeout -k 10 600 python3 $R/bench.py --no-cpu > $OUT/bench_again.json 2> $OUT/bench_again.err || exit 1
python3 $R/profiles/summarize_r3.py $OUT > $OUT/summary.txt
echo "# wall time of the plain 'python3 bench.py' above: $WALL s" >> $OUT/summary.txt
tail -5 $OUT/summary.txt
