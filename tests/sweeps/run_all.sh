#!/bin/bash
# Every sweep once with one seed, on the GPU box from the repository root:
#   tests/sweeps/run_all.sh SEED [TAG]
# One log per sweep under gpurun_out/; stops at the first sweep that ends abnormally
# (a time limit, a fault), so that nothing further is started on the GPU after it.
S=${1:-0}
TAG=${2:-sw$S}
mkdir -p gpurun_out
for s in "sweep_fit.py $S 40 100" "sweep_binning.py $S 40" "sweep_exposure.py $S 60" "sweep_spectra_blocks.py $S" "sweep_beta.py $S 40" \
         "sweep_trigsums.py $S 40" "sweep_lightcurve.py $S 30" "sweep_timeseries.py $S 40" "sweep_profile_bkg.py $S 40" \
         "sweep_batch.py $S 30"; do
    set -- $s
    log=gpurun_out/${TAG}_${1%.py}.log
    timeout -k 10 300 python3 -u tests/sweeps/$s > $log 2>&1
    rc=$?
    echo "seed $S $1: $(tail -1 $log)"
    if [ $rc -ne 0 ]; then echo "rc=$rc: stopping"; exit $rc; fi
done
