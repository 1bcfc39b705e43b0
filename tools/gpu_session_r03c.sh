#!/bin/bash
# Round 3, session C: K5L with interleaved tile ownership (the chip sweeps one contiguous window per step)
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/r03c
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -k "cumsum or config4 or scan" 2>&1 | tail -4 | tee $S/pytest_scan.log
echo "== cumZ: march vs level-major"
timeout 300 python tools/ab_tunables.py --cases cumZ,diffX --variants "scan_levels=0;scan_levels=1,scan_levels_il=0;scan_levels=1,scan_levels_il=1;scan_levels=-24,scan_levels_il=0" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_cumZ_levels.jsonl | cut -c1-170
echo "== config 4, 16 records in batches of 8: march vs level-major"
for lv in 0 1; do XG_SCAN_LEVELS=$lv timeout 300 python tools/bench_configs.py --gpus 1 --configs 4 --records 16 --batch-records 8 2>&1 | grep '^{' | sed "s/^{/{\"scan_levels\": $lv, /" | tee -a $S/config4_levels.jsonl | cut -c1-330; done
