#!/bin/bash
# Round 3, session B: K5L (level-major scan) parity + A/B against the march; fused vorticity: which loads may be
# non-temporal (timing A/B + FETCH_SIZE per variant; k_binary with / without nt as the calibration of the counter).
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/r03b
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -6 | tee $S/pytest_scan.log
echo "== cumZ: march vs level-major"
timeout 300 python tools/ab_tunables.py --cases cumZ --variants "scan_levels=0;scan_levels=1;scan_levels=-24;scan_levels=-16;scan_levels=-12;scan_levels=-8" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_cumZ_levels.jsonl | cut -c1-170
echo "== config 4, 16 records in batches of 8: march vs level-major"
for lv in 0 1; do XG_SCAN_LEVELS=$lv timeout 300 python tools/bench_configs.py --gpus 1 --configs 4 --records 16 --batch-records 8 2>&1 | grep '^{' | sed "s/^{/{\"scan_levels\": $lv, /" | tee -a $S/config4_levels.jsonl | cut -c1-330; done
echo "== vorticity: timing of the nt variants (4320 x 4320 x 90)"
timeout 400 python tools/ab_tunables.py --shape 90,4320,4320 --cases vort --variants "vec_nt=0;vec_nt=1;vec_nt=2;vec_nt=3" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_vort_nt.jsonl | cut -c1-170
echo "== vorticity + k_binary: FETCH_SIZE per variant"
timeout 600 python tools/pmc_ab.py --shape 90,4320,4320 --cases vort,mulTT --variants "vec_nt=0,nt_load=1;vec_nt=1,nt_load=1;vec_nt=2,nt_load=1;vec_nt=3,nt_load=1;nt_load=0" --pmc "FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum TCC_MISS_sum|TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum|TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" 2>&1 | tee $S/pmc_vort_nt.jsonl | cut -c1-400
