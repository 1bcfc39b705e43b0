#!/bin/bash
# Round 3, session F: fused vorticity with y-stacked workgroups (halo rows through LDS) -- parity, timing A/B, FETCH_SIZE;
# cheap march variants for cumsum along Z.
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/r03f
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_topology.py tests/test_grid_api.py -x -q -k "vort or curl or config5 or fused" 2>&1 | tail -4 | tee $S/pytest_vort.log
echo "== vorticity A/B (4320 x 4320 x 90)"
timeout 400 python tools/ab_tunables.py --shape 90,4320,4320 --cases vort --variants "vec_ystack=0;vec_ystack=1;vec_ystack=1,vec_nt=1;vec_ystack=1,vec_zk=1;vec_ystack=1,vec_zk=4;vec_ystack=0,vec_zk=1" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_vort_ystack.jsonl | cut -c1-170
echo "== FETCH_SIZE"
timeout 400 python tools/pmc_ab.py --shape 90,4320,4320 --cases vort --variants "vec_ystack=0;vec_ystack=1;vec_ystack=1,vec_zk=1" --pmc "FETCH_SIZE|WRITE_SIZE" 2>&1 | tee $S/pmc_vort_ystack.jsonl | cut -c1-300
echo "== cumZ march variants"
timeout 300 python tools/ab_tunables.py --cases cumZ --variants "scan_levels=0;scan_levels=0,march_lds_kb=22;scan_levels=0,march_lds_kb=24;scan_levels=0,march_lds_kb=32;scan_levels=0,scan_pipe=2;scan_levels=0,march_band=0;scan_levels=-24" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_cumZ_march.jsonl | cut -c1-170
