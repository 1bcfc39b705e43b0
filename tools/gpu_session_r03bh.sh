#!/bin/bash
# Round 3, session BH: HBM reads per band height with dropped output lines (vorticity / divergence; derivative Y)
S=$PWD/gpurun_out/r03bh
mkdir -p $S
export TMPDIR=/tmp
timeout 300 python tools/pmc_ab.py --cases vort,divg --variants "vec_zb_rows=16;vec_zb_rows=32" --pmc "FETCH_SIZE" 2>&1 | tee $S/pmc_vec_bands.jsonl | cut -c1-300
timeout 300 python tools/pmc_ab.py --cases dY --variants "zb_rows=16;zb_rows=32" --pmc "FETCH_SIZE" 2>&1 | tee $S/pmc_dy_bands.jsonl | cut -c1-300
timeout 400 python tools/ab_tunables.py --shape 90,4320,4320 --cases vort --variants "vec_zb_rows=16;vec_zb_rows=32" --rounds 5 --reps 3 2>&1 | grep '^{' | tee $S/ab_vort_fullsize.jsonl | cut -c1-150
