#!/bin/bash
# Round 3, session G: band height (rows of a broadcast metric kept in the XCD's L2 across the levels) -- the fused
# vorticity re-reads its area from the fabric once per level group (6.7 GB of 33.8 GB); metric stencils for comparison.
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/r03g
mkdir -p $S
export TMPDIR=/tmp
echo "== vorticity A/B (4320 x 4320 x 90)"
timeout 400 python tools/ab_tunables.py --shape 90,4320,4320 --cases vort --variants "vec_ystack=0,zb_rows=16;vec_ystack=0,zb_rows=8;vec_ystack=0,zb_rows=4;vec_ystack=0,zb_rows=2;vec_ystack=1,zb_rows=16;vec_ystack=1,zb_rows=8;vec_ystack=1,zb_rows=4;vec_ystack=1,zb_rows=2" --rounds 4 --reps 5 2>&1 | grep '^{' | tee $S/ab_vort_bands.jsonl | cut -c1-170
echo "== FETCH_SIZE"
timeout 400 python tools/pmc_ab.py --shape 90,4320,4320 --cases vort --variants "vec_ystack=0,zb_rows=8;vec_ystack=0,zb_rows=4;vec_ystack=0,zb_rows=2;vec_ystack=1,zb_rows=8;vec_ystack=1,zb_rows=4;vec_ystack=1,zb_rows=2" --pmc "FETCH_SIZE" 2>&1 | tee $S/pmc_vort_bands.jsonl | cut -c1-300
echo "== metric stencils: band height"
timeout 400 python tools/ab_tunables.py --cases dX,dY,iXmw,iYmw --variants "zb_rows=16;zb_rows=8;zb_rows=4" --rounds 4 --reps 5 2>&1 | grep '^{' | tee $S/ab_metric_bands.jsonl | cut -c1-170
timeout 400 python tools/pmc_ab.py --cases dX,dY,iXmw,iYmw --variants "zb_rows=16;zb_rows=8;zb_rows=4" --pmc "FETCH_SIZE" 2>&1 | tee $S/pmc_metric_bands.jsonl | cut -c1-300
