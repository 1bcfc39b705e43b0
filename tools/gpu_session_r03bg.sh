#!/bin/bash
# Round 3, session BG: band heights again now that output lines are dropped from the L2 (rule 16): do taller bands hold?
S=$PWD/gpurun_out/r03bg
mkdir -p $S
export TMPDIR=/tmp
timeout 400 python tools/ab_tunables.py --cases vort,divg,grad --variants "vec_zb_rows=16;vec_zb_rows=24;vec_zb_rows=32" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_vec_bands.jsonl | cut -c1-150
timeout 400 python tools/ab_tunables.py --cases dX,dY,iXmw,iYmw --variants "zb_rows=16;zb_rows=24;zb_rows=32" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_metric_bands.jsonl | cut -c1-150
