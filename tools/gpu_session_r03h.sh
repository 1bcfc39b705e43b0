#!/bin/bash
# Round 3, session H: fused vorticity at 16-row bands: x-major / y-stacked workgroups x levels per task
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/r03h
mkdir -p $S
export TMPDIR=/tmp
timeout 400 python tools/ab_tunables.py --shape 90,4320,4320 --cases vort --variants "vec_ystack=0,zb_rows=8,vec_zk=1;vec_ystack=0,zb_rows=8,vec_zk=2;vec_ystack=1,zb_rows=8,vec_zk=1;vec_ystack=1,zb_rows=8,vec_zk=2;vec_ystack=0,zb_rows=6,vec_zk=2;vec_ystack=0,zb_rows=12,vec_zk=2;vec_ystack=0,zb_rows=8,vec_zk=2,vec_nt=0;vec_ystack=0,zb_rows=8,vec_zk=2,vec_nt=1" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_vort.jsonl | cut -c1-170
timeout 400 python tools/pmc_ab.py --shape 90,4320,4320 --cases vort --variants "vec_ystack=0,zb_rows=8,vec_zk=1;vec_ystack=0,zb_rows=12,vec_zk=2;vec_ystack=0,zb_rows=6,vec_zk=2" --pmc "FETCH_SIZE" 2>&1 | tee $S/pmc_vort.jsonl | cut -c1-300
