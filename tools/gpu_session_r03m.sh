#!/bin/bash
# Round 3, session M: neighbour element by DPP instead of an 8-byte load (K1r, two-axis kernel): parity + A/B
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/r03m
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_f32.py -x -q 2>&1 | tail -4 | tee $S/pytest.log
timeout 300 python tools/ab_tunables.py --cases dX,iXmw,i2,i2mw --variants "nb_dpp=0;nb_dpp=1;nb_dpp=0,met_zk2=2;nb_dpp=1,met_zk2=2" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_nb_dpp.jsonl | cut -c1-150
