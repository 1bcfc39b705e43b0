#!/bin/bash
# Round 3, session AZ: K2Sy: non-temporal loads for the rows nobody reads again (all but the workgroup's top row)
S=$PWD/gpurun_out/r03az
mkdir -p $S
export TMPDIR=/tmp
XG_SEG_YS=2 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -1 | tee $S/pytest.log
timeout 400 python tools/ab_tunables.py --cases diffY,diffX --variants "seg_ys=1;seg_ys=2" --rounds 8 --reps 7 2>&1 | grep '^{' | tee $S/ab_k2sy_nt.jsonl | cut -c1-160
