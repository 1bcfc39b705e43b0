#!/bin/bash
# Round 3, session AC: linear transform, lean streaming loop: parity + A/B
S=$PWD/gpurun_out/r03ac
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_transform.py tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_f32.py -x -q -m gpu 2>&1 | tail -4 | tee $S/pytest.log
XG_TRANSFORM_LEAN=0 timeout 900 python -m pytest tests/test_transform.py -x -q -m gpu 2>&1 | tail -2 | tee -a $S/pytest.log
timeout 400 python tools/ab_tunables.py --cases tlin_rw,tlin_sm --variants "transform_lean=0;transform_lean=1" --rounds 6 --reps 5 2>&1 | grep '^{' | tee $S/ab_tlean.jsonl | cut -c1-150
