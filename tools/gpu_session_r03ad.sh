#!/bin/bash
# Round 3, session AD: conservative transform, lean one-window-per-wave kernel (K9e): parity + A/B; linear lean re-check
S=$PWD/gpurun_out/r03ad
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_transform.py tests/test_gpu_fuzz.py tests/test_f32.py -x -q -m gpu 2>&1 | tail -4 | tee $S/pytest.log
XG_TRANSFORM_LEAN=11 timeout 900 python -m pytest tests/test_transform.py -x -q -m gpu 2>&1 | tail -2 | tee -a $S/pytest.log
timeout 400 python tools/ab_tunables.py --cases tlin_rw,tlin_sm --variants "transform_lean=3;transform_lean=7;transform_lean=11" --rounds 6 --reps 5 2>&1 | grep '^{' | tee $S/ab_tlean.jsonl | cut -c1-150
