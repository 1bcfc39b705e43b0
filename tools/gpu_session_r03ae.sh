#!/bin/bash
# Round 3, session AE: lean transform kernels: ring / window size; parity of the final build
S=$PWD/gpurun_out/r03ae
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_transform.py tests/test_gpu_fuzz.py tests/test_f32.py -x -q -m gpu 2>&1 | tail -3 | tee $S/pytest.log
timeout 400 python tools/ab_tunables.py --cases tlin_rw,tlin_sm --variants "transform_ring=4;transform_ring=8;transform_ring=16" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_ring.jsonl | cut -c1-150
timeout 400 python tools/ab_tunables.py --cases tcon_rw,tcon_sm --variants "transform_cwin=4;transform_cwin=8;transform_cwin=16" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_cwin.jsonl | cut -c1-150
