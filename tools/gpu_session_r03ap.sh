#!/bin/bash
# Round 3, session AP: K4L block height with the new work order, f64 and f32
S=$PWD/gpurun_out/r03ap
mkdir -p $S
export TMPDIR=/tmp
XG_REDUCE_LDSW_U=16 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "lds or reduce" 2>&1 | tail -1 | tee $S/pytest.log
timeout 400 python tools/ab_tunables.py --cases sumYw,avgYw --variants "reduce_ldsw_u=8;reduce_ldsw_u=16" --rounds 6 --reps 5 2>&1 | grep '^{' | tee $S/ab_k4l_u.jsonl | cut -c1-160
timeout 400 python tools/ab_tunables.py --dtype f32 --cases sumYw,avgYw --variants "reduce_ldsw_u=8;reduce_ldsw_u=16" --rounds 6 --reps 5 2>&1 | grep '^{' | tee $S/ab_k4l_u_f32.jsonl | cut -c1-160
