#!/bin/bash
# Round 3, session AS: new / extended parity tests (pad rows per tiles-per-wave, binary op index paths, K4L orders and block heights)
S=$PWD/gpurun_out/r03as
mkdir -p $S
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -3 | tee $S/pytest.log
