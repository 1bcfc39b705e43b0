#!/bin/bash
# Round 3, session AH: K2Sm default for one metric; K2Sy with metrics (not z-banded) and with column chunks (whole-plane rows)
S=$PWD/gpurun_out/r03ah
mkdir -p $S
export TMPDIR=/tmp
echo "== parity (defaults)"; timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_f32.py tests/test_grid_api.py tests/test_topology.py -x -q -m gpu 2>&1 | tail -2 | tee -a $S/pytest.log
echo "== parity seg_ys=7"; XG_SEG_YS=7 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_f32.py tests/test_grid_api.py tests/test_topology.py -x -q -m gpu 2>&1 | tail -2 | tee -a $S/pytest.log
timeout 400 python tools/ab_tunables.py --cases diffZ,dZ,iZmw,dY3,dY --variants "seg_ys=1;seg_ys=7" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_ys_ext.jsonl | cut -c1-150
