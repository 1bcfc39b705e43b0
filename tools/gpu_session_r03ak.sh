#!/bin/bash
# Round 3, session AK: generic pad along the contiguous dim: aligned load + DPP instead of narrow loads
S=$PWD/gpurun_out/r03ak
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_f32.py tests/test_grid_api.py tests/test_topology.py -x -q -m gpu 2>&1 | tail -2 | tee -a $S/pytest.log
timeout 400 python tools/ab_tunables.py --cases padX,padYX,padYZ --variants "pad_tpw=1;pad_tpw=2;pad_tpw=4" --rounds 6 --reps 5 2>&1 | grep '^{' | tee $S/ab_pad_tpw.jsonl | cut -c1-150
