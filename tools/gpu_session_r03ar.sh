#!/bin/bash
# Round 3, session AR: K1r: the tile's boundary lane from scalar loads instead of a divergent block of per-lane loads
S=$PWD/gpurun_out/r03ar
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; XG_NB_DPP=2 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_f32.py tests/test_grid_api.py tests/test_topology.py -x -q -m gpu 2>&1 | tail -2 | tee -a $S/pytest.log
timeout 400 python tools/ab_tunables.py --cases dX,iXmw --variants "nb_dpp=1;nb_dpp=2" --rounds 8 --reps 5 2>&1 | grep '^{' | tee $S/ab_k1r_scalar.jsonl | cut -c1-160
