#!/bin/bash
# Round 3, session X: K8y (two-axis metric kernel, y-stacked workgroups): parity + A/B + traffic
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/r03x
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_f32.py tests/test_grid_api.py tests/test_gpu_graphs.py -x -q -m gpu 2>&1 | tail -4 | tee $S/pytest.log
timeout 300 python tools/ab_tunables.py --cases i2mw,i2 --variants "met_ys=0;met_ys=1;met_ys=1,nb_dpp=0" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_k8y.jsonl | cut -c1-150
timeout 300 python tools/pmc_ab.py --cases i2mw --variants "met_ys=0;met_ys=1" --pmc "FETCH_SIZE|SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU" 2>&1 | tee $S/pmc_k8y.jsonl | cut -c1-300
