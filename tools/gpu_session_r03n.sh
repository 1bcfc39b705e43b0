#!/bin/bash
# Round 3, session N: DPP neighbours in the fused two-component kernels: parity + A/B
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/r03n
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_f32.py tests/test_topology.py tests/test_grid_api.py -x -q -m gpu 2>&1 | tail -4 | tee $S/pytest.log
timeout 300 python tools/ab_tunables.py --cases vort,divg,grad,flux --variants "nb_dpp=0;nb_dpp=1;nb_dpp=1,vec_nt=2;nb_dpp=0,vec_nt=2" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_nb_dpp_vec.jsonl | cut -c1-150
