#!/bin/bash
# Round 3, session AA: flat K1 with the neighbour by DPP; K2Sy with 8 waves per workgroup
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/r03aa
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_f32.py tests/test_grid_api.py tests/test_topology.py -x -q -m gpu 2>&1 | tail -4 | tee $S/pytest.log
XG_SEG_YS=2 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -2 | tee -a $S/pytest.log
timeout 300 python tools/ab_tunables.py --cases diffX,diffY --variants "nb_dpp=0;nb_dpp=1;seg_ys=2" --rounds 6 --reps 7 2>&1 | grep '^{' | tee $S/ab_k1dpp.jsonl | cut -c1-150
for v in "XG_NB_DPP=0" "XG_NB_DPP=1" "XG_SEG_YS=2" "XG_NB_DPP=1" "XG_NB_DPP=0"; do env $v timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['per_op_ms'])"; done | tee $S/bench_variants.txt
