#!/bin/bash
# Round 3, session Y: K2Sy (plain strided-axis stencil, y-stacked workgroups): parity + A/B + the bench line
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/r03y
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_f32.py tests/test_grid_api.py tests/test_topology.py -x -q -m gpu 2>&1 | tail -4 | tee $S/pytest.log
timeout 300 python tools/ab_tunables.py --cases diffY,diffX --variants "seg_ys=0;seg_ys=1" --rounds 6 --reps 7 2>&1 | grep '^{' | tee $S/ab_k2sy.jsonl | cut -c1-150
for v in 0 1; do XG_SEG_YS=$v timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seg_ys=$v', d['value'], d['ms_per_step'], d['roofline']['per_op_ms'])"; done | tee $S/bench_seg_ys.txt
for v in 1 0; do XG_SEG_YS=$v timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seg_ys=$v', d['value'], d['ms_per_step'], d['roofline']['per_op_ms'])"; done | tee -a $S/bench_seg_ys.txt
timeout 300 python tools/pmc_ab.py --cases diffY --variants "seg_ys=0;seg_ys=1" --pmc "FETCH_SIZE|TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" 2>&1 | tee $S/pmc_k2sy.jsonl | cut -c1-300
