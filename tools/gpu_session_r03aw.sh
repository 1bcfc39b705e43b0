#!/bin/bash
# Round 3, session AW: K2Sy / K2Sm with 32-bit index arithmetic (fewer scalar instructions per wave): parity + A/B against K2S
S=$PWD/gpurun_out/r03aw
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_f32.py tests/test_grid_api.py tests/test_topology.py -x -q -m gpu 2>&1 | tail -2 | tee -a $S/pytest.log
timeout 400 python tools/ab_tunables.py --cases diffY,dY --variants "seg_ys=0,met_ys1=0;seg_ys=1,met_ys1=12" --rounds 8 --reps 7 2>&1 | grep '^{' | tee $S/ab_lean_ys.jsonl | cut -c1-170
for v in 1 0 0 1; do XG_SEG_YS=$v timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seg_ys=$v', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['per_op_ms'])"; done | tee $S/bench_seg_ys.txt
