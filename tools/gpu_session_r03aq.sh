#!/bin/bash
# Round 3, session AQ: flat K1: neighbour by DPP + SCALAR load for the wave's boundary lane (one vector-memory instruction per wave)
S=$PWD/gpurun_out/r03aq
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_f32.py tests/test_grid_api.py tests/test_topology.py -x -q -m gpu 2>&1 | tail -2 | tee -a $S/pytest.log
timeout 400 python tools/ab_tunables.py --cases diffX,diffY --variants "nb_dpp=0;nb_dpp=1" --rounds 8 --reps 7 2>&1 | grep '^{' | tee $S/ab_k1_sdpp.jsonl | cut -c1-160
for v in 0 1 1 0; do XG_NB_DPP=$v timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nb_dpp=$v', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['per_op_ms'])"; done | tee $S/bench_nb_dpp.txt
timeout 300 python tools/pmc_ab.py --cases diffX --variants "nb_dpp=0;nb_dpp=1" --pmc "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" 2>&1 | tee $S/pmc_k1_sdpp.jsonl | cut -c1-330
