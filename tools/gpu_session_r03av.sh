#!/bin/bash
# Round 3, session AV: K1r with z-stacked workgroups (metric vectors through LDS): parity + A/B
S=$PWD/gpurun_out/r03av
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_f32.py tests/test_grid_api.py tests/test_topology.py -x -q -m gpu 2>&1 | tail -2 | tee -a $S/pytest.log
timeout 400 python tools/ab_tunables.py --cases dX,iXmw --variants "rw_zw=0;rw_zw=1" --rounds 8 --reps 5 2>&1 | grep '^{' | tee $S/ab_k1r_zw.jsonl | cut -c1-160
timeout 300 python tools/pmc_ab.py --cases dX,iXmw --variants "rw_zw=0;rw_zw=1" --pmc "FETCH_SIZE|TCP_TCC_READ_REQ_sum" 2>&1 | tee $S/pmc_k1r_zw.jsonl | cut -c1-330
