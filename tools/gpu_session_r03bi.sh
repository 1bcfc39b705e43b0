#!/bin/bash
# Round 3, session BI: K2Sm with 32-row bands (one metric): parity + A/B (zb_rows = 8 -> 16-row bands, 16 -> 32-row bands)
S=$PWD/gpurun_out/r03bi
mkdir -p $S
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -1 | tee $S/pytest.log
timeout 400 python tools/ab_tunables.py --cases dY --variants "zb_rows=8;zb_rows=16" --rounds 8 --reps 5 2>&1 | grep '^{' | tee $S/ab_k2sm_band.jsonl | cut -c1-150
