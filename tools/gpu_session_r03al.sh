#!/bin/bash
# Round 3, session AL: contiguous-axis reductions specialised for the plain sums
S=$PWD/gpurun_out/r03al
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_f32.py tests/test_grid_api.py -x -q -m gpu 2>&1 | tail -2 | tee -a $S/pytest.log
timeout 400 python tools/ab_tunables.py --cases sumX,sumXw --variants "reduce_ru=0;reduce_ru=1;reduce_ru=2" --rounds 6 --reps 5 2>&1 | grep '^{' | tee $S/ab_reduce_ru8.jsonl | cut -c1-150
