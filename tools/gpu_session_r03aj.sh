#!/bin/bash
# Round 3, session AJ: contiguous-axis reductions with 4 loads in flight; elementwise binary op without 64-bit divisions
S=$PWD/gpurun_out/r03aj
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_f32.py tests/test_grid_api.py -x -q -m gpu 2>&1 | tail -2 | tee -a $S/pytest.log
timeout 400 python tools/ab_tunables.py --cases sumX,sumXw --variants "reduce_ru=0;reduce_ru=1" --rounds 6 --reps 5 2>&1 | grep '^{' | tee $S/ab_reduce_ru.jsonl | cut -c1-150
timeout 400 python tools/ab_tunables.py --cases divT,mulTT --variants "bin_idx32=0;bin_idx32=1" --rounds 6 --reps 5 2>&1 | grep '^{' | tee $S/ab_bin_idx32.jsonl | cut -c1-150
