#!/bin/bash
# Round 3, session AT: row-wise gather with 2 / 4 tiles per wave-task: parity on long rows + A/B
S=$PWD/gpurun_out/r03at
mkdir -p $S
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_topology.py tests/test_grid_api.py -x -q -m gpu 2>&1 | tail -3 | tee $S/pytest.log
timeout 400 python tools/ab_tunables.py --cases gatherYX,padX --variants "pad_tpw=1;pad_tpw=2;pad_tpw=4" --rounds 6 --reps 5 2>&1 | grep '^{' | tee $S/ab_gather_tpw.jsonl | cut -c1-160
