#!/bin/bash
# Round 3, session Q: contiguous-axis scan with two vectors per thread and pass: parity + A/B
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/r03q
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_f32.py tests/test_grid_api.py tests/test_reference_suite_more.py tests/test_reference_suite_positions.py -x -q -m gpu 2>&1 | tail -4 | tee $S/pytest.log
timeout 300 python tools/ab_tunables.py --cases cumX,cumXw,cumXp,cumXe,cumX0 --variants "scan_gv=1;scan_gv=2;scan_gv=2,scan_block=512;scan_gv=2,scan_block=128;scan_gv=1,scan_block=512" --rounds 5 --reps 5 2>&1 | grep '^{' | tee $S/ab_scan_gv.jsonl | cut -c1-150
