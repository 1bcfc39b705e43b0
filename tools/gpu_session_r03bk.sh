#!/bin/bash
# Round 3, session BK: K5p, the marching scan as loader / storer wave pairs: parity (forced for every shape with 16-B lanes) + A/B
S=$PWD/gpurun_out/r03bk
mkdir -p $S
export TMPDIR=/tmp
for v in 1 2 3; do echo "== parity scan_split=$v, forced"; XG_SCAN_SPLIT=$v XG_SCAN_NARROW_BELOW=0 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_f32.py tests/test_grid_api.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -1 | tee -a $S/pytest.log; done
echo "== parity defaults"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_chain_rescue.py -x -q -m gpu 2>&1 | tail -1 | tee -a $S/pytest.log
timeout 400 python tools/ab_tunables.py --cases cumZ,cumZr4 --variants "scan_split=0;scan_split=1;scan_split=2;scan_split=3" --rounds 6 --reps 5 2>&1 | grep '^{' | tee $S/ab_k5p.jsonl | cut -c1-150
