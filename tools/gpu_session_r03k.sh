#!/bin/bash
# Round 3, session K: readers + byte swap on the GPU, the shifted-aligned X scan (parity, A/B), the box-kind sample,
# the generated roofline table and the one-command scaling table (N = 1 here).
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/r03k
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_streaming.py tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_f32.py -x -q 2>&1 | tail -4 | tee $S/pytest.log
echo "== X scan A/B"
timeout 300 python tools/ab_tunables.py --cases cumX,cumXw,cumXp --variants "scan_vec=1" --rounds 3 --reps 5 2>&1 | grep '^{' | tee $S/ab_cumX.jsonl | cut -c1-170
echo "== box kind"; bash tools/box_kind_pmc.sh $S/box_kind.txt > /dev/null 2>&1; head -3 $S/box_kind.txt | cut -c1-200
echo "== roofline table"; timeout 900 python tools/roofline_table.py --out $S/roofline 2>&1 | tail -32
echo "== scale table"; timeout 600 python tools/scale_table.py --records 16 --out $S/scale_table 2>&1 | tail -12
