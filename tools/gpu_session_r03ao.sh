#!/bin/bash
# Round 3, session AO: float32 marches with 4-byte lanes when 8-byte lanes leave too few wave-tasks
S=$PWD/gpurun_out/r03ao
mkdir -p $S
export TMPDIR=/tmp
echo "== parity"; timeout 1200 python -m pytest tests/test_f32.py tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2 | tee -a $S/pytest.log
timeout 400 python tools/ab_tunables.py --dtype f32 --cases sumY,sumYw,avgYw,cumY,sumZ --variants "scan_narrow4=0;scan_narrow4=1" --rounds 6 --reps 5 2>&1 | grep '^{' | tee $S/ab_f32_narrow4.jsonl | cut -c1-160
timeout 400 python tools/ab_tunables.py --dtype f32 --cases sumY,cumY --variants "scan_chain=0,scan_narrow4=0;scan_chain=0,scan_narrow4=1" --rounds 4 --reps 5 2>&1 | grep '^{' | tee -a $S/ab_f32_narrow4.jsonl | cut -c1-160
