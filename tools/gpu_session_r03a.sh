#!/bin/bash
# Round 3, session A: the rescue path of the chained kernels on hardware + the whole GPU suite + where this box stands
# on the kernels VERDICT r02 lists (box kind, baseline numbers for the round's A/B work).
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/r03a
mkdir -p $S
export TMPDIR=/tmp
echo "== rescue tests"; timeout 600 python -m pytest tests/test_gpu_chain_rescue.py -x -q 2>&1 | tail -15 | tee $S/pytest_rescue.log
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $S/pytest_gpu.log
echo "== box probe"; bash tools/box_probe.sh 2>&1 | tee $S/box_probe.txt | tail -12
echo "== baseline of the round's targets"
timeout 300 python tools/ab_tunables.py --cases cumZ,cumY,cumYw,sumYw,dY,dX,cumXw,i2mw,vort,diffX,diffY --variants "scan_chain=1" --rounds 3 --reps 5 2>&1 | grep '^{' | tee $S/baseline_targets.jsonl | cut -c1-150
