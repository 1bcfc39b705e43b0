#!/bin/bash
# Round 3, session W: the generated tables and the GPU suite once more on the round's last build (K4L, gradient bands,
# contig_rw_mi = 4 are in; the X-scan and chain experiments are out)
REPO=$PWD
OUT=$REPO/gpurun_out
S=$OUT/r03w
mkdir -p $S
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | tee $S/pytest_gpu.log
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $S/smoke.log
echo "== bench"; timeout 400 python bench.py 2>&1 | tail -1 | tee $S/bench_1gpu.json | cut -c1-300
echo "== box kind"; bash tools/box_probe.sh 2>&1 | grep rates | tee $S/box_rates.txt
echo "== roofline table"; timeout 900 python tools/roofline_table.py --out $S/roofline 2>&1 | tail -34
timeout 300 python tools/survey.py --reps 7 2>&1 | grep '^{' > $S/survey.jsonl
timeout 300 python tools/bench_configs.py --configs 3 2>&1 | grep '^{' | tee $S/configs_3.jsonl | cut -c1-200
