#!/bin/bash
# Round 2, session H: full GPU suite on the current tree, one-pass average, f32 metric kernels, block-iterator streaming
OUT=$PWD/gpurun_out/r02h
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest_gpu.log
echo "== config 3 (average in one pass)"
timeout 600 python tools/bench_configs.py --configs 3 2>&1 | grep '^{' | tee $OUT/config3.jsonl
echo "== f32 kernel table"
timeout 300 python tools/microbench.py --reps 9 --dtype f32 --cases stencil,metric,cumsum,reduce,vort 2>/dev/null | grep '^{' > $OUT/microbench_f32.jsonl
python - <<'PY'
import json
for ln in open("gpurun_out/r02h/microbench_f32.jsonl"):
    r = json.loads(ln)
    print(f"{r['case']:55s} {r['ms']:7.3f} ms {r['frac_8TBps']*100:5.1f}%")
PY
echo "== host-record streaming"
timeout 600 python tools/bench_configs.py --configs stream 2>&1 | grep '^{' | tee $OUT/stream.jsonl
