#!/bin/bash
# Round 2, session J: final state -- full GPU suite, smoke, bench line, config tables, kernel tables (f64 / f32)
OUT=$PWD/gpurun_out/r02j
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | tee $OUT/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
echo "== bench"; timeout 600 python bench.py 2>&1 | tail -1 | tee $OUT/bench_1gpu.json
echo "== bench through RCCL, one rank"; XG_BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_rccl_ws1.json
echo "== configs"; timeout 900 python tools/bench_configs.py --configs 2,3 2>&1 | grep '^{' | tee $OUT/configs_2_3.jsonl
timeout 900 python tools/bench_configs.py --gpus 1 --configs 4,5 --records 45 2>&1 | grep '^{' | tee $OUT/configs_4_5_sharded_1rank.jsonl
timeout 900 python tools/bench_configs.py --configs 5x,f1 2>&1 | grep '^{' | tee $OUT/configs_5x_f1.jsonl
echo "== kernel tables"
timeout 300 python tools/microbench.py --reps 9 --cases copy,stencil,metric,cumsum,reduce,vort,generic 2>/dev/null | grep '^{' > $OUT/microbench_all_kernels.jsonl
timeout 300 python tools/microbench.py --reps 9 --dtype f32 --cases stencil,metric,cumsum,reduce,vort 2>/dev/null | grep '^{' > $OUT/microbench_f32.jsonl
python - <<'PY'
import json
for f in ("microbench_all_kernels", "microbench_f32"):
    print("--", f)
    for ln in open(f"gpurun_out/r02j/{f}.jsonl"):
        r = json.loads(ln)
        print(f"{r['case']:55s} {r['ms']:7.3f} ms {r['frac_8TBps']*100:5.1f}%")
PY
