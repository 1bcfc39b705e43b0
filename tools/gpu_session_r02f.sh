#!/bin/bash
# Round 2, session F: vertical transform kernels (slope hoisted out of the divergent loop; sliding accumulator window)
OUT=$PWD/gpurun_out/r02f
mkdir -p $OUT
export TMPDIR=/tmp
echo "== transform parity (both settings of the new paths)"
timeout 900 python -m pytest tests/test_transform.py tests/test_gpu_fullsize.py tests/test_xarray_surface.py tests/test_streaming.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest.log
XG_TRANSFORM_WIN=0 XG_DBG=4 timeout 900 python -m pytest tests/test_transform.py -m gpu -x -q 2>&1 | tail -1 | tee -a $OUT/pytest.log
echo "== A/B"
python tools/ab_tunables.py --cases tlin_rw,tlin_sm,tcon_rw,tcon_sm --rounds 5 --reps 3 --variants "dbg=4,transform_win=0;dbg=0,transform_win=1;dbg=0,transform_win=0,transform_lds_kb=0" 2>&1 | grep '^{' | tee $OUT/ab_transform.jsonl
python - <<'PY'
import json
for ln in open("gpurun_out/r02f/ab_transform.jsonl"):
    r = json.loads(ln)
    print(f"{r['case']:8s} {r['median_ms']:7.3f} ms [{r['min_ms']:.3f}-{r['max_ms']:.3f}] {r['frac_8TBps']*100:5.1f}%  {r['variant']}")
PY
echo "== sharded config 4 with the output block pre-warmed (45 records = one GPU's share)"
timeout 900 python tools/bench_configs.py --gpus 1 --configs 4 --records 45 2>&1 | grep '^{' | tee $OUT/config4_45records.jsonl
