#!/bin/bash
# Round 2, session A: parity of everything new, tunable sweeps for the metric stencils and the marching scans,
# the N-rank launcher, sharded configs 4 / 5 on one GPU.  Usage on the GPU box: bash tools/gpu_session_r02a.sh
OUT=$PWD/gpurun_out/r02a
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu (default tunables)"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.log
echo "== parity of the non-default kernel variants"
for v in "XG_CONTIG_RW=1 XG_MET_SEG=1 XG_SCAN_PIPE=0" "XG_CONTIG_RW=4 XG_MET_SEG=4 XG_SCAN_PIPE=2 XG_SCAN_U=32" "XG_CONTIG_RW=0 XG_MET_SEG=2 XG_SCAN_PIPE=1 XG_SCAN_U=8 XG_SCAN_PACE=1" "XG_SCAN_PIPE=1 XG_SCAN_U=24 XG_SCAN_NARROW_BELOW=0"; do
  echo "-- $v"
  env $v timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_f32.py -m gpu -x -q 2>&1 | tail -3 | tee -a $OUT/pytest_variants.log
done
echo "== sweeps"
bash tools/sweep_r02.sh met > $OUT/sweep_met.jsonl 2>&1
bash tools/sweep_r02.sh scan > $OUT/sweep_scan.jsonl 2>&1
python - <<'PY'
import json
for part in ("met", "scan"):
    print("--", part)
    for ln in open(f"gpurun_out/r02a/sweep_{part}.jsonl"):
        if not ln.startswith("{"):
            continue
        r = json.loads(ln)
        tag = " ".join(f"{k[3:]}={v}" for k, v in r.items() if k.startswith("XG_"))
        print(f"{r['case']:32s} {r['ms']:7.3f} ms {r['frac_8TBps']*100:5.1f}%  {tag}")
PY
echo "== bench.py"
timeout 600 python bench.py 2>&1 | tail -1 | tee $OUT/bench_1gpu.json
echo "== sharded configs 4 (45 records = one GPU's share of the 8-GPU run) and 5 on one rank"
timeout 900 python tools/bench_configs.py --gpus 1 --configs 4,5 --records 45 2>&1 | grep '^{' | tee $OUT/configs45_1gpu.jsonl
echo "== config 3 table"
timeout 600 python tools/bench_configs.py --configs 3 2>&1 | grep '^{' | tee $OUT/config3.jsonl
