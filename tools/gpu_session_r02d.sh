#!/bin/bash
# Round 2, session D: new defaults (z-shared metric rows everywhere), fused vorticity / divergence with shared area rows
OUT=$PWD/gpurun_out/r02d
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu (new defaults)"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest_gpu.log
for v in "XG_VEC_ZK=4 XG_MET_ZK=8 XG_MET_SEG=2" "XG_VEC_ZK=1 XG_MET_ZK=4 XG_MET_SEG=4"; do
  env $v timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_f32.py tests/test_topology.py -m gpu -x -q 2>&1 | tail -1 | tee -a $OUT/pytest_variants.log
done
XG_VEC_ZK=4 timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config5" 2>&1 | tail -1 | tee -a $OUT/pytest_variants.log
echo "== A: fused vorticity (75 x 2400 x 3600) and the strided metric kernels with deeper sharing"
python tools/ab_tunables.py --cases vort,iYmw,dY --rounds 6 --variants "vec_zk=1,met_zk=4,met_seg=2;vec_zk=2,met_zk=8,met_seg=2;vec_zk=4,met_zk=4,met_seg=4;vec_zk=2,zb_rows=32,met_zk=4,met_seg=2;vec_zk=4,zb_rows=32,met_zk=8,met_seg=2" 2>&1 | grep '^{' | tee $OUT/ab_vort.jsonl
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02d/ab_*.jsonl")):
    print("--", f)
    for ln in open(f):
        r = json.loads(ln)
        print(f"{r['case']:6s} {r['median_ms']:7.3f} ms [{r['min_ms']:.3f}-{r['max_ms']:.3f}] {r['frac_8TBps']*100:5.1f}%  {r['variant']}")
PY
echo "== config 5 (4320 x 4320 x 90) and config 3"
for zk in 1 2 4; do XG_VEC_ZK=$zk timeout 600 python tools/bench_configs.py --gpus 1 --configs 5 2>&1 | grep '^{' | grep fused | head -1 | tee -a $OUT/config5_zk.jsonl; done
timeout 600 python tools/bench_configs.py --configs 3 2>&1 | grep '^{' | tee $OUT/config3.jsonl
