#!/bin/bash
# Round 2, session C: z-shared metric rows in the strided-axis kernel, R = 8 in K1r
OUT=$PWD/gpurun_out/r02c
mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity of the z-shared variants"
for v in "XG_MET_ZK=2 XG_MET_SEG=2 XG_RW_ZSHARE=1 XG_CONTIG_RW=8" "XG_MET_ZK=4 XG_MET_SEG=1 XG_ZB_ROWS=5" "XG_MET_ZK=2 XG_MET_SEG=4 XG_ZB_ROWS=32" "XG_MET_ZK=4 XG_MET_SEG=2" "XG_MET_ZK=2 XG_MET_SEG=1"; do
  env $v timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_f32.py tests/test_topology.py -m gpu -x -q 2>&1 | tail -2 | tee -a $OUT/pytest_variants.log
done
XG_MET_ZK=2 XG_MET_SEG=2 XG_RW_ZSHARE=1 XG_CONTIG_RW=4 timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config3" 2>&1 | tail -2 | tee -a $OUT/pytest_variants.log
echo "== A: K1r"
python tools/ab_tunables.py --cases dX,iXmw --rounds 6 --variants "contig_rw=2,rw_zshare=1;contig_rw=4,rw_zshare=1;contig_rw=8,rw_zshare=1;contig_rw=2,rw_zshare=1,zb_rows=32;contig_rw=8,rw_zshare=1,zb_rows=32;contig_rw=4,rw_zshare=1,zb_rows=8" 2>&1 | grep '^{' | tee $OUT/ab_contig.jsonl
echo "== B: K2S"
python tools/ab_tunables.py --cases dY,iYmw,dZ --rounds 6 --variants "met_seg=2,met_zk=1;met_seg=4,met_zk=1;met_seg=1,met_zk=2;met_seg=2,met_zk=2;met_seg=4,met_zk=2;met_seg=1,met_zk=4;met_seg=2,met_zk=4;met_seg=2,met_zk=2,zb_rows=32;met_seg=2,met_zk=4,zb_rows=32;met_seg=4,met_zk=2,zb_rows=32" 2>&1 | grep '^{' | tee $OUT/ab_strided.jsonl
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02c/ab_*.jsonl")):
    print("--", f)
    for ln in open(f):
        r = json.loads(ln)
        print(f"{r['case']:6s} {r['median_ms']:7.3f} ms [{r['min_ms']:.3f}-{r['max_ms']:.3f}] {r['frac_8TBps']*100:5.1f}%  {r['variant']}")
PY
