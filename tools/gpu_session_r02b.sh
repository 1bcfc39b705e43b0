#!/bin/bash
# Round 2, session B: interleaved A/B of the metric-stencil variants and what bounds them
OUT=$PWD/gpurun_out/r02b
mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity of the z-shared K1r variants"
for v in "XG_RW_ZSHARE=1 XG_CONTIG_RW=2" "XG_RW_ZSHARE=1 XG_CONTIG_RW=4 XG_ZB_ROWS=5"; do
  env $v timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_f32.py -m gpu -x -q 2>&1 | tail -2 | tee -a $OUT/pytest_variants.log
done
timeout 300 python -m pytest tests/test_host_logic.py tests/test_gpu_fullsize.py -m gpu -x -q -k "config3 or tunable" 2>&1 | tail -2 | tee -a $OUT/pytest_variants.log
echo "== A: metric kernels, row grouping"
python tools/ab_tunables.py --cases diffX,dX,iXmw,divT --rounds 6 --variants "contig_rw=0;contig_rw=1;contig_rw=2;contig_rw=4;contig_rw=2,rw_zshare=1;contig_rw=4,rw_zshare=1;contig_rw=2,zb_rows=32;contig_rw=2,zb_rows=64;contig_rw=2,rw_zshare=1,zb_rows=32;contig_rw=4,rw_zshare=1,zb_rows=32" 2>&1 | grep '^{' | tee $OUT/ab_contig.jsonl
echo "== B: what bounds K1r (dbg 1: no divisor load; dbg 2: product instead of division; 3: both)"
python tools/ab_tunables.py --cases dX --rounds 6 --variants "contig_rw=2,dbg=0;contig_rw=2,dbg=1;contig_rw=2,dbg=2;contig_rw=2,dbg=3;contig_rw=2,rw_zshare=1,dbg=0;contig_rw=2,rw_zshare=1,dbg=2" 2>&1 | grep '^{' | tee $OUT/ab_bounds.jsonl
echo "== C: strided metric kernels"
python tools/ab_tunables.py --cases diffY,dY,dZ,iYmw --rounds 6 --variants "met_seg=1;met_seg=2;met_seg=4;met_seg=2,zb_rows=32;met_seg=4,zb_rows=32" 2>&1 | grep '^{' | tee $OUT/ab_strided.jsonl
echo "== D: marching scans"
python tools/ab_tunables.py --cases cumY,sumY,cumZ,sumZ --rounds 6 --variants "scan_pipe=0;scan_pipe=1,scan_u=8;scan_pipe=1,scan_u=16;scan_pipe=1,scan_u=24;scan_pipe=1,scan_u=32;scan_pipe=1,scan_u=24,scan_pace=1;scan_pipe=1,scan_u=24,scan_narrow_below=0;scan_pipe=1,scan_u=24,march_band=0" 2>&1 | grep '^{' | tee $OUT/ab_scan.jsonl
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02b/ab_*.jsonl")):
    print("--", f)
    for ln in open(f):
        r = json.loads(ln)
        print(f"{r['case']:6s} {r['median_ms']:7.3f} ms [{r['min_ms']:.3f}-{r['max_ms']:.3f}] {r['frac_8TBps']*100:5.1f}%  {r['variant']}")
PY
