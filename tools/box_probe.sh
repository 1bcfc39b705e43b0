#!/bin/bash
# What differs between a "fast" and a "slow" box?  Clocks / power / temperature under load next to the rates of a plain and
# a metric-carrying stencil, a scan and the copy ceiling, all in one process.   bash tools/box_probe.sh
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
python - <<'PY' &
import sys, time; sys.path.insert(0, '.')
import torch
from xgcm_amd import device as D
T = D.synthetic((75, 2400, 3600), 2)
dx = D.synthetic((1, 2400, 3600), 31, 0, 1000.0, 1000.0)
def rate(fn, bpc, reps=30):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    return T.numel() * bpc / ms / 1e6 / 8000 * 100
t0 = time.time()
while time.time() - t0 < 6:  # keep the device busy while rocm-smi samples
    D.stencil1d("diff", T, 1, 1, 0, "extend")
torch.cuda.synchronize()
print("rates (pct of 8 TB/s): diffY %.1f  dY(metric) %.1f  cumZ %.1f  cumY %.1f  torch copy %.1f" % (
    rate(lambda: D.stencil1d("diff", T, 1, 1, 0, "extend"), 16), rate(lambda: D.stencil1d("diff", T, 1, 1, 0, "extend", m_out=dx), 16),
    rate(lambda: D.cumsum1d(T, 0, 0, 1, 1, 0, "fill"), 16), rate(lambda: D.cumsum1d(T, 1, 0, 1, 1, 0, "fill"), 16), rate(lambda: T.clone(), 16)))
PY
sleep 4
echo "-- rocm-smi under load"
rocm-smi --showclocks --showpower --showtemp --showperflevel 2>/dev/null | grep -v "^=\|^$\|WARNING" | head -30
rocm-smi --showmaxpower --showmemvendor --showvbios --showmemorypartition --showcomputepartition --showfwinfo 2>/dev/null | grep -v "^=\|^$\|WARNING" | head -60
rocminfo 2>/dev/null | grep -i "Compute Unit\|Max Clock\|L2:\|L3:\|Cacheline\|Size:.*KB" | head -12
wait
