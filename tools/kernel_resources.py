#!/usr/bin/env python3
"""VGPR / SGPR / scratch / occupancy of every kernel of one unit, from hipcc's -Rpass-analysis=kernel-resource-usage.

    python tools/kernel_resources.py xg_scan [-DXG_F32] [filter]

Build container only (cross-compiles for gfx950; no GPU needed)."""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
unit = sys.argv[1]
extra = [a for a in sys.argv[2:] if a.startswith("-D")]
filt = [a for a in sys.argv[2:] if not a.startswith("-D")]
src = os.path.join(REPO, "xgcm_amd", "csrc", unit + ".hip")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-fPIC",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
txt = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
names = sorted(set(re.findall(r"Function Name: (\S+)", txt)))
dem = dict(zip(names, subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines())) if names else {}
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip()

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1

    nm = dem.get(name, name).replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if filt and not any(f in nm for f in filt):
        continue
    scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    print(f"{nm[:90]:90s} vgpr={g('VGPRs'):4d} agpr={g('AGPRs'):3d} sgpr={g('SGPRs'):4d} scratch={scratch:4d} occ={occ} lds={lds}")
