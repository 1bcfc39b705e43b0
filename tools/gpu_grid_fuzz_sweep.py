#!/usr/bin/env python3
"""A long run of tests/test_gpu_grid_fuzz.py's comparison (HIP vs the oracle double) over many seeds, on a GPU box:

    python tools/gpu_grid_fuzz_sweep.py --seeds 8000:8040 --cases 200 > gpurun_out/<name>/gpu_grid_fuzz_sweep.log
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="8000:8010")
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--fused", action="store_true", help="every grid with fuse=True: HIP's fused templates against the double's")
    args = ap.parse_args()
    import test_gpu_grid_fuzz as T
    from oracle import fake_device
    from oracle import fuzz_against_reference as F
    import xgcm_amd.device as dev

    lo, hi = (int(v) for v in args.seeds.split(":"))
    real = {n: getattr(dev, n) for n in fake_device._NAMES}
    total = differences = 0
    for seed in range(lo, hi):
        on_hip = [T._outcomes(seed, case, args.fused) for case in range(args.cases)]
        fake_device.install(F._MP())
        try:
            on_double = [T._outcomes(seed, case, args.fused) for case in range(args.cases)]
        finally:
            for n, f in real.items():
                setattr(dev, n, f)
        n = bad = 0
        for case, (hs, ds) in enumerate(zip(on_hip, on_double)):
            for (what, got, got_exc), (_, ref, ref_exc) in zip(hs, ds):
                diff = F.compare(ref, ref_exc, got, got_exc)
                n += 1
                if diff is not None:
                    bad += 1
                    print(json.dumps({"seed": seed, "case": case, "call": what[:200], "difference": diff[:200]}))
        print(json.dumps({"seed": seed, "calls": n, "differences": bad}))
        total, differences = total + n, differences + bad
    print(json.dumps({"seeds": [lo, hi], "cases_per_seed": args.cases, "fused": bool(args.fused), "calls": total, "differences": differences}))
    sys.exit(1 if differences else 0)


if __name__ == "__main__":
    main()
