"""Helper of tests/test_gpu_multi.py (not a test): a field split ALONG the operator's axis over N ranks on the real HIP
library -- `sharding.exchange_halo` (edge planes point to point), `stencil_along_sharded_axis` (halo-mode kernel),
`cumsum_along_sharded_axis` (all-gather of block totals) -- through the same launcher as the benches.

    python tests/_sharded_axis_gpu_driver.py --gpus N --out DIR        # backend RCCL (one GPU per rank)
    XG_DIST_BACKEND=gloo XG_SHARE_GPU=1 python ... --gpus 2 --out DIR   # two ranks computing on ONE GPU, transport gloo

Every rank writes its blocks of the results to DIR; the test assembles them and compares with the one-process result."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NZ, NY, NX = 37, 24, 128   # 37 levels: uneven blocks (19 + 18, 10 + 9 + 9 + 9 ...)


def build_grid():
    from xgcm_amd import Dataset, Grid

    ds = Dataset(coords={"Z": ("Z", np.arange(NZ) + 0.5), "Zl": ("Zl", np.arange(NZ) * 1.0), "Zr": ("Zr", np.arange(NZ) + 1.0)})
    return Grid(ds, coords={"Z": {"center": "Z", "left": "Zl", "right": "Zr"}}, padding="fill", autoparse_metadata=False)


CASES = [("diff", "center", "left", "periodic", 0.0), ("interp", "center", "left", "fill", 2.5),
         ("diff", "center", "right", "extend", 0.0), ("interp", "left", "center", "periodic", 0.0),
         ("max", "center", "left", "fill", -1.0), ("min", "right", "center", "extend", 0.0)]
SCANS = [("left", "fill", 0.0), ("right", "fill", 0.0), ("left", "extend", 0.0)]


def field(seed):
    from oracle import refimpl as R

    a = R.synthetic_field((NZ, NY, NX), seed)
    a[5, 3, 7] = np.nan
    return a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    from xgcm_amd import sharding as S

    S.ensure_ranks(a.gpus, os.path.abspath(__file__), sys.argv[1:])
    ranks = S.init_ranks(a.gpus)
    import torch

    from xgcm_amd import DataArray
    from xgcm_amd import device as D

    try:
        grid = build_grid()
        dist = ranks.dist
        lo, hi = S.shard_bounds(NZ, ranks.world, ranks.rank)
        dims_of = {"center": "Z", "left": "Zl", "right": "Zr"}
        full = field(77)
        for k, (fn, frm, to, bc, fill) in enumerate(CASES):
            mine = DataArray(D.asdevice(full[lo:hi]), (dims_of[frm], "Y", "X"))
            res = S.stencil_along_sharded_axis(grid, fn, mine, "Z", dist=dist, to=to, padding=bc, fill_value=fill)
            assert isinstance(res.data, torch.Tensor) and res.data.is_cuda
            np.save(os.path.join(a.out, f"stencil_{k}_{ranks.rank}.npy"), res.values)
        for k, (to, bc, fill) in enumerate(SCANS):
            mine = DataArray(D.asdevice(full[lo:hi]), ("Z", "Y", "X"))
            res = S.cumsum_along_sharded_axis(grid, mine, "Z", dist=dist, to=to, padding=bc, fill_value=fill)
            np.save(os.path.join(a.out, f"scan_{k}_{ranks.rank}.npy"), res.values)
        ranks.barrier()
        if ranks.rank == 0:
            import json

            print(json.dumps({"world": ranks.world, "backend": ranks.backend, "devices": torch.cuda.device_count()}), flush=True)
    finally:
        ranks.close()


if __name__ == "__main__":
    main()
