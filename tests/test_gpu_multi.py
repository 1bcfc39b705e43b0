"""The one data-path exchange the hot path can have -- a field split ALONG the operator's axis (the analogue of the
reference's `map_overlap(depth=padding_width)`, xgcm/grid_ufunc.py:1045-1125; SURVEY §8(e) "if the split axis were the core
axis") -- on the real library with more than one rank: `sharding.exchange_halo`, `stencil_along_sharded_axis`,
`cumsum_along_sharded_axis`.

* over RCCL with 2 (and 4, 8) ranks, one GPU each: runs the moment a box has that many GPUs, skips cleanly on one;
* on ONE GPU with 2 and 3 ranks sharing it (transport gloo): the device-side code of the exchange -- edge planes laid out
  in HBM, the halo-mode kernel, block totals and carries in HBM -- with a real neighbour, on every GPU box.

Stencils must equal the one-process result bit for bit; the scan's block carry re-associates the sum (1e-12)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "_sharded_axis_gpu_driver.py")


def _gpus():
    try:
        import torch

        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


def _run(n, out, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "XG_DIST_BACKEND", "XG_SHARE_GPU"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, DRIVER, "--gpus", str(n), "--out", str(out)], capture_output=True, text=True, env=env,
                       timeout=600, cwd=ROOT)
    return p, [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]


def _check(out, world):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _sharded_axis_gpu_driver as drv
    from xgcm_amd import DataArray

    grid = drv.build_grid()
    full = drv.field(77)
    dims_of = {"center": "Z", "left": "Zl", "right": "Zr"}
    for k, (fn, frm, to, bc, fill) in enumerate(drv.CASES):
        want = getattr(grid, fn)(DataArray(full, (dims_of[frm], "Y", "X")), "Z", to=to, padding=bc, fill_value=fill).values
        got = np.concatenate([np.load(os.path.join(out, f"stencil_{k}_{r}.npy")) for r in range(world)], axis=0)
        assert np.array_equal(got, want, equal_nan=True), (fn, frm, to, bc)
    for k, (to, bc, fill) in enumerate(drv.SCANS):
        want = grid.cumsum(DataArray(full, ("Z", "Y", "X")), "Z", to=to, padding=bc, fill_value=fill).values
        got = np.concatenate([np.load(os.path.join(out, f"scan_{k}_{r}.npy")) for r in range(world)], axis=0)
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
        r0 = np.load(os.path.join(out, f"scan_{k}_0.npy"))
        assert np.array_equal(r0, want[:r0.shape[0]], equal_nan=True)   # the first block carries nothing: bit for bit


@pytest.mark.parametrize("world", [2, 4, 8])
def test_exchange_over_rccl_one_gpu_per_rank(world, tmp_path):
    if _gpus() < world:
        pytest.skip(f"{_gpus()} GPU(s) visible, {world} ranks over RCCL need {world}")
    p, lines = _run(world, tmp_path)
    assert p.returncode == 0, p.stderr[-3000:]
    assert lines and lines[-1]["world"] == world and lines[-1]["backend"] == "nccl"
    _check(tmp_path, world)


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_between_ranks_sharing_one_gpu(world, tmp_path):
    if _gpus() < 1:
        pytest.skip("no GPU")
    p, lines = _run(world, tmp_path, {"XG_DIST_BACKEND": "gloo", "XG_SHARE_GPU": "1"})
    assert p.returncode == 0, p.stderr[-3000:]
    assert lines and lines[-1]["world"] == world and lines[-1]["backend"] == "gloo"
    _check(tmp_path, world)
