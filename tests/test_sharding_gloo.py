"""N>1 path on CPU: two `gloo` ranks shard a record axis, run their block through the operator
stack (oracle-backed device double -- no GPU here), and the sharded outputs + reduced scalars
must equal the single-process result.  Mirrors what bench.py does with RCCL on the GPU box."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from xgcm_amd.sharding import shard_bounds


def test_shard_bounds_cover_exactly():
    assert [shard_bounds(90, 8, r) for r in range(8)] == [(0, 12), (12, 24), (24, 35), (35, 46), (46, 57), (57, 68),
                                                           (68, 79), (79, 90)]
    assert [shard_bounds(360, 8, r)[1] - shard_bounds(360, 8, r)[0] for r in range(8)] == [45] * 8
    for n in (0, 1, 7, 8, 9, 75):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import fake_device
        from oracle import refimpl as R

        class MP:  # minimal monkeypatch stand-in
            def setattr(self, obj, name, val):
                setattr(obj, name, val)

        fake_device.install(MP())
        from xgcm_amd import DataArray, Dataset, Grid
        from xgcm_amd.sharding import whole_job_throughput

        nt, nz, ny, nx = 5, 6, 8, 16
        full = R.synthetic_field((nt, nz, ny, nx), 4)
        lo, hi = shard_bounds(nt, world, rank)
        ds = Dataset(coords={"XC": ("XC", np.arange(nx) * 1.0), "XG": ("XG", np.arange(nx) - 0.5),
                             "Z": ("Z", np.arange(nz) * 1.0), "Zl": ("Zl", np.arange(nz) - 0.5)})
        grid = Grid(ds, coords={"X": {"center": "XC", "left": "XG"}, "Z": {"center": "Z", "left": "Zl"}},
                    padding={"X": "periodic", "Z": "fill"}, autoparse_metadata=False)
        mine = DataArray(full[lo:hi], ("time", "Z", "YC", "XC"))
        d = grid.diff(mine, "X").values
        c = grid.cumsum(mine, "Z").values
        np.save(os.path.join(tmp, f"diff_{rank}.npy"), d)
        np.save(os.path.join(tmp, f"cumsum_{rank}.npy"), c)
        # the same through the batch loop of the product: two records per resident batch
        from xgcm_amd.sharding import Ranks, map_record_batches

        parts = map_record_batches(lambda blk: grid.cumsum(DataArray(blk, ("time", "Z", "YC", "XC")), "Z").values,
                                   lambda a, b: full[a:b], nt, Ranks(rank, world, rank, "gloo", dist), full[0].nbytes * 2, per_batch=2)
        assert [p[:2] for p in parts][0][0] == lo and parts[-1][1] == hi and all(p[1] - p[0] <= 2 for p in parts)
        assert np.array_equal(np.concatenate([p[2] for p in parts], axis=0), c)
        # scalar reductions only: checksum (order-independent integer sum) and the timing aggregate
        chk = torch.tensor([int(np.frombuffer(d.tobytes(), dtype=np.uint64).sum() % (1 << 50))], dtype=torch.int64)
        dist.all_reduce(chk, op=dist.ReduceOp.SUM)
        rate, tmax = whole_job_throughput(float(d.size), 1.0 + rank, dist)
        if rank == 0:
            np.save(os.path.join(tmp, "scalars.npy"), np.array([float(chk.item()), rate, tmax]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharded_equals_single_process(tmp_path):
    from oracle import refimpl as R

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    nt, nz, ny, nx = 5, 6, 8, 16
    full = R.synthetic_field((nt, nz, ny, nx), 4)
    want_d = R.stencil1d("diff", full, 3, 1, 0, "periodic")
    want_c = R.grid_cumsum(full, 1, "center", "left", "fill")
    got_d = np.concatenate([np.load(tmp_path / f"diff_{r}.npy") for r in range(world)], axis=0)
    got_c = np.concatenate([np.load(tmp_path / f"cumsum_{r}.npy") for r in range(world)], axis=0)
    assert np.array_equal(got_d, want_d) and np.array_equal(got_c, want_c)
    chk, rate, tmax = np.load(tmp_path / "scalars.npy")
    parts = [np.load(tmp_path / f"diff_{r}.npy") for r in range(world)]
    want_chk = sum(int(np.frombuffer(p.tobytes(), dtype=np.uint64).sum() % (1 << 50)) for p in parts)
    assert int(chk) == want_chk  # checksum of checksums
    assert tmax == 2.0 and rate == want_d.size / 2.0  # all units / max-over-ranks time


# ---- sharding ALONG the operator's axis: neighbour planes exchanged point to point ---------------
def _core_axis_worker(rank, world, port, tmp, real_device=False):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import fake_device
        from oracle import refimpl as R

        class MP:
            def setattr(self, obj, name, val):
                setattr(obj, name, val)

        if real_device:  # GPU box: both ranks compute on the one GPU through the real library, host arrays in and out
            torch.cuda.set_device(0)
        else:
            fake_device.install(MP())
        from xgcm_amd import DataArray, Dataset, Grid
        from xgcm_amd.sharding import cumsum_along_sharded_axis, stencil_along_sharded_axis

        nz, ny, nx = 7, 5, 8
        full = R.synthetic_field((nz, ny, nx), 9)
        lo, hi = shard_bounds(nz, world, rank)
        ds = Dataset(coords={"Z": ("Z", np.arange(nz) * 1.0), "Zl": ("Zl", np.arange(nz) - 0.5),
                             "Zr": ("Zr", np.arange(nz) + 0.5)})
        for bc in ("periodic", "fill", "extend"):
            grid = Grid(ds, coords={"Z": {"center": "Z", "left": "Zl", "right": "Zr"}}, padding={"Z": bc},
                        fill_value={"Z": 1.5}, autoparse_metadata=False)
            mine = DataArray(full[lo:hi], ("Z", "YC", "XC"))
            for fn in ("diff", "interp", "max"):
                for to in ("left", "right"):
                    res = stencil_along_sharded_axis(grid, fn, mine, "Z", dist=dist, to=to)
                    assert res.dims == ("Zl" if to == "left" else "Zr", "YC", "XC")
                    np.save(os.path.join(tmp, f"{bc}_{fn}_{to}_{rank}.npy"), res.values)
            back = stencil_along_sharded_axis(grid, "interp", DataArray(full[lo:hi], ("Zl", "YC", "XC")), "Z", dist=dist)
            np.save(os.path.join(tmp, f"{bc}_back_{rank}.npy"), back.values)
            if bc != "periodic":
                for to in ("left", "right"):
                    c = cumsum_along_sharded_axis(grid, mine, "Z", dist=dist, to=to)
                    assert c.dims == ("Zl" if to == "left" else "Zr", "YC", "XC")
                    np.save(os.path.join(tmp, f"{bc}_cumsum_{to}_{rank}.npy"), c.values)
        with pytest.raises(NotImplementedError, match="cumsum_along_sharded_axis"):
            stencil_along_sharded_axis(grid, "cumsum", mine, "Z", dist=dist)
        pgrid = Grid(ds, coords={"Z": {"center": "Z", "left": "Zl", "right": "Zr"}}, padding={"Z": "periodic"},
                     autoparse_metadata=False)
        with pytest.raises(NotImplementedError, match="periodic halo"):
            cumsum_along_sharded_axis(pgrid, mine, "Z", dist=dist, to="left")
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,real_device", [(2, False), (3, False), pytest.param(2, True, marks=pytest.mark.gpu)])
def test_gloo_ranks_sharded_along_the_core_axis_exchange_one_plane(tmp_path, world, real_device):
    """diff / interp / max along Z of a field split over Z: each rank gets one plane from its neighbour
    (ring closed for `periodic`, ends made locally for `fill` / `extend`); concatenated blocks ==
    the single-process operator, bit for bit.  On the GPU box the same two ranks run the real library with
    HOST (numpy) shards (the block totals of the scan are HBM tensors there)."""
    from oracle import refimpl as R

    mp.spawn(_core_axis_worker, args=(world, _free_port(), str(tmp_path), real_device), nprocs=world, join=True)
    nz, ny, nx = 7, 5, 8
    full = R.synthetic_field((nz, ny, nx), 9)
    for bc in ("periodic", "fill", "extend"):
        for fn in ("diff", "interp", "max"):
            for to, pad in (("left", (1, 0)), ("right", (0, 1))):
                got = np.concatenate([np.load(tmp_path / f"{bc}_{fn}_{to}_{r}.npy") for r in range(world)], axis=0)
                assert np.array_equal(got, R.stencil1d(fn, full, 0, pad[0], pad[1], bc, 1.5)), (bc, fn, to)
        got = np.concatenate([np.load(tmp_path / f"{bc}_back_{r}.npy") for r in range(world)], axis=0)
        assert np.array_equal(got, R.stencil1d("interp", full, 0, 0, 1, bc, 1.5))  # left -> center
        if bc != "periodic":  # scans: block totals all-gathered, carry added in rank order (re-associated sum)
            for to in ("left", "right"):
                got = np.concatenate([np.load(tmp_path / f"{bc}_cumsum_{to}_{r}.npy") for r in range(world)], axis=0)
                np.testing.assert_allclose(got, R.grid_cumsum(full, 0, "center", to, bc, 1.5), rtol=1e-12, atol=1e-13)
