"""`--gpus N` must mean N ranks: the launcher of bench.py / tools/bench_configs.py
(`xgcm_amd.sharding.ensure_ranks` + `init_ranks`) and the sharded config-4 / config-5 drivers, on CPU with
gloo and the oracle-backed device double.  The GPU box repeats the 2-rank run on the real library
(tests/test_gpu_sharded.py)."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "_sharded_cpu_driver.py")


def _run(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update({"XG_DIST_BACKEND": "gloo", "OMP_NUM_THREADS": "1"})
    env.update(env_extra or {})
    p = subprocess.run([sys.executable] + args, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


def _by_op(lines):
    return {(ln["config"], ln["op"].split(";")[0].split(",")[0] + ("outer" if "outer" in ln["op"] else "")): ln
            for ln in lines if "op" in ln}


def test_two_gloo_ranks_through_the_launcher_equal_one_rank():
    p1, one = _run([DRIVER, "--gpus", "1"])
    assert p1.returncode == 0, p1.stderr[-2000:]
    p2, two = _run([DRIVER, "--gpus", "2"])
    assert p2.returncode == 0, p2.stderr[-2000:]
    assert "torch.distributed.run" in p2.stderr  # the launcher re-executed itself with 2 ranks
    a, b = _by_op(one), _by_op(two)
    assert a.keys() == b.keys() and len(a) == 5  # config 4: two operators; config 5: fused, chain, chain as written under grid.fused()
    for k in a:
        assert a[k]["n_gpus"] == 1 and b[k]["n_gpus"] == 2 and b[k]["backend"] == "gloo"
        assert a[k]["cells"] == b[k]["cells"] and sum(b[k]["per_rank_cells"]) == a[k]["cells"]
        assert a[k]["checksum_u64"] == b[k]["checksum_u64"], k  # checksum of checksums == single-process checksum
        assert len(b[k]["per_rank_device_ms"]) == 2
    c4 = next(v for k, v in b.items() if k[0] == 4)
    assert c4["records_per_rank"] == [4, 3] and c4["records_per_resident_batch"] == 2 and c4["batch_rounds"] == 2
    c5 = next(v for k, v in b.items() if k[0] == 5)
    assert c5["levels_per_rank"] == [3, 2]
    checks = [ln for ln in two if "check" in ln]
    assert checks and all(ln["ok"] for ln in checks)


def test_eight_gloo_ranks_at_the_baseline_splits():
    """8 ranks as on the node the north star names: config 4's 360 records fall 45 per rank, config 5's 90 levels
    12,12,11,11,11,11,11,11; every checksum of checksums equals the single-rank run's."""
    args = ["--records", "360", "--batch-records", "23", "--levels", "90"]
    p1, one = _run([DRIVER, "--gpus", "1"] + args)
    assert p1.returncode == 0, p1.stderr[-2000:]
    p8, eight = _run([DRIVER, "--gpus", "8"] + args, timeout=600)
    assert p8.returncode == 0, p8.stderr[-2000:]
    a, b = _by_op(one), _by_op(eight)
    assert a.keys() == b.keys() and len(a) == 5  # config 4: two operators; config 5: fused, chain, chain as written under grid.fused()
    for k in a:
        assert b[k]["n_gpus"] == 8 and len(b[k]["per_rank_device_ms"]) == 8
        assert sum(b[k]["per_rank_cells"]) == a[k]["cells"] and a[k]["checksum_u64"] == b[k]["checksum_u64"], k
        assert 0 < b[k]["rank_balance_min_over_max"] <= 1
    c4 = next(v for k, v in b.items() if k[0] == 4)
    assert c4["records_per_rank"] == [45] * 8 and c4["records_per_resident_batch"] == 23 and c4["batch_rounds"] == 2
    c5 = next(v for k, v in b.items() if k[0] == 5)
    assert c5["levels_per_rank"] == [12, 12, 11, 11, 11, 11, 11, 11]
    assert all(ln["ok"] for ln in eight if "check" in ln)


def test_ranks_without_units_run_the_whole_driver():
    """more ranks than records / levels: the empty ranks pass every barrier, contribute nothing to the checksums and do
    not drag the batch size down to one record"""
    args = ["--records", "3", "--batch-records", "0", "--levels", "5"]
    _, one = _run([DRIVER, "--gpus", "1"] + args)
    p8, eight = _run([DRIVER, "--gpus", "8"] + args, timeout=600)
    assert p8.returncode == 0, p8.stderr[-2000:]
    a, b = _by_op(one), _by_op(eight)
    for k in a:
        assert a[k]["checksum_u64"] == b[k]["checksum_u64"] and sum(b[k]["per_rank_cells"]) == a[k]["cells"], k
    c4 = next(v for k, v in b.items() if k[0] == 4)
    assert c4["records_per_rank"] == [1, 1, 1, 0, 0, 0, 0, 0] and c4["per_rank_cells"][3:] == [0] * 5
    assert c4["records_per_resident_batch"] >= 1 and c4["batch_rounds"] == 1
    c5 = next(v for k, v in b.items() if k[0] == 5)
    assert c5["levels_per_rank"] == [1, 1, 1, 1, 1, 0, 0, 0]


def test_config4_checksum_is_the_oracles():
    """the driver's checksum of checksums against the oracle's cumsum of the same synthetic records"""
    from oracle import refimpl as R

    _, one = _run([DRIVER, "--gpus", "1", "--records", "3", "--batch-records", "2"])
    nz, ny, nx = 5, 6, 8
    T = R.synthetic_field((3, nz, ny, nx), 4)
    for to, line in zip(("left", "outer"), [ln for ln in one if ln.get("config") == 4]):
        want = R.grid_cumsum(T, 1, "center", to, "fill")
        chk = int(np.ascontiguousarray(want).view(np.uint64).sum(dtype=np.uint64))
        assert line["checksum_u64"] == f"{chk:016x}"
        assert line["cells"] == T.size


def test_rank_count_mismatches_fail_loudly():
    # a launcher-provided WORLD_SIZE that disagrees with --gpus
    p, _ = _run([DRIVER, "--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr
    # RCCL ranks need one GPU each: no GPU here => --gpus 2 with the nccl backend refuses to start
    p, _ = _run([DRIVER, "--gpus", "2"], {"XG_DIST_BACKEND": "nccl"})
    assert p.returncode != 0 and "GPU(s) are visible" in p.stderr


def test_bench_py_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus 2` on a box with fewer GPUs exits non-zero instead of printing n_gpus: 1"""
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    p, lines = _run([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], {"XG_DIST_BACKEND": ""})
    assert p.returncode != 0 and not lines


def test_records_per_batch_and_record_batches():
    from xgcm_amd.sharding import record_batches, records_per_batch, shard_bounds

    rec = 8 * (2 * 75 * 2400 * 3600 + 2400 * 3600)
    assert records_per_batch(45, rec, free_bytes=280 << 30, headroom=0.8) == 23
    assert records_per_batch(45, rec, free_bytes=1 << 30) == 1          # never zero: one record at a time
    assert records_per_batch(3, rec, free_bytes=280 << 30) == 3          # never more than the rank owns
    assert records_per_batch(45, rec, free_bytes=280 << 30, cap=8) == 8
    assert records_per_batch(0, rec, free_bytes=280 << 30) == 0
    for world in (1, 2, 8):
        cover = []
        for r in range(world):
            b = record_batches(360, world, r, 23)
            lo, hi = shard_bounds(360, world, r)
            assert b[0][0] == lo and b[-1][1] == hi and all(x[1] == y[0] for x, y in zip(b, b[1:]))
            assert all(0 < e - s <= 23 for s, e in b)
            cover += b
        assert sum(e - s for s, e in cover) == 360
    assert record_batches(3, 8, 5, 4) == []  # a rank without records
