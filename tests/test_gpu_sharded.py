"""The N-rank path on the GPU box (which has ONE GPU): the launcher and the sharded config-4 / config-5 drivers
through the real HIP library -- one rank over RCCL (`XG_BENCH_FORCE_DIST=1`), two gloo ranks computing on the
same GPU, and the refusal to start more RCCL ranks than there are GPUs."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = os.path.join(ROOT, "tools", "bench_configs.py")
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "XG_DIST_BACKEND", "XG_SHARE_GPU"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable] + args, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    return p, [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]


SHAPE = (9, 66, 256)
ARGS = ["--configs", "4,5", "--records", "7", "--batch-records", "2", "--shape", ",".join(map(str, SHAPE)), "--reps", "3"]


def test_two_ranks_on_the_real_library_equal_one_rank_and_the_oracle():
    from oracle import refimpl as R

    p1, one = _run([CONFIGS, "--gpus", "1"] + ARGS)
    assert p1.returncode == 0, p1.stderr[-3000:]
    p2, two = _run([CONFIGS, "--gpus", "2"] + ARGS, {"XG_DIST_BACKEND": "gloo", "XG_SHARE_GPU": "1"})
    assert p2.returncode == 0, p2.stderr[-3000:]
    ops1 = [ln for ln in one if "op" in ln]
    ops2 = [ln for ln in two if "op" in ln]
    assert len(ops1) == len(ops2) == 5   # config 4: two operators; config 5: fused, chain, chain as written under grid.fused()
    for a, b in zip(ops1, ops2):
        assert a["op"] == b["op"] and a["n_gpus"] == 1 and b["n_gpus"] == 2
        assert a["checksum_u64"] == b["checksum_u64"] and a["cells"] == b["cells"]
    assert ops2[0]["records_per_rank"] == [4, 3] and ops2[2]["levels_per_rank"] == [5, 4]
    assert all(ln["ok"] for ln in one + two if "check" in ln)
    # against the oracle: cumsum of the same synthetic records, vorticity of the same synthetic fields
    nz, ny, nx = SHAPE
    T = R.synthetic_field((7, nz, ny, nx), 4)
    for to, line in zip(("left", "outer"), ops1[:2]):
        want = R.grid_cumsum(T, 1, "center", to, "fill")
        assert line["checksum_u64"] == f"{int(np.ascontiguousarray(want).view(np.uint64).sum(dtype=np.uint64)):016x}"
    U = R.synthetic(nz * ny * nx, 51).reshape(nz, ny, nx)
    V = R.synthetic(nz * ny * nx, 52).reshape(nz, ny, nx)
    area = R.synthetic(ny * nx, 53, 0, 1000.0, 1000.0).reshape(ny, nx)
    want = (R.stencil1d("diff", V, 2, 1, 0, "fill") - R.stencil1d("diff", U, 1, 1, 0, "fill")) / area
    chk = f"{int(np.ascontiguousarray(want).view(np.uint64).sum(dtype=np.uint64)):016x}"
    assert ops1[2]["checksum_u64"] == chk and ops1[3]["checksum_u64"] == chk and ops1[4]["checksum_u64"] == chk


def test_bench_line_measures_its_hbm_traffic_in_the_run():
    """`roofline.traffic` of the N = 1 line comes from two `rocprofv3 --pmc` child passes of the same command made after the
    timed region (FETCH_SIZE, WRITE_SIZE, one per pass); the flat stencils move their algorithmic bytes and nothing else.
    Without rocprofv3 the committed figure is reported and flagged."""
    import shutil

    p, lines = _run([BENCH, "--steps", "3", "--warmup", "1", "--levels", "12", "--no-cpu-baseline"], timeout=500)
    assert p.returncode == 0, p.stderr[-3000:]
    roof = lines[-1]["roofline"]
    if shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3"):
        assert roof["traffic_measured_in_run"] is True, roof["traffic_source"]
        assert 0.98 <= roof["traffic_over_algorithmic"] <= 1.05, roof
    else:
        assert roof["traffic_measured_in_run"] is False
    p, lines = _run([BENCH, "--steps", "2", "--warmup", "0", "--levels", "4", "--no-cpu-baseline", "--no-pmc"])
    assert p.returncode == 0 and lines[-1]["roofline"]["traffic_measured_in_run"] is False


def test_bench_through_rccl_with_one_rank():
    p, lines = _run([BENCH, "--gpus", "1", "--steps", "2", "--warmup", "1", "--levels", "4", "--no-cpu-baseline"],
                    {"XG_BENCH_FORCE_DIST": "1"})
    assert p.returncode == 0, p.stderr[-3000:]
    assert "torch.distributed.run" in p.stderr
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1
    assert lines[0]["ranks"]["world_size"] == 1 and "nccl" in lines[0]["ranks"]["backend"]
    assert len(lines[0]["ranks"]["per_rank_ms_per_step"]) == 1


def test_more_ranks_than_gpus_is_an_error():
    import torch

    n = torch.cuda.device_count() + 1
    for script, extra in ((BENCH, ["--steps", "1", "--warmup", "0"]), (CONFIGS, ARGS)):
        p, lines = _run([script, "--gpus", str(n)] + extra)
        assert p.returncode != 0 and not lines
        assert "GPU(s) are visible" in p.stderr


@pytest.mark.parametrize("ranks", [1, 2], ids=["one-process", "two-gloo-ranks-on-one-gpu"])
def test_bench_configs_block_on_the_real_library(ranks):
    """bench.py's `configs` block (configs 3 / 4 / 5 after the headline's timed region) through HIP at a plumbing shape: nine
    operators timed with HIP events, every kept slab equal to the oracle bit for bit, the box probe's three launches; with two
    ranks (gloo, sharing the one GPU) the per-rank times of every operator and the job's aggregate."""
    env = {} if ranks == 1 else {"XG_DIST_BACKEND": "gloo", "XG_SHARE_GPU": "1"}
    p, lines = _run([BENCH, "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--shape", "6,64,256;8,64,128", "--no-pmc",
                     "--config4-records", "2", "--config-reps", "3"], env)
    assert p.returncode == 0 and len(lines) == 1, p.stderr[-3000:]
    ln = lines[0]
    assert ln["n_gpus"] == ranks and ln["config"]["workload"].startswith("NOT BASELINE's shape")
    c = ln["configs"]
    ops = [e for k in ("config3", "config4", "config5") for e in c[k]["ops"]]
    assert len(ops) == 9 and all(e["bit_exact_vs_oracle"] is True and e["ms"] > 0 for e in ops)
    assert len(c["box_probe"]["ms"]) == 3
    if ranks == 2:
        assert all(len(e["per_rank_ms"]) == 2 and e["job_gcell_s"] > 0 for e in ops)
        assert "levels per rank [4, 4]" in c["config5"]["workload"]
