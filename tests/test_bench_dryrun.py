"""The day a node with 8 GPUs appears, `bench.py --gpus 8` and `tools/scale_table.py --gpus 1,2,4,8` must simply work (VERDICT r05
"next round" 4): a CPU dry run of exactly those commands with gloo ranks.  Everything on the path is the real thing --
`sharding.ensure_ranks` (the self-relaunch under torch.distributed.run), `init_ranks`, `bind_to_gpu_numa`, the barriers, the
gathers, the max-over-ranks clock, the JSON line with its `configs` block -- except the memory the device layer computes in:
`tests/dryrun_site/usercustomize.py` swaps it for the host build of the C ABI in every process of the run (the ranks' product
path imports nothing from `oracle/`; bench.py's own oracle checks of the `configs` slabs run on rank 0 as they do on the GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPE = "5,12,32;90,8,16"  # record (Z, Y, X); config 5 with BASELINE's 90 levels (12,12,11,... on 8 ranks)


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update({"XG_DRYRUN_HOST_ABI": "1", "XG_DIST_BACKEND": "gloo", "OMP_NUM_THREADS": "1",
                "PYTHONPATH": os.path.join(ROOT, "tests", "dryrun_site") + os.pathsep + env.get("PYTHONPATH", "")})
    return env


def _bench(n, extra=(), launcher=False):
    args = ["bench.py", "--gpus", str(n), "--steps", "2", "--warmup", "1", "--shape", SHAPE, "--no-pmc", "--config4-records", "2",
            "--config-reps", "2", *extra]
    if launcher:  # as the round-end driver starts N > 1
        from xgcm_amd.sharding import _free_port

        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + args
    else:         # started by hand: bench.py re-executes itself under the launcher (sharding.ensure_ranks)
        cmd = [sys.executable] + args
    p = subprocess.run(cmd, capture_output=True, text=True, env=_env(), cwd=ROOT, timeout=600)
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


def test_the_dry_run_hook_is_inert_without_its_switch():
    env = _env()
    env.pop("XG_DRYRUN_HOST_ABI")
    p = subprocess.run([sys.executable, "-c", "import sys; print('host_abi_device' in sys.modules)"], capture_output=True, text=True, env=env, cwd=ROOT)
    assert p.stdout.strip() == "False"


def test_single_process_line_and_its_oracle_checks():
    p, lines = _bench(1)
    assert p.returncode == 0 and len(lines) == 1, p.stderr[-2000:]
    ln = lines[0]
    assert ln["n_gpus"] == 1 and ln["ranks"]["world_size"] == 1 and ln["ranks"]["backend"] == "single process"
    assert ln["config"]["workload"].startswith("NOT BASELINE's shape") and ln["roofline"]["traffic"] is None
    assert ln["parity_spot_check"] is True and "cpu_baseline" in ln
    ops = [e for c in ("config3", "config4", "config5") for e in ln["configs"][c]["ops"]]
    assert len(ops) == 9 and all(e["bit_exact_vs_oracle"] is True for e in ops)
    assert len(ln["configs"]["box_probe"]["ms"]) == 3


@pytest.mark.parametrize("launcher", [True, False], ids=["under-torchrun", "self-relaunch"])
def test_eight_ranks(launcher):
    p, lines = _bench(8, ["--no-cpu-baseline"], launcher=launcher)
    assert p.returncode == 0 and len(lines) == 1, p.stderr[-3000:]  # ONE line, from rank 0
    if not launcher:
        assert "[launcher]" in p.stderr and "--nproc-per-node=8" in p.stderr
    ln = lines[0]
    rk = ln["ranks"]
    assert ln["n_gpus"] == 8 and rk["world_size"] == 8 and rk["backend"] == "gloo" and ln["scaling"] == "weak"
    assert len(rk["placement"]) == 8 and sorted(pl["rank"] for pl in rk["placement"]) == list(range(8))
    assert sorted(pl["local_rank"] for pl in rk["placement"]) == list(range(8)) and len(rk["per_rank_ms_per_step"]) == 8
    assert 0 < rk["rank_balance_min_over_max"] <= 1
    # value = the cells of ALL ranks / the slowest rank's time
    cells = 8 * ln["config"]["cells_per_step_per_gpu"]
    assert ln["value"] == pytest.approx(cells / (ln["ms_per_step"] * 1e-3) / 1e9, rel=0.02, abs=2e-3)
    assert ln["ms_per_step"] == pytest.approx(max(rk["per_rank_ms_per_step"]), rel=1e-3, abs=1e-3)
    c = ln["configs"]
    assert "levels per rank [12, 12, 11, 11, 11, 11, 11, 11]" in c["config5"]["workload"]
    assert "2 of their 45 records" in c["config4"]["workload"]
    for key, nops in (("config3", 4), ("config4", 2), ("config5", 3)):
        assert len(c[key]["ops"]) == nops
        for e in c[key]["ops"]:
            assert len(e["per_rank_ms"]) == 8 and e["job_gcell_s"] > 0
    assert len(c["box_probe"]["per_rank_ms"]) == 8
    assert "cpu_baseline" not in ln


def test_rank_count_mismatch_is_refused():
    env = dict(_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--shape", SHAPE, "--no-pmc"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout)


def test_scale_table_one_two_four_eight(tmp_path):
    out = str(tmp_path / "scale")
    p = subprocess.run([sys.executable, os.path.join("tools", "scale_table.py"), "--gpus", "1,2,4,8", "--steps", "2", "--warmup", "1", "--records", "16",
                        "--reps", "2", "--shape", "5,6,8", "--bench-shape", SHAPE, "--out", out], capture_output=True, text=True, env=_env(), cwd=ROOT, timeout=1200)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "NOT the table" not in p.stdout and "skipped" not in p.stdout
    rows = [json.loads(ln) for ln in open(out + ".jsonl")]
    bench = {r["n"]: r for r in rows if r["what"].startswith("bench")}
    assert sorted(bench) == [1, 2, 4, 8]
    for n, r in bench.items():
        assert r["world_size"] == n and r["placements"] == n and r["configs_in_line"] == ["config3", "config4", "config5"]
        assert r["backend"] == ("single process" if n == 1 else "gloo")
    for what in {r["what"] for r in rows} - {"bench: interp+diff X,Y, one record per GPU (weak)"}:
        assert sorted(r["n"] for r in rows if r["what"] == what) == [1, 2, 4, 8], what
    md = open(out + ".md").read()
    for n in (1, 2, 4, 8):
        line = next(ln for ln in md.splitlines() if ln.startswith(f"N = {n}: "))
        assert len(json.loads(line.split(": ", 1)[1])) == n
