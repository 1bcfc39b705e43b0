"""Helper of tests/test_launcher.py (not a test): the sharded config-4 / config-5 drivers of
tools/bench_configs.py at toy sizes, on the oracle-backed device double, through the SAME launcher the GPU runs
use (`sharding.ensure_ranks` -> torch.distributed.run -> `sharding.init_ranks`), backend gloo.

    python tests/_sharded_cpu_driver.py --gpus N
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


class _Patch:  # minimal monkeypatch stand-in
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--records", type=int, default=7)
    ap.add_argument("--batch-records", type=int, default=2)
    ap.add_argument("--levels", type=int, default=5, help="config 5: levels of the field split along Z")
    ap.add_argument("--configs", default="4,5")
    a = ap.parse_args()
    from xgcm_amd import sharding as S

    S.ensure_ranks(a.gpus, os.path.abspath(__file__), sys.argv[1:])
    ranks = S.init_ranks(a.gpus)
    from oracle import fake_device

    fake_device.install(_Patch())
    import bench_configs as B

    try:
        if "4" in a.configs.split(","):
            B.run_config4(ranks, a.records, shape=(5, 6, 8), per_batch=a.batch_records or None)
        if "5" in a.configs.split(","):
            B.run_config5(ranks, shape=(a.levels, 6, 8), reps=2)
    finally:
        ranks.close()


if __name__ == "__main__":
    main()
