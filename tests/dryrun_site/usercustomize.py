"""TEST INFRASTRUCTURE ONLY (tests/test_bench_dryrun.py).  A directory the dry-run test puts on PYTHONPATH: Python imports a
module named `usercustomize` at start-up, so EVERY python process of the run -- `bench.py`, `tools/bench_configs.py`,
`tools/scale_table.py`, and the ranks `torch.distributed.run` starts for them -- finds the device layer's memory already swapped
for host memory + the host build of the C ABI (tests/host_abi_device.py), exactly as the `host_abi` fixture does inside pytest.
Nothing else changes: the launcher (`sharding.ensure_ranks`), `init_ranks`, `bind_to_gpu_numa`, the barriers, the gathers and
the JSON assembly are the product's own code, run by N gloo ranks.  Inert unless XG_DRYRUN_HOST_ABI=1."""
import os

if os.environ.get("XG_DRYRUN_HOST_ABI") == "1":
    import sys

    _tests = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for _p in (_tests, os.path.dirname(_tests)):
        if _p not in sys.path:
            sys.path.insert(0, _p)

    class _Patch:  # minimal monkeypatch stand-in: the process ends with the patch in place
        def setattr(self, obj, name, val):
            setattr(obj, name, val)

    import host_abi_device

    host_abi_device.install(_Patch())
