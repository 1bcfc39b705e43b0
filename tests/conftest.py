import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the C-ABI library is a build artefact (git-ignored): (re)build it when missing or stale so a
    # clean checkout is testable; a failed build is reported by the tests that need the library
    try:
        import __graft_entry__ as entry

        if entry._stale():
            entry.build()
        if entry._host_stale():
            entry.build_host()
    except Exception as exc:  # pragma: no cover
        print(f"[conftest] could not build libxgcm_hip.so: {exc}", file=sys.stderr)


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(params=["oracle-double", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request, monkeypatch):
    """Run a host-logic test twice: on CPU against the oracle-backed device double (no kernels),
    and on a GPU box against the real HIP library.  `backend` is the residency converter for
    inputs: identity on CPU, identity on GPU too (numpy in -> numpy out through PCIe)."""
    if request.param == "oracle-double":
        from oracle import fake_device

        fake_device.install(monkeypatch)
    return request.param


@pytest.fixture
def host_abi(monkeypatch):
    """The `Grid` stack over the HOST build of the C ABI (libxgcm_host.so: same symbols, host pointers, its own
    loops) -- BASELINE config 1, "plumbing, no GPU".  Not a product path: xgcm_amd never loads that library."""
    import host_abi_device

    host_abi_device.install(monkeypatch)
    return host_abi_device
