"""Grid-level differential fuzzing ON the GPU box: the generator of `oracle/fuzz_against_reference.py` (random grids, connected
faces, folds, fields of every served dtype, metrics, user ufuncs, `pad`, `transform`; valid and invalid calls), every call
made twice -- through the HIP library and through the oracle-backed device double -- and compared like the live fuzz
compares with the reference: same exception class, or same dims / name / dtype / attrs / coordinates / values.

The double is what the build container compares with the reference's own `Grid` on fresh seeds
(tests/test_reference_suite_live.py); this test carries that agreement over to the kernels on seeds no fixture holds."""
import warnings

import numpy as np
import pytest

from oracle import fake_device
from oracle import fuzz_against_reference as F
from xgcm_amd import Dataset, Grid
from xgcm_amd.padding import pad as our_pad

pytestmark = pytest.mark.gpu


def _outcomes(seed, case, fused=False):
    ds, gkw, variables, calls = F.build_case(Dataset, seed, case)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            grid = Grid(ds, **(dict(gkw, fuse=True) if fused else gkw))
        except Exception as exc:  # noqa: BLE001
            return [("grid", None, exc)]
        return [(f"{m}({v}, {a}, {kw})",) + F._call(grid, ds, m, v, a, kw, our_pad) for m, v, a, kw in calls]


@pytest.mark.parametrize("seed, fused", [(7001, False), (7002, False), (7003, False), (7004, True)])
def test_hip_and_the_oracle_double_agree_on_fresh_seeds(seed, fused, monkeypatch):
    """`fused`: every Grid on the HIP side built `fuse=True` (deferred results, computed for the comparison) against the EAGER
    double -- the deferred mode's kernels against the operator-by-operator chain"""
    n = 120
    on_hip = [_outcomes(seed, case, fused) for case in range(n)]
    with monkeypatch.context() as mp:
        fake_device.install(mp)
        on_double = [_outcomes(seed, case) for case in range(n)]
    compared = 0
    for case, (hs, ds) in enumerate(zip(on_hip, on_double)):
        assert len(hs) == len(ds)
        for (what, got, got_exc), (_, ref, ref_exc) in zip(hs, ds):
            diff = F.compare(ref, ref_exc, got, got_exc)
            assert diff is None, (seed, case, what, diff)
            compared += 1
    assert compared > 1000
