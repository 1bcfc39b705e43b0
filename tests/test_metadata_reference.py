"""`Grid(ds)` from metadata against the REFERENCE's own parsers, description for description.

tests/golden/metadata_reference.json was written by `oracle/make_golden_metadata.py`: the reference's
`xgcm/metadata_parsers.py`, `comodo.py`, `sgrid.py` and `Grid.__init__` imported UNMODIFIED over `oracle/xr_min.py` and run
on 57 dataset descriptions (COMODO: every staggered position x both signs of the shift, wrong lengths, malformed shifts,
several coordinates per position, axis names of any spelling; SGRID: 1-3 dimensions, every padding word, vertical
dimensions, the `Conventions` spellings the reference accepts and the ones it does not, both conventions at once,
missing and inconsistent topology attributes).  Each description is rebuilt here as an `xgcm_amd.Dataset`;
`xgcm_amd.metadata.parse_metadata` must return the same position tables (position ORDER included) and `xgcm_amd.Grid(ds)`
the same axes -- or raise the same error type with the same text.  The ORDER OF AXES is not compared: the reference
collects axis names in a `set` (xgcm/comodo.py:24-28), so its order is the process's hash seed's.

Pinned MODULO THE STAND-IN, like tests/test_grid_reference.py.  No kernels run: CPU suite only.
"""

import json
import os
import warnings

import numpy as np
import pytest

from xgcm_amd import Dataset, Grid
from xgcm_amd import metadata as M

META = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "metadata_reference.json")))
ERRORS = {"ValueError": ValueError, "KeyError": KeyError, "IndexError": IndexError, "TypeError": TypeError}


def _build(d):
    coords = {k: (k, np.arange(float(c["len"])), dict(c["attrs"])) for k, c in d["coords"].items()}
    variables = {}
    if d["topology"] is not None:
        variables["topo"] = ((), np.array(1, dtype="int32"), dict(d["topology"]))
    return Dataset(variables, coords=coords, attrs=dict(d["attrs"]))


def _check(expected, fn, where):
    if "raises" in expected:
        with pytest.raises(ERRORS[expected["raises"]["type"]]) as info:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                fn()
        got = str(info.value.args[0]) if info.value.args else str(info.value)
        assert got == expected["raises"]["message"], where
        return None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return fn()


@pytest.mark.parametrize("name", sorted(META["cases"]))
def test_parse_metadata_as_the_reference(name):
    c = META["cases"][name]
    got = _check(c["parse_metadata"], lambda: M.parse_metadata(_build(c["dataset"]))[1], name)
    if got is not None:
        want = {ax: positions for ax, positions in c["parse_metadata"]["ok"]}
        assert set(got) == {"coords"} and set(got["coords"]) == set(want), name
        for ax, positions in want.items():
            assert [list(kv) for kv in got["coords"][ax].items()] == positions, f"{name}: axis {ax}"


@pytest.mark.parametrize("name", sorted(META["cases"]))
def test_grid_from_metadata_as_the_reference(name):
    c = META["cases"][name]
    grid = _check(c["grid"], lambda: Grid(_build(c["dataset"])), name)
    if grid is not None:
        assert {ax: dict(a.coords) for ax, a in grid.axes.items()} == c["grid"]["ok"], name


def test_the_fixture_set_is_the_one_described():
    assert META["n"] == len(META["cases"]) >= 57
    assert sum("raises" in c["parse_metadata"] for c in META["cases"].values()) >= 20
    assert sum(n.startswith("sgrid") for n in META["cases"]) >= 24 and sum(n.startswith("comodo") for n in META["cases"]) >= 30
