"""Host-side index/string logic of xgcm_amd checked against fixtures produced by the reference's
own functions (signature parser, ufunc selection, dispatch table) and the C-ABI export list.
No kernels are launched.  CPU only."""

import json
import os
import re

import pytest

import xgcm_amd
from xgcm_amd import _hip, gridops
from xgcm_amd.axis import Axis
from xgcm_amd.grid import _select_grid_ufunc
from xgcm_amd.grid_ufunc import GridUFunc, _GridUFuncSignature, as_grid_ufunc
from xgcm_amd.labeled import Dataset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


SIG = _load("signatures.json")


@pytest.mark.parametrize("case", SIG["parse"], ids=lambda c: c["string"])
def test_signature_parser_matches_reference(case):
    if not case["ok"]:
        with pytest.raises(ValueError) as e:
            _GridUFuncSignature.from_string(case["string"])
        assert str(e.value) == case["error"]
        return
    sig = _GridUFuncSignature.from_string(case["string"])
    assert [list(t) for t in sig.in_ax_names] == case["in_ax_names"]
    assert [list(t) for t in sig.in_ax_positions] == case["in_ax_positions"]
    assert [list(t) for t in sig.out_ax_names] == case["out_ax_names"]
    assert [list(t) for t in sig.out_ax_positions] == case["out_ax_positions"]
    assert str(sig) == case["str"]
    assert str(_GridUFuncSignature.from_string(str(sig))) == str(sig)


@pytest.mark.parametrize("case", SIG["equivalent"], ids=lambda c: f"{c['a']} ~ {c['b']}")
def test_signature_equivalence_matches_reference(case):
    a = _GridUFuncSignature.from_string(case["a"])
    b = _GridUFuncSignature.from_string(case["b"])
    assert a.equivalent(b) == case["equivalent"]
    assert b.equivalent(a) == case["equivalent"]


def test_multi_axis_equivalence_is_order_of_appearance():
    S = _GridUFuncSignature.from_string
    assert S("(X:center,Y:center)->(X:left)").equivalent(S("(A:center,B:center)->(A:left)"))
    assert not S("(X:center,Y:center)->(X:left)").equivalent(S("(A:center,B:center)->(B:left)"))


def test_dispatch_table_matches_reference():
    table = _load("gridops_table.json")
    mine = {k: v for k, v in vars(gridops).items() if isinstance(v, GridUFunc)}
    assert set(mine) == {r["name"] for r in table}
    for r in table:
        u = mine[r["name"]]
        assert str(u.signature) == r["signature"]
        pw = None if u.padding_width is None else {k: list(v) for k, v in u.padding_width.items()}
        assert pw == r["padding_width"]
        for attr in ("padding", "fill_value", "dask", "map_overlap", "pad_before_func"):
            assert getattr(u, attr) == r[attr], (r["name"], attr)


@pytest.mark.parametrize("row", _load("select.json"), ids=lambda r: f"{r['funcname']}:{r['from']}->{r['to']}")
def test_select_grid_ufunc_matches_reference(row):
    sig = _GridUFuncSignature.from_string(f"(Q:{row['from']})->(Q:{row['to']})")
    name_of = {id(v): k for k, v in vars(gridops).items() if isinstance(v, GridUFunc)}
    if "selected" in row:
        uf, rest = _select_grid_ufunc(row["funcname"], sig, module=gridops, padding="fill")
        assert name_of[id(uf)] == row["selected"]
        assert rest == row["kwargs"]
    else:
        exc = {"NotImplementedError": NotImplementedError, "ValueError": ValueError}[row["error"]]
        with pytest.raises(exc) as e:
            _select_grid_ufunc(row["funcname"], sig, module=gridops, padding="fill")
        assert str(e.value) == row["message"]


def test_select_is_ambiguous_for_duplicate_registrations():
    """reference test/test_grid_ufunc.py:1368-1420 uses a mock namespace the same way."""

    class ns:
        pass

    f = as_grid_ufunc(signature="(X:center)->(X:left)")(lambda a: a)
    ns.diff_one, ns.diff_two, ns.other = f, as_grid_ufunc(signature="(X:center)->(X:left)")(lambda a: a), 3
    sig = _GridUFuncSignature.from_string("(Z:center)->(Z:left)")
    with pytest.raises(ValueError, match="ambiguous"):
        _select_grid_ufunc("diff", sig, module=ns)
    with pytest.raises(NotImplementedError, match="Could not find any pre-defined interp grid ufuncs"):
        _select_grid_ufunc("interp", sig, module=ns)


def test_decorator_kwarg_validation():
    with pytest.raises(TypeError, match="Unsupported keyword argument"):
        as_grid_ufunc(signature="(X:center)->(X:left)", junk=1)
    with pytest.raises(ValueError, match="renamed to 'padding'"):
        as_grid_ufunc(signature="(X:center)->(X:left)", boundary="fill")
    with pytest.raises(ValueError, match="renamed to 'padding_width'"):
        as_grid_ufunc(signature="(X:center)->(X:left)", boundary_width={"X": (1, 0)})
    u = as_grid_ufunc(signature="(X:center)->(X:left)", padding_width={"X": (1, 0)}, fill_value=10)(lambda a: a)
    assert u.fill_value == 10 and u.pad_before_func is True and u.dask == "forbidden"
    with pytest.raises(AttributeError):
        u.boundary_width


def test_signature_from_type_hints():
    """reference test/test_grid_ufunc.py:104-213 (TestParseSignatureFromTypeHints)."""
    from typing import Annotated, Tuple

    import numpy as np

    with pytest.raises(ValueError, match="Must specify axis positions"):

        @as_grid_ufunc()
        def nothing(): ...

    @as_grid_ufunc()
    def f1(a: Annotated[np.ndarray, "X:center"]) -> Annotated[np.ndarray, "X:center"]:
        return a

    assert str(f1.signature) == "(X:center)->(X:center)"

    @as_grid_ufunc()
    def f2(a: Annotated[np.ndarray, "X:center,Y:center"]) -> Annotated[np.ndarray, "X:center"]:
        return a

    assert str(f2.signature) == "(X:center,Y:center)->(X:center)"

    @as_grid_ufunc()
    def f3(a: Annotated[np.ndarray, "X:left"], b: Annotated[np.ndarray, "Y:right"]) -> Annotated[np.ndarray, "X:center"]:
        return a

    assert str(f3.signature) == "(X:left),(Y:right)->(X:center)"

    @as_grid_ufunc()
    def f4(a: Annotated[np.ndarray, "X:center"]) -> Tuple[Annotated[np.ndarray, "X:left"], Annotated[np.ndarray, "Y:right"]]:
        return a, a

    assert str(f4.signature) == "(X:center)->(X:left),(Y:right)"

    @as_grid_ufunc()
    def f5(a: Annotated[np.ndarray, "X:center"]) -> Annotated[np.ndarray, "X:left,Y:right"]:
        return a

    assert str(f5.signature) == "(X:center)->(X:left,Y:right)"
    with pytest.raises(ValueError, match="only one of either type hints or signature kwarg"):

        @as_grid_ufunc(signature="(X:center)->(X:left)")
        def both(a: Annotated[np.ndarray, "X:center"]) -> Annotated[np.ndarray, "X:left"]:
            return a


def _ds(n=9):
    import numpy as np

    return Dataset(coords={"xc": ("xc", np.arange(n) + 0.5), "xg": ("xg", np.arange(n) * 1.0),
                           "xo": ("xo", np.arange(n + 1) * 1.0)})


def test_axis_validation_and_defaults():
    ds = _ds()
    ax = Axis(ds, "X", {"center": "xc", "left": "xg", "outer": "xo"})
    assert ax.default_shifts == {"center": "left", "left": "center", "outer": "center"}
    assert ax.padding is None and ax.fill_value == 0.0 and ax.periodic is False
    assert Axis(ds, "X", {"center": "xc", "outer": "xo"}).default_shifts["center"] == "outer"
    assert Axis(ds, "X", {"center": "xc", "left": "xg"}, default_shifts={"center": "left"}, padding="periodic").periodic
    with pytest.raises(ValueError, match="Axis position must be one of"):
        Axis(ds, "X", {"centre": "xc"})
    with pytest.raises(ValueError, match="Could not find dimension `nope`"):
        Axis(ds, "X", {"center": "nope"})
    with pytest.raises(ValueError, match="cannot be assigned to multiple positions"):
        Axis(ds, "X", {"center": "xc", "left": "xc"})
    with pytest.raises(ValueError, match="Can't set the default shift for center to be to center"):
        Axis(ds, "X", {"center": "xc", "left": "xg"}, default_shifts={"center": "center"})
    with pytest.raises(ValueError, match="padding must be one of"):
        Axis(ds, "X", {"center": "xc"}, padding="wrap")
    with pytest.raises(TypeError, match="fill value must be"):
        Axis(ds, "X", {"center": "xc"}, fill_value="bad")
    with pytest.raises(TypeError, match="name argument must be of type str"):
        Axis(ds, 3, {"center": "xc"})
    with pytest.raises(ValueError, match="renamed to 'padding'"):
        Axis(ds, "X", {"center": "xc"}, boundary="fill")
    with pytest.raises(AttributeError, match="renamed to 'padding'"):
        ax.boundary


def test_axis_position_lookup():
    import numpy as np

    from xgcm_amd import DataArray

    ds = _ds()
    ax = Axis(ds, "X", {"center": "xc", "left": "xg"})
    assert ax._get_position_name(DataArray(np.zeros((3, 9)), ("t", "xg"))) == ("left", "xg")
    assert ax._get_axis_dim_num(DataArray(np.zeros((3, 9)), ("t", "xc"))) == 1
    with pytest.raises(KeyError, match="None of the DataArray's dims"):
        ax._get_position_name(DataArray(np.zeros(3), ("t",)))
    with pytest.raises(KeyError, match="cannot have more than 1 axis dimension"):
        ax._get_position_name(DataArray(np.zeros((9, 9)), ("xc", "xg")))


def test_c_abi_library_exports_every_declared_symbol():
    """The built .so loads without a GPU and exports exactly what include/xgcm_hip.h declares."""
    header = open(os.path.join(ROOT, "include", "xgcm_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|void\s*\*?)\s*(xg_\w+)\s*\(", header, flags=re.M))  # (xg_pool_alloc / _free: torch's allocator signature)
    assert declared == set(_hip.SIGNATURES), declared ^ set(_hip.SIGNATURES)
    lib = _hip.load()  # raises ImportError if not built, AttributeError if a symbol is missing
    for name in declared:
        assert hasattr(lib, name)
    assert lib.xg_version() == 1


def test_c_abi_header_compiles_as_c_and_links():
    """The header is plain C and the demo links against the library without a GPU present."""
    import subprocess
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "demo")
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", os.path.join(ROOT, "examples", "c_abi_demo.c"),
                               "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "xgcm_amd"), "-lxgcm_hip",
                               "-Wl,-rpath," + os.path.join(ROOT, "xgcm_amd"), "-lm", "-o", exe])
        res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert res.returncode == 2 and "no GPU" in res.stderr  # loads, reports the missing device, no crash


def test_fails_loudly_without_library_or_gpu(monkeypatch):
    """No silent fallback: a missing .so is an ImportError, a missing GPU a RuntimeError."""
    import numpy as np
    import torch

    from xgcm_amd import DataArray, Grid, device

    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", os.path.join(ROOT, "xgcm_amd", "no_such_library.so"))
    with pytest.raises(ImportError, match="has not been built"):
        _hip.load()
    monkeypatch.undo()
    if not torch.cuda.is_available():
        ds = Dataset(coords={"xc": ("xc", np.arange(8.0)), "xg": ("xg", np.arange(8.0))})
        grid = Grid(ds, coords={"X": {"center": "xc", "left": "xg"}}, padding="periodic", autoparse_metadata=False)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            grid.diff(DataArray(np.arange(8.0), ("xc",)), "X")
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            device.stencil1d("diff", np.arange(8.0), 0, 1, 0, "periodic")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "xgcm_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("oracle.refimpl.synthetic", "").replace("oracle/refimpl.py", ""), fn
    assert xgcm_amd.__all__


def test_tunables_set_and_get_without_a_gpu():
    """xg_set_tunable / xg_get_tunable: known names round-trip, unknown names are an error (no GPU needed)."""
    d = _hip.get_tunable("zb_rows")
    try:
        _hip.set_tunable("zb_rows", 32)
        assert _hip.get_tunable("zb_rows") == 32
    finally:
        _hip.set_tunable("zb_rows", d)
    assert _hip.get_tunable("contig_rw") >= 0
    with pytest.raises(_hip.XgcmHipError, match="unknown tunable"):
        _hip.set_tunable("no_such_knob", 1)


def test_graph_capture_needs_the_gpu():
    """xgcm_amd.graphs.capture is hipGraph capture: without a GPU it fails loudly instead of running `fn` eagerly."""
    import torch

    from xgcm_amd.graphs import capture

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="needs the GPU"):
        capture(lambda: None)


def test_every_tunable_is_documented():
    """INTEGRATION.md's tunables table names every entry of the library's table (xg_runtime.hip)."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = re.findall(r'\{"(\w+)", &Tune::', open(os.path.join(root, "xgcm_amd", "csrc", "xg_runtime.hip")).read())
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    assert len(names) >= 30 and not [n for n in names if f"`{n}`" not in doc]
