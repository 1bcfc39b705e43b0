"""xg_copy_nd: the strided N-d copy that lays out transposed / flipped / broadcast / sliced views (include/xgcm_hip.h).

CPU: the host build of the ABI against numpy (the binding and the argument conventions).  GPU: the three kernels -- rows,
32 x 32 LDS transpose, gather -- through xgcm_amd.device against numpy on the same bytes, every element size."""

import ctypes as C
import itertools

import numpy as np
import pytest

from xgcm_amd import _hip


def _host_lib():
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xgcm_amd", "libxgcm_host.so")
    if not os.path.exists(path):
        import __graft_entry__ as g

        g.build_host()
    lib = C.CDLL(path)
    res, args = _hip.SIGNATURES["xg_copy_nd"]
    lib.xg_copy_nd.restype, lib.xg_copy_nd.argtypes = res, args
    return lib


def _host_copy(lib, view: np.ndarray, out_view: np.ndarray):
    """dst view <- src view through the host ABI: element strides from the numpy views, negative source strides allowed"""
    item = view.itemsize
    assert out_view.itemsize == item and view.shape == out_view.shape
    rc = lib.xg_copy_nd(view.ctypes.data_as(C.c_void_p), _hip.i64([s // item for s in view.strides]),
                        out_view.ctypes.data_as(C.c_void_p), _hip.i64([s // item for s in out_view.strides]),
                        _hip.i64(list(view.shape)), view.ndim, item, None)
    return rc


@pytest.mark.parametrize("dtype", ["float64", "float32", "int16", "uint8"])
def test_host_abi_copy_matches_numpy(dtype):
    lib = _host_lib()
    rng = np.random.default_rng(1)
    a = (rng.standard_normal((3, 5, 7, 4)) * 100).astype(dtype)
    for view in (a.transpose(2, 0, 3, 1), a[:, ::-1], a[..., ::-1], a[::-1, :, ::-1], np.broadcast_to(a[:, :1], a.shape),
                 a[1:, 2:4], a.transpose(3, 2, 1, 0)[::-1]):
        out = np.full(view.shape, 7, dtype=dtype)
        assert _host_copy(lib, view, out) == 0
        np.testing.assert_array_equal(out, view)
    big = np.zeros((3, 12, 7, 4), dtype=dtype)  # concatenation: a part into its slice of the result
    assert _host_copy(lib, a, big[:, 2:7]) == 0
    np.testing.assert_array_equal(big[:, 2:7], a)
    assert not big[:, :2].any() and not big[:, 7:].any()
    # refusals: a destination that would be written twice, odd element sizes
    assert lib.xg_copy_nd(a.ctypes.data_as(C.c_void_p), _hip.i64([0, 0, 0, 1]), big.ctypes.data_as(C.c_void_p), _hip.i64([0, 4, 0, 1]),
                          _hip.i64([3, 5, 1, 4]), 4, a.itemsize, None) != 0
    assert lib.xg_copy_nd(a.ctypes.data_as(C.c_void_p), _hip.i64([1]), big.ctypes.data_as(C.c_void_p), _hip.i64([1]), _hip.i64([4]), 1, 3, None) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float64", "float32", "int64", "int16", "bool"])
def test_device_layouts_match_numpy(dtype):
    import torch

    from xgcm_amd import device as dev

    rng = np.random.default_rng(2)
    for shape in ((3, 40, 130), (2, 5, 64, 96), (7, 33), (4, 3, 2, 5, 6, 7)):
        a = (rng.standard_normal(shape) * 100)
        a = (a > 0) if dtype == "bool" else a.astype(dtype)
        t = torch.from_numpy(a).cuda()
        for perm in itertools.islice(itertools.permutations(range(a.ndim)), 0, 30, 3):
            got = dev.materialize(t.permute(*perm))
            assert got.is_contiguous()
            np.testing.assert_array_equal(dev.tohost(got), np.ascontiguousarray(a.transpose(perm)))
        for axes in ([0], [-1], [0, -1], list(range(a.ndim))):
            np.testing.assert_array_equal(dev.tohost(dev.flip(t, axes)), np.flip(a, axes))
            np.testing.assert_array_equal(dev.tohost(dev.flip(t.permute(*reversed(range(a.ndim))), axes)), np.flip(a.T, axes))
        np.testing.assert_array_equal(dev.tohost(dev.materialize(t[..., ::2])), a[..., ::2])              # gather: no unit stride
        np.testing.assert_array_equal(dev.tohost(dev.materialize(t[:1].expand(*shape))), np.broadcast_to(a[:1], shape))
        for ax in range(a.ndim):
            parts = [t.narrow(ax, 0, 1), t, t.narrow(ax, shape[ax] - 1, 1)]
            np.testing.assert_array_equal(dev.tohost(dev.concatenate(parts, ax)),
                                          np.concatenate([a.take([0], ax), a, a.take([shape[ax] - 1], ax)], ax))


@pytest.mark.gpu
def test_kernel_sized_transposes_and_unaligned_rows():
    import torch

    from xgcm_amd import device as dev

    rng = np.random.default_rng(3)
    a = rng.standard_normal((6, 515, 1030))
    t = torch.from_numpy(a).cuda()
    np.testing.assert_array_equal(dev.tohost(dev.materialize(t.permute(0, 2, 1))), np.ascontiguousarray(a.transpose(0, 2, 1)))
    np.testing.assert_array_equal(dev.tohost(dev.materialize(t.permute(2, 1, 0))), np.ascontiguousarray(a.transpose(2, 1, 0)))
    np.testing.assert_array_equal(dev.tohost(dev.materialize(t[:, :, 1:1024])), a[:, :, 1:1024])   # rows, starts unaligned
    np.testing.assert_array_equal(dev.tohost(dev.materialize(t[:, 3:, :1028])), a[:, 3:, :1028])   # rows, 16-byte lanes
    np.testing.assert_array_equal(dev.tohost(dev.flip(t, [2])), a[:, :, ::-1])
    f = torch.from_numpy(a.astype(np.float32)).cuda()
    np.testing.assert_array_equal(dev.tohost(dev.materialize(f.permute(1, 2, 0))), np.ascontiguousarray(a.astype(np.float32).transpose(1, 2, 0)))
