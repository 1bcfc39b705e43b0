"""Host <-> HBM record streaming (first half: out-of-core inputs).

The reference reaches larger-than-memory datasets through dask: one chunk at a time is loaded,
processed by numpy and written back (`xgcm/grid.py:786-818`).  The accelerator analogue is a
pipeline over the record (outermost, e.g. time) axis of a HOST array:

    copy-in stream:   H2D of record block k+1      |  three HIP streams, ordered by events, so the
    compute stream:   Grid operators on block k    |  PCIe transfers in both directions overlap with
    copy-out stream:  D2H of the result of k-1     |  the kernels (which take ~2 % of the time)

so a record batch that does not fit the 288 GB of HBM -- or that simply lives on the host -- is
processed at the PCIe rate instead of (PCIe in) + (kernel) + (PCIe out) + pageable-copy overheads.
Host memory is page-locked in place (`hipHostRegister` through torch's runtime handle) when
possible, else staged through pinned buffers.  PyTorch provides streams, events and the caching
allocator here; the arithmetic is the HIP library's as everywhere else.
"""

from __future__ import annotations

from typing import Callable, Iterable, Iterator, List, Optional, Tuple

import numpy as np
import torch

from . import _hip
from .labeled import DataArray

__all__ = ["record_blocks", "stream_records", "stream_apply", "stream_blocks", "iter_stream"]


def _native_view(a: np.ndarray) -> Tuple[np.ndarray, int]:
    """(the same bytes seen as NATIVE float32 / float64, element size if the byte order must be reversed on the GPU else 0).

    Files MITgcm writes (MDS, NetCDF-3: `xgcm_amd.io`) hold big-endian numbers.  Their blocks cross PCIe as raw bytes and
    are swapped in HBM by `xg_bswap`; no host core converts them."""
    dt = a.dtype
    if dt.kind == "f" and dt.itemsize in (4, 8) and not dt.isnative:
        return a.view(dt.newbyteorder("=")), dt.itemsize
    return a, 0


def _swap_on_device(x: torch.Tensor, itemsize: int, stream: "torch.cuda.Stream") -> None:
    if itemsize:
        _hip.check(_hip.load().xg_bswap(x.data_ptr(), x.numel(), itemsize, stream.cuda_stream))


def record_blocks(n_records: int, block: int) -> List[Tuple[int, int]]:
    """[start, stop) of consecutive record blocks (the last one may be short)."""
    if block < 1:
        raise ValueError("block must be >= 1")
    return [(s, min(s + block, n_records)) for s in range(0, n_records, block)]


def _as_tensor(arr: np.ndarray) -> torch.Tensor:
    """torch view of a host array that is only ever READ (a read-only memory map is fine: silence torch's warning)"""
    import warnings

    with warnings.catch_warnings():
        warnings.filterwarnings("ignore", message="The given NumPy array is not writable")
        return torch.from_numpy(arr)


class _Pinned:
    """Page-lock a numpy array in place for the lifetime of the object (fallback: not pinned)."""

    def __init__(self, arr: np.ndarray):
        self.arr = arr
        self.tensor = _as_tensor(arr)
        self.registered = False
        try:
            rt = torch.cuda.cudart()
            err = rt.cudaHostRegister(self.tensor.data_ptr(), self.tensor.numel() * self.tensor.element_size(), 0)
            self.registered = int(err) == 0
        except Exception:
            self.registered = False

    def close(self) -> None:
        if self.registered:
            try:
                torch.cuda.cudart().cudaHostUnregister(self.tensor.data_ptr())
            except Exception:
                pass
            self.registered = False


def stream_records(fn: Callable[[torch.Tensor], torch.Tensor], src: np.ndarray, block: int = 1,
                   out: Optional[np.ndarray] = None, register: bool = True) -> np.ndarray:
    """Apply `fn` (HBM tensor of `block` records -> HBM tensor with the same leading length) to a
    C-contiguous host array record block by record block, with H2D / compute / D2H overlapped."""
    if not torch.cuda.is_available():
        raise RuntimeError("xgcm_amd.streaming needs a GPU (there is no CPU fallback)")
    src, swap = _native_view(np.asarray(src))  # (a big-endian array, e.g. a memory-mapped MDS file: swapped in HBM)
    if not src.flags.c_contiguous:
        raise ValueError("the host array must be C-contiguous (records along the first axis)")
    n = src.shape[0]
    blocks = record_blocks(n, block)
    dev = torch.device("cuda", torch.cuda.current_device())
    s_in, s_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    s_cmp = torch.cuda.current_stream(dev)
    pin_src = _Pinned(src) if register else None
    src_t = pin_src.tensor if pin_src is not None else _as_tensor(src)
    # page-locking in place failed (or was declined): stage through two pinned buffers per direction
    stage_in = None
    if pin_src is None or not pin_src.registered:
        stage_in = [torch.empty((block,) + src.shape[1:], dtype=src_t.dtype).pin_memory() for _ in range(2)]
    in_done: List[Optional[torch.cuda.Event]] = [None, None]
    pin_out: Optional[_Pinned] = None
    out_t: Optional[torch.Tensor] = None
    stage_out: Optional[List[torch.Tensor]] = None
    pending: List[Optional[Tuple[int, int, torch.cuda.Event]]] = [None, None]

    def drain(slot: int) -> None:
        """finish the D2H that went into staging buffer `slot` and move it to its place in `out`"""
        if stage_out is None or pending[slot] is None:
            return
        a0, b0, ev = pending[slot]
        ev.synchronize()
        out_t[a0:b0].copy_(stage_out[slot][: b0 - a0])
        pending[slot] = None

    try:
        for k, (a, b) in enumerate(blocks):
            slot = k % 2
            with torch.cuda.stream(s_in):
                if stage_in is not None:
                    if in_done[slot] is not None:
                        in_done[slot].synchronize()      # the H2D that last read this staging buffer is done
                    stage_in[slot][: b - a].copy_(src_t[a:b])
                    host_block = stage_in[slot][: b - a]
                else:
                    host_block = src_t[a:b]
                x = host_block.to(dev, non_blocking=True)
                _swap_on_device(x, swap, s_in)
                ev_in = torch.cuda.Event()
                ev_in.record(s_in)
                in_done[slot] = ev_in
            s_cmp.wait_event(ev_in)
            x.record_stream(s_cmp)
            y = fn(x)
            if y.shape[0] != b - a:
                raise ValueError("fn must keep the record axis (first dim) of its block")
            ev_cmp = torch.cuda.Event()
            ev_cmp.record(s_cmp)
            if out_t is None:
                if out is None:
                    out = np.empty((n,) + tuple(y.shape[1:]), dtype=np.float32 if y.dtype == torch.float32 else np.float64)
                if not out.flags.c_contiguous or out.shape[0] != n or tuple(out.shape[1:]) != tuple(y.shape[1:]):
                    raise ValueError("`out` must be C-contiguous with the result's shape")
                pin_out = _Pinned(out) if register else None
                out_t = pin_out.tensor if pin_out is not None else torch.from_numpy(out)
                if pin_out is None or not pin_out.registered:
                    stage_out = [torch.empty((block,) + tuple(y.shape[1:]), dtype=y.dtype).pin_memory() for _ in range(2)]
            drain(slot)
            with torch.cuda.stream(s_out):
                s_out.wait_event(ev_cmp)
                y.record_stream(s_out)
                if stage_out is not None:
                    stage_out[slot][: b - a].copy_(y, non_blocking=True)
                else:
                    out_t[a:b].copy_(y, non_blocking=True)
                ev_out = torch.cuda.Event()
                ev_out.record(s_out)
            pending[slot] = (a, b, ev_out)
            del x, y
        drain(0)
        drain(1)
        s_out.synchronize()
        s_in.synchronize()
        _hip.chain_check()  # results are on the host now: report a chained launch that had to be redone
    finally:
        if pin_src is not None:
            pin_src.close()
        if pin_out is not None:
            pin_out.close()
    return out


# ------------------------------------------------------------------------------------------------------
# Chunk-iterator input: the shape a dask / zarr / netCDF reader hands data over in.  The reference walks the
# chunks of a dask array with `apply_ufunc(dask="parallelized")` (xgcm/grid.py:786-818); here ANY iterable of
# host record blocks -- a generator that memory-maps one file per block, `dask_array.blocks`, a zarr array
# sliced along time -- is pulled one block ahead of the GPU: while block k is computed, block k+1 is read
# (page faults of a memory map included), staged into page-locked memory and copied in, and the result of
# block k-1 is copied out.  Blocks may differ in length along the record axis (ragged last chunk).
# ------------------------------------------------------------------------------------------------------
class _Stage:
    """two rotating page-locked host buffers that grow to the largest block seen"""

    def __init__(self):
        self.buf: List[Optional[torch.Tensor]] = [None, None]

    def get(self, slot: int, shape, dtype) -> torch.Tensor:
        n = int(np.prod(shape))
        b = self.buf[slot]
        if b is None or b.dtype != dtype or b.numel() < n:
            self.buf[slot] = b = torch.empty(max(n, 1), dtype=dtype).pin_memory()
        return b[:n].view(tuple(shape))


def iter_stream(fn: Callable[[torch.Tensor], torch.Tensor], blocks: Iterable, copy: bool = True,
                mask_value: Optional[float] = None) -> Iterator[np.ndarray]:
    """Yield `fn(block)` for every host block of `blocks`, in order, as host arrays.

    `mask_value`: cells of the uploaded block equal to it become NaN before `fn` sees the block (xg_mask_value, in HBM
    after the byte swap) -- the `_FillValue` / `missing_value` of a file variable, which xarray's decoding masks for the
    reference (`xgcm_amd.io.netcdf_missing_value`).

    `blocks`: any iterable of array-likes (float32 / float64, records along the first axis; anything
    `numpy.asarray` accepts, e.g. `numpy.memmap`).  `fn`: HBM tensor -> HBM tensor (e.g. a closure over
    `Grid` operators).  H2D of block k+1, `fn` on block k and D2H of block k-1 overlap on three HIP streams.
    `copy=False` yields views of the staging buffers that are valid only until the next `next()`."""
    if not torch.cuda.is_available():
        raise RuntimeError("xgcm_amd.streaming needs a GPU (there is no CPU fallback)")
    dev = torch.device("cuda", torch.cuda.current_device())
    s_in, s_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    s_cmp = torch.cuda.current_stream(dev)
    st_in, st_out = _Stage(), _Stage()
    in_free: List[Optional[torch.cuda.Event]] = [None, None]   # H2D that last read staging slot i
    out_ready: List[Optional[Tuple[torch.Tensor, torch.cuda.Event]]] = [None, None]

    def upload(k: int, block) -> Tuple[torch.Tensor, torch.cuda.Event]:
        a, swap = _native_view(np.asarray(block))
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float64)
        slot = k % 2
        if in_free[slot] is not None:
            in_free[slot].synchronize()
        host = st_in.get(slot, a.shape, torch.float32 if a.dtype == np.float32 else torch.float64)
        np.copyto(host.numpy(), a)  # the READ of the block (disk / page cache -> pinned memory), any strides
        with torch.cuda.stream(s_in):
            x = host.to(dev, non_blocking=True)
            _swap_on_device(x, swap, s_in)  # a big-endian block: raw bytes came over, the order is reversed in HBM
            if mask_value is not None and x.numel():
                _hip.check(_hip.load().xg_mask_value(x.data_ptr(), x.numel(), x.element_size(), float(mask_value),
                                                     s_in.cuda_stream))
            ev = torch.cuda.Event()
            ev.record(s_in)
        in_free[slot] = ev
        return x, ev

    def finish(slot: int) -> np.ndarray:
        host, ev = out_ready[slot]
        ev.synchronize()
        out_ready[slot] = None
        return np.array(host.numpy(), copy=True) if copy else host.numpy()

    it = iter(blocks)
    nxt = next(it, None)
    ahead = upload(0, nxt) if nxt is not None else None
    k = 0
    while ahead is not None:
        x, ev_in = ahead
        nxt = next(it, None)                       # read + stage + enqueue the NEXT block before computing this one
        ahead = upload(k + 1, nxt) if nxt is not None else None
        s_cmp.wait_event(ev_in)
        x.record_stream(s_cmp)
        y = fn(x)
        if not isinstance(y, torch.Tensor):
            raise TypeError("fn must return an HBM tensor")
        ev_cmp = torch.cuda.Event()
        ev_cmp.record(s_cmp)
        slot = k % 2
        if out_ready[slot] is not None:            # result of block k-2 still parked there: hand it over first
            yield finish(slot)
        host_out = st_out.get(slot, tuple(y.shape), y.dtype)
        with torch.cuda.stream(s_out):
            s_out.wait_event(ev_cmp)
            y.record_stream(s_out)
            host_out.copy_(y, non_blocking=True)
            ev_out = torch.cuda.Event()
            ev_out.record(s_out)
        out_ready[slot] = (host_out, ev_out)
        del x, y
        k += 1
        other = k % 2
        if out_ready[other] is not None and ahead is None:  # tail: nothing left to overlap with
            yield finish(other)
    for slot in ((k % 2), ((k + 1) % 2)):
        if out_ready[slot] is not None:
            yield finish(slot)
    s_in.synchronize()
    s_out.synchronize()
    _hip.chain_check()


def stream_blocks(fn: Callable[[torch.Tensor], torch.Tensor], blocks: Iterable,
                  sink: Optional[Callable[[int, np.ndarray], None]] = None,
                  mask_value: Optional[float] = None) -> Optional[np.ndarray]:
    """Run `iter_stream` to completion.  With `sink(k, result_block)` every result is handed over as it arrives
    (write it to a file, a zarr store ...) and None is returned; without it the results are concatenated along
    the record axis and returned."""
    parts = []
    for k, res in enumerate(iter_stream(fn, blocks, copy=sink is None, mask_value=mask_value)):
        if sink is not None:
            sink(k, res)
        else:
            parts.append(res)
    if sink is not None:
        return None
    if not parts:
        return np.empty((0,))
    return np.concatenate(parts, axis=0)


def stream_apply(fn: Callable[[DataArray], DataArray], da: DataArray, record_dim: Optional[str] = None,
                 block: int = 1) -> DataArray:
    """Labelled form: `fn` maps a device-resident block `DataArray` (same dims as `da`, `block`
    records along the FIRST dim) to a `DataArray`; the host result keeps `fn`'s dims and name."""
    if record_dim is None:
        record_dim = da.dims[0]
    if da.dims[0] != record_dim:
        raise ValueError(f"the record dim {record_dim!r} must be the first (slowest) dim of the array, got {da.dims}")
    meta = {}

    def on_block(x: torch.Tensor) -> torch.Tensor:
        res = fn(DataArray(x, da.dims, name=da.name))
        if res.dims[0] != record_dim:
            raise ValueError("fn must keep the record dim first")
        meta.setdefault("dims", res.dims)
        meta.setdefault("name", res.name)
        return res.data if isinstance(res.data, torch.Tensor) else torch.as_tensor(res.data, device=x.device)

    host = np.ascontiguousarray(da.values)
    out = stream_records(on_block, host, block=block)
    coords = {k: c for k, c in da.coords.items() if all(d in meta["dims"] for d in c.dims) and
              all(c.sizes[d] == out.shape[meta["dims"].index(d)] for d in c.dims)}
    return DataArray(out, meta["dims"], coords=coords, name=meta["name"])
