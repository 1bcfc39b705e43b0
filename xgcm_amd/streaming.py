"""Host <-> HBM record streaming (first half: out-of-core inputs).

The reference reaches larger-than-memory datasets through dask: one chunk at a time is loaded,
processed by numpy and written back (`xgcm/grid.py:786-818`).  The accelerator analogue is a
pipeline over the record (outermost, e.g. time) axis of a HOST array:

    copy-in stream:   H2D of record block k+1      |  three HIP streams, ordered by events, so the
    compute stream:   Grid operators on block k    |  PCIe transfers in both directions overlap with
    copy-out stream:  D2H of the result of k-1     |  the kernels (which take ~2 % of the time)

so a record batch that does not fit the 288 GB of HBM -- or that simply lives on the host -- is
processed at the PCIe rate instead of (PCIe in) + (kernel) + (PCIe out) + pageable-copy overheads.
Host memory is page-locked in place (`hipHostRegister` through torch's runtime handle), record block by record block
in a helper thread that runs AHEAD of the copies -- page-locking 27 GB up front cost as much as moving half of it
(tools/pcie_probe.py) -- and blocks that cannot be locked (a read-only memory map) take the runtime's pageable copies,
the copy back to the host running in a worker thread so that both directions still overlap.  PyTorch provides
streams, events and the caching allocator here; the arithmetic is the HIP library's as everywhere else.
"""

from __future__ import annotations

import mmap
import threading
from concurrent.futures import ThreadPoolExecutor
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


def _lock(ptr: int, nbytes: int) -> bool:
    """page-lock host memory in place (xg_pin_host); False if the range cannot be locked"""
    return nbytes > 0 and _hip.load().xg_pin_host(ptr, nbytes) == 0


def _unlock(ptr: int) -> None:
    _hip.load().xg_unpin_host(ptr)


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


class _BlockPinner:
    """Page-lock the record blocks of a C-contiguous host array one after the other in a helper thread, so that the
    pipeline's first copy starts after ONE block has been locked instead of the whole array.  Block ranges are cut at
    page boundaries (a page two blocks share belongs to the earlier one: blocks are locked in order, so by the time
    block k is copied every page it touches is locked).  An asynchronous copy must stay inside ONE locked range, so a
    block is moved in `pieces(k)`: its first bytes up to the page boundary (they lie in the previous block's range),
    then the rest.  Arrays under 256 MB are locked whole.  `wait(k)` -> True if block k is page-locked."""

    WHOLE_BELOW = 256 << 20
    THREADS = 4

    def __init__(self, arr: np.ndarray, blocks: List[Tuple[int, int]], enabled: bool = True, prefault: bool = False):
        """`prefault` (an OUTPUT array, every cell of which is about to be overwritten): write one element per page
        before locking a block -- page-locking never-touched memory takes its page faults inside the driver call, one
        thread at a time; touched here, they are taken by THREADS threads at once"""
        self.tensor = _as_tensor(arr)
        self.flat = self.tensor.reshape(-1)  # (C-contiguous: a view)
        self.ready = [threading.Event() for _ in blocks]
        self.pinned = [False] * len(blocks)
        self._cancelled = False
        self._ranges: List[int] = []
        self._threads: List[threading.Thread] = []
        item = arr.itemsize
        per = (arr.size // arr.shape[0]) if arr.ndim and arr.shape[0] else 1  # elements per record
        self._pieces = [[(a * per, b * per)] for a, b in blocks]
        if not enabled or not blocks or arr.size == 0:
            for ev in self.ready:
                ev.set()
            return
        base = self.tensor.data_ptr()
        row = per * item
        page = mmap.PAGESIZE
        up = lambda p: (p + page - 1) // page * page  # noqa: E731
        if arr.nbytes < self.WHOLE_BELOW or len(blocks) == 1 or base % item:
            spans = [(base, arr.nbytes)] + [(0, 0)] * (len(blocks) - 1)
        else:
            spans = []
            for k, (a, b) in enumerate(blocks):
                lo = base + a * row if k == 0 else up(base + a * row)
                hi = base + b * row if k == len(blocks) - 1 else up(base + b * row)
                spans.append((lo, max(hi - lo, 0)))
                cut = (lo - base) // item  # first element of the block that lies in its OWN range
                if k and a * per < cut < b * per:
                    self._pieces[k] = [(a * per, cut), (cut, b * per)]
        dev = torch.cuda.current_device()
        lock = threading.Lock()
        todo = iter(range(len(spans)))
        failed = [False]

        def work():
            # a few threads lock consecutive blocks at once: page-locking never-touched memory (an output array fresh from
            # numpy.empty) is bound by the page faults of the locking thread, 25 GB/s -- under the copy rate
            torch.cuda.set_device(dev)
            while True:
                with lock:
                    k = next(todo, None)
                if k is None:
                    return
                lo, nbytes = spans[k]
                ok = not failed[0] and not self._cancelled  # (an aborted stream: the remaining blocks stay pageable)
                try:
                    if ok and nbytes and prefault:
                        e0 = (lo - base) // item
                        self.flat[e0:e0 + nbytes // item:max(page // item, 1)] = 0
                    if ok and nbytes:
                        ok = _lock(lo, nbytes)
                        if ok:
                            with lock:
                                self._ranges.append(lo)
                except Exception:  # (a read-only output array ...): never leave the pipeline waiting for this block
                    ok = False
                if not ok:
                    failed[0] = True  # (a read-only mapping ...): the remaining blocks stay pageable
                self.pinned[k] = ok
                self.ready[k].set()

        nthreads = 1 if spans[0][1] == arr.nbytes else min(self.THREADS, len(spans))
        self._threads = [threading.Thread(target=work, name="xgcm-amd-pin", daemon=True) for _ in range(nthreads)]
        for t in self._threads:
            t.start()

    def pieces(self, k: int) -> List[Tuple[int, int]]:
        """[first, last) flat element ranges of block k, each inside one page-locked range"""
        return self._pieces[k]

    def wait(self, k: int) -> bool:
        """block k may be copied: its own range and the one holding its first bytes are done"""
        if k:
            self.ready[k - 1].wait()
        self.ready[k].wait()
        return self.pinned[k]

    def cancel(self) -> None:
        """stop locking further blocks (the stream was aborted); `close` still joins and unlocks what was locked"""
        self._cancelled = True

    def close(self) -> None:
        for t in self._threads:
            t.join()
        self._threads = []
        for lo in self._ranges:
            _unlock(lo)
        self._ranges = []


def _copy_threads(dst: np.ndarray, src: np.ndarray, threads: int = 8) -> None:
    """`numpy.copyto(dst, src)` cut into pieces for a few threads: one core moves ~10 GB/s, a memory-mapped source also
    takes its page faults in parallel (numpy releases the GIL in the copy loop).  Contiguous pairs are cut flat, others
    along their first axis long enough to share."""
    src = np.asarray(src)
    if dst.nbytes < (64 << 20) or dst.ndim == 0 or src.shape != dst.shape:
        np.copyto(dst, src)
        return
    if dst.flags.c_contiguous and src.flags.c_contiguous:
        d, s_, ax = dst.reshape(-1), src.reshape(-1), 0
    else:
        ax = next((i for i, e in enumerate(dst.shape) if e >= threads), None)
        if ax is None:
            np.copyto(dst, src)
            return
        d, s_ = dst, src
    n = d.shape[ax]
    cuts = [n * i // threads for i in range(threads + 1)]
    pre = (slice(None),) * ax

    def part(i):
        sl = pre + (slice(cuts[i], cuts[i + 1]),)
        np.copyto(d[sl], s_[sl])

    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(part, range(threads)))


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
    pin_src = _BlockPinner(src, blocks, register)
    src_t = pin_src.tensor
    pin_out: Optional[_BlockPinner] = None
    out_t: Optional[torch.Tensor] = None
    pool = ThreadPoolExecutor(1, thread_name_prefix="xgcm-amd-d2h")
    futs: list = []

    def copy_out(k: int, a: int, b: int, y: torch.Tensor, ev_cmp: "torch.cuda.Event") -> None:
        """worker thread: result of block k back to its place in `out` (an asynchronous copy when that block of `out` is
        page-locked, the runtime's pageable copy otherwise -- either way off the thread that feeds the GPU)"""
        torch.cuda.set_device(dev)
        pin_out.wait(k)
        yf = y.reshape(-1)
        e0 = pin_out.pieces(k)[0][0]
        with torch.cuda.stream(s_out):
            s_out.wait_event(ev_cmp)
            for p0, p1 in pin_out.pieces(k):
                pin_out.flat[p0:p1].copy_(yf[p0 - e0:p1 - e0], non_blocking=True)
        s_out.synchronize()  # `y` is dropped when this returns

    try:
        for k, (a, b) in enumerate(blocks):
            pin_src.wait(k)
            with torch.cuda.stream(s_in):
                x = torch.empty((b - a,) + tuple(src.shape[1:]), dtype=src_t.dtype, device=dev)
                xf, e0 = x.reshape(-1), pin_src.pieces(k)[0][0]
                for p0, p1 in pin_src.pieces(k):  # (a block that could not be locked: pageable copy, blocks here)
                    xf[p0 - e0:p1 - e0].copy_(pin_src.flat[p0:p1], non_blocking=True)
                _swap_on_device(x, swap, s_in)
                ev_in = torch.cuda.Event()
                ev_in.record(s_in)
            s_cmp.wait_event(ev_in)
            x.record_stream(s_cmp)
            y = fn(x)
            if y.shape[0] != b - a:
                raise ValueError("fn must keep the record axis (first dim) of its block")
            if not y.is_contiguous():
                y = y.contiguous()
            ev_cmp = torch.cuda.Event()
            ev_cmp.record(s_cmp)
            if out_t is None:
                if out is None:
                    out = np.empty((n,) + tuple(y.shape[1:]), dtype=np.float32 if y.dtype == torch.float32 else np.float64)
                if not out.flags.c_contiguous or out.shape[0] != n or tuple(out.shape[1:]) != tuple(y.shape[1:]):
                    raise ValueError("`out` must be C-contiguous with the result's shape")
                if not out.flags.writeable:
                    raise ValueError("`out` is read-only")
                if out.dtype != (np.float32 if y.dtype == torch.float32 else np.float64):
                    raise ValueError(f"`out` has dtype {out.dtype}, the result is {str(y.dtype).replace('torch.', '')}")
                pin_out = _BlockPinner(out, blocks, register, prefault=True)
                out_t = pin_out.tensor
            while len(futs) >= 2:  # at most two results parked in HBM
                futs.pop(0).result()
            futs.append(pool.submit(copy_out, k, a, b, y, ev_cmp))
            del x, y
        for f in futs:
            f.result()
        futs = []
        s_in.synchronize()
        _hip.chain_check()  # results are on the host now: report a chained launch that had to be redone
    finally:
        torch.cuda.synchronize(dev)  # (an error path: no copy may still read / write memory that is about to be unlocked)
        for f in futs:
            try:
                f.result()
            except Exception:
                pass
        pool.shutdown(wait=True)
        pin_src.cancel()
        if pin_out is not None:
            pin_out.cancel()
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
def _prefault(a: np.ndarray, threads: int = 4) -> None:
    """first touch of a fresh array by a few threads at once (one write per page): 27 GB/s for one thread, 100 GB/s for
    four on the GPU box (tools/pcie_probe.py)"""
    flat = a.reshape(-1)
    step = max(mmap.PAGESIZE // a.itemsize, 1)
    if flat.size < threads * step * 1024:
        flat[::step] = 0
        return
    cuts = [flat.size * i // threads // step * step for i in range(threads)] + [flat.size]
    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(lambda i: flat[cuts[i]:cuts[i + 1]:step].__setitem__(slice(None), 0), range(threads)))


class _Stage:
    """two rotating page-locked host buffers that grow to the largest block seen.  A buffer is numpy memory, touched by a
    few threads and then page-locked in place: `hipHostMalloc` (torch's pin_memory) hands out 8 GB/s, which for two
    input and two output buffers of a 1.7 GB record was two thirds of the whole stream (tools/pcie_probe.py)."""

    def __init__(self):
        self.buf: List[Optional[torch.Tensor]] = [None, None]
        self._locked: List[Optional[int]] = [None, None]

    def _release(self, slot: int) -> None:
        if self._locked[slot] is not None:
            _unlock(self._locked[slot])
            self._locked[slot] = None
        self.buf[slot] = None

    def get(self, slot: int, shape, dtype) -> torch.Tensor:
        n = int(np.prod(shape))
        b = self.buf[slot]
        if b is None or b.dtype != dtype or b.numel() < n:
            self._release(slot)
            from . import dtypes as _dt

            if dtype not in _dt._TORCH_TO_NUMPY or dtype == torch.bfloat16:
                raise TypeError(f"streaming: blocks / results of dtype {dtype} are not served (numpy has no twin)")
            a = np.empty(max(n, 1), dtype=_dt._TORCH_TO_NUMPY[dtype])  # the result's OWN dtype: int64 / bool results stay what they are
            _prefault(a)
            b = torch.from_numpy(a)
            ok = _lock(b.data_ptr(), a.nbytes)
            if ok:
                self._locked[slot] = b.data_ptr()
            else:
                b = torch.empty(max(n, 1), dtype=dtype).pin_memory()
            self.buf[slot] = b
        return b[:n].view(tuple(shape))

    def close(self) -> None:
        self._release(0)
        self._release(1)


def iter_stream(fn: Callable[[torch.Tensor], torch.Tensor], blocks: Iterable, copy: bool = True,
                mask_value: Optional[float] = None, in_place: bool = True) -> Iterator[np.ndarray]:
    """Yield `fn(block)` for every host block of `blocks`, in order, as host arrays.

    Ownership of the blocks: a C-contiguous in-memory block of 8 MB or more is page-locked WHERE IT IS and copied to the
    GPU straight from the caller's memory (`in_place=True`, no staging copy).  The generator is not asked for its next
    block before that copy has completed, so a reader that refills ONE buffer (`f.readinto(buf); yield buf`) is safe --
    it only loses the overlap of its read with that copy.  `in_place=False` stages every block through the streamer's
    own page-locked buffers (a host copy per block; the reader's memory is never locked).

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

    locked: List[Optional[Tuple[int, np.ndarray]]] = [None, None]  # source blocks page-locked in place, per slot
    regs: dict = {}  # page-locked range -> [bytes, slots using it]: a reader that reuses one buffer locks it once
    direct = [bool(in_place)]
    ends: List[Optional[torch.Tensor]] = [None, None]  # page-locked scrap for the sub-page ends of a block locked in place
    in_flight: List[Optional[torch.cuda.Event]] = [None]  # H2D reading the CALLER's memory (a block locked in place)

    def release(slot: int) -> None:
        if locked[slot] is not None:
            alo = locked[slot][0]
            regs[alo][1] -= 1
            if regs[alo][1] == 0:
                _unlock(alo)
                del regs[alo]
            locked[slot] = None

    def lock_range(alo: int, nbytes: int) -> bool:
        """page-lock [alo, alo + nbytes) once, however many slots copy from it"""
        hit = regs.get(alo)
        if hit is not None and hit[0] >= nbytes:
            hit[1] += 1
            return True
        if hit is None and _lock(alo, nbytes):
            regs[alo] = [nbytes, 1]
            return True
        return False

    def before_next_block() -> None:
        """called before the generator is asked for another block: a copy that still reads the caller's memory finishes
        first (the reader may be about to overwrite that buffer)"""
        if in_flight[0] is not None:
            in_flight[0].synchronize()
            in_flight[0] = None

    def upload(k: int, block) -> Tuple[torch.Tensor, torch.cuda.Event]:
        a, swap = _native_view(np.asarray(block))
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float64)
        slot = k % 2
        if in_free[slot] is not None:
            in_free[slot].synchronize()
        release(slot)
        # a block that already sits in memory (a numpy chunk handed over by a reader) is page-locked where it is and
        # copied straight from there; one that cannot be locked (a file mapping: its pages are read here) is staged
        host, cut = None, None
        if direct[0] and a.nbytes >= (8 << 20) and a.flags.c_contiguous and not isinstance(block, np.memmap):
            # whole pages only: consecutive blocks cut from one array share their boundary pages, and a page is locked once
            t = _as_tensor(a)
            page, lo = mmap.PAGESIZE, t.data_ptr()
            alo, ahi = (lo + page - 1) // page * page, (lo + a.nbytes) // page * page
            if lo % a.itemsize == 0 and lock_range(alo, ahi - alo):
                locked[slot] = (alo, a)  # (the array stays referenced until its copy is done)
                host, cut = t, ((alo - lo) // a.itemsize, (ahi - lo) // a.itemsize)
            else:
                direct[0] = False  # memory that cannot be locked (or oddly aligned): stage from here on
        if host is None:
            host = st_in.get(slot, a.shape, torch.float32 if a.dtype == np.float32 else torch.float64)
            _copy_threads(host.numpy(), a)  # the READ of the block (disk / page cache -> pinned memory), any strides
        with torch.cuda.stream(s_in):
            if cut is None:
                x = host.to(dev, non_blocking=True)
            else:  # the locked middle asynchronously, the two sub-page ends through the pageable path
                x = torch.empty(tuple(a.shape), dtype=host.dtype, device=dev)
                xf, hf = x.reshape(-1), host.reshape(-1)
                xf[cut[0]:cut[1]].copy_(hf[cut[0]:cut[1]], non_blocking=True)
                nh, nt = cut[0], hf.numel() - cut[1]
                if nh or nt:  # (through a page-locked scrap buffer: a pageable copy would wait for the stream)
                    if ends[slot] is None or ends[slot].dtype != host.dtype:
                        ends[slot] = torch.empty(2 * mmap.PAGESIZE // a.itemsize, dtype=host.dtype).pin_memory()
                    e = ends[slot]
                    e[:nh].copy_(hf[:nh])
                    e[nh:nh + nt].copy_(hf[cut[1]:])
                    xf[:nh].copy_(e[:nh], non_blocking=True)
                    xf[cut[1]:].copy_(e[nh:nh + nt], non_blocking=True)
            _swap_on_device(x, swap, s_in)  # a big-endian block: raw bytes came over, the order is reversed in HBM
            if mask_value is not None and x.numel():
                _hip.check(_hip.load().xg_mask_value(x.data_ptr(), x.numel(), x.element_size(), float(mask_value),
                                                     s_in.cuda_stream))
            ev = torch.cuda.Event()
            ev.record(s_in)
        in_free[slot] = ev
        if cut is not None:
            in_flight[0] = ev
        return x, ev

    def finish(slot: int) -> np.ndarray:
        host, ev = out_ready[slot]
        ev.synchronize()
        out_ready[slot] = None
        if not copy:
            return host.numpy()
        res = np.empty(tuple(host.shape), dtype=host.numpy().dtype)
        _copy_threads(res, host.numpy())
        return res

    try:
        yield from _pump(fn, blocks, upload, finish, out_ready, st_out, s_cmp, s_out, before_next_block)
        s_in.synchronize()
        s_out.synchronize()
        _hip.chain_check()
    finally:
        torch.cuda.synchronize(dev)  # (an abandoned generator: no copy may still target the staging buffers)
        release(0)
        release(1)
        st_in.close()
        st_out.close()


def _pump(fn, blocks, upload, finish, out_ready, st_out, s_cmp, s_out, before_next=lambda: None):
    it = iter(blocks)
    nxt = next(it, None)
    ahead = upload(0, nxt) if nxt is not None else None
    k = 0
    while ahead is not None:
        x, ev_in = ahead
        before_next()                              # (a copy straight from the reader's buffer completes before it refills it)
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
