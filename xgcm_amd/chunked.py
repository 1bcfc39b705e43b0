"""Chunked host arrays: the shape a dask / zarr / netCDF reader hands a field over in.

The reference walks the chunks of a dask-backed array with `xr.apply_ufunc(dask="parallelized")` over the broadcast dims
(`xgcm/grid.py:786-818`, `xgcm/grid_ufunc.py:966-984`) and with `dask.array.map_overlap` along a chunked core dim
(`xgcm/grid_ufunc.py:1057-1133`; refused for `inner` / `outer` positions, `:1136-1159`).  Here ANY array that exposes

    .chunks      block lengths per dim, dask's tuple of tuples (or zarr's one chunk shape)
    .shape / .dtype / .ndim
    x[slices]    whose `numpy.asarray(...)` yields the cells (dask computes just the chunks a slice crosses)

is walked block by block by the unlabelled-array layer (`device._blockwise`): a block of the NON-core dims, whole along the
operator's own axis (chunks along it are read together -- what `map_overlap` computes, without the halo exchange), goes
through HBM with its copies overlapped (`streaming.iter_stream`) and lands in a `BlockArray` with the input's chunking.
Nothing is concatenated on the host; `numpy.asarray(result)` / `.values` assembles on demand.  `BlockArray` is also the
30-line container tests build their chunked inputs from (dask is not installable here)."""

from __future__ import annotations

import itertools
from typing import Dict, Iterator, Sequence, Tuple

import numpy as np

Chunks = Tuple[Tuple[int, ...], ...]


def is_chunked(x) -> bool:
    """a chunked CONTAINER (dask / zarr / BlockArray ...), not a plain numpy array or a tensor"""
    if isinstance(x, np.ndarray) or type(x).__module__.split(".")[0] == "torch":
        return False
    return getattr(x, "chunks", None) is not None and hasattr(x, "shape") and hasattr(x, "dtype")


def normalize_chunks(chunks, shape: Sequence[int]) -> Chunks:
    """dask's ((2, 2, 1), (4,)) passes; zarr's chunk SHAPE (2, 4) becomes block lengths per dim"""
    out = []
    for c, n in zip(chunks, shape):
        if isinstance(c, (tuple, list)):
            c = tuple(int(v) for v in c)
            if sum(c) != int(n):
                raise ValueError(f"chunks {c} do not add up to the extent {n}")
            out.append(c)
        else:
            c, n = max(1, int(c)), int(n)
            out.append(tuple([c] * (n // c) + ([n % c] if n % c else [])) if n else (0,))
    if len(out) != len(shape):
        raise ValueError(f"chunks for {len(out)} dims on an array of {len(shape)}")
    return tuple(out)


def bounds(lengths: Sequence[int]):
    """[(lo, hi), ...] of consecutive blocks"""
    edges = np.concatenate([[0], np.cumsum(lengths)]).astype(int)
    return [(int(a), int(b)) for a, b in zip(edges[:-1], edges[1:])]


def block_slices(chunks: Chunks, whole: Sequence[int] = ()) -> Iterator[Tuple[Tuple[int, ...], Tuple[slice, ...]]]:
    """(block index, slices) in C order; dims listed in `whole` are not split (index 0, the full extent)"""
    per_dim = [[(0, sum(c))] if d in whole else bounds(c) for d, c in enumerate(chunks)]
    for idx in itertools.product(*[range(len(b)) for b in per_dim]):
        yield idx, tuple(slice(*per_dim[d][i]) for d, i in enumerate(idx))


def read_ahead(x, slices, workers: int = None):
    """`numpy.asarray(x[sl])` for every `sl` of `slices`, IN ORDER, the next few fetched by helper threads while the caller
    works on the current one.  Fetching a block of a chunked container is file reads and decompression (zarr: 0.6 - 2 GB/s per
    core; a dask graph: whatever it computes) -- far below what the PCIe link takes (~50 GB/s), and both release the GIL
    (zlib / ctypes / numpy), so several blocks are fetched side by side; at most `workers` blocks are held ahead of the caller.
    `XG_READ_AHEAD` sets the number of helper threads (default: min(8, cores); 0: fetch in the caller's thread)."""
    import os

    slices = list(slices)
    if workers is None:
        workers = int(os.environ.get("XG_READ_AHEAD", min(8, os.cpu_count() or 1)))
    if workers <= 0 or len(slices) <= 1:
        for sl in slices:
            yield np.asarray(x[sl])
        return
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=workers, thread_name_prefix="xg-read") as pool:
        pending, it = deque(), iter(slices)
        try:
            for sl in it:
                pending.append(pool.submit(lambda sl=sl: np.asarray(x[sl])))
                if len(pending) >= workers:
                    break
            while pending:
                block = pending.popleft().result()
                nxt = next(it, None)
                if nxt is not None:
                    pending.append(pool.submit(lambda sl=nxt: np.asarray(x[sl])))
                yield block
        finally:
            for f in pending:
                f.cancel()


def index_spans(key, shape, what: str = "chunked array"):
    """an index of ints and unit-step slices -> ([(lo, hi) per dim], [dims an int drops]): the indexing a chunked container
    serves (`ds["T"].isel(time=0)` on a store reads exactly the chunks that record crosses)"""
    key = key if isinstance(key, tuple) else (key,)
    if len(key) > len(shape):
        raise IndexError(f"{what}: {len(key)} indices for {len(shape)} dims")
    key = key + (slice(None),) * (len(shape) - len(key))
    spans, squeeze = [], []
    for d, (k, n) in enumerate(zip(key, shape)):
        if isinstance(k, (int, np.integer)):
            k = int(k) + (n if k < 0 else 0)
            if not 0 <= k < n:
                raise IndexError(f"{what}: index {k} is out of bounds for a dim of {n}")
            squeeze.append(d)
            k = slice(k, k + 1)
        if not isinstance(k, slice) or k.step not in (None, 1):
            raise IndexError(f"{what}: integers and unit-step slices only")
        lo, hi, _ = k.indices(n)
        spans.append((lo, max(lo, hi)))
    return spans, squeeze


def pmap(fn, items, workers: int = None):
    """`fn(item)` for every item, by a few helper threads when there are several (chunk files / raw chunks of ONE request:
    reading and inflating release the GIL); results in order.  `XG_READ_AHEAD=0` keeps everything in the caller's thread."""
    import os

    items = list(items)
    if workers is None:  # (more of them than `read_ahead` has: nothing is held ahead here, the items are one request's own pieces)
        workers = int(os.environ["XG_READ_AHEAD"]) if "XG_READ_AHEAD" in os.environ else min(32, os.cpu_count() or 1)
    if workers <= 1 or len(items) <= 1:
        return [fn(it) for it in items]
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=min(workers, len(items)), thread_name_prefix="xg-chunk") as pool:
        return list(pool.map(fn, items))


class BlockArray:
    """A host array kept as a grid of numpy blocks.  `.chunks` as dask's; slicing (unit-step slices) assembles just the blocks a
    slice crosses; `numpy.asarray` the whole."""

    def __init__(self, blocks: Dict[Tuple[int, ...], np.ndarray], chunks: Chunks, dtype):
        self.blocks, self.chunks, self.dtype = blocks, tuple(tuple(int(v) for v in c) for c in chunks), np.dtype(dtype)
        self.shape = tuple(sum(c) for c in self.chunks)
        self.ndim = len(self.shape)

    @classmethod
    def from_array(cls, a, chunks) -> "BlockArray":
        a = np.asarray(a)
        ch = normalize_chunks(chunks, a.shape)
        return cls({idx: a[sl] for idx, sl in block_slices(ch)}, ch, a.dtype)

    @property
    def numblocks(self) -> Tuple[int, ...]:
        return tuple(len(c) for c in self.chunks)

    @property
    def nbytes(self) -> int:
        return int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize

    def __getitem__(self, key) -> np.ndarray:
        key = key if isinstance(key, tuple) else (key,)
        key = key + (slice(None),) * (self.ndim - len(key))
        spans, squeeze = [], []
        for d, (k, n) in enumerate(zip(key, self.shape)):
            if isinstance(k, (int, np.integer)):  # x[i]: that one index, the dim dropped
                k = int(k) + (n if k < 0 else 0)
                if not 0 <= k < n:
                    raise IndexError(f"index {k} is out of bounds for a dim of {n}")
                squeeze.append(d)
                k = slice(k, k + 1)
            if not isinstance(k, slice) or k.step not in (None, 1):
                raise IndexError("BlockArray: integers and unit-step slices only")
            lo, hi, _ = k.indices(n)
            spans.append((lo, max(lo, hi)))
        per_dim = [bounds(c) for c in self.chunks]
        hit = [[i for i, (a, b) in enumerate(pd) if a < hi and b > lo] for pd, (lo, hi) in zip(per_dim, spans)]
        if not squeeze and all(len(h) == 1 and per_dim[d][h[0]] == spans[d] for d, h in enumerate(hit)):
            # exactly one whole block: the block itself, no copy (what the block walk asks for: a 345 MB block assembled by a
            # single-threaded copy cost more than its trip through the GPU -- 5.2 GB through `diff` 0.97 s instead of 0.3)
            return self.blocks[tuple(h[0] for h in hit)]
        out = np.empty([b - a for a, b in spans], dtype=self.dtype)
        for idx in itertools.product(*hit):
            src, dst = [], []
            for d, i in enumerate(idx):
                a, b = per_dim[d][i]
                lo, hi = max(a, spans[d][0]), min(b, spans[d][1])
                src.append(slice(lo - a, hi - a))
                dst.append(slice(lo - spans[d][0], hi - spans[d][0]))
            out[tuple(dst)] = self.blocks[idx][tuple(src)]
        return out.reshape([n for d, n in enumerate(out.shape) if d not in squeeze]) if squeeze else out

    def __array__(self, dtype=None, copy=None):
        a = self[(slice(None),) * self.ndim]
        return a if dtype is None else a.astype(dtype)

    def __repr__(self) -> str:
        return f"BlockArray(shape={self.shape}, dtype={self.dtype}, chunks={self.chunks})"

    def to_dask(self):
        """the same blocks as ONE dask array (no copy, nothing computed): how a chunked result leaves for xarray where dask
        exists (`labeled.to_xarray`) -- the form the reference's `dask="parallelized"` result has (xgcm/grid.py:786-818)"""
        import dask.array as dsa

        nested = np.empty(self.numblocks, dtype=object)
        for idx, blk in self.blocks.items():
            nested[idx] = dsa.from_array(blk, chunks=blk.shape)
        return dsa.block(nested.tolist())


def rechunk_blocks(result_blocks: Dict[Tuple[int, ...], np.ndarray], chunks: Chunks, axis: int, lengths: Sequence[int]):
    """blocks that are whole along `axis` cut into `lengths` there (views): the operator's axis keeps the input's chunks when
    its length did not change (reference: `true_chunksizes`, grid_ufunc.py:1098-1104)"""
    out = {}
    cuts = bounds(lengths)
    for idx, blk in result_blocks.items():
        for j, (a, b) in enumerate(cuts):
            sl = [slice(None)] * blk.ndim
            sl[axis] = slice(a, b)
            out[idx[:axis] + (j,) + idx[axis + 1:]] = blk[tuple(sl)]
    return out, chunks[:axis] + (tuple(int(v) for v in lengths),) + chunks[axis + 1:]


class ExpandedView:
    """A chunked array seen with its dims in another order and / or extra length-1 dims (`x.T[:, None, :]` of a container that
    cannot be indexed so): what the labelled layer's name-based broadcasting hands the block walk.  `where[d]` = the base's
    dim behind view dim d, or None for a new dim."""

    def __init__(self, base, where: Sequence):
        self.base, self.where = base, tuple(where)
        own = normalize_chunks(base.chunks, base.shape)
        self.chunks = tuple((1,) if w is None else own[w] for w in self.where)
        self.shape = tuple(sum(c) for c in self.chunks)
        self.ndim, self.dtype = len(self.shape), base.dtype

    def __getitem__(self, key) -> np.ndarray:
        key = key if isinstance(key, tuple) else (key,)
        key = key + (slice(None),) * (self.ndim - len(key))
        inner = [slice(None)] * len(self.base.shape)
        for k, w in zip(key, self.where):
            if w is not None:
                inner[w] = k
        part = np.asarray(self.base[tuple(inner)])
        order = [w for w in self.where if w is not None]
        if order != sorted(order):  # the view's dims in another order than the base's: a transposed VIEW of the block
            part = np.transpose(part, order)
        out = part[tuple(slice(None) if w is not None else np.newaxis for w in self.where)]
        empty = [d for d, (k, w) in enumerate(zip(key, self.where)) if w is None and len(range(*k.indices(1))) == 0]
        return out[tuple(slice(0, 0) if d in empty else slice(None) for d in range(self.ndim))] if empty else out

    def __array__(self, dtype=None, copy=None):
        a = self[(slice(None),) * self.ndim]
        return a if dtype is None else a.astype(dtype)
