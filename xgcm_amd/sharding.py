"""Record-axis sharding of the hot path over the GPUs of one node (one process per GPU).

Every operator is 1-D along one core axis and independent along all others (the property the
reference's `dask="parallelized"` path relies on, xgcm/grid.py:786-789), so a field shards by a
contiguous block split of its outermost non-core axis with NO data-path collective: outputs
stay sharded like a dask array chunked along `time`.  `torch.distributed` (backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in CPU tests) is used only for barriers and for reducing
scalars: the max-over-ranks time and order-independent checksums.
"""

from __future__ import annotations

from typing import Optional, Sequence, Tuple


def shard_bounds(n_units: int, world: int, rank: int) -> Tuple[int, int]:
    """[start, stop) of the contiguous block of `n_units` owned by `rank`; the first
    `n_units % world` ranks own one extra unit (90 levels on 8 GPUs -> 12,12,11,11,11,11,11,11)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(int(n_units), world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def reduce_max(value: float, dist=None, device: Optional[str] = None) -> float:
    """max over ranks of a python float (identity without a process group)."""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(value: float, dist=None, device: Optional[str] = None) -> float:
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def whole_job_throughput(local_units: float, local_seconds: float, dist=None, device: Optional[str] = None) -> Tuple[float, float]:
    """(units of all ranks / max-over-ranks seconds, that max) -- the bench contract's aggregate."""
    total = reduce_sum(local_units, dist, device)
    tmax = reduce_max(local_seconds, dist, device)
    return total / tmax, tmax


# ----------------------------------------------------------------------------------------------
# Rank launcher: `script --gpus N` must mean N ranks whoever started it.  Under a launcher
# (torchrun / `python -m torch.distributed.run`, which exports WORLD_SIZE / RANK / LOCAL_RANK) the
# process is ONE of the N ranks; started by hand with `--gpus N` it re-executes itself under
# torch.distributed.run with N ranks on 127.0.0.1.  Fewer visible GPUs than ranks is an error, a
# WORLD_SIZE that disagrees with `--gpus` is an error: no run ever reports a rank count it did not use.
# ----------------------------------------------------------------------------------------------
def _free_port() -> int:
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _backend_from_env(default_gpu: str = "nccl") -> str:
    import os

    import torch

    return os.environ.get("XG_DIST_BACKEND") or (default_gpu if torch.cuda.is_available() else "gloo")


def _visible_gpus() -> int:
    import torch

    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def ensure_ranks(n_gpus: int, script: str, argv: Sequence[str]) -> None:
    """Make sure `n_gpus` ranks exist.  Returns in a process that IS one of the ranks (or the only one);
    otherwise replaces this process by `python -m torch.distributed.run --nproc-per-node n_gpus script argv`
    and exits with its status.  Call before anything touches the GPU."""
    import os
    import subprocess
    import sys

    n_gpus = int(n_gpus)
    if n_gpus < 1:
        raise SystemExit(f"--gpus {n_gpus}: need at least one rank")
    if "WORLD_SIZE" in os.environ:  # already under a launcher: it decides the rank count, we only verify
        world = int(os.environ["WORLD_SIZE"])
        if world != n_gpus:
            raise SystemExit(f"--gpus {n_gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to "
                             "report a rank count that was not used")
        return
    backend = _backend_from_env()
    shared = os.environ.get("XG_SHARE_GPU") == "1"  # tests: several gloo ranks computing on one GPU
    if backend == "nccl" or not shared:
        vis = _visible_gpus()
        if backend == "nccl" and vis < n_gpus:
            raise SystemExit(f"--gpus {n_gpus} but only {vis} GPU(s) are visible; one process per GPU, "
                             "no oversubscription")
    if n_gpus == 1 and not os.environ.get("XG_BENCH_FORCE_DIST"):
        return
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script] + list(argv)
    print("[launcher] " + " ".join(cmd), file=sys.stderr, flush=True)
    sys.exit(subprocess.call(cmd, env=env))


# ----------------------------------------------------------------------------------------------
# Placement: a rank drives its GPU from Python (4 launches of ~1.6 ms per bench step), so on a two-socket 8-GPU
# host its process belongs on the CPUs of the NUMA node its GPU hangs off -- a rank scheduled across the socket
# link pays that latency on every launch and on every pinned-memory copy.  The binding is read from sysfs
# (PCI device -> local_cpulist / numa_node), opt-out with XG_NUMA_BIND=0, and REPORTED whether or not it was
# applied, so that the first 8-GPU run explains itself.
# ----------------------------------------------------------------------------------------------
def parse_cpulist(text: str):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)"""
    cpus = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def format_cpulist(cpus) -> str:
    """[0, 1, 2, 3, 8] -> '0-3,8'"""
    cpus = sorted(set(int(c) for c in cpus))
    out, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


def pci_placement(pci_bus_id: Optional[str], sysfs: str = "/sys") -> dict:
    """{pci_bus_id, numa_node, cpus} of a PCI device from sysfs: `local_cpulist` of the device, else the cpulist of its
    `numa_node`; numa_node -1 / unreadable files leave `cpus` empty (a single-node host: nothing to bind)."""
    import os

    info = {"pci_bus_id": pci_bus_id, "numa_node": None, "cpus": []}
    if not pci_bus_id:
        return info
    dev = os.path.join(sysfs, "bus", "pci", "devices", pci_bus_id.lower())

    def read(path):
        try:
            with open(path) as f:
                return f.read().strip()
        except OSError:
            return None

    node = read(os.path.join(dev, "numa_node"))
    if node is not None:
        try:
            info["numa_node"] = int(node)
        except ValueError:
            pass
    cpulist = read(os.path.join(dev, "local_cpulist"))
    if not cpulist and info["numa_node"] is not None and info["numa_node"] >= 0:
        cpulist = read(os.path.join(sysfs, "devices", "system", "node", f"node{info['numa_node']}", "cpulist"))
    if cpulist:
        try:
            info["cpus"] = parse_cpulist(cpulist)
        except ValueError:
            info["cpus"] = []
    return info


def gpu_pci_bus_id(index: int) -> Optional[str]:
    """'0000:c1:00.0' of cuda:index (None without a GPU)"""
    import torch

    if not torch.cuda.is_available():
        return None
    try:
        p = torch.cuda.get_device_properties(index)
        return f"{int(p.pci_domain_id):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}.0"
    except Exception:
        return None


def set_affinity_all_threads(cpus) -> int:
    """`os.sched_setaffinity` for every thread of this process (/proc/self/task); returns how many were moved.  A thread
    that exits meanwhile, or a platform without /proc, leaves the calling thread bound at least."""
    import os

    cpus = sorted(cpus)
    os.sched_setaffinity(0, cpus)
    moved = 1
    import threading

    try:
        me = threading.get_native_id()  # the CALLING thread's tid (os.getpid() is the main thread's: ADVICE r05)
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        return moved
    for tid in tids:
        if tid == me:
            continue
        try:
            os.sched_setaffinity(tid, cpus)
            moved += 1
        except OSError:
            pass
    return moved


def bind_to_gpu_numa(local_rank: int, sysfs: str = "/sys", pci_bus_id: Optional[str] = None, apply: Optional[bool] = None) -> dict:
    """Restrict this process -- every thread it has -- to the CPUs local to its GPU and return what was found and done:
    {local_rank, pci_bus_id, numa_node, cpus (cpulist text), n_cpus, bound, affinity_before / affinity (cpulists the
    process ran / runs on)}.
    `apply` None = on unless XG_NUMA_BIND=0.  Never raises: an unreadable sysfs or a refused affinity call is reported."""
    import os

    if apply is None:
        apply = os.environ.get("XG_NUMA_BIND", "1") != "0"
    info = pci_placement(pci_bus_id if pci_bus_id is not None else gpu_pci_bus_id(local_rank), sysfs)
    cpus = info.pop("cpus")
    out = {"local_rank": int(local_rank), "pci_bus_id": info["pci_bus_id"], "numa_node": info["numa_node"],
           "cpus": format_cpulist(cpus), "n_cpus": len(cpus), "bound": False}
    try:
        out["affinity_before"] = format_cpulist(os.sched_getaffinity(0))  # to undo the binding (bench.py's CPU-baseline leg)
    except (AttributeError, OSError):
        out["affinity_before"] = None
    if apply and cpus and hasattr(os, "sched_setaffinity"):
        try:
            allowed = set(os.sched_getaffinity(0))
            target = sorted(set(cpus) & allowed)  # never widen a cpuset the launcher / container imposed
            if target:
                # `sched_setaffinity(0, ...)` moves the CALLING thread only: threads that already exist (torch's / OpenMP's
                # pools) would keep the old mask while new ones inherit the narrow one.  Every task of the process moves.
                n = set_affinity_all_threads(target)
                out["bound"] = True
                out["threads_bound"] = n
        except OSError as exc:
            out["error"] = str(exc)
    try:
        out["affinity"] = format_cpulist(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        out["affinity"] = None
    return out


class Ranks:
    """The process group as this rank sees it: RCCL ("nccl") on the GPU box, gloo in CPU tests.  Only
    barriers and scalar reductions go through it -- the record-axis split has no data-path collective."""

    def __init__(self, rank: int, world: int, local_rank: int, backend: Optional[str], dist, placement: Optional[dict] = None):
        self.rank, self.world, self.local_rank, self.backend, self.dist = rank, world, local_rank, backend, dist
        self.placement = placement or {"local_rank": local_rank, "bound": False}  # bind_to_gpu_numa's report

    def gather_objects(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank (small picklable records: placement, timings)"""
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    @property
    def scalar_device(self) -> str:
        return "cuda" if self.backend == "nccl" else "cpu"

    def barrier(self) -> None:
        import torch

        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if self.dist is not None:
            if self.backend == "nccl":
                self.dist.barrier(device_ids=[self.local_rank])
            else:
                self.dist.barrier()
            if torch.cuda.is_available():
                torch.cuda.synchronize()

    def gather_floats(self, value: float):
        """[value of rank 0, value of rank 1, ...] on every rank."""
        if self.dist is None:
            return [float(value)]
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64, device=self.scalar_device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def sum_int(self, value: int) -> int:
        """order-independent checksum of checksums (int64 wrap-around sum)"""
        if self.dist is None:
            return int(value)
        import torch

        t = torch.tensor([int(value)], dtype=torch.int64, device=self.scalar_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def sum_u64(self, value: int) -> int:
        """sum modulo 2^64 over the ranks (checksum of checksums); sent as two 32-bit halves so that no
        backend's integer overflow behaviour matters"""
        value = int(value) & 0xFFFFFFFFFFFFFFFF
        if self.dist is None:
            return value
        import torch

        t = torch.tensor([value & 0xFFFFFFFF, value >> 32], dtype=torch.int64, device=self.scalar_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        lo, hi = int(t[0].item()), int(t[1].item())
        return (lo + (hi << 32)) & 0xFFFFFFFFFFFFFFFF

    def max(self, value: float) -> float:
        return reduce_max(value, self.dist, self.scalar_device)

    def min(self, value: float) -> float:
        return -reduce_max(-float(value), self.dist, self.scalar_device)

    def sum(self, value: float) -> float:
        return reduce_sum(value, self.dist, self.scalar_device)

    def close(self) -> None:
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()


def init_ranks(n_gpus: int, backend: Optional[str] = None) -> Ranks:
    """Join the process group the launcher prepared (call after `ensure_ranks`).  One process per GPU: rank r
    of a node computes on cuda:LOCAL_RANK.  The world size RCCL / gloo reports must equal `n_gpus`."""
    import os

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != int(n_gpus):
        raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world}")
    backend = backend or _backend_from_env()
    shared = os.environ.get("XG_SHARE_GPU") == "1" and backend != "nccl"
    if torch.cuda.is_available():
        vis = torch.cuda.device_count()
        if shared:
            torch.cuda.set_device(local_rank % vis)
        else:
            if local_rank >= vis:
                raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {vis} GPU(s) visible")
            torch.cuda.set_device(local_rank)
    elif backend == "nccl":
        raise SystemExit("backend nccl (RCCL) needs a GPU")
    # this rank's process onto the CPUs of its GPU's NUMA node (reported in any case; XG_NUMA_BIND=0 leaves it alone)
    placement = bind_to_gpu_numa(local_rank if not shared or not torch.cuda.is_available() else local_rank % torch.cuda.device_count())
    placement["local_rank"] = local_rank
    if "WORLD_SIZE" not in os.environ:
        return Ranks(0, 1, 0, None, None, placement)
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    if dist.get_world_size() != int(n_gpus):
        raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {n_gpus} asked")
    return Ranks(rank, world, local_rank, backend, dist, placement)


# ----------------------------------------------------------------------------------------------
# Resident batches: a rank's records rarely fit HBM together with their outputs (config 4: 45 records of
# 5.18 GB in, the same out, per GPU), so it walks its block [lo, hi) of the record axis in batches that do.
# ----------------------------------------------------------------------------------------------
def records_per_batch(n_local: int, bytes_per_record: int, free_bytes: Optional[int] = None,
                      headroom: float = 0.85, cap: Optional[int] = None) -> int:
    """How many records (inputs + outputs + temporaries = `bytes_per_record` each) stay resident at once:
    `headroom` x free HBM / bytes_per_record, at least 1, at most `cap` and the rank's own count."""
    if n_local <= 0:
        return 0
    if free_bytes is None:
        import torch

        free_bytes = torch.cuda.mem_get_info()[0] if torch.cuda.is_available() else 8 << 30
    n = int(headroom * float(free_bytes) // max(1, int(bytes_per_record)))
    n = max(1, min(n, n_local))
    return min(n, int(cap)) if cap else n


def record_batches(n_records: int, world: int, rank: int, per_batch: int):
    """[(start, stop), ...] covering this rank's block of the record axis in resident batches."""
    lo, hi = shard_bounds(n_records, world, rank)
    if per_batch < 1:
        return []
    return [(s, min(s + per_batch, hi)) for s in range(lo, hi, per_batch)]


def map_record_batches(fn, load, n_records: int, ranks: "Ranks", bytes_per_record: int, per_batch: Optional[int] = None,
                       sink=None, headroom: float = 0.8):
    """The record-axis shard of a job, as a loop: this rank's block of `n_records` (`shard_bounds`) is walked in
    HBM-resident batches; per batch `load(start, stop)` brings the records in (an HBM tensor, or a host array that
    the operators upload), `fn(block)` runs the `Grid` operators, `sink(start, stop, result)` takes the result
    (default: results are collected and returned as a list of `(start, stop, result)`).  No data-path collective:
    the reference's `dask="parallelized"` independence over broadcast dims (xgcm/grid.py:786-789).  The batch size
    is agreed across the ranks (minimum of `records_per_batch` over the ranks) so that rounds line up for callers
    that put barriers around them."""
    lo, hi = shard_bounds(n_records, ranks.world, ranks.rank)
    per = per_batch or records_per_batch(hi - lo, bytes_per_record, headroom=headroom)
    # a rank without records (more ranks than records) has no say in the batch size: it contributes the job's upper
    # bound instead of 0, which would have forced one-record batches on everybody
    per = max(1, int(ranks.min(per if hi > lo else max(1, int(n_records)))))
    out = []
    for start, stop in record_batches(n_records, ranks.world, ranks.rank, per):
        result = fn(load(start, stop))
        if sink is not None:
            sink(start, stop, result)
        else:
            out.append((start, stop, result))
    return None if sink is not None else out


# ----------------------------------------------------------------------------------------------
# Sharding ALONG the operator's own axis (not needed by the BASELINE configs, which split an outer
# axis): the one data-path exchange the hot path can have.  It is the analogue of the reference's
# `map_overlap(depth=padding_width)` over chunks of the core dim (xgcm/grid_ufunc.py:1045-1125):
# each rank needs ONE plane from its neighbour, sent point to point (over xGMI: a plane of a
# 3600 x 2400 level is 69 MB against 5.2 GB of local work), after which the ordinary halo-mode
# kernel (`xg_stencil1d_halo_*`, the mechanism face connections use) reads the shard once.
# ----------------------------------------------------------------------------------------------
def exchange_halo(local, axis: int, pad: Tuple[int, int], bc: Optional[str], fill: float = 0.0, dist=None):
    """One-plane halos of a field sharded in contiguous blocks (rank order) along `axis`.

    Returns an array shaped like `local` with `axis` shortened to pad_lo + pad_hi (low halo first):
    interior shard boundaries take the neighbour rank's edge plane (isend / irecv), the two ends of
    the global axis follow the boundary mode -- `periodic` closes the ring between the last and the
    first rank, `fill` / `extend` are made locally, exactly as numpy.pad would on the whole array."""
    import numpy as np
    import torch

    lo, hi = int(pad[0]), int(pad[1])
    if lo not in (0, 1) or hi not in (0, 1):
        raise NotImplementedError("sharded core axis: halo widths of at most one plane")
    is_np = not isinstance(local, torch.Tensor)
    t = torch.as_tensor(local)
    axis = axis % t.dim()
    if t.shape[axis] == 0:
        raise ValueError("a rank owns no cells along the sharded axis")
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    # edge planes and the halo slab are laid out by the library's strided copy when they live in HBM (host tensors -- the
    # gloo tests -- stay with torch)
    def plane(at):
        p = t.narrow(axis, at, 1)
        if t.is_cuda:
            from . import device as _dev
            return _dev.materialize(p)
        return p.contiguous()

    first = plane(0)
    last = plane(t.shape[axis] - 1)
    if (lo or hi) and bc is None:
        raise ValueError("no boundary condition for the sharded axis")

    def boundary(edge_plane, wrapped_plane):
        if bc == "periodic":
            return wrapped_plane
        if bc == "fill":
            return torch.full_like(edge_plane, float(fill))
        if bc == "extend":
            return edge_plane
        raise ValueError(f"unknown boundary mode {bc!r}")

    recv_lo = torch.empty_like(first) if lo else None
    recv_hi = torch.empty_like(first) if hi else None
    if world > 1:
        ring = bc == "periodic"
        prev_r, next_r = (rank - 1) % world, (rank + 1) % world
        # RCCL moves HBM buffers peer to peer over xGMI; a transport that only moves host memory (gloo: ranks sharing one
        # GPU in the tests, or a CPU run) gets the planes staged through the host -- one plane each way, the same values
        staged = t.is_cuda and dist.get_backend() != "nccl"
        wire = (lambda p: p.cpu()) if staged else (lambda p: p)
        w_lo = wire(recv_lo) if lo else None
        w_hi = wire(recv_hi) if hi else None
        ops = []
        # order matters when prev == next (two ranks): sends (last, first) pair with recvs (lo, hi)
        if lo and (rank < world - 1 or ring):
            ops.append(dist.P2POp(dist.isend, wire(last), next_r))
        if lo and (rank > 0 or ring):
            ops.append(dist.P2POp(dist.irecv, w_lo, prev_r))
        if hi and (rank > 0 or ring):
            ops.append(dist.P2POp(dist.isend, wire(first), prev_r))
        if hi and (rank < world - 1 or ring):
            ops.append(dist.P2POp(dist.irecv, w_hi, next_r))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if staged:
            if lo:
                recv_lo.copy_(w_lo)
            if hi:
                recv_hi.copy_(w_hi)
        if lo and rank == 0 and not ring:
            recv_lo = boundary(first, None)
        if hi and rank == world - 1 and not ring:
            recv_hi = boundary(last, None)
    else:
        if lo:
            recv_lo = boundary(first, last)
        if hi:
            recv_hi = boundary(last, first)
    parts = [p for p in (recv_lo, recv_hi) if p is not None]
    if not parts:
        shape = list(t.shape)
        shape[axis] = 0
        halo = t.new_empty(shape)
    else:
        if len(parts) > 1 and all(p.is_cuda for p in parts):
            from . import device as _dev
            halo = _dev.concatenate(parts, axis)
        else:
            halo = torch.cat(parts, dim=axis) if len(parts) > 1 else parts[0]
    return halo.numpy() if is_np else halo


def stencil_along_sharded_axis(grid, funcname: str, da, axis: str, dist=None, to=None, padding=None, fill_value=None):
    """`Grid.diff / interp / min / max(da, axis)` for a `da` that holds THIS rank's contiguous block of
    the axis' own dimension (blocks in rank order).  Length-preserving position pairs (center <-> left /
    right: one halo plane); the result is this rank's block of the output, same bits as the
    corresponding rows of the single-process result."""
    from . import device as _dev
    from . import gridops
    from .grid import _select_grid_ufunc
    from .labeled import DataArray

    if funcname not in ("diff", "interp", "min", "max"):
        raise NotImplementedError(f"{funcname} along a sharded axis: see cumsum_along_sharded_axis for the scan")
    if gridops.complex_topology(grid, axis):
        raise NotImplementedError("sharding along an axis with face connections / a fold")
    sig = grid._create_1d_grid_ufunc_signatures(da, axis=[axis], to=grid._map_kwargs_over_axes(to))[0]
    ufunc, _ = _select_grid_ufunc(funcname, sig, module=gridops)
    lo, hi = next(iter(ufunc.padding_width.values())) if ufunc.padding_width else (0, 0)
    if lo + hi != 1:
        raise NotImplementedError("sharded core axis: only length-preserving position pairs (one halo plane)")
    bc = grid._complete_user_kwargs_using_axis_defaults(padding, "padding")[axis]
    fv = grid._complete_user_kwargs_using_axis_defaults(fill_value, "fill_value")[axis]
    in_dim = grid.axes[axis].coords[ufunc.from_pos]
    out_dim = grid.axes[axis].coords[ufunc.to_pos]
    num = da.get_axis_num(in_dim)
    halo = exchange_halo(da.data, num, (lo, hi), bc, 0.0 if fv is None else float(fv), dist)
    out = _dev.stencil1d_halo(funcname, da.data, halo, num, lo, hi)
    if not isinstance(da.data, type(out)) and hasattr(_dev, "tohost"):
        out = _dev.tohost(out)  # host array in -> host array out, like the Grid methods
    return DataArray(out, tuple(out_dim if d == in_dim else d for d in da.dims), name=da.name)


def cumsum_along_sharded_axis(grid, da, axis: str, dist=None, to=None, padding=None, fill_value=None):
    """`Grid.cumsum(da, axis)` for a `da` that holds THIS rank's contiguous block of the axis' own dimension.

    The carry is the reference's blockwise answer to a scan over chunks ("it would need blockwise",
    xgcm/grid.py:811-813): each rank sums its block (`xg_reduce1d_*`), the block totals -- one plane per
    rank -- are all-gathered, every rank adds the totals of the ranks before it in rank order and shifts its
    local scan by that plane.  Length-preserving position pairs, forward direction, `fill` / `extend`
    boundary.  The block offset re-associates the sum: results agree with the single-process scan to
    rounding (1e-12 relative), not bit for bit; NaN cells count as zero like `Grid.cumsum`."""
    import torch

    from . import device as _dev
    from . import gridops
    from .grid import _cumsum_trim_pad
    from .labeled import DataArray

    if gridops.complex_topology(grid, axis):
        raise NotImplementedError("sharding along an axis with face connections / a fold")
    ax = grid.axes[axis]
    pos, dim = ax._get_position_name(da)
    ax_to = to if to is not None else ax._default_shifts[pos]
    trims = _cumsum_trim_pad(pos, ax_to, False, ax)
    if trims not in ((0, 0, 0, 0), (0, 1, 1, 0)):
        raise NotImplementedError("sharded core axis: only length-preserving position pairs")
    bc = grid._complete_user_kwargs_using_axis_defaults(padding, "padding")[axis]
    fv = grid._complete_user_kwargs_using_axis_defaults(fill_value, "fill_value")[axis]
    if trims[2] and bc not in ("fill", "extend"):
        raise NotImplementedError("sharded core axis: the scan's halo needs `fill` or `extend` (a periodic halo is the far end's total)")
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    num = da.get_axis_num(dim)
    host = not isinstance(da.data, torch.Tensor)
    out_dims = tuple(ax.coords[ax_to] if d == dim else d for d in da.dims)
    # local scan: rank 0 applies the real boundary mode, later ranks start from 0 and are shifted by the carry
    first = rank == 0
    local = _dev.cumsum1d(da.data, num, *trims, (bc if first else "fill") if trims[2] else None,
                          (0.0 if fv is None else float(fv)) if first else 0.0, False, True)
    if world > 1:
        total = _dev.reduce1d(da.data, num, None, True)          # this block's sum: one plane
        t = total if isinstance(total, torch.Tensor) else torch.as_tensor(total)  # (a large host block comes back as numpy)
        if dist.get_backend() == "nccl" and not t.is_cuda:
            t = t.cuda()  # RCCL moves device buffers only
        elif dist.get_backend() != "nccl" and t.is_cuda:
            t = t.cpu()   # ... and gloo host buffers only (ranks sharing one GPU in the tests): totals staged through the host
        planes = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(planes, t.contiguous())
        if not host and not planes[0].is_cuda:
            planes = [p.cuda() for p in planes]
        if rank > 0:
            plane = (lambda p: _dev.tohost(p)) if host else (lambda p: p)  # reduce1d returns HBM tensors on a GPU box
            carry = DataArray(plane(planes[0]), tuple(d for d in da.dims if d != dim))
            for r in range(1, rank):                              # rank order: ((t0 + t1) + t2) ...
                carry = carry + DataArray(plane(planes[r]), carry.dims)
            res = DataArray(local if not host else _dev.tohost(local), out_dims) + carry
            return DataArray(res.transpose(*out_dims).data, out_dims, name=da.name)
    return DataArray(_dev.tohost(local) if host else local, out_dims, name=da.name)
