"""Operator chains on small grids as ONE hipGraph launch.

A `Grid` operator on a small HBM-resident array is launch-bound: ~45 us of Python dispatch + ctypes + kernel launch
for a kernel that runs a few microseconds (measured with a host-side profile of the dispatch path, round 2).  Every kernel of the library takes its stream as an
argument and keeps no per-launch host state -- the chained scans clean their workspace up inside the kernel, tickets
included -- so a whole sequence of operators can be captured once and replayed:

    T = DataArray(dev.asdevice(t0), ("Z", "YC", "XC"))            # static input buffer
    step = capture(lambda: (grid.derivative(T, "X"), grid.cumsum(T, "Y"), grid.interp(T, ["X", "Y"])))
    for t in timesteps:
        T.data.copy_(next_field)                                   # new values, same storage
        dudx, cs, corner = step()                                  # one graph launch; results in static output buffers

Measured on one MI355X (4 x 320 x 256 f64, five operators): 31 us per replay against 234 us eager.
The reference has no counterpart (its eager numpy / dask path pays Python + xarray overhead per operator); this is the
MI355X answer to its small-grid regime (BASELINE config 1 is "plumbing": 4.4 M cells).

Rules of the capture: inputs and metrics are HBM-resident tensors (no host arrays: a pageable copy cannot be captured),
their SHAPES are frozen, new values are written into the same storage; outputs are overwritten by every replay -- copy
what must survive.  Replays of ONE captured chain must not overlap each other (static outputs, and the chained scans'
hand-off workspace belongs to the capture); different captured chains are independent.
"""
from __future__ import annotations

from typing import Any, Callable

import torch


def _hip_chain_check():
    from . import _hip

    _hip.chain_check()


def _tensors(obj, found):
    """every HBM tensor reachable from the return value (DataArrays, tuples, lists, dicts)"""
    data = getattr(obj, "data", None)
    if isinstance(obj, torch.Tensor):
        found.append(obj)
    elif isinstance(data, torch.Tensor):
        found.append(data)
    elif isinstance(obj, dict):
        for v in obj.values():
            _tensors(v, found)
    elif isinstance(obj, (tuple, list)):
        for v in obj:
            _tensors(v, found)
    return found


class CapturedChain:
    """The result of :func:`capture`: call it to replay; `outputs` is what `fn` returned during capture."""

    def __init__(self, graph: "torch.cuda.CUDAGraph", outputs: Any, stream_handle=None):
        self._graph = graph
        self.outputs = outputs
        self._stream_handle = stream_handle  # the capture's own stream: owns the chained kernels' workspace

    def __del__(self):
        handle, self._stream_handle = getattr(self, "_stream_handle", None), None
        if handle:
            try:
                from . import _hip

                _hip.load().xg_stream_destroy(handle)
            except Exception:  # interpreter shutdown
                pass

    def __call__(self):
        # (a chained scan whose hand-off failed in an earlier replay was redone inside that replay by its marching
        # twin, which is part of the graph; the event is reported at the next replay or wherever a result is read)
        _hip_chain_check()
        self._graph.replay()
        return self.outputs

    replay = __call__


def capture(fn: Callable[[], Any], warmup: int = 2) -> CapturedChain:
    """Run `fn` (a closure over HBM-resident DataArrays calling Grid operators) `warmup` times on a side stream --
    library workspaces, metric uploads and halo maps are created there, outside the capture -- then capture it."""
    if not torch.cuda.is_available():
        raise RuntimeError("capture() needs the GPU: operator chains are hipGraphs")
    # a stream of the capture's own, not one of torch's pooled 32: the chained scans / reductions keep their hand-off
    # workspace per stream, and a graph must not share it with another capture or with eager calls
    import ctypes

    from . import _hip

    handle = ctypes.c_void_p()
    _hip.check(_hip.load().xg_stream_create(ctypes.byref(handle)))
    side = torch.cuda.ExternalStream(handle.value)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(max(1, warmup)):
            probe = fn()
    side.synchronize()
    found = _tensors(probe, [])
    if not found or not all(t.is_cuda for t in found):
        raise ValueError("capture(): fn must return HBM-resident results (DataArrays / tensors); host arrays cannot be captured")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        outputs = fn()
    torch.cuda.current_stream().wait_stream(side)
    return CapturedChain(graph, outputs, handle)
