#!/usr/bin/env python3
"""PCIe-inclusive rate of the chunked-input path (xgcm_amd.chunked): a 75 x 2400 x 3600 f64 field held as a
BlockArray of 15 blocks of 5 levels through Grid.diff / Grid.cumsum / Grid.integrate, next to the same field as ONE host array
(the pipelined numpy-in / numpy-out path) and resident in HBM.  Never the bench's `value`: host memory in, host memory out."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from xgcm_amd import DataArray  # noqa: E402
from xgcm_amd import device as D  # noqa: E402
from xgcm_amd.chunked import BlockArray  # noqa: E402
from tools.bench_configs import mitgcm_grid  # noqa: E402

nz, ny, nx = 75, 2400, 3600
grid = mitgcm_grid(nz, ny, nx)
host = D.tohost(D.synthetic((nz, ny, nx), 2))
dims = ("Z", "YC", "XC")
blocks = BlockArray.from_array(host, ((5,) * 15, (ny,), (nx,)))
resident = DataArray(D.asdevice(host), dims)


def t(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


for name, call in (("diff X", lambda da: grid.diff(da, "X")), ("derivative Y", lambda da: grid.derivative(da, "Y")),
                   ("cumsum X", lambda da: grid.cumsum(da, "X")), ("integrate Y", lambda da: grid.integrate(da, "Y"))):
    s_chunk, out_c = t(lambda: call(DataArray(blocks, dims)))
    s_host, out_h = t(lambda: call(DataArray(host, dims)))
    s_res, out_r = t(lambda: call(resident), 5)
    same = bool(np.array_equal(np.asarray(out_c.values), np.asarray(out_h.values), equal_nan=True))
    gb = host.nbytes / 1e9
    print(json.dumps({"op": name, "chunked_15_blocks_s": round(s_chunk, 4), "one_host_array_s": round(s_host, 4), "resident_ms": round(s_res * 1e3, 3),
                      "chunked_GBps_in": round(gb / s_chunk, 1), "host_GBps_in": round(gb / s_host, 1), "same_values": same,
                      "result": type(out_c.data).__name__}), flush=True)
