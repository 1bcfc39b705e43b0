"""Diagnostic: per-op host enqueue time / allocator state for the first steps (slow-start hunt)."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

import bench
from xgcm_amd import device as D

field = D.synthetic((75, 2400, 3600), 2)
grid, T = bench.build_grid(75, field)
torch.cuda.synchronize()
rows = []
for k in range(8):
    for fn, ax in bench.OPS:
        t0 = time.perf_counter()
        getattr(grid, fn)(T, ax)
        t1 = time.perf_counter()
        rows.append((k, fn + ax, round((t1 - t0) * 1e3, 2), torch.cuda.memory_reserved() >> 20,
                     torch.cuda.memory_stats().get("num_device_alloc", -1), torch.cuda.memory_stats().get("num_device_free", -1)))
    torch.cuda.synchronize()
for r in rows:
    print(r)
