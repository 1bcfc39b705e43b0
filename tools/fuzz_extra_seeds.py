"""On-demand robustness run (not part of the suites): the GPU fuzz tests of tests/test_gpu_fuzz.py with 150 further seeds,
the scans / reductions also with the chained kernels forced (1500 runs of 20-25 random cases, 30 s on one MI355X)."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tests.test_gpu_fuzz as F
from xgcm_amd import device as dev, _hip
t0 = time.time(); n = 0
orig = np.random.default_rng
for seed in range(1000, 1150):
    for dtype in F.DTYPES:
        # the test functions derive their rng from a fixed base + seed: pass large seeds for new cases
        F.test_fuzz_stencil(dev, seed, dtype); n += 1
        for chain in (1, 2):
            _hip.set_tunable("scan_chain", chain)
            F.test_fuzz_cumsum_reduce.__wrapped__(dev, seed, dtype, chain) if hasattr(F.test_fuzz_cumsum_reduce, "__wrapped__") else F.test_fuzz_cumsum_reduce(dev, seed, dtype, chain)
            n += 1
        _hip.set_tunable("scan_chain", 1)
        F.test_fuzz_row_scan_any_length(dev, seed, dtype); n += 1
        F.test_fuzz_two_axis_and_vorticity(dev, seed, dtype); n += 1
for seed in range(1000, 1020):  # the integer builds (*_i32 / *_i64): every dtype, 20 further seeds
    for dtype in F.INT_DTYPES:
        F.test_fuzz_integer_lanes(dev, seed, dtype); n += 1
print(f"{n} extra fuzz runs passed in {time.time()-t0:.0f} s")
