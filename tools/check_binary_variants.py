#!/usr/bin/env python3
"""`a OP b` with a level-shared broadcast operand (xg_set_tunable bin_zl = 2 / 4) against one level per thread, bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from xgcm_amd import _hip
from xgcm_amd import device as D

lib = _hip.load()
bad = 0
for shape in [(75, 240, 3600), (7, 600, 1024), (5, 300, 1026), (3, 1, 70000), (2, 65, 2048)]:
    T = D.synthetic(shape, 2)
    m = D.synthetic((1,) + shape[1:], 31, 0, 1000.0, 1000.0)
    for op in ("mul", "div", "add", "sub"):
        for a, b in ((T, m), (m, T)):
            assert lib.xg_set_tunable(b"bin_zl", 0) == 0
            ref = D.binary(op, a, b)
            for v in (2, 4):
                assert lib.xg_set_tunable(b"bin_zl", v) == 0
                got = D.binary(op, a, b)
                if not torch.equal(got, ref):
                    bad += 1
                    print("MISMATCH", shape, op, "broadcast first" if a is m else "broadcast second", v)
lib.xg_set_tunable(b"bin_zl", 0)
print("checked; mismatches:", bad)
sys.exit(1 if bad else 0)
