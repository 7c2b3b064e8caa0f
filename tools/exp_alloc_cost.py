#!/usr/bin/env python3
"""What does a range-sized temporary cost?  jh_bvec_create / destroy (hipMalloc / hipFree) per size, first and repeated, and zeros() on top.

    python tools/exp_alloc_cost.py
"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

J.init(0)
print("# GiB   create (no fill) ms   destroy ms   zeros() ms   [three rounds]", flush=True)
for gib in (0.0625, 1, 8, 64):
    n = int(gib * 2 ** 30 / 4)
    spc = J.JetSpace(np.float32, n)
    rows = []
    for rep in range(3):
        J.synchronize()
        t0 = time.perf_counter()
        x = J.Array(spc) if hasattr(J, "Array") else J.zeros(spc)
        J.synchronize()
        t1 = time.perf_counter()
        x.close() if hasattr(x, "close") else None
        del x
        gc.collect()
        J.synchronize()
        t2 = time.perf_counter()
        z = J.zeros(spc)
        J.synchronize()
        t3 = time.perf_counter()
        z.close() if hasattr(z, "close") else None
        del z
        gc.collect()
        rows.append((1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
    print(f"{gib:7.4f}  " + "   ".join(f"{a:9.3f} {b:9.3f} {c:9.3f}" for a, b, c in rows), flush=True)
