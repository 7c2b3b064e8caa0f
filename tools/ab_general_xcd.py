#!/usr/bin/env python3
"""A/B of the general kernels' grid decode on a 2x3 nonlinear block operator of 512^3 Float32 blocks: XCD-aware (the block
rows of a tile share an XCD's L2) vs tile-fastest (each block row streams its inputs on its own).  Prints HIP-event times;
under `rocprofv3 --pmc FETCH_SIZE` the per-dispatch counters give the HBM read traffic of the two orders (3 launches each,
XCD-aware first).

    python tools/ab_general_xcd.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

J.init(0)
edge = 512
spc = J.JetSpace(np.float32, edge, edge, edge)
n = edge ** 3
F = J.blockop([[J.JopSquare(spc) for _ in range(3)] for _ in range(2)])
m, d = J.rand(J.domain(F), seed=2, stream=0), J.rand(J.range(F), seed=3, stream=0)
unique = (4 + 3) * n * 4                                      # read d (2 blocks), m (3), write d (2)
per_use = (4 + 6) * n * 4                                     # every block row reads its three inputs itself
for xcd in (1, 0):
    J.tune(general_xcd=2 if xcd else 0)
    ts = []
    for _ in range(3):
        e0 = J.Event().record()
        J.mul_(d, F, m)
        e1 = J.Event().record()
        ts.append(e0.elapsed_ms(e1))
    t = min(ts)
    print(f"general_xcd={xcd}: f! of the 2x3 operator {t:7.3f} ms   unique bytes {unique / 1e9:.2f} GB -> {unique / t / 1e6:7.1f} GB/s   "
          f"(per-use bytes {per_use / 1e9:.2f} GB -> {per_use / t / 1e6:7.1f} GB/s)", flush=True)
J.tune(general_xcd=1)
