#!/usr/bin/env python3
"""Tall / wide / block-diagonal operators of DENSE children whose dimensions are ODD (matrix rows off the 16-byte grid) beside the even neighbour.
    python tools/bench_dense_odd.py [TOTAL_MiB]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J

J.init(0)
total = (int(sys.argv[1]) if len(sys.argv) > 1 else 512) << 20


def timed(fn, reps=5):
    fn(); fn(); fn()
    best = 1e30
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


for layout in ("tall", "wide", "blockdiag"):
    for k0 in (4096, 1024, 256, 128):
        for k in (k0 - 1, k0):
            nch = max(2, total // (4 * k * k))
            if nch > 8192:
                continue
            if layout == "blockdiag":
                nch = min(nch, 64)
            mats = [J.rand(J.JetSpace(np.float32, k, k), seed=1, stream=i) for i in range(nch)]
            spc = J.JetSpace(np.float32, k)
            if layout == "tall":
                A = J.blockop([[J.JopDense(M)] for M in mats])
            elif layout == "wide":
                A = J.blockop([[J.JopDense(M) for M in mats]])
            else:
                A = J.blockop([[J.JopDense(mats[i]) if i == j else J.JopZeroBlock(spc, spc) for j in range(nch)] for i in range(nch)])
            m = J.rand(J.domain(A), seed=2, stream=0)
            d = J.rand(J.range(A), seed=3, stream=0)
            mt = J.zeros(J.domain(A))
            tf = timed(lambda: J.mul_(d, A, m))
            ta = timed(lambda: J.mul_(mt, A.H, d))
            nbytes = nch * k * k * 4
            flag = "  <-- CLIFF" if nbytes > (32 << 20) and min(nbytes / tf, nbytes / ta) / 1e6 < 2500 else ""
            print(f"{layout:9s} {nch:5d} children of {k:5d}^2: fwd {tf:8.3f} ms {nbytes / tf / 1e6:6.0f} GB/s | adj {ta:8.3f} ms {nbytes / ta / 1e6:6.0f} GB/s{flag}", flush=True)
            J.close(A)
            del A, mats, m, d, mt
