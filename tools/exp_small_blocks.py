#!/usr/bin/env python3
"""Tall diagonal operators with MANY rows of SMALL blocks (the opposite corner from the headline shape): forward, adjoint,
fused A'A and the one-pass LSQR step at a fixed operator size, against the bytes each has to move.

    python tools/exp_small_blocks.py [TOTAL_MIB]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J

total_mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
J.init(0)
reps = 5


def timed(fn):
    fn()
    fn()
    best = 1e30
    for _ in range(reps):
        e0 = J.Event().record()
        fn()
        e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


print(f"# tall diagonal Float32 operators of {total_mib} MiB, rows x block length; best of {reps}; GB/s of algorithmic bytes")
print("# adj_split=0: the ordered walk (one thread per 16-byte vector of the domain, all rows); auto: split-row walk when that starves the chip")
from jets_jl_amd._ffi import lib, check
import ctypes as C
from jets_jl_amd.jetblock import _tall_native


def step(A, u, v, w):
    out = C.c_double(0)
    check(lib.jh_blockop_bidiag_step(_tall_native(A).handle, u.handle, v.handle, w.handle, 1.0, -0.5, C.byref(out)))
    return out.value


for n in (1024, 4096, 16384, 65536, 262144, 1048576):
    nrow = total_mib * (1 << 20) // (4 * n)
    if nrow > (1 << 18):
        continue
    spc = J.JetSpace(np.float32, n)
    coeff = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    m = J.rand(J.domain(A), seed=2, stream=0)
    d = J.rand(J.range(A), seed=3, stream=0)
    mt = J.zeros(J.domain(A))
    w = J.zeros(J.domain(A))
    C_ = A.H @ A
    b2 = (2 * nrow * n + n) * 4
    b1 = (nrow * n + 2 * n) * 4
    b3 = (3 * nrow * n + 2 * n) * 4
    tf = timed(lambda: J.mul_(d, A, m))
    line = f"{nrow:7d} x {n:8d}: fwd {tf:7.3f} ms {b2 / tf / 1e6:6.0f} GB/s"
    ref = None
    res = {}
    for split in (0, -1):
        J.tune(adj_split=split)
        ta = timed(lambda: J.mul_(mt, A.H, d))
        parts = J.tune_get("last_adj_parts")
        got = mt.to_numpy().astype(np.float64)
        if ref is None:
            ref = got
        res[split] = (ta, parts, float(np.linalg.norm(got - ref) / np.linalg.norm(ref)))
    for split in (0, -1):
        J.tune(adj_split=split)
        ta, parts, err = res[split]
        tn = timed(lambda: J.mul_(mt, C_, m))
        ts = timed(lambda: step(A, d, m, w))
        line += (f" | split={split:2d} parts {parts:4d}: adj {ta:7.3f} ms {b2 / ta / 1e6:6.0f} GB/s (rel diff vs ordered {err:.1e}), "
                 f"A'A {tn:7.3f} ms {b1 / tn / 1e6:6.0f}, step {ts:7.3f} ms {b3 / ts / 1e6:6.0f}")
    J.tune(adj_split=-1)
    print(line, flush=True)
    del A, C_, coeff, m, d, mt, w
