#!/usr/bin/env python3
"""Tall block operators of DENSE children (the reference's JopBaz, test/runtests.jl:27-33): N x 1 blocks of k x k Float32
matrices, 1 GiB of matrices per operator; forward / adjoint time and GB/s of matrix bytes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J

J.init(0)
total = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
WIDE = len(sys.argv) > 2 and sys.argv[2] == "wide"
GWS = [int(a[3:]) for a in sys.argv[2:] if a.startswith("gw=")]          # sweep of the fused adjoint's children-per-wave knob
FWS = [int(a[4:]) for a in sys.argv[2:] if a.startswith("fws=")]        # sweep of the forward's column-split target (workgroups)
OLD = "old" in sys.argv[2:]                                              # the three-launch adjoint of rounds 1-3 beside the fused kernel


def timed(fn, reps=5):
    fn(); fn(); fn()
    best = 1e30
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


print(f"# {'wide 1 x N' if WIDE else 'tall N x 1'} operators of dense k x k Float32 children, {total >> 20} MiB of matrices; best of 5")
for k in (8192, 4096, 2048, 1024, 512, 256, 128):
    nrow = total // (4 * k * k)
    if nrow > 16384:
        continue
    mats = [J.rand(J.JetSpace(np.float32, k, k), seed=1, stream=i) for i in range(nrow)]
    A = J.blockop([[J.JopDense(M) for M in mats]]) if WIDE else J.blockop([[J.JopDense(M)] for M in mats])
    m = J.rand(J.domain(A), seed=2, stream=0)
    d = J.rand(J.range(A), seed=3, stream=0)
    mt = J.zeros(J.domain(A))
    tf = timed(lambda: J.mul_(d, A, m))
    ta = timed(lambda: J.mul_(mt, A.H, d))
    nbytes = nrow * k * k * 4
    fused = J.tune_get('last_dense_fused')
    extra = ""
    for gw in GWS:
        J.tune(dense_gw=gw)
        extra += f" gw={gw}: {timed(lambda: J.mul_(mt, A.H, d)):.3f}"
    J.tune(dense_gw=0)
    for w in FWS:
        J.tune(dense_fwd_wgs=w)
        extra += f" fws={w}: {timed(lambda: J.mul_(d, A, m)):.3f}"
    J.tune(dense_fwd_wgs=0)
    if OLD:
        J.tune(dense_fused=0)
        t_old = timed(lambda: J.mul_(mt, A.H, d))
        J.tune(dense_fused=1)
        extra += f" | three-launch adj {t_old:8.3f} ms {nbytes / t_old / 1e6:6.0f} GB/s"
    print(f"{nrow:6d} children of {k:5d}^2: fwd {tf:8.3f} ms {nbytes / tf / 1e6:6.0f} GB/s | adj {ta:8.3f} ms {nbytes / ta / 1e6:6.0f} GB/s   "
          f"(fused {fused}){extra}", flush=True)
    del A, mats, m, d, mt
