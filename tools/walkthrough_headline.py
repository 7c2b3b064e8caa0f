#!/usr/bin/env python3
"""The reference's README walk-through at the HEADLINE size (1024 x 1 blocks of 256^3 Float32), written the way a user of the reference
writes it -- allocating calls (`d = A*m`), one coefficient array per block, no library-specific shortcuts -- with the wall time of every
line, twice (the second pass shows what the slab cache and the per-operator measurements leave).  Looks for lines that take seconds.

    python tools/walkthrough_headline.py [nrow] [edge]
"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 256
J.init(0)


class T:
    def __init__(self, what):
        self.what = what

    def __enter__(self):
        J.synchronize()
        self.t0 = time.perf_counter()

    def __exit__(self, *a):
        J.synchronize()
        print(f"    {1e3 * (time.perf_counter() - self.t0):10.2f} ms   {self.what}", flush=True)


blk = J.JetSpace(np.float32, edge, edge, edge)
for turn in (1, 2):
    print(f"## pass {turn}: {nrow} x 1 blocks of {edge}^3 Float32", flush=True)
    with T("coefficients: one rand(blk) per block (separate arrays, as a user's operators hold them)"):
        coeffs = [J.rand(blk, seed=1, stream=i) for i in range(nrow)]
    with T("A = @blockop [JopDiagonal(c) for c in coeffs]"):
        A = J.blockop([[J.JopDiagonal(c)] for c in coeffs])
    with T("m = rand(domain(A))"):
        m = J.rand(J.domain(A), seed=2, stream=0)
    with T("d = A*m            (allocates range(A): 64 GiB at the headline size)"):
        d = A * m
    print(f"                    (slab choice for d: {J.tune_get('last_alloc_choice')} = 100 x probed candidates + rank by fill time, -1: no choice; {J.tune_get('slab_probed')} probed slabs left in the cache)", flush=True)
    with T("d = A*m again      (allocates again)"):
        d2 = A * m
    with T("mul!(d, A, m)      (in place)"):
        J.mul_(d, A, m)
    with T("mt = A'*d"):
        mt = A.H * d
    with T("mul!(mt, A', d)"):
        J.mul_(mt, A.H, d)
    with T("dot_product_test(A, m, d)"):
        lhs, rhs = J.dot_product_test(A, m, d)
    with T("norm(d), dot(d, d2), extrema(d)"):
        nrm, dt_, ex = J.norm(d), J.dot(d, d2), J.extrema(d)
    with T("getblock(d, 7) (a view), getblock!(d, 7, out), setblock!(d, 7, 1.0)"):
        v = J.getblock(d, 7)
        out = J.zeros(blk)
        J.getblock_(d, 7, out)
        J.setblock_(d, 7, 1.0)
    with T("d .= 2 .* d .- d2   (broadcast over block arrays)"):
        d.assign(2.0 * d - 1.0 * d2)
    with T("y = (A'oA)*m"):
        y = (A.H @ A) * m
    with T("x = lsqr(A, d2; maxiter=5)   (keeps d2: works on a copy)"):
        res = J.lsqr(A, d2, maxiter=5, atol=0.0, btol=0.0, conlim=0.0)
    with T("x = lsqr(A, d2; maxiter=5) again"):
        res = J.lsqr(A, d2, maxiter=5, atol=0.0, btol=0.0, conlim=0.0)
    with T("close(A) and dropping every vector"):
        J.close(A)
        del A, coeffs, m, d, d2, mt, v, out, y, res
        gc.collect()
print(f"# the slab cache holds {J.tune_get('slab_cached_mib') / 1024:.1f} GiB at the end", flush=True)
