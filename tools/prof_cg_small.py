#!/usr/bin/env python3
"""Small-operator solver loops under rocprofv3: 64 x 64^3 Float32, 200 forced iterations each of CG on the normal equations, CGLS and LSQR
(recurrences on the device, graph-replayed): kernel durations and the gaps between them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J
J.init(0)
nrow, edge = 64, 64
blk = J.JetSpace(np.float32, edge, edge, edge)
coeff = J.rand(J.JetBSpace([blk] * nrow), seed=1, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
x_true = J.rand(J.domain(A), seed=4, stream=0)
b = J.mul(A, x_true)
which = sys.argv[1] if len(sys.argv) > 1 else "cgnr"
for _ in range(2):
    J.mul_(b, A, x_true)
    if which == "cgnr": r = J.cgnr(A, b, maxiter=200, atol=0.0, btol=0.0, force_maxiter=True)
    elif which == "cgls": r = J.cgls(A, b, maxiter=200, atol=0.0, btol=0.0, force_maxiter=True, overwrite_b=True)
    else: r = J.lsqr(A, b, maxiter=200, atol=0.0, btol=0.0, conlim=0.0, force_maxiter=True, overwrite_b=True)
J.synchronize()
print(which, "iterations", r.itn, "istop", r.istop)
