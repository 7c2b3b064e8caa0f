#!/usr/bin/env python3
"""Does it matter WHICH of two 64 GiB slabs holds the coefficients and which the range vector?  The forward reads one and writes the other, and
the (read region -> write region) rate matrix of this chip is not symmetric (profiles/exp_r03_alloc_place.txt, section 3).  Both role
assignments of the same two slabs, in one process: forward (after its per-operator measurement), adjoint, pair.

    python tools/exp_swap_roles.py        (run it several times in a row: consecutive processes land differently)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

J.init(0)
nrow, edge = 1024, 256
blk = J.JetSpace(np.float32, edge, edge, edge)
R = J.JetBSpace([blk] * nrow)
X = J.rand(R, seed=1, stream=0)           # allocated first
Y = J.rand(R, seed=3, stream=0)           # allocated second
m = J.rand(blk, seed=2, stream=0)
mt = J.zeros(blk)


def measure(coeff, d, tag):
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    k = 0
    J.mul_(d, A, m)
    while J.op_tune_get(A, "fwd_walk") == -1 and 0 < J.op_tune_get(A, "fwd_trials") and k < 32:
        J.mul_(d, A, m)
        J.synchronize()
        k += 1
    for _ in range(3):
        J.mul_(d, A, m)
        J.mul_(mt, A.H, d)
    J.synchronize()
    reps = 10
    e = [J.Event() for _ in range(3)]
    tf = ta = 0.0
    for _ in range(reps):
        e[0].record()
        J.mul_(d, A, m)
        e[1].record()
        J.mul_(mt, A.H, d)
        e[2].record()
        J.synchronize()
        tf += e[0].elapsed_ms(e[1])
        ta += e[1].elapsed_ms(e[2])
    tf, ta = tf / reps, ta / reps
    print(f"{tag}: forward {tf:7.3f} ms  adjoint {ta:7.3f} ms  pair {tf + ta:7.3f} ms = {1e3 / (tf + ta):6.3f} pairs/s   (walk {J.op_tune_get(A, 'fwd_walk')})", flush=True)
    J.close(A)
    return tf + ta


a = measure(X, Y, "coefficients in the slab allocated FIRST, range vector in the second ")
b = measure(Y, X, "coefficients in the slab allocated SECOND, range vector in the first ")
print(f"# swapping the roles changes the pair by {100 * (b - a) / a:+.2f} %", flush=True)
