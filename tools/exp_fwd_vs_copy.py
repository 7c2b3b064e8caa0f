#!/usr/bin/env python3
"""Is the tall forward's distance to a plain device copy a property of the KERNEL or of the two slabs it streams between?  The forward reads the
coefficient slab (and the cached model tile) and writes the range vector; here the library's own copy kernel moves the SAME coefficient slab
into the SAME range vector, and the Hadamard product `d .= coeff .* d0` (two reads, one write) gives the 2:1 rate between them.

    python tools/exp_fwd_vs_copy.py [NROW] [EDGE] > profiles/exp_r04_fwd_vs_copy.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J

nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 128
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 256
J.init(0)
n = edge ** 3
spc = J.JetSpace("float32", edge, edge, edge)
R = J.JetBSpace([spc] * nrow)
coeff = J.rand(R, seed=1, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
m = J.rand(spc, seed=2, stream=0)
d = J.zeros(R)
other = J.rand(R, seed=3, stream=0)


def timed(fn, reps=8, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        e0 = J.Event().record()
        fn()
        e1 = J.Event().record()
        ts.append(e0.elapsed_ms(e1))
    return min(ts)


nb = nrow * n * 4
print(f"# {nrow} x {edge}^3 Float32: {nb / 2**30:.0f} GiB per slab; GB/s over the bytes each line moves", flush=True)
for walk in (7, 8, 9, 1, 2):
    J.op_tune_set(A, "fwd_walk", walk)
    t = timed(lambda: J.mul_(d, A, m))
    print(f"forward, candidate {walk}:            d <- coeff .* m   {t:7.3f} ms  {(2 * nb + n * 4) / t / 1e6:7.1f} GB/s")
t = timed(lambda: J.copyto_(d, coeff))
print(f"copy, same slabs:                 d <- coeff        {t:7.3f} ms  {2 * nb / t / 1e6:7.1f} GB/s")
t = timed(lambda: J.copyto_(coeff, d))
print(f"copy, the other way:              coeff <- d        {t:7.3f} ms  {2 * nb / t / 1e6:7.1f} GB/s")
t = timed(lambda: J.copyto_(d, other))
print(f"copy from a third slab:           d <- other        {t:7.3f} ms  {2 * nb / t / 1e6:7.1f} GB/s")
t = timed(lambda: J.hadamard_(d, coeff, other))
print(f"product of two slabs:             d <- coeff .* other {t:7.3f} ms  {3 * nb / t / 1e6:7.1f} GB/s")
coeff2 = J.rand(R, seed=1, stream=0)                                      # restore (the copy the other way overwrote it)
