#!/usr/bin/env python3
"""The reference's benchmark groups that touch this path (benchmark/benchmarks.jl:88-157: "Block, homogeneous" /
"Block, heterogeneous": mul!, mul!-adjoint, block, block!, broadcast, fill!, dot, norm, extrema, reshape, each beside a
flat-array "(base-case)"), at MI355X scale, HIP-event timed, reported as achieved HBM GB/s over the algorithmic bytes.

    python tools/bench_suite.py [NBLOCKS EDGE] > profiles/bench_suite_r01.txt
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 256
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 256
PEAK = 8000.0
J.init(0)
n = edge ** 3
s = 4
blk = J.JetSpace(np.float32, edge, edge, edge)
R = J.JetBSpace([blk] * nblocks)
L = nblocks * n


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    e0 = J.Event().record()
    for _ in range(reps):
        fn()
    e1 = J.Event().record()
    return e0.elapsed_ms(e1) / reps


def row(name, nbytes, ms):
    gbs = nbytes / ms / 1e6
    print(f"{name:46s} {ms:10.3f} ms  {nbytes / 1e9:9.2f} GB  {gbs:8.1f} GB/s  {100 * gbs / PEAK:5.1f} % of 8 TB/s")


print(f"# {nblocks} blocks of {edge}^3 Float32 ({nblocks * n * s / 2**30:.0f} GiB per block vector), one MI355X")
coeff = J.rand(R, seed=1, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
m = J.rand(J.domain(A), seed=2, stream=0)
d, e, f = J.rand(R, seed=3, stream=0), J.rand(R, seed=4, stream=0), J.zeros(R)
mt = J.zeros(J.domain(A))
flat_d = J.rand(J.JetSpace(np.float32, L), seed=3, stream=0)
flat_e = J.rand(J.JetSpace(np.float32, L), seed=4, stream=0)
flat_f = J.zeros(J.JetSpace(np.float32, L))
host_blk = np.zeros((edge, edge, edge), dtype=np.float32, order="F")
dev_blk = J.rand(blk, seed=5, stream=0)

# the forward chooses its grid walk over its first calls (each one a timed trial of one candidate): the first row is what
# those calls average, the second the steady state every later call runs at
row("mul!(d, A, m)  [first 7 calls: walk trials]", (2 * L + n) * s, timeit(lambda: J.mul_(d, A, m)))
row("mul!(d, A, m)", (2 * L + n) * s, timeit(lambda: J.mul_(d, A, m), reps=10, warm=24))
row("mul!(m, A', d)", (2 * L + n) * s, timeit(lambda: J.mul_(mt, A.H, d)))
C = A.H @ A
row("mul!(y, A' o A, m)  [fused]", (L + 2 * n) * s, timeit(lambda: J.mul_(mt, C, m)))
row("getblock(d, 2)  [view, no copy]", 1, timeit(lambda: J.getblock(d, 2), reps=50))
row("getblock!(d, 2, device block)", 2 * n * s, timeit(lambda: J.getblock_(d, 2, dev_blk)))
row("setblock!(d, 2, device block)", 2 * n * s, timeit(lambda: J.setblock_(d, 2, dev_blk)))
row("setblock!(d, 2, scalar)", n * s, timeit(lambda: J.setblock_(d, 2, 3.14)))
row("getblock!(d, 2, host array)  [PCIe]", n * s, timeit(lambda: J.getblock_(d, 2, host_blk), reps=3))
row("f .= d .+ e", 3 * L * s, timeit(lambda: f.assign(d + e)))
row("f .= d .+ e  (base-case, flat)", 3 * L * s, timeit(lambda: flat_f.assign(flat_d + flat_e)))
row("f .= a*d .+ b*e .+ c*f", 4 * L * s, timeit(lambda: f.assign(0.3 * d + 0.5 * e + 0.2 * f)))
row("f .= d .* e", 3 * L * s, timeit(lambda: J.hadamard_(f, d, e)))
row("f .= a*d .+ b*e .+ c*f  [JIT broadcast]", 4 * L * s, timeit(lambda: J.broadcast_(f, "s0*x0 + s1*x1 + s2*x2", [d, e, f], [0.3, 0.5, 0.2])))
row("f .= exp.(-d .* d) .* e  [JIT broadcast]", 3 * L * s, timeit(lambda: J.broadcast_(f, "exp(-x0*x0) * x1", [d, e])))
row("f .= sqrt.(abs.(d)) ./ (1 .+ e)  [JIT broadcast]", 3 * L * s, timeit(lambda: J.broadcast_(f, "sqrt(abs(x0)) / (1 + x1)", [d, e])))
row("f .= f .* f  [JIT broadcast, in place]", 2 * L * s, timeit(lambda: J.broadcast_(f, "x0*x0", [f])))
row("fill!(f, 3.14)", L * s, timeit(lambda: J.fill_(f, 3.14)))
row("fill!  (base-case, flat)", L * s, timeit(lambda: J.fill_(flat_f, 3.14)))
row("dot(d, e)", 2 * L * s, timeit(lambda: J.dot(d, e)))
row("dot  (base-case, flat)", 2 * L * s, timeit(lambda: J.dot(flat_d, flat_e)))
row("norm(d)", L * s, timeit(lambda: J.norm(d)))
row("norm(d, 1)", L * s, timeit(lambda: J.norm(d, 1)))
row("norm(d, Inf)", L * s, timeit(lambda: J.norm(d, math.inf)))
row("norm(d, 3)", L * s, timeit(lambda: J.norm(d, 3)))
row("extrema(d)", L * s, timeit(lambda: J.extrema(d)))
row("rand(R) into existing storage  [counter RNG]", L * s, timeit(lambda: J._ffi.check(J._ffi.lib.jh_fill_uniform(f.handle, 9, 9, 0))))
row("reshape(flat, R)  [aliasing view]", 1, timeit(lambda: J.reshape(flat_d, R), reps=20))
row("convert(Array, d)  [device copy of the slab]", 2 * L * s, timeit(lambda: J.copyto_(flat_f, d)))
