#!/usr/bin/env python3
"""Fused chains through a tall operator against the stage-by-stage chain (round 6; jets.jl_amd/chains.py, jh_tall_chain.hip).

    python tools/bench_chains.py [nrow edge [dtype]]        default 256 256 f32

Algorithmic bytes (s = element size, N rows of n elements):  W o A: 3 N n s + n s;  (W o A)': 3 N n s + n s;  A' o W o A: 2 N n s + 2 n s;
M' o A' o W o A o M: 2 N n s + 4 n s;  A' o A + lam I: N n s + 3 n s.  The stage-by-stage chain moves a range-sized temporary in and out per stage."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J
from jets_jl_amd import chains

PEAK = 8.0e12
nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 256
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dt = {"f32": np.float32, "f64": np.float64, "c32": np.complex64, "c64": np.complex128}[sys.argv[3] if len(sys.argv) > 3 else "f32"]
J.init(0)
for kv in os.environ.get("JETS_TUNE", "").split(","):          # e.g. JETS_TUNE=adj_wg=512,adj_unroll=2 (A/B of launch shapes)
    if "=" in kv:
        J.tune(**{kv.split("=")[0]: int(kv.split("=")[1])})


def timed(fn, reps):
    fn()
    fn()
    J.synchronize()
    e0 = J.Event().record()
    for _ in range(reps):
        fn()
    e1 = J.Event().record()
    return e0.elapsed_ms(e1) / reps


blk = J.JetSpace(dt, edge, edge, edge)
n = blk.length()
s = np.dtype(dt).itemsize
R = J.JetBSpace([blk] * nrow)
coeff = J.rand(R, seed=1, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
w = J.rand(R, seed=5, stream=0)
W = J.JopDiagonal(w)
c = J.rand(blk, seed=6, stream=0)
M = J.JopDiagonal(c)
m = J.rand(J.domain(A), seed=2, stream=0)
y = J.zeros(J.domain(A))
d = J.zeros(J.range(A))
reps = max(3, int(2.0e11 / (nrow * n * s)))
print(f"# {nrow} x 1 of {edge}^3 {np.dtype(dt).name}: {nrow * n * s / 2**30:.1f} GiB per range-sized array, {reps} repetitions", flush=True)


def line(tag, op, out, x, nbytes):
    chains.ENABLED[0] = True
    before = chains.STATS["chain_calls"] + chains.STATS["sum_terms_fused"] + chains.STATS["bcast_calls"]
    ms_f = timed(lambda: J.mul_(out, op, x), reps)
    ran = chains.STATS["chain_calls"] + chains.STATS["sum_terms_fused"] + chains.STATS["bcast_calls"] - before
    chains.ENABLED[0] = False
    try:
        ms_u = timed(lambda: J.mul_(out, op, x), max(2, reps // 3))
    finally:
        chains.ENABLED[0] = True
    bw = nbytes / (ms_f * 1e-3)
    print(f"{tag:34s} fused {ms_f:9.3f} ms  {bw / 1e12:5.2f} TB/s  {100 * bw / PEAK:5.1f} % of 8 TB/s   stage by stage {ms_u:9.3f} ms   {ms_u / ms_f:5.2f}x"
          f"   ({'fused path ran' if ran else 'NOT FUSED'})", flush=True)


Nn = nrow * n * s
line("A' o W o A", J.compose(J.compose(A.H, W), A), y, m, 2 * Nn + 2 * n * s)
line("(W o A)' o (W o A)", J.compose(J.compose(W, A).H, J.compose(W, A)), y, m, 2 * Nn + 2 * n * s)
line("W o A", J.compose(W, A), d, m, 3 * Nn + n * s)
line("(W o A)'", J.compose(W, A).H, y, d, 3 * Nn + n * s)
line("M' o A' o W o A o M", J.compose(J.compose(J.compose(J.compose(M.H, A.H), W), A), M), y, m, 2 * Nn + 4 * n * s)
line("A' o A + 0.1 I", J.compose(A.H, A) + 0.1 * J.JopIdentity(J.domain(A)), y, m, Nn + 3 * n * s)
line("2.5 * (A' o W o A)", 2.5 * J.compose(J.compose(A.H, W), A), y, m, 2 * Nn + 2 * n * s)
# the plain fused A'A and adjoint of rounds 1-5 beside them (same box, same data)
ms = timed(lambda: J.mul_(y, J.compose(A.H, A), m), reps)
print(f"{'A^T o A (jh_blockop_normal_mul)':34s}       {ms:9.3f} ms  {(Nn + 2 * n * s) / ms / 1e9:5.2f} TB/s", flush=True)
ms = timed(lambda: J.mul_(y, A.H, d), reps)
print(f"{'A^T d   (jh_blockop_mul_adj)':34s}       {ms:9.3f} ms  {(2 * Nn + n * s) / ms / 1e9:5.2f} TB/s", flush=True)
ms = timed(lambda: J.mul_(d, A, m), reps)
print(f"{'A m     (jh_blockop_mul)':34s}       {ms:9.3f} ms  {(2 * Nn + n * s) / ms / 1e9:5.2f} TB/s", flush=True)

# the reference's composition benchmark (benchmark/benchmarks.jl:73-80: G = F o A o F o A, F: d .= m.^2, A a diagonal) on ONE block-sized plain space:
# algorithmic bytes 3 n s (m, a read; d written); stage by stage: four passes through three temporaries
F = J.JopSquare(blk)
A1 = J.JopDiagonal(c)
G = J.compose(J.compose(J.compose(F, A1), F), A1)
g = J.zeros(blk)
reps = max(20, int(2.0e10 / (n * s)))
line("G = F o A o F o A (one block)", G, g, m, 3 * n * s)
Jg = J.jacobian_(G, m)
line("jacobian(G) * dm", Jg, g, m, 5 * n * s)
line("jacobian(G)' * d", Jg.H, g, m, 5 * n * s)
