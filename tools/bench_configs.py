#!/usr/bin/env python3
"""One line per BASELINE.json config on ONE MI355X (HIP-event timed, inputs resident, counter-RNG data).

    python tools/bench_configs.py > profiles/bench_configs_r01.txt

config 1  4x4 JopBlock of identity JopLn, JetSpace(Float64,128): dot-product test + time per pair (launch-bound)
config 2  64x1 tall diagonal, 128^3 Float32                     : fwd+adj pairs/s (1 GiB working set: partly MALL-resident)
config 3  A'oA on a 256x1 tall diagonal (128^3 and 256^3)       : fused launch vs the unfused chain
config 4  1024x1 tall diagonal, 256^3 Float32                   : the whole operator on one GPU, and the per-rank shards
                                                                  (512 / 256 / 128 rows) a 2 / 4 / 8 GPU row partition runs locally
config 5  100 LSQR iterations on config 4, b = A x_true         : ms/iteration, relative error of x; the same solve by CGLS
                                                                  (two passes) and by CG through the fused A'A (one pass of a)
Algorithmic bytes as in SURVEY.md 8d: pair = 4*N*n*s + 2*n*s; fused A'A = N*n*s + 2*n*s.
"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

PEAK = 8.0e12
J.init(0)


def timed(fn, reps):
    fn()
    fn()
    J.synchronize()
    e0 = J.Event().record()
    for _ in range(reps):
        fn()
    e1 = J.Event().record()
    return e0.elapsed_ms(e1) / reps


def tall(nrow, edge):
    blk = J.JetSpace(np.float32, edge, edge, edge)
    coeff = J.rand(J.JetBSpace([blk] * nrow), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    return A, coeff


def pair_line(tag, nrow, edge, reps):
    A, coeff = tall(nrow, edge)
    n = edge ** 3
    m = J.rand(J.domain(A), seed=2, stream=0)
    d = J.zeros(J.range(A))
    mt = J.zeros(J.domain(A))

    def pair():
        J.mul_(d, A, m)
        J.mul_(mt, A.H, d)

    k = 0                                                      # let the forward's lazy per-operator measurement finish first (as bench.py does)
    while J.op_tune_get(A, "fwd_walk") == -1 and k < 24:
        J.mul_(d, A, m)
        J.synchronize()
        k += 1
        if J.op_tune_get(A, "fwd_trials") == 0:
            break
    ms = timed(pair, reps)
    nbytes = (4 * nrow * n + 2 * n) * 4
    bw = nbytes / (ms * 1e-3)
    print(f"{tag:58s} {ms:9.3f} ms/pair {1e3 / ms:9.2f} pairs/s {bw / 1e9:8.1f} GB/s {100 * bw / PEAK:5.1f} % of 8 TB/s", flush=True)
    J.close(A)
    del A, coeff, m, d, mt
    gc.collect()


# ---------------------------------------------------------------- config 1
spc = J.JetSpace(np.float64, 128)
A1 = J.blockop([[J.JopIdentity(spc) for _ in range(4)] for _ in range(4)])
m1, d1 = J.rand(J.domain(A1), seed=2, stream=0), J.rand(J.range(A1), seed=3, stream=0)
lhs, rhs = J.dot_product_test(A1, m1, d1)
out_d, out_m = J.zeros(J.range(A1)), J.zeros(J.domain(A1))


def pair1():
    J.mul_(out_d, A1, m1)
    J.mul_(out_m, A1.H, d1)


t0 = time.perf_counter()
ms1 = timed(pair1, 200)
print(f"{'config 1: 4x4 identity, Float64, n=128':58s} {ms1 * 1e3:9.1f} us/pair (launch-bound, 2 launches)   dot-product test "
      f"|lhs-rhs|/|lhs+rhs| = {abs(lhs - rhs) / abs(lhs + rhs):.1e}", flush=True)

# ---------------------------------------------------------------- config 2
pair_line("config 2: 64x1 diagonal, 128^3 Float32", 64, 128, 50)

# ---------------------------------------------------------------- config 3
for edge, reps in ((128, 20), (256, 10)):
    A, coeff = tall(256, edge)
    n = edge ** 3
    m = J.rand(J.domain(A), seed=2, stream=0)
    y = J.zeros(J.domain(A))
    N = A.H @ A
    ms_f = timed(lambda: J.mul_(y, N, m), reps)
    d = J.zeros(J.range(A))

    def unfused():
        J.mul_(d, A, m)
        J.mul_(y, A.H, d)

    ms_u = timed(unfused, reps)
    bf = (256 * n + 2 * n) * 4
    bu = (4 * 256 * n + 2 * n) * 4
    print(f"{'config 3: A^T o A on 256x1 diagonal, %d^3 Float32' % edge:58s} fused {ms_f:8.3f} ms ({bf / ms_f / 1e6:7.1f} GB/s, "
          f"{100 * bf / (ms_f * 1e-3) / PEAK:4.1f} %)   unfused pair {ms_u:8.3f} ms ({bu / ms_u / 1e6:7.1f} GB/s)   "
          f"speed-up {ms_u / ms_f:4.2f}x", flush=True)
    J.close(A)
    del A, coeff, N, d, y, m
    gc.collect()

# ---------------------------------------------------------------- config 3, weighted (round 6): A' o W o A on 256 x 256^3
# the normal equations of a WEIGHTED least-squares problem (data weights W on the block range): a chain of depth 3 through the tall operator -- ONE pass
# of the chain kernels (jh_chain_*: 2 N n s + 2 n s bytes) against the reference's stage-by-stage chain through two range-sized temporaries
from jets_jl_amd import chains as _chains

A, coeff = tall(256, 256)
n = 256 ** 3
w = J.rand(J.range(A), seed=5, stream=0)
N = J.compose(J.compose(A.H, J.JopDiagonal(w)), A)
m = J.rand(J.domain(A), seed=2, stream=0)
y = J.zeros(J.domain(A))
ms_f = timed(lambda: J.mul_(y, N, m), 10)
_chains.ENABLED[0] = False
ms_u = timed(lambda: J.mul_(y, N, m), 4)
_chains.ENABLED[0] = True
bf = (2 * 256 * n + 2 * n) * 4
print(f"{'config 3w: A^T o W o A on 256x1 diagonal, 256^3 Float32':58s} fused {ms_f:8.3f} ms ({bf / ms_f / 1e6:7.1f} GB/s, "
      f"{100 * bf / (ms_f * 1e-3) / PEAK:4.1f} %)   stage by stage {ms_u:8.3f} ms   speed-up {ms_u / ms_f:4.2f}x", flush=True)
J.close(A)
del A, coeff, N, w, y, m
gc.collect()

# ---------------------------------------------------------------- config 4
for nrow, note in ((128, "rank-local shard at 8 GPUs"), (256, "rank-local shard at 4 GPUs"), (512, "rank-local shard at 2 GPUs"),
                   (1024, "whole operator on 1 GPU")):
    pair_line(f"config 4: {nrow}x1 diagonal, 256^3 Float32 ({note})", nrow, 256, 10 if nrow < 1024 else 8)

# ---------------------------------------------------------------- config 5
A, coeff = tall(1024, 256)
x_true = J.rand(J.domain(A), seed=4, stream=0)
b = J.mul(A, x_true)
J.synchronize()
t0 = time.perf_counter()
res = J.lsqr(A, b, maxiter=100, atol=0.0, btol=0.0, overwrite_b=True, force_maxiter=True)
J.synchronize()
wall = time.perf_counter() - t0
err = J.zeros(J.domain(A))
J.lincomb_(err, [1.0, -1.0], [res.x, x_true])
rel = J.norm(err) / J.norm(x_true)
print(f"{'config 5: 100 LSQR iterations on config 4':58s} {1e3 * wall / max(res.itn, 1):9.2f} ms/iteration ({res.itn} iterations, "
      f"{wall:6.2f} s wall)   ||x - x_true|| / ||x_true|| = {rel:.2e}", flush=True)

for name, solve in (("CGLS", lambda rhs: J.cgls(A, rhs, maxiter=100, atol=0.0, btol=0.0, overwrite_b=True, force_maxiter=True)),
                    ("CGNR (fused A'A)", lambda rhs: J.cgnr(A, rhs, maxiter=100, atol=0.0, btol=0.0, force_maxiter=True))):
    J.mul_(b, A, x_true)                                       # LSQR / CGLS used b's storage
    J.synchronize()
    t0 = time.perf_counter()
    res = solve(b)
    J.synchronize()
    wall = time.perf_counter() - t0
    J.lincomb_(err, [1.0, -1.0], [res.x, x_true])
    rel = J.norm(err) / J.norm(x_true)
    print(f"{'config 5: 100 %s iterations on config 4' % name:58s} {1e3 * wall / max(res.itn, 1):9.2f} ms/iteration ({res.itn} iterations, "
          f"{wall:6.2f} s wall)   ||x - x_true|| / ||x_true|| = {rel:.2e}", flush=True)

# (after the headline-size sections: the 128 GiB operator of configs 4 / 5 should meet the slab cache as configs 1-3 leave it)
J.close(A)
del A, coeff, b, x_true, err, res
gc.collect()
# config 3g (round 6): the multi-parameter twin -- A'A of a 64 x 4 grid of diagonals, plain and with 4 regularisation rows (lam * I on the diagonal, zero blocks elsewhere)
for reg in (0, 1):
    spc = J.JetSpace(np.float32, 256, 256, 256)
    n, NN, KK = 256 ** 3, 64, 4
    coeff = J.rand(J.JetBSpace([spc] * (NN * KK)), seed=1, stream=0)
    rows = [[J.JopDiagonal(coeff.arrays[i * KK + j]) for j in range(KK)] for i in range(NN)]
    if reg:
        rows += [[(J.JopLn(dom=spc, rng=spc, df=J.constdiag_df, df_adj=J.constdiag_df_adj, s={"a": 0.5}) if j == k else J.JopZeroBlock(spc, spc)) for j in range(KK)]
                 for k in range(KK)]
    A = J.blockop(rows)
    N = J.compose(A.H, A)
    m, y = J.rand(J.domain(A), seed=2, stream=0), J.zeros(J.domain(A))
    ms_f = timed(lambda: J.mul_(y, N, m), 10)
    J.tune(grid_normal=0)
    ms_u = timed(lambda: J.mul_(y, N, m), 4)
    J.tune(grid_normal=1)
    bf = (NN * KK + 2 * KK) * n * 4
    tag = f"config 3g: A^T o A on a 64{' + 4 reg.' if reg else ''} x 4 grid, 256^3 Float32"
    print(f"{tag:58s} fused {ms_f:8.3f} ms ({bf / ms_f / 1e6:7.1f} GB/s, {100 * bf / (ms_f * 1e-3) / PEAK:4.1f} %)   two stages {ms_u:8.3f} ms   speed-up {ms_u / ms_f:4.2f}x", flush=True)
    J.close(A)
    del A, coeff, N, y, m, rows
    gc.collect()

