#!/usr/bin/env python3
"""Sweep of the fused forward-update kernel (u <- alpha A v + beta u, ||u||^2) in the LSQR context
(alternating with the fused adjoint update), HIP-event timed, interleaved rounds."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J
from jets_jl_amd._ffi import lib, check
from jets_jl_amd import jetblock

nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 256
J.init(0)
n = edge ** 3
blk = J.JetSpace("float32", edge, edge, edge)
coeff = J.rand(J.JetBSpace([blk] * nblocks), seed=1, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
nat = jetblock._native_op(A.jet.s["_native"], A.jet.s["ops"], A.jet.rng.eltype())
v = J.rand(J.domain(A), seed=2, stream=0)
u = J.rand(J.range(A), seed=3, stream=0)
fb, ab = (3 * nblocks * n + n) * 4, (2 * nblocks * n + 2 * n) * 4
res = {}
cfgs = [dict(fwd_wg=w, fwd_unroll=un, fwd_group=g) for (w, un) in ((1024, 8), (256, 4), (256, 1)) for g in (1, 2, 4, 8, 16, 32)]
for rnd in range(2):
    for cfg in cfgs:
        J.tune(**cfg)
        tf = ta = 0.0
        for rep in range(4):
            e0 = J.Event().record()
            check(lib.jh_blockop_mul_axpby(nat.handle, u.handle, v.handle, 1e-3, 0.5, None))
            e1 = J.Event().record()
            check(lib.jh_blockop_mul_adj_axpby(nat.handle, v.handle, u.handle, 1e-3, 0.5, 1.0, None))
            e2 = J.Event().record()
            if rep:
                tf += e0.elapsed_ms(e1); ta += e1.elapsed_ms(e2)
        res.setdefault(json.dumps(cfg, sort_keys=True), []).append((tf / 3, ta / 3))
for cfg, t in sorted(res.items(), key=lambda kv: min(x[0] for x in kv[1])):
    mf, ma = min(x[0] for x in t), min(x[1] for x in t)
    print(f"fwd_update min {mf:8.3f} ms {fb / mf / 1e6:8.1f} GB/s | adj_update {ma:8.3f} ms {ab / ma / 1e6:8.1f} GB/s  {cfg}")
