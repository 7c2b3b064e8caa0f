#!/usr/bin/env python3
"""Fused JetSum of K tall diagonal operators (jh_blocksum_mul / _mul_adj): d = sum_k +-(A_k m) reads K coefficient slabs and writes d once
(unfused: 5 range-sized streams per term).   python tools/bench_jetsum.py [K] [NROW] [EDGE]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nrow = int(sys.argv[2]) if len(sys.argv) > 2 else 64
edge = int(sys.argv[3]) if len(sys.argv) > 3 else 256
J.init(0)
n = edge ** 3
spc = J.JetSpace("float32", edge, edge, edge)
ops = []
for k in range(K):
    coeff = J.rand(J.JetBSpace([spc] * nrow), seed=10 + k, stream=0)
    ops.append(J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays]))
# SUM_SCALE=wide: every term a numpy float64 scalar times its operator -- Julia's Float64 against Float32 elements, the reference's own
# `1.0*A1 - 2.0*A2 + 3.0*A3` (src/Jets.jl:686); round 5: still ONE pass (WIDE instantiations); SUM_SCALE=narrow: Python floats, T(a)
import numpy as np
scale_mode = os.environ.get("SUM_SCALE", "")
term = (lambda k: np.float64(0.5 + 0.37 * k) * ops[k]) if scale_mode == "wide" else ((lambda k: (0.5 + 0.37 * k) * ops[k]) if scale_mode == "narrow" else (lambda k: ops[k]))
S = term(0)
for k in range(1, K):
    S = S + term(k) if k % 2 else S - term(k)
m = J.rand(spc, seed=2, stream=0)
d = J.zeros(J.range(ops[0]))
mt = J.zeros(spc)


def timed(fn, reps=7, warm=3):
    for _ in range(warm):
        fn()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record()
        fn()
        e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


b = n * 4
if os.environ.get("SUM_FWD_GROUP"):
    J.tune(fwd_group=int(os.environ["SUM_FWD_GROUP"]))                # rows per workgroup of the fused forward (default 4)
if os.environ.get("SUM_FWD_UNROLL"):
    J.tune(fwd_unroll=int(os.environ["SUM_FWD_UNROLL"]))
print(f"ALGO k_tall_sum_(fwd|adj) {((K + 1) * nrow + 1) * b}", flush=True)       # per launch when the sum is ONE launch (K <= terms per launch): tools/prof_any.sh
groups = tuple(int(v) for v in os.environ.get("SUM_GROUPS", "16,8,16,8").split(","))
if os.environ.get("SUM_SWEEP"):                                     # round 5: rows in flight per workgroup of the 9..16-term kernels, alternating in one process
    for rnd in range(2):
        for g, dd in ((1, 1), (2, 2), (1, 1)):
            J.tune(fwd_group=g, adj_depth=dd, sum_group=16, sum_adj_group=16)
            tf = timed(lambda: J.mul_(d, S, m))
            ta = timed(lambda: J.mul_(mt, S.H, d))
            print(f"JetSum of {K} tall {nrow} x {edge}^3 operators: forward {g} row(s) in flight {tf:7.3f} ms {((K + 1) * nrow + 1) * b / tf / 1e6:7.1f} GB/s | adjoint {dd} row(s) in flight {ta:7.3f} ms {((K + 1) * nrow + 1) * b / ta / 1e6:7.1f} GB/s", flush=True)
    J.tune(fwd_group=0, adj_depth=0)
    sys.exit(0)
for group in groups:                                        # forward terms per launch: 16 (round 4) against round 3's 8, alternating in one process
    J.tune(sum_group=group, sum_adj_group=group)                    # (the adjoint: 16 or 8 accumulators per launch)
    tf = timed(lambda: J.mul_(d, S, m))
    ta = timed(lambda: J.mul_(mt, S.H, d))
    print(f"JetSum of {K} tall {nrow} x {edge}^3 operators{' (' + scale_mode + ' scalars)' if scale_mode else ''}, {group} terms per launch: forward {tf:7.3f} ms {((K + 1) * nrow + 1) * b / tf / 1e6:7.1f} GB/s | adjoint {ta:7.3f} ms {((K + 1) * nrow + 1) * b / ta / 1e6:7.1f} GB/s")
