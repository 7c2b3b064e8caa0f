#!/usr/bin/env python3
"""Wide (1 x K) operators of large diagonal blocks next to the tall operator of the same blocks: forward d = d + sum_j a_j .* m_j
(reads 2 K n + n, writes n), adjoint m_j = conj(a_j) .* d (reads K n + n, writes K n).   python tools/bench_wide.py K EDGE"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J

K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 256
J.init(0)
n = edge ** 3
spc = J.JetSpace("float32", edge, edge, edge)
coeff = J.rand(J.JetBSpace([spc] * K), seed=1, stream=0)


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


W = J.blockop([[J.JopDiagonal(c) for c in coeff.arrays]])            # 1 x K
T = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])            # K x 1
mw = J.rand(J.domain(W), seed=2, stream=0)
dw = J.zeros(J.range(W))
mtw = J.zeros(J.domain(W))
mt = J.rand(J.domain(T), seed=2, stream=0)
dt_ = J.zeros(J.range(T))
mtt = J.zeros(J.domain(T))
for _ in range(18):
    J.mul_(dt_, T, mt)
    J.synchronize()
b = n * 4
t_wf = timed(lambda: J.mul_(dw, W, mw))
t_wa = timed(lambda: J.mul_(mtw, W.H, dw))
t_tf = timed(lambda: J.mul_(dt_, T, mt))
t_ta = timed(lambda: J.mul_(mtt, T.H, dt_))
print(f"wide 1 x {K} of {edge}^3: forward {t_wf:7.3f} ms {(2 * K + 2) * b / t_wf / 1e6:7.1f} GB/s | adjoint {t_wa:7.3f} ms {(2 * K + 1) * b / t_wa / 1e6:7.1f} GB/s")
print(f"tall {K} x 1 of {edge}^3: forward {t_tf:7.3f} ms {(2 * K + 1) * b / t_tf / 1e6:7.1f} GB/s | adjoint {t_ta:7.3f} ms {(2 * K + 1) * b / t_ta / 1e6:7.1f} GB/s")
