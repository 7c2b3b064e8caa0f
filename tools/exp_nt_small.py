#!/usr/bin/env python3
"""Round 5: should an operator that FITS the Infinity Cache (256 MiB) be streamed with nontemporal loads?  Every kernel of the tall family loaded
its coefficients nontemporal whatever the operator's size (knob nt = 1), so a solver that re-reads the same 64 MiB of coefficients every iteration
fetched them from HBM every time.  Here: working sets of 32 MiB ... 1 GiB, nt = 0 (temporal) against nt = 2 (always nontemporal), alternating in
one process -- the fused A'A (k_tall_diag_adj MODE 1), the forward + adjoint pair, the one-pass step (k_tall_diag_bidiag), and what one MORE
iteration of CG on the normal equations / LSQR costs inside the graph-replayed loops (k_cg_normal, the step with device-resident coefficients).
    python tools/exp_nt_small.py > profiles/exp_r05_nt_small.txt"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J
from jets_jl_amd._ffi import lib, check
from jets_jl_amd import jetblock as _blk

J.init(0)
shapes = [(32, 64), (64, 64), (128, 64), (256, 64), (64, 128), (128, 128)]          # 32, 64, 128, 256, 512, 1024 MiB of Float32 coefficients
if len(sys.argv) > 2:
    shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
print("# rows x block | coefficients | nt | fused A'A us (TB/s over N n s) | fwd + adj pair us (TB/s over 4 N n s) | one-pass step us (TB/s over 3 N n s) | CGNR / LSQR us per further iteration", flush=True)
for nrow, edge in shapes:
    n = edge ** 3
    blk = J.JetSpace(np.float32, edge, edge, edge)
    coeff = J.rand(J.JetBSpace([blk] * nrow), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    nat = _blk._tall_native(A)
    x_true = J.rand(J.domain(A), seed=4, stream=0)
    b = J.mul(A, x_true)
    y = J.zeros(J.domain(A)); w = J.zeros(J.domain(A)); u = J.rand(J.range(A), seed=3, stream=0)
    N = A.H @ A
    out = C.c_double(0)
    nat.tune_set("step_mode", 0)                                         # the plain walk (what the graph loops run)

    def timed(fn, reps):
        for _ in range(5):
            fn()
        J.synchronize()
        e0 = J.Event().record()
        for _ in range(reps):
            fn()
        e1 = J.Event().record()
        return 1e3 * e0.elapsed_ms(e1) / reps                            # us

    def pair():
        J.mul_(u, A, x_true)
        J.mul_(y, A.H, u)

    def step():
        check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, x_true.handle, w.handle, 1.0, 0.0, None))

    reps = max(20, min(400, int(2e9 / (nrow * n * 4))))
    res = {}
    for rnd in range(2):
        for nt in (0, 2):
            J.tune(nt=nt)
            t_n = timed(lambda: J.mul_(y, N, x_true), reps)
            t_p = timed(pair, max(10, reps // 2))
            t_s = timed(step, max(10, reps // 2))
            marg = []
            for solve in (lambda k: J.cgnr(A, b, maxiter=k, atol=0.0, btol=0.0, force_maxiter=True),
                          lambda k: J.lsqr(A, b, maxiter=k, atol=0.0, btol=0.0, conlim=0.0, force_maxiter=True, overwrite_b=True)):
                took = {}
                for iters in (12, 12, 112):
                    J.mul_(b, A, x_true)
                    J.synchronize()
                    t0 = time.perf_counter()
                    r = solve(iters)
                    J.synchronize()
                    took[iters] = (time.perf_counter() - t0, r.itn)
                marg.append(1e6 * (took[112][0] - took[12][0]) / max(took[112][1] - took[12][1], 1))
            key = nt
            cur = (t_n, t_p, t_s, marg[0], marg[1])
            res[key] = tuple(min(a, b_) for a, b_ in zip(cur, res[key])) if key in res else cur
    by = nrow * n * 4
    for nt in (0, 2):
        t_n, t_p, t_s, m0, m1 = res[nt]
        print(f"{nrow:4d} x {edge}^3 | {by / 2**20:6.0f} MiB | nt={nt} | A'A {t_n:8.1f} us {by / t_n / 1e6:6.2f} | pair {t_p:8.1f} us {4 * by / t_p / 1e6:6.2f} | step {t_s:8.1f} us {3 * by / t_s / 1e6:6.2f} | "
              f"CGNR {m0:7.1f} us  LSQR {m1:7.1f} us", flush=True)
    J.tune(nt=1)
    J.close(A)
    del A, coeff, N, b, y, w, u, x_true
    import gc
    gc.collect()
