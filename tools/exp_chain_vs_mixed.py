#!/usr/bin/env python3
"""Tall operators with rows of several kinds: the fused A'A (jh_blockop_normal_mul -> k_tall_diag_adj MIXED, MODE 1) and the adjoint against the chain kernels
with EMPTY stage lists (k_chain_adj NORMAL / ADJOINT: packed 8-byte row records instead of the 48-byte block table) on the same operator.

    python tools/exp_chain_vs_mixed.py [NROW N]..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J
from jets_jl_amd import chains
from jets_jl_amd._ffi import CHAIN_ADJOINT, CHAIN_NORMAL

J.init(0)


def timed(fn, reps=7):
    fn(); fn()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


def case(nrow, n, special):
    spc = J.JetSpace(np.float32, n)
    R = J.JetBSpace([spc] * nrow)
    rows = [[J.JopDiagonal(c)] for c in J.rand(R, seed=1, stream=0).arrays]
    for i in range(special):
        rows[(7 * i + 3) % nrow] = [J.JopIdentity(spc)]
    A = J.blockop(rows)
    m, y, d = J.rand(spc, seed=2, stream=0), J.zeros(spc), J.rand(R, seed=3, stream=0)
    st = chains.classify(A, None)
    hn = chains.ChainHandle(st, CHAIN_NORMAL, [], [], [])
    ha = chains.ChainHandle(chains.classify(A.H, None), CHAIN_ADJOINT, [], [], [])
    NA = J.compose(A.H, A)
    Nn = nrow * n * 4
    t_n, t_a = timed(lambda: J.mul_(y, NA, m)), timed(lambda: J.mul_(y, A.H, d))
    p1 = J.tune_get("last_adj_parts")
    c_n, c_a = timed(lambda: hn.apply(y, m)), timed(lambda: ha.apply(y, d))
    p2 = J.tune_get("last_adj_parts")
    t_f = timed(lambda: J.mul_(d, A, m))
    out = __import__("ctypes").c_double(0)
    from jets_jl_amd import jetblock as _blk
    from jets_jl_amd._ffi import lib
    nat = _blk._native_op(A.jet.s["_native"], A.jet.s["ops"], A.jet.rng.eltype())
    w = J.zeros(spc)
    st = lib.jh_blockop_bidiag_step(nat.handle, d.handle, m.handle, w.handle, 1.0, -0.5, __import__("ctypes").byref(out))
    t_s = timed(lambda: lib.jh_blockop_bidiag_step(nat.handle, d.handle, m.handle, w.handle, 1.0, -0.5, __import__("ctypes").byref(out))) if st == 0 else float("nan")
    print(f"{'':30s} forward {2 * Nn / t_f / 1e9:5.2f} TB/s   one-pass step {3 * Nn / t_s / 1e9:5.2f} TB/s", flush=True)
    print(f"{nrow:6d} x {n:9d}, {special} identity rows: A'A library {Nn / t_n / 1e9:5.2f} TB/s | chain kernel {Nn / c_n / 1e9:5.2f}     "
          f"adjoint library {2 * Nn / t_a / 1e9:5.2f} ({p1} parts) | chain kernel {2 * Nn / c_a / 1e9:5.2f} ({p2} parts)", flush=True)
    hn.close(); ha.close(); J.close(A)


args = [int(a) for a in sys.argv[1:]] or [256, 524288, 256, 1 << 24, 1024, 1 << 21, 4096, 1 << 18, 64, 1 << 24, 512, 127 ** 3]
for k in range(0, len(args), 2):
    for special in (0, 1):
        case(args[k], args[k + 1], special)
