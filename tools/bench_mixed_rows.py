#!/usr/bin/env python3
"""Tall operators whose rows are not all plain diagonals (zero blocks, identity / scalar rows): the tall kernels with a per-row
kind vs the all-diagonal fast path (round 1 gave few scalar rows constant diagonals instead and sent everything else to the general kernels).

    python tools/bench_mixed_rows.py NROW EDGE
Algorithmic bytes: a DIAG row moves 2 blocks per kernel (a_i + d_i), an IDENTITY / SCALE row 1 (d_i), a ZERO row 0."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J
from jets_jl_amd._ffi import lib, check
from jets_jl_amd import jetblock as _blk

nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 256
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 256
J.init(0)
n = edge ** 3
spc = J.JetSpace("float32", edge, edge, edge)
coeff = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0)


def build(kinds):
    rows = []
    for i, k in enumerate(kinds):
        if k == "d":
            rows.append([J.JopDiagonal(coeff.arrays[i])])
        elif k == "z":
            rows.append([J.JopZeroBlock(spc, spc)])
        elif k == "i":
            rows.append([J.JopIdentity(spc)])
        else:
            rows.append([J.JopLn(dom=spc, rng=spc, df=J.constdiag_df, df_adj=J.constdiag_df_adj, s={"a": 0.5})])
    return J.blockop(rows)


def timed(fn, reps=5):
    fn(); fn()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record()
        fn()
        e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


m = J.rand(spc, seed=2, stream=0)
d = J.rand(J.JetBSpace([spc] * nrow), seed=3, stream=0)
mt, w = J.zeros(spc), J.zeros(spc)
out = C.c_double(0)
cases = [("all diagonal (fast path)", "d" * nrow, "1"),
         ("1 identity row", "d" * (nrow - 1) + "i", "0"),
         ("1 zero row", "d" * (nrow - 1) + "z", "0"),
         ("25 % zero rows", "".join("z" if i % 4 == 3 else "d" for i in range(nrow)), "0"),
         ("50 % scalar rows", "".join("s" if i % 2 else "d" for i in range(nrow)), "0"),
         ("data + zero + lambda*I", "d" * (nrow - 2) + "zs", "0")]
for name, kinds, densify in cases:
    A = build(kinds)
    nd = kinds.count("d")
    ns = nrow - nd - kinds.count("z")
    blk = n * 4
    b_pair = (2 * nd + ns) * blk * 2 + 2 * blk
    b_normal = nd * blk + 2 * blk
    b_step = (3 * nd + 2 * ns + 2 * kinds.count("z")) * blk + 2 * blk         # a zero / scalar row of u is still read and written by the step
    C_ = A.H @ A
    nat = _blk._tall_native(A)
    t_f = timed(lambda: J.mul_(d, A, m))
    t_a = timed(lambda: J.mul_(mt, A.H, d))
    t_n = timed(lambda: J.mul_(w, C_, m))
    step = lambda: check(lib.jh_blockop_bidiag_step(nat.handle, d.handle, m.handle, w.handle, 1.0, -0.5, C.byref(out)))
    J.op_tune_set(A, "step_mode", 0)
    t_s = timed(step)
    J.op_tune_set(A, "step_mode", 2)                                          # chained row chunks (rows of any elementwise kind)
    t_c = timed(step)
    chained = J.tune_get("last_step_chain") > 0
    extra = ""
    for ch in [int(v) for v in os.environ.get("STEP_CHUNKS", "").split(",") if v]:   # STEP_CHUNKS=8,32: the chained walk again with that many rows per chunk
        J.tune(step_chunk=ch)
        t_x = timed(step)
        extra += f" | chunk {ch}: {t_x:7.3f} ms {b_step / t_x / 1e6:7.1f} GB/s ({J.tune_get('last_step_chain')} chunks)"
        J.tune(step_chunk=0)
    print(f"{nrow} x {edge}^3  {name:28s} pair {t_f + t_a:8.3f} ms {b_pair / (t_f + t_a) / 1e6:7.1f} GB/s | A'A {t_n:7.3f} ms {b_normal / t_n / 1e6:7.1f} GB/s | "
          f"one-pass step {t_s:7.3f} ms {b_step / t_s / 1e6:7.1f} GB/s | chained {t_c:7.3f} ms {b_step / t_c / 1e6:7.1f} GB/s{'' if chained else ' (not taken)'}{extra}",
          flush=True)
    J.close(A)
