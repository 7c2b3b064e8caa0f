#!/usr/bin/env python3
"""Bandwidth of the general (mixed-kind / M x K) block kernels on large blocks, HIP-event timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

J.init(0)


def timeit(fn, reps=5):
    fn(); fn()
    e0 = J.Event().record()
    for _ in range(reps):
        fn()
    e1 = J.Event().record()
    return e0.elapsed_ms(e1) / reps


def case(name, kinds, n, dt=np.float32):
    spc = J.JetSpace(dt, n)
    s = np.dtype(dt).itemsize
    rows, ndiag, nnz = [], 0, 0
    for i, row in enumerate(kinds):
        r = []
        for j, k in enumerate(row):
            if k == "z":
                r.append(J.JopZeroBlock(spc, spc))
            elif k == "i":
                r.append(J.JopIdentity(spc)); nnz += 1
            elif k == "s":
                r.append(J.JopLn(dom=spc, rng=spc, df=J.constdiag_df, df_adj=J.constdiag_df_adj, s={"a": 0.5 + i})); nnz += 1
            else:
                r.append(J.JopDiagonal(J.rand(spc, seed=9, stream=100 * i + j))); ndiag += 1; nnz += 1
        rows.append(r)
    A = J.blockop(rows)
    nrow, ncol = len(kinds), len(kinds[0])
    m = J.rand(J.domain(A), seed=2, stream=0)
    d = J.rand(J.range(A), seed=3, stream=0)
    mt = J.zeros(J.domain(A))
    # algorithmic bytes: every DIAG coefficient once, every used input block once per use, outputs once (+ read when accumulating)
    fwd_b = (ndiag + nnz + nrow * (2 if ncol > 1 else 1)) * n * s
    adj_b = (ndiag + nnz + ncol) * n * s
    tf = timeit(lambda: J.mul_(d, A, m))
    ta = timeit(lambda: J.mul_(mt, A.H, d))
    print(f"{name:28s} n={n:>9d} fwd {tf:8.3f} ms {fwd_b / tf / 1e6:8.1f} GB/s | adj {ta:8.3f} ms {adj_b / ta / 1e6:8.1f} GB/s")


n = 128 ** 3
case("tall 64x1 mixed d/s/i", [["d"] if i % 3 == 0 else (["s"] if i % 3 == 1 else ["i"]) for i in range(64)], n)
case("tall 64x1 all diag (fast)", [["d"] for _ in range(64)], n)
case("wide 1x64 diag", [["d"] * 64], n)
case("4x4 identity (config 1 big)", [["i"] * 4 for _ in range(4)], 256 ** 3)
case("8x8 diag", [["d"] * 8 for _ in range(8)], n)
case("3x4 mixed with zeros", [["d", "i", "d", "s"], ["d", "z", "d", "d"], ["s", "d", "d", "z"]], 256 ** 3)
case("wide 1x16 diag", [["d"] * 16], 256 ** 3)
case("8x8 diag", [["d"] * 8 for _ in range(8)], 256 ** 3)
case("tall 64x1 mixed d/s/i", [["d"] if i % 3 == 0 else (["s"] if i % 3 == 1 else ["i"]) for i in range(64)], 256 ** 3)


def dense_case(nr, nc, dt=np.float32):
    Am = J.rand(J.JetSpace(dt, nr, nc), seed=11, stream=0)
    A = J.JopDense(Am)
    x, y = J.rand(J.JetSpace(dt, nc), seed=12, stream=0), J.rand(J.JetSpace(dt, nr), seed=13, stream=0)
    out_r, out_c = J.zeros(J.JetSpace(dt, nr)), J.zeros(J.JetSpace(dt, nc))
    b = nr * nc * np.dtype(dt).itemsize
    tf = timeit(lambda: J.mul_(out_r, A, x))
    ta = timeit(lambda: J.mul_(out_c, A.H, y))
    print(f"dense {nr:>8d} x {nc:<8d}          fwd {tf:8.3f} ms {b / tf / 1e6:8.1f} GB/s | adj {ta:8.3f} ms {b / ta / 1e6:8.1f} GB/s")


dense_case(32768, 32768)
dense_case(4096, 262144)
dense_case(4194304, 256)
dense_case(1024, 1024)
