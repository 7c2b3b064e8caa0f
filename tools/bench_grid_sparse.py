#!/usr/bin/env python3
"""M x K grids of large blocks with STRUCTURED sparsity (block-diagonal, block-bidiagonal, arrow, banded, random fill) on the general kernels.
Unique (algorithmic) bytes: every diagonal once, every domain block that some non-zero block reads once, every range block that some non-zero block
touches read (as found) and written once.   python tools/bench_grid_sparse.py M K EDGE [PATTERN ...]
PATTERN: diag | bidiag | arrow | band5 | rand10 | rand25 | mixdiag (block-diagonal whose diagonal alternates diagonal / identity / scalar)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
edge = int(sys.argv[3]) if len(sys.argv) > 3 else 128
patterns = sys.argv[4:] or ["diag", "bidiag", "arrow", "band5", "rand10", "rand25", "mixdiag"]
J.init(0)
n = edge ** 3
spc = J.JetSpace("float32", edge, edge, edge)
RAGGED = os.environ.get("RAGGED", "0") == "1"      # block i has n - 4096 * (i % 3) elements: such grids run on the one-line general kernels (square grids only)


def kinds_of(pat):
    k = np.full((M, K), "z")
    ii, jj = np.meshgrid(np.arange(M), np.arange(K), indexing="ij")
    if pat == "diag":
        k[ii == jj] = "d"
    elif pat == "bidiag":
        k[(ii == jj) | (ii == jj + 1)] = "d"
    elif pat == "arrow":
        k[(ii == jj) | (ii == 0) | (jj == 0)] = "d"
    elif pat.startswith("band"):
        w = int(pat[4:]) // 2
        k[np.abs(ii - jj) <= w] = "d"
    elif pat.startswith("rand"):
        rng = np.random.default_rng(11)
        k[rng.random((M, K)) < int(pat[4:]) / 100.0] = "d"
    elif pat == "mixdiag":
        for i in range(min(M, K)):
            k[i, i] = "dis"[i % 3]
    else:
        raise SystemExit(f"unknown pattern {pat}")
    return k


def timed(fn, reps=6, warm=2):
    for _ in range(warm):
        fn()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record()
        fn()
        e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


knob = os.environ.get("GENERAL_LIST")
routes = [int(v) for v in knob.split(",")] if knob else [3, 2, 0, 1, 3, 2, 0, 1]
for pat in patterns:
    kinds = kinds_of(pat)
    ndiag = int((kinds == "d").sum())
    if RAGGED:
        assert M == K
        lens = [n - 4096 * (i % 3) for i in range(M)]
        spcs = [J.JetSpace("float32", v) for v in lens]
        kinds[np.array(lens)[:, None] != np.array(lens)[None, :]] = "z"      # an elementwise block is square
        ndiag = int((kinds == "d").sum())
    coeff = J.rand(J.JetBSpace([spc] * max(ndiag, 1)), seed=1, stream=0)
    rows, k = [], 0
    for i in range(M):
        row = []
        for j in range(K):
            kd = kinds[i, j]
            if RAGGED:
                if kd == "d":
                    row.append(J.JopDiagonal(J.rand(spcs[i], seed=1, stream=k))); k += 1
                else:
                    row.append(J.JopZeroBlock(spcs[j], spcs[i]))
                continue
            if kd == "d":
                row.append(J.JopDiagonal(coeff.arrays[k])); k += 1
            elif kd == "z":
                row.append(J.JopZeroBlock(spc, spc))
            elif kd == "i":
                row.append(J.JopIdentity(spc))
            else:
                row.append(J.JopLn(dom=spc, rng=spc, df=J.constdiag_df, df_adj=J.constdiag_df_adj, s={"a": 0.5 + i}))
        rows.append(row)
    A = J.blockop(rows)
    m = J.rand(J.domain(A), seed=2, stream=0)
    d = J.zeros(J.range(A))
    mt = J.zeros(J.domain(A))
    nz = kinds != "z"
    rows_t, cols_t = int(nz.any(axis=1).sum()), int(nz.any(axis=0).sum())
    b = n * 4
    fwd_bytes = (ndiag + cols_t + 2 * rows_t) * b
    adj_bytes = (ndiag + rows_t + K) * b                                   # the adjoint of a grid writes every domain block (1042)
    # algorithmic bytes per launch for tools/prof_any.sh (regex over the kernel's template name; the LIST instantiations end in `..., R, true>`)
    print(f"ALGO k_general_tile<.*,.false,.[0-9],.(true|false)>|k_block_fwd_general {fwd_bytes}")
    print(f"ALGO k_general_tile<.*,.true,.[0-9],.(true|false)>|k_block_adj_general {adj_bytes}", flush=True)
    bits = {}
    for gl in routes:
        try:
            J.tune(general_list=gl)
        except Exception:                                                   # a library without the knob: the baseline
            pass
        if edge <= 128:
            J.copyto_(d, J.rand(J.range(A), seed=9, stream=0))
            J.mul_(d, A, m)
            J.mul_(mt, A.H, d)
            got = (d.to_numpy().tobytes(), mt.to_numpy().tobytes())
            ref = bits.setdefault("ref", got)
            same = "==" if got == ref else "!="
        else:
            same = "??"
        walk = ""
        if gl == 1:                                                        # the automatic route measures over its first seven calls per direction
            for _ in range(9):
                J.mul_(d, A, m)
                J.mul_(mt, A.H, d)
            J.synchronize()
            J.mul_(d, A, m)
            J.mul_(mt, A.H, d)
            walk = f" [chose fwd {J.op_tune_get(A, 'gen_walk_fwd')} adj {J.op_tune_get(A, 'gen_walk_adj')}; last launch list {J.tune_get('last_general_list')}]"
        tf = timed(lambda: J.mul_(d, A, m))
        ta = timed(lambda: J.mul_(mt, A.H, d))
        print(f"{M} x {K} {pat:8s} of {edge}^3 ({int(nz.sum())} non-zero blocks, {ndiag} diagonals) general_list={gl} bits {same}: "
              f"forward {tf:7.3f} ms {fwd_bytes / tf / 1e6:7.1f} GB/s | adjoint {ta:7.3f} ms {adj_bytes / ta / 1e6:7.1f} GB/s{walk}", flush=True)
    del A, m, d, mt, coeff
