#!/usr/bin/env python3
"""M x K grids of large blocks of MIXED kinds (diagonal, zero, identity, scalar) on the general kernels, unique (algorithmic) bytes: every diagonal once,
every domain block once, every range block read (as found) and written once.   python tools/bench_grid_mixed.py M K EDGE"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
edge = int(sys.argv[3]) if len(sys.argv) > 3 else 256
J.init(0)
n = edge ** 3
spc = J.JetSpace("float32", edge, edge, edge)
rng = np.random.default_rng(7)
kinds = rng.choice(list(os.environ.get("GRID_KINDS", "dddzis")), size=(M, K))   # GRID_KINDS=ddz: diagonals and zero blocks only, ...
ndiag = int((kinds == "d").sum())
coeff = J.rand(J.JetBSpace([spc] * max(ndiag, 1)), seed=1, stream=0)
rows, k = [], 0
for i in range(M):
    row = []
    for j in range(K):
        kd = kinds[i, j]
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


b = n * 4
# algorithmic bytes per launch, for tools/prof_any.sh (regex over the kernel's template name: TRANSPOSED = false / true)
print(f"ALGO k_general_tile<.*false|k_block_fwd_general {(ndiag + K + 2 * M) * b}")
print(f"ALGO k_general_tile<.*true|k_block_adj_general {(ndiag + M + K) * b}", flush=True)
if os.environ.get("GENERAL_TILE_U"):
    J.tune(fwd_unroll=int(os.environ["GENERAL_TILE_U"]))
lists = [int(v) for v in os.environ.get("GENERAL_LIST", "").split(",") if v]         # step-list route per iteration (0 plain, 2 four-line lists, 3 per-line lists), cycled
knob = os.environ.get("GENERAL_TILE")
bands = [int(v) for v in os.environ.get("GENERAL_BAND", "").split(",") if v]      # tiles per band of the general kernels (8 / 16 / 32 / 64), cycled with the tile knob
bits = {}
for it, gt in enumerate([int(v) for v in knob.split(",")] if knob else [1, 0, 1, 0]):
    J.tune(general_tile=gt)
    if lists:
        J.tune(general_list=lists[it % len(lists)])
    if edge <= 128:                                               # every route's bits against the first route's (forward from a dirty d, adjoint)
        J.copyto_(d, J.rand(J.range(A), seed=9, stream=0))
        J.mul_(d, A, m)
        J.mul_(mt, A.H, d)
        got = (d.to_numpy().tobytes(), mt.to_numpy().tobytes())
        ref = bits.setdefault("ref", got)
        print(f"general_tile={gt}: bits {'==' if got == ref else '!='} first route's", flush=True)
    if bands:
        J.tune(general_band=bands[it % len(bands)])
    tf = timed(lambda: J.mul_(d, A, m))
    ta = timed(lambda: J.mul_(mt, A.H, d))
    print(f"{M} x {K} mixed grid of {edge}^3 ({ndiag} diagonals of {M * K} blocks) general_tile={gt}{' list ' + str(lists[it % len(lists)]) if lists else ''}{' band ' + str(bands[it % len(bands)]) if bands else ''}: forward {tf:7.3f} ms {(ndiag + K + 2 * M) * b / tf / 1e6:7.1f} GB/s | "
          f"adjoint {ta:7.3f} ms {(ndiag + M + K) * b / ta / 1e6:7.1f} GB/s", flush=True)
