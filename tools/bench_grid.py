#!/usr/bin/env python3
"""M x K grids of large diagonal blocks on the general kernels: forward d_i = d_i + sum_j a_ij .* m_j, adjoint m_j = sum_i conj(a_ij) .* d_i.
Algorithmic (unique) bytes: M K n coefficients + the two vectors (+ d as found in the forward).   python tools/bench_grid.py M K EDGE"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
edge = int(sys.argv[3]) if len(sys.argv) > 3 else 256
J.init(0)
n = edge ** 3
spc = J.JetSpace("float32", edge, edge, edge)
coeff = J.rand(J.JetBSpace([spc] * (M * K)), seed=1, stream=0)
colmajor = os.environ.get("GRID_COLMAJOR", "0") == "1"           # where block (i, j)'s coefficients sit in the slab: row by row, or column by column
pad = int(os.environ.get("GRID_PAD_BLOCKS", "0"))                 # GRID_PAD_BLOCKS=1: one unused block after every row / column of blocks (breaks the power-of-two strides)
if pad:
    coeff = J.rand(J.JetBSpace([spc] * (M * K + (K if colmajor else M) * pad)), seed=1, stream=0)
idx = (lambda i, j: j * (M + pad) + i) if colmajor else (lambda i, j: i * (K + pad) + j)
A = J.blockop([[J.JopDiagonal(coeff.arrays[idx(i, j)]) for j in range(K)] for i in range(M)])
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
routes = [(1, 1), (1, 2), (1, 4), (1, 8), (1, 0), (0, 0)]               # (grid_diag, grid_tile): tiled automatic / 2 x 1 / 4 x 2 / 8 x 1 (lines x packs per lane); k_grid_diag; the general kernels
if os.environ.get("GRID_ROUTES"):
    routes = [tuple(int(v) for v in r.split(":")) for r in os.environ["GRID_ROUTES"].split(",")]
bands = [int(v) for v in os.environ.get("GENERAL_BAND", "").split(",") if v]      # tiles per band of the grid kernels (8 / 16 / 32 / 64), cycled
for it, (gd, gt) in enumerate(routes * 2):
    xcd = 1
    J.tune(general_xcd=xcd, grid_diag=gd, grid_tile=gt)
    if os.environ.get("GENERAL_TILE"):                                    # with grid_diag = 0: which general kernel takes the grid
        J.tune(general_tile=int(os.environ["GENERAL_TILE"]))
    if bands:
        J.tune(general_band=bands[it % len(bands)])
    tf = timed(lambda: J.mul_(d, A, m))
    ta = timed(lambda: J.mul_(mt, A.H, d))
    print(f"{M} x {K} of {edge}^3 {'col-major' if colmajor else 'row-major'} slab pad {pad} grid_diag={gd} grid_tile={gt}{' band ' + str(bands[it % len(bands)]) if bands else ''}: forward {tf:7.3f} ms {(M * K + K + 2 * M) * b / tf / 1e6:7.1f} GB/s | adjoint {ta:7.3f} ms {(M * K + M + K) * b / ta / 1e6:7.1f} GB/s", flush=True)
