#!/usr/bin/env python3
"""Second cliff hunt (round 5, session 3): shapes tools/cliff_hunt.py does not visit -- the four element types, row lengths that are not
multiples of 16 bytes, ragged tall operators, adjointed children, sums / composites / scalar multiples of GRIDS (not tall operators),
tall operators of SQUARE children (f!, Jacobian).  GB/s of unique bytes (every coefficient once, every vector once), ~TOTAL MiB of
coefficients per case.       python tools/cliff_hunt2.py [TOTAL_MiB]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J

J.init(0)
total = (int(sys.argv[1]) if len(sys.argv) > 1 else 512) << 20


def timed(fn, reps=3):
    fn(); fn(); fn()
    best = 1e30
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


def report(tag, shape, uniq_f, tf, uniq_a=None, ta=None, extra=""):
    s = f"{tag:46s} {shape:26s} fwd {tf:8.3f} ms {uniq_f / tf / 1e6:6.0f} GB/s"
    worst = uniq_f / tf / 1e6
    if ta is not None:
        s += f" | adj {ta:8.3f} ms {uniq_a / ta / 1e6:6.0f} GB/s"
        worst = min(worst, uniq_a / ta / 1e6)
    flag = "  <-- CLIFF" if worst < 2500 and uniq_f > (48 << 20) else ""
    print(s + extra + flag, flush=True)


def grid(tag, dtype, nrow, ncol, lens_r, kind_of, adjoint_children=False):
    """nrow x ncol grid; row i has length lens_r[i]; all columns the length of ... square blocks only (lens_r[i] == lens_c[j] where non-zero)"""
    es = np.dtype(dtype).itemsize
    spcs = [J.JetSpace(dtype, int(n)) for n in lens_r]
    cells = [(i, j) for i in range(nrow) for j in range(ncol) if kind_of(i, j) == "diag"]
    slab = J.rand(J.JetBSpace([spcs[i] for i, _ in cells] or [spcs[0]]), seed=1, stream=0).arrays
    k = 0
    rows = []
    for i in range(nrow):
        row = []
        for j in range(ncol):
            kd = kind_of(i, j)
            if kd == "diag":
                op = J.JopDiagonal(slab[k]); k += 1
                row.append(op.H if adjoint_children else op)
            elif kd == "identity":
                row.append(J.JopIdentity(spcs[i]))
            elif kd == "scale":
                row.append(J.JopLn(dom=spcs[i], rng=spcs[i], df=J.constdiag_df, df_adj=J.constdiag_df_adj, s={"a": 0.5 + i}))
            else:
                row.append(J.JopZeroBlock(spcs[j] if ncol > 1 else spcs[i], spcs[i]))
        rows.append(row)
    A = J.blockop(rows)
    m = J.rand(J.domain(A), seed=2, stream=0)
    d = J.rand(J.range(A), seed=3, stream=0)
    mt = J.zeros(J.domain(A))
    ncoef = sum(int(lens_r[i]) for i, _ in cells)
    uniq = (ncoef + sum(int(x) for x in lens_r) + J.domain(A).length()) * es
    tf = timed(lambda: J.mul_(d, A, m))
    ta = timed(lambda: J.mul_(mt, A.H, d))
    report(tag, f"{nrow} x {ncol} {np.dtype(dtype).name}", uniq, tf, uniq, ta)
    return A, m, d, mt


diag = lambda i, j: "diag"
mixed = lambda i, j: ("diag", "identity", "scale", "zero")[(i + j) % 4]

# 1. element types x (tall, grid, mixed tall)
for dt in ("float32", "float64", "complex64", "complex128"):
    es = np.dtype(dt).itemsize
    n = total // es // 256
    grid("tall all diagonal", dt, 256, 1, [n] * 256, diag)
    grid("tall all diagonal, ADJOINTED children", dt, 256, 1, [n] * 256, diag, adjoint_children=True)
    grid("tall mixed kinds", dt, 256, 1, [n] * 256, mixed)
    n = total // es // 256
    grid("16 x 16 grid all diagonal", dt, 16, 16, [n] * 16, diag)
    grid("16 x 16 grid mixed kinds", dt, 16, 16, [n] * 16, mixed)

# 2. row lengths that break 16-byte alignment (odd n: every second row of the slab starts 4 bytes off a 16-byte boundary)
for dt in ("float32", "float64"):
    es = np.dtype(dt).itemsize
    n = total // es // 256 + 1
    grid("tall, ODD row length", dt, 256, 1, [n] * 256, diag)
    grid("16 x 16 grid, ODD block length", dt, 16, 16, [n] * 16, diag)
    grid("tall mixed, ODD row length", dt, 256, 1, [n] * 256, mixed)

# 3. ragged tall operators (rows of different lengths cannot share m: every row needs its own ... no -- a tall operator's rows all
#    map the SAME domain, so rows have equal length for diagonal children; raggedness lives in GRIDS: block-diagonal with unequal blocks)
lens = [(total // 4 // 64) * (1 + (i % 3)) // 2 for i in range(64)]
grid("64 x 64 block-diagonal, RAGGED blocks", "float32", 64, 64, lens, lambda i, j: "diag" if i == j else "zero")
lens = [(total // 4 // 64) + 3 * i for i in range(64)]
grid("64 x 64 block-diagonal, ragged + unaligned", "float32", 64, 64, lens, lambda i, j: "diag" if i == j else "zero")
grid("64 x 64 bidiagonal", "float32", 64, 64, [total // 4 // 128] * 64, lambda i, j: "diag" if j in (i, i + 1) else "zero")
grid("64 x 64 arrow (row 0, col 0, diagonal)", "float32", 64, 64, [total // 4 // 190] * 64, lambda i, j: "diag" if (i == 0 or j == 0 or i == j) else "zero")
grid("1 x 1024 wide", "float32", 1, 1024, [total // 4 // 1024], diag)
grid("2 x 512 wide", "float32", 2, 512, [total // 4 // 1024] * 2, diag)

# 4. sums / composites / scalar multiples of GRIDS and of mixed tall operators
n = total // 4 // 256
spc = J.JetSpace("float32", n)


def mk_grid(seed):
    slab = J.rand(J.JetBSpace([spc] * 256), seed=seed, stream=0).arrays
    return J.blockop([[J.JopDiagonal(slab[i * 16 + j]) for j in range(16)] for i in range(16)])


A1, A2, A3 = mk_grid(11), mk_grid(12), mk_grid(13)
m = J.rand(J.domain(A1), seed=2, stream=0); d = J.rand(J.range(A1), seed=3, stream=0); mt = J.zeros(J.domain(A1))
by = (256 + 32) * n * 4
S = A1 + A2 - A3
report("JetSum of 3 grids 16 x 16", "float32", 3 * by, timed(lambda: J.mul_(d, S, m)), 3 * by, timed(lambda: J.mul_(mt, S.H, d)))
try:
    C = J.compose(A2, A1)
    report("composite A2 o A1 of grids 16 x 16", "float32", 2 * by, timed(lambda: J.mul_(d, C, m)), 2 * by, timed(lambda: J.mul_(mt, C.H, d)))
except Exception as e:  # noqa: BLE001
    print("composite of grids:", repr(e))
N = J.compose(A1.H, A1)
report("normal equations A1' o A1 of a grid 16 x 16", "float32", 2 * by, timed(lambda: J.mul_(mt, N, m)))
Sc = J.scale_op(2.5, A1)
report("2.5 * grid 16 x 16", "float32", by, timed(lambda: J.mul_(d, Sc, m)), by, timed(lambda: J.mul_(mt, Sc.H, d)))

# tall mixed: sums and scalar multiples
slab = J.rand(J.JetBSpace([spc] * 256), seed=21, stream=0).arrays
T1 = J.blockop([[J.JopDiagonal(slab[i]) if i % 4 else J.JopIdentity(spc)] for i in range(256)])
slab2 = J.rand(J.JetBSpace([spc] * 256), seed=22, stream=0).arrays
T2 = J.blockop([[J.JopDiagonal(slab2[i])] for i in range(256)])
m = J.rand(J.domain(T1), seed=2, stream=0); d = J.rand(J.range(T1), seed=3, stream=0); mt = J.zeros(J.domain(T1))
byt = (2 * 256 + 1) * n * 4
S = T1 + T2
report("JetSum of tall mixed + tall diagonal", "256 x 1 float32", byt * 448 // 512 + 256 * n * 4, timed(lambda: J.mul_(d, S, m)), byt, timed(lambda: J.mul_(mt, S.H, d)))
N = J.compose(T1.H, T1)
report("normal equations of a tall MIXED operator", "256 x 1 float32", 192 * n * 4, timed(lambda: J.mul_(mt, N, m)))
Sc = J.scale_op(2.5, T1)
report("2.5 * tall mixed", "256 x 1 float32", (192 + 256) * n * 4, timed(lambda: J.mul_(d, Sc, m)), (192 + 256) * n * 4, timed(lambda: J.mul_(mt, Sc.H, d)))

# 5. nonlinear: tall operator of SQUARE children
F = J.blockop([[J.JopSquare(spc)] for _ in range(256)])
m = J.rand(J.domain(F), seed=2, stream=0); d = J.rand(J.range(F), seed=3, stream=0); mt = J.zeros(J.domain(F))
report("F(m), tall of 256 SQUARE children", "float32", 257 * n * 4, timed(lambda: J.mul_(d, F, m)))
Jm = J.jacobian_(F, m)
report("jacobian of it: J dm | J' dd", "float32", 258 * n * 4, timed(lambda: J.mul_(d, Jm, m)), 258 * n * 4, timed(lambda: J.mul_(mt, Jm.H, d)))
