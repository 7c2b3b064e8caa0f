#!/usr/bin/env python3
"""Block-diagonal (and block-bidiagonal) operators of DENSE children: M x M grid, N x N Float32 matrices on the diagonal (BAND=1: and below it), zero
blocks elsewhere.  Algorithmic bytes = the matrices (+ the vectors).    python tools/bench_dense_blockdiag.py M N [BAND]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
band = int(sys.argv[3]) if len(sys.argv) > 3 else 0
J.init(0)
if os.environ.get("DENSE_GRID"):
    J.tune(dense_grid=int(os.environ["DENSE_GRID"]))
if os.environ.get("DENSE_DIRECT"):
    J.tune(dense_direct=int(os.environ["DENSE_DIRECT"]))
if os.environ.get("DENSE_LIST_CPW"):
    J.tune(dense_list_cpw=int(os.environ["DENSE_LIST_CPW"]))
if os.environ.get("SMALL_LOOP_MAX_KIB"):
    J.tune(small_loop_max_kib=int(os.environ["SMALL_LOOP_MAX_KIB"]))
spc = J.JetSpace("float32", N)
mat = J.JetSpace("float32", N, N)
nd = 0
rows = []
for i in range(M):
    row = []
    for j in range(M):
        if band == 2 or i == j or (band and i == j + 1):                # BAND = 2: every block dense (a full grid)
            row.append(J.JopDense(J.rand(mat, seed=7, stream=i * M + j))); nd += 1
        else:
            row.append(J.JopZeroBlock(spc, spc))
    rows.append(row)
A = J.blockop(rows)
m = J.rand(J.domain(A), seed=2, stream=0)
d = J.zeros(J.range(A))
mt = J.zeros(J.domain(A))


def timed(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record()
        fn()
        e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


b = nd * N * N * 4 + 3 * M * N * 4
# algorithmic bytes per launch of the children's kernels alone, for tools/prof_any.sh: the matrices, the input vector, the products
print(f"ALGO k_gemv_rows_list|k_gemv_cols_list|k_gemv_rows_mixed|k_gemv_cols_mixed {nd * N * N * 4 + M * N * 4 + nd * N * 4}", flush=True)
for dl in [int(v) for v in os.environ.get("DENSE_LIST", "1,0,1,0").split(",")]:
  J.tune(dense_list=dl)
  tf = timed(lambda: J.mul_(d, A, m))
  lf = J.tune_get("last_launches")
  rl = J.tune_get("last_dense_rl")
  ta = timed(lambda: J.mul_(mt, A.H, d))
  print(f"dense_list={dl} rl={rl} " +f"{M} x {M} {'full grid' if band == 2 else ('block-bidiagonal' if band else 'block-diagonal')} of {N} x {N} dense Float32 children ({nd} matrices, {b / 2**20:.0f} MiB): forward {tf:7.3f} ms {b / tf / 1e6:7.1f} GB/s "
      f"({lf} launches) | adjoint {ta:7.3f} ms {b / ta / 1e6:7.1f} GB/s ({J.tune_get('last_launches')} launches)", flush=True)
