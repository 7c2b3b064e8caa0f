#!/usr/bin/env python3
"""The split-row walk under rocprofv3: 1 GiB of Float32 as NROW x N (default 65536 x 4096), 20 adjoints, 20 fused A'A,
20 one-pass LSQR steps (used by tools/prof_split.sh)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J
from jets_jl_amd._ffi import check, lib
from jets_jl_amd.jetblock import _tall_native

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nrow = (1 << 28) // n
J.init(0)
spc = J.JetSpace(np.float32, n)
coeff = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
m = J.rand(J.domain(A), seed=2, stream=0)
d = J.rand(J.range(A), seed=3, stream=0)
mt, w = J.zeros(J.domain(A)), J.zeros(J.domain(A))
N = A.H @ A
out = C.c_double(0)
for _ in range(20):
    J.mul_(mt, A.H, d)
for _ in range(20):
    J.mul_(mt, N, m)
for _ in range(20):
    check(lib.jh_blockop_bidiag_step(_tall_native(A).handle, d.handle, m.handle, w.handle, 1.0, -0.5, C.byref(out)))
J.synchronize()
print(f"{nrow} x {n} Float32, parts {J.tune_get('last_adj_parts')}; algorithmic bytes: adjoint {(2 * nrow * n + n) * 4}, "
      f"fused A'A {(nrow * n + 2 * n) * 4}, step {(3 * nrow * n + 2 * n) * 4}")
