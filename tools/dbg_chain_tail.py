#!/usr/bin/env python3
"""The chained one-pass step on rows off the 16-byte grid (17 rows of 4099 Float32 elements), nine calls alternating the workgroup sizes: prints which scalars of w
differ from the first call's.  Found the hand-off race of the first TAIL version (lanes past the end handed on packs computed from u as already updated in place:
1 ... 3 wrong tail scalars in one run out of three at 1024 lanes); clean since a lane hands on only the scalars it owns."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J
from jets_jl_amd._ffi import check, lib
from jets_jl_amd import jetblock
J.init(0)
nrow, n = 17, 4099
spc = J.JetSpace("float32", n)
diags = [J.rand(spc, seed=1, stream=i) for i in range(nrow)]
A = J.blockop([[J.JopDiagonal(g)] for g in diags])
nat = jetblock._native_op(A.jet.s["_native"], A.jet.s["ops"], A.jet.rng.eltype())
v = J.rand(spc, seed=2, stream=0)
hu = J.rand(J.range(A), seed=3, stream=0).to_numpy()
out = C.c_double(0)
ref = None
for wg in (512, 1024, 1024, 0, 256, 1024, 0, 512, 1024):
    J.tune(step_chain=1, adj_wg=wg)
    u = J.from_numpy(hu, J.range(A)); w = J.rand(spc, seed=53, stream=0)
    check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, 0.75, -1.375, C.byref(out)))
    hw = w.to_numpy().ravel()
    if ref is None: ref = hw.copy()
    bad = np.flatnonzero(hw != ref)
    print("wg", wg, "chain", J.tune_get("last_step_chain"), "bad", bad[:8], bad.size, hw[-4:], flush=True)
