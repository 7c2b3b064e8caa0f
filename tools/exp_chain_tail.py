#!/usr/bin/env python3
"""EXPERIMENT: the one-pass step on rows off the 16-byte grid: plain walk (step_chain = 0) against the chained walk's TAIL instantiations (step_chain = 1)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J
from jets_jl_amd._ffi import check, lib
from jets_jl_amd import jetblock
J.init(0)
nrow, e = int(sys.argv[1]), int(sys.argv[2])
def timed(fn, reps=5):
    fn(); fn()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best
spc = J.JetSpace("float32", e, e, e); n = e ** 3
diags = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0).arrays
A = J.blockop([[J.JopDiagonal(g)] for g in diags])
nat = jetblock._native_op(A.jet.s["_native"], A.jet.s["ops"], A.jet.rng.eltype())
v = J.rand(spc, seed=2, stream=0); u = J.rand(J.range(A), seed=3, stream=0); w = J.zeros(spc)
out = C.c_double(0)
by = (3 * nrow + 2) * n * 4
for rnd in range(2):
    for mode, wg in ((0, 0), (1, 1024), (1, 512), (1, 256)):
        J.tune(step_chain=mode, adj_wg=wg if mode else 0)
        t = timed(lambda: check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, 1.0, -0.5, C.byref(out))))
        print(f"{nrow} x {e}^3 step_chain={mode} wg={wg}: {t:8.3f} ms {by / t / 1e6:6.0f} GB/s  (chunks {J.tune_get('last_step_chain')})", flush=True)
J.tune(step_chain=-1, adj_wg=0)
