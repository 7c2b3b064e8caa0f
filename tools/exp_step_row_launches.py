#!/usr/bin/env python3
"""One-pass LSQR step at 1024 x 256^3 walked in 1, 2, 4 row launches (knob adj_rows_per_launch): w's ordered sum continues
across launches (same bits), ||u||^2 accumulates on the device and is read back once."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J
from jets_jl_amd._ffi import lib, check
from jets_jl_amd import jetblock as _blk
J.init(0)
nblocks, edge = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 256
n = edge ** 3
blk = J.JetSpace("float32", edge, edge, edge)
coeff = J.rand(J.JetBSpace([blk] * nblocks), seed=1, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
nat = _blk._tall_native(A)
u = J.rand(J.range(A), seed=3, stream=0); v = J.rand(J.domain(A), seed=2, stream=0); w = J.zeros(J.domain(A))
out = C.c_double(0)
nat.tune_set("step_mode", 0)
b3 = (3 * nblocks * n + 2 * n) * 4
for rpl in (0, nblocks // 2, nblocks // 4, nblocks // 8, 0):
    J.tune(adj_rows_per_launch=rpl)
    ts = []
    for k in range(7):
        e0 = J.Event().record()
        check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, 1.0, -0.5, C.byref(out)))
        e1 = J.Event().record()
        ts.append(e0.elapsed_ms(e1))
    t = min(ts[2:])
    print(f"{nblocks} x {edge}^3 one-pass step, rows per launch {rpl or nblocks:5d}: {t:8.3f} ms  {b3 / t / 1e6:7.1f} GB/s   ||u||^2 {out.value:.6e}", flush=True)
