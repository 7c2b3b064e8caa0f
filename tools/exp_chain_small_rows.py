#!/usr/bin/env python3
"""Does the chained one-pass step pay for rows SMALLER than 16 MiB (where it is not a measured candidate today)?  Plain walk vs chained
row chunks forced with the knob step_chain=1 (any workgroup size whose tiles divide the row).   python tools/exp_chain_small_rows.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J
from jets_jl_amd._ffi import lib, check
from jets_jl_amd import jetblock as _blk
J.init(0)
for nblocks, edge in ((64, 128), (256, 128), (1024, 128), (512, 160), (4096, 64)):
    n = edge ** 3
    blk = J.JetSpace("float32", edge, edge, edge)
    coeff = J.rand(J.JetBSpace([blk] * nblocks), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    nat = _blk._tall_native(A)
    u = J.rand(J.range(A), seed=3, stream=0); v = J.rand(J.domain(A), seed=2, stream=0); w = J.zeros(J.domain(A))
    out = C.c_double(0)
    def one_pass():
        check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, 1.0, -0.5, C.byref(out)))
    def timed(reps=9):
        one_pass(); one_pass()
        best = 1e9
        for _ in range(reps):
            e0 = J.Event().record(); one_pass(); e1 = J.Event().record()
            best = min(best, e0.elapsed_ms(e1))
        return best
    b3 = (3 * nblocks * n + 2 * n) * 4
    res = []
    for rnd in range(2):
        J.tune(step_chain=0); t0 = timed()
        J.tune(step_chain=1); t1 = timed(); chunks = J.tune_get("last_step_chain")
        res.append((t0, t1, chunks))
    J.tune(step_chain=-1)
    print(f"{nblocks} x {edge}^3 ({n * 4 / 2**20:.1f} MiB rows): plain {min(r[0] for r in res):7.3f} ms {b3 / min(r[0] for r in res) / 1e6:7.1f} GB/s | chained {min(r[1] for r in res):7.3f} ms {b3 / min(r[1] for r in res) / 1e6:7.1f} GB/s ({res[0][2]} chunks)", flush=True)
    J.close(A)
