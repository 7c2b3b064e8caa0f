#!/usr/bin/env python3
"""One-pass LSQR step, the three ways it can walk (same bits): 0 plain walk (a workgroup lives for all rows of its tile),
1 the same with XCD-contiguous tiles, 2 chained row chunks (one batch of 8 rows per workgroup, ordered sum handed on) -- each
forced through jh_blockop_tune_set("step_mode"), then what the lazy per-operator measurement picks.   python tools/ab_step_chain.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J
from jets_jl_amd._ffi import lib, check
from jets_jl_amd import jetblock as _blk
J.init(0)
shapes = [(64, 256), (128, 256), (256, 256), (512, 256), (1024, 256), (64, 128), (1024, 128), (4096, 64)]
if len(sys.argv) > 2:
    shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
for nblocks, edge in shapes:
    n = edge ** 3
    blk = J.JetSpace("float32", edge, edge, edge)
    coeff = J.rand(J.JetBSpace([blk] * nblocks), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    nat = _blk._tall_native(A)
    u = J.rand(J.range(A), seed=3, stream=0); v = J.rand(J.domain(A), seed=2, stream=0); w = J.zeros(J.domain(A))
    out = C.c_double(0)
    def one_pass():
        check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, 1.0, -0.5, C.byref(out)))
    def timed(reps=7):
        best = 1e9
        for _ in range(reps):
            e0 = J.Event().record(); one_pass(); e1 = J.Event().record()
            best = min(best, e0.elapsed_ms(e1))
        return best
    b3 = (3 * nblocks * n + 2 * n) * 4
    res = {}
    for mode in (0, 1, 2, 0, 2):
        nat.tune_set("step_mode", mode)
        one_pass(); one_pass()
        t = timed()
        res[mode] = min(t, res.get(mode, 1e9))
    chained = J.tune_get("last_step_chain")
    nat.tune_set("step_mode", -1)
    for _ in range(10): one_pass()
    chosen = nat.tune_get("step_mode")
    t = timed()
    print(f"{nblocks:5d} x {edge}^3 one-pass step: plain {res[0]:8.3f} ms {b3/res[0]/1e6:7.1f} GB/s | XCD-contiguous {res[1]:8.3f} ms {b3/res[1]/1e6:7.1f} | "
          f"chained{'' if chained else ' (n/a: plain)'} {res[2]:8.3f} ms {b3/res[2]/1e6:7.1f} | measured choice {chosen}: {t:8.3f} ms {b3/t/1e6:7.1f} GB/s", flush=True)
    del u, v, w, coeff; J.close(A)
