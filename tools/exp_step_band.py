#!/usr/bin/env python3
"""Round 5: the chained one-pass LSQR step (k_tall_diag_bidiag_chain: one batch of 8 rows of one tile per workgroup) in COLUMN bands -- `step_band`
consecutive tiles of chunk 0, the same tiles of chunk 1, ..., then the next band -- against tiles fastest over the whole row (step_band = 0) and
against the plain walk; w and ||u||^2 are compared bit for bit between the walks.   python tools/exp_step_band.py [NROW EDGE] [bands ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J
from jets_jl_amd._ffi import lib, check
from jets_jl_amd import jetblock as _blk
J.init(0)
shapes = [(1024, 256), (256, 256), (128, 256), (1024, 128)]
bands = [0, 64, 128, 256, 512, 1024, 2048]
if len(sys.argv) > 2:
    shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
if len(sys.argv) > 3:
    bands = [int(v) for v in sys.argv[3:]]
for nblocks, edge in shapes:
    n = edge ** 3
    blk = J.JetSpace("float32", edge, edge, edge)
    coeff = J.rand(J.JetBSpace([blk] * nblocks), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    nat = _blk._tall_native(A)
    u = J.rand(J.range(A), seed=3, stream=0); v = J.rand(J.domain(A), seed=2, stream=0); w = J.zeros(J.domain(A))
    out = C.c_double(0)
    def one_pass(alpha=1.0, beta=-0.5):
        check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, alpha, beta, C.byref(out)))
    def timed(reps=6):
        best = 1e9
        for _ in range(reps):
            e0 = J.Event().record(); one_pass(); e1 = J.Event().record()
            best = min(best, e0.elapsed_ms(e1))
        return best
    b3 = (3 * nblocks * n + 2 * n) * 4
    res = {}
    for rnd in range(2):
        for key in ["plain"] + bands:
            if key == "plain":
                nat.tune_set("step_mode", 0)
            else:
                nat.tune_set("step_mode", 2); J.tune(step_band=key)
            one_pass(); one_pass()
            t = timed()
            res[key] = min(t, res.get(key, 1e9))
    # same bits: one step from identical state under every walk (beta = 0: u is write-only, so the state repeats)
    ref = None
    for key in ["plain"] + bands:
        if key == "plain":
            nat.tune_set("step_mode", 0)
        else:
            nat.tune_set("step_mode", 2); J.tune(step_band=key)
        one_pass(0.75, 0.0)
        got = (w.to_numpy().tobytes(), J.getblock(u, nblocks - 1).to_numpy().tobytes(), out.value)
        if ref is None:
            ref = got
        assert got[:2] == ref[:2], f"walk {key}: w / last block of u differ from the plain walk"
        assert abs(got[2] - ref[2]) <= 1e-12 * abs(ref[2]), f"walk {key}: ||u||^2"       # (fp64 partial sums folded in the walk's own order: tolerance)
    J.tune(step_band=-1); nat.tune_set("step_mode", -1)
    print(f"{nblocks:5d} x {edge}^3 one-pass step, TB/s over 3 N n s: " + " | ".join(f"{k if k == 'plain' else 'band ' + str(k)} {b3 / res[k] / 1e9:5.2f} ({res[k]:.3f} ms)" for k in ["plain"] + bands) + "   [bits identical]", flush=True)
    del u, v, w, coeff; J.close(A)
