#!/usr/bin/env python3
"""Round 5, structural attempts at the one-pass LSQR step beside the plain walk and the chained row chunks (8 rows per workgroup): chunks of
16 rows (512 or 256 lanes) and of 32 rows (256 lanes) -- half / a quarter of the hand-offs and of the re-reads of v -- alternating in one
process; w and the last block of u compared bit for bit.  (The first version of this tool also timed a software-pipelined plain walk:
34.27 against 34.50 ms at 1024 x 256^3, dropped -- profiles/exp_r05_step_pipe.txt.)   python tools/exp_step_pipe.py [NROW EDGE]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J
from jets_jl_amd._ffi import lib, check
from jets_jl_amd import jetblock as _blk
J.init(0)
shapes = [(1024, 256), (256, 256), (128, 256), (1024, 128)]
if len(sys.argv) > 2:
    shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
variants = [("plain", 0, 0, 0), ("chained", 2, 0, 0), ("chained, 16-row chunks", 2, 0, -16), ("chained, 16 rows x 256 lanes", 2, 0, -1016), ("chained, 32-row chunks", 2, 0, -32)]
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
    def select(mode, pipe, depth):
        nat.tune_set("step_mode", mode)
        if depth < 0:                                                  # chained with 16 / 32 rows per chunk (forced: step_chain = 1 takes the first workgroup size that divides, or adj_wg)
            J.tune(step_chunk=(-depth) % 1000, step_chain=1, adj_depth=0, adj_unroll=0, adj_wg=256 if depth <= -1000 else 0)
        else:
            J.tune(step_chunk=8, step_chain=-1, adj_depth=0, adj_unroll=0, adj_wg=0)
    def timed(reps=6):
        best = 1e9
        for _ in range(reps):
            e0 = J.Event().record(); one_pass(); e1 = J.Event().record()
            best = min(best, e0.elapsed_ms(e1))
        return best
    b3 = (3 * nblocks * n + 2 * n) * 4
    res = {}
    for rnd in range(2):
        for name, mode, pipe, depth in variants:
            select(mode, pipe, depth)
            one_pass(); one_pass()
            res[name] = min(timed(), res.get(name, 1e9))
    ref = None
    for name, mode, pipe, depth in variants:
        select(mode, pipe, depth)
        one_pass(0.75, 0.0)
        got = (w.to_numpy().tobytes(), J.getblock(u, nblocks - 1).to_numpy().tobytes())
        ref = ref or got
        assert got == ref, f"{name}: w / the last block of u differ from the plain walk"
    select(-1, 0, 0)
    J.tune(step_chunk=8, step_chain=-1)
    print(f"{nblocks:5d} x {edge}^3 one-pass step, TB/s over 3 N n s: " + " | ".join(f"{k} {b3 / res[k] / 1e9:5.2f} ({res[k]:.3f} ms)" for k, *_ in variants) + "   [bits identical]", flush=True)
    del u, v, w, coeff; J.close(A)
