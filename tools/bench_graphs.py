#!/usr/bin/env python3
"""The reference's own 3x4 mixed block operator shape (test/runtests.jl:622-695) with dense children, one adjointed:
per-block loop eager / replayed as a hipGraph (jh_tune "graphs") vs the whole loop in ONE launch (jh_tune "small_loop",
k_block_loop_small; matrices up to 256 KiB).

    python tools/bench_graphs.py > profiles/bench_graphs_r02.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

J.init(0)
for n in (10, 64, 128, 256, 512, 2048):
    spc = J.JetSpace(np.float32, n)
    blk = lambda s: J.JopDense(J.rand(J.JetSpace(np.float32, n, n), seed=1, stream=s))
    Z = lambda: J.JopZeroBlock(spc, spc)
    A = J.blockop([[blk(1), blk(2), blk(3), blk(4)], [blk(5), Z(), blk(6), blk(7).H], [blk(8), blk(9), blk(10), Z()]])
    m, d = J.rand(J.domain(A), seed=2, stream=0), J.zeros(J.range(A))
    mt = J.zeros(J.domain(A))
    out = []
    for graphs, small in ((0, 0), (1, 0), (1, 1)):
        J.tune(graphs=graphs, small_loop=small)
        for _ in range(3):
            J.mul_(d, A, m)
            J.mul_(mt, A.H, d)
        J.synchronize()
        reps = 200
        e0 = J.Event().record()
        for _ in range(reps):
            J.mul_(d, A, m)
            J.mul_(mt, A.H, d)
        e1 = J.Event().record()
        out.append(1e3 * e0.elapsed_ms(e1) / reps)
    J.tune(graphs=1, small_loop=1)
    print(f"3x4 block operator, dense {n:5d} x {n:<5d} children (10 dense + 2 zero blocks): per-block loop eager {out[0]:8.1f} us/pair   "
          f"graph replay {out[1]:8.1f}   one launch each way {out[2]:8.1f} us/pair ({'per-block loop: a child exceeds 256 KiB' if n * n * 4 > (256 << 10) else f'{out[0] / out[2]:4.1f}x vs eager'})")
