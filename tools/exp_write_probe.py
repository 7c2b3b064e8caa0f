#!/usr/bin/env python3
"""Round 4, placement for callers of the reference API: when the slab cache holds several 64 GiB slabs, which one should `A*m` write
into?  The library does not know at allocation time which operator will use the vector, so it needs a per-SLAB predictor.  Candidates:
the time of a pure fill of the slab (10 ms), of a fill of its first 8 GiB, against what matters: the forward (walk pinned to the
headline's candidate 7) from ONE coefficient slab into each candidate.  Four slabs of 64 GiB: coefficients in the first.

    python tools/exp_write_probe.py          (several processes in a row: placements differ from process to process)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

J.init(0)
nrow, edge = 1024, 256
blk = J.JetSpace(np.float32, edge, edge, edge)
R = J.JetBSpace([blk] * nrow)
C = J.rand(R, seed=1, stream=0)
cands = [J.Array(R, undef=True) for _ in range(3)]
m = J.rand(blk, seed=2, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in C.arrays])
J.op_tune_set(A, "fwd_walk", 7)


def timed(fn, reps=5):
    fn(); fn()
    J.synchronize()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        J.synchronize()
        best = min(best, e0.elapsed_ms(e1))
    return best


part = J.JetBSpace([blk] * 128)                       # the first 8 GiB of a candidate, as a view
for k, Y in enumerate(cands):
    t_fill = timed(lambda: J.fill_(Y, 1.0))
    head = J.reshape(J.getblock(Y, 0), blk) if False else None
    t_fwd = timed(lambda: J.mul_(Y, A, m))
    t_norm = timed(lambda: J.norm(Y))
    print(f"candidate {k}: fill {t_fill:7.3f} ms ({64 * 1.073741824 / t_fill:5.2f} TB/s)   forward into it {t_fwd:7.3f} ms   norm (read only) {t_norm:7.3f} ms", flush=True)
# and the other direction: each candidate as the COEFFICIENT slab writing into candidate 0 ... shows whether 'slow to write' is a property of the slab
for k, Y in enumerate(cands[1:], start=1):
    J.rand_(Y, seed=1, stream=0)
    B = J.blockop([[J.JopDiagonal(c)] for c in Y.arrays])
    J.op_tune_set(B, "fwd_walk", 7)
    t = timed(lambda: J.mul_(cands[0], B, m))
    print(f"coefficients in candidate {k} -> candidate 0: forward {t:7.3f} ms", flush=True)
    J.close(B)
