#!/usr/bin/env python3
"""The pipelined multi-GPU adjoint launches the ranged kernel (jh_blockop_mul_adj_range) once per chunk of the domain
vector.  What does chunking cost on one GPU (no exchange), per shard size and kernel shape?

    python tools/sweep_adj_chunks.py EDGE NROW [NROW ...]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J
from jets_jl_amd._ffi import lib, check
from jets_jl_amd import jetblock as _blk

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rows = [int(v) for v in sys.argv[2:]] or [128]
J.init(0)
n = edge ** 3
for nrow in rows:
    blk = J.JetSpace("float32", edge, edge, edge)
    coeff = J.rand(J.JetBSpace([blk] * nrow), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    d = J.rand(J.range(A), seed=3, stream=0)
    m = J.rand(J.domain(A), seed=2, stream=0)
    mt = J.zeros(J.domain(A))
    J.mul_(d, A, m)
    nat = _blk._native_op(A.jet.s["_native"], A.jet.s["ops"], A.jet.rng.eltype())
    nbytes = (2 * nrow * n + n) * 4

    def run(nchunks):
        step = -(-n // nchunks)
        step = -(-step // 16384) * 16384
        lo = 0
        while lo < n:
            cnt = min(step, n - lo)
            check(lib.jh_blockop_mul_adj_range(nat.handle, mt.handle, d.handle, lo, cnt))
            lo += cnt

    for shape in (dict(adj_wg=0, adj_unroll=0, adj_depth=0), dict(adj_wg=256, adj_unroll=4, adj_depth=4), dict(adj_wg=256, adj_unroll=4, adj_depth=2),
                  dict(adj_wg=512, adj_unroll=4, adj_depth=2), dict(adj_wg=256, adj_unroll=2, adj_depth=4), dict(adj_wg=256, adj_unroll=1, adj_depth=8),
                  dict(adj_wg=512, adj_unroll=2, adj_depth=4), dict(adj_wg=1024, adj_unroll=4, adj_depth=2)):
        J.tune(**shape)
        line = []
        for nchunks in (1, 2, 4, 8, 16):
            best = 1e9
            for _ in range(3):
                J.mul_(d, A, m)                         # the solver alternates forward / adjoint
                e0 = J.Event().record()
                run(nchunks)
                e1 = J.Event().record()
                best = min(best, e0.elapsed_ms(e1))
            line.append(f"{nchunks:2d} chunks {best:7.3f} ms {nbytes / best / 1e6:7.1f} GB/s")
        print(f"{nrow:5d} x {edge}^3  {str(shape):58s} " + " | ".join(line), flush=True)
    J.tune(adj_wg=0, adj_unroll=0, adj_depth=0)
    J.close(A)
    del A, coeff, d, m, mt, nat
    import gc
    gc.collect()
