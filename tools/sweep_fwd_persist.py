#!/usr/bin/env python3
"""Forward-only sweep including COLUMN-PERSISTENT shapes (fwd_group = nrow: a workgroup keeps its m tile in registers
and streams every block row through it, like the adjoint does), interleaved rounds, each launch event-timed.

    python tools/sweep_fwd_persist.py NBLOCKS EDGE
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J

nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 256
J.init(0)
n = edge ** 3
blk = J.JetSpace("float32", edge, edge, edge)
coeff = J.rand(J.JetBSpace([blk] * nblocks), seed=1, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
m = J.rand(J.domain(A), seed=2, stream=0)
d = J.rand(J.range(A), seed=3, stream=0)
mt = J.zeros(J.domain(A))
nbytes = (2 * nblocks * n + n) * 4
cfgs = [dict(fwd_wg=0, fwd_unroll=0, fwd_group=0, fwd_order=-1)]
for g in (nblocks, max(nblocks // 2, 1), max(nblocks // 4, 1), 64, 16):
    for (w, u) in ((256, 1), (256, 2), (256, 4), (512, 1), (512, 2), (512, 4), (1024, 1), (1024, 2), (1024, 4), (1024, 8)):
        for o in ((0,) if g == nblocks else (0, 1)):
            c = dict(fwd_wg=w, fwd_unroll=u, fwd_group=g, fwd_order=o)
            if c not in cfgs:
                cfgs.append(c)
res = {}
for rnd in range(3):
    for cfg in cfgs:
        J.tune(**cfg)
        J.mul_(d, A, m)
        J.mul_(mt, A.H, d)                      # the solver alternates: keep the adjoint between forwards
        e0 = J.Event().record()
        J.mul_(d, A, m)
        e1 = J.Event().record()
        res.setdefault(json.dumps(cfg, sort_keys=True), []).append(e0.elapsed_ms(e1))
for cfg, ms in sorted(res.items(), key=lambda kv: min(kv[1])):
    print(f"fwd  min {min(ms):8.3f} ms  med {sorted(ms)[len(ms) // 2]:8.3f} ms  {nbytes / min(ms) / 1e6:8.1f} GB/s  {cfg}")
