#!/usr/bin/env python3
"""Kernel-shape sweep in the solver's context: forward and adjoint ALTERNATE (fwd, adj, fwd, adj ...),
each launch timed with its own HIP events; interleaved rounds in one process.

    python tools/sweep_pair.py NBLOCKS EDGE [fwd|adj|both]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J

nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 256
which = sys.argv[3] if len(sys.argv) > 3 else "both"
rounds, reps = 2, 3
J.init(0)
n = edge ** 3
blk = J.JetSpace("float32", edge, edge, edge)
coeff = J.rand(J.JetBSpace([blk] * nblocks), seed=1, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
m = J.rand(J.domain(A), seed=2, stream=0)
d = J.rand(J.range(A), seed=3, stream=0)
mt = J.zeros(J.domain(A))
nbytes = (2 * nblocks * n + n) * 4
AUTO_F = dict(fwd_wg=0, fwd_unroll=0, fwd_group=0, fwd_order=-1)
AUTO_A = dict(adj_wg=0, adj_unroll=0, adj_depth=0)


def pair(fcfg, acfg):
    J.tune(**fcfg)
    J.tune(**acfg)
    J.mul_(d, A, m)
    J.mul_(mt, A.H, d)
    tf = ta = 0.0
    for _ in range(reps):
        e0 = J.Event().record()
        J.mul_(d, A, m)
        e1 = J.Event().record()
        J.mul_(mt, A.H, d)
        e2 = J.Event().record()
        tf += e0.elapsed_ms(e1)
        ta += e1.elapsed_ms(e2)
    return tf / reps, ta / reps


fwd_cfgs = [dict(fwd_wg=w, fwd_unroll=u, fwd_group=g, fwd_order=o)
            for o in (0, 1) for (w, u) in ((256, 1), (256, 4), (512, 1), (512, 4), (1024, 4), (1024, 8)) for g in (2, 4, 8, 16, 32) if g <= nblocks]
adj_cfgs = [dict(adj_wg=w, adj_unroll=u, adj_depth=dp) for w in (256, 512, 1024) for u in (1, 2, 4) for dp in (1, 2, 4, 8) if not (u == 4 and dp == 8)]
res = {}
for rnd in range(rounds):
    if which in ("fwd", "both"):
        for cfg in fwd_cfgs:
            tf, _ = pair(cfg, AUTO_A)
            res.setdefault(("fwd", json.dumps(cfg, sort_keys=True)), []).append(tf)
    if which in ("adj", "both"):
        for cfg in adj_cfgs:
            _, ta = pair(AUTO_F, cfg)
            res.setdefault(("adj", json.dumps(cfg, sort_keys=True)), []).append(ta)
    tf, ta = pair(AUTO_F, AUTO_A)
    res.setdefault(("fwd", "AUTO"), []).append(tf)
    res.setdefault(("adj", "AUTO"), []).append(ta)
for (kind, cfg), ms in sorted(res.items(), key=lambda kv: (kv[0][0], min(kv[1]))):
    print(f"{kind:4s} min {min(ms):8.3f} ms  med {sorted(ms)[len(ms) // 2]:8.3f} ms  {nbytes / min(ms) / 1e6:8.1f} GB/s  {cfg}")
