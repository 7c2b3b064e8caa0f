#!/usr/bin/env python3
"""A/B of the tall-forward grid order (fwd_order 0/1) x rows per workgroup, interleaved rounds."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J

J.init(0)
nblocks, edge = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = edge ** 3
blk = J.JetSpace("float32", edge, edge, edge)
coeff = J.rand(J.JetBSpace([blk] * nblocks), seed=1, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
m = J.rand(J.domain(A), seed=2, stream=0)
d = J.rand(J.range(A), seed=3, stream=0)
nbytes = (2 * nblocks * n + n) * 4
res = {}
cfgs = [dict(fwd_order=o, fwd_group=g, fwd_unroll=u, fwd_wg=w) for o in (0, 1) for g in (8, 16, 32, 64, 128, 256) for (u, w) in ((8, 1024), (4, 512), (4, 256))]
for rnd in range(3):
    for cfg in cfgs:
        J.tune(**cfg)
        J.mul_(d, A, m)
        e0 = J.Event().record()
        for _ in range(3):
            J.mul_(d, A, m)
        e1 = J.Event().record()
        res.setdefault(json.dumps(cfg, sort_keys=True), []).append(e0.elapsed_ms(e1) / 3)
for cfg, ms in sorted(res.items(), key=lambda kv: min(kv[1])):
    print(f"min {min(ms):8.3f} ms  med {sorted(ms)[1]:8.3f} ms  {nbytes / min(ms) / 1e6:8.1f} GB/s  {cfg}")
