#!/usr/bin/env python3
"""Per-block norms / inner products of a block vector: one pass (jh_norm_blocks / jh_dot_blocks, round 6) against a reduction per block through the
whole-vector entry points on views (a launch and a host round trip each).

    python tools/bench_block_reductions.py [nblocks edge]        default 1024 256"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 256
J.init(0)
blk = J.JetSpace(np.float32, edge, edge, edge)
R = J.JetBSpace([blk] * nb)
x, y = J.rand(R, seed=1, stream=0), None
nbytes = nb * blk.length() * 4


def wall(fn, reps):
    fn()
    J.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    J.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print(f"# {nb} blocks of {edge}^3 Float32 ({nbytes / 2**30:.1f} GiB)")
for p in (2, 1, np.inf):
    ms = wall(lambda: J.norm_blocks(x, p), 5)
    print(f"norm_blocks(x, {p}):  {ms:8.3f} ms  {nbytes / ms / 1e9:5.2f} TB/s  ({100 * nbytes / ms / 1e9 / 8:4.1f} % of 8 TB/s)   all {nb} block norms in one pass")
ms_w = wall(lambda: J.norm(x, 2), 5)
print(f"norm(x, 2) (whole vector):  {ms_w:8.3f} ms  {nbytes / ms_w / 1e9:5.2f} TB/s")
views = [J.getblock(x, i) for i in range(nb)]
ms_v = wall(lambda: [J.norm(v, 2) for v in views], 2)
print(f"[norm(getblock(x, i)) for i]:  {ms_v:8.3f} ms  {nbytes / ms_v / 1e9:5.2f} TB/s   ({ms_v / nb * 1e3:.1f} us per block)")
if nb * blk.length() * 8 < 200e9:
    y = J.rand(R, seed=2, stream=0)
    ms = wall(lambda: J.dot_blocks(x, y), 5)
    print(f"dot_blocks(x, y):  {ms:8.3f} ms  {2 * nbytes / ms / 1e9:5.2f} TB/s")
