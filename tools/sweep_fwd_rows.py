#!/usr/bin/env python3
"""The tall forward per candidate walk (jh_tall.hip: k_fwd_candidates) at the row counts a rank owns on 1 / 2 / 4 / 8 GPUs, plus a
grid of knob shapes around them: which (workgroup, vectors per lane, rows per workgroup, order) is fastest at 128 / 256 / 512 rows?

    python tools/sweep_fwd_rows.py NROW [EDGE] [--grid]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J

nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 256
edge = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 256
grid = "--grid" in sys.argv
J.init(0)
n = edge ** 3
spc = J.JetSpace("float32", edge, edge, edge)
coeff = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
m = J.rand(spc, seed=2, stream=0)
d = J.zeros(J.range(A))
b = (2 * nrow + 1) * n * 4


def timed(reps=6, warm=2):
    for _ in range(warm):
        J.mul_(d, A, m)
    ts = []
    for _ in range(reps):
        e0 = J.Event().record()
        J.mul_(d, A, m)
        e1 = J.Event().record()
        ts.append(e0.elapsed_ms(e1))
    ts.sort()
    return ts[0], ts[len(ts) // 2]


for rnd in range(2):
    for walk in range(10):
        J.op_tune_set(A, "fwd_walk", walk)
        lo, med = timed()
        print(f"{nrow} x {edge}^3 candidate {walk}: min {lo:7.3f} ms {b / lo / 1e6:7.1f} GB/s  median {med:7.3f} ms  rows/wg {J.tune_get('last_fwd_rows_per_wg')}", flush=True)
if grid:
    J.op_tune_set(A, "fwd_walk", 0)
    best = []
    for wg in (256, 512, 1024):
        for un in (1, 2, 4, 8):
            for grp in (1, 2, 4, 8, 16, 32, 1 << 20):
                for order in (0, 1):
                    try:
                        J.tune(fwd_wg=wg, fwd_unroll=un, fwd_group=grp, fwd_order=order)
                        lo, med = timed(reps=4, warm=1)
                    except Exception as e:  # a shape that is not instantiated
                        continue
                    best.append((lo, wg, un, grp, order))
                    print(f"  wg {wg:4d} unroll {un} group {grp:7d} order {order}: min {lo:7.3f} ms {b / lo / 1e6:7.1f} GB/s", flush=True)
    best.sort()
    print("best five:", [(f"{t:.3f} ms {b / t / 1e6:.0f} GB/s", wg, un, grp, order) for t, wg, un, grp, order in best[:5]])
