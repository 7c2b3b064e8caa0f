#!/usr/bin/env python3
"""The tall adjoint over a grid of (workgroup, vectors per lane, rows in flight) at the row counts a rank owns on 1 / 2 / 4 / 8 GPUs:
how far is the fixed rule (pick_adj_shape) from the best shape?     python tools/sweep_adj_rows.py NROW [EDGE]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J

nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 256
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 256
J.init(0)
n = edge ** 3
spc = J.JetSpace("float32", edge, edge, edge)
coeff = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
d = J.rand(J.range(A), seed=3, stream=0)
mt = J.zeros(spc)
b = (2 * nrow + 1) * n * 4


def timed(reps=5, warm=2):
    for _ in range(warm):
        J.mul_(mt, A.H, d)
    ts = []
    for _ in range(reps):
        e0 = J.Event().record()
        J.mul_(mt, A.H, d)
        e1 = J.Event().record()
        ts.append(e0.elapsed_ms(e1))
    return min(ts)


for rnd in range(2):
    t = timed()
    print(f"{nrow} x {edge}^3 default rule: {t:7.3f} ms {b / t / 1e6:7.1f} GB/s  launches {J.tune_get('last_adj_launches')}", flush=True)
res = []
for wg in (256, 512, 1024):
    for un in (1, 2, 4):
        for dp in (1, 2, 4, 8):
            for rpl in (0, 512) if nrow > 512 else (0,):
                try:
                    J.tune(adj_wg=wg, adj_unroll=un, adj_depth=dp, adj_rows_per_launch=rpl if rpl else nrow)
                    t = timed(reps=4, warm=1)
                except Exception:
                    continue
                res.append((t, wg, un, dp, rpl))
                print(f"  wg {wg:4d} unroll {un} depth {dp} rows/launch {rpl or nrow}: {t:7.3f} ms {b / t / 1e6:7.1f} GB/s", flush=True)
res.sort()
print("best five:", [(f"{t:.3f} ms {b / t / 1e6:.0f} GB/s", wg, un, dp, rpl) for t, wg, un, dp, rpl in res[:5]])
J.tune(adj_wg=0, adj_unroll=0, adj_depth=0, adj_rows_per_launch=0)
t = timed()
print(f"{nrow} x {edge}^3 default rule again: {t:7.3f} ms {b / t / 1e6:7.1f} GB/s")
