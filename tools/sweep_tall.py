#!/usr/bin/env python3
"""Kernel-shape sweep for the tall fast path (interleaved rounds in one process, HIP-event timed).

    python tools/sweep_tall.py --nblocks 1024 --edge 256 --rounds 2 > gpurun_out/sweep.txt
"""
import argparse
import itertools
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nblocks", type=int, default=1024)
    ap.add_argument("--edge", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--which", default="fwd,adj,normal")
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    J.init(0)
    n = args.edge ** 3
    blk = J.JetSpace("float32", args.edge, args.edge, args.edge)
    R = J.JetBSpace([blk] * args.nblocks)
    coeff = J.rand(R, seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    m = J.rand(J.domain(A), seed=2, stream=0)
    d = J.rand(J.range(A), seed=3, stream=0)
    mt = J.zeros(J.domain(A))
    C = A.H @ A
    J.synchronize()
    nbytes = (2 * args.nblocks * n + n) * 4
    nbytes_normal = (args.nblocks * n + 2 * n) * 4

    def timeit(fn):
        fn()
        e0 = J.Event().record()
        for _ in range(args.reps):
            fn()
        e1 = J.Event().record()
        return e0.elapsed_ms(e1) / args.reps

    which = args.which.split(",")
    results = {}
    if args.quick:
        fwd_cfgs = [dict(fwd_wg=w, fwd_unroll=u, fwd_group=g, nt=t) for w in (256, 512) for u in (2, 4) for g in (8, 32) for t in (0, 1)]
        adj_cfgs = [dict(adj_wg=w, adj_unroll=u, adj_depth=dp, nt=t) for w in (256, 512) for u in (1, 2) for dp in (2, 4) for t in (0, 1)]
    else:
        fwd_cfgs = [dict(fwd_wg=w, fwd_unroll=u, fwd_group=g, fwd_order=o, nt=1) for w in (256, 512, 1024) for u in (1, 2, 4, 8) for g in (2, 4, 8, 16, 32, 64) for o in (0, 1)
                    if g <= args.nblocks]
        adj_cfgs = [dict(adj_wg=w, adj_unroll=u, adj_depth=dp, nt=1) for w in (256, 512, 1024) for u in (1, 2, 4) for dp in (1, 2, 4, 8)
                    if not (u == 4 and dp == 8)]
    for rnd in range(args.rounds):
        if "fwd" in which:
            for cfg in fwd_cfgs:
                J.tune(**cfg)
                ms = timeit(lambda: J.mul_(d, A, m))
                results.setdefault(("fwd", json.dumps(cfg, sort_keys=True)), []).append(ms)
        if "adj" in which:
            for cfg in adj_cfgs:
                J.tune(**cfg)
                ms = timeit(lambda: J.mul_(mt, A.H, d))
                results.setdefault(("adj", json.dumps(cfg, sort_keys=True)), []).append(ms)
        if "normal" in which:
            for cfg in adj_cfgs:
                J.tune(**cfg)
                ms = timeit(lambda: J.mul_(mt, C, m))
                results.setdefault(("normal", json.dumps(cfg, sort_keys=True)), []).append(ms)
    rows = []
    for (kind, cfg), ms in results.items():
        b = nbytes_normal if kind == "normal" else nbytes
        rows.append((kind, min(ms), sorted(ms)[len(ms) // 2], b / min(ms) / 1e6, cfg))
    rows.sort(key=lambda r: (r[0], r[1]))
    for kind, mn, med, gbs, cfg in rows:
        print(f"{kind:6s} min {mn:8.3f} ms  med {med:8.3f} ms  {gbs:8.1f} GB/s  {cfg}")


if __name__ == "__main__":
    main()
