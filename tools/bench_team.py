#!/usr/bin/env python3
"""The headline pair and the LSQR iteration through a SINGLE-PROCESS team (rowpart.Team: one context per member, jh_comm_init_all,
grouped ranged all-reduces) -- SURVEY section 8e's form, next to bench.py's one-process-per-GPU form.

    python tools/bench_team.py [--members M] [--nblocks N] [--edge E] [--steps K] [--lsqr ITERS]

With >= M devices visible every member gets its own GPU (RCCL over xGMI); on a one-GPU box the members are M streams of that GPU
(the grouped sum is then a device kernel), which measures the host-side cost of the team flow, not scaling."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jets_jl_amd as J
from jets_jl_amd import rowpart

ap = argparse.ArgumentParser()
ap.add_argument("--members", type=int, default=2)
ap.add_argument("--nblocks", type=int, default=256)
ap.add_argument("--edge", type=int, default=256)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--lsqr", type=int, default=0)
args = ap.parse_args()

M, n = args.members, args.edge ** 3
ndev = J.device_count()
J.init(0)
if ndev >= M:
    ctxs = []
    for dev in range(M):
        J.init(dev)
        ctxs.append(J.context_current()[0])
    placement = f"{M} device(s), one context each, RCCL"
else:
    ctxs = [J.context_current()[0]] + [J.context_create(0) for _ in range(M - 1)]
    placement = f"{M} contexts of ONE device" if M > 1 else "one context, RCCL team of one"
team = rowpart.Team(ctxs)
spc = J.JetSpace("float32", args.edge, args.edge, args.edge)
parts = [rowpart.partition_rows(args.nblocks, M, k) for k in range(M)]
ops, keep = [], []
for k, _ in team.each():
    coeff = J.rand(J.JetBSpace([spc] * parts[k].count), seed=1, stream=0, index_base=parts[k].first * n)
    keep.append(coeff)
    ops.append(J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays]))
T = team.operator(ops)
m = rowpart.TeamVec([J.rand(spc, seed=2, stream=0) for _ in team.each()])
d = team.zeros(T.ranges())
mt = team.zeros(T.domain())


def pair():
    T.mul_(d, m)
    T.mul_adj_(mt, d)


for _ in range(args.warmup + 16):                        # + the forward's lazy walk trials
    pair()
team.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    pair()
team.synchronize()
dt = (time.perf_counter() - t0) / args.steps
pair_bytes = (4 * args.nblocks * n + 2 * n * M) * 4
out = {"metric": "fwd+adj mul! pairs/s, single-process team", "value": 1.0 / dt, "ms_per_pair": 1e3 * dt, "members": M, "placement": placement,
       "nblocks": args.nblocks, "edge": args.edge, "GBps_algorithmic": pair_bytes / dt / 1e9}
if args.lsqr:
    x_true = rowpart.TeamVec([J.rand(spc, seed=4, stream=0) for _ in team.each()])
    b = team.zeros(T.ranges())
    T.mul_(b, x_true)
    team.synchronize()
    t0 = time.perf_counter()
    res = J.lsqr(T, b, atol=0.0, btol=0.0, conlim=0.0, maxiter=args.lsqr, overwrite_b=True, force_maxiter=True)
    team.synchronize()
    tl = time.perf_counter() - t0
    err = (res.x[0] - x_true[0]).materialize()
    out["lsqr"] = {"iterations": res.itn, "ms_per_iteration": 1e3 * tl / max(res.itn, 1), "rel_err_vs_x_true": float(J.norm(err)) / float(J.norm(x_true[0])),
                   "driver": "jh_lsqr_solve_team"}
print(json.dumps(out))
team.close()
