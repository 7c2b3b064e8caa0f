#!/usr/bin/env python3
"""The three chain kernels (jh_tall_chain.hip) on one weighted tall operator, for tools/prof_any.sh: prints the ALGO lines its summary needs.

    TAG=chains REGEX='k_chain' bash tools/prof_any.sh tools/prof_chains.py 256 256"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J
from jets_jl_amd import chains

nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 256
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
J.init(0)
blk = J.JetSpace(np.float32, edge, edge, edge)
n, s = blk.length(), 4
R = J.JetBSpace([blk] * nrow)
A = J.blockop([[J.JopDiagonal(c)] for c in J.rand(R, seed=1, stream=0).arrays])
W = J.JopDiagonal(J.rand(R, seed=5, stream=0))
m, y, d = J.rand(blk, seed=2, stream=0), J.zeros(blk), J.zeros(R)
Nn = nrow * n * s
cases = [("A' o W o A  (NORMAL)", J.compose(J.compose(A.H, W), A), y, m, 2 * Nn + 2 * n * s, r"k_chain_adj<float,\s1,\s4,\s\d,\s\d,\s(true|false),\s1,"),
         ("(W o A)'    (ADJOINT)", J.compose(W, A).H, y, d, 3 * Nn + n * s, r"k_chain_adj<float,\s1,\s4,\s\d,\s\d,\s(true|false),\s0,"),
         ("W o A       (FORWARD)", J.compose(W, A), d, m, 3 * Nn + n * s, r"k_chain_fwd<float")]
for tag, op, out, x, nbytes, rx in cases:
    print(f"ALGO {rx} {nbytes}")
for tag, op, out, x, nbytes, rx in cases:
    before = chains.STATS["chain_calls"]
    J.mul_(out, op, x)
    J.synchronize()
    e0 = J.Event().record()
    for _ in range(reps):
        J.mul_(out, op, x)
    e1 = J.Event().record()
    ms = e0.elapsed_ms(e1) / reps
    assert chains.STATS["chain_calls"] == before + reps + 1
    print(f"{tag:24s} {nrow} x {edge}^3 Float32: {ms:8.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s over {nbytes} algorithmic bytes", flush=True)
