#!/usr/bin/env python3
"""Fuzz campaign for fused composite chains and sums of chains (round 6; jets.jl_amd/chains.py, jh_tall_chain.hip): random tall operators (row
counts, block lengths on and off the 16-byte grid, four element types, rows all-diagonal or of mixed kinds), random chains of elementwise stages on
either side of A and A' (scalars of the elements' precision or Float64, diagonals and their adjoints on the domain, weight vectors / block-diagonal
block operators on the range, identities, an opaque closure that splits the chain), their adjoints, and sums of two to four such chains with random
signs.  Every case: the fused result BIT-EXACT against the same composite applied stage by stage on the device (chains.ENABLED = False: the path of
rounds 1-5, itself pinned against the CPU oracle by the test suite), and against the oracle's stages (tests/test_gpu_chains.py: Rig.ora_apply) for the
plain chains.  The split-row walk is switched off (adj_split = 0): it is tolerance parity by design.

    python tools/fuzz_chains.py NCASES [SEED0]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J
from jets_jl_amd import chains
from oracle import jets_oracle as oracle
from tests.helpers import DTYPES, assert_bits_equal, u01
from tests.test_gpu_chains import Rig

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
J.init(0)
J.tune(adj_split=0)
t0 = time.time()
stats = {"chains": 0, "fused_runs": 0, "sums": 0, "sum_terms_fused": 0, "declined": 0}


def dom_stage(rng):
    r = rng.random()
    if r < 0.35:
        return ("M", int(rng.integers(2)), bool(rng.integers(2)))
    if r < 0.7:
        return ("s", float(rng.choice([0.5, -1.25, 3.0, 0.375, 2.0 ** -20, 1e3])), "d")
    if r < 0.9:
        return ("I",)
    return ("opaque",)


def rng_stage(rng, wb):
    r = rng.random()
    if r < 0.45:
        return ("W", int(rng.integers(2)), bool(rng.integers(2)))
    if r < 0.6 and wb:
        return ("Wb", int(rng.integers(2)), bool(rng.integers(2)))
    return ("s", float(rng.choice([0.5, -1.25, 3.0, 0.375, 7.0])), "r")


def random_chain(rng, wb, allow_opaque=True):
    """Tokens in application order and the side the chain starts on ('d' domain / 'r' range)."""
    shape = rng.integers(4)                                   # 0: dom -> rng (.. A ..), 1: rng -> dom (.. A' ..), 2: dom -> dom (.. A .. A' ..), 3: two normal blocks
    toks = []

    def doms(k):
        for _ in range(k):
            t = dom_stage(rng)
            if t == ("opaque",) and not allow_opaque:
                t = ("I",)
            toks.append(t)

    def rngs(k):
        for _ in range(k):
            toks.append(rng_stage(rng, wb))

    if shape == 0:
        doms(int(rng.integers(0, 4))); toks.append("A"); rngs(int(rng.integers(0, 4)))
        return toks, "d"
    if shape == 1:
        rngs(int(rng.integers(0, 4))); toks.append("At"); doms(int(rng.integers(0, 4)))
        return toks, "r"
    doms(int(rng.integers(0, 3))); toks.append("A"); rngs(int(rng.integers(0, 6))); toks.append("At"); doms(int(rng.integers(0, 3)))
    if shape == 3:
        toks.append("A"); rngs(int(rng.integers(0, 2))); toks.append("At")
    return toks, "d"


for case in range(seed0, seed0 + ncases):
    rng = np.random.default_rng(91_000 + case)
    dt = DTYPES[rng.integers(len(DTYPES))]
    nrow = int(rng.choice([2, 3, 5, 8, 9, 17, 33]))
    per16 = 16 // np.dtype(dt).itemsize
    if rng.random() < 0.5:
        n = int(rng.choice([1, 4, 16, 64, 256, 1024, 2048])) * per16 * int(rng.choice([1, 1, 4]))
    else:
        n = int(rng.integers(per16, 6000))
    wb = nrow <= 9
    rig = Rig(J, oracle, dt, nrow, n, "mixed" if rng.random() < 0.5 else "diag", seed=1000 + case, with_wb=wb)
    R, D = J.range(rig.A), J.domain(rig.A)

    def both(C, x, out_space):
        before = chains.STATS["chain_calls"], chains.STATS["sum_terms_fused"]
        y1 = J.mul_(J.rand(out_space, seed=77, stream=1), C, x)
        ran = chains.STATS["chain_calls"] - before[0], chains.STATS["sum_terms_fused"] - before[1]
        chains.ENABLED[0] = False
        try:
            y0 = J.mul_(J.rand(out_space, seed=77, stream=1), C, x)        # (the same dirty output: a bare operator leaves the rows of its zero blocks as found, 1022)
        finally:
            chains.ENABLED[0] = True
        return y1.to_numpy().ravel(order="F"), y0.to_numpy().ravel(order="F"), ran

    # ---- a plain chain and its adjoint
    toks, side = random_chain(rng, wb)
    C = rig.compose(toks)
    hx = [u01(oracle, dt, 91, i, n) for i in range(nrow if side == "r" else 1)]
    x = J.from_numpy(np.concatenate(hx), R if side == "r" else D)
    y1, y0, ran = both(C, x, J.range(C))
    assert_bits_equal(y1, y0, f"case {case} {np.dtype(dt).name} {nrow} x {n} {toks}: fused vs stage by stage")
    if len(toks) > 1:                                        # (a bare operator is no composite: mul! leaves the rows of its zero blocks as found, 1022)
        assert_bits_equal(y1, np.concatenate(rig.ora_apply(toks, hx)), f"case {case} {toks}: fused vs the oracle's stages")
    stats["chains"] += 1
    stats["fused_runs"] += ran[0]
    stats["declined"] += ran[0] == 0
    hz = [u01(oracle, dt, 92, i, n) for i in range(y1.size // n)]
    z = J.from_numpy(np.concatenate(hz), J.range(C))
    a1, a0, ran = both(C.H, z, J.domain(C))
    assert_bits_equal(a1, a0, f"case {case} ({toks})': fused vs stage by stage")
    stats["fused_runs"] += ran[0]
    # ---- a sum of chains that share their spaces
    if rng.random() < 0.6:
        shape = int(rng.integers(3))
        terms = []
        for _ in range(int(rng.integers(2, 5))):
            if shape == 2 and rng.random() < 0.3:                   # a regularisation term: a * I, I, or the bare A'A (`A'A + lam I`: fused since late round 6)
                pick = rng.integers(3)
                terms.append(rig.compose([("s", float(rng.choice([0.25, -1.5, 3.0])), "d"), ("I",)] if pick == 0 else ([("I",)] if pick == 1 else ["A", "At"])))
                continue
            while True:
                t, s = random_chain(rng, wb, allow_opaque=False)
                kind = (s, "r" if (t.count("A") > t.count("At")) else "d")
                want = [("d", "r"), ("r", "d"), ("d", "d")][shape]
                if kind == want:
                    break
            terms.append(rig.compose(t))
        S = terms[0]
        for t in terms[1:]:
            S = (S + t) if rng.random() < 0.5 else (S - t)
        xin = R if shape == 1 else D
        hx = np.concatenate([u01(oracle, dt, 93, i, n) for i in range(nrow if shape == 1 else 1)])
        y1, y0, ran = both(S, J.from_numpy(hx, xin), J.range(S))
        assert_bits_equal(y1, y0, f"case {case} sum of {len(terms)} chains (shape {shape}): fused vs the reference's loop")
        hz = np.concatenate([u01(oracle, dt, 94, i, n) for i in range(y1.size // n)])
        a1, a0, ran2 = both(S.H, J.from_numpy(hz, J.range(S)), J.domain(S))
        assert_bits_equal(a1, a0, f"case {case} adjoint of a sum of {len(terms)} chains (shape {shape})")
        stats["sums"] += 1
        stats["sum_terms_fused"] += ran[1] + ran2[1]
    rig.close()
    if (case - seed0 + 1) % 100 == 0:
        print(f"{case - seed0 + 1} cases, {time.time() - t0:.0f} s, {stats}", flush=True)
print(f"fuzz_chains: {ncases} cases from seed {seed0}: all bit-exact; {stats}; {time.time() - t0:.0f} s")
