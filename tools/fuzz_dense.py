#!/usr/bin/env python3
"""Fuzz campaign for block operators with DENSE children: random M x K shapes (tall / wide / grid), uniform children (the batched
kernels) or ragged / adjointed / mixed-with-elementwise ones (the reference's per-block loop), four eltypes, dirty outputs.
Forward: bit-exact vs the CPU oracle while a child stays below 1 MiB and a wide operator has at most 64 children; adjoint: within
1e-6 / 1e-14 of an 80-bit host sum.

    python tools/fuzz_dense.py NCASES [SEED0]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J
from oracle import jets_oracle as oracle
from tests.helpers import DTYPES, assert_bits_equal, u01

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
J.init(0)
# late round 5: FUZZ_LISTS=1 sends every operator that CAN take the batched list route there (small_loop_max_kib = 0) with the list kernels' own lane
# layout (column groups: tolerance parity) -- every forward check is then a tolerance check; the default pins the one-launch loop and columns in order, so
# that the bit-for-bit expectations of rounds 2-4 below still describe what runs
LISTS = os.environ.get("FUZZ_LISTS", "0") == "1"
J.tune(small_loop_max_kib=0 if LISTS else 1 << 40, dense_list_split=1 if LISTS else 0)


def werr(a, b):
    a, b = np.asarray(a, dtype=np.clongdouble).ravel(), np.asarray(b, dtype=np.clongdouble).ravel()
    den = np.linalg.norm(np.abs(b).astype(np.longdouble))
    return float(np.linalg.norm(np.abs(a - b).astype(np.longdouble)) / (den if den else 1.0))


t0 = time.time()
stats = {"batched": 0, "loop": 0}
for case in range(seed0, seed0 + ncases):
    rng = np.random.default_rng(31_000 + case)
    dt = DTYPES[rng.integers(len(DTYPES))]
    shape = rng.random()
    if shape < 0.4:
        nrow, ncol = int(rng.integers(2, 80)), 1
    elif shape < 0.7:
        nrow, ncol = 1, int(rng.integers(2, 80))
    else:
        nrow, ncol = int(rng.integers(2, 7)), int(rng.integers(2, 7))
    flavour = rng.random()                       # 0.6 uniform dense; 0.15 ragged; 0.15 some adjointed (square); 0.1 mixed with elementwise
    pool = [1, 2, 3, 4, 5, 8, 12, 16, 33, 64, 100, 256]
    nr0, nc0 = int(rng.choice(pool)), int(rng.choice(pool))
    if flavour >= 0.75:
        nc0 = nr0                                # adjointed / mixed kinds need square blocks
    row_len = [nr0] * nrow
    col_len = [nc0] * ncol
    if 0.6 <= flavour < 0.75:
        if ncol == 1:
            row_len = [int(rng.choice(pool)) for _ in range(nrow)]
        elif nrow == 1:
            col_len = [int(rng.choice(pool)) for _ in range(ncol)]
        else:
            row_len = [int(rng.choice(pool)) for _ in range(nrow)]
    dev, ora, mats = [], [], {}
    for i in range(nrow):
        dr, orow = [], []
        for j in range(ncol):
            nr, nc = row_len[i], col_len[j]
            kind = "dense"
            if flavour >= 0.9 and rng.random() < 0.4:
                kind = ["diag", "identity", "zero"][rng.integers(3)]
            adj = flavour >= 0.75 and flavour < 0.9 and rng.random() < 0.4
            if kind == "dense":
                hA = np.asfortranarray(u01(oracle, dt, 800 + case, 1000 * i + j, nr * nc).reshape((nr, nc), order="F"))
                mats[(i, j)] = (hA, adj)
                op = J.JopDense(J.from_numpy(hA))
                dr.append(op.H if adj else op)
                orow.append(oracle.Block("dense", nr, nc, coeff=hA, adjoint=adj))
            elif kind == "diag":
                spc = J.JetSpace(dt, nr)
                g = J.rand(spc, seed=801 + case, stream=1000 * i + j)
                dr.append(J.JopDiagonal(g)); orow.append(oracle.Block("diag", nr, coeff=u01(oracle, dt, 801 + case, 1000 * i + j, nr)))
                mats[(i, j)] = (np.diag(u01(oracle, dt, 801 + case, 1000 * i + j, nr)), False)
            elif kind == "identity":
                dr.append(J.JopIdentity(J.JetSpace(dt, nr))); orow.append(oracle.Block("identity", nr))
                mats[(i, j)] = (np.eye(nr, dtype=dt), False)
            else:
                dr.append(J.JopZeroBlock(J.JetSpace(dt, nc), J.JetSpace(dt, nr))); orow.append(oracle.Block("zero", nr, nc))
        dev.append(dr); ora.append(orow)
    A = J.blockop(dev)
    uniform = flavour < 0.6
    stats["batched" if uniform else "loop"] += 1
    tag = f"case {case}: {np.dtype(dt).name} {nrow}x{ncol} rows={row_len[:5]} cols={col_len[:5]} flavour={flavour:.2f}"
    try:
        NR, NC = sum(row_len), sum(col_len)
        m = J.rand(J.domain(A), seed=1, stream=case); hm = u01(oracle, dt, 1, case, NC)
        d = J.rand(J.range(A), seed=2, stream=case); hd = u01(oracle, dt, 2, case, NR)
        offr, offc = np.cumsum([0] + row_len), np.cumsum([0] + col_len)
        J.mul_(d, A, m)
        ref = oracle.block_df(ora, [hd[offr[i]:offr[i + 1]].copy() for i in range(nrow)], [hm[offc[j]:offc[j + 1]].copy() for j in range(ncol)])
        has_adj = any(a for (_, a) in mats.values())
        big = max(row_len) * max(col_len) * np.dtype(dt).itemsize >= (1 << 20)     # a child of 1 MiB or more may have its columns split
        # adjointed dense children / dense next to other kinds, every matrix <= 256 KiB: the one-launch block loop
        # (k_block_loop_small), whose in-thread sequential dots are the oracle's -- forward AND adjoint bit for bit
        # ... as long as an output element needs at most 512 sequential products (round 3; with more, the batched launches + combine launch
        # of dense_mixed_apply take over: forward still bit-exact without adjointed children, every B' x an fp64 wave reduction)
        dense_at = [[o.kind == "dense" for o in r_] for r_ in ora]
        line_work = max([sum(col_len[j] for j in range(ncol) if dense_at[i][j]) for i in range(nrow)] +
                        [sum(row_len[i] for i in range(nrow) if dense_at[i][j]) for j in range(ncol)])
        one_launch = (has_adj or flavour >= 0.9) and not uniform and max(row_len) * max(col_len) * np.dtype(dt).itemsize <= (256 << 10) \
            and any(o.kind == "dense" for r_ in ora for o in r_) and not all(o.kind == "dense" and not o.adjoint for r_ in ora for o in r_) \
            and line_work <= 512
        if LISTS:
            one_launch = False
        if one_launch:
            stats["one_launch"] = stats.get("one_launch", 0) + 1
            assert_bits_equal(d.to_numpy(), np.concatenate(ref), "forward (one-launch loop), " + tag)
        elif LISTS or (nrow == 1 and ncol > 64 and uniform) or has_adj or big:
            single = np.dtype(dt).itemsize // (2 if np.dtype(dt).kind == "c" else 1) == 4
            assert werr(d.to_numpy(), np.concatenate(ref)) < (2e-6 if single else 1e-14), "forward (tolerance), " + tag
        else:
            assert_bits_equal(d.to_numpy(), np.concatenate(ref), "forward, " + tag)
        dd = J.rand(J.range(A), seed=3, stream=case); hdd = u01(oracle, dt, 3, case, NR)
        mt = J.rand(J.domain(A), seed=4, stream=case)
        J.mul_(mt, A.H, dd)
        wide = np.clongdouble
        truth = []
        for j in range(ncol):
            acc = np.zeros(col_len[j], dtype=wide)
            for i in range(nrow):
                if (i, j) in mats:
                    M, adj = mats[(i, j)]
                    Mw = M.astype(wide)
                    blockT = Mw if adj else np.conj(Mw).T          # the block is M' when adjointed, so its adjoint is M
                    acc = acc + blockT @ hdd[offr[i]:offr[i + 1]].astype(wide)
            truth.append(acc)
        if nrow == 1 and not any((0, j) in mats for j in range(ncol)):
            pass
        single = np.dtype(dt).itemsize // (2 if np.dtype(dt).kind == "c" else 1) == 4
        # a single-row operator leaves the domain blocks of zero blocks untouched (1047-1051): compare only where a block wrote
        got = mt.to_numpy()
        keep = np.concatenate([np.full(col_len[j], any((i, j) in mats for i in range(nrow)) or nrow > 1) for j in range(ncol)])
        assert werr(got[keep], np.concatenate(truth)[keep]) < (2e-6 if single else 1e-14), "adjoint, " + tag
        if one_launch:
            hmt = u01(oracle, dt, 4, case, NC)
            refm = oracle.block_df_adj(ora, [hmt[offc[j]:offc[j + 1]].copy() for j in range(ncol)], [hdd[offr[i]:offr[i + 1]].copy() for i in range(nrow)])
            assert_bits_equal(got, np.concatenate(refm), "adjoint (one-launch loop), " + tag)
    except Exception as e:
        print("FAIL", tag)
        print(repr(e)[:1500])
        raise SystemExit(1)
    if (case - seed0 + 1) % 100 == 0:
        print(f"{case - seed0 + 1} cases ok ({stats}), {time.time() - t0:.0f} s", flush=True)
print(f"fuzz_dense: {ncases} cases ok ({stats['batched']} on the batched kernels, {stats['loop']} others, of which {stats.get('one_launch', 0)} on the one-launch loop "
      f"-- forward and adjoint bit-exact), {time.time() - t0:.0f} s")
