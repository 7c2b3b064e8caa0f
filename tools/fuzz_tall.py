#!/usr/bin/env python3
"""Fuzz campaign for TALL operators of elementwise rows (the hot path): random row counts, block lengths (aligned and not), four
eltypes, rows all-diagonal (one slab or separate arrays) or of mixed kinds (zero / identity / scalar / adjointed), dirty outputs.
Every case: forward, adjoint, adjoint in random 16-byte aligned ranges, fused A'A, fused forward update and the one-pass LSQR
step -- plain, with XCD-contiguous tiles, chained, and chained in random ranges with the deferred ||u||^2 -- all BIT-EXACT against
the CPU oracle's loops (src/Jets.jl:1010-1057).  The split-row walk is switched off (adj_split=0): it is tolerance parity by design.

    python tools/fuzz_tall.py NCASES [SEED0]
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J
from jets_jl_amd import jetblock as _blk
from jets_jl_amd._ffi import check, lib
from oracle import jets_oracle as oracle
from tests.helpers import DTYPES, assert_bits_equal, u01

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
J.init(0)
J.tune(adj_split=0)
for kv in os.environ.get("JETS_TUNE", "").split(","):          # e.g. JETS_TUNE=fwd_anchor=1: every off-grid forward on the anchored kernel (round 6)
    if "=" in kv:
        J.tune(**{kv.split("=")[0]: int(kv.split("=")[1])})
KINDS = ["diag", "zero", "identity", "scale", "diag_adj", "scale_adj"]
out = C.c_double(0)
OFFGRID = os.environ.get("FUZZ_OFFGRID", "0") == "1"           # most cases off the 16-byte pack grid (round 5, last session)
t0 = time.time()
stats = {"all_diag": 0, "mixed": 0, "chained": 0, "general": 0}
for case in range(seed0, seed0 + ncases):
    rng = np.random.default_rng(77_000 + case)
    dt = DTYPES[rng.integers(len(DTYPES))]
    cplx = np.dtype(dt).kind == "c"
    nrow = int(rng.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 24, 31, 40, 64, 65]))
    per16 = 16 // np.dtype(dt).itemsize
    if rng.random() < (0.3 if OFFGRID else 0.75):                 # 16-byte multiples: the tall kernels; full chain tiles sometimes
        n = int(rng.choice([1, 3, 16, 64, 100, 256, 512, 1024, 4096])) * per16 * int(rng.choice([1, 1, 4]))
    elif OFFGRID and rng.random() < 0.5:                          # off the 16-byte grid, several tiles (FUZZ_OFFGRID=1)
        n = int(rng.choice([1025, 4097, 4099, 8193, 16385, 20001, 40003, 65535])) + int(rng.integers(0, 3))
    else:
        n = int(rng.integers(1, 3000))                            # anything: the general kernels, or (from one pack per row on) the under-aligned tall kernels
    n = min(n, 1 << 16)
    spc = J.JetSpace(dt, n)
    mixed = rng.random() < 0.5 and nrow > 1
    seed = 500 + case
    dev, ora = [], []
    if not mixed and rng.random() < 0.5:                          # one slab of coefficients (strided addressing)
        coeff = J.rand(J.JetBSpace([spc] * nrow), seed=seed, stream=0)
        dev = [J.JopDiagonal(c) for c in coeff.arrays]
        ora = [oracle.Block("diag", n, coeff=oracle.rng_u01(dt, seed, 0, i * n, n)) for i in range(nrow)]
        kinds = ["diag"] * nrow
    else:
        kinds = [KINDS[k] for k in rng.integers(0, len(KINDS), size=nrow)] if mixed else ["diag"] * nrow
        if mixed:
            kinds[int(rng.integers(nrow))] = "zero"
        for i, k in enumerate(kinds):
            if k == "zero":
                dev.append(J.JopZeroBlock(spc, spc)); ora.append(oracle.Block("zero", n))
            elif k == "identity":
                dev.append(J.JopIdentity(spc)); ora.append(oracle.Block("identity", n))
            elif k.startswith("scale"):
                a = (0.375 + 0.125 * i) - (0.25j * (i + 1) if cplx else 0)
                # scalars of every TYPE class (tests/test_scalar_types.py): numpy 64-bit ones are promoted arithmetic against 32-bit elements
                tk = [lambda v: v, np.complex64 if cplx else np.float32, np.complex128 if cplx else np.float64] + ([lambda v: float(v.real), lambda v: complex(v.real, 0.0)] if cplx else [])
                a = tk[(i + seed) % len(tk)](a)
                op = J.JopLn(dom=spc, rng=spc, df=J.constdiag_df, df_adj=J.constdiag_df_adj, s={"a": a})
                dev.append(op.H if k.endswith("adj") else op); ora.append(oracle.Block("scale", n, scale=a, adjoint=k.endswith("adj")))
                ora[-1].scale_src = a
            else:
                op = J.JopDiagonal(J.rand(spc, seed=seed, stream=i))
                dev.append(op.H if k.endswith("adj") else op)
                ora.append(oracle.Block("diag", n, coeff=u01(oracle, dt, seed, i, n), adjoint=k.endswith("adj")))
    A = J.blockop([[op] for op in dev])
    ops = [[b] for b in ora]
    stats["mixed" if mixed else "all_diag"] += 1
    hm = u01(oracle, dt, 2, case, n)
    hd = [u01(oracle, dt, 3, 50 + i, n) for i in range(nrow)]
    m = J.from_numpy(hm)
    d = J.from_numpy(np.concatenate(hd), J.range(A))
    tag = f"case {case}: {np.dtype(dt).name} {nrow} x {n} {'mixed ' + str(kinds) if mixed else 'diag'}"
    # forward (zero rows keep what was found, 1022) and adjoint into a dirty vector (1042)
    J.mul_(d, A, m)
    want_d = oracle.block_df(ops, [b.copy() for b in hd], [hm])
    assert_bits_equal(d.to_numpy(), np.concatenate(want_d), tag + " forward")
    mt = J.from_numpy(u01(oracle, dt, 9, case, n))
    J.mul_(mt, A.H, d)
    want_m = oracle.block_df_adj(ops, [np.zeros(n, dt)], want_d)[0]
    assert_bits_equal(mt.to_numpy().ravel(order="F"), want_m, tag + " adjoint")
    y = (A.H @ A) * m
    tmp = oracle.block_df(ops, [np.zeros(n, dt) for _ in range(nrow)], [hm])
    assert_bits_equal(y.to_numpy().ravel(order="F"), oracle.block_df_adj(ops, [np.zeros(n, dt)], tmp)[0], tag + " fused A'A")
    nat = _blk._tall_native(A)
    wide = np.dtype(dt).itemsize // (2 if cplx else 1) == 4 and any(isinstance(getattr(b, "scale_src", None), (np.float64, np.complex128)) for b in ora)
    # rows off the 16-byte pack grid (round 5, last session): the whole-vector fused passes take them (under-aligned packs), the ranged ones do not
    off_grid = nat is not None and not wide and n % per16 != 0 and nrow >= 2 and n * np.dtype(dt).itemsize >= 16
    if (nat is None or n % per16 != 0 or wide) and not off_grid:          # (a wide scalar: the per-block loop -- no ranged / fused entry points)
        stats["general"] += 1
        J.close(A)
        continue
    if off_grid:
        stats["off_grid"] = stats.get("off_grid", 0) + 1
    # ranges: random cuts at 16-byte multiples
    ncut = int(rng.integers(1, 4))
    cuts = sorted(set([0, n] + [int(c) * per16 for c in rng.integers(0, n // per16 + 1, size=ncut)]))
    if not off_grid:
        mt2 = J.from_numpy(u01(oracle, dt, 10, case, n))
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            check(lib.jh_blockop_mul_adj_range(nat.handle, mt2.handle, d.handle, lo, hi - lo))
        assert_bits_equal(mt2.to_numpy().ravel(order="F"), want_m, tag + f" ranged adjoint {cuts}")
    alpha, beta = float(rng.choice([1.0, 0.75, -1.25])), float(rng.choice([0.0, -0.5, 1.0, 0.3]))
    hu = [u01(oracle, dt, 11, i, n) for i in range(nrow)]
    # u_i .= alpha .* tmp_i .+ beta .* u_i with REAL alpha, beta (LSQR's are norms): Julia scales a complex number by a real one
    # component by component (no cross terms, so a zero row under a negative alpha is (-0, -0)), each product rounded, then the sum
    S = np.zeros(1, dt).real.dtype.type
    want_u = []
    for t_, u_ in zip(tmp, hu):
        r = S(alpha) * t_.view(S)
        if beta:
            r = r + S(beta) * u_.view(S)
        want_u.append(r.view(dt))
    want_w = oracle.block_df_adj(ops, [np.zeros(n, dt)], want_u)[0]
    nrm = float(sum(np.vdot(b.astype(np.complex128), b.astype(np.complex128)).real for b in want_u))
    u2 = J.from_numpy(np.concatenate(hu), J.range(A))
    check(lib.jh_blockop_mul_axpby(nat.handle, u2.handle, m.handle, alpha, beta, C.byref(out)))
    assert_bits_equal(u2.to_numpy(), np.concatenate(want_u), tag + " fused forward update")
    for mode, chain in ((0, -1), (1, -1), (2, 1)):
        J.tune(step_chain=chain)
        J.op_tune_set(A, "step_mode", mode if chain < 0 else -1)
        u = J.from_numpy(np.concatenate(hu), J.range(A))
        w = J.from_numpy(u01(oracle, dt, 12, case, n))
        check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, m.handle, w.handle, alpha, beta, C.byref(out)))
        stats["chained"] += 1 if J.tune_get("last_step_chain") > 0 else 0
        assert_bits_equal(u.to_numpy(), np.concatenate(want_u), tag + f" step mode {mode}: u ({alpha}, {beta})")
        assert_bits_equal(w.to_numpy().ravel(order="F"), want_w, tag + f" step mode {mode}: w")
        assert abs(out.value - nrm) <= 1e-12 * max(nrm, 1e-300), tag + f" step mode {mode}: ||u||^2"
        if off_grid:
            continue
        u = J.from_numpy(np.concatenate(hu), J.range(A))
        w = J.from_numpy(u01(oracle, dt, 13, case, n))
        check(lib.jh_normsq_reset())
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            check(lib.jh_blockop_bidiag_step_range(nat.handle, u.handle, m.handle, w.handle, alpha, beta, lo, hi - lo, None))
        check(lib.jh_normsq_read(C.byref(out)))
        assert_bits_equal(u.to_numpy(), np.concatenate(want_u), tag + f" ranged step mode {mode} {cuts}: u")
        assert_bits_equal(w.to_numpy().ravel(order="F"), want_w, tag + f" ranged step mode {mode} {cuts}: w")
        assert abs(out.value - nrm) <= 1e-12 * max(nrm, 1e-300), tag + f" ranged step mode {mode}: ||u||^2"
    J.tune(step_chain=-1)
    J.close(A)
    if (case - seed0 + 1) % 100 == 0:
        print(f"{case - seed0 + 1} cases, {time.time() - t0:.0f} s, {stats}", flush=True)
print(f"fuzz_tall: {ncases} cases from seed {seed0}: all bit-exact; {stats}; {time.time() - t0:.0f} s")
