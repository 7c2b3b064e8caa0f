"""GPU: tall operators whose rows are NOT all plain diagonals -- zero blocks, identity / scalar rows, adjointed (conjugated)
diagonals -- on the tall kernels' tiling with a per-row kind (jh_tall.hip / jh_tall_step.hip: MIXED instantiations), incl. the fused A'A, the
fused forward update, the ranged adjoint and the one-pass LSQR step.  Bit-exact against the CPU oracle's loops
(src/Jets.jl:1010-1057: zero rows skipped 1022 / 1047), every eltype, aligned block lengths (others take the general kernels)."""
import ctypes as C

import numpy as np
import pytest

from .helpers import DTYPES, assert_bits_equal, u01

pytestmark = pytest.mark.gpu
KINDS = ["diag", "zero", "identity", "scale", "diag_adj", "scale_adj"]


def _build(J, oracle, dt, kinds, n, seed):
    spc = J.JetSpace(dt, n)
    dev, ora = [], []
    cplx = np.dtype(dt).kind == "c"
    for i, k in enumerate(kinds):
        if k == "zero":
            dev.append(J.JopZeroBlock(spc, spc)); ora.append(oracle.Block("zero", n))
        elif k == "identity":
            dev.append(J.JopIdentity(spc)); ora.append(oracle.Block("identity", n))
        elif k.startswith("scale"):
            a = (0.375 + i) - (0.25j * (i + 1) if cplx else 0)
            op = J.JopLn(dom=spc, rng=spc, df=J.constdiag_df, df_adj=J.constdiag_df_adj, s={"a": a})
            dev.append(op.H if k.endswith("adj") else op); ora.append(oracle.Block("scale", n, scale=a, adjoint=k.endswith("adj")))
        else:
            op = J.JopDiagonal(J.rand(spc, seed=seed, stream=i))
            dev.append(op.H if k.endswith("adj") else op)
            ora.append(oracle.Block("diag", n, coeff=u01(oracle, dt, seed, i, n), adjoint=k.endswith("adj")))
    return J.blockop([[op] for op in dev]), [[b] for b in ora]


def _native(J, A):
    from jets_jl_amd import jetblock as _blk

    return _blk._tall_native(A)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", range(6))
def test_mixed_tall_rows_bit_exact(Jets, oracle, dt, case):
    J = Jets
    rs = np.random.RandomState(100 + case)
    nrow = [2, 3, 9, 17, 40, 70][case]
    n = [64, 4096, 260, 1024, 128, 32][case]
    kinds = [KINDS[k] for k in rs.randint(0, len(KINDS), size=nrow)]
    kinds[rs.randint(nrow)] = "zero"                                           # never all-diagonal: the mixed path, not the fast one
    A, ops = _build(J, oracle, dt, kinds, n, seed=40 + case)
    hm = u01(oracle, dt, 2, case, n)
    hd_found = [u01(oracle, dt, 3, 50 + i, n) for i in range(nrow)]
    m = J.from_numpy(hm)
    d = J.from_numpy(np.concatenate(hd_found), J.range(A))
    J.mul_(d, A, m)
    want = oracle.block_df(ops, [b.copy() for b in hd_found], [hm])
    assert_bits_equal(d.to_numpy(), np.concatenate(want), f"forward {kinds}")
    assert J.tune_get("last_fwd_rows_per_wg") > 0
    mt = J.from_numpy(u01(oracle, dt, 9, case, n))                             # dirty: zeroed first (1042)
    J.mul_(mt, A.H, d)
    want_m = oracle.block_df_adj(ops, [u01(oracle, dt, 9, case, n)], want)[0]
    assert_bits_equal(mt.to_numpy().ravel(order="F"), want_m, f"adjoint {kinds}")
    # fused A'A == the unfused chain through a zeros() temporary (530-534)
    y = (A.H @ A) * m
    tmp = oracle.block_df(ops, [np.zeros(n, dt) for _ in range(nrow)], [hm])
    want_y = oracle.block_df_adj(ops, [np.zeros(n, dt)], tmp)[0]
    assert_bits_equal(y.to_numpy().ravel(order="F"), want_y, f"fused A'A {kinds}")
    # ranged adjoint (multi-GPU pipelining): two 16-byte aligned halves give the whole
    from jets_jl_amd._ffi import lib, check
    nat = _native(J, A)
    assert nat is not None
    mt2 = J.zeros(J.domain(A))
    half = (n // 2) // 4 * 4
    check(lib.jh_blockop_mul_adj_range(nat.handle, mt2.handle, d.handle, 0, half))
    check(lib.jh_blockop_mul_adj_range(nat.handle, mt2.handle, d.handle, half, n - half))
    assert_bits_equal(mt2.to_numpy().ravel(order="F"), want_m, "ranged adjoint")
    # one-pass step: u <- alpha (A v) + beta u ; w <- A'u ; ||u||^2  ==  the two-call sequence
    alpha, beta = 0.75, -0.5
    hu = [u01(oracle, dt, 11, i, n) for i in range(nrow)]
    u = J.from_numpy(np.concatenate(hu), J.range(A))
    w = J.zeros(J.domain(A))
    out = C.c_double(0)
    check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, m.handle, w.handle, alpha, beta, C.byref(out)))
    av = oracle.block_df(ops, [np.zeros(n, dt) for _ in range(nrow)], [hm])
    want_u = oracle.barr_lincomb([np.empty(n, dt) for _ in range(nrow)], [alpha, beta], [av, hu])
    assert_bits_equal(u.to_numpy(), np.concatenate(want_u), "one-pass step: u")
    want_w = oracle.block_df_adj(ops, [np.zeros(n, dt)], want_u)[0]
    assert_bits_equal(w.to_numpy().ravel(order="F"), want_w, "one-pass step: w")
    nrm = float(sum(np.vdot(b.astype(np.complex128), b.astype(np.complex128)).real for b in want_u))
    assert abs(out.value - nrm) <= 1e-12 * max(nrm, 1e-300)
    # fused forward update (jh_blockop_mul_axpby) the same way
    u2 = J.from_numpy(np.concatenate(hu), J.range(A))
    check(lib.jh_blockop_mul_axpby(nat.handle, u2.handle, m.handle, alpha, beta, C.byref(out)))
    assert_bits_equal(u2.to_numpy(), np.concatenate(want_u), "fused forward update")


def test_lsqr_on_data_rows_plus_regularisation_and_a_muted_row(Jets, oracle):
    """[A_1; ...; A_6; 0; lambda*I]: the native one-pass LSQR loop runs on the mixed operator (no densifying, no general kernels)."""
    J = Jets
    dt, n = np.float32, 8192
    kinds = ["diag"] * 6 + ["zero", "scale"]
    A, ops = _build(J, oracle, dt, kinds, n, seed=77)
    x_true = J.rand(J.domain(A), seed=4, stream=0)
    b = A * x_true
    res = J.lsqr(A, b, atol=0.0, btol=0.0, conlim=0.0, maxiter=25)
    hx = u01(oracle, dt, 4, 0, n)
    assert np.linalg.norm(res.x.to_numpy().ravel() - hx) <= 1e-4 * np.linalg.norm(hx)
    # same recurrences on the CPU (fp64) for the same operator
    from oracle.lsqr_ref import lsqr_fp64
    a64 = [None if b_[0].kind == "zero" else (np.full(n, b_[0].scale.real) if b_[0].kind == "scale" else b_[0].coeff.astype(np.float64)) for b_ in ops]
    mv = lambda v: np.concatenate([np.zeros(n) if a is None else a * v for a in a64])
    rmv = lambda u: sum(a * u[i * n:(i + 1) * n] for i, a in enumerate(a64) if a is not None)
    xr, _ = lsqr_fp64(mv, rmv, b.to_numpy().astype(np.float64), n, atol=0.0, btol=0.0, conlim=0.0, maxiter=25)
    assert np.linalg.norm(res.x.to_numpy().ravel() - xr) <= 1e-4 * np.linalg.norm(xr)


@pytest.mark.parametrize("dt", [np.float32, np.complex128])
@pytest.mark.parametrize("nterms", [2, 4, 5, 9, 13, 17])
def test_long_jetsum_keeps_the_unfused_rounding_sequence(Jets, oracle, dt, nterms):
    """JetSum of MORE than four tall diagonal operators (src/Jets.jl:628-655): the fused kernels take the terms four at a time,
    later launches continuing the left-to-right sum from what the output holds -- bit-identical to the unfused chain
    d .= 0; d = d +- s_t (A_t m), whatever the grouping; same for the adjoint."""
    J = Jets
    nrow, n = 6, 512
    spc = J.JetSpace(dt, n)
    ops, hcoef = [], []
    for t in range(nterms):
        ops.append(J.blockop([[J.JopDiagonal(J.rand(spc, seed=90 + t, stream=i))] for i in range(nrow)]))
        hcoef.append([u01(oracle, dt, 90 + t, i, n) for i in range(nrow)])
    scales = [1.0 if t % 3 else 0.5 + t for t in range(nterms)]
    signs = [1.0 if t % 2 == 0 or t == 0 else -1.0 for t in range(nterms)]
    S = scales[0] * ops[0] if scales[0] != 1.0 else ops[0]
    for t in range(1, nterms):
        term = scales[t] * ops[t] if scales[t] != 1.0 else ops[t]
        S = S + term if signs[t] > 0 else S - term
    hm = u01(oracle, dt, 2, 0, n)
    m = J.from_numpy(hm)
    d = S * m
    want = [np.zeros(n, dt) for _ in range(nrow)]
    for t in range(nterms):
        ora = [[oracle.Block("diag", n, coeff=c)] for c in hcoef[t]]
        tmp = oracle.block_df(ora, [np.zeros(n, dt) for _ in range(nrow)], [hm])
        if scales[t] != 1.0:
            tmp = oracle.barr_lincomb([np.empty(n, dt) for _ in range(nrow)], [scales[t]], [tmp])      # (s*A) m = s * (A m)
        want = oracle.barr_lincomb([np.empty(n, dt) for _ in range(nrow)], [1.0, signs[t]], [want, tmp])
    assert_bits_equal(d.to_numpy(), np.concatenate(want), f"{nterms}-term sum, forward")
    hd = [u01(oracle, dt, 3, i, n) for i in range(nrow)]
    mt = S.H * J.from_numpy(np.concatenate(hd), J.range(ops[0]))
    want_m = [np.zeros(n, dt)]
    for t in range(nterms):
        ora = [[oracle.Block("diag", n, coeff=c)] for c in hcoef[t]]
        din = hd if scales[t] == 1.0 else oracle.barr_lincomb([np.empty(n, dt) for _ in range(nrow)], [np.conj(scales[t])], [hd])   # (s*A)' d = A'(conj(s) d)
        tmp = oracle.block_df_adj(ora, [np.zeros(n, dt)], din)
        want_m = oracle.barr_lincomb([np.empty(n, dt)], [1.0, signs[t]], [want_m, tmp])
    assert_bits_equal(mt.to_numpy().ravel(order="F"), want_m[0], f"{nterms}-term sum, adjoint")
