"""Golden fixtures (tests/golden/jets_block_path_v1.npz, made by tests/golden/make_golden.py).

CPU: the oracle still reproduces every stored output bit for bit (regression pin on the checker).
GPU: the HIP path, fed the stored INPUTS through the C ABI, reproduces the stored outputs bit for
bit (reductions: within the stated tolerance).
"""
import os

import numpy as np
import pytest

from .helpers import assert_bits_equal

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "jets_block_path_v1.npz"))

TALL = [("tall_f32", np.float32), ("tall_f64", np.float64), ("tall_c32", np.complex64), ("tall_c64", np.complex128)]
MIXED = [("mixed_f64", np.float64), ("mixed_c32", np.complex64)]
VEC = [("vec_f32", np.float32), ("vec_c64", np.complex128)]


def _split(flat, lens):
    offs = np.cumsum([0] + list(lens))
    return [np.ascontiguousarray(flat[offs[i]:offs[i + 1]]) for i in range(len(lens))]


# ------------------------------------------------------------------ CPU: oracle vs fixtures
@pytest.mark.parametrize("tag,dt", TALL)
def test_oracle_reproduces_tall_fixtures(oracle, tag, dt):
    a, m, d = G[f"{tag}_a"], G[f"{tag}_m"], G[f"{tag}_d"]
    assert a.dtype == np.dtype(dt)
    nrow, n = a.shape
    ops = [[oracle.Block("diag", n, coeff=np.ascontiguousarray(a[i]))] for i in range(nrow)]
    fwd = oracle.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(nrow)], [m])
    assert_bits_equal(np.stack(fwd), G[f"{tag}_fwd"], f"{tag} forward")
    adj = oracle.block_df_adj(ops, [np.zeros(n, dtype=dt)], [np.ascontiguousarray(d[i]) for i in range(nrow)])
    assert_bits_equal(adj[0], G[f"{tag}_adj"], f"{tag} adjoint")
    assert_bits_equal(oracle.normal_df(ops, [np.zeros(n, dtype=dt)], [m])[0], G[f"{tag}_normal"], f"{tag} normal")


@pytest.mark.parametrize("tag,dt", MIXED)
def test_oracle_reproduces_mixed_fixtures(oracle, tag, dt):
    from .golden.make_golden import mixed_blocks

    coeffs, m, d, d0 = G[f"{tag}_coeffs"], G[f"{tag}_m"], G[f"{tag}_d"], G[f"{tag}_d0"]
    n = m.shape[1]
    ops = mixed_blocks(dt, n, coeffs)
    ms = [np.ascontiguousarray(m[j]) for j in range(4)]
    fwd = oracle.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(3)], ms)
    assert_bits_equal(np.stack(fwd), G[f"{tag}_fwd"], f"{tag} forward")
    dirty = oracle.block_df(ops, [d0[i].copy() for i in range(3)], ms)
    assert_bits_equal(np.stack(dirty), G[f"{tag}_fwd_dirty"], f"{tag} forward into dirty d")
    adj = oracle.block_df_adj(ops, [np.zeros(n, dtype=dt) for _ in range(4)], [np.ascontiguousarray(d[i]) for i in range(3)])
    assert_bits_equal(np.stack(adj), G[f"{tag}_adj"], f"{tag} adjoint")


@pytest.mark.parametrize("tag,dt", VEC)
def test_oracle_reproduces_vector_fixtures(oracle, tag, dt):
    lens = G[f"{tag}_lens"]
    u, v, w = (_split(G[f"{tag}_{k}"], lens) for k in "uvw")
    coef = [c.item() for c in G[f"{tag}_coef"]]     # plain Python numbers: taken in the element type (a numpy float64 would be Julia's Float64: promoted)
    x = oracle.barr_lincomb([np.empty_like(t) for t in u], coef, [u, v, w])
    assert_bits_equal(np.concatenate(x), G[f"{tag}_x"], f"{tag} lincomb")
    for p, want in zip(G[f"{tag}_norm_p"], G[f"{tag}_norms"]):
        assert oracle.barr_norm(u, float(p)) == want
    dv = oracle.barr_dot(u, v)
    assert (np.real(dv), np.imag(dv)) == tuple(G[f"{tag}_dot"])


# ------------------------------------------------------------------ GPU: HIP path vs fixtures
@pytest.mark.gpu
@pytest.mark.parametrize("tag,dt", TALL)
def test_hip_reproduces_tall_fixtures(Jets, tag, dt):
    a, m, d = G[f"{tag}_a"], G[f"{tag}_m"], G[f"{tag}_d"]
    nrow, n = a.shape
    diags = [Jets.from_numpy(a[i]) for i in range(nrow)]
    A = Jets.blockop([[Jets.JopDiagonal(g)] for g in diags])
    dm = Jets.from_numpy(m)
    assert_bits_equal((A * dm).to_numpy(), G[f"{tag}_fwd"].ravel(), f"{tag} forward")
    dd = Jets.from_numpy(d.ravel(), Jets.range(A))
    assert_bits_equal((A.H * dd).to_numpy(), G[f"{tag}_adj"], f"{tag} adjoint")
    assert_bits_equal(((A.H @ A) * dm).to_numpy(), G[f"{tag}_normal"], f"{tag} fused normal")


@pytest.mark.gpu
@pytest.mark.parametrize("tag,dt", MIXED)
def test_hip_reproduces_mixed_fixtures(Jets, tag, dt):
    from .golden.make_golden import MIXED_KINDS

    coeffs, m, d, d0 = G[f"{tag}_coeffs"], G[f"{tag}_m"], G[f"{tag}_d"], G[f"{tag}_d0"]
    n = m.shape[1]
    spc = Jets.JetSpace(dt, n)
    rows = []
    for i, row in enumerate(MIXED_KINDS):
        r = []
        for j, k in enumerate(row):
            if k == "zero":
                r.append(Jets.JopZeroBlock(spc, spc))
            elif k == "identity":
                r.append(Jets.JopIdentity(spc))
            elif k == "scale":
                a = (0.5 + i) - (0.25j * (j + 1) if np.dtype(dt).kind == "c" else 0)
                r.append(Jets.JopLn(dom=spc, rng=spc, df=Jets.constdiag_df, df_adj=Jets.constdiag_df_adj, s={"a": a}))
            else:
                op = Jets.JopDiagonal(Jets.from_numpy(coeffs[i, j]))
                r.append(op.H if k == "diag_adj" else op)
        rows.append(r)
    A = Jets.blockop(rows)
    dm = Jets.from_numpy(m.ravel(), Jets.domain(A))
    assert_bits_equal((A * dm).to_numpy(), G[f"{tag}_fwd"].ravel(), f"{tag} forward")
    dirty = Jets.mul_(Jets.from_numpy(d0.ravel(), Jets.range(A)), A, dm)
    assert_bits_equal(dirty.to_numpy(), G[f"{tag}_fwd_dirty"].ravel(), f"{tag} forward into dirty d")
    dd = Jets.from_numpy(d.ravel(), Jets.range(A))
    assert_bits_equal((A.H * dd).to_numpy(), G[f"{tag}_adj"].ravel(), f"{tag} adjoint")


@pytest.mark.gpu
@pytest.mark.parametrize("tag,dt", VEC)
def test_hip_reproduces_vector_fixtures(Jets, tag, dt):
    lens = [int(k) for k in G[f"{tag}_lens"]]
    R = Jets.JetBSpace([Jets.JetSpace(dt, k) for k in lens])
    u, v, w = (Jets.from_numpy(G[f"{tag}_{k}"], R) for k in "uvw")
    c = [k.item() for k in G[f"{tag}_coef"]]        # (plain Python numbers, as in the oracle's test above)
    x = (c[0] * u + c[1] * v + c[2] * w).materialize()
    assert_bits_equal(x.to_numpy(), G[f"{tag}_x"], f"{tag} lincomb")
    tol = 1e-5 if np.dtype(dt) == np.dtype(np.float32) else 1e-12
    for p, want in zip(G[f"{tag}_norm_p"], G[f"{tag}_norms"]):
        assert float(Jets.norm(u, float(p))) == pytest.approx(float(want), rel=10 * tol)
    dv = complex(Jets.dot(u, v))
    want = complex(*G[f"{tag}_dot"])
    hu, hv = G[f"{tag}_u"].astype(np.complex128), G[f"{tag}_v"].astype(np.complex128)
    scale = float(np.linalg.norm(hu) * np.linalg.norm(hv))          # signed data: the sum cancels, so bound by |u||v|
    assert abs(dv - np.vdot(hu, hv)) <= tol * scale                 # device (fp64 accumulation) vs fp64 host value
    assert abs(dv - want) <= 10 * tol * scale                       # vs the oracle's eltype-precision sequential sum
    if np.dtype(dt).kind != "c":
        assert tuple(float(t) for t in Jets.extrema(u)) == tuple(G[f"{tag}_extrema"])


# ------------------------------------------------------------------ nonlinear path + Golub-Kahan step (jets_nonlinear_v1.npz)
GN = np.load(os.path.join(HERE, "golden", "jets_nonlinear_v1.npz"))
NL = [("nl_f64", np.float64), ("nl_c32", np.complex64)]
GK = [("gk_f32", np.float32), ("gk_c64", np.complex128)]


@pytest.mark.parametrize("tag,dt", NL)
def test_oracle_reproduces_nonlinear_fixtures(oracle, tag, dt):
    from .golden.make_golden_nonlinear import nl_blocks

    coeffs, mo, dm, d0, dd = (GN[f"{tag}_{k}"] for k in ("coeffs", "mo", "dm", "d0", "dd"))
    n = mo.shape[1]
    ops = nl_blocks(dt, n, coeffs, mo)
    f = oracle.block_f(ops, [d0[i].copy() for i in range(2)], [np.ascontiguousarray(mo[j]) for j in range(3)])
    assert_bits_equal(np.stack(f), GN[f"{tag}_f"], f"{tag} f!")
    jv = oracle.block_df(ops, [d0[i].copy() for i in range(2)], [np.ascontiguousarray(dm[j]) for j in range(3)])
    assert_bits_equal(np.stack(jv), GN[f"{tag}_jv"], f"{tag} J dm")
    jt = oracle.block_df_adj(ops, [np.zeros(n, dtype=dt) for _ in range(3)], [np.ascontiguousarray(dd[i]) for i in range(2)])
    assert_bits_equal(np.stack(jt), GN[f"{tag}_jt"], f"{tag} J' dd")


@pytest.mark.gpu
@pytest.mark.parametrize("tag,dt", NL)
def test_hip_reproduces_nonlinear_fixtures(Jets, tag, dt):
    from .golden.make_golden_nonlinear import NL_KINDS

    coeffs, mo, dm, d0, dd = (GN[f"{tag}_{k}"] for k in ("coeffs", "mo", "dm", "d0", "dd"))
    n = mo.shape[1]
    spc = Jets.JetSpace(dt, n)
    rows = []
    for i, row in enumerate(NL_KINDS):
        r = []
        for j, k in enumerate(row):
            r.append({"zero": lambda: Jets.JopZeroBlock(spc, spc), "identity": lambda: Jets.JopIdentity(spc), "square": lambda: Jets.JopSquare(spc),
                      "diag": lambda: Jets.JopDiagonal(Jets.from_numpy(np.ascontiguousarray(coeffs[i, j])))}[k]())
        rows.append(r)
    F = Jets.blockop(rows)
    dmo = Jets.from_numpy(mo.ravel(), Jets.domain(F))
    d = Jets.from_numpy(d0.ravel(), Jets.range(F))
    Jets.mul_(d, F, dmo)
    assert_bits_equal(d.to_numpy(), GN[f"{tag}_f"].ravel(), f"{tag} f!")
    J = Jets.jacobian_(F, dmo)
    d = Jets.from_numpy(d0.ravel(), Jets.range(F))
    Jets.mul_(d, J, Jets.from_numpy(dm.ravel(), Jets.domain(F)))
    assert_bits_equal(d.to_numpy(), GN[f"{tag}_jv"].ravel(), f"{tag} J dm")
    mt = Jets.mul(J.H, Jets.from_numpy(dd.ravel(), Jets.range(F)))
    assert_bits_equal(mt.to_numpy(), GN[f"{tag}_jt"].ravel(), f"{tag} J' dd")


@pytest.mark.parametrize("tag,dt", GK)
def test_oracle_reproduces_golub_kahan_fixtures(oracle, tag, dt):
    a, v, u = GN[f"{tag}_a"], GN[f"{tag}_v"], GN[f"{tag}_u"]
    alpha, beta = (float(x) for x in GN[f"{tag}_alpha_beta"])   # plain Python numbers: taken in the element type
    nrow, n = a.shape
    ops = [[oracle.Block("diag", n, coeff=np.ascontiguousarray(a[i]))] for i in range(nrow)]
    tmp = oracle.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(nrow)], [v])
    unew = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [alpha, beta], [tmp, [np.ascontiguousarray(u[i]) for i in range(nrow)]])
    assert_bits_equal(np.stack(unew), GN[f"{tag}_unew"], f"{tag} u")
    assert_bits_equal(oracle.block_df_adj(ops, [np.zeros(n, dtype=dt)], unew)[0], GN[f"{tag}_w"], f"{tag} w")


@pytest.mark.gpu
@pytest.mark.parametrize("tag,dt", GK)
def test_hip_reproduces_golub_kahan_fixtures(Jets, tag, dt):
    import ctypes as C

    from jets_jl_amd._ffi import lib, check
    from jets_jl_amd import jetblock

    a, v, u = GN[f"{tag}_a"], GN[f"{tag}_v"], GN[f"{tag}_u"]
    alpha, beta = (float(x) for x in GN[f"{tag}_alpha_beta"])
    nrow, n = a.shape
    A = Jets.blockop([[Jets.JopDiagonal(Jets.from_numpy(np.ascontiguousarray(a[i])))] for i in range(nrow)])
    nat = jetblock._native_op(A.jet.s["_native"], A.jet.s["ops"], A.jet.rng.eltype())
    du, dv, dw = Jets.from_numpy(u.ravel(), Jets.range(A)), Jets.from_numpy(v), Jets.zeros(Jets.domain(A))
    out = C.c_double(0)
    check(lib.jh_blockop_bidiag_step(nat.handle, du.handle, dv.handle, dw.handle, alpha, beta, C.byref(out)))
    assert_bits_equal(du.to_numpy(), GN[f"{tag}_unew"].ravel(), f"{tag} u")
    assert_bits_equal(dw.to_numpy(), GN[f"{tag}_w"], f"{tag} w")
    truth = float(np.sum(np.abs(GN[f"{tag}_unew"].astype(np.complex128)) ** 2))
    assert out.value == pytest.approx(truth, rel=1e-6)


# ------------------------------------------------------------------ dense children (tall / ragged / wide / grid)
GD = np.load(os.path.join(HERE, "golden", "jets_dense_blocks_v1.npz"))


def _dense_case(tag):
    from .golden.make_golden_dense import CASES, split

    dt, rows, cols = CASES[tag]
    flat, mats, k = GD[f"{tag}_A"], [], 0
    for i in range(len(rows)):
        row = []
        for j in range(len(cols)):
            n = rows[i] * cols[j]
            row.append(np.asfortranarray(flat[k:k + n].reshape((rows[i], cols[j]), order="F")))
            k += n
        mats.append(row)
    return dt, rows, cols, mats, split


DENSE_TAGS = ["tall_f32", "tall_c64", "ragged_f64", "wide_f64", "wide_c32", "grid_f32"]


@pytest.mark.parametrize("tag", DENSE_TAGS)
def test_oracle_reproduces_dense_fixtures(oracle, tag):
    dt, rows, cols, mats, split = _dense_case(tag)
    ops = [[oracle.Block("dense", rows[i], cols[j], coeff=mats[i][j]) for j in range(len(cols))] for i in range(len(rows))]
    fwd = oracle.block_df(ops, split(GD[f"{tag}_d0"].copy(), rows), split(GD[f"{tag}_m"], cols))
    assert_bits_equal(np.concatenate(fwd), GD[f"{tag}_fwd_dirty"], f"{tag} forward into dirty d")
    adj = oracle.block_df_adj(ops, [np.zeros(c, dtype=dt) for c in cols], split(GD[f"{tag}_d"], rows))
    assert_bits_equal(np.concatenate(adj), GD[f"{tag}_adj"], f"{tag} adjoint")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", DENSE_TAGS)
def test_hip_path_reproduces_dense_fixtures(Jets, tag):
    """Batched GEMV kernels (jh_dense.hip) fed the stored matrices and vectors: forward bit for bit, adjoint within 1e-6 / 1e-14."""
    dt, rows, cols, mats, split = _dense_case(tag)
    A = Jets.blockop([[Jets.JopDense(Jets.from_numpy(mats[i][j])) for j in range(len(cols))] for i in range(len(rows))])
    m = Jets.from_numpy(GD[f"{tag}_m"]) if len(cols) == 1 else Jets.zeros(Jets.domain(A))
    if len(cols) > 1:
        Jets.upload_from(m, GD[f"{tag}_m"])
    d = Jets.zeros(Jets.range(A))
    Jets.upload_from(d, GD[f"{tag}_d0"])
    Jets.mul_(d, A, m)
    want = GD[f"{tag}_fwd_dirty"]
    if len(cols) == 1:                                   # one block column overwrites (1026); the fixture's dirty d is then irrelevant
        assert_bits_equal(d.to_numpy(), want, f"{tag} forward")
    else:
        assert_bits_equal(d.to_numpy(), want, f"{tag} forward into dirty d")
    dd = Jets.zeros(Jets.range(A))
    Jets.upload_from(dd, GD[f"{tag}_d"])
    mt = Jets.mul(A.H, dd)
    got, ref = mt.to_numpy().ravel(order="F").astype(np.complex128), GD[f"{tag}_adj"].astype(np.complex128)
    tol = 1e-6 if np.dtype(dt) in (np.dtype(np.float32), np.dtype(np.complex64)) else 1e-14
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < tol
