"""GPU: CGLS (SURVEY.md 8 f-1 "LSQR/CGLS").  Like LSQR it has no counterpart inside Jets.jl (src/Jets.jl:1143-1152 points at the
un-vendored IterativeSolvers.jl), so parity is pinned on the published recurrence: the fp64 CPU CGLS of oracle/cgls_ref.py -- the
TEXTBOOK form with its q = A p vector, while jh_cgls_solve runs two passes per iteration and no q (||A p||^2 = <p, A'A p>; r update,
||r||^2 and A'r in one pass of the step kernel).  Bar after a fixed number of iterations on Float32 data: iterate within 1e-4 (rel l2),
||r|| history within 1e-4; Float64: 1e-10."""
import numpy as np
import pytest

from oracle.cgls_ref import cgls_fp64

from .helpers import make_tall_diag, u01

pytestmark = pytest.mark.gpu


def _host_ops(a_blocks, dt64):
    a64 = [g.astype(dt64) for g in a_blocks]
    n = a64[0].size
    return (lambda x: np.concatenate([g * x for g in a64])), (lambda y: sum(np.conj(g) * y[i * n:(i + 1) * n] for i, g in enumerate(a64)))


@pytest.mark.parametrize("dt,xtol", [(np.float32, 1e-4), (np.float64, 1e-10), (np.complex64, 1e-4), (np.complex128, 1e-10)])
@pytest.mark.parametrize("native", ["1", "0"])
def test_cgls_matches_fp64_cpu_cgls(Jets, oracle, dt, xtol, native, monkeypatch):
    """native = 1: jh_cgls_solve (two passes, no q); 0: the textbook loop of cgls_core over the two fused halves."""
    monkeypatch.setenv("JETS_CGLS_NATIVE", native)
    nrow, shape, iters = 6, (16, 16, 16), 12
    A, _, _, diags = make_tall_diag(Jets, oracle, dt, nrow, shape)
    n = int(np.prod(shape))
    dt64 = np.complex128 if np.dtype(dt).kind == "c" else np.float64
    matvec, rmatvec = _host_ops(diags, dt64)
    hb = (u01(oracle, dt, 51, 0, nrow * n) - dt(0.5)).astype(dt)                    # inconsistent right-hand side
    b = Jets.from_numpy(hb, Jets.range(A))
    res = Jets.cgls(A, b, atol=0.0, btol=0.0, maxiter=iters)
    xr, info = cgls_fp64(matvec, rmatvec, hb.astype(dt64), n, atol=0.0, btol=0.0, maxiter=iters)
    assert res.itn == iters == info["itn"] and res.istop == 7
    x = res.x.to_numpy().ravel(order="F").astype(dt64)
    assert np.linalg.norm(x - xr) / np.linalg.norm(xr) < xtol
    for (i1, r1, ar1), (i2, r2, ar2) in zip(res.history, info["history"]):
        assert i1 == i2 and r1 == pytest.approx(r2, rel=max(xtol, 1e-9)) and ar1 == pytest.approx(ar2, rel=max(50 * xtol, 1e-7))
    assert np.array_equal(b.to_numpy(), hb)                                          # b untouched (overwrite_b=False)
    assert res.r1norm == pytest.approx(info["r1norm"], rel=max(xtol, 1e-9)) and res.xnorm == pytest.approx(info["xnorm"], rel=max(xtol, 1e-9))


def test_cgls_damping_warm_start_and_overwrite(Jets, oracle):
    dt, nrow, shape, iters, damp = np.float64, 5, (24, 24, 3), 9, 0.35
    A, _, _, diags = make_tall_diag(Jets, oracle, dt, nrow, shape)
    n = int(np.prod(shape))
    matvec, rmatvec = _host_ops(diags, np.float64)
    hb = u01(oracle, dt, 52, 0, nrow * n) - 0.5
    hx0 = u01(oracle, dt, 53, 0, n) - 0.5
    b = Jets.from_numpy(hb, Jets.range(A))
    x0 = Jets.from_numpy(hx0.reshape(shape, order="F"))
    res = Jets.cgls(A, b, x0=x0, damp=damp, atol=0.0, btol=0.0, maxiter=iters, overwrite_b=True)
    xr, info = cgls_fp64(matvec, rmatvec, hb, n, x0=hx0, damp=damp, atol=0.0, btol=0.0, maxiter=iters)
    x = res.x.to_numpy().ravel(order="F")
    assert np.linalg.norm(x - xr) <= 1e-10 * np.linalg.norm(xr)
    r = hb - matvec(x)                                                               # b's storage now holds the residual
    assert np.linalg.norm(b.to_numpy() - r) <= 1e-10 * np.linalg.norm(r)
    assert res.r2norm == pytest.approx(np.sqrt(np.linalg.norm(r) ** 2 + damp ** 2 * np.linalg.norm(x) ** 2), rel=1e-10)
    assert np.array_equal(x0.to_numpy().ravel(order="F"), hx0)                       # x0 itself is not written


def test_cgls_consistent_system_stops_on_the_residual(Jets, oracle):
    dt, nrow, shape = np.float32, 8, (32, 32, 8)
    A, _, _, _ = make_tall_diag(Jets, oracle, dt, nrow, shape)
    x_true = Jets.rand(Jets.domain(A), seed=4, stream=0)
    b = A * x_true
    res = Jets.cgls(A, b, atol=1e-7, btol=1e-5, maxiter=200, overwrite_b=True)
    assert res.istop in (1, 2) and res.itn < 200
    err = (res.x - x_true).materialize()
    assert float(Jets.norm(err)) / float(Jets.norm(x_true)) < 1e-3


def test_cgls_generic_operator_runs_the_textbook_loop(Jets, oracle):
    """Rows of several kinds (identity, scalar) and vec(A): not an all-diagonal operator, so cgls_core drives mul! / the fused halves."""
    dt, n, shape = np.float64, 24, (4, 6)
    spc = Jets.JetSpace(dt, *shape)
    g1, g2 = Jets.rand(spc, seed=61, stream=1), Jets.rand(spc, seed=61, stream=2)
    scale = Jets.JopLn(dom=spc, rng=spc, df=Jets.constdiag_df, df_adj=Jets.constdiag_df_adj, s={"a": 0.5})
    A = Jets.blockop([Jets.JopDiagonal(g1), Jets.JopIdentity(spc), scale, Jets.JopDiagonal(g2)])
    h1, h2 = g1.to_numpy().ravel(order="F"), g2.to_numpy().ravel(order="F")
    matvec = lambda x: np.concatenate([h1 * x, x, 0.5 * x, h2 * x])
    rmatvec = lambda y: h1 * y[:n] + y[n:2 * n] + 0.5 * y[2 * n:3 * n] + h2 * y[3 * n:]
    hb = u01(oracle, dt, 62, 0, 4 * n) - 0.5
    b = Jets.from_numpy(hb, Jets.range(A))
    res = Jets.cgls(Jets.vec_op(A) if hasattr(Jets, "vec_op") else A, b, atol=0.0, btol=0.0, maxiter=8)
    xr, info = cgls_fp64(matvec, rmatvec, hb, n, atol=0.0, btol=0.0, maxiter=8)
    assert np.linalg.norm(res.x.to_numpy().ravel(order="F") - xr) <= 1e-10 * np.linalg.norm(xr)


def test_cgls_over_a_team_and_over_the_abi_communicator(Jets, oracle):
    """The same solve (a) over a single-process team of two contexts of this GPU (jh_cgls_solve_team: grouped ranged exchange of
    A'r, scalars added on the host) and (b) row-partitioned over the ABI's RCCL communicator with ONE rank and the exchange forced
    on (jh_cgls_solve_partitioned: the pipelined step + jh_comm_allreduce_normsq) against jh_cgls_solve on the whole operator."""
    import gc

    from jets_jl_amd import rowpart

    J = Jets
    dt, nrow, shape, iters = np.float32, 10, (64, 64, 20), 15                        # 81 920 elements: three exchange ranges
    n = int(np.prod(shape))
    spc = J.JetSpace(dt, *shape)
    home = J.context_current()[0]
    coeff = J.rand(J.JetBSpace([spc] * nrow), seed=81, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    hb = u01(oracle, dt, 82, 0, nrow * n) - dt(0.5)
    whole = J.cgls(A, J.from_numpy(hb, J.range(A)), atol=0.0, btol=0.0, maxiter=iters)
    xw = whole.x.to_numpy()
    whole_n = J.cgnr(A, J.from_numpy(hb, J.range(A)), atol=0.0, btol=0.0, maxiter=iters)       # CG through the fused A'A: the same iterates
    np.testing.assert_allclose(whole_n.x.to_numpy(), xw, rtol=2e-4, atol=1e-5)
    # (b) one-rank communicator, exchange forced
    comm = rowpart.AbiComm(nranks=1, rank=0)
    try:
        shard = rowpart.for_device(rowpart.partition_rows(nrow, 1, 0), A, comm=comm)
        J.tune(force_dist=1)
        dist = J.cgls(shard, J.from_numpy(hb, J.range(A)), atol=0.0, btol=0.0, maxiter=iters)
        dist_n = J.cgnr(shard, J.from_numpy(hb, J.range(A)), atol=0.0, btol=0.0, maxiter=iters)   # jh_cgnr_solve_partitioned: one vector all-reduce per iteration
    finally:
        J.tune(force_dist=0)
        comm.close()
    assert dist.itn == whole.itn == iters == dist_n.itn
    np.testing.assert_allclose(dist.x.to_numpy(), xw, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(dist_n.x.to_numpy(), whole_n.x.to_numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose([h[1] for h in dist.history], [h[1] for h in whole.history], rtol=1e-5)
    # (a) a team of two contexts: rows 0..5 and 6..9 (in a function of its own: every handle of the second context is gone when it returns)
    other = J.context_create(0)
    J.context_use(home)
    try:
        _team_leg(J, rowpart, [home, other], spc, n, hb, iters, xw, [h[1] for h in whole.history])
    finally:
        gc.collect()
        J.context_use(home)
        J.context_destroy(other)


def _team_leg(J, rowpart, ctxs, spc, n, hb, iters, xw, r_hist):
    team = rowpart.Team(ctxs)
    try:
        cuts = [(0, 6), (6, 10)]
        ops, bs, keep = [], [], []
        for k, _ in team.each():
            lo, hi = cuts[k]
            ck = J.rand(J.JetBSpace([spc] * (hi - lo)), seed=81, stream=0, index_base=lo * n)
            keep.append(ck)
            ops.append(J.blockop([[J.JopDiagonal(c)] for c in ck.arrays]))
            bs.append(J.from_numpy(hb[lo * n:hi * n], J.range(ops[-1])))
        T = team.operator(ops)
        res = J.cgls(T, rowpart.TeamVec(bs), atol=0.0, btol=0.0, maxiter=iters)
        assert res.itn == iters
        x0, x1 = res.x[0].to_numpy(), res.x[1].to_numpy()
        assert x0.tobytes() == x1.tobytes(), "the members' replicas of x differ"
        np.testing.assert_allclose(x0, xw, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose([h[1] for h in res.history], r_hist, rtol=1e-5)
        bs2 = []
        for k, _ in team.each():                                           # every member's rows of b in ITS context
            lo, hi = cuts[k]
            bs2.append(J.from_numpy(hb[lo * n:hi * n], J.range(ops[k])))
        resn = J.cgnr(T, rowpart.TeamVec(bs2), atol=0.0, btol=0.0, maxiter=iters)                   # jh_cgnr_solve_team
        assert resn.x[0].to_numpy().tobytes() == resn.x[1].to_numpy().tobytes()
        np.testing.assert_allclose(resn.x[0].to_numpy(), xw, rtol=2e-4, atol=1e-5)
        for A in ops:
            J.close(A)
    finally:
        team.close()


def test_cgls_argument_checks(Jets, oracle):
    import ctypes as C

    from jets_jl_amd._ffi import LsqrResultC, lib

    A, _, _, _ = make_tall_diag(Jets, oracle, np.float32, 1, (64,))                  # ONE row: the fused normal operator needs two
    b = Jets.rand(Jets.range(A), seed=1, stream=0)
    res = Jets.cgls(A, b, maxiter=3)                                                 # -> the generic loop, quietly
    assert res.itn >= 1
    from jets_jl_amd import jetblock

    nat = jetblock._native_op(A.jet.s["_native"], A.jet.s["ops"], A.jet.rng.eltype())
    x = Jets.zeros(Jets.domain(A))
    out = LsqrResultC()
    assert lib.jh_cgls_solve(nat.handle, b.handle, x.handle, 0, 0.0, 0.0, 0.0, 3, 0, C.byref(out), None) == 4      # JH_ERR_UNSUPPORTED, nothing touched
    assert lib.jh_cgls_solve(None, b.handle, x.handle, 0, 0.0, 0.0, 0.0, 3, 0, C.byref(out), None) != 0


@pytest.mark.parametrize("dt,xtol", [(np.float32, 2e-4), (np.float64, 1e-9), (np.complex64, 2e-4), (np.complex128, 1e-9)])
@pytest.mark.parametrize("native", ["1", "0"])
def test_cgnr_through_the_fused_normal_operator_matches_textbook_cgls(Jets, oracle, dt, xtol, native, monkeypatch):
    """CG on (A'A) x = A'b with the normal operator as ONE fused pass (jh_cgnr_solve; native = 0: A then A' through the engines) has, in
    exact arithmetic, the iterates of CGLS: checked against the TEXTBOOK fp64 CGLS (explicit q = A p and r) -- iterate, the ||r||
    recurrence and ||A'r|| per iteration.  b is only read."""
    monkeypatch.setenv("JETS_CGLS_NATIVE", native)
    nrow, shape, iters = 6, (16, 16, 16), 12
    A, _, _, diags = make_tall_diag(Jets, oracle, dt, nrow, shape)
    n = int(np.prod(shape))
    dt64 = np.complex128 if np.dtype(dt).kind == "c" else np.float64
    matvec, rmatvec = _host_ops(diags, dt64)
    hb = (u01(oracle, dt, 51, 0, nrow * n) - dt(0.5)).astype(dt)
    b = Jets.from_numpy(hb, Jets.range(A))
    res = Jets.cgnr(A, b, atol=0.0, btol=0.0, maxiter=iters)
    xr, info = cgls_fp64(matvec, rmatvec, hb.astype(dt64), n, atol=0.0, btol=0.0, maxiter=iters)
    assert res.itn == iters == info["itn"] and res.istop == 7
    x = res.x.to_numpy().ravel(order="F").astype(dt64)
    assert np.linalg.norm(x - xr) / np.linalg.norm(xr) < xtol
    for (i1, r1, ar1), (i2, r2, ar2) in zip(res.history, info["history"]):
        assert i1 == i2 and r1 == pytest.approx(r2, rel=max(10 * xtol, 1e-8)) and ar1 == pytest.approx(ar2, rel=max(100 * xtol, 1e-6))
    assert np.array_equal(b.to_numpy(), hb), "b is read, never written"
    assert res.r1norm == pytest.approx(info["r1norm"], rel=max(10 * xtol, 1e-8))


def test_cgnr_damping_and_warm_start(Jets, oracle):
    dt, nrow, shape, iters, damp = np.float64, 5, (24, 24, 3), 9, 0.35
    A, _, _, diags = make_tall_diag(Jets, oracle, dt, nrow, shape)
    n = int(np.prod(shape))
    matvec, rmatvec = _host_ops(diags, np.float64)
    hb = u01(oracle, dt, 52, 0, nrow * n) - 0.5
    hx0 = u01(oracle, dt, 53, 0, n) - 0.5
    res = Jets.cgnr(A, Jets.from_numpy(hb, Jets.range(A)), x0=Jets.from_numpy(hx0.reshape(shape, order="F")), damp=damp, atol=0.0, btol=0.0, maxiter=iters)
    xr, info = cgls_fp64(matvec, rmatvec, hb, n, x0=hx0, damp=damp, atol=0.0, btol=0.0, maxiter=iters)
    x = res.x.to_numpy().ravel(order="F")
    assert np.linalg.norm(x - xr) <= 1e-9 * np.linalg.norm(xr)
    r = hb - matvec(x)
    assert res.r1norm == pytest.approx(np.linalg.norm(r), rel=1e-7)                 # ||r|| from the recurrence, never from a pass over the range
    assert res.r2norm == pytest.approx(np.sqrt(np.linalg.norm(r) ** 2 + damp ** 2 * np.linalg.norm(x) ** 2), rel=1e-7)
