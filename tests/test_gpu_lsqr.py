"""GPU: fused solver updates (jh_blockop_mul_axpby / _mul_adj_axpby) and the LSQR driver.

Bar: the fused updates are BIT-EXACT against the unfused chain restated by the oracle (block loop into a
temporary, then `y .= alpha*tmp .+ beta*y`); their ||.||^2 within 1e-6 (Float32) / 1e-13 (Float64) of an
fp64 host value.  LSQR (Float32 data) after a fixed number of iterations: solution within 1e-4 (rel l2),
residual-norm history within 1e-4, of the fp64 CPU LSQR (oracle/lsqr_ref.py) on the same operator --
IterativeSolvers.jl, the reference's solver caller, is un-vendored, so LSQR parity is pinned on the
published algorithm only.
"""
import ctypes as C
import math

import numpy as np
import pytest

from oracle.lsqr_ref import lsqr_fp64

from .helpers import DTYPES, assert_bits_equal, make_tall_diag, u01

pytestmark = pytest.mark.gpu


def _native(Jets, A):
    from jets_jl_amd import jetblock

    j = A.jet
    return jetblock._native_op(j.s["_native"], j.s["ops"], j.rng.eltype())


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nrow,shape", [(4, (8, 8, 4)), (3, (100, 100, 27)), (2, (128, 128, 33)), (1, (64,)), (5, (1 << 20,))])
def test_fused_updates_bit_exact_vs_unfused_chain(Jets, oracle, dt, nrow, shape):
    from jets_jl_amd._ffi import lib, check

    if np.dtype(dt).itemsize * int(np.prod(shape)) * nrow > 2 ** 28:
        pytest.skip("kept small")
    A, _, ops, _ = make_tall_diag(Jets, oracle, dt, nrow, shape)
    n = int(np.prod(shape))
    nat = _native(Jets, A)
    alpha, beta = 0.75, -1.375
    m = Jets.rand(Jets.domain(A), seed=41, stream=0)
    d = Jets.rand(Jets.range(A), seed=42, stream=0)
    hm = u01(oracle, dt, 41, 0, n)
    hd = u01(oracle, dt, 42, 0, nrow * n)
    hd_blocks = [hd[i * n:(i + 1) * n].copy() for i in range(nrow)]

    # forward half:  d <- alpha*(A m) + beta*d
    out = C.c_double(0)
    check(lib.jh_blockop_mul_axpby(nat.handle, d.handle, m.handle, alpha, beta, C.byref(out)))
    tmp = oracle.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(nrow)], [hm])
    ref = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [alpha, beta], [tmp, hd_blocks])
    assert_bits_equal(d.to_numpy(), np.concatenate(ref), "d <- alpha*A m + beta*d")
    truth = float(np.sum(np.abs(np.concatenate(ref).astype(np.complex128)) ** 2))
    tol = 1e-6 if np.dtype(dt) in (np.dtype(np.float32), np.dtype(np.complex64)) else 1e-13
    assert out.value == pytest.approx(truth, rel=tol)

    # adjoint half:  m <- alpha*(A' d) + beta*m   (d is now `ref`)
    check(lib.jh_blockop_mul_adj_axpby(nat.handle, m.handle, d.handle, alpha, beta, 1.0, C.byref(out)))
    tmpm = oracle.block_df_adj(ops, [np.zeros(n, dtype=dt)], ref)
    refm = oracle.barr_lincomb([np.empty(n, dtype=dt)], [alpha, beta], [tmpm, [hm]])
    assert_bits_equal(m.to_numpy().ravel(order="F"), refm[0], "m <- alpha*A'd + beta*m")
    assert out.value == pytest.approx(float(np.sum(np.abs(refm[0].astype(np.complex128)) ** 2)), rel=tol)
    # normsq == NULL: asynchronous, same result
    check(lib.jh_blockop_mul_adj_axpby(nat.handle, m.handle, d.handle, 0.0, 1.0, 1.0, None))
    Jets.synchronize()


def _host_ops(a_blocks, dt64):
    a64 = [g.astype(dt64) for g in a_blocks]
    n = a64[0].size
    matvec = lambda x: np.concatenate([g * x for g in a64])
    rmatvec = lambda y: sum(np.conj(g) * y[i * n:(i + 1) * n] for i, g in enumerate(a64))
    return matvec, rmatvec


@pytest.mark.parametrize("dt,xtol", [(np.float32, 1e-4), (np.float64, 1e-10), (np.complex64, 1e-4)])
def test_lsqr_matches_fp64_cpu_lsqr(Jets, oracle, dt, xtol):
    nrow, shape, iters = 6, (16, 16, 16), 12
    A, _, _, diags = make_tall_diag(Jets, oracle, dt, nrow, shape)
    n = int(np.prod(shape))
    dt64 = np.complex128 if np.dtype(dt).kind == "c" else np.float64
    matvec, rmatvec = _host_ops(diags, dt64)
    hb = (u01(oracle, dt, 51, 0, nrow * n) - dt(0.5)).astype(dt)                    # inconsistent right-hand side
    b = Jets.from_numpy(hb, Jets.range(A))
    res = Jets.lsqr(A, b, atol=0.0, btol=0.0, conlim=0.0, maxiter=iters)
    xr, info = lsqr_fp64(matvec, rmatvec, hb.astype(dt64), n, atol=0.0, btol=0.0, conlim=0.0, maxiter=iters)
    assert res.itn == iters == info["itn"]
    x = res.x.to_numpy().ravel(order="F").astype(dt64)
    assert np.linalg.norm(x - xr) / np.linalg.norm(xr) < xtol
    for (i1, r1, ar1), (i2, r2, ar2) in zip(res.history, info["history"]):
        assert i1 == i2 and r1 == pytest.approx(r2, rel=max(xtol, 1e-9))
    assert np.array_equal(b.to_numpy(), hb)                                          # b untouched (overwrite_b=False)
    # closed form: the normal equations are diagonal
    a64 = np.stack([g.astype(dt64) for g in diags])
    x_ls = (np.conj(a64) * hb.astype(dt64).reshape(nrow, n)).sum(0) / (np.abs(a64) ** 2).sum(0)
    e_gpu, e_cpu = np.linalg.norm(x - x_ls), np.linalg.norm(xr - x_ls)              # both are `iters` steps from x_ls
    assert e_gpu <= 1.01 * e_cpu + xtol * np.linalg.norm(x_ls)


def test_lsqr_generic_path_and_vec_and_warm_start(Jets, oracle):
    """Operators that are not all-diagonal take the mul! + fused-broadcast + norm path; vec(A) is unwrapped."""
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
    xr, info = lsqr_fp64(matvec, rmatvec, hb, n, atol=1e-14, btol=1e-14, maxiter=50)
    res = Jets.lsqr(Jets.vec_op(A), b, atol=1e-14, btol=1e-14, maxiter=50)           # lsqr(vec(A), vec(d))  (docs/src/index.md:240)
    assert res.x.shape == shape
    assert np.allclose(res.x.to_numpy().ravel(order="F"), xr, rtol=1e-10, atol=1e-12)
    x0 = Jets.from_numpy((xr + 0.01).reshape(shape, order="F"))
    res2 = Jets.lsqr(A, b, x0=x0, atol=1e-14, btol=1e-14, maxiter=50)
    assert np.allclose(res2.x.to_numpy().ravel(order="F"), xr, rtol=1e-9, atol=1e-11)
    xd, _ = lsqr_fp64(matvec, rmatvec, hb, n, damp=0.3, atol=1e-14, btol=1e-14, maxiter=50)
    resd = Jets.lsqr(A, b, damp=0.3, atol=1e-14, btol=1e-14, maxiter=50)
    assert np.allclose(resd.x.to_numpy().ravel(order="F"), xd, rtol=1e-9, atol=1e-11)


def test_lsqr_consistent_system_recovers_x_true_and_stops(Jets, oracle):
    dt, nrow, shape = np.float32, 8, (32, 32, 8)
    A, _, _, _ = make_tall_diag(Jets, oracle, dt, nrow, shape)
    x_true = Jets.rand(Jets.domain(A), seed=4, stream=0)
    b = A * x_true
    res = Jets.lsqr(A, b, atol=1e-6, btol=1e-6, maxiter=100, overwrite_b=True)
    assert res.istop in (1, 2) and res.itn < 100
    err = (res.x - x_true).materialize()
    assert float(Jets.norm(err)) / float(Jets.norm(x_true)) < 1e-4


def test_rccl_path_with_one_rank(Jets, oracle):
    """The product wiring of the exchange step (rowpart.for_device): zero-copy torch view of the library's
    slab, collective ordered on the library's HIP stream.  One rank only here (one GPU); the two-rank logic
    is covered by the gloo tests."""
    import os

    import torch
    import torch.distributed as dist

    if dist.is_initialized():
        pytest.skip("a process group already exists")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        dt, nrow, shape = np.float32, 4, (16, 8, 8)
        A, _, ops, _ = make_tall_diag(Jets, oracle, dt, nrow, shape)
        n = int(np.prod(shape))
        part = Jets.rowpart.partition_rows(nrow, 1, 0)
        shard = Jets.rowpart.for_device(part, A)
        d = Jets.rand(Jets.range(A), seed=71, stream=0)
        hd = u01(oracle, dt, 71, 0, nrow * n)
        mt = Jets.zeros(Jets.domain(A))
        shard.mul_adj_(mt, d, force_collective=True)                 # local adjoint + ncclAllReduce over 1 rank
        Jets.synchronize()
        ref = oracle.block_df_adj(ops, [np.zeros(n, dtype=dt)], [hd[i * n:(i + 1) * n].copy() for i in range(nrow)])
        assert_bits_equal(mt.to_numpy().ravel(order="F"), ref[0], "adjoint through the RCCL path")
        assert shard.dot_range(d, d) == pytest.approx(float(np.dot(hd.astype(np.float64), hd.astype(np.float64))), rel=1e-6)
        assert shard.norm_range(d, 2) == pytest.approx(float(np.linalg.norm(hd.astype(np.float64))), rel=1e-6)
        v = Jets.rand(Jets.domain(A), seed=70, stream=0)
        yn = Jets.rand(Jets.domain(A), seed=69, stream=0)
        shard.normal_mul_(yn, v, force_collective=True)              # the fused A'A in ranges, each all-reduced by torch.distributed
        Jets.synchronize()
        assert_bits_equal(yn.to_numpy(), Jets.mul_(Jets.zeros(Jets.domain(A)), A.H @ A, v).to_numpy(), "fused A'A through the RCCL path")
    finally:
        dist.destroy_process_group()


def test_c_abi_rccl_entry_points_with_one_rank(Jets, oracle):
    """jh_comm_* (the RCCL path a non-Python host uses): id, init, vector and scalar all-reduce, destroy."""
    comm = Jets.rowpart.AbiComm(nranks=1, rank=0)
    try:
        dt, nrow, shape = np.float32, 3, (16, 16, 4)
        A, _, ops, _ = make_tall_diag(Jets, oracle, dt, nrow, shape)
        n = int(np.prod(shape))
        shard = Jets.rowpart.for_device(Jets.rowpart.partition_rows(nrow, 1, 0), A, comm=comm)
        d = Jets.rand(Jets.range(A), seed=72, stream=0)
        hd = u01(oracle, dt, 72, 0, nrow * n)
        mt = Jets.zeros(Jets.domain(A))
        shard.mul_adj_(mt, d, force_collective=True)                # ncclAllReduce on the library stream
        ref = oracle.block_df_adj(ops, [np.zeros(n, dtype=dt)], [hd[i * n:(i + 1) * n].copy() for i in range(nrow)])
        assert_bits_equal(mt.to_numpy().ravel(order="F"), ref[0], "adjoint through jh_comm_allreduce_sum")
        assert comm.all_reduce_scalars([1.5, -2.0], "sum") == [1.5, -2.0]
        assert comm.all_reduce_scalars([3.0], "max") == [3.0]
        x_true = Jets.rand(Jets.domain(A), seed=73, stream=0)
        res = Jets.lsqr(shard, A * x_true, atol=1e-5, btol=1e-5, maxiter=200)          # row-partitioned engine, one rank
        err = (res.x - x_true).materialize()
        assert float(Jets.norm(err)) / float(Jets.norm(x_true)) < 1e-3
        # the pipelined exchange of the ABI (its own stream + events) with the one-rank communicator: 81 920 elements = three ranges
        shape2 = (64, 64, 20)
        n2 = int(np.prod(shape2))
        B, _, ops2, _ = make_tall_diag(Jets, oracle, dt, 5, shape2)
        shard2 = Jets.rowpart.for_device(Jets.rowpart.partition_rows(5, 1, 0), B, comm=comm)
        d2 = Jets.rand(Jets.range(B), seed=74, stream=0)
        hd2 = u01(oracle, dt, 74, 0, 5 * n2)
        mt2 = Jets.rand(Jets.domain(B), seed=75, stream=0)            # dirty
        shard2.mul_adj_(mt2, d2, force_collective=True)               # jh_blockop_mul_adj_range + jh_comm_allreduce_sum_range x 3, jh_comm_join
        ref2 = oracle.block_df_adj(ops2, [np.zeros(n2, dtype=dt)], [hd2[i * n2:(i + 1) * n2].copy() for i in range(5)])
        assert_bits_equal(mt2.to_numpy().ravel(order="F"), ref2[0], "pipelined adjoint through the ABI's exchange stream")
        v = Jets.rand(Jets.domain(B), seed=76, stream=0)
        yn = Jets.rand(Jets.domain(B), seed=79, stream=0)            # dirty
        shard2.normal_mul_(yn, v, force_collective=True)              # jh_blockop_normal_mul_range + jh_comm_allreduce_sum_range x 3, jh_comm_join
        assert_bits_equal(yn.to_numpy(), Jets.mul_(Jets.zeros(Jets.domain(B)), B.H @ B, v).to_numpy(), "pipelined fused A'A through the ABI's exchange stream")
        u1, u2 = Jets.rand(Jets.range(B), seed=77, stream=0), Jets.rand(Jets.range(B), seed=77, stream=0)
        w1, w2 = Jets.zeros(Jets.domain(B)), Jets.zeros(Jets.domain(B))
        nrm2 = shard2.bidiag_step_(u1, v, w1, 0.75, -0.5, force_collective=True)       # ranged steps + ranged all-reduces + jh_comm_allreduce_normsq
        import ctypes as C
        from jets_jl_amd._ffi import lib, check
        out = C.c_double(0)
        check(lib.jh_blockop_bidiag_step(_native(Jets, B).handle, u2.handle, v.handle, w2.handle, 0.75, -0.5, C.byref(out)))
        assert_bits_equal(u1.to_numpy(), u2.to_numpy(), "pipelined step: u")
        assert_bits_equal(w1.to_numpy(), w2.to_numpy(), "pipelined step: w")
        assert nrm2 == pytest.approx(out.value, rel=1e-13)
        # jh_lsqr_solve_partitioned with its exchange forced on (knob force_dist): the C++ loop's pipelined distributed step
        x2 = Jets.rand(Jets.domain(B), seed=78, stream=0)
        try:
            Jets.tune(force_dist=1)
            r_dist = Jets.lsqr(shard2, B * x2, atol=0.0, btol=0.0, conlim=0.0, maxiter=20)
        finally:
            Jets.tune(force_dist=0)
        r_loc = Jets.lsqr(B, B * x2, atol=0.0, btol=0.0, conlim=0.0, maxiter=20)
        assert r_dist.itn == r_loc.itn == 20
        np.testing.assert_allclose(r_dist.x.to_numpy(), r_loc.x.to_numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose([h[1] for h in r_dist.history], [h[1] for h in r_loc.history], rtol=1e-6)
    finally:
        comm.close()
    with pytest.raises(Jets.JetsHipError):
        from jets_jl_amd._ffi import lib, check
        check(lib.jh_comm_allreduce_sum(mt.handle))                  # no communicator any more: loud


@pytest.mark.parametrize("dt", [np.float32, np.complex128])
def test_ranged_adjoint_equals_whole_adjoint(Jets, oracle, dt):
    """jh_blockop_mul_adj_range over consecutive element ranges == jh_blockop_mul_adj, bit for bit."""
    from jets_jl_amd._ffi import lib, check

    nrow, shape = 5, (64, 64, 20)                                   # 81920 elements
    A, _, ops, _ = make_tall_diag(Jets, oracle, dt, nrow, shape)
    nat = _native(Jets, A)
    d = Jets.rand(Jets.range(A), seed=81, stream=0)
    whole = Jets.mul_(Jets.zeros(Jets.domain(A)), A.H, d)
    parts = Jets.rand(Jets.domain(A), seed=82, stream=0)            # dirty: every element must be overwritten
    n = parts.length()
    for lo, cnt in ((0, 16384), (16384, 32768), (49152, 4), (49156, n - 49156)):
        check(lib.jh_blockop_mul_adj_range(nat.handle, parts.handle, d.handle, lo, cnt))
    assert_bits_equal(parts.to_numpy(), whole.to_numpy(), "ranged adjoint")
    if np.dtype(dt).itemsize < 16:
        with pytest.raises(Jets.JetsHipError):
            check(lib.jh_blockop_mul_adj_range(nat.handle, parts.handle, d.handle, 1, 4))   # chunk start off a 16-byte boundary
    with pytest.raises(Jets.JetsHipError):
        check(lib.jh_blockop_mul_adj_range(nat.handle, parts.handle, d.handle, n - 4, 8))   # past the end


@pytest.mark.parametrize("dt", [np.float32, np.complex128])
@pytest.mark.parametrize("mixed", [False, True])
def test_ranged_normal_operator_equals_the_whole_one(Jets, oracle, dt, mixed):
    """jh_blockop_normal_mul_range over consecutive element ranges == jh_blockop_normal_mul (== the unfused chain), bit for bit; all-diagonal
    rows and rows of several kinds."""
    from jets_jl_amd._ffi import lib, check

    nrow, shape = 5, (64, 64, 20)
    A, diags, ops, _ = make_tall_diag(Jets, oracle, dt, nrow, shape)
    if mixed:
        spc = Jets.JetSpace(dt, *shape)
        A = Jets.blockop([[Jets.JopDiagonal(diags[0])], [Jets.JopIdentity(spc)], [Jets.JopZeroBlock(spc, spc)], [Jets.JopDiagonal(diags[1]).H], [Jets.JopDiagonal(diags[2])]])
    nat = _native(Jets, A)
    m = Jets.rand(Jets.domain(A), seed=83, stream=0)
    whole = Jets.mul_(Jets.zeros(Jets.domain(A)), A.H @ A, m)
    chain = Jets.mul_(Jets.zeros(Jets.domain(A)), A.H, Jets.mul_(Jets.zeros(Jets.range(A)), A, m))
    assert_bits_equal(whole.to_numpy(), chain.to_numpy(), "fused A'A vs the chain")
    parts = Jets.rand(Jets.domain(A), seed=84, stream=0)            # dirty: every element must be overwritten
    n = parts.length()
    for lo, cnt in ((0, 16384), (16384, 32768), (49152, 4), (49156, n - 49156)):
        check(lib.jh_blockop_normal_mul_range(nat.handle, parts.handle, m.handle, lo, cnt))
    assert_bits_equal(parts.to_numpy(), whole.to_numpy(), "ranged A'A")
    if np.dtype(dt).itemsize < 16:
        with pytest.raises(Jets.JetsHipError):
            check(lib.jh_blockop_normal_mul_range(nat.handle, parts.handle, m.handle, 1, 4))
    with pytest.raises(Jets.JetsHipError):
        check(lib.jh_blockop_normal_mul_range(nat.handle, parts.handle, m.handle, n - 4, 8))
    with pytest.raises(Jets.JetsHipError):
        check(lib.jh_blockop_normal_mul_range(nat.handle, m.handle, m.handle, 0, 16))          # y aliases m


# ---------------------------------------------------------------------------------- one-pass Golub-Kahan step
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nrow,shape,beta", [(4, (8, 8, 4), -1.375), (3, (100, 100, 27), 0.5), (2, (128, 128, 33), -2.0), (1, (64,), 0.25),
                                             (5, (1 << 20,), -1.375), (7, (40, 40, 12), 0.0), (9, (1 << 18,), -0.3)])
def test_bidiag_step_bit_exact_vs_the_two_halves(Jets, oracle, dt, nrow, shape, beta):
    """jh_blockop_bidiag_step == jh_blockop_mul_axpby then jh_blockop_mul_adj, in one pass: u and w BIT-EXACT against the
    oracle's unfused chain (block loop into a temporary, broadcast, adjoint block loop), ||u||^2 within 1e-6 / 1e-13."""
    from jets_jl_amd._ffi import lib, check

    A, _, ops, _ = make_tall_diag(Jets, oracle, dt, nrow, shape)
    n = int(np.prod(shape))
    nat = _native(Jets, A)
    alpha = 0.75
    v = Jets.rand(Jets.domain(A), seed=51, stream=0)
    u = Jets.rand(Jets.range(A), seed=52, stream=0)
    w = Jets.rand(Jets.domain(A), seed=53, stream=0)                                  # dirty: must be overwritten
    hv, hu = u01(oracle, dt, 51, 0, n), u01(oracle, dt, 52, 0, nrow * n)
    hu_blocks = [hu[i * n:(i + 1) * n].copy() for i in range(nrow)]
    out = C.c_double(0)
    check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, alpha, beta, C.byref(out)))
    tmp = oracle.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(nrow)], [hv])
    if beta != 0.0:
        ref_u = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [alpha, beta], [tmp, hu_blocks])
    else:
        ref_u = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [alpha], [tmp])     # beta == 0: u is write-only
    assert_bits_equal(u.to_numpy(), np.concatenate(ref_u), "u <- alpha*A v + beta*u")
    ref_w = oracle.block_df_adj(ops, [np.full(n, 5, dtype=dt)], ref_u)
    assert_bits_equal(w.to_numpy().ravel(order="F"), ref_w[0], "w <- A'u")
    truth = float(np.sum(np.abs(np.concatenate(ref_u).astype(np.complex128)) ** 2))
    tol = 1e-6 if np.dtype(dt) in (np.dtype(np.float32), np.dtype(np.complex64)) else 1e-13
    assert out.value == pytest.approx(truth, rel=tol)


@pytest.mark.parametrize("knobs", [dict(adj_wg=256, adj_unroll=1, adj_depth=4), dict(adj_wg=256, adj_unroll=2, adj_depth=2),
                                   dict(adj_wg=256, adj_unroll=4, adj_depth=1), dict(adj_wg=256, adj_unroll=4, adj_depth=2),
                                   dict(adj_wg=512, adj_unroll=1, adj_depth=4), dict(adj_wg=512, adj_unroll=2, adj_depth=2),
                                   dict(adj_wg=512, adj_unroll=4, adj_depth=1), dict(adj_wg=512, adj_unroll=4, adj_depth=2),
                                   dict(adj_wg=1024, adj_unroll=1, adj_depth=4), dict(adj_wg=1024, adj_unroll=2, adj_depth=2),
                                   dict(adj_wg=1024, adj_unroll=4, adj_depth=1)])
def test_bidiag_step_every_shape_gives_identical_bits(Jets, oracle, knobs):
    from jets_jl_amd._ffi import lib, check

    dt, nrow, shape = np.float32, 11, (40, 40, 12)
    A, _, ops, _ = make_tall_diag(Jets, oracle, dt, nrow, shape)
    n = int(np.prod(shape))
    nat = _native(Jets, A)
    hv, hu = u01(oracle, dt, 51, 0, n), u01(oracle, dt, 52, 0, nrow * n)
    tmp = oracle.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(nrow)], [hv])
    ref_u = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [1.0, -0.625], [tmp, [hu[i * n:(i + 1) * n].copy() for i in range(nrow)]])
    ref_w = oracle.block_df_adj(ops, [np.zeros(n, dtype=dt)], ref_u)
    saved = {k: Jets.tune_get(k) for k in knobs}
    try:
        Jets.tune(**knobs)
        v = Jets.rand(Jets.domain(A), seed=51, stream=0)
        u = Jets.rand(Jets.range(A), seed=52, stream=0)
        w = Jets.zeros(Jets.domain(A))
        out = C.c_double(0)
        check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, 1.0, -0.625, C.byref(out)))
        assert_bits_equal(u.to_numpy(), np.concatenate(ref_u), f"u {knobs}")
        assert_bits_equal(w.to_numpy().ravel(order="F"), ref_w[0], f"w {knobs}")
    finally:
        Jets.tune(**saved)


def test_bidiag_step_argument_checks(Jets, oracle):
    from jets_jl_amd._ffi import lib

    A, _, _, _ = make_tall_diag(Jets, oracle, np.float32, 3, (64,))
    nat = _native(Jets, A)
    u, v = Jets.zeros(Jets.range(A)), Jets.zeros(Jets.domain(A))
    out = C.c_double(0)
    assert lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, v.handle, 1.0, 0.0, C.byref(out)) == 1     # w aliases v
    short = Jets.zeros(Jets.JetSpace(np.float32, 63))
    assert lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, short.handle, 1.0, 0.0, C.byref(out)) == 1
    spc = Jets.JetSpace(np.float32, 64)
    B = Jets.blockop([[Jets.JopDiagonal(Jets.rand(spc))], [Jets.JopIdentity(spc)]])                              # not all-DIAG: per-row kinds
    assert lib.jh_blockop_bidiag_step(_native(Jets, B).handle, Jets.zeros(Jets.range(B)).handle, v.handle,
                                      Jets.zeros(spc).handle, 1.0, 0.0, C.byref(out)) == 0
    odd = Jets.JetSpace(np.float32, 63)                                                                          # 252-byte rows: off the 16-byte pack grid --
    Cop = Jets.blockop([[Jets.JopDiagonal(Jets.rand(odd))], [Jets.JopIdentity(odd)]])                            # under-aligned packs since round 5 (tests/test_gpu_tall_unaligned.py)
    assert lib.jh_blockop_bidiag_step(_native(Jets, Cop).handle, Jets.zeros(Jets.range(Cop)).handle, Jets.zeros(odd).handle,
                                      Jets.zeros(odd).handle, 1.0, 0.0, C.byref(out)) == 0
    tiny = Jets.JetSpace(np.float32, 3)                                                                          # 12-byte rows: less than one pack
    Dop = Jets.blockop([[Jets.JopDiagonal(Jets.rand(tiny))], [Jets.JopIdentity(tiny)]])
    assert lib.jh_blockop_bidiag_step(_native(Jets, Dop).handle, Jets.zeros(Jets.range(Dop)).handle, Jets.zeros(tiny).handle,
                                      Jets.zeros(tiny).handle, 1.0, 0.0, C.byref(out)) == 4                        # JH_ERR_UNSUPPORTED


def test_lsqr_one_pass_and_two_pass_iterations_agree(Jets, oracle, monkeypatch):
    """The solver's default (one fused pass per iteration) against the two-half schedule: same residual history and
    solution to fp32 rounding (the adjoint half rounds conj(a)*(u/beta) vs (conj(a)*u)/beta)."""
    dt, nrow, shape = np.float32, 6, (32, 32, 8)
    A, _, _, _ = make_tall_diag(Jets, oracle, dt, nrow, shape)
    x_true = Jets.rand(Jets.domain(A), seed=61, stream=0)
    b = Jets.mul(A, x_true)
    monkeypatch.setenv("JETS_LSQR_FUSED_STEP", "1")
    r1 = Jets.lsqr(A, b, maxiter=15, atol=0.0, btol=0.0, force_maxiter=True)
    monkeypatch.setenv("JETS_LSQR_FUSED_STEP", "0")
    r0 = Jets.lsqr(A, b, maxiter=15, atol=0.0, btol=0.0, force_maxiter=True)
    assert r1.itn == r0.itn == 15
    h1, h0 = np.array([h[1] for h in r1.history]), np.array([h[1] for h in r0.history])
    np.testing.assert_allclose(h1[:8], h0[:8], rtol=1e-3)
    np.testing.assert_allclose(r1.x.to_numpy(), r0.x.to_numpy(), rtol=2e-4, atol=2e-5)
    e1 = np.linalg.norm(r1.x.to_numpy() - x_true.to_numpy()) / np.linalg.norm(x_true.to_numpy())
    e0 = np.linalg.norm(r0.x.to_numpy() - x_true.to_numpy()) / np.linalg.norm(x_true.to_numpy())
    assert e1 == pytest.approx(e0, rel=1e-2) and e1 < 0.1                              # same convergence after 15 iterations


def test_lsqr_one_pass_with_damping_and_warm_start(Jets, oracle):
    """damp (Tikhonov, handled in the scalar recurrences) and x0 (u = b - A x0 through the fused forward) on the tall
    fast path vs the fp64 CPU LSQR."""
    dt, nrow, shape, iters = np.float64, 5, (24, 24, 6), 25
    A, _, _, diags = make_tall_diag(Jets, oracle, dt, nrow, shape)
    n = int(np.prod(shape))
    matvec, rmatvec = _host_ops(diags, np.float64)
    hb = u01(oracle, dt, 71, 0, nrow * n) - 0.5
    b = Jets.from_numpy(hb, Jets.range(A))
    xd, info = lsqr_fp64(matvec, rmatvec, hb, n, damp=0.4, atol=0.0, btol=0.0, conlim=0.0, maxiter=iters)
    res = Jets.lsqr(A, b, damp=0.4, atol=0.0, btol=0.0, conlim=0.0, maxiter=iters)
    assert res.itn == info["itn"]
    np.testing.assert_allclose(res.x.to_numpy().ravel(order="F"), xd, rtol=1e-9, atol=1e-11)
    x_un = (np.stack(diags) * hb.reshape(nrow, n)).sum(0) / (np.stack(diags) ** 2).sum(0)              # undamped least squares
    x0 = Jets.from_numpy((x_un + 0.01).reshape(shape, order="F"))
    res2 = Jets.lsqr(A, b, x0=x0, atol=0.0, btol=0.0, conlim=0.0, maxiter=40)
    e0 = 0.01 * np.sqrt(n)
    e1 = np.linalg.norm(res2.x.to_numpy().ravel(order="F") - x_un)
    assert e1 < 0.2 * e0                                                             # the warm start is improved, not discarded


@pytest.mark.parametrize("dt", DTYPES)
def test_ranged_bidiag_step_equals_the_whole_step(Jets, oracle, dt):
    """jh_blockop_bidiag_step_range over a partition of the domain == jh_blockop_bidiag_step: same u, same w, shares of
    ||u||^2 adding up (what the pipelined multi-GPU LSQR step runs per chunk)."""
    from jets_jl_amd._ffi import lib, check

    nrow, shape = 5, (48, 40, 9)
    A, _, _, _ = make_tall_diag(Jets, oracle, dt, nrow, shape)
    n = int(np.prod(shape))
    nat = _native(Jets, A)
    v = Jets.rand(Jets.domain(A), seed=81, stream=0)
    u1, u2 = Jets.rand(Jets.range(A), seed=82, stream=0), Jets.rand(Jets.range(A), seed=82, stream=0)
    w1, w2 = Jets.zeros(Jets.domain(A)), Jets.zeros(Jets.domain(A))
    out = C.c_double(0)
    check(lib.jh_blockop_bidiag_step(nat.handle, u1.handle, v.handle, w1.handle, 0.75, -1.375, C.byref(out)))
    whole = out.value
    total, lo = 0.0, 0
    for cnt in (4096, 16, n // 2 // 16 * 16, 0):                                      # ragged chunks, one empty
        check(lib.jh_blockop_bidiag_step_range(nat.handle, u2.handle, v.handle, w2.handle, 0.75, -1.375, lo, cnt, C.byref(out)))
        total += out.value
        lo += cnt
    check(lib.jh_blockop_bidiag_step_range(nat.handle, u2.handle, v.handle, w2.handle, 0.75, -1.375, lo, n - lo, C.byref(out)))
    total += out.value
    assert_bits_equal(u2.to_numpy(), u1.to_numpy(), "u: chunks == whole")
    assert_bits_equal(w2.to_numpy(), w1.to_numpy(), "w: chunks == whole")
    assert total == pytest.approx(whole, rel=1e-12)
    if np.dtype(dt).itemsize < 16:
        assert lib.jh_blockop_bidiag_step_range(nat.handle, u2.handle, v.handle, w2.handle, 1.0, 0.0, 1, 16, C.byref(out)) == 1   # unaligned bound
    assert lib.jh_blockop_bidiag_step_range(nat.handle, u2.handle, v.handle, w2.handle, 1.0, 0.0, 0, n + 16, C.byref(out)) == 1


def test_pipelined_distributed_step_with_one_rank(Jets, oracle, monkeypatch):
    """The row-partitioned one-pass LSQR step with its chunked all-reduce, forced to run with ONE rank over RCCL:
    same iterates as the single-process solver."""
    import os

    import torch
    import torch.distributed as dist

    if dist.is_initialized():
        pytest.skip("a process group already exists")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        dt, nrow, shape = np.float32, 6, (64, 64, 16)
        A, _, _, _ = make_tall_diag(Jets, oracle, dt, nrow, shape)
        x_true = Jets.rand(Jets.domain(A), seed=91, stream=0)
        b = Jets.mul(A, x_true)
        ref = Jets.lsqr(A, b, maxiter=12, atol=0.0, btol=0.0, force_maxiter=True)
        shard = Jets.rowpart.for_device(Jets.rowpart.partition_rows(nrow, 1, 0), A)
        monkeypatch.setenv("BENCH_FORCE_DIST", "1")
        res = Jets.lsqr(shard, b, maxiter=12, atol=0.0, btol=0.0, force_maxiter=True)
        assert res.itn == ref.itn == 12
        np.testing.assert_allclose([h[1] for h in res.history], [h[1] for h in ref.history], rtol=1e-5)
        np.testing.assert_allclose(res.x.to_numpy(), ref.x.to_numpy(), rtol=1e-5, atol=1e-6)
        u, v, w = Jets.rand(Jets.range(A), seed=92, stream=0), Jets.rand(Jets.domain(A), seed=93, stream=0), Jets.zeros(Jets.domain(A))
        nrm2 = shard.bidiag_step_(u, v, w, 1.0, -0.5, force_collective=True)          # the chunked path really ran
        assert nrm2 is not None and nrm2 > 0
        Jets.synchronize()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64])
def test_native_lsqr_behind_the_abi_matches_the_python_driver(Jets, oracle, dt, monkeypatch):
    """jh_lsqr_solve (the loop in C++ behind the ABI) against lsqr_core (the same recurrences in Python over the same
    kernels): identical iterates, with damping, a warm start and early stopping; and against the fp64 CPU LSQR."""
    nrow, shape = 6, (24, 16, 8)
    A, _, _, diags = make_tall_diag(Jets, oracle, dt, nrow, shape)
    n = int(np.prod(shape))
    hb = (u01(oracle, dt, 95, 0, nrow * n) - dt(0.5)).astype(dt)
    b = Jets.from_numpy(hb, Jets.range(A))
    x0 = Jets.rand(Jets.domain(A), seed=96, stream=0)
    for kw in (dict(maxiter=14, atol=0.0, btol=0.0, conlim=0.0), dict(maxiter=14, atol=0.0, btol=0.0, damp=0.3),
               dict(maxiter=40, atol=1e-5, btol=1e-5), dict(maxiter=9, atol=0.0, btol=0.0, x0=x0), dict(maxiter=6, atol=0.0, btol=0.0, force_maxiter=True)):
        monkeypatch.setenv("JETS_LSQR_NATIVE", "1")
        r_nat = Jets.lsqr(A, b, **kw)
        monkeypatch.setenv("JETS_LSQR_NATIVE", "0")
        r_py = Jets.lsqr(A, b, **kw)
        # same kernels for the big pass; the native loop fuses the domain-side updates and sums its norms in another
        # (deterministic) order, and the Python driver rounds every norm to the element precision (norm() returns
        # real(eltype), like the reference): equal to fp64 round-off for Float64, to single-precision round-off otherwise
        exact = np.dtype(dt) == np.float64
        rt = 1e-12 if exact else 2e-6
        assert (r_nat.itn, r_nat.istop) == (r_py.itn, r_py.istop), kw
        np.testing.assert_allclose([h[1] for h in r_nat.history], [h[1] for h in r_py.history], rtol=rt)
        if exact:
            np.testing.assert_allclose(r_nat.x.to_numpy(), r_py.x.to_numpy(), rtol=1e-11, atol=1e-13)
        else:
            np.testing.assert_allclose(r_nat.x.to_numpy(), r_py.x.to_numpy(), rtol=1e-4, atol=1e-5)
        for f in ("r1norm", "r2norm", "anorm", "xnorm"):
            assert getattr(r_nat, f) == pytest.approx(getattr(r_py, f), rel=max(rt, 1e-6)), f
    assert np.array_equal(b.to_numpy(), hb)                                          # overwrite_b=False: b untouched
    dt64 = np.complex128 if np.dtype(dt).kind == "c" else np.float64
    matvec, rmatvec = _host_ops(diags, dt64)
    xr, info = lsqr_fp64(matvec, rmatvec, hb.astype(dt64), n, atol=0.0, btol=0.0, conlim=0.0, maxiter=14)
    monkeypatch.setenv("JETS_LSQR_NATIVE", "1")
    r = Jets.lsqr(A, b, maxiter=14, atol=0.0, btol=0.0, conlim=0.0)
    tol = 1e-10 if np.dtype(dt) == np.float64 else 1e-4
    assert np.linalg.norm(r.x.to_numpy().ravel(order="F") - xr) / np.linalg.norm(xr) < tol


def test_bidiag_step_in_several_row_launches(Jets, oracle):
    """knob adj_rows_per_launch on the one-pass step: w's ordered sum continues across launches, ||u||^2 adds up."""
    from jets_jl_amd._ffi import lib, check

    dt, nrow, shape = np.float32, 11, (40, 40, 12)
    A, _, _, _ = make_tall_diag(Jets, oracle, dt, nrow, shape)
    nat = _native(Jets, A)
    v = Jets.rand(Jets.domain(A), seed=51, stream=0)
    outs = []
    for rows in (0, 4):
        Jets.tune(adj_rows_per_launch=rows)
        try:
            u = Jets.rand(Jets.range(A), seed=52, stream=0)
            w = Jets.rand(Jets.domain(A), seed=53, stream=0)
            out = C.c_double(0)
            check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, 0.75, -1.375, C.byref(out)))
            outs.append((u.to_numpy(), w.to_numpy(), out.value))
        finally:
            Jets.tune(adj_rows_per_launch=0)
    assert_bits_equal(outs[1][0], outs[0][0], "u")
    assert_bits_equal(outs[1][1], outs[0][1], "w")
    assert outs[1][2] == pytest.approx(outs[0][2], rel=1e-12)


@pytest.mark.parametrize("native", ["1", "0"])
def test_forced_iterations_far_past_convergence_stay_finite(Jets, monkeypatch, native):
    """force_maxiter (throughput runs) keeps iterating after every stopping rule has fired; once the recurrences underflow the loop
    must end with x at its last finite update -- not divide by a rhobar that has become 0 (C++: NaN everywhere; Python: ZeroDivisionError)."""
    J = Jets
    spc = J.JetSpace(np.float32, 64, 64, 32)
    coeff = J.rand(J.JetBSpace([spc] * 12), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    x_true = J.rand(spc, seed=4, stream=0)
    b = A * x_true
    monkeypatch.setenv("JETS_LSQR_NATIVE", native)
    res = J.lsqr(A, b, atol=0.0, btol=0.0, conlim=0.0, maxiter=4000, force_maxiter=True)
    x = res.x.to_numpy()
    assert np.isfinite(x).all() and res.itn < 4000 and res.istop != 0
    assert all(np.isfinite(h[1]) and np.isfinite(h[2]) for h in res.history)
    assert float(np.linalg.norm((x - x_true.to_numpy()).ravel()) / np.linalg.norm(x_true.to_numpy().ravel())) < 1e-5


@pytest.mark.parametrize("shape", [(32, 16, 8), (31, 17, 7)], ids=["aligned", "odd-blocks"])    # 3689 elements: rows off the 16-byte pack grid (round 5)
@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64])
def test_graph_replayed_loop_of_small_operators_has_the_bits_of_the_host_loop(Jets, oracle, dt, shape):
    """Small operators: the recurrences live on the device and one iteration is replayed as a hipGraph (jh_lsqr.hip: lsqr_graph_impl)
    -- same kernels, same fp64 operations in the same order as the host loop, so x, the iteration count, the stopping rule and the
    whole history must be IDENTICAL, with early stopping, damping, a warm start, forced iterations and a single iteration."""
    J = Jets
    nrow = 9
    A, _, _, _ = make_tall_diag(J, oracle, dt, nrow, shape)
    n = int(np.prod(shape))
    hb = (u01(oracle, dt, 71, 0, nrow * n) - dt(0.5)).astype(dt)
    b = J.from_numpy(hb, J.range(A))
    x0 = J.rand(J.domain(A), seed=72, stream=0)
    cases = (dict(maxiter=25, atol=0.0, btol=0.0, conlim=0.0), dict(maxiter=60, atol=1e-4, btol=1e-4), dict(maxiter=30, atol=0.0, btol=0.0, damp=0.25),
             dict(maxiter=17, atol=0.0, btol=0.0, x0=x0), dict(maxiter=1, atol=0.0, btol=0.0), dict(maxiter=9, atol=1e-2, btol=1e-2, force_maxiter=True),
             dict(maxiter=2500, atol=0.0, btol=0.0, conlim=0.0, force_maxiter=True))
    for kw in cases:
        out = {}
        for graph in (1, 0):
            J.tune(lsqr_graph=graph)
            try:
                r = J.lsqr(A, b, **kw)
                out[graph] = (r, J.tune_get("last_lsqr_graph"))
            finally:
                J.tune(lsqr_graph=1)
        (rg, replays), (rh, zero) = out[1], out[0]
        assert zero == 0 and (replays > 0 or kw["maxiter"] == 1), "the graph path ran (and only when asked)"
        assert (rg.itn, rg.istop) == (rh.itn, rh.istop), kw
        assert_bits_equal(rg.x.to_numpy(), rh.x.to_numpy(), f"x, {kw}")
        assert rg.history == rh.history, kw
        for f in ("r1norm", "r2norm", "anorm", "acond", "arnorm", "xnorm"):
            assert getattr(rg, f) == getattr(rh, f), (f, kw)
