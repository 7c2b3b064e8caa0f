"""Tall operators whose rows are NOT whole, 16-byte aligned packs (round 5, session 3).  A block vector is one contiguous slab (src/Jets.jl:742-748), so with
an odd block length -- 101^3 elements -- most block rows start off a 16-byte boundary and end inside a pack.  Such operators run the MIXED instantiations of
the tall kernels on under-aligned packs (jh_tall.hip: tall_unaligned_ok; jh_blockop_common.h: ldu / st_pack): the same terms in the same order as the general
kernels they used to fall to, so the oracle's bits -- forward (src/Jets.jl:1015-1031), adjoint (1042-1053), the fused A'A (530-534) -- and nothing written
outside the rows."""
import numpy as np
import pytest

from .helpers import assert_bits_equal, u01
from .test_gpu_blockop import _mixed_ops

pytestmark = pytest.mark.gpu

DTYPES = [np.float32, np.float64, np.complex64, np.complex128]
# block lengths: odd, 2 mod 4, 3 mod 4; shorter than a tile (256 packs), a tile plus a bit, several tiles with a ragged last one; the smallest that holds a pack
LENGTHS = [5, 7, 67, 1025, 1026, 1027, 4099, 2 * 1024 * 4 + 1, 6 * 1024 + 3]


def _kinds(nrow, name):
    names = ["diag", "diag_adj", "identity", "scale", "zero"]
    if name == "diag":
        return [["diag"] for _ in range(nrow)]
    return [[names[(3 * i + i // 5) % 5]] for i in range(nrow)]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("n", LENGTHS)
@pytest.mark.parametrize("nrow,name", [(2, "diag"), (7, "diag"), (33, "diag"), (3, "mixed"), (18, "mixed")])
def test_tall_rows_off_the_pack_grid_have_the_oracles_bits(Jets, oracle, dt, n, nrow, name):
    J = Jets
    if n * np.dtype(dt).itemsize < 16:
        pytest.skip("less than one pack per row: the general kernels")
    kinds = _kinds(nrow, name)
    A, ops = _mixed_ops(J, oracle, dt, kinds, [n] * nrow, [n])
    hm = [u01(oracle, dt, 81, 0, n)]
    hd = [u01(oracle, dt, 82, i, n) for i in range(nrow)]
    want_d = oracle.block_df(ops, [b.copy() for b in hd], hm)           # into a DIRTY d: zero rows stay as found (1022)
    want_m = oracle.block_df_adj(ops, [np.zeros(n, dt)], want_d)
    want_y = oracle.block_df_adj(ops, [np.zeros(n, dt)], oracle.block_df(ops, [np.zeros(n, dt) for _ in range(nrow)], hm))
    try:
        for route in (1, 0):                                            # the under-aligned tall kernels / the general kernels (as before)
            J.tune(tall_unaligned=route)
            d = J.from_numpy(np.concatenate(hd), J.range(A))
            J.mul_(d, A, J.from_numpy(hm[0], J.domain(A)))
            assert_bits_equal(d.to_numpy(), np.concatenate(want_d), f"{name} {nrow} x {n} forward, tall_unaligned={route}")
            mt = J.mul_(J.rand(J.domain(A), seed=3, stream=3), A.H, d)
            assert_bits_equal(mt.to_numpy().ravel(order="F"), want_m[0], f"{name} {nrow} x {n} adjoint, tall_unaligned={route}")
    finally:
        J.tune(tall_unaligned=1)
    # the fused normal operator (the chain's bits: forward into zeros, then the adjoint)
    N = J.compose(A.H, A)
    y = J.mul_(J.rand(J.domain(A), seed=5, stream=5), N, J.from_numpy(hm[0], J.domain(A)))
    assert_bits_equal(y.to_numpy().ravel(order="F"), want_y[0], f"{name} {nrow} x {n} fused A'A")
    J.close(A)


@pytest.mark.parametrize("dt", [np.float32, np.complex64, np.float64])
def test_the_under_aligned_rows_write_nothing_outside_themselves(Jets, oracle, dt):
    """The operator's range is the MIDDLE of a longer slab (a view of blocks 1 .. nrow of nrow + 2): the blocks in front and behind keep their bits
    (the partial last pack of a row is stored scalar by scalar, never as a whole pack that would reach into the next block)."""
    import ctypes as C

    from jets_jl_amd._ffi import check, lib
    from jets_jl_amd.arrays import BlockArray

    J = Jets
    n, nrow = 3 * 1024 + 1, 9
    spc = J.JetSpace(dt, n)
    big = J.rand(J.JetBSpace([spc] * (nrow + 2)), seed=9, stream=1)
    before = big.to_numpy().copy()
    # the diagonals are blocks of ONE slab as well: their rows are off the 16-byte grid like the range vector's
    slab = J.rand(J.JetBSpace([spc] * nrow), seed=21, stream=4)
    A = J.blockop([[J.JopDiagonal(slab.arrays[i])] for i in range(nrow)])
    hs = slab.to_numpy()
    ops = [[oracle.Block("diag", n, coeff=hs[i * n:(i + 1) * n].copy())] for i in range(nrow)]
    hm = [u01(oracle, dt, 91, 0, n)]
    want_d = oracle.block_df(ops, [np.zeros(n, dt) for _ in range(nrow)], hm)
    h = C.c_void_p()
    check(lib.jh_bvec_view(big.handle, 1, nrow, C.byref(h)))
    view = BlockArray(h, [spc] * nrow, np.dtype(dt), owner=big)
    J.mul_(view, A, J.from_numpy(hm[0], J.domain(A)))
    got = big.to_numpy()
    assert_bits_equal(got[:n], before[:n], "the block in front of the rows")
    assert_bits_equal(got[(nrow + 1) * n:], before[(nrow + 1) * n:], "the block behind the rows")
    assert_bits_equal(got[n:(nrow + 1) * n], np.concatenate(want_d), "the rows")
    # and the adjoint reads exactly its rows: poison the neighbours, same result
    big[0:n] = np.full(n, np.nan, dt)
    big[(nrow + 1) * n:(nrow + 2) * n] = np.full(n, np.nan, dt)
    want_m = oracle.block_df_adj(ops, [np.zeros(n, dt)], want_d)
    mt = J.mul_(J.rand(J.domain(A), seed=3, stream=3), A.H, view)
    assert_bits_equal(mt.to_numpy().ravel(order="F"), want_m[0], "adjoint from the view")
    J.close(A)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("ncol,n", [(5, 1027), (40, 4099)])
def test_a_wide_operator_off_the_pack_grid_runs_on_its_tall_twin(Jets, oracle, dt, ncol, n):
    """1 x K of elementwise blocks: forward = the twin's ordered adjoint sum continued from what d holds (1024: no zeroing), adjoint = the twin's forward."""
    J = Jets
    names = ["diag", "diag_adj", "identity", "scale", "zero"]
    kinds = [[names[(2 * j + j // 3) % 5] for j in range(ncol)]]
    A, ops = _mixed_ops(J, oracle, dt, kinds, [n], [n] * ncol)
    hm = [u01(oracle, dt, 61, j, n) for j in range(ncol)]
    hd = [u01(oracle, dt, 62, 0, n)]
    want_d = oracle.block_df(ops, [hd[0].copy()], hm)
    want_m = oracle.block_df_adj(ops, [u01(oracle, dt, 63, j, n) for j in range(ncol)], want_d)
    try:
        for tw in (2, 0):
            J.tune(wide_twin=tw)
            d = J.from_numpy(hd[0], J.range(A))
            J.mul_(d, A, J.from_numpy(np.concatenate(hm), J.domain(A)))
            assert_bits_equal(d.to_numpy().ravel(order="F"), want_d[0], f"wide 1 x {ncol} of {n} forward, wide_twin={tw}")
            mt = J.from_numpy(np.concatenate([u01(oracle, dt, 63, j, n) for j in range(ncol)]), J.domain(A))
            J.mul_(mt, A.H, d)
            assert_bits_equal(mt.to_numpy(), np.concatenate(want_m), f"wide 1 x {ncol} of {n} adjoint, wide_twin={tw}")
    finally:
        J.tune(wide_twin=1)
    J.close(A)


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64])
def test_many_small_rows_off_the_pack_grid_take_the_split_walk_and_its_fold(Jets, oracle, dt):
    """300 rows of 1027 elements: fewer element tiles than CUs, so the adjoint sums row ranges into slabs and folds them (deterministic, tolerance parity
    like every split sum); with adj_split = 0 the ordered walk gives the oracle's bits.  The fold's last pack is partial too."""
    J = Jets
    n, nrow = 1027, 300
    kinds = [["diag"] if i % 7 else ["identity"] for i in range(nrow)]
    A, ops = _mixed_ops(J, oracle, dt, kinds, [n] * nrow, [n])
    hd = [u01(oracle, dt, 82, i, n) for i in range(nrow)]
    want_m = oracle.block_df_adj(ops, [np.zeros(n, dt)], hd)
    d = J.from_numpy(np.concatenate(hd), J.range(A))
    try:
        J.tune(adj_split=0)
        mt = J.mul_(J.rand(J.domain(A), seed=3, stream=3), A.H, d)
        assert_bits_equal(mt.to_numpy().ravel(order="F"), want_m[0], "ordered walk")
        J.tune(adj_split=-1)
        mt = J.mul_(J.rand(J.domain(A), seed=3, stream=3), A.H, d)
        got = mt.to_numpy().ravel(order="F")
        tol = 1e-5 if np.dtype(dt).itemsize in (4, 8) and np.dtype(dt) != np.float64 else 1e-12
        assert np.linalg.norm(got - want_m[0]) <= tol * np.linalg.norm(want_m[0])
        J.tune(adj_split=5)
        mt = J.mul_(J.rand(J.domain(A), seed=3, stream=3), A.H, d)
        got = mt.to_numpy().ravel(order="F")
        assert np.linalg.norm(got - want_m[0]) <= tol * np.linalg.norm(want_m[0])
    finally:
        J.tune(adj_split=-1)
    J.close(A)


# ---------------------------------------------------------------------------------- the fused solver passes on rows off the pack grid
def _native(A):
    from jets_jl_amd import jetblock

    j = A.jet
    return jetblock._native_op(j.s["_native"], j.s["ops"], j.rng.eltype())


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nrow,n,name,beta", [(2, 67, "diag", -1.375), (5, 1027, "diag", 0.5), (33, 4099, "diag", -2.0), (7, 35937, "diag", 0.0),
                                              (3, 1025, "mixed", 0.25), (18, 8193, "mixed", -0.3), (9, 1030301, "diag", -1.375), (6, 6 * 1024 + 3, "mixed", 0.0)])
def test_the_one_pass_step_and_the_forward_update_off_the_pack_grid(Jets, oracle, dt, nrow, n, name, beta):
    """jh_blockop_mul_axpby and jh_blockop_bidiag_step (whole vector) on odd block lengths: u and w BIT-EXACT against the oracle's unfused chain,
    ||u||^2 within 1e-6 / 1e-13 -- the partial last pack of a row counts the scalars it owns once."""
    import ctypes as C

    from jets_jl_amd._ffi import check, lib

    J = Jets
    if n * nrow * np.dtype(dt).itemsize > 2 ** 27:
        pytest.skip("kept small")
    A, ops = _mixed_ops(J, oracle, dt, _kinds(nrow, name), [n] * nrow, [n])
    nat = _native(A)
    alpha = 0.75
    hv, hu = u01(oracle, dt, 51, 0, n), u01(oracle, dt, 52, 0, nrow * n)
    hu_blocks = [hu[i * n:(i + 1) * n].copy() for i in range(nrow)]
    tmp = oracle.block_df(ops, [np.zeros(n, dtype=dt) for _ in range(nrow)], [hv])
    if beta != 0.0:
        ref_u = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [alpha, beta], [tmp, hu_blocks])
    else:
        ref_u = oracle.barr_lincomb([np.empty(n, dtype=dt) for _ in range(nrow)], [alpha], [tmp])
    ref_w = oracle.block_df_adj(ops, [np.full(n, 5, dtype=dt)], ref_u)
    truth = float(np.sum(np.abs(np.concatenate(ref_u).astype(np.complex128)) ** 2))
    tol = 1e-6 if np.dtype(dt) in (np.dtype(np.float32), np.dtype(np.complex64)) else 1e-13
    out = C.c_double(0)
    # the forward update alone
    v = J.from_numpy(hv, J.domain(A))
    u = J.from_numpy(hu, J.range(A))
    check(lib.jh_blockop_mul_axpby(nat.handle, u.handle, v.handle, alpha, beta, C.byref(out)))
    assert_bits_equal(u.to_numpy(), np.concatenate(ref_u), "u <- alpha*A v + beta*u (forward update)")
    assert out.value == pytest.approx(truth, rel=tol)
    # the one-pass step
    u = J.from_numpy(hu, J.range(A))
    w = J.rand(J.domain(A), seed=53, stream=0)                                        # dirty: must be overwritten
    check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, alpha, beta, C.byref(out)))
    assert_bits_equal(u.to_numpy(), np.concatenate(ref_u), "u <- alpha*A v + beta*u (one-pass step)")
    assert_bits_equal(w.to_numpy().ravel(order="F"), ref_w[0], "w <- A'u")
    assert out.value == pytest.approx(truth, rel=tol)
    J.close(A)


@pytest.mark.parametrize("dt,xtol", [(np.float32, 1e-4), (np.float64, 1e-10), (np.complex64, 1e-4)])
def test_the_solver_loops_behind_the_abi_take_operators_off_the_pack_grid(Jets, oracle, dt, xtol, monkeypatch):
    """LSQR, CGLS and CG on the normal equations on 6 rows of 17 x 17 x 15 = 4335 elements: the native loops (one pass per iteration) accept the operator
    -- jh_lsqr_solve itself is called, so a fall-back to the generic loop cannot hide -- and agree with the fp64 CPU solvers."""
    import ctypes as C

    from jets_jl_amd._ffi import check, lib
    from oracle.lsqr_ref import lsqr_fp64

    J = Jets
    nrow, shape, iters = 6, (17, 17, 15), 12
    n = int(np.prod(shape))
    spc = J.JetSpace(dt, *shape)
    slab = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=7)                       # the diagonals: blocks of ONE slab (off the grid like the range vector)
    A = J.blockop([[J.JopDiagonal(slab.arrays[i])] for i in range(nrow)])
    diags = [g.copy() for g in np.split(slab.to_numpy(), nrow)]
    dt64 = np.complex128 if np.dtype(dt).kind == "c" else np.float64
    a64 = [g.astype(dt64) for g in diags]
    matvec = lambda x: np.concatenate([g * x for g in a64])
    rmatvec = lambda y: sum(np.conj(g) * y[i * n:(i + 1) * n] for i, g in enumerate(a64))
    hb = (u01(oracle, dt, 51, 0, nrow * n) - dt(0.5)).astype(dt)
    b = J.from_numpy(hb, J.range(A))
    xr, info = lsqr_fp64(matvec, rmatvec, hb.astype(dt64), n, atol=0.0, btol=0.0, conlim=0.0, maxiter=iters)
    # the ABI's own loop, called directly: JH_OK, not JH_ERR_UNSUPPORTED
    nat = _native(A)
    u = J.from_numpy(hb, J.range(A))
    x = J.zeros(J.domain(A))
    from jets_jl_amd._ffi import LsqrResultC as _R
    res = _R()
    check(lib.jh_lsqr_solve(nat.handle, u.handle, x.handle, 0, 0.0, 0.0, 0.0, 0.0, iters, 0, C.byref(res), None))
    assert res.itn == iters
    got = x.to_numpy().ravel(order="F").astype(dt64)
    assert np.linalg.norm(got - xr) / np.linalg.norm(xr) < xtol
    # and through the drivers (native by default)
    monkeypatch.setenv("JETS_LSQR_NATIVE", "1")
    r = J.lsqr(A, b, atol=0.0, btol=0.0, conlim=0.0, maxiter=iters)
    assert np.linalg.norm(r.x.to_numpy().ravel(order="F").astype(dt64) - xr) / np.linalg.norm(xr) < xtol
    a = np.stack(a64)
    x_ls = (np.conj(a) * hb.astype(dt64).reshape(nrow, n)).sum(0) / (np.abs(a) ** 2).sum(0)
    for solver in (J.cgls, J.cgnr):
        rc = solver(A, b, maxiter=60, atol=0.0, btol=0.0)
        xc = rc.x.to_numpy().ravel(order="F").astype(dt64)
        assert np.linalg.norm(xc - x_ls) / np.linalg.norm(x_ls) < max(10 * xtol, 1e-6), solver.__name__
    J.close(A)


# ---------------------------------------------------------------------------------- M x K grids off the pack grid (the general kernels)
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 2), (3, 4), (9, 7), (16, 16), (5, 33)])
@pytest.mark.parametrize("n", [5, 67, 1027, 2 * 1024 * 4 + 1])
def test_grids_of_equal_odd_blocks_have_the_oracles_bits_on_every_route(Jets, oracle, dt, shape, n):
    """Equal blocks of an odd length: the register-tiled general kernel (2 / 4 lines per workgroup, step lists) and the one-line kernels, all on
    under-aligned packs, against the 4-byte-per-lane kernels (tall_unaligned = 0) and the oracle: forward into a dirty d (1024), adjoint (1042-1053)."""
    J = Jets
    if n * np.dtype(dt).itemsize < 16:
        pytest.skip("less than one pack per block: the 4-byte-per-lane kernels")
    M, K = shape
    names = ["diag", "diag_adj", "identity", "scale", "zero", "diag", "zero"]
    kinds = [[names[(3 * i + 5 * j + (i * j) // 3) % 7] for j in range(K)] for i in range(M)]
    A, ops = _mixed_ops(J, oracle, dt, kinds, [n] * M, [n] * K)
    hm = [u01(oracle, dt, 31, j, n) for j in range(K)]
    hd = [u01(oracle, dt, 32, i, n) for i in range(M)]
    hmt = [u01(oracle, dt, 33, j, n) for j in range(K)]
    want_d = oracle.block_df(ops, [b.copy() for b in hd], hm)
    want_m = oracle.block_df_adj(ops, [b.copy() for b in hmt], want_d)
    try:
        for knobs in (dict(), dict(general_tile=2), dict(general_tile=4), dict(general_tile=0), dict(general_list=2), dict(general_list=3), dict(tall_unaligned=0)):
            J.tune(general_tile=1, general_list=1, tall_unaligned=1)
            J.tune(**knobs)
            d = J.from_numpy(np.concatenate(hd), J.range(A))
            J.mul_(d, A, J.from_numpy(np.concatenate(hm), J.domain(A)))
            assert_bits_equal(d.to_numpy(), np.concatenate(want_d), f"{M} x {K} of {n} forward, {knobs}")
            mt = J.from_numpy(np.concatenate(hmt), J.domain(A))
            J.mul_(mt, A.H, d)
            assert_bits_equal(mt.to_numpy(), np.concatenate(want_m), f"{M} x {K} of {n} adjoint, {knobs}")
    finally:
        J.tune(general_tile=1, general_list=1, tall_unaligned=1)
    J.close(A)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("pattern", ["blockdiag", "bidiag", "full"])
def test_ragged_grids_of_odd_blocks(Jets, oracle, dt, pattern):
    """Block lengths that differ from line to line and are odd: the one-line 16-byte kernels on under-aligned packs (every line has its own last, partial
    pack), with and without the step lists."""
    J = Jets
    lens = [1027, 67, 4099, 5 if np.dtype(dt).itemsize >= 8 else 9, 2051, 1027, 333, 4099, 129]
    M = len(lens)
    names = ["diag", "diag_adj", "identity", "scale"]
    on = {"blockdiag": lambda i, j: i == j, "bidiag": lambda i, j: i == j or i == j + 1, "full": lambda i, j: True}[pattern]
    # elementwise blocks are square: off-diagonal blocks of a ragged grid can only be zero blocks unless the two lengths agree
    kinds = [[(names[(i + 2 * j) % 4] if on(i, j) and lens[i] == lens[j] else "zero") for j in range(M)] for i in range(M)]
    A, ops = _mixed_ops(J, oracle, dt, kinds, lens, lens)
    hm = [u01(oracle, dt, 31, j, lens[j]) for j in range(M)]
    hd = [u01(oracle, dt, 32, i, lens[i]) for i in range(M)]
    want_d = oracle.block_df(ops, [b.copy() for b in hd], hm)
    want_m = oracle.block_df_adj(ops, [np.zeros(lens[j], dt) for j in range(M)], want_d)
    try:
        for knobs in (dict(), dict(general_list=0), dict(tall_unaligned=0)):
            J.tune(general_list=1, tall_unaligned=1)
            J.tune(**knobs)
            d = J.from_numpy(np.concatenate(hd), J.range(A))
            J.mul_(d, A, J.from_numpy(np.concatenate(hm), J.domain(A)))
            assert_bits_equal(d.to_numpy(), np.concatenate(want_d), f"ragged {pattern} forward, {knobs}")
            mt = J.rand(J.domain(A), seed=4, stream=4)
            J.mul_(mt, A.H, d)
            assert_bits_equal(mt.to_numpy(), np.concatenate(want_m), f"ragged {pattern} adjoint, {knobs}")
    finally:
        J.tune(general_list=1, tall_unaligned=1)
    J.close(A)


# ---------------------------------------------------------------------------------- JIT broadcast off the pack grid
@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("n", [5, 7, 1027, 4099, 2 * 1024 * 4 + 1])
def test_broadcast_on_odd_lengths_and_views_off_the_grid_is_bit_exact(Jets, oracle, dt, n):
    """The 16-byte-per-lane JIT kernel on under-aligned packs with a partial last pack (src/Jets.jl:889-911): whole vectors of odd length, blocks that start
    off the 16-byte grid, in place (dst aliases an operand), and many such items in ONE launch -- numpy's bits (every operation rounded as written)."""
    J = Jets
    T = np.dtype(dt).type
    count = 9
    R = J.JetBSpace([J.JetSpace(dt, n)] * count)
    u, v = J.rand(R, seed=1, stream=0), J.rand(R, seed=2, stream=0)
    hu, hv = u01(oracle, dt, 1, 0, count * n), u01(oracle, dt, 2, 0, count * n)
    a, b = T(0.37), T(-1.25)
    x = J.zeros(R)
    J.broadcast_(x, "s0*x0 + s1*x1*x0", [u, v], [a, b])
    want = a * hu + (b * hv) * hu
    assert_bits_equal(x.to_numpy(), want, "whole vector")
    # block by block (every block but the first few starts off the grid), into a dirty destination whose neighbours must not change
    y = J.rand(R, seed=3, stream=0)
    hy = u01(oracle, dt, 3, 0, count * n)
    for k in (1, 2, 3, 6):
        J.broadcast_(y.arrays[k], "s0*x0 + s1*x1*x0", [u.arrays[k], v.arrays[k]], [a, b])
        hy[k * n:(k + 1) * n] = want[k * n:(k + 1) * n]
    assert_bits_equal(y.to_numpy(), hy, "single blocks: their bits, and nothing outside them")
    # in place
    z = J.rand(R, seed=4, stream=0)
    hz = u01(oracle, dt, 4, 0, count * n)
    J.broadcast_(z, "x0*x0 + x1", [z, u], [])
    assert_bits_equal(z.to_numpy(), hz * hz + hu, "in place, whole vector")
    J.broadcast_(z.arrays[5], "x0 + x0", [z.arrays[5]], [])
    hz = hz * hz + hu
    hz[5 * n:6 * n] = hz[5 * n:6 * n] + hz[5 * n:6 * n]
    assert_bits_equal(z.to_numpy(), hz, "in place, one block off the grid")
    # many items in one launch
    one, many = J.zeros(R), J.zeros(R)
    for k in range(count):
        J.broadcast_(one.arrays[k], "s0*x0*x1 + s1", [u.arrays[k], v.arrays[k]], [0.5 + k, -0.25 * k])
    J.broadcast_many_((many.arrays[k], "s0*x0*x1 + s1", [u.arrays[k], v.arrays[k]], [0.5 + k, -0.25 * k]) for k in range(count))
    assert many.to_numpy().tobytes() == one.to_numpy().tobytes()
    if n * np.dtype(dt).itemsize >= 16:
        assert_bits_equal(one.to_numpy()[n:2 * n], T(1.5) * hu[n:2 * n] * hv[n:2 * n] + T(-0.25), "item 1 of the batch")


# ---------------------------------------------------------------------------------- fused JetSum off the pack grid
@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64])
@pytest.mark.parametrize("nterms", [2, 4, 7, 11, 16, 19])
@pytest.mark.parametrize("n", [1027, 4099])
def test_the_fused_jetsum_takes_odd_blocks_with_the_chains_bits(Jets, oracle, dt, nterms, n):
    """JetSum of tall diagonal operators of an odd block length (src/Jets.jl:628-655): the fused kernels (four / eight / twelve / sixteen streams per
    launch, later launches continuing the sum) on under-aligned packs -- jh_blocksum_mul_typed itself must accept the operators -- bit-identical to
    the unfused chain d .= 0; d = d +- s_t (A_t m); same for the adjoint."""
    import ctypes as C

    from jets_jl_amd._ffi import check, lib

    J = Jets
    nrow = 5
    spc = J.JetSpace(dt, n)
    ops, hcoef = [], []
    for t in range(nterms):
        slab = J.rand(J.JetBSpace([spc] * nrow), seed=90 + t, stream=0)                 # one slab per term: rows off the grid, one stride (STRIDED addressing)
        if t % 3 == 2:                                                                # ... and every third term from separate arrays (the block TABLES)
            arrs = [J.rand(spc, seed=90 + t, stream=100 + i) for i in range(nrow)]
            hcoef.append([u01(oracle, dt, 90 + t, 100 + i, n) for i in range(nrow)])
        else:
            arrs = slab.arrays
            hcoef.append([g.copy() for g in np.split(slab.to_numpy(), nrow)])
        ops.append(J.blockop([[J.JopDiagonal(a)] for a in arrs]))
    scales = [1.0 if t % 3 else 0.5 + t for t in range(nterms)]
    signs = [1.0 if t % 2 == 0 else -1.0 for t in range(nterms)]
    hm = u01(oracle, dt, 2, 0, n)
    hd = [u01(oracle, dt, 3, i, n) for i in range(nrow)]
    want = [np.zeros(n, dt) for _ in range(nrow)]
    want_m = [np.zeros(n, dt)]
    for t in range(nterms):
        ora = [[oracle.Block("diag", n, coeff=c)] for c in hcoef[t]]
        tmp = oracle.block_df(ora, [np.zeros(n, dt) for _ in range(nrow)], [hm])
        if scales[t] != 1.0:
            tmp = oracle.barr_lincomb([np.empty(n, dt) for _ in range(nrow)], [scales[t]], [tmp])
        want = oracle.barr_lincomb([np.empty(n, dt) for _ in range(nrow)], [1.0, signs[t]], [want, tmp])
        din = hd if scales[t] == 1.0 else oracle.barr_lincomb([np.empty(n, dt) for _ in range(nrow)], [scales[t]], [hd])
        tm = oracle.block_df_adj(ora, [np.zeros(n, dt)], din)
        want_m = oracle.barr_lincomb([np.empty(n, dt)], [1.0, signs[t]], [want_m, tm])
    # the ABI call itself: JH_OK, not JH_ERR_UNSUPPORTED
    nats = [_native(A) for A in ops]
    hs = (C.c_void_p * nterms)(*[x.handle for x in nats])
    sc = (C.c_double * nterms)(*scales)
    fg = (C.c_int32 * nterms)(*([0] * nterms))
    sg = (C.c_double * nterms)(*signs)
    m = J.from_numpy(hm, J.domain(ops[0]))
    d = J.rand(J.range(ops[0]), seed=7, stream=7)
    check(lib.jh_blocksum_mul_typed(nterms, hs, sc, fg, sg, d.handle, m.handle))
    assert_bits_equal(d.to_numpy(), np.concatenate(want), f"{nterms}-term sum of {n}-element rows, forward")
    mt = J.rand(J.domain(ops[0]), seed=8, stream=8)
    din = J.from_numpy(np.concatenate(hd), J.range(ops[0]))                           # (kept alive across the call: a temporary would be destroyed first)
    check(lib.jh_blocksum_mul_adj_typed(nterms, hs, sc, fg, sg, mt.handle, din.handle))
    assert_bits_equal(mt.to_numpy().ravel(order="F"), want_m[0], f"{nterms}-term sum, adjoint")
    # and through the operator algebra
    S = scales[0] * ops[0] if scales[0] != 1.0 else ops[0]
    for t in range(1, nterms):
        term = scales[t] * ops[t] if scales[t] != 1.0 else ops[t]
        S = S + term if signs[t] > 0 else S - term
    assert_bits_equal((S * m).to_numpy(), np.concatenate(want), "A1 +- s2*A2 ... forward")


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_the_fused_normal_operator_on_big_odd_blocks(Jets, oracle, dt):
    """Rows of >= 64 MiB select the fat shape of the fused A'A (1024 lanes x 4 packs): off the grid it runs two rows in flight with the partial-pack logic,
    on the grid four rows without (k_tall_diag_adj's TAIL) -- the chain's bits both ways."""
    J = Jets
    nrow = 3
    for n in ((1 << 24) + 1, 1 << 24) if np.dtype(dt).itemsize == 4 else ((1 << 23) + 1, 1 << 23):
        kinds = [["diag"], ["identity"], ["diag_adj"]]
        A, ops = _mixed_ops(J, oracle, dt, kinds, [n] * nrow, [n])
        hm = [u01(oracle, dt, 81, 0, n)]
        want_y = oracle.block_df_adj(ops, [np.zeros(n, dt)], oracle.block_df(ops, [np.zeros(n, dt) for _ in range(nrow)], hm))
        y = J.mul_(J.rand(J.domain(A), seed=5, stream=5), J.compose(A.H, A), J.from_numpy(hm[0], J.domain(A)))
        assert_bits_equal(y.to_numpy().ravel(order="F"), want_y[0], f"fused A'A, rows of {n}")
        J.close(A)


# ---------------------------------------------------------------------------------- dense children of odd dimensions: the adjoint's column kernels
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("layout", ["tall", "wide", "blockdiag", "single"])
@pytest.mark.parametrize("nch,nr,nc", [(3, 1023, 515), (2, 2047, 1021), (5, 67, 33), (4, 5, 9)])
def test_dense_children_of_odd_dimensions_adjoint_on_under_aligned_packs(Jets, oracle, dt, layout, nch, nr, nc):
    """k x l children with k odd: every column of the matrix starts off the 16-byte grid and ends inside a pack.  The adjoint's column kernels (single,
    batched with row chunks, list) sum under-aligned packs, the last one of a column (of a chunk) from its own first scalar on: within the dense kernels'
    tolerance of an extended-precision host product, identical on a second run, and equal to the 4-byte-per-lane kernels' result within the same tolerance."""
    J = Jets
    if layout == "single":
        nch = 1
    wide_t = np.clongdouble if np.dtype(dt).kind == "c" else np.longdouble
    tol = 2e-5 if np.dtype(dt).itemsize // (2 if np.dtype(dt).kind == "c" else 1) == 4 else 1e-12
    mats = [np.asfortranarray(u01(oracle, dt, 11, z, nr * nc).reshape((nr, nc), order="F")) for z in range(nch)]
    dev = [J.JopDense(J.from_numpy(hA)) for hA in mats]
    if layout in ("tall", "single"):
        A = J.blockop([[op] for op in dev])
        hd = u01(oracle, dt, 3, 0, nch * nr).reshape(nch, nr)
        truth = sum(np.conj(mats[z].astype(wide_t)).T @ hd[z].astype(wide_t) for z in range(nch))
    elif layout == "wide":
        A = J.blockop([dev])
        hd = u01(oracle, dt, 3, 0, nr).reshape(1, nr)
        truth = np.concatenate([np.conj(mats[z].astype(wide_t)).T @ hd[0].astype(wide_t) for z in range(nch)])
    else:
        A = J.blockop([[dev[i] if i == j else J.JopZeroBlock(J.JetSpace(dt, nc), J.JetSpace(dt, nr)) for j in range(nch)] for i in range(nch)])
        hd = u01(oracle, dt, 3, 0, nch * nr).reshape(nch, nr)
        truth = np.concatenate([np.conj(mats[z].astype(wide_t)).T @ hd[z].astype(wide_t) for z in range(nch)])
    d = J.from_numpy(hd.ravel(), J.range(A))
    err = lambda a, b: float(np.linalg.norm(a.astype(wide_t) - b) / np.linalg.norm(b))
    got = {}
    try:
        for knob in (1, 0):
            J.tune(tall_unaligned=knob)
            mt = J.rand(J.domain(A), seed=6, stream=6)                                   # dirty: zeroed first (1042)
            J.mul_(mt, A.H, d)
            got[knob] = mt.to_numpy().ravel(order="F").copy()
            assert err(got[knob], truth) < tol, f"{layout} adjoint, tall_unaligned={knob}"
            again = J.zeros(J.domain(A))
            J.mul_(again, A.H, d)
            assert_bits_equal(again.to_numpy().ravel(order="F"), got[knob], "second run")
    finally:
        J.tune(tall_unaligned=1)
    m = J.rand(J.domain(A), seed=2, stream=0)
    lhs, rhs = J.dot_product_test(A, m, d)
    assert abs(lhs - rhs) / abs(lhs + rhs) < (1e-5 if tol > 1e-10 else 1e-12)
    J.close(A)


# ---------------------------------------------------------------------------------- the ranged calls (the multi-GPU exchange's pipelining) off the pack grid
@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64])
@pytest.mark.parametrize("name", ["diag", "mixed"])
@pytest.mark.parametrize("n", [69615, 49153, 16385 + 2])
def test_ranged_calls_on_rows_off_the_pack_grid_equal_the_whole_ones(Jets, oracle, dt, name, n):
    """jh_blockop_mul_adj_range / _normal_mul_range / _bidiag_step_range over the exchange's ranges (16-byte aligned bounds in the DOMAIN; the last range ends
    with the vector -- inside a pack, and may be a single element) == the whole-vector calls, bit for bit; what a row-partitioned solver runs per chunk."""
    import ctypes as C

    from jets_jl_amd._ffi import check, lib

    J = Jets
    nrow = 5
    A, ops = _mixed_ops(J, oracle, dt, _kinds(nrow, name), [n] * nrow, [n])
    nat = _native(A)
    step = 16384
    ranges = [(lo, min(step, n - lo)) for lo in range(0, n, step)]
    d = J.rand(J.range(A), seed=81, stream=0)
    m = J.rand(J.domain(A), seed=80, stream=0)
    whole = J.mul_(J.zeros(J.domain(A)), A.H, d)
    parts = J.rand(J.domain(A), seed=82, stream=0)                                    # dirty: every element must be overwritten
    for lo, cnt in ranges:
        check(lib.jh_blockop_mul_adj_range(nat.handle, parts.handle, d.handle, lo, cnt))
    assert_bits_equal(parts.to_numpy(), whole.to_numpy(), f"ranged adjoint {ranges[-1]}")
    yw = J.mul_(J.zeros(J.domain(A)), J.compose(A.H, A), m)
    yp = J.rand(J.domain(A), seed=83, stream=0)
    for lo, cnt in ranges:
        check(lib.jh_blockop_normal_mul_range(nat.handle, yp.handle, m.handle, lo, cnt))
    assert_bits_equal(yp.to_numpy(), yw.to_numpy(), "ranged A'A")
    out = C.c_double(0)
    u1, u2 = J.rand(J.range(A), seed=84, stream=0), J.rand(J.range(A), seed=84, stream=0)
    w1, w2 = J.zeros(J.domain(A)), J.rand(J.domain(A), seed=85, stream=0)
    check(lib.jh_blockop_bidiag_step(nat.handle, u1.handle, m.handle, w1.handle, 0.75, -1.375, C.byref(out)))
    total_whole = out.value
    total = 0.0
    for lo, cnt in ranges:
        check(lib.jh_blockop_bidiag_step_range(nat.handle, u2.handle, m.handle, w2.handle, 0.75, -1.375, lo, cnt, C.byref(out)))
        total += out.value
    assert_bits_equal(u2.to_numpy(), u1.to_numpy(), "u: ranges == whole")
    assert_bits_equal(w2.to_numpy(), w1.to_numpy(), "w: ranges == whole")
    assert total == pytest.approx(total_whole, rel=1e-12)
    if np.dtype(dt).itemsize < 16 and (n * np.dtype(dt).itemsize) % 16:
        with pytest.raises(J.JetsHipError):                                           # a MIDDLE range must still end on a 16-byte bound
            check(lib.jh_blockop_mul_adj_range(nat.handle, parts.handle, d.handle, 0, 16385 if np.dtype(dt).itemsize == 4 else 16383))
    J.close(A)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nrow,n", [(2, 67), (5, 1027), (33, 4099), (7, 35937), (3, 2 * 1024 * 4 * 4 + 1)])
def test_the_fused_adjoint_update_off_the_pack_grid(Jets, oracle, dt, nrow, n):
    """jh_blockop_mul_adj_axpby (m <- alpha A'(gamma d) + beta m, ||m||^2) and (a * A)' d = jh_blockop_mul_adj_scaled on odd block lengths: bit-exact
    against the unfused chain, the norm counted once per scalar."""
    import ctypes as C

    from jets_jl_amd._ffi import check, lib

    J = Jets
    A, ops = _mixed_ops(J, oracle, dt, _kinds(nrow, "diag"), [n] * nrow, [n])
    nat = _native(A)
    alpha, beta, gamma = 0.75, -1.375, 0.5
    hm = u01(oracle, dt, 41, 0, n)
    hd = [u01(oracle, dt, 42, i, n) for i in range(nrow)]
    m = J.from_numpy(hm, J.domain(A))
    d = J.from_numpy(np.concatenate(hd), J.range(A))
    out = C.c_double(0)
    check(lib.jh_blockop_mul_adj_axpby(nat.handle, m.handle, d.handle, alpha, beta, gamma, C.byref(out)))
    din = oracle.barr_lincomb([np.empty(n, dt) for _ in range(nrow)], [gamma], [hd])
    tmpm = oracle.block_df_adj(ops, [np.zeros(n, dtype=dt)], din)
    refm = oracle.barr_lincomb([np.empty(n, dtype=dt)], [alpha, beta], [tmpm, [hm]])
    assert_bits_equal(m.to_numpy().ravel(order="F"), refm[0], "m <- alpha*A'(gamma d) + beta*m")
    tol = 1e-6 if np.dtype(dt) in (np.dtype(np.float32), np.dtype(np.complex64)) else 1e-13
    assert out.value == pytest.approx(float(np.sum(np.abs(refm[0].astype(np.complex128)) ** 2)), rel=tol)
    mt = J.rand(J.domain(A), seed=9, stream=9)
    check(lib.jh_blockop_mul_adj_scaled(nat.handle, mt.handle, d.handle, gamma, 0))
    assert_bits_equal(mt.to_numpy().ravel(order="F"), tmpm[0], "(a * A)' d")
    J.close(A)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 2), (4, 3), (9, 16), (33, 5)])
@pytest.mark.parametrize("n", [7, 1027, 2 * 1024 * 4 + 1])
def test_grids_of_plain_diagonals_of_odd_blocks_on_the_all_diagonal_kernel(Jets, oracle, dt, shape, n):
    """M x K grids whose blocks are ALL plain diagonals of an odd length: k_grid_tile (2 / 4 / 8 lines per workgroup) on under-aligned packs, against the
    general register-tiled kernel (grid_tile = 0 ... grid_diag = 0) and the oracle: forward into a dirty d (1024), adjoint (1042-1053), bit for bit."""
    J = Jets
    if n * np.dtype(dt).itemsize < 16:
        pytest.skip("less than one pack per block")
    M, K = shape
    A, ops = _mixed_ops(J, oracle, dt, [["diag"] * K for _ in range(M)], [n] * M, [n] * K)
    hm = [u01(oracle, dt, 31, j, n) for j in range(K)]
    hd = [u01(oracle, dt, 32, i, n) for i in range(M)]
    hmt = [u01(oracle, dt, 33, j, n) for j in range(K)]
    want_d = oracle.block_df(ops, [b.copy() for b in hd], hm)
    want_m = oracle.block_df_adj(ops, [b.copy() for b in hmt], want_d)
    try:
        for knobs in (dict(), dict(grid_tile=2), dict(grid_tile=4), dict(grid_tile=8), dict(grid_tile=0), dict(grid_diag=0), dict(tall_unaligned=0)):
            J.tune(grid_tile=1, grid_diag=1, tall_unaligned=1)
            J.tune(**knobs)
            d = J.from_numpy(np.concatenate(hd), J.range(A))
            J.mul_(d, A, J.from_numpy(np.concatenate(hm), J.domain(A)))
            assert_bits_equal(d.to_numpy(), np.concatenate(want_d), f"{M} x {K} of {n} forward, {knobs}")
            mt = J.from_numpy(np.concatenate(hmt), J.domain(A))
            J.mul_(mt, A.H, d)
            assert_bits_equal(mt.to_numpy(), np.concatenate(want_m), f"{M} x {K} of {n} adjoint, {knobs}")
    finally:
        J.tune(grid_tile=1, grid_diag=1, tall_unaligned=1)
    J.close(A)


# ---------------------------------------------------------------------------------- F(m) of a tall nonlinear operator on the tall tiling
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("n", [64, 4100, 1027, 2 * 1024 * 4 + 1])
@pytest.mark.parametrize("nrow", [2, 9, 40])
def test_f_of_a_tall_nonlinear_operator_on_the_tall_tiling(Jets, oracle, dt, n, nrow):
    """JetBlock_f! (src/Jets.jl:988-1008) of a tall operator whose children are SQUARE (the reference's JopBar), diagonals, identity / scalar rows and zero
    blocks: every row is WRITTEN (a zero block's zeros included -- nothing is skipped in f!, 1003), SQUARE children square the model; the tall tiling
    (tall_f = 1, aligned and odd block lengths) and the general kernels (tall_f = 0) against the oracle's block_f, bit for bit, into a dirty d."""
    J = Jets
    if n * np.dtype(dt).itemsize < 16:
        pytest.skip("less than one pack per row")
    spc = J.JetSpace(dt, n)
    hm = u01(oracle, dt, 11, 0, n)
    names = ["square", "diag", "zero", "identity", "square", "scale", "diag_adj"]
    rows, ops = [], []
    for i in range(nrow):
        k = names[(2 * i + i // 3) % 7] if i else "square"
        if k == "square":
            rows.append([J.JopSquare(spc)]); ops.append([oracle.Block("square", n, coeff=hm)])
        elif k == "zero":
            rows.append([J.JopZeroBlock(spc, spc)]); ops.append([oracle.Block("zero", n, n)])
        elif k == "identity":
            rows.append([J.JopIdentity(spc)]); ops.append([oracle.Block("identity", n)])
        elif k == "scale":
            a = 0.3 + i
            rows.append([J.JopLn(dom=spc, rng=spc, df=J.constdiag_df, df_adj=J.constdiag_df_adj, s={"a": a})]); ops.append([oracle.Block("scale", n, scale=a)])
        else:
            g = J.rand(spc, seed=21, stream=i)
            op = J.JopDiagonal(g)
            rows.append([op.H if k == "diag_adj" else op]); ops.append([oracle.Block("diag", n, coeff=u01(oracle, dt, 21, i, n), adjoint=(k == "diag_adj"))])
    F = J.blockop(rows)
    assert isinstance(F, J.JopNl)
    hd = [u01(oracle, dt, 32, i, n) for i in range(nrow)]
    want = oracle.block_f(ops, [b.copy() for b in hd], [hm])
    try:
        for knob in (1, 0):
            J.tune(tall_f=knob)
            d = J.from_numpy(np.concatenate(hd), J.range(F))
            J.mul_(d, F, J.from_numpy(hm, J.domain(F)))
            assert_bits_equal(d.to_numpy(), np.concatenate(want), f"F(m), {nrow} rows of {n}, tall_f={knob}")
    finally:
        J.tune(tall_f=1)
    J.close(F)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nrow,n", [(3, 1024), (18, 4100), (5, 1027), (300, 67)])
def test_scalar_times_a_tall_operator_of_several_kinds_adjoint_in_one_pass(Jets, oracle, dt, nrow, n):
    """(a * A)' d = A'(conj(a) d) (src/Jets.jl:1160) for a tall operator with rows of several kinds (and / or off the pack grid): the MIXED adjoint scales d_i on
    the way in -- jh_blockop_mul_adj_scaled itself accepts the operator -- with the bits of the chain tmp .= a * d; A' tmp."""
    import ctypes as C

    from jets_jl_amd._ffi import check, lib

    J = Jets
    if n * np.dtype(dt).itemsize < 16:
        pytest.skip("less than one pack per row")
    A, ops = _mixed_ops(J, oracle, dt, _kinds(nrow, "mixed"), [n] * nrow, [n])
    nat = _native(A)
    a = 0.625
    hd = [u01(oracle, dt, 42, i, n) for i in range(nrow)]
    d = J.from_numpy(np.concatenate(hd), J.range(A))
    din = oracle.barr_lincomb([np.empty(n, dt) for _ in range(nrow)], [a], [hd])
    want = oracle.block_df_adj(ops, [np.zeros(n, dtype=dt)], din)[0]
    try:
        J.tune(adj_split=0)                                                           # the ordered walk: the chain's bits (the split walk: tolerance, below)
        mt = J.rand(J.domain(A), seed=9, stream=9)
        check(lib.jh_blockop_mul_adj_scaled(nat.handle, mt.handle, d.handle, a, 0))
        assert_bits_equal(mt.to_numpy().ravel(order="F"), want, "(a * A)' d, one pass")
        T = np.dtype(dt).type(a) if np.dtype(dt).kind != "c" else np.zeros(1, dt).real.dtype.type(a)
        got = ((T * A).H * d).to_numpy().ravel(order="F")
        assert_bits_equal(got, want, "through the operator algebra")
    finally:
        J.tune(adj_split=-1)
    mt2 = J.rand(J.domain(A), seed=9, stream=9)
    check(lib.jh_blockop_mul_adj_scaled(nat.handle, mt2.handle, d.handle, a, 0))
    tol = 1e-5 if np.dtype(dt).itemsize // (2 if np.dtype(dt).kind == "c" else 1) == 4 else 1e-12
    assert np.linalg.norm(mt2.to_numpy().ravel(order="F") - want) <= tol * np.linalg.norm(want)
    # the plain adjoint afterwards is unscaled again
    plain = J.mul_(J.zeros(J.domain(A)), A.H, d)
    ref = oracle.block_df_adj(ops, [np.zeros(n, dtype=dt)], hd)[0]
    assert np.linalg.norm(plain.to_numpy().ravel(order="F") - ref) <= tol * np.linalg.norm(ref)
    J.close(A)


# ---------------------------------------------------------------------------------- round 6: the forward on lanes anchored to each row's own 16-byte grid
@pytest.fixture()
def anchored(Jets):
    Jets.tune(fwd_anchor=1)                       # every eligible operator (the default rule starts at rows of 64 KiB)
    yield
    Jets.tune(fwd_anchor=-1)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("n", [9, 67, 1025, 1026, 1027, 4099, 2 * 1024 * 4 + 1, 6 * 1024 + 3, 65537])
@pytest.mark.parametrize("nrow,name", [(2, "diag"), (7, "diag"), (33, "diag"), (3, "mixed"), (18, "onezero")])
def test_the_anchored_forward_has_the_oracles_bits(Jets, oracle, anchored, dt, n, nrow, name):
    """k_tall_fwd_anchored (round 6): a lane owns an ALIGNED 16-byte slot of the range slab -- elements [p NS - ph, p NS - ph + NS) of its row, ph the row's
    phase -- instead of the element-indexed pack p; a row's first and last slot are partial and go element by element.  The same products: the oracle's bits
    (src/Jets.jl:1015-1031) for all-diagonal rows and rows of several kinds (zero rows stay as found, 1022), the operator's range in the MIDDLE of a longer
    slab (a view at an odd offset: every phase class occurs), the blocks in front and behind untouched; diagonals in one slab (same phase as the range) and
    in separate arrays; and F(m) of the nonlinear mode (every row written)."""
    import ctypes as C

    from jets_jl_amd._ffi import check, lib
    from jets_jl_amd.arrays import BlockArray

    J = Jets
    if n * np.dtype(dt).itemsize < 32:
        pytest.skip("fewer than two packs per row: the element-indexed kernel")
    if (n * np.dtype(dt).itemsize) % 16 == 0:
        pytest.skip("rows of whole packs (every ComplexF64 row is): nothing to anchor")
    # "onezero": rows of several kinds with ONE zero block (an operator with an eighth or more zero rows walks the list of its non-zero rows instead)
    kinds = [[["diag", "diag_adj", "identity", "scale"][i % 4] if i != 5 else "zero"] for i in range(nrow)] if name == "onezero" else _kinds(nrow, name)
    A, ops = _mixed_ops(J, oracle, dt, kinds, [n] * nrow, [n])
    spc = J.JetSpace(dt, n)
    hm = [u01(oracle, dt, 81, 0, n)]
    for lead in (1, 2):                                                          # blocks in front of the operator's rows: shifts every row's phase
        big = J.rand(J.JetBSpace([spc] * (nrow + lead + 1)), seed=9, stream=lead)
        before = big.to_numpy().copy()
        hd = [before[(lead + i) * n:(lead + i + 1) * n].copy() for i in range(nrow)]
        want = oracle.block_df(ops, [b.copy() for b in hd], hm)                  # into the DIRTY rows: zero rows stay as found
        h = C.c_void_p()
        check(lib.jh_bvec_view(big.handle, lead, nrow, C.byref(h)))
        view = BlockArray(h, [spc] * nrow, np.dtype(dt), owner=big)
        J.mul_(view, A, J.from_numpy(hm[0], J.domain(A)))
        assert J.tune_get("last_fwd_walk") == 3, "the anchored kernel should have run"
        got = big.to_numpy()
        assert_bits_equal(got[:lead * n], before[:lead * n], "the blocks in front of the rows")
        assert_bits_equal(got[(lead + nrow) * n:], before[(lead + nrow) * n:], "the block behind the rows")
        assert_bits_equal(got[lead * n:(lead + nrow) * n], np.concatenate(want), f"{name} {nrow} x {n} anchored forward, {lead} blocks in front")
    J.close(A)


@pytest.mark.parametrize("dt", [np.float32, np.complex64, np.float64])
def test_the_anchored_forward_with_diagonals_in_one_slab_and_in_f_mode(Jets, oracle, anchored, dt):
    J = Jets
    n, nrow = 4099, 11
    spc = J.JetSpace(dt, n)
    slab = J.rand(J.JetBSpace([spc] * nrow), seed=21, stream=4)                  # the diagonals share the range vector's layout: aligned loads
    A = J.blockop([[J.JopDiagonal(slab.arrays[i])] for i in range(nrow)])
    hs = slab.to_numpy()
    ops = [[oracle.Block("diag", n, coeff=hs[i * n:(i + 1) * n].copy())] for i in range(nrow)]
    hm = [u01(oracle, dt, 91, 0, n)]
    want = oracle.block_df(ops, [np.zeros(n, dt) for _ in range(nrow)], hm)
    d = J.mul_(J.rand(J.range(A), seed=5, stream=5), A, J.from_numpy(hm[0], J.domain(A)))
    assert J.tune_get("last_fwd_walk") == 3
    assert_bits_equal(d.to_numpy(), np.concatenate(want), "diagonals in one slab")
    # F(m) of a tall nonlinear operator (SQUARE children between diagonal ones): every row written (src/Jets.jl:1003)
    F = J.blockop([[J.JopSquare(spc)] if i % 3 == 0 else [J.JopDiagonal(slab.arrays[i])] for i in range(nrow)])
    fo = [[oracle.Block("square", n)] if i % 3 == 0 else [oracle.Block("diag", n, coeff=hs[i * n:(i + 1) * n].copy())] for i in range(nrow)]
    wantf = oracle.block_f(fo, [np.zeros(n, dt) for _ in range(nrow)], hm)
    df = J.mul_(J.rand(J.range(F), seed=6, stream=6), F, J.from_numpy(hm[0], J.domain(F)))
    assert_bits_equal(df.to_numpy(), np.concatenate(wantf), "F(m) on anchored lanes")
    J.close(A)
    J.close(F)
