"""GPU: hipGraph replay of the launch-bound per-block loop (operators with DENSE children; jh_blockop.hip: run_loop_graphed).
Call 1 with a given (output, input) pair is eager, call 2 is captured, calls 3+ replay the graph with one launch.
The bar is the eager path's: forward bit-exact vs the oracle, dense adjoint within tolerance (wave reduction).
Since round 2 operators whose dense children are all small run the whole loop in ONE launch (tests/test_gpu_small_loop.py);
the per-block loop and its graphs remain for bigger children, so these tests switch the one-launch loop off."""
import numpy as np
import pytest

from .helpers import assert_bits_equal, u01

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _per_block_loop(Jets):
    Jets.tune(small_loop=0)
    yield
    Jets.tune(small_loop=1)


def _dense_mix(Jets, oracle, dt, seed):
    from .test_gpu_nonlinear import _mixed, _split
    len_r, len_c = [40, 24, 40], [40, 24, 40, 40]
    kinds = [["dense", "dense", "diag", "zero"], ["dense", "dense", "dense", "dense"], ["identity", "dense", "dense", "scale"]]
    A, ops = _mixed(Jets, oracle, dt, len_r, len_c, kinds, seed=seed, hmo=None)
    return A, ops, len_r, len_c, _split


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64])
def test_per_block_loop_replayed_as_a_graph_gives_the_eager_bits(Jets, oracle, dt):
    """Operators with dense children run the reference's per-block loop (2 launches per block).  Call 1 is eager, call 2
    is captured into a hipGraph, calls 3+ replay it: same bits every time, with fresh vector CONTENTS each call (the graph
    reads the buffers, not a snapshot) and with the knob off."""
    A, ops, len_r, len_c, _split = _dense_mix(Jets, oracle, dt, seed=1234)
    m = Jets.zeros(Jets.domain(A))
    d = Jets.zeros(Jets.range(A))
    mt = Jets.zeros(Jets.domain(A))
    NR, NC = sum(len_r), sum(len_c)
    replays0 = Jets.tune_get("graph_replays")
    for call in range(5):
        if call == 4:
            Jets.tune(graphs=0)
        try:
            hm = u01(oracle, dt, 70, call, NC)
            hd0 = u01(oracle, dt, 71, call, NR)
            Jets.copyto_(m, Jets.from_numpy(hm, Jets.domain(A)))
            Jets.copyto_(d, Jets.from_numpy(hd0, Jets.range(A)))                      # dirty output: ncol > 1 accumulates into it
            Jets.mul_(d, A, m)
            ref = oracle.block_df(ops, _split(hd0, len_r), _split(hm, len_c))
            assert_bits_equal(d.to_numpy(), np.concatenate(ref), f"forward, call {call}")
            Jets.mul_(mt, A.H, d)
            refm = oracle.block_df_adj(ops, _split(u01(oracle, dt, 72, call, NC), len_c), ref)
            tol = 2e-5 if np.dtype(dt).itemsize <= 8 and np.dtype(dt) != np.float64 else 1e-12
            np.testing.assert_allclose(mt.to_numpy(), np.concatenate(refm), rtol=tol, atol=tol)   # dense adjoint: wave reduction
        finally:
            Jets.tune(graphs=1)
        # forward + adjoint: eager on call 0, captured-and-launched on call 1, replayed on calls 2 and 3, eager again with the knob off
        assert Jets.tune_get("graph_replays") - replays0 == 2 * min(call, 3), f"call {call}"


def test_graph_replay_survives_scratch_growth_and_other_operators(Jets, oracle):
    """A captured loop holds the context's scratch pointer: when a bigger operator makes the scratch buffer move, the
    stale graph must be dropped and re-captured, not replayed."""
    dt = np.float32
    A, ops, len_r, len_c, _split = _dense_mix(Jets, oracle, dt, seed=4321)
    m = Jets.rand(Jets.domain(A), seed=80, stream=0)
    hm = u01(oracle, dt, 80, 0, sum(len_c))
    d = Jets.zeros(Jets.range(A))
    want = np.concatenate(oracle.block_df(ops, [np.zeros(n, dt) for n in len_r], _split(hm, len_c)))
    for _ in range(3):                                                               # eager, capture, replay
        Jets.fill_(d, 0)
        Jets.mul_(d, A, m)
        assert_bits_equal(d.to_numpy(), want, "before growth")
    big = Jets.blockop([[Jets.JopDense(Jets.rand(Jets.JetSpace(dt, 600_000, 2), seed=85, stream=j)) for j in range(2)]])
    xb = Jets.rand(Jets.domain(big), seed=86, stream=0)
    for _ in range(3):
        Jets.mul(big, xb)                                                            # 2.4 MB dtmp > the 1 MiB scratch: it is reallocated
    for _ in range(3):
        Jets.fill_(d, 0)
        Jets.mul_(d, A, m)
        assert_bits_equal(d.to_numpy(), want, "after the scratch buffer moved")


# ---- round 4: CGLS and CG on the normal equations with the recurrences on the device (jh_lsqr.hip: cg_dev_impl) --------------------------
@pytest.mark.parametrize("shape", [(32, 16, 8), (31, 17, 7)], ids=["aligned", "odd-blocks"])    # 3689 elements: rows off the 16-byte pack grid (round 5)
@pytest.mark.parametrize("solver", ["cgnr", "cgls"])
@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64, np.complex128])
def test_graph_replayed_cg_loops_of_small_operators_have_the_bits_of_the_host_driven_loop(Jets, oracle, dt, solver, shape):
    """Small operators (docs/src/index.md:235-246: the iterative solver over the block operator): an iteration of CGLS / of CG through
    the fused A'A is 4-5 graph nodes whose coefficients live in device memory; with lsqr_graph = 0 the same kernels run eagerly and the
    host applies the same two scalar updates between them.  x, the iteration count, the stopping rule and the whole history must be
    IDENTICAL -- with early stopping, damping, a warm start, forced iterations and a single iteration; and within solver tolerance of the
    loops large / partitioned operators take (cg_dev = 0) and of the textbook fp64 CPU CGLS."""
    from oracle import cgls_ref

    from .helpers import make_tall_diag

    J = Jets
    nrow = 9
    A, _, _, diags = make_tall_diag(J, oracle, dt, nrow, shape)
    n = int(np.prod(shape))
    hb = (u01(oracle, dt, 71, 0, nrow * n) - dt(0.5)).astype(dt)
    x0 = J.rand(J.domain(A), seed=72, stream=0)
    solve = getattr(J, solver)
    cases = (dict(maxiter=25, atol=0.0, btol=0.0), dict(maxiter=60, atol=1e-4, btol=1e-4), dict(maxiter=30, atol=0.0, btol=0.0, damp=0.25),
             dict(maxiter=17, atol=0.0, btol=0.0, x0=x0), dict(maxiter=1, atol=0.0, btol=0.0), dict(maxiter=9, atol=1e-2, btol=1e-2, force_maxiter=True),
             dict(maxiter=11, atol=0.0, btol=0.0, damp=0.5, x0=x0))
    for kw in cases:
        out = {}
        for mode in ("graph", "host", "large"):
            J.tune(lsqr_graph=0 if mode == "host" else 1, cg_dev=0 if mode == "large" else 1)
            try:
                r = solve(A, J.from_numpy(hb, J.range(A)), **kw)
                out[mode] = (r, J.tune_get("last_cg_graph"))
            finally:
                J.tune(lsqr_graph=1, cg_dev=1)
        (rg, replays), (rh, zero), (rl, _) = out["graph"], out["host"], out["large"]
        assert zero == 0 and (replays > 0 or kw["maxiter"] == 1), "the graph path ran (and only when asked)"
        assert (rg.itn, rg.istop) == (rh.itn, rh.istop), kw
        assert_bits_equal(rg.x.to_numpy(), rh.x.to_numpy(), f"x, {kw}")
        assert rg.history == rh.history, kw
        for f in ("r1norm", "r2norm", "arnorm", "xnorm"):
            assert getattr(rg, f) == getattr(rh, f), (f, kw)
        single = np.dtype(dt) in (np.dtype(np.float32), np.dtype(np.complex64))
        tol = 2e-4 if single else 1e-9
        if "force_maxiter" not in kw and kw["atol"] == 0.0:
            assert rl.itn == rg.itn
            np.testing.assert_allclose(rg.x.to_numpy(), rl.x.to_numpy(), rtol=tol, atol=tol * 1e-1)
            # the textbook fp64 loop (with its q = A p in a range-sized vector) on the host copies of the same data
            D = np.stack([d.astype(np.complex128 if np.iscomplexobj(d) else np.float64) for d in diags])
            mv = lambda v: (D * v[None, :]).ravel()
            rmv = lambda u: (np.conj(D) * u.reshape(nrow, n)).sum(axis=0)
            ref = cgls_ref.cgls_fp64(mv, rmv, hb.astype(D.dtype), n, x0=None if "x0" not in kw else x0.to_numpy().ravel(order="F").astype(D.dtype),
                                     damp=kw.get("damp", 0.0), atol=0.0, btol=0.0, maxiter=kw["maxiter"])
            xr = ref[0] if isinstance(ref, tuple) else ref.x
            np.testing.assert_allclose(rg.x.to_numpy().ravel(order="F"), np.asarray(xr).astype(dt), rtol=tol, atol=tol * 1e-1)
    J.close(A)


def test_many_rows_take_the_device_loops_when_a_pass_is_one_plain_launch(Jets):
    """256 rows and more: the split-row walk (pick_adj_parts) is what keeps the step from being one launch, not the row count.  256
    rows of 64^3 Float32 fill the chip with the plain walk (one workgroup per CU), so all three loops replay graphs -- with the bits of
    the host-driven loops; 256 rows of 16^3 do not (4 workgroups), so LSQR / CGLS keep the split walk and CG through the fused A'A the
    host loop, whose A'A splits the rows."""
    J = Jets
    for edge, expect_graph in ((64, True), (16, False)):
        spc = J.JetSpace(np.float32, edge, edge, edge)
        coeff = J.rand(J.JetBSpace([spc] * 256), seed=81, stream=0)
        A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
        b = J.mul(A, J.rand(spc, seed=82, stream=0))
        for name, knob in (("lsqr", "last_lsqr_graph"), ("cgls", "last_cg_graph"), ("cgnr", "last_cg_graph")):
            kw = dict(maxiter=21, atol=0.0, btol=0.0, force_maxiter=True)
            if name == "lsqr":
                kw["conlim"] = 0.0
            out = {}
            for graph in (1, 0):
                J.tune(lsqr_graph=graph)
                try:
                    r = getattr(J, name)(A, b, **kw)
                    out[graph] = (r, J.tune_get(knob))
                finally:
                    J.tune(lsqr_graph=1)
            (rg, replays), (rh, zero) = out[1], out[0]
            assert zero == 0 and (replays > 0) == expect_graph, (name, edge, replays)
            assert (rg.itn, rg.istop) == (rh.itn, rh.istop) == (21, 7), (name, edge)
            if expect_graph:
                assert_bits_equal(rg.x.to_numpy(), rh.x.to_numpy(), f"{name} x, 256 x {edge}^3")
                assert rg.history == rh.history, (name, edge)
        J.close(A)


@pytest.mark.parametrize("dt", [np.float32, np.complex64, np.float64])
@pytest.mark.parametrize("shape,nrow", [((128, 128, 32), 5), ((256, 256, 64), 3), ((256, 256, 256), 3)])
def test_cg_normal_pass_at_its_three_launch_shapes(Jets, dt, shape, nrow):
    """k_cg_normal launches thin workgroups for launch-bound domains and the fused normal operator's fat ones from 2 MiB / 64 MiB blocks on
    (jh_launch_cg_normal): whichever shape, the replayed loop has the bits of the host-driven one (same kernels), both agree with the loop
    of large operators (cg_dev = 0: separate lincombs around jh_blockop_normal_mul) to solver tolerance, and the solve converges."""
    J = Jets
    if np.dtype(dt).itemsize * int(np.prod(shape)) * nrow > (1 << 30):
        pytest.skip("more than 1 GiB of coefficients for a shape check")
    spc = J.JetSpace(dt, *shape)
    coeff = J.rand(J.JetBSpace([spc] * nrow), seed=91, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    x_true = J.rand(spc, seed=92, stream=0)
    b = J.mul(A, x_true)
    out = {}
    for mode in ("graph", "host", "large"):
        J.tune(lsqr_graph=0 if mode == "host" else 1, cg_dev=0 if mode == "large" else 1)
        try:
            out[mode] = (J.cgnr(A, b, maxiter=12, atol=0.0, btol=0.0, damp=0.125), J.tune_get("last_cg_graph"))
        finally:
            J.tune(lsqr_graph=1, cg_dev=1)
    (rg, replays), (rh, zero), (rl, _) = out["graph"], out["host"], out["large"]
    assert replays > 0 and zero == 0
    assert (rg.itn, rg.istop) == (rh.itn, rh.istop) == (rl.itn, rl.istop)
    assert_bits_equal(rg.x.to_numpy(), rh.x.to_numpy(), f"x, {shape}")
    assert rg.history == rh.history
    tol = 2e-4 if np.dtype(dt).itemsize <= 8 and np.dtype(dt) != np.dtype(np.float64) else 1e-9
    np.testing.assert_allclose(rg.x.to_numpy(), rl.x.to_numpy(), rtol=tol, atol=tol * 1e-1)
    J.close(A)
