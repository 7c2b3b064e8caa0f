"""GPU: several CONTEXTS in one process (include/jetship.h conventions; SURVEY section 8e's single-process form).

The one-GPU test box cannot hold two devices, so the contexts here are several streams of ONE GPU: everything but RCCL itself is
exercised -- per-context workspaces and knobs, handles that carry their context, the refusal of mixed handles, and a TEAM
(rowpart.Team: jh_comm_init_all + grouped all-reduces, here the one-device sum kernel) running the row-partitioned forward, the
pipelined adjoint, the one-pass LSQR step and a whole LSQR solve.  With >= 2 devices visible the same team test also runs over
RCCL (ncclCommInitAll), one context per device."""
import ctypes as C

import numpy as np
import pytest

from .helpers import assert_bits_equal, rel_err, u01

pytestmark = pytest.mark.gpu


@pytest.fixture
def two_contexts(Jets):
    J = Jets
    base = J.context_current()[0]
    extra = J.context_create(0)
    J.context_use(base)
    yield base, extra
    import gc

    gc.collect()                                                     # a context is destroyed after everything created in it
    J.context_use(base)
    J.context_destroy(extra)


def test_handles_carry_their_context_and_mixed_handles_are_refused(Jets, two_contexts):
    from jets_jl_amd._ffi import lib

    J = Jets
    base, extra = two_contexts
    spc = J.JetSpace(np.float32, 4096)
    x0 = J.rand(spc, seed=1, stream=0)
    with J.using_context(extra) as c:
        assert c == extra and J.context_current() == (extra, 0)
        x1 = J.rand(spc, seed=1, stream=0)
        y1 = J.zeros(spc)
        s1 = J.stream_handle()
    assert J.context_current()[0] == base
    assert J.context_of(x0) == base and J.context_of(x1) == extra and J.context_of(y1) == extra
    assert J.stream_handle() != s1                                   # a stream of its own
    assert_bits_equal(x0.to_numpy(), x1.to_numpy(), "the generator does not depend on the context")
    # an operation runs in its handles' context, whatever is current -- and makes it current
    J.lincomb_(y1, [2.0], [x1])
    assert J.context_current()[0] == extra
    assert_bits_equal(y1.to_numpy(), np.float32(2.0) * x1.to_numpy(), "lincomb in the other context")
    with pytest.raises(J.JetsHipError, match="lives in context"):
        J.lincomb_(y1, [1.0, 1.0], [x0, x1])
    with pytest.raises(J.JetsHipError, match="different contexts"):
        J.hadamard_(y1, x0, x1)
    # views inherit the context
    b = J.zeros(J.JetBSpace([spc, spc]))
    assert J.context_of(b) == extra and J.context_of(J.getblock(b, 1)) == extra
    # knobs are per context
    J.context_use(base)
    J.tune(adj_split=0)
    try:
        assert J.tune_get("adj_split") == 0
        J.context_use(extra)
        assert J.tune_get("adj_split") == -1
    finally:
        J.context_use(base)
        J.tune(adj_split=-1)
    # an unknown context / a device without a context / a context that still owns handles
    assert lib.jh_context_use(63) == 5 and lib.jh_set_device(7) == 5
    assert lib.jh_context_destroy(extra) == 5 and b"still owns" in lib.jh_last_error()
    for x in (x1, y1, b):
        x.close()


def test_an_operator_lives_where_its_coefficients_live(Jets, oracle, two_contexts):
    J = Jets
    base, extra = two_contexts
    dt, nrow, n = np.float32, 5, 4096
    spc = J.JetSpace(dt, n)
    with J.using_context(extra):
        coeff = [J.rand(spc, seed=5, stream=i) for i in range(nrow)]
        m = J.rand(spc, seed=6, stream=0)
    J.context_use(base)                                              # build and apply while ANOTHER context is current
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff])
    d = A * m                                                        # the result is allocated in m's context
    assert J.context_of(d) == extra
    ops = [[oracle.Block("diag", n, coeff=u01(oracle, dt, 5, i, n))] for i in range(nrow)]
    hm = u01(oracle, dt, 6, 0, n)
    want = oracle.block_df(ops, [np.zeros(n, dt) for _ in range(nrow)], [hm])
    assert_bits_equal(d.to_numpy(), np.concatenate(want), "forward in the second context")
    mt = A.H * d
    assert J.context_of(mt) == extra
    assert_bits_equal(mt.to_numpy().ravel(order="F"), oracle.block_df_adj(ops, [np.zeros(n, dt)], want)[0], "adjoint in the second context")
    y = (A.H @ A) * m                                                # the composite's temporaries too
    assert_bits_equal(y.to_numpy().ravel(order="F"), mt.to_numpy().ravel(order="F"), "fused A'A in the second context")
    x0 = J.rand(spc, seed=7, stream=0)                               # base context (current again after the explicit use)
    if J.context_of(x0) != extra:
        with pytest.raises(J.JetsHipError, match="different contexts"):
            J.mul_(d, A, x0)
    J.close(A)


def _team_contexts(J, nmem):
    """One context per device when there are enough devices (RCCL), else `nmem` contexts of device 0."""
    ndev = J.device_count()
    if ndev >= nmem:
        ctxs = []
        for dev in range(nmem):
            J.init(dev)
            ctxs.append(J.context_current()[0])
        return ctxs, []
    base = J.context_current()[0]
    extra = [J.context_create(0) for _ in range(nmem - 1)]
    return [base] + extra, extra


@pytest.mark.parametrize("shape", [(64, 64, 16), (65, 63, 17)], ids=["aligned", "odd-blocks"])     # 69 615 elements: rows off the 16-byte grid, the last exchange range ends inside a pack
@pytest.mark.parametrize("nmem", [2, 3])
def test_a_team_of_contexts_runs_the_row_partitioned_flow(Jets, oracle, nmem, shape):
    import gc

    from jets_jl_amd import rowpart

    J = Jets
    J.init(0)
    home = J.context_current()[0]
    ctxs, extra = _team_contexts(J, nmem)
    team = rowpart.Team(ctxs)
    try:
        _team_flow(J, oracle, rowpart, team, ctxs, nmem, home, shape)
    finally:
        team.close()
        gc.collect()                                                 # the members' vectors and operators die before their contexts
        J.context_use(home)
        for c in extra:
            J.context_destroy(c)


def _team_flow(J, oracle, rowpart, team, ctxs, nmem, home, shape=(64, 64, 16)):
    dt, nrow = np.float32, 11
    n = int(np.prod(shape))
    spc = J.JetSpace(dt, *shape)
    if True:
        parts = [rowpart.partition_rows(nrow, nmem, k) for k in range(nmem)]
        local_ops, coeffs = [], []
        for k, _ in team.each():
            cs = [J.rand(spc, seed=1, stream=0, index_base=(parts[k].first + i) * n) for i in range(parts[k].count)]
            coeffs.append(cs)
            local_ops.append(J.blockop([[J.JopDiagonal(c)] for c in cs]))
        T = team.operator(local_ops)
        ha = [oracle.rng_u01(dt, 1, 0, i * n, n) for i in range(nrow)]
        ops = [[oracle.Block("diag", n, coeff=a)] for a in ha]
        hm = u01(oracle, dt, 2, 0, n)
        m = rowpart.TeamVec([J.rand(spc, seed=2, stream=0) for _ in team.each()])
        d = team.zeros(T.ranges())
        T.mul_(d, m)
        want_d = oracle.block_df(ops, [np.zeros(n, dt) for _ in range(nrow)], [hm])
        for k in range(nmem):
            lo, cnt = parts[k].first, parts[k].count
            assert J.context_of(d[k]) == ctxs[k]
            assert_bits_equal(d[k].to_numpy(), np.concatenate(want_d[lo:lo + cnt]), f"forward rows of member {k}")
        # adjoint: every member's ordered sum, members added by the grouped all-reduce -> tolerance parity, replicas identical
        mt = team.zeros(T.domain())
        T.mul_adj_(mt, d)
        want_m = oracle.block_df_adj(ops, [np.zeros(n, dt)], want_d)[0]
        got = [x.to_numpy().ravel(order="F") for x in mt.members]
        assert rel_err(got[0], want_m) < 1e-6
        for k in range(1, nmem):
            assert_bits_equal(got[k], got[0], f"replica {k} of the summed adjoint")
        # the fused normal operator over the team (ranged A_k'A_k m + grouped ranged all-reduces): A'(A m), replicas identical
        yn = team.zeros(T.domain())
        T.normal_mul_(yn, m)
        gy = [x.to_numpy().ravel(order="F") for x in yn.members]
        assert rel_err(gy[0], want_m) < 1e-6
        for k in range(nmem):
            assert_bits_equal(gy[k], got[0], f"replica {k} of (A'A) m against the team's adjoint of the forward")
        # one-pass step through the team vs the oracle's unfused sequence
        hu = [u01(oracle, dt, 3, i, n) for i in range(nrow)]
        u = rowpart.TeamVec([J.from_numpy(np.concatenate(hu[parts[k].first:parts[k].first + parts[k].count]), T.ranges()[k]) for k, _ in team.each()])
        w = team.zeros(T.domain())
        nrm2 = T.bidiag_step_(u, m, w, 0.75, -0.5)
        av = oracle.block_df(ops, [np.zeros(n, dt) for _ in range(nrow)], [hm])
        want_u = oracle.barr_lincomb([np.empty(n, dt) for _ in range(nrow)], [0.75, -0.5], [av, hu])
        for k in range(nmem):
            lo, cnt = parts[k].first, parts[k].count
            assert_bits_equal(u[k].to_numpy(), np.concatenate(want_u[lo:lo + cnt]), f"one-pass step: rows of member {k}")
        want_w = oracle.block_df_adj(ops, [np.zeros(n, dt)], want_u)[0]
        gw = [x.to_numpy().ravel(order="F") for x in w.members]
        assert rel_err(gw[0], want_w) < 1e-6 and all(np.array_equal(gw[k], gw[0]) for k in range(1, nmem))
        want_n = float(sum(np.vdot(b.astype(np.float64), b.astype(np.float64)) for b in want_u))
        assert abs(nrm2 - want_n) <= 1e-10 * want_n
        # LSQR over the team against the single-context solver on the whole operator
        J.context_use(home)
        call = [J.rand(spc, seed=1, stream=0, index_base=i * n) for i in range(nrow)]
        A = J.blockop([[J.JopDiagonal(c)] for c in call])
        x_true = J.rand(spc, seed=4, stream=0)
        b = A * x_true
        ref = J.lsqr(A, b, maxiter=15, atol=0.0, btol=0.0, conlim=0.0)
        hb = b.to_numpy()
        bt = rowpart.TeamVec([J.from_numpy(hb[parts[k].first * n:(parts[k].first + parts[k].count) * n], T.ranges()[k]) for k, _ in team.each()])
        import os
        for native in ("1", "0"):                                # jh_lsqr_solve_team (one call), then lsqr_core driving the team from Python
            os.environ["JETS_LSQR_NATIVE"] = native
            try:
                res = J.lsqr(T, bt, maxiter=15, atol=0.0, btol=0.0, conlim=0.0)
            finally:
                os.environ.pop("JETS_LSQR_NATIVE", None)
            assert res.itn == ref.itn == 15
            np.testing.assert_allclose([h[1] for h in res.history], [h[1] for h in ref.history], rtol=2e-4)
            xs = [x.to_numpy() for x in res.x.members]
            np.testing.assert_allclose(xs[0], ref.x.to_numpy(), rtol=1e-4, atol=1e-6)
            for k in range(1, nmem):
                assert_bits_equal(xs[k], xs[0], f"replica {k} of the LSQR solution (native {native})")
        # early stopping and a warm start through the one-call team solver
        x0 = rowpart.TeamVec([J.rand(spc, seed=5, stream=0) for _ in team.each()])
        J.context_use(home)
        x0h = J.rand(spc, seed=5, stream=0)
        ref2 = J.lsqr(A, b, x0=x0h, maxiter=60, atol=1e-5, btol=1e-5, damp=0.1)
        res2 = J.lsqr(T, bt, x0=x0, maxiter=60, atol=1e-5, btol=1e-5, damp=0.1)
        assert res2.istop == ref2.istop and abs(res2.itn - ref2.itn) <= 1 and res2.itn < 60
        np.testing.assert_allclose(res2.x[0].to_numpy(), ref2.x.to_numpy(), rtol=1e-3, atol=1e-5)
        # the members' handles in the wrong order are refused
        from jets_jl_amd._ffi import lib, LsqrResultC
        if nmem >= 2:
            arr = lambda hs: (C.c_void_p * nmem)(*[h.value for h in hs])
            order = list(range(nmem))[::-1]
            r = LsqrResultC()
            rc = lib.jh_lsqr_solve_team(nmem, arr([T._natives[k].handle for k in order]), arr([bt[k].handle for k in order]),
                                        arr([res.x[k].handle for k in order]), 0, 0.0, 0.0, 0.0, 0.0, 3, 0, C.byref(r), None)
            assert rc == 1 and b"must live in member" in lib.jh_last_error()
        # a member's collective outside a group is refused, not deadlocked
        with pytest.raises(J.JetsHipError, match="jh_comm_group_begin"):
            from jets_jl_amd._ffi import lib, check
            check(lib.jh_comm_allreduce_sum(mt[0].handle))
        J.close(A)
        for Ak in local_ops:
            J.close(Ak)


def test_a_team_of_one_goes_through_rccl(Jets, oracle):
    """jh_comm_init_all with ONE context takes the RCCL branch (ncclCommInitAll over one device, ncclGroupStart / ncclGroupEnd
    around the ranged all-reduces) -- the code a multi-device team runs, minus the peers: the pipelined adjoint and the one-pass
    step then equal the single-context ordered walk bit for bit."""
    from jets_jl_amd import rowpart

    import gc

    J = Jets
    J.init(0)
    home = J.context_current()[0]
    ctx = J.context_create(0)
    team = rowpart.Team([ctx])
    try:
        _team_of_one(J, oracle, rowpart, team, ctx)
    finally:
        team.close()
        gc.collect()
        J.context_use(home)
        J.context_destroy(ctx)


def _team_of_one(J, oracle, rowpart, team, ctx):
    dt, nrow, shape = np.float32, 6, (64, 64, 16)
    n = int(np.prod(shape))
    spc = J.JetSpace(dt, *shape)
    if True:
        with J.using_context(ctx):
            cs = [J.rand(spc, seed=1, stream=0, index_base=i * n) for i in range(nrow)]
            A = J.blockop([[J.JopDiagonal(c)] for c in cs])
            m = rowpart.TeamVec([J.rand(spc, seed=2, stream=0)])
        T = team.operator([A])
        d = team.zeros(T.ranges())
        T.mul_(d, m)
        mt = team.zeros(T.domain())
        T.mul_adj_(mt, d)
        ops = [[oracle.Block("diag", n, coeff=oracle.rng_u01(dt, 1, 0, i * n, n))] for i in range(nrow)]
        hm = u01(oracle, dt, 2, 0, n)
        want_d = oracle.block_df(ops, [np.zeros(n, dt) for _ in range(nrow)], [hm])
        assert_bits_equal(mt[0].to_numpy().ravel(order="F"), oracle.block_df_adj(ops, [np.zeros(n, dt)], want_d)[0], "team of one: adjoint")
        u = rowpart.TeamVec([J.rand(T.ranges()[0], seed=3, stream=0)])
        hu = [oracle.rng_u01(dt, 3, 0, i * n, n) for i in range(nrow)]
        w = team.zeros(T.domain())
        nrm2 = T.bidiag_step_(u, m, w, 1.0, -0.5)
        want_u = oracle.barr_lincomb([np.empty(n, dt) for _ in range(nrow)], [1.0, -0.5], [want_d, hu])
        assert_bits_equal(u[0].to_numpy(), np.concatenate(want_u), "team of one: one-pass step, u")
        assert_bits_equal(w[0].to_numpy().ravel(order="F"), oracle.block_df_adj(ops, [np.zeros(n, dt)], want_u)[0], "team of one: one-pass step, w")
        assert nrm2 > 0
        J.close(A)


def test_two_host_threads_each_driving_its_own_context(Jets, oracle, two_contexts):
    """One host thread per context, concurrently (the current context is per THREAD; ctypes releases the GIL around every call):
    each builds its own operator and runs forward / adjoint / fused A'A / a broadcast / LSQR in a loop; every result must be
    the oracle's bits -- nothing of one context (stream, scratch, partial sums, the deferred accumulators) leaks into the other."""
    import threading

    J = Jets
    base, extra = two_contexts
    dt, shape = np.float32, (64, 32, 16)
    n = int(np.prod(shape))
    errors = []

    def work(ctx, nrow, seed):
        try:
            J.context_use(ctx)
            spc = J.JetSpace(dt, *shape)
            cs = [J.rand(spc, seed=seed, stream=i) for i in range(nrow)]
            A = J.blockop([[J.JopDiagonal(c)] for c in cs])
            ops = [[oracle.Block("diag", n, coeff=u01(oracle, dt, seed, i, n))] for i in range(nrow)]
            for rep in range(12):
                hm = u01(oracle, dt, seed + 1, rep, n)
                m = J.from_numpy(hm.reshape(shape, order="F"))
                assert J.context_of(m) == ctx
                d = A * m
                want_d = oracle.block_df(ops, [np.zeros(n, dt) for _ in range(nrow)], [hm])
                assert_bits_equal(d.to_numpy(), np.concatenate(want_d), f"thread of context {ctx}: forward, repetition {rep}")
                mt = A.H * d
                want_m = oracle.block_df_adj(ops, [np.zeros(n, dt)], want_d)[0]
                assert_bits_equal(mt.to_numpy().ravel(order="F"), want_m, f"thread of context {ctx}: adjoint, repetition {rep}")
                y = (A.H @ A) * m
                assert_bits_equal(y.to_numpy().ravel(order="F"), want_m, f"thread of context {ctx}: fused A'A, repetition {rep}")
                z = J.zeros(spc)
                J.broadcast_(z, "x0*x1+s0", [m, mt], [0.5])
                assert_bits_equal(z.to_numpy().ravel(order="F"), hm * want_m + np.float32(0.5), f"thread of context {ctx}: broadcast")
                res = J.lsqr(A, d, maxiter=8, atol=0.0, btol=0.0, conlim=0.0)
                assert J.context_of(res.x) == ctx and res.itn == 8
                assert rel_err(res.x.to_numpy().ravel(order="F"), hm) < 5e-2 and res.history[-1][1] < 0.1 * res.history[0][1]   # 8 iterations in
            J.close(A)
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((ctx, repr(e)))

    ts = [threading.Thread(target=work, args=(base, 7, 40)), threading.Thread(target=work, args=(extra, 10, 50))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=240)
    assert not any(t.is_alive() for t in ts), "a worker thread hangs"
    assert not errors, errors
    J.context_use(base)


def test_a_finaliser_never_changes_the_current_context(Jets, two_contexts):
    """jh_bvec_destroy / jh_blockop_destroy run whenever the garbage collector says so: collecting an array or an operator of
    context B inside `using_context(A)` must leave A current (round-2 advisor finding: the destructors went through jh_enter)."""
    import gc

    J = Jets
    base, extra = two_contexts
    spc = J.JetSpace(np.float32, 8192)
    with J.using_context(extra):
        x1 = J.rand(spc, seed=5, stream=0)
        g1 = J.rand(spc, seed=6, stream=0)
        A1 = J.blockop([[J.JopDiagonal(g1)], [J.JopDiagonal(g1)]])
        d1 = A1 * x1                                                   # builds the device operator
        del d1
    with J.using_context(base):
        assert J.context_current()[0] == base
        del x1
        gc.collect()
        assert J.context_current()[0] == base, "collecting a vector of another context switched the current context"
        y = J.zeros(spc)                                               # a factory call right after: allocates where the caller is
        assert J.context_of(y) == base
        J.close(A1)
        del A1, g1
        gc.collect()
        assert J.context_current()[0] == base, "collecting an operator of another context switched the current context"
        assert J.context_of(J.rand(spc, seed=1, stream=0)) == base


def test_a_handle_that_outlives_its_context_never_resolves_to_a_later_one(Jets):
    """Context ids carry a generation (slot + 64 * generation): after jh_context_destroy the slot is reused, the id is not."""
    from jets_jl_amd._ffi import lib

    J = Jets
    base = J.context_current()[0]
    a = J.context_create(0)
    J.context_use(base)
    J.context_destroy(a)
    b = J.context_create(0)                                            # lands in the slot `a` had
    J.context_use(base)
    try:
        assert b != a and (b & 63) == (a & 63)
        assert lib.jh_context_use(a) != 0 and b"does not exist" in lib.jh_last_error()
        assert lib.jh_context_destroy(a) != 0                          # "no context": the newer one is not touched
        J.context_use(b)
        assert J.context_current()[0] == b
    finally:
        J.context_use(base)
        J.context_destroy(b)
