"""GPU: the one-pass LSQR step as CHAINED ROW CHUNKS (k_tall_diag_bidiag_chain): one batch of 8 rows per workgroup, the ordered
sum w = sum_i conj(a_i) .* u_i handed from chunk to chunk through memory -- the same additions in the same order, so u, w are
bit-identical to the plain walk and to the oracle's unfused sequence (mul! into a temporary, axpby, then the adjoint loop,
src/Jets.jl:1042-1049).  Forced here on small operators with the knob step_chain = 1 (automatic only for big ones)."""
import ctypes as C

import numpy as np
import pytest

from .helpers import DTYPES, assert_bits_equal, u01

pytestmark = pytest.mark.gpu


def _op(J, oracle, dt, nrow, n, slab, seed=51):
    spc = J.JetSpace(dt, n)
    if slab:                                                   # one slab of coefficients: strided addressing in the kernel
        coeff = J.rand(J.JetBSpace([spc] * nrow), seed=seed, stream=0)
        dev = [J.JopDiagonal(c) for c in coeff.arrays]
        host = [oracle.rng_u01(dt, seed, 0, i * n, n) for i in range(nrow)]
    else:                                                      # separate arrays: the row table
        dev = [J.JopDiagonal(J.rand(spc, seed=seed, stream=i)) for i in range(nrow)]
        host = [u01(oracle, dt, seed, i, n) for i in range(nrow)]
    return J.blockop([[d] for d in dev]), [[oracle.Block("diag", n, coeff=h)] for h in host]


def _native(J, A):
    from jets_jl_amd import jetblock as _blk

    return _blk._tall_native(A)


def _expected_chunks(nrow, chunk):
    """row chunks of the chained walk of an ALL-DIAGONAL operator under the knob step_chunk (jh_tall_step.hip: launch_bidiag): 0 = automatic =
    32 rows of a 256-lane tile per workgroup (round 5) when there are more than 32 rows, else 8-row chunks; one chunk = the plain walk (0)"""
    cd = 32 if (chunk in (0, 32) and nrow > 32) else (16 if chunk == 16 else 8)
    return (nrow + cd - 1) // cd if nrow > cd else 0


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("chunk", [0, 8, 16])
@pytest.mark.parametrize("nrow,slab", [(2, True), (8, False), (9, True), (16, False), (33, False), (37, True), (64, False), (100, True)])
def test_chained_step_has_the_bits_of_the_ordered_walk(Jets, oracle, dt, nrow, slab, chunk):
    from jets_jl_amd._ffi import lib, check

    J = Jets
    if chunk and nrow < 16 and dt != np.float32:
        pytest.skip("the forced chunk sizes differ from the automatic rule only above their own row count: one element type suffices below")
    n = 4096                                                   # 256 lanes x 16 B divide every eltype's row
    A, ops = _op(J, oracle, dt, nrow, n, slab)
    nat = _native(J, A)
    hv = u01(oracle, dt, 2, 0, n)
    hu = [u01(oracle, dt, 3, i, n) for i in range(nrow)]
    v = J.from_numpy(hv)
    out = C.c_double(0)
    for alpha, beta in ((0.75, -0.5), (1.0, 0.0)):
        av = oracle.block_df(ops, [np.zeros(n, dt) for _ in range(nrow)], [hv])
        want_u = oracle.barr_lincomb([np.empty(n, dt) for _ in range(nrow)], [alpha, beta], [av, hu]) if beta else \
            oracle.barr_lincomb([np.empty(n, dt) for _ in range(nrow)], [alpha], [av])
        want_w = oracle.block_df_adj(ops, [np.zeros(n, dt)], want_u)[0]
        nrm = float(sum(np.vdot(b.astype(np.complex128), b.astype(np.complex128)).real for b in want_u))
        try:
            J.tune(step_chain=1, step_chunk=chunk)
            u = J.from_numpy(np.concatenate(hu), J.range(A))
            w = J.rand(J.domain(A), seed=9, stream=0)           # dirty
            check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, alpha, beta, C.byref(out)))
            assert J.tune_get("last_step_chain") == _expected_chunks(nrow, chunk)            # one chunk = nothing to chain: the plain walk
            assert_bits_equal(u.to_numpy(), np.concatenate(want_u), f"chained step: u ({alpha}, {beta})")
            assert_bits_equal(w.to_numpy().ravel(order="F"), want_w, f"chained step: w ({alpha}, {beta})")
            assert abs(out.value - nrm) <= 1e-12 * nrm
            # ranged (the pipelined multi-GPU form): two halves, deferred ||u||^2
            u2 = J.from_numpy(np.concatenate(hu), J.range(A))
            w2 = J.rand(J.domain(A), seed=10, stream=0)
            check(lib.jh_normsq_reset())
            half = n // 2
            check(lib.jh_blockop_bidiag_step_range(nat.handle, u2.handle, v.handle, w2.handle, alpha, beta, 0, half, None))
            check(lib.jh_blockop_bidiag_step_range(nat.handle, u2.handle, v.handle, w2.handle, alpha, beta, half, n - half, None))
            check(lib.jh_normsq_read(C.byref(out)))
            assert_bits_equal(u2.to_numpy(), np.concatenate(want_u), "chained ranged step: u")
            assert_bits_equal(w2.to_numpy().ravel(order="F"), want_w, "chained ranged step: w")
            assert abs(out.value - nrm) <= 1e-12 * nrm
        finally:
            J.tune(step_chain=-1, step_chunk=0)
        # and the plain walk agrees, of course
        J.tune(step_chain=0)
        try:
            u3 = J.from_numpy(np.concatenate(hu), J.range(A))
            w3 = J.zeros(J.domain(A))
            check(lib.jh_blockop_bidiag_step(nat.handle, u3.handle, v.handle, w3.handle, alpha, beta, C.byref(out)))
            assert J.tune_get("last_step_chain") == 0
            assert_bits_equal(u3.to_numpy(), np.concatenate(want_u), "plain step: u")
            assert_bits_equal(w3.to_numpy().ravel(order="F"), want_w, "plain step: w")
        finally:
            J.tune(step_chain=-1)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nrow", [9, 16, 37])
def test_chained_step_with_rows_of_several_kinds(Jets, oracle, dt, nrow):
    """Zero, identity, scalar and adjointed rows inside the chained walk (the MIXED instantiation of k_tall_diag_bidiag_chain):
    the bits of the oracle's unfused sequence, whole-vector and in two ranges -- e.g. LSQR on [A; lambda*I]."""
    from jets_jl_amd._ffi import lib, check
    from .test_gpu_mixed_rows import KINDS, _build

    J = Jets
    n = 4096
    rs = np.random.RandomState(300 + nrow)
    kinds = [KINDS[k] for k in rs.randint(0, len(KINDS), size=nrow)]
    kinds[rs.randint(nrow)] = "zero"
    kinds[-1] = "scale"                                        # the regularisation row sits in the ragged last chunk
    A, ops = _build(J, oracle, dt, kinds, n, seed=70 + nrow)
    nat = _native(J, A)
    hv = u01(oracle, dt, 2, 0, n)
    hu = [u01(oracle, dt, 3, i, n) for i in range(nrow)]
    v = J.from_numpy(hv)
    out = C.c_double(0)
    for alpha, beta in ((0.75, -0.5), (1.0, 0.0)):
        av = oracle.block_df(ops, [np.zeros(n, dt) for _ in range(nrow)], [hv])
        want_u = oracle.barr_lincomb([np.empty(n, dt) for _ in range(nrow)], [alpha, beta], [av, hu]) if beta else \
            oracle.barr_lincomb([np.empty(n, dt) for _ in range(nrow)], [alpha], [av])
        want_w = oracle.block_df_adj(ops, [np.zeros(n, dt)], want_u)[0]
        nrm = float(sum(np.vdot(b.astype(np.complex128), b.astype(np.complex128)).real for b in want_u))
        try:
            J.tune(step_chain=1)
            u = J.from_numpy(np.concatenate(hu), J.range(A))
            w = J.rand(J.domain(A), seed=9, stream=0)
            check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, alpha, beta, C.byref(out)))
            assert J.tune_get("last_step_chain") == (nrow + 7) // 8
            assert_bits_equal(u.to_numpy(), np.concatenate(want_u), f"chained mixed step: u {kinds}")
            assert_bits_equal(w.to_numpy().ravel(order="F"), want_w, f"chained mixed step: w {kinds}")
            assert abs(out.value - nrm) <= 1e-12 * nrm
            u2 = J.from_numpy(np.concatenate(hu), J.range(A))
            w2 = J.rand(J.domain(A), seed=10, stream=0)
            check(lib.jh_normsq_reset())
            half = n // 2
            check(lib.jh_blockop_bidiag_step_range(nat.handle, u2.handle, v.handle, w2.handle, alpha, beta, 0, half, None))
            check(lib.jh_blockop_bidiag_step_range(nat.handle, u2.handle, v.handle, w2.handle, alpha, beta, half, n - half, None))
            check(lib.jh_normsq_read(C.byref(out)))
            assert J.tune_get("last_step_chain") == (nrow + 7) // 8
            assert_bits_equal(u2.to_numpy(), np.concatenate(want_u), "chained mixed ranged step: u")
            assert_bits_equal(w2.to_numpy().ravel(order="F"), want_w, "chained mixed ranged step: w")
            assert abs(out.value - nrm) <= 1e-12 * nrm
        finally:
            J.tune(step_chain=-1)


def test_step_mode_is_measured_per_operator_and_every_mode_has_the_same_bits(Jets, oracle):
    """40 x 256^3 Float32 (rows of 64 MiB: the chained walk is a candidate): the first seven calls each try one mode (plain,
    XCD-contiguous tiles, chained), then the choice is kept and can be exported / imported; slices of u and w vs the oracle under
    every mode; 30 LSQR iterations through the native loop."""
    from jets_jl_amd._ffi import lib, check

    J = Jets
    dt, nrow, edge = np.float32, 40, 256
    n = edge ** 3
    spc = J.JetSpace(dt, edge, edge, edge)
    coeff = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    nat = _native(J, A)
    v = J.rand(J.domain(A), seed=2, stream=0)
    w = J.zeros(J.domain(A))
    out = C.c_double(0)
    W = 4096

    def check_slices(u, what):
        for off in (0, n // 2 + 64, n - W):
            hv = oracle.rng_u01(dt, 2, 0, off, W)
            ha = [oracle.rng_u01(dt, 1, 0, i * n + off, W) for i in range(nrow)]
            hu = [oracle.rng_u01(dt, 3, 0, i * n + off, W) for i in range(nrow)]
            want_u = [np.float32(0.75) * (a * hv) + np.float32(-1.375) * uu for a, uu in zip(ha, hu)]
            for i in (0, 7, 8, 39):
                assert_bits_equal(u._download(i * n + off, W), want_u[i], f"{what}: u row {i} slice at {off}")
            want_w = oracle.block_df_adj([[oracle.Block("diag", W, coeff=a)] for a in ha], [np.zeros(W, dt)], want_u)[0]
            assert_bits_equal(w._download(off, W), want_w, f"{what}: w slice at {off}")

    for mode, chunks in ((0, 0), (1, 0), (2, 5)):
        J.op_tune_set(A, "step_mode", mode)
        u = J.rand(J.range(A), seed=3, stream=0)
        check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, 0.75, -1.375, C.byref(out)))
        assert J.tune_get("last_step_chain") == chunks
        check_slices(u, f"mode {mode}")
    # the chained walk again and again on the same buffers (hand-off buffers and flags are re-used from launch to launch and, in
    # the pipelined multi-GPU form, from range to range): every repetition must have the bits of the ordered walk
    J.op_tune_set(A, "step_mode", 2)
    f32 = np.float32
    for rep, (alpha, beta, vseed) in enumerate([(1.0, -0.5, 21), (0.5, 0.25, 22), (-1.25, 1.0, 23), (1.0, 0.0, 24)]):
        v2 = J.rand(J.domain(A), seed=vseed, stream=0)
        u = J.rand(J.range(A), seed=3, stream=rep)
        check(lib.jh_normsq_reset())
        q = n // 4
        for r in range(4):                                        # four ranges, like the pipelined exchange
            check(lib.jh_blockop_bidiag_step_range(nat.handle, u.handle, v2.handle, w.handle, alpha, beta, r * q, q, None))
        check(lib.jh_normsq_read(C.byref(out)))
        assert J.tune_get("last_step_chain") == 5
        for off in (0, q - W, q, 2 * q + 8192, n - W):
            hv = oracle.rng_u01(dt, vseed, 0, off, W)
            ha = [oracle.rng_u01(dt, 1, 0, i * n + off, W) for i in range(nrow)]
            hu = [oracle.rng_u01(dt, 3, rep, i * n + off, W) for i in range(nrow)]
            want_u = [f32(alpha) * (a * hv) + f32(beta) * uu if beta else f32(alpha) * (a * hv) for a, uu in zip(ha, hu)]
            assert_bits_equal(u._download(17 * n + off, W), want_u[17], f"repetition {rep}: u row 17 slice at {off}")
            want_w = oracle.block_df_adj([[oracle.Block("diag", W, coeff=a)] for a in ha], [np.zeros(W, dt)], want_u)[0]
            assert_bits_equal(w._download(off, W), want_w, f"repetition {rep}: w slice at {off}")
    J.op_tune_set(A, "step_mode", -1)                             # measure: 1 warm-up + 3 modes x 2 passes, one per real call
    seen = set()
    for _ in range(10):
        check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, 1.0, -0.5, C.byref(out)))
        seen.add(J.tune_get("last_step_chain"))
    assert seen == {0, 5}, "the trials ran both the plain and the chained walk"
    assert J.op_tune_get(A, "step_mode") in (0, 1, 2) and J.op_tune_get(A, "step_trials") == 7
    x_true = J.rand(J.domain(A), seed=4, stream=0)
    res = J.lsqr(A, A * x_true, atol=0.0, btol=0.0, conlim=0.0, maxiter=30)
    err = (res.x - x_true).materialize()
    assert float(J.norm(err)) / float(J.norm(x_true)) < 1e-4


@pytest.mark.parametrize("chunk", [8, 0])
def test_chained_ranged_step_beside_a_busy_second_stream(Jets, chunk):
    """The pipelined distributed iteration runs the ranged chained step while RCCL's reduce kernels of the previous range occupy
    part of the same device.  Stand-in: a second context (own stream) of this device streams 768 MiB triads without pause while
    the chained step runs in four ranges; u, w keep the bits of the plain walk and no hand-off poll expires (jh_normsq_read
    turns the sticky error word into a failed call).  tools/soak_step_chain.py --ranged --beside is the long form."""
    from jets_jl_amd import jetblock as _blk
    from jets_jl_amd._ffi import check, lib

    J = Jets
    home = J.context_current()[0]
    nrow, edge = 48, 128                                               # 8 MiB rows: chained only because the knob says so
    chunks, lanes = (6, 1024) if chunk == 8 else (2, 256)              # 8-row chunks of 1024-lane tiles (rounds 2-4) / round 5's 32-row chunks of 256-lane tiles
    n = edge ** 3
    spc = J.JetSpace(np.float32, edge, edge, edge)
    coeff = J.rand(J.JetBSpace([spc] * nrow), seed=61, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    nat = _blk._tall_native(A)
    u1, u2 = J.rand(J.range(A), seed=3, stream=0), J.rand(J.range(A), seed=3, stream=0)
    v = J.rand(spc, seed=2, stream=0)
    w1, w2 = J.zeros(spc), J.zeros(spc)
    o1, o2 = C.c_double(0), C.c_double(0)
    other = J.context_create(0)
    try:
        big = J.JetSpace(np.float32, 64 * 1024 * 1024)
        with J.using_context(other):
            nx, ny, nz = J.rand(big, seed=7, stream=0), J.rand(big, seed=8, stream=0), J.zeros(big)
            marks = [J.Event() for _ in range(4)]
        J.context_use(home)
        handoffs = 0
        for k in range(120):
            alpha, beta = (1.0, -0.5) if k % 3 else (0.75, 0.25)
            e = marks[k % 4]
            if k >= 4:
                e.elapsed_ms(e)                                        # never more than four batches of noise ahead
            for _ in range(4):
                J.lincomb_(nz, [0.5, 0.25], [nx, ny])                  # the other context's stream
            e.record()
            J.context_use(home)
            J.tune(step_chain=0)
            check(lib.jh_blockop_bidiag_step(nat.handle, u1.handle, v.handle, w1.handle, alpha, beta, C.byref(o1)))
            J.tune(step_chain=1, step_chunk=chunk)
            check(lib.jh_normsq_reset())
            q = n // 4
            for r in range(4):
                check(lib.jh_blockop_bidiag_step_range(nat.handle, u2.handle, v.handle, w2.handle, alpha, beta, r * q, q, None))
                assert J.tune_get("last_step_chain") == chunks
            check(lib.jh_normsq_read(C.byref(o2)))                     # an expired poll would fail here
            handoffs += 4 * chunks * (q // (4 * lanes))
            assert abs(o1.value - o2.value) <= 1e-12 * o1.value
            if k % 40 == 39:
                assert_bits_equal(w2.to_numpy(), w1.to_numpy(), f"step {k}: w beside the busy stream")
                assert_bits_equal(u2.to_numpy(), u1.to_numpy(), f"step {k}: u beside the busy stream")
        assert handoffs >= 300_000
        with J.using_context(other):
            J.synchronize()
            del nx, ny, nz, marks, e
    finally:
        J.tune(step_chain=-1, step_chunk=0)
        import gc

        gc.collect()
        J.context_use(home)
        J.context_destroy(other)


def test_a_step_that_asks_for_no_norm_never_chains(Jets):
    """An expired hand-off poll is reported by the call that reads ||u||^2 back; a whole-vector step with normsq = NULL has no
    such reader, so it keeps the plain walk even when the knob forces chaining."""
    from jets_jl_amd import jetblock as _blk
    from jets_jl_amd._ffi import check, lib

    J = Jets
    spc = J.JetSpace(np.float32, 4096)
    coeff = J.rand(J.JetBSpace([spc] * 32), seed=5, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    nat = _blk._tall_native(A)
    u, v, w = J.rand(J.range(A), seed=3, stream=0), J.rand(spc, seed=2, stream=0), J.zeros(spc)
    out = C.c_double(0)
    try:
        J.tune(step_chain=1)
        check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, 1.0, -0.5, C.byref(out)))
        assert J.tune_get("last_step_chain") == 4
        check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, 1.0, -0.5, None))
        assert J.tune_get("last_step_chain") == 0
    finally:
        J.tune(step_chain=-1)
