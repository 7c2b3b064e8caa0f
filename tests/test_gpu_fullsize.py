"""GPU parity at BASELINE.json's full sizes.

configs[1]  (64x1, 128^3 Float32): the whole forward and adjoint are compared bit for bit with the
            CPU oracle (512 MiB per vector; the oracle finishes in seconds).
configs[2]  (A' o A on a 256x1 tall operator, 128^3 and 256^3 blocks): fused kernel vs oracle slices, vs the unfused chain on
            the whole vector, and <m, A'A m> = ||A m||^2.
configs[3]  (1024x1, 256^3 Float32; 64 GiB of coefficients + 64 GiB range vector, cannot be held on a
            host): inputs are generated on the device by the counter-based generator and checked
            through size-independent properties --
              * slices of the forward / adjoint / fused-normal results vs the oracle run on the
                regenerated slices of the inputs (bit-exact; the adjoint slice sums all 1024 rows in order),
              * the dot-product test (src/Jets.jl:1211-1226) at |lhs-rhs|/|lhs+rhs| < 1e-5,
              * fused A'A == chained A'(A m) on the whole vector (max |difference| == 0),
              * norm/dot of the 2^34-element range vector vs closed-form expectations of U[0,1).
"""
import math

import numpy as np
import pytest

from .helpers import assert_bits_equal

pytestmark = pytest.mark.gpu


def _build(Jets, nblocks, edge):
    blk = Jets.JetSpace(np.float32, edge, edge, edge)
    coeff = Jets.rand(Jets.JetBSpace([blk] * nblocks), seed=1, stream=0)
    A = Jets.blockop([[Jets.JopDiagonal(c)] for c in coeff.arrays])
    m = Jets.rand(Jets.domain(A), seed=2, stream=0)
    d = Jets.rand(Jets.range(A), seed=3, stream=0)
    return A, coeff, m, d


def test_config2_64x128cubed_whole_vectors_bit_exact(Jets, oracle):
    nblocks, edge = 64, 128
    n = edge ** 3
    A, coeff, m, d = _build(Jets, nblocks, edge)
    ha = [oracle.rng_u01(np.float32, 1, 0, i * n, n) for i in range(nblocks)]
    hm = oracle.rng_u01(np.float32, 2, 0, 0, n)
    hd = [oracle.rng_u01(np.float32, 3, 0, i * n, n) for i in range(nblocks)]
    ops = [[oracle.Block("diag", n, coeff=g)] for g in ha]
    mt = Jets.mul_(Jets.zeros(Jets.domain(A)), A.H, d)
    ref_m = oracle.block_df_adj(ops, [np.zeros(n, dtype=np.float32)], hd)
    assert_bits_equal(mt.to_numpy().ravel(order="F"), ref_m[0], "adjoint, 64 x 128^3")
    out = Jets.mul_(d, A, m)
    ref_d = oracle.block_df(ops, hd, [hm])                                  # overwrites hd in place
    assert_bits_equal(out.to_numpy(), np.concatenate(ref_d), "forward, 64 x 128^3")
    lhs, rhs = Jets.dot_product_test(A, m, Jets.rand(Jets.range(A), seed=3, stream=1))
    assert abs(lhs - rhs) / abs(lhs + rhs) < 1e-5


@pytest.mark.parametrize("edge", [128, 256])
def test_config3_normal_equations_matvec_on_256x1(Jets, oracle, edge):
    """BASELINE.json configs[2]: JetComposite A' o A on a 256 x 1 tall JopBlock (src/Jets.jl:530-534 over (A', A)), at its own
    size with both block sizes SURVEY.md 8d names (128^3: 2 GiB of coefficients, 256^3: 16 GiB).  The fused kernel against
      * the oracle's normal_df on regenerated slices of the inputs -- all 256 rows summed in order, bit for bit,
      * the unfused chain A'(A m) on the device, on the WHOLE vector (max |difference| == 0),
      * the reductions the solver takes of it (<m, A'A m> = ||A m||^2 to 1e-5)."""
    import math

    nblocks = 256
    n = edge ** 3
    A, coeff, m, d = _build(Jets, nblocks, edge)
    C = A.H @ A
    y = Jets.mul(C, m)
    W = 4096
    for off in (0, (n // 3) // 4 * 4, n // 2 + 64, n - W):
        ha = [oracle.rng_u01(np.float32, 1, 0, i * n + off, W) for i in range(nblocks)]
        hm = oracle.rng_u01(np.float32, 2, 0, off, W)
        ref = oracle.normal_df([[oracle.Block("diag", W, coeff=g)] for g in ha], [np.zeros(W, dtype=np.float32)], [hm])[0]
        assert_bits_equal(y._download(off, W), ref, f"fused A'A slice at {off}, 256 x {edge}^3")
    Jets.mul_(d, A, m)                                                     # the chain through the range vector
    y_chain = Jets.mul_(Jets.zeros(Jets.domain(A)), A.H, d)
    assert float(Jets.norm((y - y_chain).materialize(), math.inf)) == 0.0  # bit-identical on every element
    lhs, rhs = float(Jets.dot(m, y)), float(Jets.norm(d, 2)) ** 2          # <m, A'A m> == ||A m||^2
    assert abs(lhs - rhs) <= 1e-5 * abs(rhs)
    # composite plumbing at this size: the adjoint of the composite is the same operator
    y2 = Jets.mul(C.H, m)
    assert float(Jets.norm((y - y2).materialize(), math.inf)) == 0.0


def test_config4_1024x256cubed_properties(Jets, oracle):
    info = Jets.device_info()
    if info["free_mem"] < 200 * 2 ** 30:
        pytest.skip(f"needs ~195 GiB of free HBM (coefficients, range vector and a second range vector), device reports {info['free_mem'] / 2**30:.0f} GiB")
    nblocks, edge = 1024, 256
    n = edge ** 3
    A, coeff, m, d = _build(Jets, nblocks, edge)
    assert Jets.range(A).length() == 2 ** 34

    # --- reductions over 2^34 elements of U[0,1): E[x] = 1/2, E[x^2] = 1/3, E[xy] = 1/4
    N = float(2 ** 34)
    assert float(Jets.norm(d, 1)) == pytest.approx(N / 2, rel=1e-4)
    assert float(Jets.norm(d, 2)) == pytest.approx(math.sqrt(N / 3), rel=1e-4)
    assert float(Jets.dot(d, coeff)) == pytest.approx(N / 4, rel=1e-4)
    assert float(Jets.norm(d, math.inf)) < 1.0 and Jets.extrema(d)[0] >= 0.0

    # --- adjoint slices: all 1024 rows summed in order, bit-exact vs the oracle on regenerated slices
    mt = Jets.mul_(Jets.zeros(Jets.domain(A)), A.H, d)
    W = 4096
    for off in (0, 12345 * 4, n // 2 + 64, n - W):
        ha = [oracle.rng_u01(np.float32, 1, 0, i * n + off, W) for i in range(nblocks)]
        hd = [oracle.rng_u01(np.float32, 3, 0, i * n + off, W) for i in range(nblocks)]
        ops = [[oracle.Block("diag", W, coeff=g)] for g in ha]
        ref = oracle.block_df_adj(ops, [np.zeros(W, dtype=np.float32)], hd)[0]
        assert_bits_equal(mt._download(off, W), ref, f"adjoint slice at {off}")

    # --- dot-product test with the SAME d (before the forward overwrites it)
    md = float(Jets.dot(m, mt))                                              # <m, A'd>

    # --- fused A'A == chained, whole vector, and a slice vs the oracle
    y_fused = Jets.mul(A.H @ A, m)
    hm0 = oracle.rng_u01(np.float32, 2, 0, 0, W)
    ha0 = [oracle.rng_u01(np.float32, 1, 0, i * n, W) for i in range(nblocks)]
    ref_y = oracle.normal_df([[oracle.Block("diag", W, coeff=g)] for g in ha0], [np.zeros(W, dtype=np.float32)], [hm0])[0]
    assert_bits_equal(y_fused._download(0, W), ref_y, "fused normal slice")

    # --- forward (overwrites d), slices of several rows vs the oracle
    dcopy_first = d._download(0, W)                                          # to prove the forward overwrote it
    Jets.mul_(d, A, m)
    for i in (0, 1, 511, 1023):
        for off in (0, n // 3 // 4 * 4, n - W):
            ha = oracle.rng_u01(np.float32, 1, 0, i * n + off, W)
            hm = oracle.rng_u01(np.float32, 2, 0, off, W)
            assert_bits_equal(d._download(i * n + off, W), ha * hm, f"forward row {i} slice at {off}")
    assert not np.array_equal(d._download(0, W), dcopy_first)
    y_chain = Jets.mul_(Jets.zeros(Jets.domain(A)), A.H, d)
    diff = (y_fused - y_chain).materialize()
    assert float(Jets.norm(diff, math.inf)) == 0.0                           # bit-identical on all 2^24 elements

    # --- dot-product test: <A m, d0> == <m, A' d0>, with d0 regenerated into the coefficient-free spare
    d0 = Jets.rand(Jets.range(A), seed=3, stream=0)                          # needs a third 64 GiB slab
    lhs = float(Jets.dot(d, d0))                                             # <A m, d0>
    assert abs(lhs - md) / abs(lhs + md) < 1e-5
    del d0

    # --- JIT broadcast over 2^34 elements (2^32 packs: beyond one lane per pack, the generated kernel strides), in place
    Jets.broadcast_(d, "s0*x0 + x0*x0", [d], [3.0])
    for i in (0, 511, 1023):
        for off in (0, n - W):
            t = oracle.rng_u01(np.float32, 1, 0, i * n + off, W) * oracle.rng_u01(np.float32, 2, 0, off, W)   # d_i = a_i .* m
            assert_bits_equal(d._download(i * n + off, W), np.float32(3.0) * t + t * t, f"broadcast row {i} slice at {off}")
    Jets.mul_(d, A, m)                                                       # back to d = A m for the step below

    # --- one-pass Golub-Kahan step on the full-size vectors (d = A m at this point): u <- 0.75*(A m) - 1.375*u, w <- A'u
    import ctypes as C

    from jets_jl_amd._ffi import lib, check
    from jets_jl_amd import jetblock

    nat = jetblock._native_op(A.jet.s["_native"], A.jet.s["ops"], A.jet.rng.eltype())
    w, out = Jets.zeros(Jets.domain(A)), C.c_double(0)
    check(lib.jh_blockop_bidiag_step(nat.handle, d.handle, m.handle, w.handle, 0.75, -1.375, C.byref(out)))
    f32 = np.float32
    for off in (0, n // 2 + 64, n - W):
        hm = oracle.rng_u01(f32, 2, 0, off, W)
        ha = [oracle.rng_u01(f32, 1, 0, i * n + off, W) for i in range(nblocks)]
        hu = [f32(0.75) * (g * hm) + f32(-1.375) * (g * hm) for g in ha]     # product, scale, scale, add: each rounded
        for i in (0, 700, 1023):
            assert_bits_equal(d._download(i * n + off, W), hu[i], f"step: u row {i} slice at {off}")
        ref_w = oracle.block_df_adj([[oracle.Block("diag", W, coeff=g)] for g in ha], [np.zeros(W, dtype=f32)], hu)[0]
        assert_bits_equal(w._download(off, W), ref_w, f"step: w slice at {off}")
    assert out.value == pytest.approx(N * 0.625 ** 2 / 9, rel=1e-3)          # E[(0.625 a m)^2] = 0.625^2 / 9


def test_config5_100_lsqr_iterations_on_1024x256cubed(Jets, oracle):
    """BASELINE.json configs[4]: 100-iteration LSQR on the 1024 x 256^3 operator, b = A x_true.
    fp tolerance: rel l2 error of x vs x_true <= 1e-4 (Float32 data; the fp64 CPU LSQR of
    oracle/lsqr_ref.py reaches the same on a 4096-element slice of the problem, which is separable)."""
    info = Jets.device_info()
    if info["free_mem"] < 140 * 2 ** 30:
        pytest.skip(f"needs ~130 GiB of free HBM, device reports {info['free_mem'] / 2**30:.0f} GiB")
    from oracle.lsqr_ref import lsqr_fp64

    nblocks, edge = 1024, 256
    n = edge ** 3
    blk = Jets.JetSpace(np.float32, edge, edge, edge)
    coeff = Jets.rand(Jets.JetBSpace([blk] * nblocks), seed=1, stream=0)
    A = Jets.blockop([[Jets.JopDiagonal(c)] for c in coeff.arrays])
    x_true = Jets.rand(Jets.domain(A), seed=4, stream=0)
    b = A * x_true
    res = Jets.lsqr(A, b, atol=0.0, btol=0.0, conlim=0.0, maxiter=100, overwrite_b=True)
    assert res.itn <= 100 and res.istop in (4, 5, 6, 7)       # runs to machine precision (tolerances are zero) or 100
    err = (res.x - x_true).materialize()
    rel = float(Jets.norm(err)) / float(Jets.norm(x_true))
    assert rel <= 1e-4, rel
    r = [h[1] for h in res.history]
    assert r[5] < 1e-3 * r[0] or r[0] < 1e-3 * float(Jets.norm(x_true))       # converges within a few iterations
    # the problem is separable per element: a W-element slice solved by the fp64 CPU LSQR must agree
    W = 4096
    ha = np.stack([oracle.rng_u01(np.float32, 1, 0, i * n, W) for i in range(nblocks)]).astype(np.float64)
    hx = oracle.rng_u01(np.float32, 4, 0, 0, W).astype(np.float64)
    hb = (ha.astype(np.float32) * hx.astype(np.float32)).astype(np.float64)   # b as the device computed it (one rounded product)
    xr, _ = lsqr_fp64(lambda v: (ha * v).ravel(), lambda y: (ha * y.reshape(nblocks, W)).sum(0), hb.ravel(), W,
                      atol=0.0, btol=0.0, conlim=0.0, maxiter=30)
    got = res.x._download(0, W).astype(np.float64)
    assert np.linalg.norm(got - xr) / np.linalg.norm(xr) <= 1e-4
    # round 3: the same solve by CGLS (two passes per iteration) and by CG through the fused A'A (one pass of the coefficients),
    # on the operator already resident; both must reach x_true, and their recorded ||r|| must be the residual's real norm
    for name, solve, tol in (("cgls", lambda rhs: Jets.cgls(A, rhs, atol=0.0, btol=0.0, maxiter=30, overwrite_b=True), 1e-4),
                             ("cgnr", lambda rhs: Jets.cgnr(A, rhs, atol=0.0, btol=0.0, maxiter=30), 1e-4)):
        Jets.mul_(b, A, x_true)                                   # LSQR / CGLS consumed b's storage
        bnorm = float(Jets.norm(b))
        out = solve(b)
        err = (out.x - x_true).materialize()
        rel = float(Jets.norm(err)) / float(Jets.norm(x_true))
        assert rel <= tol, (name, rel)
        assert 1 <= out.itn <= 30 and out.istop in (1, 2, 6, 7), (name, out.itn, out.istop)
        assert out.history[0][1] < 0.5 * bnorm and out.history[min(5, out.itn - 1)][1] < 1e-2 * bnorm, (name, out.history[:6])
        got = out.x._download(0, W).astype(np.float64)
        assert np.linalg.norm(got - xr) / np.linalg.norm(xr) <= 1e-4, name


def test_headline_size_with_an_odd_edge_1024x255cubed_properties(Jets, oracle):
    """The headline operator with an ODD edge -- 1024 x 1 of 255^3 Float32 blocks (63 GiB of coefficients + 63 GiB range vector): three rows in four start off a
    16-byte boundary of their slab and every row ends inside a 16-byte pack (round 5, last session: under-aligned packs).  The same
    size-independent properties as config 4: slices of the adjoint (all 1024 rows in order), of the fused A'A, of the forward and of the one-pass step against
    the oracle on regenerated slices -- including the LAST elements of rows and of the domain, where the partial packs are -- bit for bit; fused == chained
    on the whole vector; the dot-product test; ||u||^2 counted once per scalar."""
    import ctypes as C

    from jets_jl_amd import jetblock
    from jets_jl_amd._ffi import check, lib

    info = Jets.device_info()
    if info["free_mem"] < 200 * 2 ** 30:
        pytest.skip(f"needs ~195 GiB of free HBM, device reports {info['free_mem'] / 2**30:.0f} GiB")
    nblocks, edge = 1024, 255
    n = edge ** 3
    assert (n * 4) % 16 != 0
    A, coeff, m, d = _build(Jets, nblocks, edge)
    f32 = np.float32
    W = 4096
    offs = (0, 12345 * 4 + 1, n // 2 + 63, n - W)                                # the last slice ends with the row: the partial pack
    mt = Jets.mul_(Jets.zeros(Jets.domain(A)), A.H, d)
    for off in offs:
        ha = [oracle.rng_u01(f32, 1, 0, i * n + off, W) for i in range(nblocks)]
        hd = [oracle.rng_u01(f32, 3, 0, i * n + off, W) for i in range(nblocks)]
        ref = oracle.block_df_adj([[oracle.Block("diag", W, coeff=g)] for g in ha], [np.zeros(W, dtype=f32)], hd)[0]
        assert_bits_equal(mt._download(off, W), ref, f"adjoint slice at {off}")
    md = float(Jets.dot(m, mt))
    y_fused = Jets.mul(A.H @ A, m)
    for off in (0, n - W):
        hm0 = oracle.rng_u01(f32, 2, 0, off, W)
        ha0 = [oracle.rng_u01(f32, 1, 0, i * n + off, W) for i in range(nblocks)]
        ref_y = oracle.normal_df([[oracle.Block("diag", W, coeff=g)] for g in ha0], [np.zeros(W, dtype=f32)], [hm0])[0]
        assert_bits_equal(y_fused._download(off, W), ref_y, f"fused normal slice at {off}")
    Jets.mul_(d, A, m)
    for i in (0, 1, 2, 3, 511, 1023):                                          # rows 1 .. 3: the three misalignments
        for off in (0, n // 3, n - W):
            ha = oracle.rng_u01(f32, 1, 0, i * n + off, W)
            hm = oracle.rng_u01(f32, 2, 0, off, W)
            assert_bits_equal(d._download(i * n + off, W), ha * hm, f"forward row {i} slice at {off}")
    y_chain = Jets.mul_(Jets.zeros(Jets.domain(A)), A.H, d)
    diff = (y_fused - y_chain).materialize()
    assert float(Jets.norm(diff, math.inf)) == 0.0
    d0 = Jets.rand(Jets.range(A), seed=3, stream=0)
    lhs = float(Jets.dot(d, d0))
    assert abs(lhs - md) / abs(lhs + md) < 1e-5
    del d0
    nat = jetblock._native_op(A.jet.s["_native"], A.jet.s["ops"], A.jet.rng.eltype())
    w, out = Jets.zeros(Jets.domain(A)), C.c_double(0)
    check(lib.jh_blockop_bidiag_step(nat.handle, d.handle, m.handle, w.handle, 0.75, -1.375, C.byref(out)))
    for off in (0, n // 2 + 63, n - W):
        hm = oracle.rng_u01(f32, 2, 0, off, W)
        ha = [oracle.rng_u01(f32, 1, 0, i * n + off, W) for i in range(nblocks)]
        hu = [f32(0.75) * (g * hm) + f32(-1.375) * (g * hm) for g in ha]
        for i in (0, 1, 700, 1023):
            assert_bits_equal(d._download(i * n + off, W), hu[i], f"step: u row {i} slice at {off}")
        ref_w = oracle.block_df_adj([[oracle.Block("diag", W, coeff=g)] for g in ha], [np.zeros(W, dtype=f32)], hu)[0]
        assert_bits_equal(w._download(off, W), ref_w, f"step: w slice at {off}")
    N = float(nblocks) * n
    assert out.value == pytest.approx(N * 0.625 ** 2 / 9, rel=1e-3)
