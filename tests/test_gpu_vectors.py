"""GPU parity: device BlockArray (one HIP slab) vs the CPU oracle, through the C ABI.

Re-encodes the reference's block-array test sets on seeded inputs:
  test/runtests.jl:512-551 "block arrays", 553-600 "block arrays, broadcasting",
  602-620 "block array, reshaped from array".
Bar: bit-exact for layout / fill / copies / random fill / broadcast; reductions within
1e-5 (Float32, ComplexF32) or 1e-12 (Float64, ComplexF64) relative to an fp64 host value
(tighter than the reference's own `isapprox` default rtol = sqrt(eps)).
"""
import math

import numpy as np
import pytest

from .helpers import DTYPES, assert_bits_equal, dev_blocks_to_numpy, u01

pytestmark = pytest.mark.gpu

RAGGED = [(2,), (2, 2), (2, 3)]                    # test/runtests.jl:513
SHAPES_MIX = [(5,), (1,), (7, 3), (4, 4, 4), (129,), (1000, 3)]   # odd sizes: unaligned block starts


def _tol(dt):
    return 1e-5 if np.dtype(dt) in (np.dtype(np.float32), np.dtype(np.complex64)) else 1e-12


def _bspace(J, dt, shapes):
    return J.JetBSpace([J.JetSpace(dt, *s) for s in shapes])


@pytest.mark.parametrize("dt", DTYPES)
def test_layout_matches_reference_ranges(Jets, oracle, dt):
    R = _bspace(Jets, dt, SHAPES_MIX)
    x = Jets.zeros(R)
    lens = [int(np.prod(s)) for s in SHAPES_MIX]
    start1, stop1 = oracle.bspace_indices(lens)            # 1-based inclusive (src/Jets.jl:742-748)
    for i in range(len(lens)):
        assert Jets.indices(x, i) == range(start1[i] - 1, stop1[i])
        assert Jets.indices(R, i) == range(start1[i] - 1, stop1[i])
    assert x.length() == stop1[-1] == R.length()
    assert Jets.nblocks(x) == len(lens)
    assert Jets.space(x) == R
    assert np.all(x.to_numpy() == 0)


def test_pi_fill_literals(Jets):
    """test/runtests.jl:513-526 (literal values)."""
    R = _bspace(Jets, np.float64, RAGGED)
    x = Jets.ones(R)
    assert np.array_equal(Jets.getblock(x, 0).to_numpy(), np.ones(2))
    assert np.array_equal(Jets.getblock(x, 1).to_numpy(), np.ones((2, 2)))
    assert np.array_equal(Jets.getblock(x, 2).to_numpy(), np.ones((2, 3)))
    Jets.setblock_(x, 0, math.pi)
    Jets.setblock_(x, 1, 2 * math.pi)
    Jets.setblock_(x, 2, 3 * math.pi * np.ones((2, 3)))
    out = Jets.getblock_(x, 1, np.empty((2, 2)))
    assert np.array_equal(out, 2 * math.pi * np.ones((2, 2)))
    _x = Jets.convert_array(x).to_numpy()
    assert np.array_equal(_x, np.concatenate([np.full(2, math.pi), np.full(4, 2 * math.pi), np.full(6, 3 * math.pi)]))
    assert np.array_equal(_x, x.to_numpy())
    assert Jets.norm(x) == pytest.approx(np.linalg.norm(_x), rel=1e-14)
    assert Jets.norm(x, 0) == np.count_nonzero(_x)
    assert Jets.norm(x, math.inf) == 3 * math.pi


@pytest.mark.parametrize("dt", DTYPES)
def test_rand_is_the_oracle_stream_bit_exact(Jets, oracle, dt):
    R = _bspace(Jets, dt, SHAPES_MIX)
    x = Jets.rand(R, seed=7, stream=11)
    assert_bits_equal(x.to_numpy(), u01(oracle, dt, 7, 11, R.length()), "rand slab")
    # a shard regenerates its slice of the global vector (index_base)
    y = Jets.rand(Jets.JetSpace(dt, 1000), seed=7, stream=11, index_base=37)
    assert_bits_equal(y.to_numpy(), u01(oracle, dt, 7, 11, 1000, index0=37), "rand slice")
    for n, base in ((1003, 37), (1001, 38), (3, 5), (70001, 2 ** 33 + 1)):        # odd / even bases, tails that are not whole 16-byte packs, a far offset
        z = Jets.rand(Jets.JetSpace(dt, n), seed=7, stream=11, index_base=base)
        assert_bits_equal(z.to_numpy(), u01(oracle, dt, 7, 11, n, index0=base), f"rand slice of {n} from {base}")
    v = x.to_numpy()
    assert (v.real >= 0).all() and (v.real < 1).all()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("p", [2, 1, 0, math.inf, -math.inf, 3, 2.5])
def test_norm(Jets, oracle, dt, p):
    R = _bspace(Jets, dt, SHAPES_MIX)
    x = Jets.rand(R, seed=3, stream=1)
    if p == 0:
        Jets.setblock_(x, 1, 0.0)                           # some exact zeros to count
    blocks = dev_blocks_to_numpy(x)
    got = float(Jets.norm(x, p))
    flat = np.concatenate(blocks).astype(np.complex128)
    truth = float(np.linalg.norm(flat, p)) if p not in (0,) else float(np.count_nonzero(flat))
    assert got == pytest.approx(truth, rel=_tol(dt))
    assert got == pytest.approx(oracle.barr_norm(blocks, float(p)), rel=10 * _tol(dt) if np.dtype(dt).itemsize <= 8 else 1e-10)


@pytest.mark.parametrize("dt", DTYPES)
def test_dot(Jets, oracle, dt):
    R = _bspace(Jets, dt, SHAPES_MIX)
    x, y = Jets.rand(R, seed=3, stream=1), Jets.rand(R, seed=3, stream=2)
    bx, by = dev_blocks_to_numpy(x), dev_blocks_to_numpy(y)
    truth = np.vdot(np.concatenate(bx).astype(np.complex128), np.concatenate(by).astype(np.complex128))
    got = complex(Jets.dot(x, y))
    assert abs(got - truth) <= _tol(dt) * abs(truth)
    ora = complex(oracle.barr_dot(bx, by))
    assert abs(got - ora) <= 20 * _tol(dt) * abs(truth)
    # dot(x,x) ~ dot(_x,_x)  (test/runtests.jl:550), result type follows eltype
    assert type(Jets.dot(x, x)) == np.dtype(dt).type


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_extrema(Jets, oracle, dt):
    R = _bspace(Jets, dt, SHAPES_MIX)
    x = Jets.rand(R, seed=9, stream=4)
    x.assign(2.0 * x - 1.0 * Jets.ones(R))                  # signed values
    blocks = dev_blocks_to_numpy(x)
    mn, mx = Jets.extrema(x)
    omn, omx = oracle.barr_extrema(blocks)
    assert (float(mn), float(mx)) == (omn, omx)             # exact: compares only
    flat = np.concatenate(blocks)
    assert (mn, mx) == (flat.min(), flat.max())


@pytest.mark.parametrize("dt", DTYPES)
def test_fill_and_setblock_bit_exact(Jets, oracle, dt):
    R = _bspace(Jets, dt, SHAPES_MIX)
    x = Jets.rand(R, seed=1, stream=1)
    val = 3.14 if np.dtype(dt).kind != "c" else 3.14 - 2.5j
    Jets.fill_(x, val)                                      # x .= 3.14  (test/runtests.jl:584)
    ref = oracle.barr_fill([np.empty(int(np.prod(s)), dtype=dt) for s in SHAPES_MIX], val)
    assert_bits_equal(x.to_numpy(), np.concatenate(ref), "fill!")
    for i, s in enumerate(SHAPES_MIX):                      # scalar setblock! on every (unaligned) block
        Jets.setblock_(x, i, float(i) + 0.5)
    ref = np.concatenate([np.full(int(np.prod(s)), i + 0.5, dtype=dt) for i, s in enumerate(SHAPES_MIX)])
    assert_bits_equal(x.to_numpy(), ref, "setblock! scalar")
    blk = u01(oracle, dt, 5, 5, int(np.prod(SHAPES_MIX[3])))
    Jets.setblock_(x, 3, blk.reshape(SHAPES_MIX[3], order="F"))
    ref[Jets.indices(x, 3).start:Jets.indices(x, 3).stop] = blk
    assert_bits_equal(x.to_numpy(), ref, "setblock! array")
    dev_src = Jets.rand(Jets.JetSpace(dt, *SHAPES_MIX[2]), seed=6, stream=6)
    Jets.setblock_(x, 2, dev_src)                           # device-to-device
    ref[Jets.indices(x, 2).start:Jets.indices(x, 2).stop] = dev_src.to_numpy().ravel(order="F")
    assert_bits_equal(x.to_numpy(), ref, "setblock! device array")


def test_getblock_is_a_reference_not_a_copy(Jets):
    """docs/src/index.md:223 / src/Jets.jl:914."""
    R = _bspace(Jets, np.float64, RAGGED)
    x = Jets.zeros(R)
    b1 = Jets.getblock(x, 1)
    Jets.fill_(b1, 5.0)
    assert np.array_equal(x.to_numpy(), np.concatenate([np.zeros(2), np.full(4, 5.0), np.zeros(6)]))
    assert b1.shape == (2, 2)


@pytest.mark.parametrize("dt", DTYPES)
def test_broadcast_lincomb_bit_exact(Jets, oracle, dt):
    """x = a*u .+ b*v .+ c*w  (test/runtests.jl:555-568), evaluated left to right in eltype T."""
    R = _bspace(Jets, dt, SHAPES_MIX)
    u, v, w = (Jets.rand(R, seed=2, stream=k) for k in (1, 2, 3))
    a, b, c = 0.37, 0.81, 0.59
    if np.dtype(dt).kind == "c":
        a, b, c = 0.37 + 0.2j, 0.81 - 0.4j, -0.59 + 1.5j
    x = (a * u + b * v + c * w).materialize()
    assert isinstance(x, Jets.BlockArray) and x.dtype == np.dtype(dt)      # :562
    bu, bv, bw = (dev_blocks_to_numpy(t) for t in (u, v, w))
    ref = oracle.barr_lincomb([np.empty_like(t) for t in bu], [a, b, c], [bu, bv, bw])
    assert_bits_equal(x.to_numpy(), np.concatenate(ref), "a*u .+ b*v .+ c*w")
    y = Jets.zeros(R)
    y.assign(x)                                                             # y .= x  (:570-571)
    assert_bits_equal(y.to_numpy(), x.to_numpy(), "y .= x")
    # in-place with aliasing: u .= u .- v
    u.assign(u - v)
    ref2 = oracle.barr_lincomb([np.empty_like(t) for t in bu], [1.0, -1.0], [bu, bv])
    assert_bits_equal(u.to_numpy(), np.concatenate(ref2), "u .= u .- v")


@pytest.mark.parametrize("dt", DTYPES)
def test_hadamard_bit_exact(Jets, oracle, dt):
    R = _bspace(Jets, dt, SHAPES_MIX)
    x, y = Jets.rand(R, seed=2, stream=7), Jets.rand(R, seed=2, stream=8)
    hx, hy = x.to_numpy(), y.to_numpy()
    blk = oracle.Block("diag", hx.size, coeff=hx)            # d .= diagonal .* m / m .= conj.(diagonal) .* d
    z = Jets.hadamard_(Jets.zeros(R), x, y)
    assert_bits_equal(z.to_numpy(), oracle.child_mul(blk, np.empty_like(hx), hy), "x .* y")
    zc = Jets.hadamard_(Jets.zeros(R), x, y, conj_x=True)
    assert_bits_equal(zc.to_numpy(), oracle.child_mul_adj(blk, np.empty_like(hx), hy), "conj.(x) .* y")


def test_similar_variants(Jets):
    """test/runtests.jl:595-599."""
    R = _bspace(Jets, np.float64, RAGGED)
    y = Jets.rand(R)
    assert isinstance(Jets.similar(y, np.float32), Jets.BlockArray)
    assert isinstance(Jets.similar(y, np.float32, y.length()), Jets.BlockArray)
    assert isinstance(Jets.similar(y, np.float32, (y.length(),)), Jets.BlockArray)
    assert isinstance(Jets.similar(y, np.float32, 5), Jets.DeviceArray)
    assert R.similar((0,)) == Jets.JetSpace(np.float64, 0)                 # :619


def test_reshape_shares_memory(Jets):
    """test/runtests.jl:602-610: reshape(x, R) aliases x; linear indexing reads through."""
    x = Jets.rand(Jets.JetSpace(np.float64, 5, 10), seed=4, stream=4)
    R = Jets.JetBSpace([Jets.JetSpace(np.float64, 5) for _ in range(10)])
    _x = Jets.reshape(x, R)
    host = x.to_numpy().ravel(order="F")
    for i in range(len(_x)):
        assert _x[i] == host[i]
        _x[i] = float(i)
    assert np.array_equal(x.to_numpy().ravel(order="F"), np.arange(50.0))
    assert Jets.reshape(_x, R) is _x                                        # src/Jets.jl:1115-1118
    with pytest.raises(ValueError):
        Jets.reshape(_x, Jets.JetBSpace([Jets.JetSpace(np.float64, 7)]))


def test_empty_and_degenerate_blocks(Jets):
    R = Jets.JetBSpace([Jets.JetSpace(np.float32, 0), Jets.JetSpace(np.float32, 3), Jets.JetSpace(np.float32, 0)])
    x = Jets.ones(R)
    assert x.length() == 3 and Jets.indices(x, 0) == range(0, 0) and Jets.indices(x, 2) == range(3, 3)
    assert float(Jets.norm(x)) == pytest.approx(math.sqrt(3.0))
    assert float(Jets.dot(x, x)) == 3.0
    Jets.setblock_(x, 0, 9.0)                                               # no-op on an empty block
    assert np.array_equal(x.to_numpy(), np.ones(3, dtype=np.float32))


def test_errors_are_loud(Jets):
    R = _bspace(Jets, np.float32, RAGGED)
    x = Jets.zeros(R)
    with pytest.raises(Jets.JetsHipError):
        Jets.dot(x, Jets.zeros(Jets.JetSpace(np.float32, 5)))               # length mismatch
    with pytest.raises(Jets.JetsHipError):
        Jets.dot(x, Jets.zeros(_bspace(Jets, np.float64, RAGGED)))          # dtype mismatch
    with pytest.raises(IndexError):
        x[12]
    with pytest.raises(IndexError):
        Jets.getblock(x, 3)


@pytest.mark.parametrize("dt", DTYPES)
def test_randn_and_abs(Jets, dt):
    """randn(R) (test/runtests.jl:535-540) and abs.(x) (545-547)."""
    R = Jets.JetBSpace([Jets.JetSpace(dt, 200_000), Jets.JetSpace(dt, 50, 7), Jets.JetSpace(dt, 3)])
    x = Jets.randn(R, seed=5, stream=2)
    v = x.to_numpy().astype(np.complex128)
    assert abs(v.mean()) < 0.01 and abs(np.mean(np.abs(v) ** 2) - 1.0) < 0.01            # zero mean, unit variance
    assert np.array_equal(x.to_numpy(), Jets.randn(R, seed=5, stream=2).to_numpy())       # pure function of (seed, stream, index)
    y = Jets.randn(Jets.JetSpace(dt, 1000), seed=5, stream=2, index_base=100)
    assert np.array_equal(y.to_numpy(), x.to_numpy()[100:1100])
    a = Jets.abs_(x)
    single = np.dtype(dt) in (np.dtype(np.float32), np.dtype(np.complex64))
    assert isinstance(a, Jets.BlockArray) and a.dtype == np.dtype(np.float32 if single else np.float64)   # eltype(abs.(x)) is real
    assert np.allclose(a.to_numpy(), np.abs(x.to_numpy()), rtol=1e-6 if single else 1e-14)
    if np.dtype(dt).kind != "c":
        mn, mx = Jets.extrema(x)
        assert (mn, mx) == (x.to_numpy().min(), x.to_numpy().max()) and mn < 0 < mx


# ---------------------------------------------------------------------------------- pinned host memory
def test_pinned_host_buffers_round_trip(Jets, oracle):
    """jh_host_alloc / jh_host_register: page-locked host arrays for getblock!/setblock!/convert at the PCIe rate.
    Same bytes as the pageable path."""
    dt, n = np.float32, 1 << 20
    R = Jets.JetBSpace([Jets.JetSpace(dt, n // 4)] * 4)
    x = Jets.rand(R, seed=5, stream=0)
    ref = u01(oracle, dt, 5, 0, n)
    pin = Jets.pinned_empty(n, dt)
    assert pin.shape == (n,) and pin.dtype == np.dtype(dt)
    got = x.to_numpy(out=pin)
    assert got.ctypes.data == pin.ctypes.data
    assert_bits_equal(np.asarray(pin), ref, "download into a pinned array")
    y = Jets.zeros(R)
    pin[...] = ref[::-1]
    Jets.upload_from(y, pin)
    assert_bits_equal(y.to_numpy(), ref[::-1].copy(), "upload from a pinned array")
    blockview = pin[: n // 4]                                                         # a view keeps the buffer alive
    del pin, got
    Jets.getblock_(y, 2, blockview)
    assert_bits_equal(np.asarray(blockview), ref[::-1][n // 2: 3 * n // 4].copy(), "getblock! into a pinned view")
    host = np.empty(n, dt)                                                            # an ordinary array, pinned in place
    Jets.host_register(host)
    try:
        Jets.download_into(x, host)
        assert_bits_equal(host, ref, "download into a registered array")
    finally:
        Jets.host_unregister(host)
    with pytest.raises(Jets.JetsHipError):
        Jets.host_unregister(host)                                                    # not registered any more
    cube = Jets.pinned_empty((8, 4, 2), np.float64)
    assert cube.flags.f_contiguous and cube.shape == (8, 4, 2)


# ---- IEEE special values in the reductions (round 3) ---------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_nan_in_norm_inf_and_extrema_follows_the_references_folds(Jets, oracle, dt):
    """`max` / `min` answer NaN when they meet one (src/Jets.jl:835-838); extrema folds the blocks' extrema with `<` / `>` (:870-878):
    a NaN in the first block is the answer, a later block holding one drops out whole.  Big blocks too (several workgroups)."""
    R = Jets.JetBSpace([Jets.JetSpace(dt, 3), Jets.JetSpace(dt, 70001), Jets.JetSpace(dt, 2, 5)])
    x = Jets.rand(R, seed=5, stream=2)
    x.assign(2.0 * x - 1.0 * Jets.ones(R))
    clean = dev_blocks_to_numpy(x)
    assert (float(Jets.extrema(x)[0]), float(Jets.extrema(x)[1])) == oracle.barr_extrema(clean)
    Jets.setblock_(x, 1, np.where(np.arange(70001) == 65000, np.nan, clean[1]).astype(dt))   # a NaN deep inside the second block (0-based index)
    blocks = dev_blocks_to_numpy(x)
    assert math.isnan(float(Jets.norm(x, math.inf))) and math.isnan(oracle.barr_norm(blocks, math.inf))
    assert math.isnan(float(Jets.norm(x, -math.inf))) and math.isnan(oracle.barr_norm(blocks, -math.inf))
    assert math.isnan(float(Jets.norm(x))) and math.isnan(float(Jets.norm(x, 1)))
    mn, mx = Jets.extrema(x)
    assert (float(mn), float(mx)) == oracle.barr_extrema(blocks)
    flat02 = np.concatenate([clean[0].ravel(), clean[2].ravel()])
    assert (float(mn), float(mx)) == (flat02.min(), flat02.max())                # the block with the NaN is not seen at all
    Jets.setblock_(x, 0, np.array([0.25, np.nan, -0.5], dtype=dt))                # ... and a NaN in the FIRST block is the answer
    mn, mx = Jets.extrema(x)
    omn, omx = oracle.barr_extrema(dev_blocks_to_numpy(x))
    assert math.isnan(float(mn)) and math.isnan(float(mx)) and math.isnan(omn) and math.isnan(omx)
    y = Jets.from_numpy(np.array([1, 2, np.inf, -3], dtype=dt))                  # infinities are ordinary values
    assert float(Jets.norm(y, math.inf)) == math.inf and float(Jets.norm(y)) == math.inf
    assert tuple(float(v) for v in Jets.extrema(y)) == (-3.0, math.inf)


@pytest.mark.parametrize("dt", DTYPES)
def test_norm_of_huge_and_tiny_values_neither_overflows_nor_vanishes(Jets, oracle, dt):
    """The stdlib's block norms rescale (BLAS nrm2, generic_normp): norm([1e200, 1e200]) is 1.41e200 and norm([1e-200]) is 1e-200.  The
    device sums squares in fp64: Float32 data cannot leave that range, Float64 data can -- then the pass is repeated on x / 2^k."""
    rt = np.float32 if dt in (np.float32, np.complex64) else np.float64
    big, tiny = (rt(1e30), rt(1e-30)) if rt == np.float32 else (rt(1e200), rt(1e-200))
    for v in (big, tiny):
        h = (np.arange(1, 5001) % 7 + 1).astype(rt) * v
        h = h.astype(dt) if np.dtype(dt).kind != "c" else (h + 1j * h[::-1]).astype(dt)
        x = Jets.from_numpy(h)
        truth = float(v) * float(np.linalg.norm((h / v).astype(np.complex128)))
        assert float(Jets.norm(x)) == pytest.approx(truth, rel=1e-6 if rt == np.float32 else 1e-14)
        # the reference itself leaves the range one level up: it squares the BLOCK norms in real(T) (`norm(_x,p)^_p`, src/Jets.jl:843-846),
        # so it answers Inf / 0 here -- a deviation kept on purpose: the device answer is the true norm
        assert oracle.barr_norm([h], 2.0) == (math.inf if v == big else 0.0)
        t3 = float(v) * float(np.sum(np.abs((h / v).astype(np.complex128)) ** 3) ** (1 / 3))
        assert float(Jets.norm(x, 3)) == pytest.approx(t3, rel=1e-6 if rt == np.float32 else 1e-13)
    assert float(Jets.norm(Jets.zeros(Jets.JetSpace(dt, 100)))) == 0.0


# ---------------------------------------------------------------------------------- round 6: per-block reductions in one pass
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("lens", [[1000] * 7, [4099, 5, 1 << 16, 33, 1027], [3], [257] * 300, [1 << 20, 1 << 20 | 1, 7]])
def test_norm_blocks_and_dot_blocks_equal_the_block_by_block_reductions(Jets, oracle, dt, lens):
    """jh_norm_blocks / jh_dot_blocks (round 6): norm(x_i, p) and dot(x_i, y_i) of EVERY block in one pass over the slab -- the quantities src/Jets.jl:836-846 and
    850-856 form block by block.  Against the fp64 truth per block (stated tolerance of the reductions: 1e-5 for 32-bit, 1e-12 for 64-bit elements -- fp64
    lanes, fixed order), against the oracle's own block norms where its Float32 accumulation is accurate enough to compare (blocks of <= 4100 elements; it
    sums a Float32 block in Float32 like the reference's generic norm: 2e-4 off at 2^20 elements), and against the library's whole-vector entry points on views."""
    J = Jets
    R = J.JetBSpace([J.JetSpace(dt, n) for n in lens])
    x, y = J.rand(R, seed=31, stream=1), J.rand(R, seed=32, stream=2)
    hx, hy = x.to_numpy(), y.to_numpy()
    offs = np.concatenate([[0], np.cumsum(lens)])
    tol = 1e-5 if np.dtype(dt) in (np.dtype(np.float32), np.dtype(np.complex64)) else 1e-12
    wide = np.complex128 if np.dtype(dt).kind == "c" else np.float64
    for p in (2, 1, 0, np.inf, -np.inf, 3, 2.5):
        got = J.norm_blocks(x, p)
        assert got.shape == (len(lens),)
        for i in range(len(lens)):
            ab = np.abs(hx[offs[i]:offs[i + 1]].astype(wide))
            want = float(ab.max() if p == np.inf else ab.min() if p == -np.inf else np.count_nonzero(ab) if p == 0 else ab.sum() if p == 1 else (ab ** p).sum() ** (1.0 / p))
            assert abs(float(got[i]) - want) <= tol * max(abs(want), 1e-30), f"p={p} block {i}: {got[i]} vs {want}"
            if lens[i] <= 4100 and p in (2, 1, 0, np.inf, -np.inf):               # the oracle's block norm (jo.barr_norm: the reference's formula on one block)
                ora = oracle.barr_norm([hx[offs[i]:offs[i + 1]].copy()], p)
                assert abs(float(got[i]) - ora) <= 1e-4 * max(abs(ora), 1e-30), f"p={p} block {i} vs the oracle"
            if len(lens) <= 8:                                                   # the whole-vector entry point on the block's view (another grouping of the same fp64 sums)
                ref = float(J.norm(J.getblock(x, i), p))
                assert abs(float(got[i]) - ref) <= tol * max(abs(ref), 1e-30), f"p={p} block {i} vs jh_norm on the view"
    gd = J.dot_blocks(x, y)
    for i in range(len(lens)):
        want = np.vdot(hx[offs[i]:offs[i + 1]].astype(np.complex128), hy[offs[i]:offs[i + 1]].astype(np.complex128))
        assert abs(complex(gd[i]) - complex(want)) <= tol * max(abs(complex(want)), 1e-30), f"dot block {i}"
    # the blocks' norms combine to the vector's (836-846): p = 2
    assert abs(float(np.sqrt(np.sum(J.norm_blocks(x, 2).astype(np.float64) ** 2))) - float(J.norm(x, 2))) <= 1e-5 * float(J.norm(x, 2))


def test_norm_blocks_rescales_a_block_whose_squares_leave_the_double_range(Jets):
    J = Jets
    R = J.JetBSpace([J.JetSpace(np.float64, 100)] * 3)
    h = np.ones(300)
    h[100:200] = 1e200
    h[200:] = 1e-200
    x = J.from_numpy(h, R)
    got = J.norm_blocks(x, 2)
    assert np.allclose(got, [10.0, 1e201, 1e-199], rtol=1e-12)


@pytest.mark.parametrize("dt", DTYPES)
def test_per_block_reductions_of_many_short_blocks(Jets, oracle, dt):
    """Thousands of short ragged blocks (traces rather than volumes; some empty): a wave per block in one launch (k_reduce_blocks_wave) -- against the fp64 truth
    per block and against the workgroup-per-block kernels (knob red_blocks_wave = 0) within the reductions' tolerance."""
    J = Jets
    rng = np.random.default_rng(5)
    lens = [int(v) for v in rng.integers(0, 700, size=1500)]
    lens[7] = 0
    lens[8] = 4096 // (np.dtype(dt).itemsize // 4)                                 # (the longest block the wave kernel takes: 16 KiB)
    R = J.JetBSpace([J.JetSpace(dt, n) for n in lens])
    x, y = J.rand(R, seed=31, stream=1), J.rand(R, seed=32, stream=2)
    hx, hy = x.to_numpy(), y.to_numpy()
    offs = np.concatenate([[0], np.cumsum(lens)])
    tol = 1e-5 if np.dtype(dt) in (np.dtype(np.float32), np.dtype(np.complex64)) else 1e-12
    wide = np.complex128 if np.dtype(dt).kind == "c" else np.float64
    ax = np.abs(hx.astype(wide))
    for p in (2, 1, 0, np.inf, -np.inf, 3):
        got = J.norm_blocks(x, p).astype(np.float64)
        J.tune(red_blocks_wave=0)
        try:
            other = J.norm_blocks(x, p).astype(np.float64)
        finally:
            J.tune(red_blocks_wave=1)
        want = np.zeros(len(lens))
        for i in range(len(lens)):
            ab = ax[offs[i]:offs[i + 1]]
            if ab.size:
                want[i] = ab.max() if p == np.inf else ab.min() if p == -np.inf else np.count_nonzero(ab) if p == 0 else ab.sum() if p == 1 else (ab ** p).sum() ** (1.0 / p)
        assert np.all(np.abs(got - want) <= tol * np.maximum(np.abs(want), 1e-30)), f"p={p}: wave per block vs the truth"
        assert np.all(np.abs(got - other) <= tol * np.maximum(np.abs(want), 1e-30)), f"p={p}: wave per block vs workgroup per block"
    gd = J.dot_blocks(x, y)
    want = np.array([np.vdot(hx[offs[i]:offs[i + 1]].astype(np.complex128), hy[offs[i]:offs[i + 1]].astype(np.complex128)) for i in range(len(lens))])
    assert np.all(np.abs(gd.astype(np.complex128) - want) <= tol * np.maximum(np.abs(want), 1e-30))
