"""GPU: the HIP path (through the C ABI) against known answers that do NOT come from the oracle.

tests/golden/known_answers.npz is derived from the reference's source lines with exact rational arithmetic + an IEEE
round-to-nearest-even written for the purpose (tests/golden/make_known_answers.py); tests/test_known_answers.py holds the CPU
oracle to the same arrays.  Bit-exact for every case, under every instantiated shape of the ordered tall kernels."""
import numpy as np
import pytest

from . import known_answers as ka

pytestmark = pytest.mark.gpu

ADJ_SHAPES = [dict(adj_wg=0, adj_unroll=0, adj_depth=0)] + [dict(adj_wg=w, adj_unroll=u, adj_depth=d) for w in (256, 1024) for (u, d) in ((1, 1), (1, 8), (2, 4), (4, 2))]


@pytest.fixture(scope="module")
def z():
    return ka.load()


def _dom(J, A, c, blocks):
    flat = np.concatenate(blocks)
    return J.from_numpy(flat, J.domain(A)) if c.ncol > 1 else J.from_numpy(flat)


def _rng(J, A, blocks):
    return J.from_numpy(np.concatenate(blocks), J.range(A))


def _blocks_of(x, lens):
    flat = x.to_numpy().ravel(order="F") if not hasattr(x, "arrays") else x.to_numpy()
    out, o = [], 0
    for n in lens:
        out.append(flat[o:o + n])
        o += n
    return out


@pytest.mark.parametrize("name", ka.LINEAR_CASES)
def test_hip_path_reproduces_the_independent_known_answers(Jets, z, name):
    J = Jets
    c = ka.Case(z, name)
    A = ka.device_ops(J, c)
    m = _dom(J, A, c, c.blocks("m", c.ncol))
    tall = c.ncol == 1 and c.nrow > 1
    shapes = ADJ_SHAPES if tall else ADJ_SHAPES[:1]
    try:
        for knobs in shapes:
            J.tune(**knobs)
            d = _rng(J, A, c.blocks("d_found", c.nrow))                       # dirty range vector: overwritten (1026) or accumulated into (1024)
            J.mul_(d, A, m)
            for i, got in enumerate(_blocks_of(d, c.row_len)):
                assert ka.bits(got) == ka.bits(c.get(f"fwd_{i}")), f"{name} {knobs}: forward block {i}"
            mt = _dom(J, A, c, c.blocks("m_found", c.ncol))                    # dirty domain vector: zeroed (1042) / written (1051) / untouched (1047)
            J.mul_(mt, A.H, _rng(J, A, c.adjoint_input()))
            for j, got in enumerate(_blocks_of(mt, c.col_len)):
                assert ka.bits(got) == ka.bits(c.get(f"adj_{j}")), f"{name} {knobs}: adjoint block {j}"
            if c.has("normal_0"):                                             # JetComposite (A', A): the fused kernel
                y = (A.H @ A) * m
                assert ka.bits(y.to_numpy().ravel(order="F")) == ka.bits(c.get("normal_0")), f"{name} {knobs}: fused A'A"
    finally:
        J.tune(adj_wg=0, adj_unroll=0, adj_depth=0)


@pytest.mark.parametrize("name", ["rand_tall_f32", "rand_tall_c64", "order_tall_f32", "mixed_tall_f32", "mixed_tall_c64"])
def test_one_pass_step_reproduces_the_known_adjoint(Jets, z, name):
    """jh_blockop_bidiag_step with alpha = 1, beta = 0 is forward-then-adjoint in one pass: u must be the known forward, w the
    known adjoint of it (order_tall: of the stored d_in, so only its forward half is compared there).  mixed_tall: the step's
    forward half is A v into a zeros() temporary, so a zero row of u is 0 (not "as found") and w is the known A'A v."""
    import ctypes as C
    from jets_jl_amd._ffi import lib, check
    from jets_jl_amd import jetblock as _blk

    J = Jets
    c = ka.Case(z, name)
    A = ka.device_ops(J, c)
    nat = _blk._tall_native(A)
    v = _dom(J, A, c, c.blocks("m", 1))
    u = _rng(J, A, c.blocks("d_found", c.nrow))
    w = J.zeros(J.domain(A))
    out = C.c_double(0)
    check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, 1.0, 0.0, C.byref(out)))
    mixed = name.startswith("mixed")
    for i, got in enumerate(_blocks_of(u, c.row_len)):
        if mixed and int(c.kind[i, 0]) == 0:
            assert not np.any(got.view(np.uint8)), f"{name}: zero row {i} of u must be +0"
        else:
            assert ka.bits(got) == ka.bits(c.get(f"fwd_{i}")), f"{name}: u block {i}"
    if mixed:
        assert ka.bits(w.to_numpy().ravel(order="F")) == ka.bits(c.get("normal_0")), f"{name}: w"
    elif not c.has("d_in_0"):
        assert ka.bits(w.to_numpy().ravel(order="F")) == ka.bits(c.get("adj_0")), f"{name}: w"
        want = sum(float(np.vdot(c.get(f"fwd_{i}").astype(np.complex128), c.get(f"fwd_{i}").astype(np.complex128)).real) for i in range(c.nrow))
        assert abs(out.value - want) <= 1e-12 * want


@pytest.mark.parametrize("name", ka.SUM_CASES)
def test_jetsum_sign_rules_and_rounding_sequence(Jets, z, name):
    """A1 - (A2 - A3): the signs flatten to (+, -, +) (src/Jets.jl:667-676) and the fused sum kernels keep the unfused chain's
    rounding sequence, incl. `0 + (-0) = +0` after `d .= 0` (640).  Round 3: sums of 6 and 8 terms (ONE launch of the eight-term
    kernels) and of 11 (eight + three: the second launch continues the left-to-right sum), spelled as nested differences."""
    J = Jets
    c = ka.Case(z, name)
    nrow = c.nrow
    nt = len(ka.sum_signs(c))
    ops = [J.blockop([[J.JopDiagonal(J.from_numpy(c.get(f"coeff_{t}_{i}")))] for i in range(nrow)]) for t in range(nt)]
    S = ka.SUM_EXPRESSIONS[name](ops)
    m = J.from_numpy(c.get("m_0"))
    d = S * m
    n = c.get("m_0").size
    for i, got in enumerate(_blocks_of(d, [n] * nrow)):
        assert ka.bits(got) == ka.bits(c.get(f"fwd_{i}")), f"{name}: sum forward block {i}"
    din = _rng(J, ops[0], c.blocks("d_in", nrow))
    mt = S.H * din
    assert ka.bits(mt.to_numpy().ravel(order="F")) == ka.bits(c.get("adj_0")), f"{name}: sum adjoint"


@pytest.mark.parametrize("name", ka.CHAIN_CASES)
def test_fused_chains_reproduce_the_independent_known_answers(Jets, z, name):
    """Round 6: composites of depth 3 to 7 around a tall operator (A' o W o A, (W o A)' o (W o A), a * (A' o A), M' o A' o (b W) o A o (a M), (b W') o A and
    its adjoint) and sums whose terms are such chains, through the FUSED chain kernels (jh_chain_*; the test asserts the fused path ran) -- bit for bit the
    arrays derived in softfloat from src/Jets.jl:530-540, 639-655, 1159-1160."""
    from jets_jl_amd import chains

    J = Jets
    c = ka.Case(z, name)
    A = ka.device_ops(J, c)
    R, D = J.range(A), J.domain(A)
    w = [J.JopDiagonal(J.from_numpy(c.get(f"w_{k}"), R)) for k in range(2)]
    cdom = J.JopDiagonal(J.from_numpy(c.get("c_0"), D))
    m, din = J.from_numpy(c.get("m_0"), D), J.from_numpy(c.get("d_in"), R)
    for key, stages in ka.CHAINS.items():
        C = ka.device_chain(J, c, A, w, cdom, stages)
        x = din if key.startswith("a_") else m
        before = chains.STATS["chain_calls"]
        y = J.mul_(J.rand(J.range(C), seed=5, stream=5), C, x)                  # into a dirty output
        assert chains.STATS["chain_calls"] == before + 1, f"{name}: {key} did not take the fused path"
        got = y.to_numpy() if hasattr(y, "arrays") else y.to_numpy().ravel(order="F")
        assert ka.bits(got) == ka.bits(c.get(key)), f"{name}: {key}"
    for key, (terms, xin) in ka.CHAIN_SUMS.items():
        x = din if xin == "d_in" else m
        S = None
        for sign, stages in terms:
            T = ka.device_chain(J, c, A, w, cdom, stages)
            S = T if S is None else (S + T if sign > 0 else S - T)
        before = chains.STATS["sum_terms_fused"]
        y = J.mul_(J.rand(J.range(S), seed=6, stream=6), S, x)
        assert chains.STATS["sum_terms_fused"] > before, f"{name}: {key} did not fuse any term"
        got = y.to_numpy() if hasattr(y, "arrays") else y.to_numpy().ravel(order="F")
        assert ka.bits(got) == ka.bits(c.get(key)), f"{name}: {key}"
    J.close(A)
