"""CPU: the oracle against known answers that do NOT come from the oracle (tests/golden/known_answers.npz).

The expected arrays were derived from the reference's source lines with exact rational arithmetic and a round-to-nearest-even
written from the IEEE standard (tests/golden/make_known_answers.py) -- exact-integer cases (answer independent of rounding and
order: pins layout and the skip / accumulate / overwrite rules) and order-revealing cases (2^24, 1, 1, ... sums; random values
in full softfloat emulation: pin the summation ORDER and the rounding SEQUENCE of src/Jets.jl:1042-1049).  The oracle must
reproduce every one bit for bit; tests/test_gpu_known_answers.py holds the HIP path to the same arrays."""
import os
import subprocess
import sys

import numpy as np
import pytest

from . import known_answers as ka

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def z():
    return ka.load()


@pytest.mark.parametrize("name", ka.LINEAR_CASES)
def test_oracle_reproduces_the_independent_known_answers(oracle, z, name):
    c = ka.Case(z, name)
    ops = ka.oracle_ops(oracle, c)
    m = c.blocks("m", c.ncol)
    d = oracle.block_df(ops, c.blocks("d_found", c.nrow), m)                    # forward from the dirty range vector
    for i in range(c.nrow):
        assert ka.bits(d[i]) == ka.bits(c.get(f"fwd_{i}")), f"{name}: forward block {i}"
    mt = oracle.block_df_adj(ops, c.blocks("m_found", c.ncol), c.adjoint_input())
    for j in range(c.ncol):
        assert ka.bits(mt[j]) == ka.bits(c.get(f"adj_{j}")), f"{name}: adjoint block {j}"
    if c.has("normal_0"):                                                       # JetComposite (A', A)
        y = oracle.normal_df(ops, [np.zeros_like(m[0])], m)
        assert ka.bits(y[0]) == ka.bits(c.get("normal_0")), f"{name}: A'A"


@pytest.mark.parametrize("name", ka.SUM_CASES)
def test_oracle_chain_reproduces_the_jetsum_known_answers(oracle, z, name):
    """JetSum_df! / df'! (src/Jets.jl:639-655) as the oracle's unfused chain: d .= 0; d = d +- mul!(tmp, A_t, m), signs (+, -, +) for
    the three-term cases, the stored `signs` for the long sums of round 3 (6, 8 and 11 terms)."""
    c = ka.Case(z, name)
    nrow, dt = c.nrow, c.dtype
    n = c.get("m_0").size
    sg = ka.sum_signs(c)
    nt = len(sg)
    terms = [[[oracle.Block("diag", n, coeff=c.get(f"coeff_{t}_{i}").copy())] for i in range(nrow)] for t in range(nt)]
    m = [c.get("m_0").copy()]
    d = [np.zeros(n, dt) for _ in range(nrow)]
    for t in range(nt):
        tmp = oracle.block_df(terms[t], [np.zeros(n, dt) for _ in range(nrow)], m)
        d = oracle.barr_lincomb([np.empty(n, dt) for _ in range(nrow)], [1.0, sg[t]], [d, tmp])
    for i in range(nrow):
        assert ka.bits(d[i]) == ka.bits(c.get(f"fwd_{i}")), f"{name}: sum forward block {i}"
    din = c.blocks("d_in", nrow)
    mt = [np.zeros(n, dt)]
    for t in range(nt):
        tmp = oracle.block_df_adj(terms[t], [np.zeros(n, dt)], din)
        mt = oracle.barr_lincomb([np.empty(n, dt)], [1.0, sg[t]], [mt, tmp])
    assert ka.bits(mt[0]) == ka.bits(c.get("adj_0")), f"{name}: sum adjoint"


@pytest.mark.parametrize("name", ka.CHAIN_CASES)
def test_oracle_stages_reproduce_the_chain_known_answers(oracle, z, name):
    """Round 6: composites of depth 3 to 7 around a tall operator with rows of every kind (JetComposite_df! / df'!, src/Jets.jl:530-540: every stage
    into its own zeros()) and sums whose terms are chains (639-655), derived in softfloat from the reference's lines -- the oracle applying the stages
    one by one must reproduce every array bit for bit (tests/test_gpu_known_answers.py holds the FUSED device path to the same arrays)."""
    c = ka.Case(z, name)
    ops = ka.oracle_ops(oracle, c)
    n, dt = c.col_len[0], c.dtype
    for key, stages in ka.CHAINS.items():
        x = c.get("d_in") if key.startswith("a_") else c.get("m_0")
        assert ka.bits(ka.oracle_chain(oracle, c, ops, stages, x)) == ka.bits(c.get(key)), f"{name}: {key}"
    for key, (terms, xin) in ka.CHAIN_SUMS.items():
        x = c.get(xin)
        acc = None
        for sign, stages in terms:
            t = ka.oracle_chain(oracle, c, ops, stages, x)
            if acc is None:
                acc = np.zeros(t.size, dt)                                       # d .= 0   (640 / 649)
            acc = oracle.barr_lincomb([np.empty(t.size, dt)], [1.0, sign], [[acc], [t]])[0]
        assert ka.bits(acc) == ka.bits(c.get(key)), f"{name}: {key}"


@pytest.mark.parametrize("name", ka.SUM_CASES)
def test_the_sum_expressions_flatten_to_the_stored_signs(z, name):
    """The nested differences the GPU test builds (ka.SUM_EXPRESSIONS) flatten, by the reference's rule (667-676: a minus flips the
    signs of everything inside), to the sign sequence the fixture was derived with."""
    class Sym:
        def __init__(self, terms):
            self.t = terms

        def __add__(self, o):
            return Sym(self.t + o.t)

        def __sub__(self, o):
            return Sym(self.t + [(i, -sg) for i, sg in o.t])

    c = ka.Case(z, name)
    want = ka.sum_signs(c)
    e = ka.SUM_EXPRESSIONS[name]([Sym([(i, 1.0)]) for i in range(len(want))])
    assert [i for i, _ in e.t] == list(range(len(want))) and [sg for _, sg in e.t] == want


def test_the_order_revealing_case_really_reveals_the_order(z):
    """Guard on the fixture itself: in `order_tall_f32` the reference's sequential Float32 sum differs from the exactly rounded
    sum (and hence from pairwise / split / fp64-accumulated sums) in most columns."""
    c = ka.Case(z, "order_tall_f32")
    a = np.stack([c.get(f"coeff_{i}_0") for i in range(c.nrow)]).astype(np.float64)
    exact = a.sum(axis=0)                                                        # every term is an integer <= 2^24: exact in float64
    got = c.get("adj_0").astype(np.float64)
    assert np.count_nonzero(got != exact) >= 24
    assert got[0] == 2.0 ** 24 and exact[0] == 2.0 ** 24 + 10                    # (2^24, 1, 1, ...): ties to even, stays put


def test_the_committed_fixture_is_what_the_generator_writes(tmp_path):
    """Regenerate into a scratch copy and compare array by array (the generator is pure Python, ~2 s)."""
    gen = os.path.join(ROOT, "tests", "golden", "make_known_answers.py")
    scratch = tmp_path / "golden"
    scratch.mkdir()
    code = open(gen).read().replace("HERE = os.path.dirname(os.path.abspath(__file__))", f"HERE = {str(scratch)!r}")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    new, old = np.load(scratch / "known_answers.npz"), ka.load()
    assert sorted(new.files) == sorted(old.files)
    for k in old.files:
        assert ka.bits(new[k]) == ka.bits(old[k]) and new[k].dtype == old[k].dtype, k
