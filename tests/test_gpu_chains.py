"""GPU parity: JetComposite chains of any depth and JetSum terms that are chains, fused around a tall block operator (round 6;
jets.jl_amd/chains.py, jh_tall_chain.hip: jh_chain_*).

The reference applies a composite stage by stage, right to left, each stage into a fresh zeros(range(op_i)) (src/Jets.jl:524-540), a sum term by
term through one temporary (630-655), `a * A` as one more stage (1159-1164).  The fused kernels keep every stage's rounding, so the bar is
BIT-EXACT three ways: against the same chain applied stage by stage on the device (chains.ENABLED = False: rounds 1-5's path), against the CPU
oracle applying the stages one by one, and (tests/test_gpu_known_answers.py) against the softfloat known answers.  All four element types, block
lengths on and off the 16-byte grid, rows of several kinds (zero blocks, identities, scalars, adjointed diagonals), Float64 scalars on 32-bit
elements, weights in one slab and as a block-diagonal block operator."""
import numpy as np
import pytest

from .helpers import DTYPES, assert_bits_equal, u01
from .test_gpu_blockop import _mixed_ops

pytestmark = pytest.mark.gpu


def _kinds(nrow, name):
    names = ["diag", "diag_adj", "identity", "scale", "zero"]
    if name == "diag":
        return [["diag"] for _ in range(nrow)]
    return [[names[(3 * i + i // 5) % 5]] for i in range(nrow)]


class Rig:
    """One tall operator A (nrow x 1, blocks of n elements) with two weight vectors on its range and two diagonals on its domain, on the device and
    in the oracle.  A chain is a list of tokens in APPLICATION order: "A", "At", ("W", k, conj), ("M", k, conj), ("s", a), ("Wb", k, conj) (the k-th
    weights as a block-diagonal block operator), ("I",) identity on the domain, ("opaque",) a user-written closure on the domain (d .= 2 .* m)."""

    def __init__(self, J, oracle, dt, nrow, n, name="diag", seed=31, with_wb=True):
        self.J, self.o, self.dt, self.nrow, self.n = J, oracle, dt, nrow, n
        self.A, self.ora = _mixed_ops(J, oracle, dt, _kinds(nrow, name), [n] * nrow, [n], seed=seed)
        R, D = J.range(self.A), J.domain(self.A)
        self.w = [J.rand(R, seed=seed + 1 + k, stream=0) for k in range(2)]
        self.hw = [[b.copy() for b in np.split(w.to_numpy(), nrow)] for w in self.w]
        self.c = [J.rand(D, seed=seed + 5 + k, stream=0) for k in range(2)]
        self.hc = [c.to_numpy().ravel(order="F").copy() for c in self.c]
        self.W = [J.JopDiagonal(w) for w in self.w]
        self.M = [J.JopDiagonal(c) for c in self.c]
        self.Wb = []
        for k in range(2 if with_wb else 0):
            spc = J.JetSpace(dt, n)
            rows = []
            for i in range(nrow):
                row = [J.JopZeroBlock(spc, spc) for _ in range(nrow)]
                if i % 4 == 3:
                    row[i] = J.JopIdentity(spc)
                elif i % 7 == 5:
                    pass                                                       # a zero block on the diagonal
                else:
                    d = J.JopDiagonal(self.w[k].arrays[i])
                    row[i] = d.H if i % 3 == 1 else d
                rows.append(row)
            self.Wb.append(J.blockop(rows))

        def twice(d, m, **kw):
            return J.lincomb_(d, [2.0], [m])

        self.opaque = J.JopLn(dom=D, rng=D, df=twice, df_adj=twice)

    # ---- device
    def op(self, tok):
        J = self.J
        if tok == "A":
            return self.A
        if tok == "At":
            return self.A.H
        kind = tok[0]
        if kind == "W":
            return self.W[tok[1]].H if tok[2] else self.W[tok[1]]
        if kind == "Wb":
            return self.Wb[tok[1]].H if tok[2] else self.Wb[tok[1]]
        if kind == "M":
            return self.M[tok[1]].H if tok[2] else self.M[tok[1]]
        if kind == "I":
            return J.JopIdentity(J.domain(self.A))
        if kind == "opaque":
            return self.opaque
        if kind == "s":
            side = tok[2]
            spc = J.range(self.A) if side == "r" else J.domain(self.A)
            return J.JopLn(dom=spc, rng=spc, df=J.constdiag_df, df_adj=J.constdiag_df_adj, s={"a": tok[1]})
        raise ValueError(tok)

    def compose(self, toks):
        ops = [self.op(t) for t in toks]
        out = ops[0]
        for o in ops[1:]:
            out = self.J.compose(o, out)                                        # applied later = further left
        return out

    # ---- oracle, stage by stage (every stage into zeros, src/Jets.jl:525/531/537)
    def ora_apply(self, toks, x):
        o, n, nrow, dt = self.o, self.n, self.nrow, self.dt
        cur = [b.copy() for b in x]
        for tok in toks:
            if tok == "A":
                cur = o.block_df(self.ora, [np.zeros(n, dt) for _ in range(nrow)], cur)
            elif tok == "At":
                cur = o.block_df_adj(self.ora, [np.zeros(n, dt)], cur)
            elif tok[0] in ("W", "M"):
                coef = self.hw[tok[1]] if tok[0] == "W" else [self.hc[tok[1]]]
                cur = [o.child_mul(o.Block("diag", n, coeff=cf, adjoint=bool(tok[2])), np.zeros(n, dt), b) for cf, b in zip(coef, cur)]
            elif tok[0] == "Wb":
                nxt = []
                for i, b in enumerate(cur):                                    # a block operator of several columns ACCUMULATES its rows into zeros
                    if i % 4 == 3:                                             # (src/Jets.jl:1024 `_d .+= mul!(dtmp, ...)`, 1042 / 1049): 0 + product
                        nxt.append(np.zeros(n, dt) + b)
                    elif i % 7 == 5:
                        nxt.append(np.zeros(n, dt))
                    else:
                        nxt.append(np.zeros(n, dt) + o.child_mul(o.Block("diag", n, coeff=self.hw[tok[1]][i], adjoint=(i % 3 == 1) != bool(tok[2])), np.zeros(n, dt), b))
                cur = nxt
            elif tok[0] == "I":
                cur = [b.copy() for b in cur]
            elif tok[0] == "opaque":
                cur = o.barr_lincomb([np.empty(n, dt) for _ in cur], [2.0], [cur])
            elif tok[0] == "s":
                cur = o.barr_lincomb([np.empty(n, dt) for _ in cur], [tok[1]], [cur])
            else:
                raise ValueError(tok)
        return cur

    def close(self):
        self.J.close(self.A)


def _run_both(J, C, x, out_space, chains):
    """The composite C applied fused and stage by stage; returns (fused, unfused, number of fused runs applied)."""
    before = chains.STATS["chain_calls"]
    y1 = J.mul_(J.rand(out_space, seed=77, stream=1), C, x)                    # into a DIRTY output
    ran = chains.STATS["chain_calls"] - before
    chains.ENABLED[0] = False
    try:
        y0 = J.mul_(J.rand(out_space, seed=78, stream=2), C, x)
    finally:
        chains.ENABLED[0] = True
    return y1, y0, ran


CHAINS = {
    # name: (tokens in application order, fused runs expected)
    "W o A": (["A", ("W", 0, False)], 1),
    "W' o A": (["A", ("W", 0, True)], 1),
    "A' o W o A": (["A", ("W", 0, False), "At"], 1),
    "(W o A)' o (W o A)": (["A", ("W", 0, False), ("W", 0, True), "At"], 1),
    "A' o W1' o W0 o A": (["A", ("W", 0, False), ("W", 1, True), "At"], 1),
    "M' o A' o W o A o M": ([("M", 0, False), "A", ("W", 0, False), "At", ("M", 0, True)], 1),
    "a * (A' o A)": (["A", "At", ("s", 0.375, "d")], 1),
    "A' o (a W) o A o (b M)": ([("M", 1, False), ("s", -1.25, "d"), "A", ("W", 1, False), ("s", 3.0, "r"), "At"], 1),
    "A' o Wb o A": (["A", ("Wb", 0, False), "At"], 1),
    "Wb' o A": (["A", ("Wb", 1, True)], 1),
    "A o M": ([("M", 0, False), "A"], 1),
    "A o I o M1 o M0": ([("M", 0, False), ("M", 1, True), ("I",), "A"], 1),
    "M o A'  (range -> domain)": (["At", ("M", 0, False)], 1),
    "M o A' o W'": ([("W", 0, True), "At", ("M", 1, False)], 1),
    "A' o A o opaque o A' o W o A": (["A", ("W", 0, False), "At", ("opaque",), "A", "At"], 2),
    "opaque o A' o W o A o opaque": ([("opaque",), "A", ("W", 0, False), "At", ("opaque",)], 1),
    "five range-side stages": (["A", ("s", 2.0, "r"), ("W", 0, False), ("s", 0.5, "r"), ("W", 1, False), ("s", -1.0, "r"), "At"], 2),   # four stages ride with A, the fifth with A'
}


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("name", list(CHAINS))
@pytest.mark.parametrize("nrow,n,kinds", [(5, 4096 + 64, "diag"), (7, 1027, "diag"), (18, 2051, "mixed"), (3, 67, "mixed")])
def test_chains_of_any_depth_have_the_bits_of_the_stage_by_stage_chain(Jets, oracle, dt, name, nrow, n, kinds):
    from jets_jl_amd import chains

    J = Jets
    toks, runs = CHAINS[name]
    rig = Rig(J, oracle, dt, nrow, n, kinds)
    C = rig.compose(toks)
    rng_in = toks[0] == "At" or (toks[0] != "A" and toks[0][0] in ("W", "Wb"))
    xs = J.range(rig.A) if rng_in else J.domain(rig.A)
    hx = [u01(oracle, dt, 91, i, n) for i in range(nrow if rng_in else 1)]
    x = J.from_numpy(np.concatenate(hx), xs)
    y1, y0, ran = _run_both(J, C, x, J.range(C), chains)
    assert ran == runs, f"{name}: {ran} fused runs applied, expected {runs}"
    assert_bits_equal(y1.to_numpy().ravel(order="F"), y0.to_numpy().ravel(order="F"), f"{name}: fused vs stage by stage on the device")
    want = np.concatenate(rig.ora_apply(toks, hx))
    assert_bits_equal(y1.to_numpy().ravel(order="F"), want, f"{name}: fused vs the oracle's stages")
    # the adjoint of the whole composite: the same stages adjointed, in reverse (src/Jets.jl:536-540)
    hz = [u01(oracle, dt, 92, i, n) for i in range(len(want) // n)]
    z = J.from_numpy(np.concatenate(hz), J.range(C))
    a1, a0, ran = _run_both(J, C.H, z, J.domain(C), chains)
    assert ran == runs
    assert_bits_equal(a1.to_numpy().ravel(order="F"), a0.to_numpy().ravel(order="F"), f"({name})': fused vs stage by stage on the device")
    rig.close()


@pytest.mark.parametrize("dt", [np.float32, np.complex64])
@pytest.mark.parametrize("n", [1024, 1027])
def test_float64_scalars_on_32_bit_elements_inside_a_chain(Jets, oracle, dt, n):
    """`3.14 * (A' o W o A)` and `A' o (2.5 W) o A` with numpy float64 scalars (Julia's Float64): the scalar stage is the promoted product rounded once
    (include/jetship.h: JH_SCALAR_WIDE) -- fused, with the bits of the chain whose scalar stage is the typed lincomb."""
    from jets_jl_amd import chains

    J = Jets
    rig = Rig(J, oracle, dt, 6, n, "mixed")
    for toks in (["A", ("W", 0, False), "At", ("s", np.float64(3.14), "d")], ["A", ("s", np.float64(2.5), "r"), ("W", 0, True), "At"],
                 [("s", np.float64(-0.1), "d"), "A", ("W", 1, False), ("s", np.float64(1.0 / 3.0), "r")]):
        C = rig.compose(toks)
        hx = [u01(oracle, dt, 91, 0, n)]
        x = J.from_numpy(hx[0], J.domain(rig.A))
        y1, y0, ran = _run_both(J, C, x, J.range(C), chains)
        assert ran == 1
        assert_bits_equal(y1.to_numpy().ravel(order="F"), y0.to_numpy().ravel(order="F"), f"{toks}: fused vs stage by stage")
        assert_bits_equal(y1.to_numpy().ravel(order="F"), np.concatenate(rig.ora_apply(toks, hx)), f"{toks}: fused vs the oracle")
    rig.close()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nrow,n,kinds", [(5, 4096, "diag"), (9, 1027, "mixed")])
def test_sums_whose_terms_are_chains(Jets, oracle, dt, nrow, n, kinds):
    """JetSum (src/Jets.jl:628-655) with composite terms: A'oWoA + lam*I - B'oB on the domain, W0oA - 0.5*(W1oB) + B on the range and their adjoints --
    every term that is one fusable run adds itself to the output in its own last stage; bit-identical to the reference's loop over one temporary."""
    from jets_jl_amd import chains

    J = Jets
    ra, rb = Rig(J, oracle, dt, nrow, n, kinds, seed=31), Rig(J, oracle, dt, nrow, n, "diag", seed=57)
    A, B = ra.A, rb.A
    dom = J.domain(A)
    lamI = 0.25 * J.JopIdentity(dom)
    N1 = J.compose(J.compose(A.H, ra.W[0]), A)
    N2 = J.compose(B.H, B)
    S_dom = N1 + lamI - N2
    F1 = J.compose(ra.W[0], A)
    F2 = 0.5 * J.compose(ra.W[1], B)
    S_rng = F1 - F2 + B
    hx = u01(oracle, dt, 91, 0, n)
    x = J.from_numpy(hx, dom)
    for S, nfused in ((S_dom, 2), (S_rng, 2)):          # (A'WA and, in one lincomb pass, lam * I; the bare B'B term takes the tuned fused A'A + one accumulate pass)
        before = chains.STATS["sum_terms_fused"]
        y1 = J.mul_(J.rand(J.range(S), seed=77, stream=1), S, x)
        assert chains.STATS["sum_terms_fused"] - before == nfused
        chains.ENABLED[0] = False
        try:
            y0 = J.mul_(J.rand(J.range(S), seed=78, stream=2), S, x)
        finally:
            chains.ENABLED[0] = True
        assert_bits_equal(y1.to_numpy().ravel(order="F"), y0.to_numpy().ravel(order="F"), "sum of chains: fused vs the reference's loop")
    # oracle for the domain sum: d = ((0 + N1 x) + lam x) - N2 x
    t1 = ra.ora_apply(["A", ("W", 0, False), "At"], [hx])
    t2 = oracle.barr_lincomb([np.empty(n, dt)], [0.25], [[hx]])
    t3 = rb.ora_apply(["A", "At"], [hx])
    want = oracle.barr_lincomb([np.empty(n, dt)], [1.0, 1.0], [[np.zeros(n, dt)], t1])
    want = oracle.barr_lincomb([np.empty(n, dt)], [1.0, 1.0], [want, t2])
    want = oracle.barr_lincomb([np.empty(n, dt)], [1.0, -1.0], [want, t3])
    y = J.mul_(J.rand(dom, seed=7, stream=7), S_dom, x)
    assert_bits_equal(y.to_numpy().ravel(order="F"), want[0], "A'WA + lam I - B'B vs the oracle")
    # adjoints of both sums
    hz = np.concatenate([u01(oracle, dt, 92, i, n) for i in range(nrow)])
    for S, z in ((S_dom, x), (S_rng, J.from_numpy(hz, J.range(S_rng)))):
        a1 = J.mul_(J.rand(J.domain(S), seed=71, stream=1), S.H, z)
        chains.ENABLED[0] = False
        try:
            a0 = J.mul_(J.rand(J.domain(S), seed=72, stream=2), S.H, z)
        finally:
            chains.ENABLED[0] = True
        assert_bits_equal(a1.to_numpy().ravel(order="F"), a0.to_numpy().ravel(order="F"), "adjoint of a sum of chains: fused vs the reference's loop")
    ra.close()
    rb.close()


def test_the_first_term_of_a_fused_sum_is_zero_plus_the_term(Jets, oracle):
    """`d .= 0` then `d .= d - tmp` (src/Jets.jl:640, 643): 0 - (+0) = +0 and 0 + (-0) = +0, not the term's own zero."""
    J = Jets
    dt, n, nrow = np.float32, 1024, 3
    rig = Rig(J, oracle, dt, nrow, n, "diag")
    N = J.compose(J.compose(rig.A.H, rig.W[0]), rig.A)
    S = N - N                                                                 # two fused terms: 0 + t, then - t
    x = J.zeros(J.domain(rig.A))                                              # every product is +0
    y = J.mul_(J.rand(J.domain(rig.A), seed=3, stream=3), S, x)
    got = y.to_numpy().ravel(order="F")
    assert not np.signbit(got).any() and not got.any()
    xm = J.from_numpy(np.full(n, -0.0, dt), J.domain(rig.A))                  # products of -0: t = sum of (+0 + -0...) = +0; 0 + t = +0; 0 - t ...
    S2 = J.compose(rig.W[0], rig.A) - J.compose(rig.W[1], rig.A)
    from jets_jl_amd import chains

    y1 = J.mul_(J.rand(J.range(S2), seed=4, stream=4), S2, xm)
    chains.ENABLED[0] = False
    try:
        y0 = J.mul_(J.rand(J.range(S2), seed=5, stream=5), S2, xm)
    finally:
        chains.ENABLED[0] = True
    assert_bits_equal(y1.to_numpy(), y0.to_numpy(), "signed zeros through a fused sum")
    rig.close()


def test_a_sum_with_a_bare_operator_that_has_zero_rows_keeps_the_references_loop(Jets, oracle):
    """The reference reuses ONE temporary for all terms (src/Jets.jl:641): a block operator with a zero block leaves that row as the previous term left it
    (1022).  A fused neighbour would not have written the temporary, so such sums are not fused -- the result still equals the reference's loop."""
    from jets_jl_amd import chains

    J = Jets
    dt, n, nrow = np.float64, 1027, 6
    rig = Rig(J, oracle, dt, nrow, n, "mixed")                                # has zero rows
    S = J.compose(rig.W[0], rig.A) + rig.A
    x = J.rand(J.domain(rig.A), seed=9, stream=9)
    before = chains.STATS["sum_terms_fused"]
    y1 = J.mul_(J.zeros(J.range(S)), S, x)
    assert chains.STATS["sum_terms_fused"] == before
    chains.ENABLED[0] = False
    try:
        y0 = J.mul_(J.zeros(J.range(S)), S, x)
    finally:
        chains.ENABLED[0] = True
    assert_bits_equal(y1.to_numpy(), y0.to_numpy(), "stale-row quirk kept")
    rig.close()


# ---------------------------------------------------------------------------------- chains of elementwise stages with no tall operator: one JIT broadcast
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("n", [100, 4099, 1 << 16])
def test_the_references_composition_benchmark_is_one_pass(Jets, oracle, dt, n):
    """G = F o A o F o A with F: d .= m.^2 (JopBar) and A a diagonal (JopFoo) on one plain space -- benchmark/benchmarks.jl:33-38, 55-60, 73-80 (there
    Float64, n = 100).  mul!(d, G, m) is four stages through three temporaries in the reference (src/Jets.jl:524-528); here ONE JIT-compiled elementwise
    pass.  J = jacobian!(G, m): mul!(d, J, dm) and mul!(m, J', d) likewise (530-540 with the children's points set by point!, 578-589).  Bit-identical
    to the stage-by-stage chain on the device, and to numpy applying the stages one by one."""
    from jets_jl_amd import chains

    J = Jets
    spc = J.JetSpace(dt, n)
    a = J.rand(spc, seed=11, stream=0)
    ha = a.to_numpy().ravel(order="F").copy()
    A, F = J.JopDiagonal(a), J.JopSquare(spc)
    G = J.compose(J.compose(J.compose(F, A), F), A)
    hm = u01(oracle, dt, 12, 0, n)
    m = J.from_numpy(hm, spc)

    def both(op, x):
        before = chains.STATS["bcast_calls"]
        y1 = J.mul_(J.rand(spc, seed=5, stream=5), op, x)
        ran = chains.STATS["bcast_calls"] - before
        chains.ENABLED[0] = False
        try:
            y0 = J.mul_(J.rand(spc, seed=6, stream=6), op, x)
        finally:
            chains.ENABLED[0] = True
        return y1.to_numpy().ravel(order="F"), y0.to_numpy().ravel(order="F"), ran

    y1, y0, ran = both(G, m)
    assert ran == 1
    assert_bits_equal(y1, y0, "G(m): one pass vs four stages")

    def cmul(x, y):                                                           # Julia's complex product, every operation rounded in the element type
        if np.dtype(dt).kind != "c":
            return x * y
        rt = np.float32 if dt == np.complex64 else np.float64
        xr, xi, yr, yi = x.real.astype(rt), x.imag.astype(rt), y.real.astype(rt), y.imag.astype(rt)
        out = np.empty(x.shape, dt)
        out.real = xr * yr - xi * yi
        out.imag = xr * yi + xi * yr
        return out

    t1 = cmul(ha, hm)
    t2 = cmul(t1, t1)
    t3 = cmul(ha, t2)
    assert_bits_equal(y1, cmul(t3, t3), "G(m) vs numpy, stage by stage")
    Jg = J.jacobian_(G, m)
    hdm = u01(oracle, dt, 13, 0, n)
    dm = J.from_numpy(hdm, spc)
    j1, j0, ran = both(Jg, dm)
    assert ran == 1
    assert_bits_equal(j1, j0, "J dm: one pass vs four stages")
    a1, a0, ran = both(Jg.H, dm)
    assert ran == 1
    assert_bits_equal(a1, a0, "J' d: one pass vs four stages")


def test_an_elementwise_nonlinear_childs_expression_is_spliced_into_the_chain(Jets, oracle):
    """JopElementwise children carry their f as a C expression: in a chain of elementwise stages the expression is spliced in, with its parameters."""
    from jets_jl_amd import chains

    J = Jets
    dt, n = np.float32, 5000
    spc = J.JetSpace(dt, n)
    a = J.rand(spc, seed=21, stream=0)
    E = J.JopElementwise(spc, "s0*x0*x0 + s1", "2*s0*x0", [0.5, -0.25])
    G = J.compose(J.compose(E, J.JopDiagonal(a)), E)
    m = J.rand(spc, seed=22, stream=0)
    before = chains.STATS["bcast_calls"]
    y1 = J.mul_(J.zeros(spc), G, m)
    assert chains.STATS["bcast_calls"] == before + 1
    chains.ENABLED[0] = False
    try:
        y0 = J.mul_(J.zeros(spc), G, m)
    finally:
        chains.ENABLED[0] = True
    assert_bits_equal(y1.to_numpy(), y0.to_numpy(), "E o A o E")


def test_a_chain_through_the_jacobian_of_a_nonlinear_operator_follows_point(Jets, oracle):
    """J(m0)' o W o J(m0) for a tall nonlinear operator of SQUARE children (test/runtests.jl:19-24): jh_blockop_point moves the rows' coefficient arrays, and the
    chain's row table (built from the operator's blocks when the chain was created) must follow -- fused == stage by stage at the first point and again
    after the operator has been pointed somewhere else."""
    from jets_jl_amd import chains

    J = Jets
    dt, n, nrow = np.float64, 1027, 5
    spc = J.JetSpace(dt, n)
    F = J.blockop([[J.JopSquare(spc)] if i % 2 == 0 else [J.JopDiagonal(J.rand(spc, seed=40 + i, stream=0))] for i in range(nrow)])
    w = J.rand(J.range(F), seed=50, stream=0)
    W = J.JopDiagonal(w)
    x = J.rand(spc, seed=51, stream=0)
    for seed in (60, 61, 62):
        mo = J.rand(spc, seed=seed, stream=0)
        Jm = J.jacobian_(F, mo)
        N = J.compose(J.compose(Jm.H, W), Jm)
        before = chains.STATS["chain_calls"]
        y1 = J.mul_(J.zeros(spc), N, x)
        assert chains.STATS["chain_calls"] == before + 1
        chains.ENABLED[0] = False
        try:
            y0 = J.mul_(J.zeros(spc), N, x)
        finally:
            chains.ENABLED[0] = True
        assert_bits_equal(y1.to_numpy(), y0.to_numpy(), f"J' W J at point {seed}")


SPLIT_CHAINS = ["A' o W o A", "(W o A)' o (W o A)", "M' o A' o W o A o M", "a * (A' o A)", "A' o (a W) o A o (b M)", "M o A'  (range -> domain)", "M o A' o W'"]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("name", SPLIT_CHAINS)
@pytest.mark.parametrize("nrow,n,kinds", [(512, 1027, "diag"), (700, 260, "mixed")])
def test_chains_over_many_small_rows_take_the_split_walk(Jets, oracle, dt, name, nrow, n, kinds):
    """Hundreds of rows of a few KiB: the ordered walk of A' would run on a handful of workgroups, so the library cuts the row sum into parts (adj_split,
    DESIGN.md section 3: tolerance parity, deterministic) -- the fused chain too, its stages after A' and its accumulation applied to the folded sum.  With
    adj_split = 0 (the ordered, bit-exact walk) the fused chain has the bits of the stage-by-stage chain and of the oracle as everywhere else."""
    from jets_jl_amd import chains

    J = Jets
    toks, runs = CHAINS[name]
    rig = Rig(J, oracle, dt, nrow, n, kinds, with_wb=False)
    C = rig.compose(toks)
    rng_in = toks[0] == "At" or (toks[0] != "A" and toks[0][0] == "W")
    xs = J.range(rig.A) if rng_in else J.domain(rig.A)
    hx = [u01(oracle, dt, 91, i, n) for i in range(nrow if rng_in else 1)]
    x = J.from_numpy(np.concatenate(hx), xs)
    want = np.concatenate(rig.ora_apply(toks, hx))
    tol = (2e-5 if np.dtype(dt).itemsize // (2 if np.dtype(dt).kind == "c" else 1) == 4 else 1e-13) * np.sqrt(nrow)
    y1, y0, ran = _run_both(J, C, x, J.range(C), chains)
    assert ran == runs
    scale = np.abs(want).max()
    assert np.abs(y1.to_numpy().ravel(order="F") - want).max() <= tol * scale, f"{name}: split fused chain vs the oracle"
    assert np.abs(y0.to_numpy().ravel(order="F") - want).max() <= tol * scale
    y2 = J.mul_(J.rand(J.range(C), seed=79, stream=3), C, x)
    assert J.tune_get("last_adj_parts") > 1, "the shape was chosen to take the split walk"
    assert_bits_equal(y2.to_numpy().ravel(order="F"), y1.to_numpy().ravel(order="F"), "the split walk is deterministic")
    J.tune(adj_split=0)
    try:
        y1, y0, ran = _run_both(J, C, x, J.range(C), chains)
        assert ran == runs and J.tune_get("last_adj_parts") == 1
        assert_bits_equal(y1.to_numpy().ravel(order="F"), y0.to_numpy().ravel(order="F"), f"{name}: ordered fused vs stage by stage")
        assert_bits_equal(y1.to_numpy().ravel(order="F"), want, f"{name}: ordered fused vs the oracle")
    finally:
        J.tune(adj_split=-1)
    rig.close()


@pytest.mark.parametrize("dt", [np.float32, np.complex128])
def test_a_sum_of_chains_over_many_small_rows(Jets, oracle, dt):
    """A'oWoA + lam*I - (M' o B' o B o M) with 600 rows of 515 elements: the second and third terms ADD themselves to what the output holds after the fold
    (k_chain_finish); against the reference's loop on the device (tolerance: both split their row sums) and bit-exact with adj_split = 0."""
    from jets_jl_amd import chains

    J = Jets
    nrow, n = 600, 515
    ra, rb = Rig(J, oracle, dt, nrow, n, "mixed", seed=31, with_wb=False), Rig(J, oracle, dt, nrow, n, "diag", seed=57, with_wb=False)
    A, B = ra.A, rb.A
    dom = J.domain(A)
    N1 = J.compose(J.compose(A.H, ra.W[0]), A)
    N2 = J.compose(ra.M[0].H, J.compose(J.compose(B.H, B), ra.M[0]))
    S = N1 + 0.25 * J.JopIdentity(dom) - N2
    hx = u01(oracle, dt, 91, 0, n)
    x = J.from_numpy(hx, dom)

    def both():
        before = chains.STATS["sum_terms_fused"]
        y1 = J.mul_(J.rand(dom, seed=77, stream=1), S, x)
        fused = chains.STATS["sum_terms_fused"] - before
        chains.ENABLED[0] = False
        try:
            y0 = J.mul_(J.rand(dom, seed=78, stream=2), S, x)
        finally:
            chains.ENABLED[0] = True
        return y1.to_numpy().ravel(order="F"), y0.to_numpy().ravel(order="F"), fused

    y1, y0, fused = both()
    assert fused == 3                                   # (lam * I adds itself in one lincomb pass)
    tol = (2e-5 if dt == np.float32 else 1e-13) * np.sqrt(nrow) * np.abs(y0).max()
    assert np.abs(y1 - y0).max() <= tol
    J.tune(adj_split=0)
    try:
        y1, y0, fused = both()
        assert fused == 3
        assert_bits_equal(y1, y0, "ordered walk: sum of chains fused vs the reference's loop")
    finally:
        J.tune(adj_split=-1)
    ra.close()
    rb.close()


@pytest.mark.parametrize("dt", DTYPES)
def test_a_block_diagonal_block_operator_stage_accumulates_into_zeros(Jets, oracle, dt):
    """The sign of a zero (found by tools/fuzz_chains.py).  A block-diagonal BLOCK OPERATOR has several block columns, so the reference accumulates each
    of its rows into zeros -- `_d .+= mul!(dtmp, op, _m)` (src/Jets.jl:1024), adjoint `_m .= 0; _m .+= ...` (1042 / 1049) -- and a product of -0 becomes
    +0; a plain diagonal operator (`W`) stores its product and keeps -0.  A zero row of A under a negative scalar makes the -0 (JH_STAGE_ROWSUM)."""
    from jets_jl_amd import chains

    J = Jets
    nrow, n = 8, 70
    rig = Rig(J, oracle, dt, nrow, n, "mixed")                                   # rows of A: diag, diag', identity, scale, ZERO, ...
    for toks in (["A", ("s", -1.25, "r"), ("Wb", 0, False), ("s", 0.375, "r")], ["A", ("s", -1.25, "r"), ("Wb", 1, True)], ["A", ("s", -1.25, "r"), ("W", 0, False)]):
        C = rig.compose(toks)
        hx = [u01(oracle, dt, 91, 0, n)]
        x = J.from_numpy(hx[0], J.domain(rig.A))
        y1, y0, ran = _run_both(J, C, x, J.range(C), chains)
        assert ran == 1
        want = np.concatenate(rig.ora_apply(toks, hx))
        assert_bits_equal(y1.to_numpy().ravel(order="F"), y0.to_numpy().ravel(order="F"), f"{toks}: fused vs stage by stage (signs of zeros included)")
        assert_bits_equal(y1.to_numpy().ravel(order="F"), want, f"{toks}: fused vs the oracle")
    plain = np.concatenate(rig.ora_apply(["A", ("s", -1.25, "r"), ("W", 0, False)], hx))
    parts = plain.view(np.float32 if np.dtype(dt).itemsize // (2 if np.dtype(dt).kind == "c" else 1) == 4 else np.float64)
    assert (np.signbit(parts) & (parts == 0)).any(), "the case must contain a -0 under the plain diagonal"
    rig.close()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nrow,n,kinds", [(6, 1024, "diag"), (9, 1027, "mixed")])
def test_the_regularised_normal_operator_is_two_passes(Jets, oracle, dt, nrow, n, kinds):
    """A'A + lam*I - mu*I (src/Jets.jl:639-655 over (A', A) and scalar terms): the first term IS `0 + A'A x` (the fused pass sums its rows from +0, 640), every
    `a * I` adds itself as d .= d +- T(a x) in one pass (product rounded, then the add: 634) -- no fill, no temporary; the bits of the reference's loop."""
    from jets_jl_amd import chains

    J = Jets
    rig = Rig(J, oracle, dt, nrow, n, kinds, with_wb=False)
    A, dom = rig.A, J.domain(rig.A)
    S = J.compose(A.H, A) + 0.25 * J.JopIdentity(dom) - 1.5 * J.JopIdentity(dom) + J.JopIdentity(dom)
    hx = u01(oracle, dt, 91, 0, n) - dt(0.5)                                       # (both signs: -0 products do not occur, cancellations do)
    x = J.from_numpy(hx.astype(dt), dom)
    before = chains.STATS["sum_terms_fused"]
    y1 = J.mul_(J.rand(dom, seed=77, stream=1), S, x)
    assert chains.STATS["sum_terms_fused"] - before == 4
    chains.ENABLED[0] = False
    try:
        y0 = J.mul_(J.rand(dom, seed=78, stream=2), S, x)
    finally:
        chains.ENABLED[0] = True
    assert_bits_equal(y1.to_numpy().ravel(order="F"), y0.to_numpy().ravel(order="F"), "A'A + lam I - mu I + I: fused vs the reference's loop")
    hxl = [hx.astype(dt)]
    t1 = rig.ora_apply(["A", "At"], hxl)
    want = oracle.barr_lincomb([np.empty(n, dt)], [1.0, 1.0], [[np.zeros(n, dt)], t1])
    for a, sg in ((0.25, 1.0), (1.5, -1.0), (1.0, 1.0)):
        tmp = oracle.barr_lincomb([np.empty(n, dt)], [a], [hxl])
        want = oracle.barr_lincomb([np.empty(n, dt)], [1.0, sg], [want, tmp])
    assert_bits_equal(y1.to_numpy().ravel(order="F"), want[0], "vs the oracle's loop")
    a1 = J.mul_(J.rand(dom, seed=71, stream=1), S.H, x)                            # the adjoint of the sum: the same terms adjointed
    chains.ENABLED[0] = False
    try:
        a0 = J.mul_(J.rand(dom, seed=72, stream=2), S.H, x)
    finally:
        chains.ENABLED[0] = True
    assert_bits_equal(a1.to_numpy().ravel(order="F"), a0.to_numpy().ravel(order="F"), "adjoint of the sum")
    rig.close()
