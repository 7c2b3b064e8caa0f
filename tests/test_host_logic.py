"""CPU: host-side mirror of the reference's operator API -- spaces, operator algebra, dispatch
bookkeeping.  No device arrays are created and no kernels run (the C-ABI library is only loaded).

Re-encodes: test/runtests.jl:58-94 (JetSpace), 296-326 / 425-451 (composition), 453-488 (sums),
622-660 / 697-702 / 720-742 / 760-787 (block operator structure), 789-795 (scalar * op),
797-838 (vec), 840-886 (close), 888-899 (perfstat).
"""
import numpy as np
import pytest

import jets_jl_amd as J

F64 = np.float64


def lin(n, name="A", rng_n=None, **s):
    """A linear operator with no arithmetic attached (structure tests only)."""
    def df(d, m, **kw):
        return d
    df.__name__ = name
    return J.JopLn(dom=J.JetSpace(F64, n), rng=J.JetSpace(F64, rng_n or n), df=df, s=s)


def nonlin(n):
    def f(d, m, **kw):
        return d

    def df(d, m, mo=None, **kw):
        return d
    return J.JopNl(dom=J.JetSpace(F64, n), rng=J.JetSpace(F64, n), f=f, df=df)


# ------------------------------------------------------------------ spaces
@pytest.mark.parametrize("n", [(2,), (2, 3), (2, 3, 4)])
@pytest.mark.parametrize("T", [np.float32, np.float64, np.complex64, np.complex128])
def test_jetspace(n, T):
    R = J.JetSpace(T, *n)                                            # test/runtests.jl:58-66
    assert R.size() == n and R.eltype() == np.dtype(T) and R.ndims() == len(n)
    assert R.length() == int(np.prod(n)) and R.size(0) == n[0]
    assert R == J.JetSpace(T, n) and R.vec() == J.JetSpace(T, int(np.prod(n)))
    assert R.similar(*n) == R


def test_jetbspace_layout_and_accessors():
    R = J.JetBSpace([J.JetSpace(F64, 2), J.JetSpace(F64, 2, 2), J.JetSpace(F64, 2, 3)])
    assert [J.indices(R, i) for i in range(3)] == [range(0, 2), range(2, 6), range(6, 12)]   # Julia 1:2, 3:6, 7:12
    assert R.size() == (12,) and R.length() == 12 and R.ndims() == 1 and R.eltype() == np.dtype(F64)
    assert J.nblocks(R) == 3 and J.nblocks(J.JetSpace(F64, 4)) == 1
    assert R.vec() is R and R.similar(5) == J.JetSpace(F64, 5)
    assert R == J.JetBSpace([J.JetSpace(F64, 2), J.JetSpace(F64, 2, 2), J.JetSpace(F64, 2, 3)])
    assert R != J.JetBSpace([J.JetSpace(F64, 2), J.JetSpace(F64, 4), J.JetSpace(F64, 2, 3)])
    with pytest.raises(TypeError):
        J.JetBSpace([J.JetSpace(np.float32, 2), J.JetSpace(F64, 2)])
    big = J.JetBSpace([J.JetSpace(np.float32, 256, 256, 256)] * 1024)
    assert big.length() == 17_179_869_184 and J.indices(big, 1023).start == 1023 * 256 ** 3


# ------------------------------------------------------------------ Jet / Jop core
def test_jet_constructor_defaults():
    with pytest.raises(ValueError):                                  # src/Jets.jl:178-180
        J.Jet(dom=J.JetSpace(F64, 2), rng=J.JetSpace(F64, 2))
    A = lin(3)
    assert A.jet.f is A.jet.df is A.jet.df_adj                       # :181-186
    assert J.JopLn(A) is A                                           # :223
    assert J.adjoint(J.adjoint(A)) is A and A.H.H is A               # :382-383
    assert J.JopLn(A.H) is not None and isinstance(J.JopLn(A.H), J.JopAdjoint)   # :235
    assert J.domain(A.H) == J.range(A) and J.range(A.H) == J.domain(A)
    F = nonlin(3)
    assert isinstance(J.JopLn(F), J.JopLn) and J.JopLn(F).jet is F.jet              # :224
    assert J.shape(A) == ((3,), (3,)) and J.size(A) == (3, 3) and J.size(lin(3, rng_n=5), 1) == 5
    assert J.eltype(A) == np.dtype(F64)
    with pytest.raises(TypeError):
        J.adjoint(F)


def test_state_and_point():
    A = lin(2, diagonal="x")
    assert J.state(A)["diagonal"] == "x" and J.state(A, "diagonal") == "x"
    J.state_(A, {"extra": 3})
    assert J.state(A) == {"diagonal": "x", "extra": 3}               # merge (:272)
    calls = []
    F = J.JopNl(dom=J.JetSpace(F64, 2), rng=J.JetSpace(F64, 2), f=lambda d, m, **k: d, df=lambda d, m, **k: d,
                upstate=lambda m, s: calls.append(m), s={"J": 1})
    mo = object()
    L = J.jacobian_(F, mo)                                           # :364: shares the jet
    assert isinstance(L, J.JopLn) and L.jet is F.jet and J.point(L) is mo and calls == [mo]
    assert J.jacobian_(L, "other") is L                              # :366


def test_composition_flattens_and_orders():
    A1, A2, A3, A4 = (lin(10, f"A{i}") for i in range(1, 5))
    A21 = A2 @ A1
    A4321 = A4 @ A3 @ A2 @ A1
    assert J.state(A4321)["ops"] == (A4, A3, A2, A1) and len(J.state(A4321)["ops"]) == 4      # :309
    assert isinstance(A4321, J.JopLn) and J.domain(A4321) == J.JetSpace(F64, 10)             # :318
    C = A4 @ A3 @ A21.H                                                                        # :324
    ops = J.state(C)["ops"]
    assert ops[0] is A4 and ops[1] is A3 and ops[2].op is A1 and ops[3].op is A2             # (A2 A1)' = A1' A2'
    F = nonlin(10)
    assert isinstance(A1 @ F, J.JopNl) and isinstance(F @ A1, J.JopNl)                          # :570


def test_composite_state_lookup():
    A, B = lin(2, "A", diagonal=1), lin(2, "B", diagonal=2)
    G = A @ nonlin(2)
    assert J.state(G, "diagonal") == 1                               # :443
    with pytest.raises(KeyError):
        J.state(G, "foo")                                            # :446
    with pytest.raises(KeyError):
        J.state(A @ B, "diagonal")                                   # ambiguous (:450)


def test_sum_sign_flattening():
    A1, A2, A3 = (lin(10, f"A{i}") for i in range(1, 4))
    A12 = A1 + A2
    A123 = A1 + A2 - A3
    assert J.state(A123)["ops"] == (A1, A2, A3) and J.state(A123)["sgns"] == ("+", "+", "-")  # :456
    A12312 = (A12 + A3) - A12                                                                    # :464
    assert J.state(A12312)["ops"] == (A1, A2, A3, A1, A2) and J.state(A12312)["sgns"] == ("+", "+", "+", "-", "-")
    B = A1 - (A2 - A3)
    assert J.state(B)["sgns"] == ("+", "-", "+")                                                # flipsgn (:667-671)
    assert isinstance(A123, J.JopLn) and isinstance(A1 + nonlin(10), J.JopNl)
    assert J.domain(A123) == J.domain(A1) and J.range(A123) == J.range(A1)


def test_scalar_times_operator_structure():
    A = lin(10)
    B = 3.14 * A                                                     # :1161-1164
    ops = J.state(B)["ops"]
    assert len(ops) == 2 and ops[1] is A and ops[0].jet.df is J.constdiag_df and J.state(ops[0])["a"] == 3.14
    S = 1.0 * A - 2.0 * lin(10, "A2")                                # docs/src/index.md: linear combination
    assert J.state(S)["sgns"] == ("+", "-")


# ------------------------------------------------------------------ block operators (structure)
def test_block_operator_spaces_and_queries():
    ops = [[lin(10) for _ in range(4)] for _ in range(3)]
    Z = J.JopZeroBlock(J.JetSpace(F64, 10), J.JetSpace(F64, 10))
    ops[1][1] = Z
    assert J.iszero(Z) and not J.iszero(ops[0][0])                   # test/runtests.jl:629-631
    F = J.blockop(ops)
    assert J.nblocks_op(F) == (3, 4) and J.nblocks_op(F, 1) == 3 and J.nblocks_op(F, 2) == 4   # :645-647
    assert J.nblocks_op(ops[0][0]) == (1, 1)                                                    # :648
    assert J.domain(F).length() == 40 and J.range(F).length() == 30                             # :658-660
    assert isinstance(J.domain(F), J.JetBSpace) and isinstance(F, J.JopLn) and J.isblockop(F) and not J.isblockop(Z)
    assert J.getblock_op(F, 0, 0) is ops[0][0] and J.getblock_op(F, 1, 1) is Z
    ops[0][1] = nonlin(10)
    G = J.blockop(ops)
    assert isinstance(G, J.JopNl)                                                               # :639
    assert J.getblock_op(G, 0, 1, kind=J.JopNl) is ops[0][1]                                    # :650
    assert isinstance(J.getblock_op(G, 0, 0, kind=J.JopLn), J.JopLn)                            # :654


def test_tall_operator_domain_is_a_plain_space():
    B = [lin(5) for _ in range(3)]
    A = J.blockop(B)                                                 # vector form -> 3 x 1 (src/Jets.jl:933)
    assert J.nblocks_op(A) == (3, 1) and isinstance(J.domain(A), J.JetSpace)                    # :927, test :739-741
    assert J.getblock_op(A, 1, 0) is B[1] and isinstance(J.getblock_op(A, 0, 0), J.JopLn)       # :735-736
    Ad = J.blockop([[b] for b in B], dadom=True)
    assert isinstance(J.domain(Ad), J.JetBSpace) and J.nblocks_op(Ad) == (3, 1)                 # dadom keyword (:926-927)
    W = J.blockop([[lin(5), lin(5), lin(5)]])
    assert J.nblocks_op(W) == (1, 3) and J.range(W).length() == 5


def test_blockop_keyword_arguments_land_in_state():
    x = J.blockop([[lin(2)], [lin(2)]], foo=3, bar=4)                # test/runtests.jl:697-702
    assert isinstance(x, J.Jop) and J.state(x)["foo"] == 3 and J.state(x)["bar"] == 4


def test_getblock_of_adjoint_operator():
    B = [[lin(5, f"B{i}{j}", tag=(i, j)) for j in range(3)] for i in range(2)]
    C = J.blockop(B).H
    for i in range(2):
        for j in range(3):
            Cji = J.getblock_op(C, j, i)                             # :766-768
            assert isinstance(Cji, J.JopAdjoint) and Cji.op is B[i][j] and J.state(Cji)["tag"] == (i, j)
    Bh = [[lin(5, tag=(i, j)).H for j in range(3)] for i in range(2)]
    Ch = J.blockop(Bh).H
    for i in range(2):
        for j in range(3):
            Cji = J.getblock_op(Ch, j, i)                            # :781-783
            assert isinstance(Cji, J.JopLn) and J.state(Cji)["tag"] == (i, j)


def test_getblock_of_composite_with_block_operator():
    A1 = lin(2, "A1")
    A2 = J.blockop([nonlin(2), nonlin(2)])
    A = A2 @ A1                                                      # test/runtests.jl:425-431
    A11 = J.getblock_op(A, 0, 0)
    ops = J.state(A11)["ops"]
    assert ops[0] is J.getblock_op(A2, 0, 0) and ops[1] is A1


# ------------------------------------------------------------------ vec
def test_vectorised_operator():
    A = J.JopLn(dom=J.JetSpace(F64, 10, 11), rng=J.JetSpace(F64, 10, 11), df=lambda d, m, **k: d)
    B = J.vec_op(A)                                                  # test/runtests.jl:803-804
    assert B.jet.f is J.JetVec_f and J.domain(B).size() == (110,) and J.range(B).size() == (110,)
    C = lin(7)
    assert J.vec_op(C) is C                                          # already 1-D: no-op (:1130)
    Ab = J.blockop([A, A])
    assert J.range(Ab).size() == J.range(Ab).vec().size() == (220,) and J.domain(Ab).vec().size() == (110,)   # :823-825
    assert J.vec_op(Ab).jet.f is J.JetVec_f


# ------------------------------------------------------------------ close / perfstat
def _closeable(log, name):
    def df(d, m, **kw):
        return d
    op = J.JopLn(dom=J.JetSpace(F64, 2), rng=J.JetSpace(F64, 2), df=df, s={"file": name})
    J.register_close(df, lambda j: log.append(j.s["file"]))
    return op


def test_close_cascades():
    log = []
    A = J.blockop([[_closeable(log, "11"), _closeable(log, "12")], [_closeable(log, "21"), _closeable(log, "22")]])
    J.close(A)                                                       # test/runtests.jl:847-860
    assert sorted(log) == ["11", "12", "21", "22"]
    log.clear()
    J.close(_closeable(log, "a") @ _closeable(log, "b"))             # :862-873
    assert sorted(log) == ["a", "b"]
    log.clear()
    J.close(_closeable(log, "a") + _closeable(log, "b"))             # :875-886
    assert sorted(log) == ["a", "b"]
    assert J.close(lin(2)) is False                                  # default (:290)


def test_perfstat():
    def foo_df(d, m, **kw):
        return d
    A1 = J.JopLn(dom=J.JetSpace(F64, 2), rng=J.JetSpace(F64, 2), df=foo_df)
    J.register_perfstat(foo_df, lambda j: np.pi)                     # test/runtests.jl:9
    A2 = nonlin(2)
    assert J.perfstat(A1) == np.pi and J.perfstat(A2) is None        # :893-894
    assert J.perfstat(A2 @ A1) == np.pi and J.perfstat(A2 + A1) == np.pi   # :895-898


def test_broadcast_expressions_cross_compile_without_a_device():
    """jh_bcast_check: the JIT source generator + hiprtc for gfx950, no GPU needed (the launch path is GPU-tested)."""
    import jets_jl_amd as J
    from jets_jl_amd._ffi import lib

    for dt in range(4):
        assert lib.jh_bcast_check(b"s0*x0 + x1/x2 - 2", dt, 3, 1) == 0, lib.jh_last_error()
        assert lib.jh_bcast_check(b"exp(-abs2(x0)) * conj(x1) + real(x0)", dt, 2, 0) == 0, lib.jh_last_error()
    assert lib.jh_bcast_check(b"x0 +", 0, 1, 0) == 1 and b"x0 +" in lib.jh_last_error()
    assert lib.jh_bcast_check(b"x0", 0, 9, 0) == 1
    e = 2.0 * J.lazy(3.0) + J.bc.exp(J.lazy(1.0))
    code, vecs, scal = e.program()
    assert code == "((s0 * s1) + exp(s2))" and vecs == [] and scal == [2.0, 3.0, 1.0]


def test_bench_self_spawn_propagates_a_failing_rank_without_a_gpu():
    """`python bench.py --gpus 2` with no launcher: the GPU-free supervisor plans the launch before any worker exists (the device
    count comes from a short-lived child); on a box without a GPU it refuses with "no MI355X visible", a non-zero exit and no JSON
    line -- no rank process is ever started (tests/test_bench_supervisor.py covers the supervisor's other paths with fake workers,
    the GPU suite the real flows)."""
    import os
    import subprocess
    import sys

    import torch

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: the GPU suite covers the self-spawned flow")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                         text=True, timeout=120, env=env, cwd=ROOT)
    assert out.returncode != 0
    assert "no MI355X visible" in out.stderr and "exited with" not in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


# ---------------------------------------------------------------------------------- round 6: the chain planner's segmentation (no GPU needed)
def _fake_stages(spec):
    """spec: a string over A (tall), a (its adjoint), B / b (another tall operator), e (an elementwise stage), i (identity), x (opaque)."""
    from jets_jl_amd.chains import Stage

    nats = {"A": object(), "B": object()}
    out = []
    for ch in spec:
        if ch in "AaBb":
            out.append(Stage("tall", None, None, nat=nats[ch.upper()], base=nats[ch.upper()], adj=ch.islower()))
        elif ch == "e":
            out.append(Stage("scale", None, None, a=2.0, flags=0))
        elif ch == "i":
            out.append(Stage("identity", None, None))
        else:
            out.append(Stage("opaque", None, None))
    return out


@pytest.mark.parametrize("spec,want", [
    ("Aa", [("N", 0, 0, 0, 0, 1)]),                                   # A' o A
    ("Aea", [("N", 0, 1, 0, 0, 2)]),                                  # A' o W o A
    ("eAeeae", [("N", 1, 2, 1, 0, 5)]),                               # M' o A' o W2 o W1 o A o M
    ("Ae", [("F", 0, 1, 0, 0, 1)]),                                   # W o A
    ("eA", [("F", 1, 0, 0, 0, 1)]),                                   # A o M
    ("ea", [("T", 0, 1, 0, 0, 1)]),                                   # A' o W'
    ("aee", [("T", 0, 0, 2, 0, 2)]),                                  # M2 o M1 o A'
    ("A", [("op", 0)]),                                               # a bare operator: nothing to fuse
    ("a", [("op", 0)]),
    ("ee", [("op", 0), ("op", 1)]),                                   # elementwise stages with no tall operator to lean on
    ("AexAa", [("F", 0, 1, 0, 0, 1), ("op", 2), ("N", 0, 0, 0, 3, 4)]),   # an opaque stage splits the chain; both sides fuse
    ("Aeb", [("F", 0, 1, 0, 0, 1), ("op", 2)]),                       # B' is not A': forward run, then B' alone
    ("AeeeeeA"[:6] + "a", [("F", 0, 4, 0, 0, 4), ("T", 0, 1, 0, 5, 6)]),   # five range-side stages: four ride with A, the fifth with A'
    ("eeeeeA", [("op", 0), ("F", 4, 0, 0, 1, 5)]),                    # five domain-side stages: the earliest runs alone
    ("Aiea", [("N", 0, 1, 0, 0, 3)]),                                 # an identity stage costs nothing and is not a stage of the kernel
    ("xAa", [("op", 0), ("N", 0, 0, 0, 1, 2)]),
])
def test_chain_planner_segments(spec, want):
    from jets_jl_amd import chains

    st = _fake_stages(spec)
    got = []
    for step in chains._segments(st):
        if step[0] == "op":
            got.append(("op", step[1]))
        else:
            _, ctype, _tall, pre, mid, post, first, last = step
            got.append(({chains.CHAIN_FORWARD: "F", chains.CHAIN_ADJOINT: "T", chains.CHAIN_NORMAL: "N"}[ctype], len(pre), len(mid), len(post), first, last))
    assert got == want
    covered = []
    for step in chains._segments(st):
        covered += [step[1]] if step[0] == "op" else list(range(step[6], step[7] + 1))
    assert covered == list(range(len(st))), "every stage is applied exactly once, in order"
