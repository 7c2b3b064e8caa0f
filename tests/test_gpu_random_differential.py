"""GPU: randomized differential test -- HIP path vs the CPU oracle, bit for bit, over seeded random block operators.

Each case draws an element type, a block-matrix shape (1..5 x 1..5), per-row/column block lengths (including
zero-length and 16-byte-unaligned ones), a kind per block (zero / identity / scale / diag / diag' ) and dirty outputs,
builds the SAME operator through the product package (-> C ABI) and in the oracle, and compares mul!(d, A, m) and
mul!(m, A', d) bitwise.  This sweeps the dispatch between the tall fast path, the vectorised general kernels and the
scalar general kernels, and the reference's quirks (zero-block skip, accumulate-into-dirty-output, zero-then-accumulate).
"""
import numpy as np
import pytest

from .helpers import DTYPES, assert_bits_equal, u01

pytestmark = pytest.mark.gpu

KINDS = ["zero", "identity", "scale", "diag", "diag_adj"]


def _case(rng):
    dt = DTYPES[rng.integers(len(DTYPES))]
    nrow, ncol = int(rng.integers(1, 6)), int(rng.integers(1, 6))
    uniform = rng.random() < 0.5
    pool = [0, 1, 3, 4, 7, 16, 64, 100, 257, 1024, 4100]
    if uniform:
        n = int(rng.choice(pool[1:]))
        lens = [n] * max(nrow, ncol)
    else:
        lens = [int(rng.choice(pool)) for _ in range(max(nrow, ncol))]
    # elementwise blocks are square: block (i, j) needs len_r[i] == len_c[j]; otherwise it must be a zero block
    len_r, len_c = lens[:nrow], lens[:ncol]
    kinds = []
    for i in range(nrow):
        row = []
        for j in range(ncol):
            if len_r[i] != len_c[j]:
                row.append("zero")
            else:
                row.append(KINDS[rng.integers(len(KINDS))] if rng.random() < 0.85 else "zero")
        kinds.append(row)
    return dt, nrow, ncol, len_r, len_c, kinds


def _typed(a, which, dt):
    """The same value as a scalar of another TYPE -- Julia's arithmetic follows it (tests/test_scalar_types.py): a plain Python number (taken in
    the element type), numpy's 32-bit and 64-bit scalars (the latter promoted arithmetic against 32-bit elements: such an operator runs
    the per-block loop), and for complex element types a Real scalar and a Complex one with a zero imaginary part."""
    cplx = np.dtype(dt).kind == "c"
    kinds = [lambda v: v, np.complex64 if cplx else np.float32, np.complex128 if cplx else np.float64, lambda v: v]
    if cplx:
        kinds += [lambda v: float(v.real), lambda v: np.float64(v.real), lambda v: complex(v.real, 0.0)]
    return kinds[which % len(kinds)](a)


def _build(Jets, oracle, dt, len_r, len_c, kinds, seed):
    dev_rows, ora_rows = [], []
    for i, row in enumerate(kinds):
        dr, orow = [], []
        for j, k in enumerate(row):
            nr, nc = len_r[i], len_c[j]
            dom, rng_ = Jets.JetSpace(dt, nc), Jets.JetSpace(dt, nr)
            if k == "zero":
                dr.append(Jets.JopZeroBlock(dom, rng_)); orow.append(oracle.Block("zero", nr, nc))
            elif k == "identity":
                dr.append(Jets.JopIdentity(dom)); orow.append(oracle.Block("identity", nr))
            elif k == "scale":
                a = (0.3 + 0.5 * i - 0.25 * j) - (0.125j * (j + 1) if np.dtype(dt).kind == "c" else 0)
                a = _typed(a, i + 3 * j + seed, dt)
                dr.append(Jets.JopLn(dom=dom, rng=dom, df=Jets.constdiag_df, df_adj=Jets.constdiag_df_adj, s={"a": a}))
                orow.append(oracle.Block("scale", nr, scale=a))
            else:
                stream = 1000 * i + j
                op = Jets.JopDiagonal(Jets.rand(dom, seed=seed, stream=stream))
                hb = oracle.Block("diag", nr, coeff=u01(oracle, dt, seed, stream, nr), adjoint=(k == "diag_adj"))
                dr.append(op.H if k == "diag_adj" else op); orow.append(hb)
        dev_rows.append(dr); ora_rows.append(orow)
    return Jets.blockop(dev_rows), ora_rows


def _split(v, lens):
    offs = np.cumsum([0] + list(lens))
    return [v[offs[i]:offs[i + 1]].copy() for i in range(len(lens))]


@pytest.mark.parametrize("case", range(60))
def test_random_block_operator_matches_oracle_bitwise(Jets, oracle, case):
    rng = np.random.default_rng(10_000 + case)
    dt, nrow, ncol, len_r, len_c, kinds = _case(rng)
    while sum(len_r) == 0 or sum(len_c) == 0:                 # an entirely empty range or domain is not an operator: redraw
        dt, nrow, ncol, len_r, len_c, kinds = _case(rng)
    A, ops = _build(Jets, oracle, dt, len_r, len_c, kinds, seed=500 + case)
    tag = f"case {case}: {np.dtype(dt).name} {nrow}x{ncol} rows={len_r} cols={len_c} kinds={kinds}"
    NR, NC = sum(len_r), sum(len_c)
    # forward into a dirty range vector
    m = Jets.rand(Jets.domain(A), seed=1, stream=case)
    d = Jets.rand(Jets.range(A), seed=2, stream=case)
    hm, hd = u01(oracle, dt, 1, case, NC), u01(oracle, dt, 2, case, NR)
    Jets.mul_(d, A, m)
    ref_d = oracle.block_df(ops, _split(hd, len_r), _split(hm, len_c))
    assert_bits_equal(d.to_numpy(), np.concatenate(ref_d) if ref_d else np.empty(0, dt), "forward, " + tag)
    # adjoint into a dirty domain vector
    dd = Jets.rand(Jets.range(A), seed=3, stream=case)
    mt = Jets.rand(Jets.domain(A), seed=4, stream=case)
    hdd, hmt = u01(oracle, dt, 3, case, NR), u01(oracle, dt, 4, case, NC)
    Jets.mul_(mt, A.H, dd)
    ref_m = oracle.block_df_adj(ops, _split(hmt, len_c), _split(hdd, len_r))
    assert_bits_equal(mt.to_numpy().ravel(order="F"), np.concatenate(ref_m), "adjoint, " + tag)
