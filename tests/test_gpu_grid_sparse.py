"""Sparse M x K grids on the step-list walk of the register-tiled general kernel (k_general_tile LIST, late round 5): per group of four
lines the kernel visits only the summed block indices at which some line of the group has a non-zero block.  The reference skips zero
blocks itself (src/Jets.jl:1022 forward, 1047 adjoint), so leaving their steps out changes no term and no order: the oracle's bits."""
import numpy as np
import pytest

from .helpers import assert_bits_equal, u01
from .test_gpu_blockop import _mixed_ops

pytestmark = pytest.mark.gpu

DTYPES = [np.float32, np.float64, np.complex64, np.complex128]


def _pattern(name, M, K, rng):
    names = ["diag", "diag_adj", "identity", "scale"]
    pick = lambda: names[rng.integers(len(names))]
    kinds = [["zero"] * K for _ in range(M)]
    for i in range(M):
        for j in range(K):
            if name == "blockdiag":
                on = i == j
            elif name == "bidiag":
                on = i == j or i == j + 1
            elif name == "arrow":
                on = i == j or i == 0 or j == 0
            elif name == "band":
                on = abs(i - j) <= 2
            elif name == "rand":
                on = rng.random() < 0.15
            elif name == "holes":                      # whole groups of four lines without a block, in both directions
                on = (i // 4) % 2 == 0 and (j // 4) % 2 == 1 and rng.random() < 0.5
            else:
                raise AssertionError(name)
            if on:
                kinds[i][j] = pick()
    return kinds


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(4, 4), (9, 7), (16, 16), (6, 33), (37, 5)])
@pytest.mark.parametrize("name", ["blockdiag", "bidiag", "arrow", "band", "rand", "holes"])
def test_sparse_grids_walk_their_step_lists_with_the_oracles_bits(Jets, oracle, dt, shape, name):
    J = Jets
    M, K = shape
    n = 1024 + 64                                                      # a full tile and a ragged one, 16-byte multiples for every eltype
    rng = np.random.default_rng(1000 * M + K + 7 * sum(map(ord, name)))
    kinds = _pattern(name, M, K, rng)
    A, ops = _mixed_ops(J, oracle, dt, kinds, [n] * M, [n] * K)
    hm = [u01(oracle, dt, 31, j, n) for j in range(K)]
    hd = [u01(oracle, dt, 32, i, n) for i in range(M)]
    hmt = [u01(oracle, dt, 33, j, n) for j in range(K)]
    want_d = oracle.block_df(ops, [b.copy() for b in hd], hm)          # forward into a DIRTY d (`_d .+=`, 1024; rows of zero blocks stay as found, 1022)
    want_m = oracle.block_df_adj(ops, [b.copy() for b in hmt], want_d)  # adjoint into a dirty m (zeroed, 1042)
    used = {}
    try:
        for route in (3, 2, 1, 0):                                     # the per-line lists / the four-line lists always / by the automatic rule / never
            J.tune(general_list=route)
            m = J.from_numpy(np.concatenate(hm), J.domain(A))
            d = J.from_numpy(np.concatenate(hd), J.range(A))
            J.mul_(d, A, m)
            fwd_list = J.tune_get("last_general_list")
            assert_bits_equal(d.to_numpy(), np.concatenate(want_d), f"{M}x{K} {name} forward, general_list={route}")
            mt = J.from_numpy(np.concatenate(hmt), J.domain(A))
            J.mul_(mt, A.H, d)
            adj_list = J.tune_get("last_general_list")
            assert_bits_equal(mt.to_numpy(), np.concatenate(want_m), f"{M}x{K} {name} adjoint, general_list={route}")
            used[route] = (fwd_list, adj_list)
    finally:
        J.tune(general_list=1)
    assert used[0] == (0, 0)
    assert used[2] == (1 if M >= 4 else 0, 1 if K >= 4 else 0), used    # four-line lists exist for every direction with at least four lines
    assert used[3] == (2, 2), used
    J.close(A)


def test_the_automatic_rule_takes_the_lists_only_where_they_drop_steps(Jets):
    """A block-diagonal 16 x 16 grid walks 4 of 16 steps per line group (lists); a full grid with a single zero block keeps the plain walk."""
    J = Jets
    n = 4096
    spc = J.JetSpace(np.float32, n)

    def grid(pred):
        return J.blockop([[J.JopDiagonal(J.rand(spc, seed=3, stream=16 * i + j)) if pred(i, j) else J.JopZeroBlock(spc, spc) for j in range(16)] for i in range(16)])

    m = J.rand(J.JetBSpace([spc] * 16), seed=4, stream=0)
    for pred, want in ((lambda i, j: i == j, 1), (lambda i, j: (i, j) != (3, 5), 0)):
        A = grid(pred)
        d = J.zeros(J.range(A))
        J.mul_(d, A, m)
        assert J.tune_get("last_general_list") == want
        J.mul_(J.zeros(J.domain(A)), A.H, d)
        assert J.tune_get("last_general_list") == want
        J.close(A)


def test_a_big_sparse_grid_measures_its_walk_over_its_first_calls_and_keeps_its_bits(Jets):
    """16 x 16 block-bidiagonal of 4 MiB blocks (124 MiB of coefficients per call): the first seven calls per direction each run ONE candidate walk
    (four-line lists / per-line lists / plain) between two events, then the operator keeps the fastest -- every call with the bits of the plain walk;
    the choice can be read, forced and measured again through the per-operator knobs."""
    J = Jets
    n = 1 << 20
    spc = J.JetSpace(np.float32, n)
    A = J.blockop([[J.JopDiagonal(J.rand(spc, seed=5, stream=16 * i + j)) if i == j or i == j + 1 else J.JopZeroBlock(spc, spc) for j in range(16)] for i in range(16)])
    m = J.rand(J.domain(A), seed=6, stream=0)
    d0 = J.rand(J.range(A), seed=7, stream=0)
    try:
        J.tune(general_list=0)
        d = J.copyto_(J.zeros(J.range(A)), d0)
        J.mul_(d, A, m)
        mt = J.mul_(J.zeros(J.domain(A)), A.H, d)
        want = (d.to_numpy().tobytes(), mt.to_numpy().tobytes())
        J.tune(general_list=1)
        assert J.op_tune_get(A, "gen_walk_fwd") == -1 and J.op_tune_get(A, "gen_walk_adj") == -1
        seen = set()
        for _ in range(12):
            J.copyto_(d, d0)
            J.mul_(d, A, m)
            seen.add(J.tune_get("last_general_list"))
            J.mul_(mt, A.H, d)
            assert (d.to_numpy().tobytes(), mt.to_numpy().tobytes()) == want
        assert seen == {0, 1, 2}                                           # every candidate ran as a trial
        assert J.op_tune_get(A, "gen_walk_fwd") in (0, 1, 2) and J.op_tune_get(A, "gen_walk_adj") in (0, 1, 2)
        assert J.op_tune_get(A, "gen_trials") == 14                        # a warm-up + three candidates x two passes, per direction
        for walk, lst in ((2, 0), (0, 1), (1, 2)):                         # forced choices
            J.op_tune_set(A, "gen_walk_fwd", walk)
            J.copyto_(d, d0)
            J.mul_(d, A, m)
            assert J.tune_get("last_general_list") == lst and d.to_numpy().tobytes() == want[0]
        J.op_tune_set(A, "gen_walk_fwd", -1)                               # measure again
        J.copyto_(d, d0)
        J.mul_(d, A, m)
        assert J.op_tune_get(A, "gen_walk_fwd") == -1 and J.op_tune_get(A, "gen_trials") == 7 + 1 and d.to_numpy().tobytes() == want[0]
    finally:
        J.tune(general_list=1)
    J.close(A)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("aligned", [True, False], ids=["packs", "odd-lengths"])
@pytest.mark.parametrize("M", [6, 21, 40])
def test_ragged_sparse_grids_walk_step_lists_on_the_one_line_kernels(Jets, oracle, dt, aligned, M):
    """Block-diagonal / block-banded grids whose blocks differ in length run on the one-line general kernels (16-byte packs, or element by element when a length
    or offset is odd); they too walk each line's step list when the grid is sparse -- the same terms in the same order, the oracle's bits; a line without any
    block keeps d as found (1022) / is zeroed in the adjoint (1042)."""
    J = Jets
    rng = np.random.default_rng(50 + M)
    per16 = 16 // np.dtype(dt).itemsize
    lens = [int(rng.choice([3, 8, 20, 65])) * (per16 if aligned else 1) + (0 if aligned else int(rng.integers(0, 2))) for _ in range(M)]
    names = ["diag", "diag_adj", "identity", "scale"]
    kinds = [["zero"] * M for _ in range(M)]
    for i in range(M):
        for j in range(M):
            if lens[i] == lens[j] and (i == j or abs(i - j) == 3 or rng.random() < 0.05) and i != M - 1 and j != 1:   # row M-1 and column 1 hold no block
                kinds[i][j] = names[rng.integers(len(names))]
    A, ops = _mixed_ops(J, oracle, dt, kinds, lens, lens)
    hm = [u01(oracle, dt, 61, j, lens[j]) for j in range(M)]
    hd = [u01(oracle, dt, 62, i, lens[i]) for i in range(M)]
    hmt = [u01(oracle, dt, 63, j, lens[j]) for j in range(M)]
    want_d = oracle.block_df(ops, [b.copy() for b in hd], hm)
    want_m = oracle.block_df_adj(ops, [b.copy() for b in hmt], want_d)
    try:
        for route in (1, 0):
            J.tune(general_list=route, adj_split=0)
            d = J.from_numpy(np.concatenate(hd), J.range(A))
            J.mul_(d, A, J.from_numpy(np.concatenate(hm), J.domain(A)))
            assert_bits_equal(d.to_numpy(), np.concatenate(want_d), f"ragged {M} x {M} forward, general_list={route}")
            mt = J.from_numpy(np.concatenate(hmt), J.domain(A))
            J.mul_(mt, A.H, d)
            assert_bits_equal(mt.to_numpy(), np.concatenate(want_m), f"ragged {M} x {M} adjoint, general_list={route}")
    finally:
        J.tune(general_list=1, adj_split=-1)
    assert_bits_equal(d.to_numpy()[-lens[-1]:], hd[-1], "the row without blocks keeps d as found")
    J.close(A)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nrow,keep", [(9, 2), (64, 8), (37, 5), (5, 7)])
def test_a_tall_operator_with_many_zero_rows_launches_only_its_other_rows(Jets, oracle, dt, nrow, keep):
    """Tall operators whose rows are mostly zero blocks (muted shots): the forward covers the listed non-zero rows only (row i of the launch = the list's i-th
    row); every zero row keeps d as found (src/Jets.jl:1022), the others the oracle's bits; the adjoint and the fused A'A are unchanged."""
    J = Jets
    n = 2048 + 64
    names = ["diag", "diag_adj", "identity", "scale"]
    kinds = [[names[(i // keep) % 4] if i % keep == 0 else "zero"] for i in range(nrow)]
    A, ops = _mixed_ops(J, oracle, dt, kinds, [n] * nrow, [n])
    hm = [u01(oracle, dt, 71, 0, n)]
    hd = [u01(oracle, dt, 72, i, n) for i in range(nrow)]
    want_d = oracle.block_df(ops, [b.copy() for b in hd], hm)
    want_m = oracle.block_df_adj(ops, [np.zeros(n, dt)], want_d)
    try:
        for route in (1, 0):
            J.tune(general_list=route)
            d = J.from_numpy(np.concatenate(hd), J.range(A))
            J.mul_(d, A, J.from_numpy(hm[0], J.domain(A)))
            assert_bits_equal(d.to_numpy(), np.concatenate(want_d), f"tall {nrow} rows, one in {keep} non-zero, forward, general_list={route}")
            mt = J.mul_(J.rand(J.domain(A), seed=3, stream=3), A.H, d)
            assert_bits_equal(mt.to_numpy().ravel(order="F"), want_m[0], f"adjoint, general_list={route}")
    finally:
        J.tune(general_list=1)
    for i in range(nrow):
        if i % keep:
            assert_bits_equal(d.to_numpy()[i * n:(i + 1) * n], hd[i], "a zero row keeps d as found")
    J.close(A)
