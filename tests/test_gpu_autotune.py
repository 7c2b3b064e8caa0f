"""GPU: the tall forward's grid walk is chosen lazily -- every candidate runs as one of the caller's own calls (same bits),
jh_blockop_mul never synchronises, and the choice can be exported / imported per operator (include/jetship.h)."""
import numpy as np
import pytest

from .helpers import assert_bits_equal, u01

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt", [np.float32, np.complex64, np.float64])
def test_lazy_forward_autotune_converges_without_changing_bits(Jets, oracle, dt):
    J = Jets
    nrow, n = 64, (64 << 20) // np.dtype(dt).itemsize           # 2 x 64 x 64 MiB = 8 GiB streamed: the autotuned regime
    spc = J.JetSpace(dt, n)
    coeff = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    m = J.rand(spc, seed=2, stream=0)
    d = J.zeros(J.range(A))
    ha0, ha9 = u01(oracle, dt, 1, 0, 4096), oracle.rng_u01(dt, 1, 0, 9 * n + 12345, 4096)
    hm0, hm9 = u01(oracle, dt, 2, 0, 4096), oracle.rng_u01(dt, 2, 0, 12345, 4096)
    def prod(a, b):
        """a .* b as Julia rounds it: for complex elements the four products and the two sums, each rounded (numpy's complex multiply may contract)"""
        if np.dtype(dt).kind != "c":
            return a * b
        out = np.empty_like(a)
        out.real = a.real * b.real - a.imag * b.imag
        out.imag = a.real * b.imag + a.imag * b.real
        return out

    ha0, ha9 = prod(ha0, hm0), prod(ha9, hm9)                     # (the expected slices of rows 0 and 9)
    assert J.op_tune_get(A, "fwd_walk") is None or J.op_tune_get(A, "fwd_walk") == -1
    walks = set()
    for call in range(27):                                      # 20 trials (ten candidate walks below 1024 rows), a play-off of 4 when the two best are within 3 %, the harvesting calls
        J.fill_(d, 0)
        J.mul_(d, A, m)
        walks.add((J.tune_get("last_fwd_walk"), J.tune_get("last_fwd_rows_per_wg")))
        flat = d.to_numpy() if call in (0, 1, 5, 11, 13, 15, 17, 19, 22, 26) else None  # slices of two rows, bit for bit, under whichever candidate ran
        if flat is not None:
            assert_bits_equal(flat[:4096], ha0, f"call {call}: row 0")
            assert_bits_equal(flat[9 * n + 12345:9 * n + 12345 + 4096], ha9, f"call {call}: row 9")
    assert len(walks) >= 3 and any(w[0] == 2 for w in walks), "the first calls must have tried several candidate shapes, the column bands among them"
    for _ in range(4):                                          # the choice is made by the first call that finds every trial FINISHED (the library never waits):
        if J.op_tune_get(A, "fwd_walk") != -1:                  # the host runs ahead of the device, so let the last play-off trial complete (round 5: this
            break                                               # raced once in a while)
        J.synchronize()
        J.mul_(d, A, m)
    trials, po = J.op_tune_get(A, "fwd_trials"), J.op_tune_get(A, "fwd_playoff")
    assert (trials, po >= 0) in ((20, False), (24, True)), "20 timed calls, or 24 with a play-off between the two best"
    pick = J.op_tune_get(A, "fwd_walk")
    assert 0 <= pick < 10, "after the timed calls (+ their completion) the choice is made"
    if po >= 0:
        assert pick in (po // 16, po % 16), "the play-off is between the winner and the runner-up of the regular trials"
    # the periodic re-check (every 64th call timed, three slow samples in a row rotate the runner-up in) never changes the bits either
    for call in range(70):
        J.mul_(d, A, m)
    flat = d.to_numpy()
    assert_bits_equal(flat[:4096], ha0, "after the re-check: row 0")
    assert J.op_tune_get(A, "fwd_switches") >= 0 and 0 <= J.op_tune_get(A, "fwd_walk") < 10
    pick = J.op_tune_get(A, "fwd_walk")
    # round 4: a NEW operator of the same shape starts with what this one found (no trials of its own; the re-check still applies) ...
    B = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    J.mul_(d, B, m)
    assert J.op_tune_get(B, "fwd_walk") == pick and J.op_tune_get(B, "fwd_walk_inherited") == 1 and J.op_tune_get(B, "fwd_trials") == 0
    assert_bits_equal(d.to_numpy()[:4096], ha0, "inherited walk: row 0")
    J.close(B)
    # ... unless the knob says every operator measures
    J.tune(walk_memory=0)
    try:
        B = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
        J.mul_(d, B, m)                                         # builds the handle (and runs trial 0)
        assert J.op_tune_get(B, "fwd_walk") == -1 and J.op_tune_get(B, "fwd_trials") == 1 and J.op_tune_get(B, "fwd_walk_inherited") == 0
    finally:
        J.tune(walk_memory=1)
    # export / import: a second operator starts in the steady state, and -1 measures again (also with the memory on)
    J.op_tune_set(B, "fwd_walk", pick)
    J.mul_(d, B, m)
    assert J.op_tune_get(B, "fwd_walk") == pick and J.op_tune_get(B, "fwd_trials") == 0
    J.op_tune_set(B, "fwd_walk", -1)
    J.mul_(d, B, m)
    assert J.op_tune_get(B, "fwd_trials") == 1
    with pytest.raises(Exception):
        J.op_tune_set(B, "fwd_walk", 99)


def test_the_column_persistent_candidate_never_runs_on_small_blocks(Jets, oracle):
    """Candidate 5 (a workgroup keeps its m tile and streams EVERY block row through it) has one workgroup per 128 KiB of a row: on rows of
    1 MiB that is eight workgroups walking thousands of rows -- 17.8 ms where the other walks take 1.4-1.7 (4096 x 64^3).  Its TRIAL therefore
    ran candidate 0's shape on small blocks, but round 4 stored a "5" that won such a trial (a tie, a play-off) and from then on launched the
    real column-persistent walk (round-4 advisor finding).  Since round 5 the substitution is part of what candidate 5 IS on such rows -- trial,
    chosen walk, re-check and inherited choice alike (jh_tall.hip: fwd_candidate_shape)."""
    J = Jets
    nrow, edge = 4096, 64                                       # 1 MiB rows, 8 GiB streamed per forward: inside the lazily measured regime
    dt = np.float32
    spc = J.JetSpace(dt, edge, edge, edge)
    n = edge ** 3
    coeff = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    m = J.rand(spc, seed=2, stream=0)
    d = J.zeros(J.range(A))
    want = oracle.rng_u01(dt, 1, 0, 77 * n + 4096, 4096) * oracle.rng_u01(dt, 2, 0, 4096, 4096)
    J.op_tune_set(A, "fwd_walk", 5)                             # "the measurement chose 5"
    for _ in range(3):
        J.mul_(d, A, m)
        assert J.tune_get("last_fwd_rows_per_wg") == 16, "candidate 5 on rows below 4 M packs runs candidate 0's 16-row sweep, not one workgroup per column of tiles"
    e0 = J.Event().record()
    J.mul_(d, A, m)
    e1 = J.Event().record()
    assert e0.elapsed_ms(e1) < 6.0, "the column-persistent walk itself takes ~18 ms on this shape"
    assert_bits_equal(d._download(77 * n + 4096, 4096), want, "row 77 under the substituted shape")
    B = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])   # a new operator of the shape: whatever it inherits or measures, never the real candidate 5
    for _ in range(30):
        J.mul_(d, B, m)
        assert J.tune_get("last_fwd_rows_per_wg") < nrow
    J.close(A)
    J.close(B)
