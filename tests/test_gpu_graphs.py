"""GPU: hipGraph replay of the launch-bound per-block loop (operators with DENSE children; jh_blockop.hip: run_loop_graphed).
Call 1 with a given (output, input) pair is eager, call 2 is captured, calls 3+ replay the graph with one launch.
The bar is the eager path's: forward bit-exact vs the oracle, dense adjoint within tolerance (wave reduction).
Since round 2 operators whose dense children are all small run the whole loop in ONE launch (tests/test_gpu_small_loop.py);
the per-block loop and its graphs remain for bigger children, so these tests switch the one-launch loop off."""
import numpy as np
import pytest

from .helpers import assert_bits_equal, u01

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _per_block_loop(Jets):
    Jets.tune(small_loop=0)
    yield
    Jets.tune(small_loop=1)


def _dense_mix(Jets, oracle, dt, seed):
    from .test_gpu_nonlinear import _mixed, _split
    len_r, len_c = [40, 24, 40], [40, 24, 40, 40]
    kinds = [["dense", "dense", "diag", "zero"], ["dense", "dense", "dense", "dense"], ["identity", "dense", "dense", "scale"]]
    A, ops = _mixed(Jets, oracle, dt, len_r, len_c, kinds, seed=seed, hmo=None)
    return A, ops, len_r, len_c, _split


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64])
def test_per_block_loop_replayed_as_a_graph_gives_the_eager_bits(Jets, oracle, dt):
    """Operators with dense children run the reference's per-block loop (2 launches per block).  Call 1 is eager, call 2
    is captured into a hipGraph, calls 3+ replay it: same bits every time, with fresh vector CONTENTS each call (the graph
    reads the buffers, not a snapshot) and with the knob off."""
    A, ops, len_r, len_c, _split = _dense_mix(Jets, oracle, dt, seed=1234)
    m = Jets.zeros(Jets.domain(A))
    d = Jets.zeros(Jets.range(A))
    mt = Jets.zeros(Jets.domain(A))
    NR, NC = sum(len_r), sum(len_c)
    replays0 = Jets.tune_get("graph_replays")
    for call in range(5):
        if call == 4:
            Jets.tune(graphs=0)
        try:
            hm = u01(oracle, dt, 70, call, NC)
            hd0 = u01(oracle, dt, 71, call, NR)
            Jets.copyto_(m, Jets.from_numpy(hm, Jets.domain(A)))
            Jets.copyto_(d, Jets.from_numpy(hd0, Jets.range(A)))                      # dirty output: ncol > 1 accumulates into it
            Jets.mul_(d, A, m)
            ref = oracle.block_df(ops, _split(hd0, len_r), _split(hm, len_c))
            assert_bits_equal(d.to_numpy(), np.concatenate(ref), f"forward, call {call}")
            Jets.mul_(mt, A.H, d)
            refm = oracle.block_df_adj(ops, _split(u01(oracle, dt, 72, call, NC), len_c), ref)
            tol = 2e-5 if np.dtype(dt).itemsize <= 8 and np.dtype(dt) != np.float64 else 1e-12
            np.testing.assert_allclose(mt.to_numpy(), np.concatenate(refm), rtol=tol, atol=tol)   # dense adjoint: wave reduction
        finally:
            Jets.tune(graphs=1)
        # forward + adjoint: eager on call 0, captured-and-launched on call 1, replayed on calls 2 and 3, eager again with the knob off
        assert Jets.tune_get("graph_replays") - replays0 == 2 * min(call, 3), f"call {call}"


def test_graph_replay_survives_scratch_growth_and_other_operators(Jets, oracle):
    """A captured loop holds the context's scratch pointer: when a bigger operator makes the scratch buffer move, the
    stale graph must be dropped and re-captured, not replayed."""
    dt = np.float32
    A, ops, len_r, len_c, _split = _dense_mix(Jets, oracle, dt, seed=4321)
    m = Jets.rand(Jets.domain(A), seed=80, stream=0)
    hm = u01(oracle, dt, 80, 0, sum(len_c))
    d = Jets.zeros(Jets.range(A))
    want = np.concatenate(oracle.block_df(ops, [np.zeros(n, dt) for n in len_r], _split(hm, len_c)))
    for _ in range(3):                                                               # eager, capture, replay
        Jets.fill_(d, 0)
        Jets.mul_(d, A, m)
        assert_bits_equal(d.to_numpy(), want, "before growth")
    big = Jets.blockop([[Jets.JopDense(Jets.rand(Jets.JetSpace(dt, 600_000, 2), seed=85, stream=j)) for j in range(2)]])
    xb = Jets.rand(Jets.domain(big), seed=86, stream=0)
    for _ in range(3):
        Jets.mul(big, xb)                                                            # 2.4 MB dtmp > the 1 MiB scratch: it is reallocated
    for _ in range(3):
        Jets.fill_(d, 0)
        Jets.mul_(d, A, m)
        assert_bits_equal(d.to_numpy(), want, "after the scratch buffer moved")
