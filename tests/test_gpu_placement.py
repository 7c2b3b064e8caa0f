"""GPU: Jets.stream_pair (jets.jl_amd/placement.py) -- two vectors of a block space ordered (read side, write side) by a measured probe.  On
MI355X the tall forward between two 64 GiB slabs runs up to 10 % apart between the two directions, differently in every process
(profiles/exp_r03_swap_roles.txt); the probe itself is exercised here at a small size."""
from __future__ import annotations

import numpy as np
import pytest

from .helpers import assert_bits_equal

pytestmark = pytest.mark.gpu


def test_small_spaces_are_returned_in_allocation_order_unprobed(Jets):
    J = Jets
    R = J.JetBSpace([J.JetSpace(np.float32, 64, 64)] * 5)
    x, y, info = J.stream_pair(R)
    assert info == {"probed": False}
    assert x.ptr != y.ptr and J.space(x) == R and J.space(y) == R
    plain = J.JetSpace(np.float32, 1 << 20)                          # not a block space: no operator to probe with
    x2, y2, info2 = J.stream_pair(plain)
    assert info2 == {"probed": False} and x2.length() == y2.length() == 1 << 20


@pytest.mark.parametrize("candidates", [2, 3])
def test_the_probe_measures_every_ordered_pair_and_keeps_two_vectors(Jets, oracle, monkeypatch, candidates):
    J = Jets
    from jets_jl_amd import placement

    import gc

    monkeypatch.setattr(placement, "PROBE_FROM_BYTES", 1 << 20)
    gc.collect()                                                       # (vectors of earlier tests go to the cache now, not in the middle of this one)
    J.trim()
    assert J.tune_get("slab_cached_mib") == 0
    blk = J.JetSpace(np.float32, 128, 128, 64)                         # 4 MiB blocks, 8 of them: a 32 MiB vector (cached when destroyed)
    R = J.JetBSpace([blk] * 8)
    x, y, info = J.stream_pair(R, candidates=candidates)
    assert info["probed"] and info["candidates"] == candidates
    assert len(info["pair_ms_other"]) == candidates * (candidates - 1) - 1
    assert info["pair_ms_kept"] == pytest.approx(info["fwd_ms_kept"] + info["adj_ms_kept"])
    assert info["pair_ms_kept"] <= min(info["pair_ms_other"]) + 0.5e-3        # (the others are reported rounded to a microsecond)
    assert x.ptr != y.ptr
    assert J.tune_get("slab_cached_mib") == (candidates - 2) * 32       # the candidates that were not kept went back (to the slab cache)
    # the two vectors are ordinary vectors: the operator built on them gives the oracle's bits
    n = blk.length()
    J.rand_(x, seed=1, stream=0)
    J.rand_(y, seed=3, stream=0)
    assert_bits_equal(x.to_numpy(), oracle.rng_u01(np.float32, 1, 0, 0, 8 * n), "rand_ writes rand's values")
    A = J.blockop([[J.JopDiagonal(c)] for c in x.arrays])
    m = J.rand(blk, seed=2, stream=0)
    J.mul_(y, A, m)
    ha, hm = oracle.rng_u01(np.float32, 1, 0, 0, 8 * n), oracle.rng_u01(np.float32, 2, 0, 0, n)
    ref = oracle.block_df([[oracle.Block("diag", n, coeff=ha[i * n:(i + 1) * n].copy())] for i in range(8)], [np.zeros(n, dtype=np.float32) for _ in range(8)], [hm])
    assert_bits_equal(y.to_numpy(), np.concatenate(ref), "forward into the write side")
    J.close(A)
    J.trim()
