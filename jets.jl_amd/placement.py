"""Which of two equally sized device vectors should a streaming kernel WRITE?

On MI355X the rate of a kernel that reads one huge vector and writes another depends on where the two lie physically, and NOT
symmetrically: with the same two 64 GiB slabs the tall forward d_i = a_i .* m runs at 20.7-20.9 ms reading the first and writing
the second, and at 23.4-23.5 ms the other way round -- which of the two directions is the fast one changes from process to
process (profiles/exp_r03_swap_roles.txt; over three 64 GiB candidates typically ONE is slow to write into, 24.3-24.8 ms from either
of the others, and fine to read from: profiles/exp_r03_step_placement.txt; on some boxes none is).  The adjoint reads both and varies by a
few per cent.  An application decides which allocation holds which operand only once, when
it builds its data, so that is where the choice belongs:

    coeff, d, info = Jets.stream_pair(R)      # two vectors of the block space R, UNINITIALISED, ordered (read side, write side)
    ... fill coeff with the operator's coefficients, use d (and the solver's u) as the range-side vector ...

The order is measured: a tall diagonal operator over the blocks of one vector is applied into the other and back (forward + adjoint), both
ways, three calls each (about 0.3 s at 2 x 64 GiB), and the direction with the faster pair is kept.  Small spaces (under 4 GiB) and spaces that are not block spaces of equal
blocks are returned in allocation order, unprobed.  There is no counterpart in the reference (host memory has no such asymmetry).
"""
from __future__ import annotations

import builtins
import os

import numpy as np

from . import arrays as _arr
from . import device as _dev
from .spaces import JetBSpace

__all__ = ["stream_pair", "probe_stream_direction"]

PROBE_FROM_BYTES = 4 << 30


def _pair_ms(src, dst, calls: int):
    """(forward ms, adjoint ms) per call of the tall diagonal operator over the blocks of `src` with `dst` as its range vector: the forward
    reads src and writes dst (one block row per workgroup, all rows concurrent), the adjoint reads both.  Overwrites dst; src is only read."""
    from . import jetblock as _blk
    from .jets import mul_, close, adjoint

    A = _blk.blockop([[_blk.JopDiagonal(c)] for c in src.arrays])
    keep = {k: _dev.tune_get(k) for k in ("adj_wg", "adj_unroll", "adj_depth")}
    try:
        # The probe runs OTHER kernel instantiations than the ones an operator of this size ends up on (forward: candidate 6 of the measured
        # walks, 512 x 8 packs, where 7 = 256 x 1 usually wins; adjoint: 256 x 2 x 2 instead of the shape rule's pick): placement moves them all
        # alike, and a profile of the application (rocprofv3 --stats averages per kernel name) is not diluted by the probe's launches.
        try:
            _blk.op_tune_set(A, "fwd_walk", int(os.environ.get("JETS_PROBE_WALK", "6")))
        except Exception:
            pass                                                   # (an operator too small for the measured walks: whatever it runs is the same both ways)
        if os.environ.get("JETS_PROBE_ADJ", "alt") == "alt":
            _dev.tune(adj_wg=256, adj_unroll=2, adj_depth=2)
        m, mt = _arr.Array(src.spaces[0]), _arr.Array(src.spaces[0])
        At = adjoint(A)
        mul_(dst, A, m)
        mul_(mt, At, dst)
        _dev.synchronize()
        e = [_dev.Event() for _ in builtins.range(3)]
        e[0].record()
        for _ in builtins.range(calls):
            mul_(dst, A, m)
        e[1].record()
        for _ in builtins.range(calls):
            mul_(mt, At, dst)
        e[2].record()
        _dev.synchronize()
        return e[0].elapsed_ms(e[1]) / calls, e[1].elapsed_ms(e[2]) / calls
    finally:
        _dev.tune(**keep)
        close(A)


def _forward_ms(src, dst, calls: int) -> float:
    return _pair_ms(src, dst, calls)[0]


def probe_stream_direction(x, y, calls: int = 3):
    """(ms reading x and writing y, ms reading y and writing x) for two BlockArrays of the same block space of equal blocks.  DESTROYS the
    contents of both (meant for vectors that have just been allocated)."""
    return _forward_ms(x, y, calls), _forward_ms(y, x, calls)


def stream_pair(R, calls: int = 3, candidates: int = 2, keep_all: bool = False):
    """Two uninitialised vectors of the space R, ordered (the one to read from, the one to write to), and what was measured:
    {"probed": bool, "fwd_ms_kept", "adj_ms_kept", "pair_ms_kept": float, "pair_ms_other": [...]}.  `candidates` > 2 allocates that many vectors, measures every ordered pair
    and gives the others back (a third candidate makes it likely that one of them covers the device's fast-write region; it needs the
    memory for the moment of the probe).  info["kept"] = the indices, in allocation order, of the two vectors returned."""
    nbytes = R.length() * np.dtype(R.eltype()).itemsize
    uniform = isinstance(R, JetBSpace) and len({s.size() for s in R.spaces}) == 1 and len(R.spaces) >= 2
    if nbytes < PROBE_FROM_BYTES or not uniform:
        return _arr.Array(R, undef=True), _arr.Array(R, undef=True), {"probed": False}
    free = _dev.device_info()["free_mem"]
    candidates = builtins.max(2, builtins.min(int(candidates), int(0.9 * free // nbytes)))
    xs = [_arr.Array(R, undef=True), _arr.Array(R, undef=True)]
    from ._ffi import JetsHipError

    while len(xs) < candidates:                                    # a further candidate is a bonus: without the memory for it, go on with what there is
        try:
            xs.append(_arr.Array(R, undef=True))
        except JetsHipError as e:
            if e.status != 3:                                      # JH_ERR_NOMEM
                raise
            break
    candidates = len(xs)
    timed = {}
    for i in builtins.range(candidates):
        for j in builtins.range(candidates):
            if i != j:
                timed[(i, j)] = _pair_ms(xs[i], xs[j], calls)
    (i, j), best = builtins.min(timed.items(), key=lambda kv: kv[1][0] + kv[1][1])     # forward + adjoint: the pair a solver iteration pays for
    if not keep_all:
        for k, v in enumerate(xs):
            if k not in (i, j):
                v.close()
    others = sorted(round(f + a, 3) for k, (f, a) in timed.items() if k != (i, j))
    info = {"probed": True, "candidates": candidates, "fwd_ms_kept": best[0], "adj_ms_kept": best[1], "pair_ms_kept": best[0] + best[1],
            "pair_ms_other": others, "kept": [i, j],                       # allocation order would have been [0, 1]
            "pair_ms_allocation_order": timed[(0, 1)][0] + timed[(0, 1)][1]}
    if keep_all:                                                           # the caller closes what it does not keep (bench.py: its allocation-order leg)
        info["all"] = xs
    return xs[i], xs[j], info
