"""GPU: the slab cache (include/jetship.h, jh_trim): device memory of destroyed vectors of 16 MiB or more is kept for the next vector of
exactly that size -- hipMalloc of a range-sized slab costs seconds on this machine (profiles/exp_r03_alloc_cost.txt), and the
reference's style allocates such temporaries per call (src/Jets.jl:399, 526-533)."""
from __future__ import annotations

import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GIB = 1 << 30


def _cached_mib(J):
    return J.tune_get("slab_cached_mib")


def test_a_destroyed_big_vector_is_reused_by_the_next_of_its_size_and_comes_back_zeroed(Jets):
    J = Jets
    gc.collect()                                                  # (vectors of earlier tests go to the cache now, not in the middle of this one)
    J.trim()
    spc = J.JetSpace(np.float32, GIB // 2)                        # 2 GiB
    x = J.ones(spc)
    ptr = x.ptr
    x.close()
    del x
    gc.collect()
    assert _cached_mib(J) == 2048
    free_with_cache = J.device_info()["free_mem"]
    y = J.zeros(J.JetSpace(np.float32, GIB // 2 - 4))             # another size: not served from the cache
    assert y.ptr != ptr and _cached_mib(J) == 2048
    z = J.zeros(spc)                                              # the same size: the cached slab, zero-filled again
    assert z.ptr == ptr and _cached_mib(J) == 0
    assert float(J.norm(z)) == 0.0
    z.close(); y.close()
    del y, z
    gc.collect()
    assert _cached_mib(J) == 2048 + 2047                          # both went into the cache (MiB, rounded down)
    J.trim()
    assert _cached_mib(J) == 0
    assert abs(J.device_info()["free_mem"] - free_with_cache) < 64 << 20     # cached memory was counted as free all along


def test_small_vectors_and_a_switched_off_cache_go_straight_back_to_the_driver(Jets):
    J = Jets
    J.trim()
    small = J.zeros(J.JetSpace(np.float32, 1 << 20))              # 4 MiB: below the floor
    small.close()
    assert _cached_mib(J) == 0
    try:
        J.tune(slab_cache=0)
        big = J.zeros(J.JetSpace(np.float32, GIB // 2))
        big.close()
        del big
        gc.collect()
        assert _cached_mib(J) == 0
    finally:
        J.tune(slab_cache=1)
    assert J.tune_get("slab_cache") == 1


def test_an_allocation_that_needs_the_cached_memory_gets_it(Jets):
    """Two 100 GiB slabs wait in the cache, then a request the rest of the device cannot satisfy: slabs go back to the driver -- the one
    that covers the shortfall with the fewest bytes first -- until the request fits."""
    J = Jets
    info = J.device_info()
    if info["free_mem"] < 270 * GIB:
        pytest.skip(f"needs an empty 288 GiB device, {info['free_mem'] / GIB:.0f} GiB free")
    J.trim()
    n100 = 25 * GIB                                               # Float32 elements of a 100 GiB slab
    a, b = J.Array(J.JetSpace(np.float32, n100)), J.Array(J.JetSpace(np.float32, n100 + 4))
    a.close(); b.close()
    del a, b
    gc.collect()
    held = _cached_mib(J)
    assert held in (200 * 1024, 200 * 1024 + 1)                   # both (the cap leaves the last 32 GiB of the device alone)
    small = J.Array(J.JetSpace(np.float32, 20 * GIB))             # 80 GiB: fits beside the cache -- nothing is given back for it
    assert _cached_mib(J) == held
    small.close()
    del small
    gc.collect()
    assert _cached_mib(J) in (180 * 1024, 180 * 1024 + 1)         # 280 GiB would pass the cap: one 100 GiB slab went back to the driver
    c = J.Array(J.JetSpace(np.float32, 55 * GIB))                 # 220 GiB: more than what is free beside the cache -- both remaining slabs go
    assert _cached_mib(J) == 0
    J.fill_(c, 1.0)
    assert float(J.norm(c, np.inf)) == 1.0
    c.close()
    del c
    gc.collect()
    J.trim()


def test_a_times_m_skips_the_zero_fill_only_where_every_row_is_overwritten(Jets, oracle):
    """`A*m` = mul!(zeros(range(A)), A, m) (src/Jets.jl:399).  A one-column operator of native children without zero blocks overwrites every
    row (1026), so its output comes from jh_bvec_create_uninit -- here out of a cached slab full of garbage; an operator with a zero block
    (its row stays as found, 1022) or with several columns (accumulates, 1024) still gets zeros."""
    from jets_jl_amd.jetblock import overwrites_its_whole_range
    from .helpers import assert_bits_equal

    J = Jets
    J.trim()
    dt, n, nrow = np.float32, 6 << 20, 3                                      # rows of 24 MiB: the 72 MiB range vector is cached when destroyed
    spc = J.JetSpace(dt, n)
    coeffs = [J.rand(spc, seed=1, stream=i) for i in range(nrow)]
    hc = [oracle.rng_u01(dt, 1, i, 0, n) for i in range(nrow)]
    hm = oracle.rng_u01(dt, 2, 0, 0, n)
    m = J.rand(spc, seed=2, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeffs])
    Z = J.blockop([[J.JopDiagonal(coeffs[0])], [J.JopZeroBlock(spc, spc)], [J.JopDiagonal(coeffs[2])]])
    G = J.blockop([[J.JopDiagonal(coeffs[0]), J.JopDiagonal(coeffs[1])], [J.JopDiagonal(coeffs[2]), J.JopZeroBlock(spc, spc)], [J.JopIdentity(spc), J.JopIdentity(spc)]])
    assert overwrites_its_whole_range(A) and not overwrites_its_whole_range(Z) and not overwrites_its_whole_range(G) and not overwrites_its_whole_range(A.H)
    for _ in range(2):
        junk = J.fill_(J.zeros(J.range(A)), float("nan"))                      # a range-sized slab of NaNs goes into the cache ...
        ptr = junk.ptr
        junk.close()
        d = A * m                                                              # ... and comes back as the output of A*m, not zeroed
        assert d.ptr == ptr
        ref = oracle.block_df([[oracle.Block("diag", n, coeff=c)] for c in hc], [np.zeros(n, dtype=dt) for _ in range(nrow)], [hm])
        assert_bits_equal(d.to_numpy(), np.concatenate(ref), "A*m into an uninitialised slab")
        d.close()
        junk = J.fill_(J.zeros(J.range(Z)), float("nan"))
        junk.close()
        dz = Z * m                                                             # the zero block's row must read 0, not the slab's NaNs
        zops = [[oracle.Block("diag", n, coeff=hc[0])], [oracle.Block("zero", n, n)], [oracle.Block("diag", n, coeff=hc[2])]]
        assert_bits_equal(dz.to_numpy(), np.concatenate(oracle.block_df(zops, [np.zeros(n, dtype=dt) for _ in range(nrow)], [hm])), "zero row of Z*m")
        dz.close()
    x2 = J.rand(J.domain(G), seed=3, stream=0)
    junk = J.fill_(J.zeros(J.range(G)), float("nan"))
    junk.close()
    dg = (G * x2).to_numpy()
    assert np.isfinite(dg).all()
    J.trim()


def test_churn_of_cached_slabs_never_hands_out_memory_that_is_still_in_use(Jets):
    """Random create / combine / destroy of vectors of a few sizes (all above the cache's floor), destroying operands right after the
    kernels that read them were enqueued: every survivor must hold what a host model says.  A slab handed out while a kernel still
    reads or writes it would show up as a wrong value (the destroy waits for the vector's stream before the slab goes to the cache)."""
    J = Jets
    J.trim()
    rng = np.random.default_rng(11)
    sizes = [(4 << 20) + 8 * k for k in range(3)]                   # Float32 elements: 16 MiB + a little, three distinct sizes
    live = []                                                        # (device vector, host value of every element)
    hits = 0
    for it in range(400):
        action = rng.integers(0, 4)
        if action <= 1 or len(live) < 3:
            n = sizes[rng.integers(0, len(sizes))]
            before = J.tune_get("slab_cached_mib")
            x = J.zeros(J.JetSpace(np.float32, n))
            hits += J.tune_get("slab_cached_mib") < before
            val = float(rng.integers(1, 9))
            J.fill_(x, val)
            live.append((x, val))
        elif action == 2:
            i, j = rng.integers(0, len(live), 2)
            (a, va), (b, vb) = live[i], live[j]
            if a.length() == b.length() and i != j:
                out = J.zeros(J.space(a))
                J.lincomb_(out, [2.0, -1.0], [a, b])                 # enqueued ...
                live.append((out, 2.0 * va - vb))
                k = max(i, j)
                live[k][0].close()                                   # ... and one operand destroyed at once: its slab goes to the cache
                live.pop(k)
        else:
            k = rng.integers(0, len(live))
            live[k][0].close()
            live.pop(k)
        if len(live) > 12:
            live[0][0].close()
            live.pop(0)
    assert hits > 50                                                  # the cache was really in play
    for x, val in live:
        mn, mx = J.extrema(x)
        assert float(mn) == float(mx) == val
        x.close()
    J.trim()


def test_big_slabs_are_probed_when_they_enter_the_cache_and_an_allocation_with_a_role_takes_the_right_one(Jets, oracle):
    """Round 4: which of several cached slabs `A*m` (src/Jets.jl:399) writes into.  A slab of 4 GiB or more is probed once when it enters the
    cache (its fill time: a property of the slab on this chip, profiles/exp_r04_write_probe.txt); an allocation that says what it is for takes
    the fastest-to-write (an operator's output) or the slowest-to-write (data read from then on) cached slab of its size; one that does
    not, the most recently freed.  `A*m` says 'output', rand(R) says 'data' -- and the results do not depend on where they live."""
    from jets_jl_amd import arrays

    from .helpers import assert_bits_equal

    J = Jets
    gc.collect()
    J.trim()
    blk = J.JetSpace(np.float32, 1 << 20)                          # 4 MiB blocks, 1024 of them: a 4 GiB range
    nrow = 1024
    R = J.JetBSpace([blk] * nrow)
    xs = [J.Array(R, undef=True) for _ in range(3)]
    ptrs = [x.ptr for x in xs]
    for x in xs:
        x.close()
    del xs, x
    gc.collect()
    assert _cached_mib(J) == 3 * 4096 and J.tune_get("slab_probed") == 3
    a = J.Array(R, undef=True)                                      # no role: the most recently freed slab, no choice made
    assert a.ptr == ptrs[2] and J.tune_get("last_alloc_choice") == -1
    a.close()
    del a
    gc.collect()
    assert J.tune_get("slab_probed") == 3                           # (probed once per slab, not once per visit to the cache)
    out = J.Array(R, undef=True, role=arrays.ROLE_OUTPUT)
    assert J.tune_get("last_alloc_choice") == 300 and out.ptr in ptrs        # 3 probed candidates, the fastest fill
    data = J.rand(R, seed=1, stream=0)                              # rand says 'data': of the two left, the slowest fill
    assert J.tune_get("last_alloc_choice") == 201 and data.ptr in ptrs and data.ptr != out.ptr
    assert J.tune_get("alloc_role") == 0, "the hint is for one allocation"
    A = J.blockop([[J.JopDiagonal(c)] for c in data.arrays])
    m = J.rand(J.domain(A), seed=2, stream=0)
    out.close()
    del out
    gc.collect()
    d = A * m                                                       # 'output': the fastest of the two cached again
    assert J.tune_get("last_alloc_choice") == 200 and d.ptr in ptrs and d.ptr != data.ptr
    n = blk.length()
    ha = oracle.rng_u01(np.float32, 1, 0, 5 * n, n)
    hm = oracle.rng_u01(np.float32, 2, 0, 0, n)
    assert_bits_equal(J.getblock(d, 5).to_numpy().ravel(), ha * hm, "row 5 of A*m")
    J.close(A)
    for v in (d, data, m):
        v.close()
    del d, data, m, A
    gc.collect()
    J.trim()
    assert J.tune_get("slab_probed") == 0


def test_a_slab_is_not_handed_out_while_another_stream_still_writes_it(Jets):
    """Round-3 advisor finding: the destroy waited for the owning context's CURRENT stream only, and then put the slab into the cache --
    work of any other stream (a communicator's exchange stream under an early error return, a stream the application installed with
    jh_set_stream and has replaced since, torch / RCCL streams on a wrapped vector) was no longer covered, where hipFree used to wait
    for the whole device.  Here a second context's stream is installed in the first, a long chain of fills of a vector is enqueued on
    it, the own stream is restored, the vector destroyed and a new one of its size created at once (zeros): it must BE zeros."""
    J = Jets
    gc.collect()
    J.trim()
    home = J.context_current()[0]
    other = J.context_create(0)
    try:
        J.context_use(other)
        foreign = J.stream_handle()                                  # the second context's stream
        J.context_use(home)
        spc = J.JetSpace(np.float32, 1 << 29)                        # 2 GiB: a fill takes ~0.3 ms
        for trial in range(3):
            x = J.Array(spc, undef=True)
            ptr = x.ptr
            J.set_stream(foreign)
            for k in range(60):                                      # ~20 ms of writes into x on the foreign stream
                J.fill_(x, float(k + 1))
            J.set_stream(None)                                       # back on the context's own (idle) stream
            x.close()                                                # -> the slab goes to the cache
            del x
            y = J.zeros(spc)                                         # the same slab, zero-filled on the own stream
            assert y.ptr == ptr
            J.synchronize()
            J.context_use(other)
            J.synchronize()
            J.context_use(home)
            mn, mx = J.extrema(y)
            assert float(mn) == 0.0 and float(mx) == 0.0, f"trial {trial}: the old writer was still running when the slab was handed out (extrema {mn}, {mx})"
            y.close()
            del y
    finally:
        J.context_use(home)
        J.set_stream(None)
        gc.collect()
        J.context_destroy(other)
        J.trim()
