#!/usr/bin/env python3
"""Soak of the chained one-pass step's hand-off protocol: STEPS steps on two copies of the same state, one through the plain
ordered walk, one through the chained row chunks; w (whole) and slices of u must stay bit-identical all the way, and the sticky
error word must never be raised (an expired poll fails the call that reads ||u||^2).

    python tools/soak_step_chain.py [NROW] [STEPS] [--ranged] [--beside]

--ranged : the chained copy runs as the pipelined multi-GPU step does -- four element ranges (jh_blockop_bidiag_step_range),
           ||u||^2 deferred on the device, one jh_normsq_read per step.
--beside : while the steps run, a SECOND context (its own stream) of the same device streams a bandwidth-heavy kernel over
           1.5 GiB without pause -- the stand-in for RCCL's reduce kernels, which share the device with the chained step in the
           pipelined distributed iteration (VERDICT r2, weak point 5)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jets_jl_amd as J
from jets_jl_amd import jetblock as _blk
from jets_jl_amd._ffi import check, lib

flags = [a for a in sys.argv[1:] if a.startswith("--")]
pos = [a for a in sys.argv[1:] if not a.startswith("--")]
nrow = int(pos[0]) if len(pos) > 0 else 1024
steps = int(pos[1]) if len(pos) > 1 else 1000
edge = int(pos[2]) if len(pos) > 2 else 256
ranged, beside = "--ranged" in flags, "--beside" in flags
J.init(0)
home = J.context_current()[0]
n = edge ** 3
spc = J.JetSpace("float32", edge, edge, edge)
coeff = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
nat = _blk._tall_native(A)
u1 = J.rand(J.range(A), seed=3, stream=0)
u2 = J.rand(J.range(A), seed=3, stream=0)
v = J.rand(spc, seed=2, stream=0)
w1, w2 = J.zeros(spc), J.zeros(spc)
o1, o2 = C.c_double(0), C.c_double(0)

noise = None
if beside:
    other = J.context_create(0)
    big = J.JetSpace("float32", 128 * 1024 * 1024)                  # 512 MiB per vector
    with J.using_context(other):
        nx, ny, nz = J.rand(big, seed=7, stream=0), J.rand(big, seed=8, stream=0), J.zeros(big)
        nev = [J.Event() for _ in range(8)]
    J.context_use(home)
    noise = {"launches": 0, "k": 0}

    def make_noise(count):
        """`count` triads z = 0.5 x + 0.25 y (1.5 GiB of traffic each) on the other context's stream; never more than 8 batches ahead."""
        e = nev[noise["k"] % 8]
        if noise["k"] >= 8:
            e.elapsed_ms(e)                                          # hipEventSynchronize on the batch enqueued 8 batches ago
        for _ in range(count):
            J.lincomb_(nz, [0.5, 0.25], [nx, ny])
        e.record()
        noise["k"] += 1
        noise["launches"] += count
        J.context_use(home)


def bounds(total, parts=4):
    step = -(-total // parts)
    step = -(-step // 16384) * 16384
    lo = 0
    while lo < total:
        cnt = min(step, total - lo)
        yield lo, cnt
        lo += cnt


t0 = time.time()
chunks, handoffs = 0, 0
for k in range(steps):
    alpha, beta = (1.0, -0.5) if k % 3 else (0.75, 0.25)          # u stays bounded
    if noise is not None:
        make_noise(6)
    J.op_tune_set(A, "step_mode", 0)
    check(lib.jh_blockop_bidiag_step(nat.handle, u1.handle, v.handle, w1.handle, alpha, beta, C.byref(o1)))
    J.op_tune_set(A, "step_mode", 2)
    if ranged:
        check(lib.jh_normsq_reset())
        for lo, cnt in bounds(n):
            check(lib.jh_blockop_bidiag_step_range(nat.handle, u2.handle, v.handle, w2.handle, alpha, beta, lo, cnt, None))
            assert J.tune_get("last_step_chain") > 0, "the chained walk did not run on a range"
        check(lib.jh_normsq_read(C.byref(o2)))                     # fails if any hand-off poll expired
    else:
        check(lib.jh_blockop_bidiag_step(nat.handle, u2.handle, v.handle, w2.handle, alpha, beta, C.byref(o2)))
    chunks = J.tune_get("last_step_chain")
    assert chunks > 0, "the chained walk did not run"
    tile = 4096 if J.tune_get('step_chunk') in (8, 16) or nrow <= 32 else 1024     # scalars per tile: 1024 lanes x 4 (8-row chunks) / 256 lanes x 4 (round 5's 32-row chunks)
    handoffs += chunks * (n // tile)
    assert abs(o1.value - o2.value) <= 1e-12 * o1.value, (k, o1.value, o2.value)
    if k % 100 == 99 or k == steps - 1:
        a, b = w1.to_numpy(), w2.to_numpy()
        assert a.tobytes() == b.tobytes(), f"step {k}: w differs"
        for off in (0, (nrow // 2) * n + 12345, nrow * n - 65536):
            assert u1._download(off, 65536).tobytes() == u2._download(off, 65536).tobytes(), f"step {k}: u differs at {off}"
        print(f"step {k + 1}: w and u slices bit-identical, ||u||^2 {o1.value:.6e}, {handoffs / 1e6:.1f} M hand-offs, {time.time() - t0:.0f} s", flush=True)
err = np.zeros(1, dtype=np.uint32)
extra = ""
if noise is not None:
    with J.using_context(other):
        J.synchronize()
    extra = f", beside {noise['launches']} concurrent 1.5 GiB triads on a second stream of the device"
print(f"soak ok: {steps} chained steps{' in 4 ranges' if ranged else ''} of {nrow} x {edge}^3 ({chunks} chunks x {n // tile} tiles per step, "
      f"{handoffs / 1e6:.1f} M hand-offs in all){extra}, no expired poll, bits of the plain walk, {time.time() - t0:.0f} s")
