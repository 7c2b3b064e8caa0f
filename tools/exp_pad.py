#!/usr/bin/env python3
"""Does the forward's process-dependent rate come from how the coefficient slab and the range slab alias in DRAM?
Same coefficient slab, range vectors allocated with a leading pad block of varying size (the operator writes the blocks
after the pad), forward timed with the sequential and the row-concurrent walk."""
import ctypes as C
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J
from jets_jl_amd import arrays as _a
from jets_jl_amd._ffi import lib, check

J.init(0)
J.tune(autotune=0)
edge, N = 256, 1024
n = edge ** 3
blk = J.JetSpace(np.float32, edge, edge, edge)
coeff = J.rand(J.JetBSpace([blk] * N), seed=1, stream=0)
A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
m = J.rand(blk, seed=2, stream=0)
mt = J.zeros(blk)
MiB = 1 << 20


def timed(fn, reps=4):
    fn(); fn()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


print("pad before the range vector | forward sequential | forward all rows | adjoint      (ms; 1024 x 256^3 Float32)")
for pad_mib in (0, 2, 4, 6, 8, 16, 32, 34, 64, 66, 128, 512, 1024, 2050, 0):
    spaces = ([J.JetSpace(np.float32, pad_mib * MiB // 4)] if pad_mib else []) + [blk] * N
    big = J.zeros(J.JetBSpace(spaces))
    if pad_mib:
        h = C.c_void_p()
        check(lib.jh_bvec_view(big.handle, 1, N, C.byref(h)))
        d = _a.BlockArray(h, [blk] * N, big.dtype, owner=big)
    else:
        d = big
    J.tune(fwd_wg=1024, fwd_unroll=8, fwd_group=16, fwd_order=0)
    t_seq = timed(lambda: J.mul_(d, A, m))
    J.tune(fwd_wg=512, fwd_unroll=1, fwd_group=2, fwd_order=1)
    t_all = timed(lambda: J.mul_(d, A, m))
    J.tune(fwd_wg=0, fwd_unroll=0, fwd_group=0, fwd_order=-1)
    t_adj = timed(lambda: J.mul_(mt, A.H, d))
    print(f"{pad_mib:6d} MiB   ptr 0x{d.ptr:x}   {t_seq:8.3f}   {t_all:8.3f}   {t_adj:8.3f}", flush=True)
    del d, big
    gc.collect()
