#!/usr/bin/env python3
"""Streaming rate by position in HBM: read (norm), write (fill) and copy over 8 GiB windows of four 64 GiB slabs allocated
back to back (256 GiB of the 288 GB).  Do all regions of the memory stream equally fast?"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J
from jets_jl_amd import arrays as _a
from jets_jl_amd._ffi import lib, check

J.init(0)
GiB = 1 << 30
blk = J.JetSpace(np.float32, 2 * GiB // 4)            # 2 GiB blocks
slabs = [J.zeros(J.JetBSpace([blk] * 32)) for _ in range(4)]
for s in slabs:
    J.fill_(s, 1.0)


def window(s, first, count):
    h = C.c_void_p()
    check(lib.jh_bvec_view(s.handle, first, count, C.byref(h)))
    return _a.BlockArray(h, [blk] * count, s.dtype, owner=s)


def timed(fn, reps=4):
    fn()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best


print("slab  window[GiB]   read(norm) GB/s   write(fill) GB/s   copy within window GB/s   device ptr")
for si, s in enumerate(slabs):
    for w0 in range(0, 32, 4):                        # 8 GiB windows
        win = window(s, w0, 4)
        half_a, half_b = window(s, w0, 2), window(s, w0 + 2, 2)
        nb = 8 * GiB
        tr = timed(lambda: J.norm(win))
        tw = timed(lambda: J.fill_(win, 2.0))
        tc = timed(lambda: J.copyto_(half_a, half_b))
        print(f"{si:3d}   [{2 * w0:3d}, {2 * w0 + 8:3d})      {nb / tr / 1e6:8.1f}        {nb / tw / 1e6:8.1f}          {nb / tc / 1e6:8.1f}              0x{win.ptr:x}", flush=True)
