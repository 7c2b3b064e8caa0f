#!/usr/bin/env python3
"""Host <-> device copy rates of the boundary (jh_upload / jh_download): pageable, page-locked (jh_host_alloc) and
registered-in-place (jh_host_register) host arrays, one 256^3 Float32 block (64 MiB) and a 1 GiB vector.

    python tools/bench_pcie.py > profiles/bench_pcie_r01.txt
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J

J.init(0)
for n in (256 ** 3, 1 << 28):
    x = J.rand(J.JetSpace(np.float32, n), seed=1, stream=0)
    nbytes = 4 * n
    bufs = {"pageable": np.empty(n, np.float32), "page-locked (jh_host_alloc)": J.pinned_empty(n, np.float32)}
    reg = np.empty(n, np.float32)
    reg[:] = 0
    t0 = time.perf_counter()
    J.host_register(reg)
    t_reg = time.perf_counter() - t0
    bufs["registered in place (jh_host_register, %.1f ms to pin)" % (1e3 * t_reg)] = reg
    for name, h in bufs.items():
        h[:] = 1.0
        res = []
        for fn in (lambda: J.download_into(x, h), lambda: J.upload_from(x, h)):
            fn()
            best = 1e9
            for _ in range(5):
                J.synchronize()
                t0 = time.perf_counter()
                fn()
                J.synchronize()
                best = min(best, time.perf_counter() - t0)
            res.append(nbytes / best / 1e9)
        print(f"{nbytes / 2**20:7.0f} MiB  {name:58s} device->host {res[0]:6.1f} GB/s   host->device {res[1]:6.1f} GB/s", flush=True)
    J.host_unregister(reg)
