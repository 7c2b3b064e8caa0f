#!/usr/bin/env python3
"""The batched dense kernels under rocprofv3: a tall operator of N dense K x K Float32 children (1 GiB), 20 forwards, 20 adjoints."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jets_jl_amd as J
k = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nrow = (1 << 28) // (k * k)
J.init(0)
mats = [J.rand(J.JetSpace(np.float32, k, k), seed=1, stream=i) for i in range(nrow)]
A = J.blockop([[J.JopDense(M)] for M in mats])
m = J.rand(J.domain(A), seed=2, stream=0); d = J.rand(J.range(A), seed=3, stream=0); mt = J.zeros(J.domain(A))
for _ in range(20):
    J.mul_(d, A, m)
for _ in range(20):
    J.mul_(mt, A.H, d)
J.synchronize()
print(f"{nrow} x 1 of {k}^2 Float32 dense children; algorithmic bytes per call: {nrow * k * k * 4}")
