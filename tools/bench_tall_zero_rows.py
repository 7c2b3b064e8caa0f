import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import jets_jl_amd as J
J.init(0)
nrow, edge, keep = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
spc = J.JetSpace("float32", edge, edge, edge)
n = edge ** 3
rows = [[J.JopDiagonal(J.rand(spc, seed=1, stream=i))] if i % keep == 0 else [J.JopZeroBlock(spc, spc)] for i in range(nrow)]
A = J.blockop(rows)
nz = sum(1 for i in range(nrow) if i % keep == 0)
m = J.rand(J.domain(A), seed=2, stream=0)
d = J.zeros(J.range(A)); mt = J.zeros(J.domain(A))
def timed(fn, reps=8, warm=3):
    for _ in range(warm): fn()
    best = 1e9
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best
tf = timed(lambda: J.mul_(d, A, m)); ta = timed(lambda: J.mul_(mt, A.H, d))
b = n * 4
print(f"{nrow} x 1 of {edge}^3, {nz} diagonal rows, the rest zero blocks: forward {tf:.3f} ms {(2 * nz + 1) * b / tf / 1e6:.0f} GB/s | adjoint {ta:.3f} ms {(2 * nz + 1) * b / ta / 1e6:.0f} GB/s")
