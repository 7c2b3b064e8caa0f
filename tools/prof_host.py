"""Host-side cost of the everyday calls on a tall operator / block vector of MANY blocks (round 6): what Python adds per call once the kernels take microseconds.

    python tools/prof_host.py [NBLOCKS]"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, jets_jl_amd as J
J.init(0)
N, n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 1024
spc = J.JetSpace(np.float32, n)
R = J.JetBSpace([spc] * N)
t0 = time.perf_counter(); coeff = J.rand(R, seed=1, stream=0); t1 = time.perf_counter()
print(f"rand(R) with {N} blocks: {(t1 - t0) * 1e3:.1f} ms")
t0 = time.perf_counter(); arrs = coeff.arrays; t1 = time.perf_counter()
print(f"x.arrays (views): {(t1 - t0) * 1e3:.1f} ms")
t0 = time.perf_counter(); A = J.blockop([[J.JopDiagonal(c)] for c in arrs]); t1 = time.perf_counter()
print(f"blockop: {(t1 - t0) * 1e3:.1f} ms")
m = J.rand(spc, seed=2, stream=0); d = J.zeros(R); y = J.zeros(spc); w = J.rand(R, seed=3, stream=0)
W = J.JopDiagonal(w)
NA, NW = J.compose(A.H, A), J.compose(J.compose(A.H, W), A)
def wall(tag, fn, reps=5):
    fn(); J.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    J.synchronize()
    print(f"{tag:44s} {(time.perf_counter() - t0) / reps * 1e3:9.3f} ms per call")
wall("first A*m (native handle)", lambda: J.mul_(d, A, m), 1)
wall("mul_(d, A, m)", lambda: J.mul_(d, A, m))
wall("mul_(y, A', d)", lambda: J.mul_(y, A.H, d))
wall("d = A * m (allocating)", lambda: A * m)
wall("(A'A) m", lambda: J.mul_(y, NA, m))
wall("(A'WA) m", lambda: J.mul_(y, NW, m))
wall("norm(d)", lambda: J.norm(d))
wall("dot(d, w)", lambda: J.dot(d, w))
wall("lincomb_(d, [2, 3], [d, w])", lambda: J.lincomb_(d, [2.0, 3.0], [d, w]))
wall("broadcast_(d, 's0*x0 + x1')", lambda: J.broadcast_(d, "s0*x0 + x1", [d, w], [0.5]))
wall("zeros(R)", lambda: J.zeros(R))
wall("getblock(d, 7)", lambda: J.getblock(d, 7))
wall("norm_blocks(d)", lambda: J.norm_blocks(d))
wall("fill_(d, 1)", lambda: J.fill_(d, 1.0))
wall("copyto_(d, w)", lambda: J.copyto_(d, w))
wall("d2 = d .- w (allocating)", lambda: d - w)
