import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import jets_jl_amd as J
J.init(0)
def timed(fn, reps=5):
    fn(); fn()
    best = 1e30
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best
for nrow, n in ((16384, 16384), (4096, 65536), (1024, 262144)):
    spc = J.JetSpace(np.float32, n)
    for table in (False, True):
        if table:
            diags = [J.rand(spc, seed=1, stream=i) for i in range(nrow)]
        else:
            diags = J.rand(J.JetBSpace([spc] * nrow), seed=1, stream=0).arrays
        A = J.blockop([[J.JopDiagonal(g)] for g in diags])
        m = J.rand(J.domain(A), seed=2, stream=0); d = J.rand(J.range(A), seed=3, stream=0); mt = J.zeros(J.domain(A))
        tf = timed(lambda: J.mul_(d, A, m)); ta = timed(lambda: J.mul_(mt, A.H, d))
        ptrs = sorted(g.ptr for g in diags)
        gaps = np.diff(ptrs)
        print(f"{nrow} x {n} {'separate allocations' if table else 'one slab':22s}: fwd {tf:7.3f} ms adj {ta:7.3f} ms; pointer spacing min {gaps.min()} median {int(np.median(gaps))} max {gaps.max()}", flush=True)
        del A, diags, m, d, mt
