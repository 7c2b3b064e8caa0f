import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import jets_jl_amd as J
J.init(0)
def timed(fn, reps=3):
    fn(); fn()
    best = 1e30
    for _ in range(reps):
        e0 = J.Event().record(); fn(); e1 = J.Event().record()
        best = min(best, e0.elapsed_ms(e1))
    return best
EXPR = os.environ.get("EXPR", "exp(x0)")
if os.environ.get("ITEM_FAST"):
    J.tune(bcast_item_fast=int(os.environ["ITEM_FAST"]))
if os.environ.get("BCAST_BAND"):
    J.tune(bcast_band=int(os.environ["BCAST_BAND"]))              # tiles per column band of the batched broadcast (1: items fastest, no bands; 0: the default, 32)
for nrow, n in ((256, 1 << 24), (64, 1 << 22), (1024, 1 << 18), (4096, 1 << 16), (16384, 1 << 14)):
    spc = J.JetSpace(np.float32, n)
    F = J.blockop([[J.JopElementwise(spc, EXPR, EXPR)] for _ in range(nrow)])
    m = J.rand(J.domain(F), seed=2, stream=0)
    d = J.zeros(J.range(F))
    tf = timed(lambda: J.mul_(d, F, m))
    tp = timed(lambda: J.jacobian_(F, m))
    Jm = J.jacobian_(F, m)
    tj = timed(lambda: J.mul_(d, Jm, m))
    b = nrow * n * 4
    print(f"{nrow:6d} x {n:8d} JopElementwise({EXPR}): F(m) {tf:8.3f} ms ({b / tf / 1e6:6.0f} GB/s written) | jacobian! {tp:8.3f} ms | J*dm {tj:8.3f} ms", flush=True)
    del F, Jm, m, d
