import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import jets_jl_amd as J
from jets_jl_amd._ffi import lib, check
from jets_jl_amd import jetblock as _blk
J.init(0)
for nblocks, edge in ((128, 256), (256, 256), (1024, 256), (1024, 128), (64, 128), (250, 250)):
    n = edge ** 3
    blk = J.JetSpace("float32", edge, edge, edge)
    coeff = J.rand(J.JetBSpace([blk] * nblocks), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    nat = _blk._tall_native(A)
    u = J.rand(J.range(A), seed=3, stream=0); v = J.rand(J.domain(A), seed=2, stream=0); w = J.zeros(J.domain(A))
    out = C.c_double(0)
    def one_pass():
        check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, 1.0, -0.5, C.byref(out)))
    def timed(reps=7):
        best = 1e9
        for _ in range(reps):
            e0 = J.Event().record(); one_pass(); e1 = J.Event().record()
            best = min(best, e0.elapsed_ms(e1))
        return best
    b3 = (3 * nblocks * n + 2 * n) * 4
    res = {}
    for forced in (0, 1):
        nat.tune_set("step_remap", forced)
        one_pass(); one_pass()
        res[forced] = timed()
    nat.tune_set("step_remap", -1)
    for _ in range(8): one_pass()
    chosen = nat.tune_get("step_remap")
    t = timed()
    print(f"{nblocks} x {edge}^3 one-pass step: tiles by id {res[0]:8.3f} ms {b3/res[0]/1e6:7.1f} GB/s | XCD-contiguous {res[1]:8.3f} ms {b3/res[1]/1e6:7.1f} GB/s | measured choice {chosen}: {t:8.3f} ms {b3/t/1e6:7.1f} GB/s", flush=True)
    del u, v, w, coeff; J.close(A)
