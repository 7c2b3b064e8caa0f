import ctypes as C, sys
sys.path.insert(0, "/root/repo")
import jets_jl_amd as J
from jets_jl_amd._ffi import lib, check
from jets_jl_amd import jetblock as _blk
J.init(0)
edge=256; n=edge**3
for nblocks in (1024, 256, 128):
    blk = J.JetSpace("float32", edge, edge, edge)
    coeff = J.rand(J.JetBSpace([blk] * nblocks), seed=1, stream=0)
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    nat = _blk._native_op(A.jet.s["_native"], A.jet.s["ops"], A.jet.rng.eltype())
    u = J.rand(J.range(A), seed=3, stream=0); v = J.rand(J.domain(A), seed=2, stream=0); w = J.zeros(J.domain(A))
    out = C.c_double(0)
    b2 = (2*nblocks*n+2*n)*4
    def timed(fn, reps=5):
        fn(); fn(); best=1e9
        for _ in range(reps):
            e0=J.Event().record(); fn(); e1=J.Event().record(); best=min(best,e0.elapsed_ms(e1))
        return best
    ms = timed(lambda: J.mul_(u, A, v))
    print(f"{nblocks} rows: plain forward {ms:.3f} ms {b2/ms/1e6:.0f} GB/s", flush=True)
    for sh in [dict(adj_wg=wg, adj_unroll=U, adj_depth=D) for wg in (256,512) for (U,D) in ((1,4),(1,8),(2,2),(4,1),(4,2))]:
        J.tune(**sh)
        ms1 = timed(lambda: check(lib.jh_blockop_bidiag_step(nat.handle, u.handle, v.handle, w.handle, 1.0, 0.0, C.byref(out))))
        print(f"{nblocks} rows: bidiag beta=0 (forward + free adjoint) {sh} {ms1:.3f} ms {b2/ms1/1e6:.0f} GB/s", flush=True)
    J.tune(adj_wg=0, adj_unroll=0, adj_depth=0)
    J.close(A); del A, coeff, u, v, w, nat
    import gc; gc.collect()
