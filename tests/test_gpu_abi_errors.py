"""GPU: the C ABI's argument validation -- every misuse returns a status code with a message (never a crash,
never a silent fallback).  Driven through raw ctypes, below the host mirror."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def abi(Jets):
    from jets_jl_amd._ffi import lib, BlockDesc

    return lib, BlockDesc


def _vec(lib, lens, dtype=0):
    h = C.c_void_p()
    arr = (C.c_int64 * len(lens))(*lens)
    assert lib.jh_bvec_create(len(lens), arr, dtype, C.byref(h)) == 0
    return h


def test_vector_argument_validation(abi):
    lib, _ = abi
    h = C.c_void_p()
    one = (C.c_int64 * 1)(4)
    assert lib.jh_bvec_create(0, one, 0, C.byref(h)) == 1                       # no blocks
    assert lib.jh_bvec_create(1, one, 9, C.byref(h)) == 1                       # unknown dtype
    assert lib.jh_bvec_create(1, (C.c_int64 * 1)(-3), 0, C.byref(h)) == 1       # negative length
    assert b"negative" in lib.jh_last_error()
    assert lib.jh_bvec_create(1, one, 0, None) == 1
    v = _vec(lib, [4, 6, 2])
    buf = (C.c_float * 8)()
    assert lib.jh_getblock_copy(v, 3, buf, 0) == 1 and b"out of range" in lib.jh_last_error()
    assert lib.jh_getblock_copy(v, -1, buf, 0) == 1
    assert lib.jh_setblock_fill(v, 7, 1.0, 0.0) == 1
    w = C.c_void_p()
    assert lib.jh_bvec_view(v, 2, 2, C.byref(w)) == 1                           # runs past the last block
    assert lib.jh_bvec_view(v, 1, 2, C.byref(w)) == 0
    n, length = C.c_int64(0), C.c_int64(0)
    assert lib.jh_bvec_info(w, C.byref(n), C.byref(length), None, None) == 0 and (n.value, length.value) == (2, 8)
    assert lib.jh_download(v, 10, 5, buf) == 1                                   # range outside the slab
    assert lib.jh_fill_uniform(v, 1, 1, -5) == 1
    assert lib.jh_bvec_destroy(w) == 0 and lib.jh_bvec_destroy(v) == 0 and lib.jh_bvec_destroy(None) == 0


def test_elementwise_and_reduction_validation(abi):
    lib, _ = abi
    x, y, z64, c = _vec(lib, [8]), _vec(lib, [8]), _vec(lib, [8], 1), _vec(lib, [8], 2)
    short = _vec(lib, [7])
    coef = (C.c_double * 4)(1, 0, 1, 0)
    hs = (C.c_void_p * 2)(x, y)
    assert lib.jh_lincomb(x, 0, coef, hs) == 1 and lib.jh_lincomb(x, 9, coef, hs) == 1
    assert lib.jh_lincomb(short, 2, coef, hs) == 1 and b"length mismatch" in lib.jh_last_error()
    assert lib.jh_lincomb(z64, 2, coef, hs) == 1                                 # dtype mismatch
    assert lib.jh_lincomb(x, 2, (C.c_double * 4)(1, 0.5, 1, 0), hs) == 1         # complex coefficient on a real vector
    assert lib.jh_hadamard(x, y, short, 0) == 1
    out = C.c_double(0)
    assert lib.jh_norm(x, float("nan"), C.byref(out)) == 1
    assert lib.jh_norm(x, 2.0, None) == 1
    assert lib.jh_dot(x, z64, C.byref(out), None) == 1
    mn, mx = C.c_double(0), C.c_double(0)
    assert lib.jh_extrema(c, C.byref(mn), C.byref(mx)) == 1 and b"complex" in lib.jh_last_error()
    assert lib.jh_abs(x, c) == 0 and lib.jh_abs(z64, c) == 1                     # |ComplexF32| is Float32
    for h in (x, y, z64, c, short):
        lib.jh_bvec_destroy(h)


def test_operator_validation(abi):
    lib, BlockDesc = abi
    coeff = _vec(lib, [8])
    p = C.c_void_p()
    lib.jh_bvec_info(coeff, None, None, None, C.byref(p))
    op = C.c_void_p()
    rows, cols = (C.c_int64 * 2)(8, 8), (C.c_int64 * 1)(8)

    def desc(kind, nr=8, nc=8, ptr=p.value):
        b = (BlockDesc * 2)()
        for k in range(2):
            b[k].kind, b[k].adjoint, b[k].coeff, b[k].nr, b[k].nc = kind, 0, ptr, nr, nc
        return b

    assert lib.jh_blockop_create(0, 1, desc(3), rows, cols, 0, C.byref(op)) == 1
    assert lib.jh_blockop_create(2, 1, desc(7), rows, cols, 0, C.byref(op)) == 1 and b"unknown block kind" in lib.jh_last_error()
    assert lib.jh_blockop_create(2, 1, desc(3, 8, 9), rows, cols, 0, C.byref(op)) == 1       # block does not fit its row/column
    assert lib.jh_blockop_create(2, 1, desc(3, ptr=None), rows, cols, 0, C.byref(op)) == 1   # DIAG without coefficients
    assert lib.jh_blockop_create(2, 1, desc(3), rows, cols, 0, C.byref(op)) == 0
    d, m, bad = _vec(lib, [8, 8]), _vec(lib, [8]), _vec(lib, [9])
    assert lib.jh_blockop_mul(op, d, bad) == 1 and b"domain vector" in lib.jh_last_error()
    assert lib.jh_blockop_mul(op, bad, m) == 1 and b"range vector" in lib.jh_last_error()
    assert lib.jh_blockop_mul(op, d, m) == 0 and lib.jh_blockop_mul_adj(op, m, d) == 0
    assert lib.jh_blockop_normal_mul(op, m, m) == 1                                           # y aliases m
    nrm = C.c_double(0)
    assert lib.jh_blockop_mul_axpby(op, d, m, 1.0, 0.0, C.byref(nrm)) == 0
    assert lib.jh_gemv(p, 4, 2, 0, d, m, 0) == 1 and b"needs vectors" in lib.jh_last_error()
    assert lib.jh_comm_allreduce_sum(m) == 5                                                  # JH_ERR_STATE: no communicator
    assert lib.jh_tune_set(b"no_such_knob", 1) == 1 and lib.jh_tune_set(b"fwd_wg", 300) == 1
    assert lib.jh_blockop_destroy(op) == 0 and lib.jh_blockop_destroy(None) == 0
    for h in (coeff, d, m, bad):
        lib.jh_bvec_destroy(h)


def test_stream_and_events(abi, Jets):
    lib, _ = abi
    s = C.c_void_p()
    assert lib.jh_get_stream(C.byref(s)) == 0 and s.value
    assert lib.jh_set_stream(None) == 0                                            # back to the library's own stream
    e0, e1 = Jets.Event().record(), None
    x = Jets.rand(Jets.JetSpace(np.float32, 1 << 20))
    Jets.fill_(x, 2.0)
    e1 = Jets.Event().record()
    assert e0.elapsed_ms(e1) >= 0.0
    assert lib.jh_event_record(None) == 1 and lib.jh_event_destroy(None) == 0
    assert lib.jh_init(0) == 0                                                     # idempotent for the same device
    if Jets.device_count() == 1:
        assert lib.jh_init(1) == 1 and b"out of range" in lib.jh_last_error()      # no such device
    else:                                                                          # a second device gets its own primary context
        assert lib.jh_init(1) == 0 and lib.jh_set_device(0) == 0


def test_create_destroy_cycles_do_not_leak_device_memory(Jets):
    """Handles are freed for real: 300 cycles of vectors, operators (fast path, general path, per-block loop with graphs),
    JIT broadcast programs and pinned buffers leave the free-memory figure where it was."""
    import gc

    def free_bytes():
        Jets.synchronize()
        return Jets.device_info()["free_mem"]

    def cycle(k):
        spc = Jets.JetSpace(np.float32, 64 * 1024)
        diags = [Jets.rand(spc, seed=1, stream=i) for i in range(4)]
        A = Jets.blockop([[Jets.JopDiagonal(g)] for g in diags])                       # tall fast path
        B = Jets.blockop([[Jets.JopDiagonal(diags[0]), Jets.JopIdentity(spc)], [Jets.JopZeroBlock(spc, spc), Jets.JopSquare(spc)]])
        Cd = Jets.blockop([[Jets.JopDense(Jets.rand(Jets.JetSpace(np.float32, 32, 32)))] for _ in range(3)])   # per-block loop
        m = Jets.rand(spc)
        d = A * m
        _ = A.H * d
        _ = Jets.mul(A.H @ A, m)
        mb = Jets.rand(Jets.domain(B))
        _ = Jets.jacobian_(B, mb) * mb
        x = Jets.rand(Jets.domain(Cd))
        for _i in range(3):
            _ = Cd * x                                                               # eager, captured, replayed
        Jets.broadcast_(m, "x0*x0 + s0", [m], [float(k % 3)])
        pin = Jets.pinned_empty(1 << 16, np.float32)
        m.to_numpy(out=pin)
        Jets.lsqr(A, d, maxiter=3)
        for op in (A, B, Cd):
            Jets.close(op)

    cycle(0)
    gc.collect()
    base = free_bytes()
    for k in range(300):
        cycle(k)
    gc.collect()
    after = free_bytes()
    assert abs(after - base) <= 64 << 20, f"free device memory moved by {(base - after) / 2**20:.1f} MiB over 300 cycles"
