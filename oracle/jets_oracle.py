"""numpy-facing wrapper of the C oracle (oracle/libjets_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the product package jets.jl_amd.  See oracle/jets_oracle.h for what the oracle
restates (file:line of /root/reference/src/Jets.jl) and how its parity is pinned.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libjets_oracle.so")

DT = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.complex64): 2, np.dtype(np.complex128): 3}
KIND = {"zero": 0, "identity": 1, "scale": 2, "diag": 3, "dense": 4, "square": 5}


class _Block(C.Structure):
    _fields_ = [("kind", C.c_int32), ("adjoint", C.c_int32), ("coeff", C.c_void_p), ("sre", C.c_double), ("sim", C.c_double),
                ("nr", C.c_int64), ("nc", C.c_int64), ("sflags", C.c_int32), ("reserved", C.c_int32)]


SCALAR_COMPLEX, SCALAR_WIDE = 1, 2


def scalar_flags(a) -> int:
    """What Julia knows from a scalar's TYPE (jets_oracle.h JO_SCALAR_*), read off the Python object the way the host mirror does: a Python or
    numpy complex is a Complex scalar; numpy's float64 / complex128 are Float64-based (promoted arithmetic against 32-bit elements); plain
    Python numbers and 32-bit numpy scalars are taken in the vectors' element type (Julia's `T(a)`)."""
    f = SCALAR_COMPLEX if isinstance(a, (complex, np.complexfloating)) else 0
    return f | (SCALAR_WIDE if isinstance(a, (np.float64, np.complex128)) else 0)


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("jets_oracle.c", "jets_oracle_body.inc", "jets_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def _load():
    build()
    lib = C.CDLL(_SO)
    lib.jo_barr_norm.restype = C.c_double
    lib.jo_rng_key.restype = C.c_uint64
    lib.jo_rng_key.argtypes = [C.c_uint64, C.c_uint64]
    lib.jo_rng_u01.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_int64, C.c_int64, C.c_void_p]
    lib.jo_barr_locate.restype = C.c_int
    return lib


_lib = _load()


def _ptrs(arrays):
    return (C.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])


def _lens(arrays):
    return (C.c_int64 * len(arrays))(*[a.size for a in arrays])


def _check_blocks(arrays, dtype=None):
    out = []
    for a in arrays:
        if not (isinstance(a, np.ndarray) and (a.flags.f_contiguous or a.flags.c_contiguous)):
            raise ValueError("oracle blocks must be contiguous numpy arrays")
        if dtype is not None and a.dtype != dtype:
            raise TypeError("block dtype mismatch")
        out.append(a)
    return out


# ------------------------------------------------------------------ index arithmetic ---------------
def bspace_indices(lens):
    """1-based inclusive (start, stop) per block: src/Jets.jl:739-750."""
    n = len(lens)
    L = (C.c_int64 * n)(*lens)
    s, e = (C.c_int64 * n)(), (C.c_int64 * n)()
    _lib.jo_bspace_indices(C.c_int64(n), L, s, e)
    return list(s), list(e)


def barr_locate(lens, i1):
    """1-based linear index -> (block, local), both 1-based: src/Jets.jl:820-823."""
    s, e = bspace_indices(lens)
    n = len(lens)
    ib, il = C.c_int64(0), C.c_int64(0)
    rc = _lib.jo_barr_locate(C.c_int64(n), (C.c_int64 * n)(*s), (C.c_int64 * n)(*e), C.c_int64(i1), C.byref(ib), C.byref(il))
    if rc != 0:
        raise IndexError(i1)
    return ib.value, il.value


# ------------------------------------------------------------------ random -------------------------
def rng_u01(dtype, seed, stream, index0, count) -> np.ndarray:
    dt = np.dtype(dtype)
    out = np.empty(count, dtype=dt)
    _lib.jo_rng_u01(DT[dt], seed, stream, index0, count, out.ctypes.data)
    return out


# ------------------------------------------------------------------ BlockArray ops -----------------
def barr_norm(arrays, p=2.0) -> float:
    arrays = _check_blocks(arrays)
    return _lib.jo_barr_norm(C.c_int(DT[arrays[0].dtype]), C.c_int64(len(arrays)), _ptrs(arrays), _lens(arrays), C.c_double(p))


def barr_dot(x, y):
    x, y = _check_blocks(x), _check_blocks(y, x[0].dtype)
    re, im = C.c_double(0), C.c_double(0)
    _lib.jo_barr_dot(C.c_int(DT[x[0].dtype]), C.c_int64(len(x)), _ptrs(x), _ptrs(y), _lens(x), C.byref(re), C.byref(im))
    return complex(re.value, im.value) if x[0].dtype.kind == "c" else re.value


def barr_extrema(arrays):
    arrays = _check_blocks(arrays)
    mn, mx = C.c_double(0), C.c_double(0)
    _lib.jo_barr_extrema(C.c_int(DT[arrays[0].dtype]), C.c_int64(len(arrays)), _ptrs(arrays), _lens(arrays), C.byref(mn), C.byref(mx))
    return mn.value, mx.value


def barr_fill(arrays, a):
    arrays = _check_blocks(arrays)
    a = complex(a)
    _lib.jo_barr_fill(C.c_int(DT[arrays[0].dtype]), C.c_int64(len(arrays)), _ptrs(arrays), _lens(arrays), C.c_double(a.real), C.c_double(a.imag))
    return arrays


def barr_convert(arrays) -> np.ndarray:
    arrays = _check_blocks(arrays)
    flat = np.empty(sum(a.size for a in arrays), dtype=arrays[0].dtype)
    _lib.jo_barr_convert(C.c_int(DT[arrays[0].dtype]), C.c_int64(len(arrays)), _ptrs(arrays), _lens(arrays), C.c_void_p(flat.ctypes.data))
    return flat


def barr_lincomb(dst, coefs, srcs):
    dst = _check_blocks(dst)
    k = len(srcs)
    cf = (C.c_double * (2 * k))()
    fl = (C.c_int32 * k)()
    for j, c in enumerate(coefs):
        fl[j] = scalar_flags(c)
        c = complex(c)
        cf[2 * j], cf[2 * j + 1] = c.real, c.imag
    ptrs = [_ptrs(_check_blocks(s, dst[0].dtype)) for s in srcs]
    pp = (C.POINTER(C.c_void_p) * k)(*[C.cast(p, C.POINTER(C.c_void_p)) for p in ptrs])
    _lib.jo_barr_lincomb(C.c_int(DT[dst[0].dtype]), C.c_int64(len(dst)), _ptrs(dst), _lens(dst), C.c_int(k), cf, fl, pp)
    return dst


# ------------------------------------------------------------------ block operators ----------------
class Block:
    """One child operator of a block matrix (device-native kinds only).  kind "square" is the nonlinear
    d .= m.^2 (test/runtests.jl:19-24); its `coeff` is the linearisation point mo of that block."""

    def __init__(self, kind, nr, nc=None, coeff=None, scale=0.0, adjoint=False):
        self.kind, self.nr, self.nc = kind, int(nr), int(nr if nc is None else nc)
        self.coeff = None if coeff is None else np.asfortranarray(coeff)
        self.scale, self.adjoint, self.scale_flags = complex(scale), bool(adjoint), scalar_flags(scale)

    @property
    def rng_len(self):
        return self.nc if self.adjoint else self.nr

    @property
    def dom_len(self):
        return self.nr if self.adjoint else self.nc


def _ops_array(ops):
    """ops: list of rows of Block -> column-major jo_block array (like a Julia Matrix)."""
    nrow, ncol = len(ops), len(ops[0])
    arr = (_Block * (nrow * ncol))()
    for i in range(nrow):
        for j in range(ncol):
            b, o = arr[i + j * nrow], ops[i][j]
            b.kind, b.adjoint = KIND[o.kind], 1 if o.adjoint else 0
            b.coeff = o.coeff.ctypes.data if o.coeff is not None else None
            b.sre, b.sim, b.nr, b.nc, b.sflags = o.scale.real, o.scale.imag, o.nr, o.nc, o.scale_flags
    return arr, nrow, ncol


def block_df(ops, d_blocks, m_blocks):
    """JetBlock_df! (src/Jets.jl:1010-1032): mutates d_blocks in place."""
    arr, nrow, ncol = _ops_array(ops)
    d_blocks = _check_blocks(d_blocks)
    m_blocks = _check_blocks(m_blocks, d_blocks[0].dtype)
    _lib.jo_block_df(C.c_int(DT[d_blocks[0].dtype]), C.c_int64(nrow), C.c_int64(ncol), arr, _ptrs(d_blocks), _ptrs(m_blocks))
    return d_blocks


def block_f(ops, d_blocks, m_blocks):
    """JetBlock_f! (src/Jets.jl:988-1008): mutates d_blocks in place."""
    arr, nrow, ncol = _ops_array(ops)
    d_blocks = _check_blocks(d_blocks)
    m_blocks = _check_blocks(m_blocks, d_blocks[0].dtype)
    _lib.jo_block_f(C.c_int(DT[d_blocks[0].dtype]), C.c_int64(nrow), C.c_int64(ncol), arr, _ptrs(d_blocks), _ptrs(m_blocks))
    return d_blocks


def block_df_adj(ops, m_blocks, d_blocks):
    """JetBlock_df'! (src/Jets.jl:1034-1057): mutates m_blocks in place."""
    arr, nrow, ncol = _ops_array(ops)
    m_blocks = _check_blocks(m_blocks)
    d_blocks = _check_blocks(d_blocks, m_blocks[0].dtype)
    _lib.jo_block_df_adj(C.c_int(DT[m_blocks[0].dtype]), C.c_int64(nrow), C.c_int64(ncol), arr, _ptrs(m_blocks), _ptrs(d_blocks))
    return m_blocks


def normal_df(ops, y_blocks, m_blocks):
    """JetComposite_df! over (A', A) (src/Jets.jl:530-534): mutates y_blocks."""
    arr, nrow, ncol = _ops_array(ops)
    y_blocks = _check_blocks(y_blocks)
    m_blocks = _check_blocks(m_blocks, y_blocks[0].dtype)
    _lib.jo_normal_df(C.c_int(DT[y_blocks[0].dtype]), C.c_int64(nrow), C.c_int64(ncol), arr, _ptrs(y_blocks), _ptrs(m_blocks))
    return y_blocks


def child_mul(block: Block, d, m):
    arr, _, _ = _ops_array([[block]])
    _lib.jo_child_mul(C.c_int(DT[d.dtype]), arr, C.c_void_p(d.ctypes.data), C.c_void_p(m.ctypes.data))
    return d


def child_mul_adj(block: Block, m, d):
    arr, _, _ = _ops_array([[block]])
    _lib.jo_child_mul_adj(C.c_int(DT[m.dtype]), arr, C.c_void_p(m.ctypes.data), C.c_void_p(d.ctypes.data))
    return m


def tall_diag_pair_omp_f32(a_blocks, m, d_blocks, mt):
    """All-cores forward + adjoint of a tall Float32 diagonal operator (separately labelled CPU baseline only)."""
    nrow, n = len(a_blocks), m.size
    pa = (C.c_void_p * nrow)(*[x.ctypes.data for x in a_blocks])
    pd = (C.c_void_p * nrow)(*[x.ctypes.data for x in d_blocks])
    _lib.jo_tall_diag_fwd_omp_f32.restype = C.c_int
    _lib.jo_tall_diag_adj_omp_f32.restype = C.c_int
    nt = _lib.jo_tall_diag_fwd_omp_f32(C.c_int64(nrow), C.c_int64(n), pa, C.c_void_p(m.ctypes.data), pd)
    _lib.jo_tall_diag_adj_omp_f32(C.c_int64(nrow), C.c_int64(n), pa, C.c_void_p(mt.ctypes.data), pd)
    return nt


def fill_u01_omp_f32(rows, seed, row0=0):
    """Fill freshly allocated (untouched) Float32 rows with the counter generator, in parallel, with the thread partition of
    tall_diag_pair_omp_f32 -- the first touch puts every thread's chunks on its own NUMA node."""
    nrow, n = len(rows), rows[0].size
    pr = (C.c_void_p * nrow)(*[x.ctypes.data for x in rows])
    _lib.jo_fill_u01_omp_f32.restype = C.c_int
    return _lib.jo_fill_u01_omp_f32(C.c_int64(nrow), C.c_int64(n), C.c_uint64(seed), C.c_int64(row0), pr)


def dot_product_test(ops, m_blocks, d_blocks, mmask=None, dmask=None):
    """dot_product_test (src/Jets.jl:1211-1226) on a block operator of native kinds."""
    dt = m_blocks[0].dtype
    mm = [m * (mk if mmask is not None else 1) for m, mk in zip(m_blocks, mmask or m_blocks)]
    dd = [d * (dk if dmask is not None else 1) for d, dk in zip(d_blocks, dmask or d_blocks)]
    mm = [np.ascontiguousarray(x, dtype=dt) for x in mm]
    dd = [np.ascontiguousarray(x, dtype=dt) for x in dd]
    ds = [np.zeros(o.rng_len, dtype=dt) for o in [row[0] for row in ops]]          # op * (mmask .* m): zeros(range)  (:399)
    block_df(ops, ds, mm)
    ms = [np.zeros(o.dom_len, dtype=dt) for o in ops[0]]                           # op' * (dmask .* d)
    block_df_adj(ops, ms, dd)
    lhs, rhs = barr_dot(mm, ms), barr_dot(ds, dd)
    if isinstance(lhs, complex) and isinstance(rhs, complex):
        return lhs, rhs
    return np.real(lhs), np.real(rhs)
