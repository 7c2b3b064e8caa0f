"""Device-backed vectors: DeviceArray (plain N-d array) and BlockArray (north_star's "JetBArray").

Mirrors /root/reference/src/Jets.jl:809-924 (BlockArray, norm/dot/extrema/fill!/convert,
broadcast, getblock/getblock!/setblock!, factories) and 1112-1118 (reshape to a block space) over
the C ABI of include/jetship.h.  A BlockArray is ONE contiguous HBM slab; block i starts at
element offset indices[i].start (the layout of JetBSpace.indices), so `reshape(flat, R)` and the
per-block views alias the slab with no copy, exactly like the reference's views.

All arithmetic runs in HIP kernels through libjetship.so; nothing here computes on the host.
"""
from __future__ import annotations

import builtins
import ctypes as C
import itertools
import math
from typing import Sequence

import numpy as np

from ._ffi import lib, check, scalar_flags
from . import device as _device
from .spaces import JetAbstractSpace, JetSpace, JetBSpace, JetSSpace, dtype_code

__all__ = [
    "DeviceArray", "BlockArray", "LinExpr", "zeros", "ones", "rand", "rand_", "randn", "Array", "from_numpy", "space", "nblocks",
    "indices", "getblock", "getblock_", "setblock_", "norm", "dot", "extrema", "fill_", "copyto_", "lincomb_", "norm_blocks", "dot_blocks",
    "hadamard_", "similar", "convert_array", "reshape", "vec", "length", "abs_", "pinned_empty", "host_register",
    "host_unregister", "download_into", "upload_from",
]

_rand_counter = itertools.count(1)
_DEFAULT_SEED = 0x4A455453  # "JETS"


def _i64arr(vals: Sequence[int]):
    return (C.c_int64 * len(vals))(*[int(v) for v in vals])


class _DevVec:
    """Common base: owns (or borrows) one jh_bvec handle."""

    _h = None
    _owner = None  # keeps the parent alive for views / wraps
    __array_ufunc__ = None  # numpy scalars defer to __rmul__ / __radd__ instead of iterating the vector

    @property
    def handle(self):
        if self._h is None:
            raise ValueError("device vector already closed")
        return self._h

    @property
    def dtype(self) -> np.dtype:
        return self._dtype

    def eltype(self):
        return self._dtype

    def length(self) -> int:
        return self._length

    def __len__(self) -> int:
        return self._length

    @property
    def ptr(self) -> int:
        p = C.c_void_p()
        check(lib.jh_bvec_info(self.handle, None, None, None, C.byref(p)))
        return p.value or 0

    def close(self) -> None:
        if self._h is not None:
            h, self._h = self._h, None
            lib.jh_bvec_destroy(h)
        self._owner = None

    def __del__(self):  # the Julia wrapper registers the same thing as a finalizer
        try:
            self.close()
        except Exception:
            pass

    # --- host transfer (convert(Array, x), src/Jets.jl:862-868) -----------------------------------
    def _download(self, offset: int = 0, count: int | None = None) -> np.ndarray:
        count = self._length - offset if count is None else count
        out = np.empty(count, dtype=self._dtype)
        if count:
            check(lib.jh_download(self.handle, offset, count, out.ctypes.data_as(C.c_void_p)))
        return out

    def _upload(self, host: np.ndarray, offset: int = 0) -> None:
        host = np.ascontiguousarray(host, dtype=self._dtype).ravel()
        if host.size:
            check(lib.jh_upload(self.handle, offset, host.size, host.ctypes.data_as(C.c_void_p)))

    # --- LinExpr sugar:  y.assign(a*u + b*v - w) is  y .= a*u .+ b*v .- w  ------------------------
    def __mul__(self, a):
        if isinstance(a, (int, float, complex, np.number)):
            return LinExpr([(a, self)])
        return NotImplemented

    __rmul__ = __mul__

    def __neg__(self):
        return LinExpr([(-1.0, self)])

    def __add__(self, other):
        return LinExpr([(1.0, self)]) + other

    def __radd__(self, other):
        return other + LinExpr([(1.0, self)]) if isinstance(other, LinExpr) else NotImplemented

    def __sub__(self, other):
        return LinExpr([(1.0, self)]) - other

    def assign(self, expr) -> "_DevVec":
        """`self .= expr` (src/Jets.jl:905-911)."""
        if isinstance(expr, LinExpr):
            lincomb_(self, [c for c, _ in expr.terms], [x for _, x in expr.terms])
        elif isinstance(expr, _DevVec):
            copyto_(self, expr)
        elif isinstance(expr, (int, float, complex, np.number)):
            fill_(self, expr)
        else:
            self._upload(np.asarray(expr).ravel(order="F"))
        return self


class LinExpr:
    """Lazy `c1*x1 .+ c2*x2 .+ ...` (the Broadcasted tree of src/Jets.jl:889-911, restricted to linear
    combinations); evaluated left to right in one fused kernel by `assign` / `materialize`."""

    __array_ufunc__ = None

    def __init__(self, terms):
        self.terms = list(terms)

    def __add__(self, other):
        if isinstance(other, _DevVec):
            other = LinExpr([(1.0, other)])
        if not isinstance(other, LinExpr):
            return NotImplemented
        return LinExpr(self.terms + other.terms)

    def __sub__(self, other):
        if isinstance(other, _DevVec):
            other = LinExpr([(1.0, other)])
        if not isinstance(other, LinExpr):
            return NotImplemented
        return LinExpr(self.terms + [(-c, x) for c, x in other.terms])

    def __mul__(self, a):
        if isinstance(a, (int, float, complex, np.number)):
            return LinExpr([(a * c, x) for c, x in self.terms])
        return NotImplemented

    __rmul__ = __mul__

    def __neg__(self):
        return LinExpr([(-c, x) for c, x in self.terms])

    def materialize(self):
        """similar(find_blockarray(bc)) then copyto! (src/Jets.jl:893-898)."""
        first = next((x for _, x in self.terms if isinstance(x, BlockArray)), self.terms[0][1])
        return similar(first).assign(self)


class DeviceArray(_DevVec):
    """A plain N-d array in HBM (column-major shape metadata), e.g. the domain vector of a one-column
    block operator (src/Jets.jl:927) or one block of a BlockArray (getblock, src/Jets.jl:914)."""

    def __init__(self, handle, shape, dtype, owner=None):
        self._h = handle
        self.shape = tuple(int(s) for s in shape)
        self._dtype = np.dtype(dtype)
        self._length = math.prod(self.shape)                     # (exact Python integers; np.prod cost 2 us per block view of a vector of 65 536 blocks)
        self._owner = owner

    @property
    def ndim(self) -> int:
        return len(self.shape)

    @property
    def size(self) -> int:
        return self._length

    @property
    def __cuda_array_interface__(self):
        item = self._dtype.itemsize
        strides, acc = [], item
        for s in self.shape:  # Fortran order
            strides.append(acc)
            acc *= s
        return {"shape": self.shape, "typestr": self._dtype.str, "data": (self.ptr, False), "version": 3,
                "strides": tuple(strides) if len(self.shape) > 1 else None}

    def to_numpy(self, out: np.ndarray | None = None) -> np.ndarray:
        """Host copy (column-major).  `out`: a flat or same-shape host array to fill instead -- pass a `pinned_empty`
        array to move at the PCIe rate."""
        if out is not None:
            return download_into(self, out).reshape(self.shape, order="F")
        return self._download().reshape(self.shape, order="F")

    def reshape(self, *shape) -> "DeviceArray":
        """reshape(x, dims): shares memory (src/Jets.jl:38)."""
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        if int(np.prod(shape, dtype=np.int64)) != self._length:
            raise ValueError("dimension mismatch in reshape")
        h = C.c_void_p()
        check(lib.jh_bvec_view(self.handle, 0, 1, C.byref(h)))
        return DeviceArray(h, shape, self._dtype, owner=self)

    def __repr__(self):
        return f"DeviceArray({self._dtype.name}, shape={self.shape})"


class BlockArray(_DevVec):
    """Jets.BlockArray (src/Jets.jl:809-812) backed by one HIP slab."""

    def __init__(self, handle, spaces: Sequence[JetAbstractSpace], dtype, owner=None, indices=None):
        self._h = handle
        self._dtype = np.dtype(dtype)
        if indices is not None:
            # the block ranges of the JetBSpace this vector lives in, shared (never mutated): a vector of 16 384 blocks no longer walks them again --
            # 18 ms per zeros(range(A)) of a composite's stage, every call, where the stage's kernels take 2.5 (tools/prof_host.py)
            self.spaces, self.indices = spaces, indices
            self._length = indices[-1].stop if indices else 0
        else:
            self.spaces = list(spaces)
            self.indices = []
            stop = 0
            for s in self.spaces:
                self.indices.append(builtins.range(stop, stop + s.length()))
                stop += s.length()
            self._length = stop
        self._owner = owner
        self._views = None

    @property
    def arrays(self) -> list:
        """x.arrays: the blocks, by reference (views of the slab)."""
        if self._views is None:
            views = []
            for i, s in enumerate(self.spaces):
                h = C.c_void_p()
                check(lib.jh_bvec_view(self.handle, i, 1, C.byref(h)))
                views.append(DeviceArray(h, s.size(), self._dtype, owner=self))
            self._views = views
        return self._views

    @property
    def shape(self):  # src/Jets.jl:818
        return (self._length,)

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self._length,), "typestr": self._dtype.str, "data": (self.ptr, False), "version": 3, "strides": None}

    def close(self) -> None:
        if self._views is not None:
            for v in self._views:
                v.close()
            self._views = None
        super().close()

    def to_numpy(self, out: np.ndarray | None = None) -> np.ndarray:
        """convert(Array, x) (src/Jets.jl:862-868) brought to the host (`out`: host array to fill, e.g. a pinned one)."""
        if out is not None:
            return download_into(self, out)
        return self._download()

    # linear indexing (src/Jets.jl:820-827): slow path by design, like the reference's findfirst
    def _locate(self, i: int):
        if i < 0:
            i += self._length
        for j, r in enumerate(self.indices):
            if i in r:
                return j, i - r.start
        raise IndexError(i)

    def __getitem__(self, i):
        if isinstance(i, slice):
            start, stop, step = i.indices(self._length)
            if step != 1:
                raise IndexError("only unit-stride slices")
            return self._download(start, builtins.max(0, stop - start))
        self._locate(i)
        return self._download(int(i) % self._length if i < 0 else int(i), 1)[0]

    def __setitem__(self, i, v):
        if isinstance(i, slice):
            start, stop, step = i.indices(self._length)
            if step != 1:
                raise IndexError("only unit-stride slices")
            vals = np.broadcast_to(np.asarray(v, dtype=self._dtype), (builtins.max(0, stop - start),))
            self._upload(vals, start)
            return
        self._locate(i)
        self._upload(np.asarray([v], dtype=self._dtype), int(i) % self._length if i < 0 else int(i))

    def __repr__(self):
        return f"BlockArray({self._dtype.name}, {len(self.spaces)} blocks, length {self._length})"


# ------------------------------------------------------------------------------ factories ----------
ROLE_OUTPUT, ROLE_DATA = 1, 2      # what a big vector is for (knob alloc_role): an operator's output / data written once and read from then on
_ROLE_FROM_BYTES = 4 << 30


def _new_handle(block_lens: Sequence[int], T, undef: bool = False, role: int = 0, lens_c=None) -> C.c_void_p:
    _device.init()
    h = C.c_void_p()
    create = lib.jh_bvec_create_uninit if undef else lib.jh_bvec_create
    lens = lens_c if lens_c is not None else _i64arr(block_lens)
    # vectors of 4 GiB and more: the library's slab cache may hold several slabs of this size, and which of them is fast to WRITE is a
    # property of the slab it has measured (include/jetship.h, knob alloc_role) -- say what this one is for
    hint = role != 0 and builtins.sum(block_lens) * np.dtype(T).itemsize >= _ROLE_FROM_BYTES
    if hint:
        check(lib.jh_tune_set(b"alloc_role", role))
    try:
        st = create(len(block_lens), lens, dtype_code(T), C.byref(h))
        if st == 3:                              # JH_ERR_NOMEM: vectors caught in reference cycles still hold device memory -- collect them and ask again
            import gc

            gc.collect()
            st = create(len(block_lens), lens, dtype_code(T), C.byref(h))
    finally:
        if hint:
            lib.jh_tune_set(b"alloc_role", 0)
    check(st)
    return h


def Array(R: JetAbstractSpace, undef: bool = False, role: int = 0):
    """Array(R) / zeros(R): device storage for the space (src/Jets.jl:105-108, 922-924).  Zero-filled unless `undef=True`
    (Julia's Array{T}(undef, ...): for an output the next call overwrites entirely; the fill of 64 GiB is 12 ms).  `role`: what a
    vector of 4 GiB or more is for (ROLE_OUTPUT / ROLE_DATA), a hint for the library's choice among cached slabs."""
    if isinstance(R, JetSSpace):  # src/Jets.jl:514-516
        from .symmetric import zeros_sym

        return zeros_sym(R)
    if isinstance(R, JetBSpace):
        return BlockArray(_new_handle(R.block_lengths(), R.eltype(), undef, role, lens_c=R.block_lengths_c()), R.spaces, R.eltype(), indices=R.indices)
    return DeviceArray(_new_handle([R.length()], R.eltype(), undef, role), R.size(), R.eltype())



def zeros(R: JetAbstractSpace):
    return Array(R)


def ones(R: JetAbstractSpace):
    if isinstance(R, JetSSpace):
        from .symmetric import ones_sym

        return ones_sym(R)
    return fill_(Array(R, undef=True), 1.0)           # (every element is written: no zero fill first)


def rand(R: JetAbstractSpace, seed: int | None = None, stream: int | None = None, index_base: int = 0):
    """rand(R): U[0,1) from the counter-based generator.  With seed/stream given the
    values are a pure function of (seed, stream, element index) and reproducible on the CPU oracle."""
    if isinstance(R, JetSSpace):
        from .symmetric import rand_sym

        return rand_sym(R, seed=seed, stream=stream, index_base=index_base)
    x = Array(R, undef=True, role=ROLE_DATA)               # the generator writes every element (once: data to read from then on)
    if seed is None:
        seed, stream = _DEFAULT_SEED, next(_rand_counter)
    check(lib.jh_fill_uniform(x.handle, int(seed), int(stream or 0), int(index_base)))
    return x


def rand_(x: "_DevVec", seed: int | None = None, stream: int | None = None, index_base: int = 0):
    """rand!(x): the same generator into existing storage (the values of rand(space(x), seed, stream, index_base))."""
    if seed is None:
        seed, stream = _DEFAULT_SEED, next(_rand_counter)
    check(lib.jh_fill_uniform(x.handle, int(seed), int(stream or 0), int(index_base)))
    return x


def randn(R: JetAbstractSpace, seed: int | None = None, stream: int | None = None, index_base: int = 0):
    """randn(R): standard normal values generated on the device (Box-Muller over the counter generator)."""
    if isinstance(R, JetSSpace):
        from .symmetric import randn_sym

        return randn_sym(R, seed=seed, stream=stream, index_base=index_base)
    x = Array(R, undef=True)
    if seed is None:
        seed, stream = _DEFAULT_SEED, next(_rand_counter)
    check(lib.jh_fill_normal(x.handle, int(seed), int(stream or 0), int(index_base)))
    return x


def abs_(x: "_DevVec"):
    """abs.(x): a real vector with the block structure of x (test/runtests.jl:545-547)."""
    if not isinstance(x, _DevVec):
        from .symmetric import SymmetricArray, abs_sym

        if isinstance(x, SymmetricArray):
            return abs_sym(x)
    real_t = np.float32 if x.dtype in (np.dtype(np.float32), np.dtype(np.complex64)) else np.float64
    out = similar(x, real_t)
    check(lib.jh_abs(out.handle, x.handle))
    return out


def from_numpy(a: np.ndarray, R: JetAbstractSpace | None = None):
    """Upload a host array; shape metadata is column-major (Julia order)."""
    a = np.asarray(a)
    if R is None:
        R = JetSpace(a.dtype, *a.shape)
    x = Array(R, undef=a.size == R.length())               # no zero fill when the upload covers the whole vector
    x._upload(a.ravel(order="F") if not isinstance(R, JetBSpace) else a.ravel())
    return x


# ------------------------------------------------------------------------------ pinned host memory --
class _PinnedOwner:
    """Frees a jh_host_alloc buffer when the last numpy view of it dies."""

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes
        self.buf = (C.c_char * nbytes).from_address(ptr)

    def __del__(self):
        try:
            lib.jh_host_free(C.c_void_p(self.ptr))
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float32) -> np.ndarray:
    """A page-locked (DMA-able) host array, column-major: transfers to / from it run at the PCIe rate instead of through
    the runtime's staging copy (jh_host_alloc)."""
    _device.init()
    dt = np.dtype(dtype)
    shape = (int(shape),) if np.isscalar(shape) else tuple(int(v) for v in shape)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
    if nbytes == 0:
        return np.empty(shape, dtype=dt, order="F")
    p = C.c_void_p()
    check(lib.jh_host_alloc(nbytes, C.byref(p)))
    owner = _PinnedOwner(p.value, nbytes)
    flat = np.frombuffer(owner.buf, dtype=dt)          # keeps `owner.buf` (and through the closure below, owner) alive
    flat = flat.view(_PinnedArray)
    flat._pinned_owner = owner
    return flat.reshape(shape, order="F")


class _PinnedArray(np.ndarray):
    """ndarray that carries its pinned-buffer owner through views and reshapes."""

    def __array_finalize__(self, obj):
        if obj is not None:
            self._pinned_owner = getattr(obj, "_pinned_owner", None)


def host_register(a: np.ndarray) -> np.ndarray:
    """Pin an existing contiguous host array in place (jh_host_register); undo with host_unregister."""
    _device.init()
    if not (a.flags.c_contiguous or a.flags.f_contiguous):
        raise ValueError("host_register needs a contiguous array")
    check(lib.jh_host_register(C.c_void_p(a.ctypes.data), a.nbytes))
    return a


def host_unregister(a: np.ndarray) -> None:
    check(lib.jh_host_unregister(C.c_void_p(a.ctypes.data)))


def download_into(x: "_DevVec", out: np.ndarray) -> np.ndarray:
    """Copy a device vector into an existing host array (flat, column-major order)."""
    if out.dtype != x.dtype or out.size != x.length() or not (out.flags.c_contiguous or out.flags.f_contiguous):
        raise ValueError("download_into: need a contiguous host array of the vector's dtype and length")
    if out.size:
        check(lib.jh_download(x.handle, 0, out.size, C.c_void_p(out.ctypes.data)))
    return out


def upload_from(x: "_DevVec", host: np.ndarray) -> "_DevVec":
    """Copy a contiguous host array (column-major order) into an existing device vector."""
    if host.dtype != x.dtype or host.size != x.length() or not (host.flags.c_contiguous or host.flags.f_contiguous):
        raise ValueError("upload_from: need a contiguous host array of the vector's dtype and length")
    if host.size:
        check(lib.jh_upload(x.handle, 0, host.size, C.c_void_p(host.ctypes.data)))
    return x


# ------------------------------------------------------------------------------ generic functions --
def space(x):
    """space(x) (src/Jets.jl:126, 452, 814)."""
    if not isinstance(x, _DevVec) and hasattr(x, "map") and hasattr(x, "A"):
        from .symmetric import space_sym

        return space_sym(x)
    if isinstance(x, BlockArray):
        return JetBSpace([JetSpace(x.dtype, *s.size()) for s in x.spaces])
    if isinstance(x, DeviceArray):
        return JetSpace(x.dtype, *x.shape)
    a = np.asarray(x)
    return JetSpace(a.dtype, *a.shape)


def length(x) -> int:
    return x.length() if hasattr(x, "length") else len(x)


def nblocks(x) -> int:
    """nblocks (src/Jets.jl:806-807, 860)."""
    if isinstance(x, (BlockArray, JetBSpace)):
        return len(x.spaces)
    return 1


def indices(x, i: int):
    """indices(R, i) / indices(x, i) (src/Jets.jl:780, 858): 0-based half-open range."""
    return x.indices[i]


def getblock(x, iblock: int):
    """getblock(x, i): by reference (src/Jets.jl:914); a plain array is its own block (918)."""
    if isinstance(x, BlockArray):
        return x.arrays[iblock]
    return x


def getblock_(x, iblock: int, out):
    """getblock!(x, i, out) (src/Jets.jl:915, 919): copies block i into `out` (device or numpy array)."""
    src = getblock(x, iblock)
    if isinstance(out, _DevVec):
        if out.length() != src.length():
            raise ValueError("DimensionMismatch in getblock!")
        check(lib.jh_getblock_copy(src.handle, 0, C.c_void_p(out.ptr), 1))
        return out
    if out.size != src.length():
        raise ValueError("DimensionMismatch in getblock!")
    if out.dtype == src.dtype and (out.flags.f_contiguous or out.ndim == 1 and out.flags.c_contiguous):
        return download_into(src, out)                       # straight into the caller's array: no temporary, no first-touch faults
    out[...] = src.to_numpy().reshape(out.shape, order="F")
    return out


def setblock_(x, iblock: int, value):
    """setblock!(x, i, v) (src/Jets.jl:916, 920): v is a scalar, a device array or a host array."""
    dst = getblock(x, iblock)
    if isinstance(value, (int, float, complex, np.number)):
        v = complex(value)
        if isinstance(x, BlockArray):
            check(lib.jh_setblock_fill(x.handle, iblock, v.real, v.imag))
        else:
            check(lib.jh_fill(x.handle, v.real, v.imag))
    elif isinstance(value, _DevVec):
        if value.length() != dst.length():
            raise ValueError("DimensionMismatch in setblock!")
        check(lib.jh_setblock_copy(dst.handle, 0, C.c_void_p(value.ptr), 1))
    else:
        a = np.asarray(value, dtype=dst.dtype)
        if a.size != dst.length():
            raise ValueError("DimensionMismatch in setblock!")
        dst._upload(a.ravel(order="F"))
    return dst


def norm(x: _DevVec, p: float = 2) -> float:
    """norm(x, p) (src/Jets.jl:834-848); returned in real(eltype) precision like the reference."""
    if not isinstance(x, _DevVec) and hasattr(x, "map") and hasattr(x, "A"):
        from .symmetric import norm_sym

        return norm_sym(x, p)
    out = C.c_double(0)
    check(lib.jh_norm(x.handle, float(p), C.byref(out)))
    real_t = np.float32 if x.dtype in (np.dtype(np.float32), np.dtype(np.complex64)) else np.float64
    return real_t(out.value)


def dot(x: _DevVec, y: _DevVec):
    """dot(x, y) (src/Jets.jl:850-856): conjugates x; result cast to eltype T."""
    re, im = C.c_double(0), C.c_double(0)
    check(lib.jh_dot(x.handle, y.handle, C.byref(re), C.byref(im)))
    if x.dtype.kind == "c":
        return x.dtype.type(complex(re.value, im.value))
    return x.dtype.type(re.value)


def norm_blocks(x: _DevVec, p: float = 2) -> np.ndarray:
    """[norm(getblock(x, i), p) for i in 1:nblocks(x)] in ONE pass over the slab (jh_norm_blocks; src/Jets.jl:836-846 forms these block norms before it
    combines them): per-shot residual norms of a block range without a launch and a host round trip per block.  Real(eltype) precision like `norm`."""
    nb = nblocks(x)
    out = (C.c_double * nb)()
    check(lib.jh_norm_blocks(x.handle, float(p), out))
    real_t = np.float32 if x.dtype in (np.dtype(np.float32), np.dtype(np.complex64)) else np.float64
    return np.ctypeslib.as_array(out).astype(real_t)                          # (no Python list of nblocks floats on the way: 50 ns per block)


def dot_blocks(x: _DevVec, y: _DevVec) -> np.ndarray:
    """[dot(getblock(x, i), getblock(y, i)) for i in 1:nblocks(x)] in one pass (jh_dot_blocks; 850-856 adds exactly these); conjugates x."""
    nb = nblocks(x)
    re, im = (C.c_double * nb)(), (C.c_double * nb)()
    check(lib.jh_dot_blocks(x.handle, y.handle, re, im))
    if x.dtype.kind == "c":
        return (np.ctypeslib.as_array(re) + 1j * np.ctypeslib.as_array(im)).astype(x.dtype)
    return np.ctypeslib.as_array(re).astype(x.dtype)


def extrema(x: _DevVec):
    """extrema(x) (src/Jets.jl:870-878)."""
    mn, mx = C.c_double(0), C.c_double(0)
    check(lib.jh_extrema(x.handle, C.byref(mn), C.byref(mx)))
    return x.dtype.type(mn.value), x.dtype.type(mx.value)


def fill_(x: _DevVec, a):
    """fill!(x, a) (src/Jets.jl:880-885)."""
    v = complex(a)
    check(lib.jh_fill(x.handle, v.real, v.imag))
    return x


def copyto_(dst: _DevVec, src: _DevVec):
    """dst .= src for equal-length device vectors."""
    check(lib.jh_copy(dst.handle, src.handle))
    return dst


def lincomb_(dst: _DevVec, coefs: Sequence, xs: Sequence[_DevVec]):
    """dst .= c1*x1 .+ c2*x2 .+ ... in one fused pass, evaluated left to right in eltype T."""
    k = len(xs)
    cf = (C.c_double * (2 * k))()
    fl = (C.c_int32 * k)()
    for j, c in enumerate(coefs):
        fl[j] = scalar_flags(c)                      # the coefficient's TYPE: Julia's arithmetic follows it (include/jetship.h JH_SCALAR_*)
        cc = complex(c)
        cf[2 * j], cf[2 * j + 1] = cc.real, cc.imag
    hs = (C.c_void_p * k)(*[x.handle for x in xs])
    check(lib.jh_lincomb_typed(dst.handle, k, cf, fl, hs))
    return dst


def hadamard_(dst: _DevVec, x: _DevVec, y: _DevVec, conj_x: bool = False, twice_x: bool = False):
    """dst .= x .* y   (conj_x: conj.(x) .* y; twice_x: (2 .* x) .* y)."""
    check(lib.jh_hadamard(dst.handle, x.handle, y.handle, (1 if conj_x else 0) | (2 if twice_x else 0)))
    return dst


def similar(x: _DevVec, T=None, n: int | None = None):
    """similar(x[, T[, n]]) (src/Jets.jl:829-832): a BlockArray when n == length(x), else a plain array."""
    if not isinstance(x, _DevVec) and hasattr(x, "map") and hasattr(x, "A"):
        from .symmetric import similar_sym

        return similar_sym(x, T)
    T = x.dtype if T is None else np.dtype(T)
    if isinstance(n, tuple):
        n = n[0]
    if _device.several_contexts():                         # similar(x) lives where x lives
        _device.context_use(_device.context_of(x))
    if isinstance(x, BlockArray):
        if n is None or n == x.length():
            return Array(JetBSpace([JetSpace(T, *s.size()) for s in x.spaces]))
        return Array(JetSpace(T, n))
    if n is None:
        return Array(JetSpace(T, *x.shape))
    return Array(JetSpace(T, n))


def convert_array(x: _DevVec) -> DeviceArray:
    """convert(Array, x::BlockArray) (src/Jets.jl:862-868) as a flat device array: the slab is
    already in that layout, so this is one device-to-device copy."""
    out = Array(JetSpace(x.dtype, x.length()), undef=True)
    check(lib.jh_copy(out.handle, x.handle))
    return out


def reshape(x, R):
    """reshape(x, R) (src/Jets.jl:38, 1112-1118): shares memory with x."""
    if isinstance(R, JetBSpace):
        if isinstance(x, BlockArray):  # :1115-1118
            if x.length() != R.length():
                raise ValueError("dimension mismatch, unable to reshape block array")
            return x
        if x.length() != R.length():
            raise ValueError("dimension mismatch, unable to reshape array into block space")
        h = C.c_void_p()
        check(lib.jh_bvec_wrap(C.c_void_p(x.ptr), R.nblocks(), _i64arr(R.block_lengths()), dtype_code(x.dtype), C.byref(h)))
        return BlockArray(h, R.spaces, x.dtype, owner=x, indices=R.indices)  # :1112 views of x
    if isinstance(x, BlockArray):
        if x.length() != R.length():
            raise ValueError("dimension mismatch in reshape")
        h = C.c_void_p()
        check(lib.jh_bvec_wrap(C.c_void_p(x.ptr), 1, _i64arr([x.length()]), dtype_code(x.dtype), C.byref(h)))
        return DeviceArray(h, R.size(), x.dtype, owner=x)
    return x.reshape(R.size())


def vec(x):
    """vec(x): 1-D view sharing memory; vec(R): the space backed by one-dimensional arrays (src/Jets.jl:1127-1128)."""
    if isinstance(x, JetBSpace):
        return x
    if isinstance(x, JetSpace):
        return JetSpace(x.eltype(), x.length())
    if isinstance(x, BlockArray):
        return x
    return x.reshape((x.length(),))
