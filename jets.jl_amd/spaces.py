"""Jet spaces: shape + eltype metadata (host side, no device work).

Mirrors /root/reference/src/Jets.jl:5-129 (JetAbstractSpace, JetSpace) and 736-807 (JetBSpace).
Python conventions: block indices and element indices are 0-based; `indices(R, i)` is a Python
`range(start, stop)` (half-open), i.e. Julia's `a:b` (src/Jets.jl:742-748) maps to `range(a-1, b)`.
Arrays are column-major like Julia's (a block is contiguous; shape metadata only).
"""
from __future__ import annotations

import builtins
from typing import Sequence

import numpy as np

__all__ = ["JetAbstractSpace", "JetSpace", "JetBSpace", "JetSSpace", "DTYPE_CODES", "dtype_code"]

DTYPE_CODES = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.complex64): 2, np.dtype(np.complex128): 3}


def dtype_code(T) -> int:
    dt = np.dtype(T)
    if dt not in DTYPE_CODES:
        raise TypeError(f"element type {dt} is not supported on the device (Float32/Float64/ComplexF32/ComplexF64)")
    return DTYPE_CODES[dt]


class JetAbstractSpace:
    """src/Jets.jl:5-38."""

    def eltype(self):  # :12
        return self._T

    def size(self, i: int | None = None):  # :30
        s = self._size()
        return s if i is None else s[i]

    def ndims(self) -> int:  # :15
        return len(self._size())

    def length(self) -> int:  # :22  prod(size(R))
        n = 1
        for k in self._size():
            n *= int(k)
        return n

    def __len__(self) -> int:
        return self.length()


class JetSpace(JetAbstractSpace):
    """JetSpace(T, n...)  (src/Jets.jl:40-68)."""

    def __init__(self, T, *n):
        if len(n) == 1 and isinstance(n[0], (tuple, list)):  # JetSpace(T, (n1, n2))  :61
            n = tuple(n[0])
        self._T = np.dtype(T)
        self.n = tuple(int(k) for k in n)
        if any(k < 0 for k in self.n):
            raise ValueError("negative dimension")

    def _size(self):  # :63
        return self.n

    def vec(self) -> "JetSpace":  # :66
        return JetSpace(self._T, self.length())

    def similar(self, *dims) -> "JetSpace":  # :67-68
        if len(dims) == 1 and isinstance(dims[0], (tuple, list)):
            dims = tuple(dims[0])
        return JetSpace(self._T, *dims)

    def __eq__(self, other):
        return isinstance(other, JetSpace) and self._T == other._T and self.n == other.n

    def __hash__(self):
        return hash((self._T.str, self.n))

    def __repr__(self):
        return f"JetSpace({self._T.name}, {', '.join(map(str, self.n))})"


class JetSSpace(JetAbstractSpace):
    """JetSSpace(T, n, M, map)  (src/Jets.jl:407-446): a space of size `n` whose arrays store only the block `M` (the
    non-redundant part, e.g. the non-negative frequencies of the spectrum of a real signal); `map` takes the 0-based index
    tuple of an element OUTSIDE the stored block to the 0-based index tuple of the stored element it is the conjugate of."""

    def __init__(self, T, n, M, map):  # noqa: A002  (the reference's field name)
        self._T = np.dtype(T)
        self.n = tuple(int(k) for k in n)
        self.M = tuple(int(k) for k in M)
        self.map = map
        if len(self.n) != len(self.M):
            raise ValueError("n and M must have the same number of dimensions")

    def _size(self):  # :438
        return self.n

    def similar(self, *dims) -> "JetSSpace":  # :442-443
        if len(dims) == 1 and isinstance(dims[0], (tuple, list)):
            dims = tuple(dims[0])
        return JetSSpace(self._T, dims, self.M, self.map)

    def __eq__(self, other):
        return isinstance(other, JetSSpace) and self._T == other._T and self.n == other.n and self.M == other.M and self.map is other.map

    def __hash__(self):
        return hash((self._T.str, self.n, self.M, id(self.map)))

    def __repr__(self):
        return f"JetSSpace({self._T.name}, {self.n}, stored {self.M})"


class JetBSpace(JetAbstractSpace):
    """Block space: always 1-D, contiguous cumulative ranges (src/Jets.jl:736-760)."""

    def __init__(self, spaces: Sequence[JetAbstractSpace]):
        spaces = list(spaces)
        if not spaces:
            raise ValueError("JetBSpace needs at least one block")
        T = spaces[0].eltype()
        for s in spaces:
            if s.eltype() != T:
                raise TypeError("all blocks of a JetBSpace must share one element type")
        self._T = T
        self.spaces = spaces
        self.indices = []
        stop = 0  # :743  (1-based inclusive `stop`; 0-based half-open range(stop_prev, stop))
        for s in spaces:  # :744-748
            start = stop + 1
            stop = start + s.length() - 1
            self.indices.append(builtins.range(start - 1, stop))

    def _size(self):  # :755
        return (self.indices[-1].stop,)

    def vec(self) -> "JetBSpace":  # :758
        return self

    def similar(self, *dims) -> JetSpace:  # :759-760
        if len(dims) == 1 and isinstance(dims[0], (tuple, list)):
            dims = tuple(dims[0])
        if len(dims) != 1:
            raise ValueError("a block space is one dimensional")
        return JetSpace(self._T, *dims)

    def nblocks(self) -> int:  # :806
        return len(self.spaces)

    def block_lengths(self) -> list[int]:
        lens = getattr(self, "_lens", None)
        if lens is None:
            lens = self._lens = [len(r) for r in self.indices]      # (a space is immutable: computed once -- every zeros(R) asks)
        return lens

    def block_lengths_c(self):
        """The same as a ctypes int64 array (what jh_bvec_create takes), built once."""
        arr = getattr(self, "_lens_c", None)
        if arr is None:
            import ctypes as _C

            lens = self.block_lengths()
            arr = self._lens_c = (_C.c_int64 * len(lens))(*lens)
        return arr

    def __eq__(self, other):  # :753
        return isinstance(other, JetBSpace) and self.spaces == other.spaces and self.indices == other.indices

    def __hash__(self):
        return hash(tuple(self.spaces))

    def __repr__(self):
        return f"JetBSpace({self._T.name}, {self.nblocks()} blocks, length {self.length()})"
