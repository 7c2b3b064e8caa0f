"""Shared builders for the parity tests: the SAME seeded operator on the device (through the product
package / C ABI) and in the CPU oracle."""
from __future__ import annotations

import numpy as np

DTYPES = [np.float32, np.float64, np.complex64, np.complex128]
SEED_A, SEED_M, SEED_D = 1, 2, 3


def u01(oracle, dtype, seed, stream, n, index0=0):
    return oracle.rng_u01(dtype, seed, stream, index0, n)


def centered(x):
    """map U[0,1) to a signed / complex-spread value without changing bits on either side: host only."""
    return x


def make_tall_diag(J, oracle, dtype, nrow, shape, seed=SEED_A):
    """nrow x 1 block operator of diagonal blocks, coefficients from the counter RNG, stream = row."""
    spc = J.JetSpace(dtype, *shape)
    n = spc.length()
    diags_dev = [J.rand(spc, seed=seed, stream=i) for i in range(nrow)]
    A = J.blockop([[J.JopDiagonal(dg)] for dg in diags_dev])
    diags_np = [u01(oracle, dtype, seed, i, n) for i in range(nrow)]
    ops = [[oracle.Block("diag", n, coeff=dg)] for dg in diags_np]
    return A, diags_dev, ops, diags_np


def dev_blocks_to_numpy(x):
    """device BlockArray / DeviceArray -> list of flat numpy blocks."""
    if hasattr(x, "arrays"):
        flat = x.to_numpy()
        return [flat[r.start:r.stop].copy() for r in x.indices]
    return [x.to_numpy().ravel(order="F").copy()]


def bits_equal(a: np.ndarray, b: np.ndarray) -> bool:
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


def assert_bits_equal(a, b, what=""):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.dtype == b.dtype, f"{what}: dtype {a.dtype} vs {b.dtype}"
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    if a.tobytes() != b.tobytes():
        bad = np.flatnonzero(a.view(np.uint8).reshape(a.size, -1).any(axis=1) != b.view(np.uint8).reshape(b.size, -1).any(axis=1))
        diff = np.flatnonzero(a.ravel() != b.ravel())
        raise AssertionError(f"{what}: {diff.size} of {a.size} elements differ bitwise; first at {diff[:5]}: "
                             f"{a.ravel()[diff[:5]]} vs {b.ravel()[diff[:5]]} (zero-pattern mismatches: {bad.size})")


def rel_err(a, b) -> float:
    a, b = np.asarray(a, dtype=np.complex128), np.asarray(b, dtype=np.complex128)
    den = np.linalg.norm(b.ravel())
    return float(np.linalg.norm((a - b).ravel()) / (den if den else 1.0))


def assert_same_values(a, b, what=""):
    """Bit for bit, except that a NaN matches any NaN: x86 and CDNA pick different payloads / signs for the NaN an invalid operation
    produces, and which operand's payload survives -- no Julia program can observe that through `==` or `isnan`."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.dtype == b.dtype and a.shape == b.shape, f"{what}: {a.dtype}{a.shape} vs {b.dtype}{b.shape}"
    if a.dtype.kind == "c":
        rt = np.float32 if a.dtype == np.complex64 else np.float64
        a, b = a.view(rt), b.view(rt)
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), f"{what}: NaN in different places ({int(na.sum())} vs {int(nb.sum())}; first at {np.flatnonzero(na != nb)[:5]})"
    it = np.uint32 if a.dtype == np.float32 else np.uint64
    ia, ib = a.view(it)[~na.ravel().reshape(a.shape)] if a.ndim else a.view(it), b.view(it)[~nb.ravel().reshape(b.shape)] if b.ndim else b.view(it)
    if not np.array_equal(ia, ib):
        bad = np.flatnonzero(ia != ib)
        raise AssertionError(f"{what}: {bad.size} non-NaN elements differ bitwise; first at {bad[:5]}")


# ---- Julia's scalar arithmetic, spelled out with real numpy operations (each one IEEE-rounded, no contraction) ----------------------------------
def julia_scalar_term(a, x):
    """`a * x[k]` as Julia computes it for a scalar of a's TYPE (src/Jets.jl:1159 `d .= a * m`; base/complex.jl), BEFORE the store's
    conversion: (re, im, precision) with re / im arrays in the arithmetic's precision.  A Python / numpy complex is a Complex (full product,
    also with a zero imaginary part), a Real multiplies part by part; numpy's float64 / complex128 are Float64-based (promoted arithmetic
    against 32-bit elements), everything else is taken in x's precision."""
    x = np.asarray(x)
    R = np.float32 if x.dtype in (np.dtype(np.float32), np.dtype(np.complex64)) else np.float64
    W = np.float64 if isinstance(a, (np.float64, np.complex128)) else R
    cplx = isinstance(a, (complex, np.complexfloating))
    ar, ai = W(complex(a).real), W(complex(a).imag)
    xr = np.real(x).astype(W)
    xi = np.imag(x).astype(W) if x.dtype.kind == "c" else None
    with np.errstate(all="ignore"):
        if xi is None:
            assert not cplx
            return ar * xr, None, W
        if not cplx:
            return ar * xr, ar * xi, W
        return ar * xr - ai * xi, ar * xi + ai * xr, W


def julia_lincomb(coefs, xs):
    """`dst .= c1 .* x1 .+ c2 .* x2 .+ ...` left to right with Julia's promotion: a sum with a Float64 operand is a Float64 sum; ONE rounding
    into the vectors' element type at the end."""
    x0 = np.asarray(xs[0])
    R = np.float32 if x0.dtype in (np.dtype(np.float32), np.dtype(np.complex64)) else np.float64
    acc_r = acc_i = None
    prec = R
    with np.errstate(all="ignore"):
        for c, x in zip(coefs, xs):
            tr, ti, W = julia_scalar_term(c, x)
            if acc_r is None:
                acc_r, acc_i, prec = tr, ti, W
                continue
            prec = np.float64 if np.float64 in (prec, W) else R
            acc_r = acc_r.astype(prec) + tr.astype(prec)
            if ti is not None:
                acc_i = acc_i.astype(prec) + ti.astype(prec)
    with np.errstate(all="ignore"):
        if acc_i is None:
            return acc_r.astype(R)
        out = np.empty(x0.shape, dtype=x0.dtype)
        out.real, out.imag = acc_r.astype(R), acc_i.astype(R)
    return out
