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
