"""CPU: the fp64 LSQR checker (oracle/lsqr_ref.py) against closed-form least-squares solutions."""
import numpy as np

from oracle.lsqr_ref import lsqr_fp64

RNG = np.random.default_rng(7)


def test_lsqr_solves_a_consistent_tall_diagonal_system():
    n, N = 40, 6
    a = [RNG.random(n) + 0.1 for _ in range(N)]
    x_true = RNG.standard_normal(n)
    matvec = lambda x: np.concatenate([g * x for g in a])
    rmatvec = lambda y: sum(g * y[i * n:(i + 1) * n] for i, g in enumerate(a))
    b = matvec(x_true)
    x, info = lsqr_fp64(matvec, rmatvec, b, n, atol=1e-14, btol=1e-14, maxiter=200)
    assert np.linalg.norm(x - x_true) / np.linalg.norm(x_true) < 1e-10
    assert info["istop"] in (1, 2, 4, 5) and info["r1norm"] < 1e-8 * np.linalg.norm(b)


def test_lsqr_matches_lstsq_on_an_inconsistent_dense_system_with_damping():
    A = RNG.standard_normal((30, 8))
    b = RNG.standard_normal(30)
    x, info = lsqr_fp64(lambda v: A @ v, lambda y: A.T @ y, b, 8, atol=1e-13, btol=1e-13, maxiter=100)
    assert np.allclose(x, np.linalg.lstsq(A, b, rcond=None)[0], atol=1e-9)
    damp = 0.7
    xd, _ = lsqr_fp64(lambda v: A @ v, lambda y: A.T @ y, b, 8, damp=damp, atol=1e-13, btol=1e-13, maxiter=100)
    assert np.allclose(xd, np.linalg.solve(A.T @ A + damp ** 2 * np.eye(8), A.T @ b), atol=1e-9)
    h = [r for _, r, _ in info["history"]]
    assert all(h[i + 1] <= h[i] * (1 + 1e-12) for i in range(len(h) - 1))      # residual norm is monotone


def test_lsqr_warm_start_and_complex():
    A = RNG.standard_normal((20, 5)) + 1j * RNG.standard_normal((20, 5))
    b = RNG.standard_normal(20) + 1j * RNG.standard_normal(20)
    ref = np.linalg.lstsq(A, b, rcond=None)[0]
    x, _ = lsqr_fp64(lambda v: A @ v, lambda y: A.conj().T @ y, b, 5, x0=ref + 0.1, atol=1e-13, btol=1e-13, maxiter=100)
    assert np.allclose(x, ref, atol=1e-9)
