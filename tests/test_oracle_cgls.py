"""CPU: the fp64 CGLS checker (oracle/cgls_ref.py) against closed forms -- it is what pins jets.jl_amd/cgls.py and jh_cgls_solve
(CGLS, like LSQR, has no counterpart inside Jets.jl: SURVEY.md 8 f-1)."""
import numpy as np

from oracle.cgls_ref import cgls_fp64
from oracle.lsqr_ref import lsqr_fp64


def test_cgls_solves_a_consistent_tall_diagonal_system():
    rng = np.random.default_rng(3)
    nrow, n = 4, 50
    a = rng.random((nrow, n)) + 0.1
    xt = rng.standard_normal(n)
    b = (a * xt).ravel()
    x, info = cgls_fp64(lambda v: (a * v).ravel(), lambda y: (a * y.reshape(nrow, n)).sum(0), b, n, atol=1e-14, btol=1e-12, maxiter=200)
    assert info["istop"] in (1, 2) and np.linalg.norm(x - xt) <= 1e-9 * np.linalg.norm(xt)


def test_cgls_matches_lstsq_on_an_inconsistent_dense_system_with_damping():
    rng = np.random.default_rng(4)
    A = rng.standard_normal((40, 12))
    b = rng.standard_normal(40)
    damp = 0.3
    x, info = cgls_fp64(lambda v: A @ v, lambda y: A.T @ y, b, 12, damp=damp, atol=1e-13, btol=0.0, maxiter=100)
    want = np.linalg.solve(A.T @ A + damp ** 2 * np.eye(12), A.T @ b)
    assert np.linalg.norm(x - want) <= 1e-10 * np.linalg.norm(want)
    assert info["istop"] == 2 and info["itn"] <= 13                       # exact termination in n steps, give or take rounding
    # the history is (itn, ||r||, ||A'r - damp^2 x||): monotone ||r|| is NOT guaranteed for the damped problem, the last entries are small
    assert info["history"][-1][2] <= 1e-12 * info["history"][0][2] * 10 or info["history"][-1][2] < 1e-10


def test_cgls_warm_start_complex_and_agreement_with_lsqr():
    rng = np.random.default_rng(5)
    A = rng.standard_normal((30, 8)) + 1j * rng.standard_normal((30, 8))
    b = rng.standard_normal(30) + 1j * rng.standard_normal(30)
    x0 = rng.standard_normal(8) + 1j * rng.standard_normal(8)
    mv, rmv = (lambda v: A @ v), (lambda y: A.conj().T @ y)
    x, info = cgls_fp64(mv, rmv, b, 8, x0=x0, atol=1e-14, btol=0.0, maxiter=60)
    want = np.linalg.lstsq(A, b, rcond=None)[0]
    assert np.linalg.norm(x - want) <= 1e-10 * np.linalg.norm(want)
    # CGLS and LSQR are mathematically the same Krylov method: after k steps from x0 = 0 their iterates agree
    xc, _ = cgls_fp64(mv, rmv, b, 8, atol=0.0, btol=0.0, maxiter=5)
    xl, _ = lsqr_fp64(mv, rmv, b, 8, atol=0.0, btol=0.0, conlim=0.0, maxiter=5)
    assert np.linalg.norm(xc - xl) <= 1e-10 * np.linalg.norm(xl)


def test_cgls_breakdown_and_zero_rhs():
    A = np.zeros((6, 3))
    x, info = cgls_fp64(lambda v: A @ v, lambda y: A.T @ y, np.ones(6), 3, maxiter=10)
    assert info["itn"] == 0 and info["istop"] == 0 and not x.any()        # A'b = 0: nothing to do
