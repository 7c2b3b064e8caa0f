// jh_dense.hip -- the dense child operator (JopBaz of test/runtests.jl:27-33): d .= A*m ; m .= A'*d for a
// column-major nr x nc matrix in HBM.  Bandwidth-bound on A (2 flop per element): plain VALU, no MFMA.
//  * y = A x        : one thread per row, columns walked in order, product rounded then added -- bit-identical
//                     to the sequential loop `for c: s += A[r,c]*x[c]`; consecutive lanes read consecutive rows
//                     of a column (coalesced), x[c] is a wave-uniform broadcast.
//  * y = A^H x      : one wave per column (a column is contiguous): lanes stride the rows with fp64 partials,
//                     wave64 shuffle reduction; tolerance-level parity (the reference's BLAS order is unknown).
#include "jh_internal.h"

namespace {

template <typename S, int E>
__global__ void k_gemv_rows(const S *__restrict__ A, int64_t nr, int64_t nc, const S *__restrict__ x, S *__restrict__ y)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nr) return;
    S accr = 0, acci = 0;
    for (int64_t c = 0; c < nc; c++) {
        const S *a = A + (r + c * nr) * E;
        if (E == 1) {
            accr = accr + a[0] * x[c];
        } else {
            S ar = a[0], ai = a[1], xr = x[2 * c], xi = x[2 * c + 1];
            accr = accr + (ar * xr - ai * xi);
            acci = acci + (ar * xi + ai * xr);
        }
    }
    y[r * E] = accr;
    if (E == 2) y[r * E + 1] = acci;
}

template <typename S, int E>
__global__ void k_gemv_cols(const S *__restrict__ A, int64_t nr, int64_t nc, const S *__restrict__ x, S *__restrict__ y)
{
    const int64_t c = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= nc) return;                                   // whole wave exits together
    const int lane = threadIdx.x & 63;
    const S *col = A + c * nr * E;
    double sr = 0.0, si = 0.0;
    for (int64_t r = lane; r < nr; r += 64) {
        if (E == 1) {
            sr += (double)col[r] * (double)x[r];
        } else {
            const double ar = col[2 * r], ai = -(double)col[2 * r + 1], xr = x[2 * r], xi = x[2 * r + 1];   // conj(A)
            sr += ar * xr - ai * xi;
            si += ar * xi + ai * xr;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sr += __shfl_down(sr, off, 64);
        if (E == 2) si += __shfl_down(si, off, 64);
    }
    if (lane == 0) {
        y[c * E] = (S)sr;
        if (E == 2) y[c * E + 1] = (S)si;
    }
}

template <typename S, int E>
int gemv(const void *A, int64_t nr, int64_t nc, void *y, const void *x, int adjoint)
{
    hipStream_t st = jh_ctx().stream;
    if (!adjoint) {
        if (nr == 0) return JH_OK;
        hipLaunchKernelGGL((k_gemv_rows<S, E>), dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, (const S *)A, nr, nc, (const S *)x, (S *)y);
    } else {
        if (nc == 0) return JH_OK;
        hipLaunchKernelGGL((k_gemv_cols<S, E>), dim3((unsigned)((nc + 3) / 4)), dim3(256), 0, st, (const S *)A, nr, nc, (const S *)x, (S *)y);
    }
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

}  // namespace

int jh_launch_gemv(const void *A, int64_t nr, int64_t nc, int dtype, void *y, const void *x, int adjoint)
{
    switch (dtype) {
    case JH_F32: return gemv<float, 1>(A, nr, nc, y, x, adjoint);
    case JH_F64: return gemv<double, 1>(A, nr, nc, y, x, adjoint);
    case JH_C32: return gemv<float, 2>(A, nr, nc, y, x, adjoint);
    case JH_C64: return gemv<double, 2>(A, nr, nc, y, x, adjoint);
    }
    return jh_fail(JH_ERR_INVALID, "gemv: unknown dtype %d", dtype);
}

extern "C" int jh_gemv(const void *A_device, int64_t nr, int64_t nc, int dtype, jh_bvec *y, const jh_bvec *x, int adjoint)
{
    JH_TRY(jh_require_ready());
    JH_REQUIRE(A_device && y && x, "jh_gemv: null argument");
    JH_REQUIRE(nr >= 0 && nc >= 0, "jh_gemv: negative dimension");
    JH_REQUIRE(y->dtype == dtype && x->dtype == dtype, "jh_gemv: dtype mismatch");
    const int64_t ylen = adjoint ? nc : nr, xlen = adjoint ? nr : nc;
    JH_REQUIRE(y->length == ylen && x->length == xlen, "jh_gemv: %s of a %lld x %lld matrix needs vectors of %lld and %lld elements, got %lld and %lld",
               adjoint ? "adjoint" : "forward", (long long)nr, (long long)nc, (long long)ylen, (long long)xlen, (long long)y->length,
               (long long)x->length);
    JH_REQUIRE(y->data != x->data, "jh_gemv: y must not alias x");
    return jh_launch_gemv(A_device, nr, nc, dtype, y->data, x->data, adjoint);
}
