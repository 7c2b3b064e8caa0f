// jh_dense.hip -- the dense child operator (JopBaz of test/runtests.jl:27-33): d .= A*m ; m .= A'*d for a
// column-major nr x nc matrix in HBM.  Bandwidth-bound on A (2 flop per element, ridge ~20 flop/B): plain VALU with
// 16-byte loads, no MFMA.  Algorithmic bytes: nr*nc*s (the matrix once) + the two vectors.
//  * y = A x   (rows kernel): a thread owns VEC consecutive rows (one 16-byte load per column: consecutive lanes read
//              consecutive 16-byte pieces of a column), walks its column range in order, product rounded then added.
//              With ONE column chunk the result is bit-identical to the sequential loop `for c: s += A[r,c]*x[c]`;
//              large matrices with few rows split the columns over grid.y into fp partial rows that a second kernel
//              adds in chunk order (deterministic; tolerance-level parity, like any BLAS).
//  * y = A^H x (cols kernel): a column is contiguous: one wave per (column, row chunk), 16-byte loads, fp64 partials,
//              wave64 shuffle reduction; row chunks are added in order by a second kernel.
#include "jh_internal.h"

namespace {

template <typename S, int NS> struct vec_of { typedef S type __attribute__((ext_vector_type(NS))); };
template <typename S> struct vec_of<S, 1> { typedef S type; };

// ---- y = A x ---------------------------------------------------------------------------------------
// E scalars per element, NS scalars per 16-byte vector (NS == E: one element per lane, the unaligned fallback).
template <typename S, int E, int NS>
__global__ __launch_bounds__(256) void k_gemv_rows(const S *__restrict__ A, int64_t nr, int64_t nc, const S *__restrict__ x,
                                                   S *__restrict__ out, int64_t cols_per_chunk)
{
    typedef typename vec_of<S, NS>::type V;
    const int64_t s = ((int64_t)blockIdx.x * 256 + threadIdx.x) * NS;          // first scalar row index of this lane
    const int64_t ns = nr * E;
    if (s >= ns) return;
    const int64_t c0 = (int64_t)blockIdx.y * cols_per_chunk;
    const int64_t c1 = c0 + cols_per_chunk < nc ? c0 + cols_per_chunk : nc;
    V acc = (V)(S)0;
    const S *col = A + c0 * ns + s;
    for (int64_t c = c0; c < c1; c++, col += ns) {
        V a = __builtin_nontemporal_load(reinterpret_cast<const V *>(col));
        if constexpr (E == 1) {
            acc = acc + a * (V)x[c];                                           // product rounded, then added (no FMA: -ffp-contract=off)
        } else {
            const S xr = x[2 * c], xi = x[2 * c + 1];
            V p;
#pragma unroll
            for (int e = 0; e < NS; e += 2) {
                p[e] = a[e] * xr - a[e + 1] * xi;
                p[e + 1] = a[e] * xi + a[e + 1] * xr;
            }
            acc = acc + p;
        }
    }
    *reinterpret_cast<V *>(out + (int64_t)blockIdx.y * ns + s) = acc;          // chunk 0 of a one-chunk launch is y itself
}

// out[k] = sum over chunks (in order) of partial[chunk][k]
template <typename S>
__global__ void k_sum_chunks(const S *__restrict__ partial, int64_t n, int nchunks, S *__restrict__ out)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    S acc = partial[k];
    for (int c = 1; c < nchunks; c++) acc = acc + partial[(int64_t)c * n + k];
    out[k] = acc;
}

// ---- y = A^H x -------------------------------------------------------------------------------------
template <typename S, int E, int NS>
__global__ __launch_bounds__(256) void k_gemv_cols(const S *__restrict__ A, int64_t nr, int64_t nc, const S *__restrict__ x,
                                                   double *__restrict__ partial, int64_t rows_per_chunk)
{
    typedef typename vec_of<S, NS>::type V;
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= nc) return;                                                       // whole wave exits together
    const int lane = threadIdx.x & 63;
    const int64_t ns = nr * E;
    const int64_t s0 = (int64_t)blockIdx.y * rows_per_chunk * E;
    const int64_t s1 = s0 + rows_per_chunk * E < ns ? s0 + rows_per_chunk * E : ns;
    const S *col = A + c * ns;
    double sr = 0.0, si = 0.0;
    for (int64_t s = s0 + (int64_t)lane * NS; s < s1; s += 64 * NS) {
        V a = __builtin_nontemporal_load(reinterpret_cast<const V *>(col + s));
        V xv = *reinterpret_cast<const V *>(x + s);
        if constexpr (E == 1) {
#pragma unroll
            for (int e = 0; e < NS; e++) {
                if constexpr (NS == 1) sr += (double)a * (double)xv;
                else sr += (double)a[e] * (double)xv[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < NS; e += 2) {                                   // conj(a) * x
                const double ar = a[e], ai = -(double)a[e + 1], xr = xv[e], xi = xv[e + 1];
                sr += ar * xr - ai * xi;
                si += ar * xi + ai * xr;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sr += __shfl_down(sr, off, 64);
        if (E == 2) si += __shfl_down(si, off, 64);
    }
    if (lane == 0) {
        double *p = partial + ((int64_t)blockIdx.y * nc + c) * 2;
        p[0] = sr;
        p[1] = si;
    }
}

template <typename S, int E>
__global__ void k_sum_col_chunks(const double *__restrict__ partial, int64_t nc, int nchunks, S *__restrict__ y)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= nc) return;
    double sr = 0.0, si = 0.0;
    for (int k = 0; k < nchunks; k++) {
        sr += partial[((int64_t)k * nc + c) * 2];
        si += partial[((int64_t)k * nc + c) * 2 + 1];
    }
    y[c * E] = (S)sr;
    if (E == 2) y[c * E + 1] = (S)si;
}

template <typename S, int E>
int gemv(const void *A, int64_t nr, int64_t nc, void *y, const void *x, int adjoint)
{
    jh_context &c = jh_ctx();
    hipStream_t st = c.stream;
    constexpr int NSV = (16 / sizeof(S)) >= E ? (16 / sizeof(S)) : E;
    const int64_t ns = nr * E;
    // 16-byte path: every column start and both vectors aligned, whole vectors per column
    const bool vec_ok = ((ns * (int64_t)sizeof(S)) % 16 == 0) && ((((uintptr_t)A) | ((uintptr_t)x) | ((uintptr_t)y)) & 15u) == 0;
    const double bytes = (double)nr * (double)nc * sizeof(S) * E;
    if (!adjoint) {
        if (nr == 0) return JH_OK;
        if (nc == 0) return jh_launch_fill_range(y, E == 2 ? (sizeof(S) == 4 ? JH_C32 : JH_C64) : (sizeof(S) == 4 ? JH_F32 : JH_F64), nr, 0.0, 0.0);
        const int NS = vec_ok ? NSV : E;
        const int64_t row_wgs = (ns / NS + 255) / 256;
        int64_t nchunks = 1;
        if (bytes >= (double)(1 << 20) && row_wgs < 2048) {                    // few rows: split the columns (tiny matrices stay one ordered sum)
            nchunks = (2048 + row_wgs - 1) / row_wgs;
            const int64_t maxc = (nc + 31) / 32;
            if (nchunks > maxc) nchunks = maxc;
            if (nchunks > 65535) nchunks = 65535;
        }
        const int64_t cpc = (nc + nchunks - 1) / nchunks;
        nchunks = (nc + cpc - 1) / cpc;
        S *out = (S *)y;
        if (nchunks > 1) {   // partial rows live in the partials buffer (the block loop's dtmp/mtmp own the scratch buffer)
            JH_TRY(jh_ensure_partials(((int64_t)nchunks * ns * (int64_t)sizeof(S) + 7) / 8 + 2));
            out = (S *)c.part_dev;
        }
        if (vec_ok)
            hipLaunchKernelGGL((k_gemv_rows<S, E, NSV>), dim3((unsigned)row_wgs, (unsigned)nchunks), dim3(256), 0, st, (const S *)A, nr, nc,
                               (const S *)x, out, cpc);
        else
            hipLaunchKernelGGL((k_gemv_rows<S, E, E>), dim3((unsigned)row_wgs, (unsigned)nchunks), dim3(256), 0, st, (const S *)A, nr, nc,
                               (const S *)x, out, cpc);
        JH_CHECK_HIP(hipGetLastError());
        if (nchunks > 1) {
            hipLaunchKernelGGL((k_sum_chunks<S>), dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, st, out, ns, (int)nchunks, (S *)y);
            JH_CHECK_HIP(hipGetLastError());
        }
        return JH_OK;
    }
    if (nc == 0) return JH_OK;
    int64_t nchunks = 1;
    const int64_t col_wgs = (nc + 3) / 4;
    if (bytes >= (double)(1 << 20) && col_wgs < 2048) {                        // few columns: split the rows
        nchunks = (2048 + col_wgs - 1) / col_wgs;
        const int64_t maxc = (nr + 4095) / 4096;
        if (nchunks > maxc) nchunks = maxc;
        if (nchunks > 65535) nchunks = 65535;
        if (nchunks < 1) nchunks = 1;
    }
    int64_t rpc = (nr + nchunks - 1) / nchunks;
    rpc = (rpc + 3) / 4 * 4;                                                   // keep chunk starts 16-byte aligned
    if (rpc < 4) rpc = 4;
    nchunks = nr ? (nr + rpc - 1) / rpc : 1;
    JH_TRY(jh_ensure_partials(2 * nchunks * nc));
    if (vec_ok)
        hipLaunchKernelGGL((k_gemv_cols<S, E, NSV>), dim3((unsigned)col_wgs, (unsigned)nchunks), dim3(256), 0, st, (const S *)A, nr, nc,
                           (const S *)x, c.part_dev, rpc);
    else
        hipLaunchKernelGGL((k_gemv_cols<S, E, E>), dim3((unsigned)col_wgs, (unsigned)nchunks), dim3(256), 0, st, (const S *)A, nr, nc,
                           (const S *)x, c.part_dev, rpc);
    JH_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL((k_sum_col_chunks<S, E>), dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, st, c.part_dev, nc, (int)nchunks, (S *)y);
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
