// gemv_mfma.hip -- A/B for the dense-GEMV block kind (JopBaz, test/runtests.jl:27-33): the library's VALU structure (a thread owns
// four consecutive rows of a column-major matrix and walks its columns in order, 16-byte loads) against an MFMA formulation
// (v_mfma_f32_16x16x4_f32 with the vector broadcast into all 16 B-columns).  north_star names MFMA "only where a block's df! is
// a true dense GEMV"; a GEMV moves 4 bytes of matrix per 2 flops (0.5 flop/B against a ridge of ~20 flop/B), so both are
// HBM-bound -- this harness records the measurement instead of the argument.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/gemv_mfma tools/micro/gemv_mfma.hip && tools/micro/gemv_mfma
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float V4 __attribute__((ext_vector_type(4)));

__device__ inline V4 ldnt(const float *p)
{
    typedef const V4 __attribute__((address_space(1))) *gp;
    return __builtin_nontemporal_load((gp)p);
}

__global__ void k_fill(float *p, int64_t n, uint64_t seed)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t z = seed + (uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = (float)((z ^ (z >> 31)) >> 40) * (1.0f / 16777216.0f);
    }
}

// VALU: thread = 4 consecutive rows, all columns in order, product rounded then added (no FMA): the sequential loop's bits
template <int UNR>
__global__ __launch_bounds__(256) void k_gemv_valu(const float *__restrict__ A, const float *__restrict__ x, float *__restrict__ y, int64_t nr, int64_t nc)
{
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (r >= nr) return;
    V4 acc = (V4)0.f;
    int64_t c = 0;
    for (; c + UNR <= nc; c += UNR) {
        V4 a[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) a[u] = ldnt(A + r + (c + u) * nr);
#pragma unroll
        for (int u = 0; u < UNR; u++) { V4 p = a[u] * x[c + u]; acc = acc + p; }
    }
    for (; c < nc; c++) { V4 p = ldnt(A + r + c * nr) * x[c]; acc = acc + p; }
    *reinterpret_cast<V4 *>(y + r) = acc;
}

// MFMA: a wave owns 64 rows; per step it loads 64 rows x 4 columns (16 B per lane: lane (q, kk) holds rows 4q..4q+3 of column k0+kk)
// and issues four 16x16x4 MFMAs, MFMA r covering the rows {4q + r}; B = x[k0 + kk] in every column, so all 16 result columns agree
template <int UNR>
__global__ __launch_bounds__(256) void k_gemv_mfma(const float *__restrict__ A, const float *__restrict__ x, float *__restrict__ y, int64_t nr, int64_t nc)
{
    const int lane = threadIdx.x & 63, q = lane & 15, kk = lane >> 4;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (row0 >= nr) return;
    V4 acc[4] = {(V4)0.f, (V4)0.f, (V4)0.f, (V4)0.f};
    const float *base = A + row0 + 4 * q + (int64_t)kk * nr;
    for (int64_t k0 = 0; k0 < nc; k0 += 4 * UNR) {
        V4 a[UNR];
        float b[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) { a[u] = ldnt(base + (k0 + 4 * u) * nr); b[u] = x[k0 + 4 * u + kk]; }
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][0], b[u], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][1], b[u], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][2], b[u], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][3], b[u], acc[3], 0, 0, 0);
        }
    }
    if (q == 0) {                                              // result column 0: lane 16g holds D[4g + i][0] in acc[r][i]
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 4; i++) y[row0 + 4 * (4 * kk + i) + r] = acc[r][i];
    }
}

template <typename F> static float timeit(F &&f, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    std::vector<float> t;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[0];
}

int main()
{
    CK(hipSetDevice(0));
    const int64_t shapes[][2] = {{1 << 20, 256}, {1 << 18, 1024}, {1 << 22, 64}, {1 << 16, 4096}};
    float *A, *x, *y0, *y1;
    CK(hipMalloc(&A, (size_t)1 << 30)); CK(hipMalloc(&x, 4096 * 4)); CK(hipMalloc(&y0, ((size_t)1 << 22) * 4)); CK(hipMalloc(&y1, ((size_t)1 << 22) * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, A, (int64_t)1 << 28, 1ull);
    hipLaunchKernelGGL(k_fill, dim3(16), dim3(256), 0, 0, x, (int64_t)4096, 2ull);
    CK(hipDeviceSynchronize());
    printf("dense GEMV y = A x, column-major Float32, 1 GiB of matrix per shape (MI355X); GB/s of matrix bytes\n");
    for (auto &sh : shapes) {
        const int64_t nr = sh[0], nc = sh[1];
        const double bytes = (double)nr * nc * 4;
        auto valu = [&] { hipLaunchKernelGGL((k_gemv_valu<8>), dim3((unsigned)(nr / 4 / 256)), dim3(256), 0, 0, A, x, y0, nr, nc); };
        auto mfma = [&] { hipLaunchKernelGGL((k_gemv_mfma<4>), dim3((unsigned)(nr / 64 / 4)), dim3(256), 0, 0, A, x, y1, nr, nc); };
        const float tv = timeit(valu, 7), tm = timeit(mfma, 7);
        std::vector<float> h0((size_t)nr), h1((size_t)nr);
        CK(hipMemcpy(h0.data(), y0, (size_t)nr * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h1.data(), y1, (size_t)nr * 4, hipMemcpyDeviceToHost));
        double num = 0, den = 0;
        size_t same = 0;
        for (size_t i = 0; i < (size_t)nr; i++) { num += (double)(h0[i] - h1[i]) * (h0[i] - h1[i]); den += (double)h0[i] * h0[i]; same += (h0[i] == h1[i]); }
        printf("%8lld x %-5lld  VALU (ordered, bit-exact vs the sequential loop) %7.3f ms %7.1f GB/s | MFMA 16x16x4 f32 %7.3f ms %7.1f GB/s | "
               "rel l2 difference %.1e, %5.1f %% of the elements identical\n", (long long)nr, (long long)nc, tv, bytes / tv / 1e6, tm, bytes / tm / 1e6,
               std::sqrt(num / den), 100.0 * same / nr);
    }
    return 0;
}
