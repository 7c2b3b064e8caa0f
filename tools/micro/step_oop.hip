// Experiment: does the one-pass LSQR step run faster when the updated u goes to a SEPARATE buffer instead of in place?
// (in place, every workgroup writes the DRAM pages it has just read; out of place the write stream has pages of its own, like
// the forward's, which reaches 6.45 TB/s where the in-place step stops at 6.05.)  Same arithmetic, base structure of
// k_tall_diag_bidiag (U = 1, DEPTH = 4, 512 lanes).      ./step_oop NROW [EDGE]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float V4 __attribute__((ext_vector_type(4)));

__global__ void k_fill(float *p, int64_t n, uint64_t seed)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        uint64_t z = (uint64_t)i + seed * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = (float)((z >> 40) & 0xFFFFFF) * (1.0f / 16777216.0f);
    }
}

template <int DEPTH, int BLK, int WHAT>      // WHAT 0: in place; 1: u_out separate; 2: triad only (no ordered sum: u_out = alpha a v + beta u)
__global__ __launch_bounds__(BLK) void k_step(const float *__restrict__ a, const float *u, float *uo, const float *__restrict__ v,
                                              float *__restrict__ w, int64_t n, int64_t nrow, float alpha, float beta)
{
    const int64_t s = ((int64_t)blockIdx.x * BLK + threadIdx.x) * 4;
    const V4 vv = *reinterpret_cast<const V4 *>(v + s);
    V4 acc = (V4)0.f;
    int64_t i = 0;
    for (; i + DEPTH <= nrow; i += DEPTH) {
        V4 av[DEPTH], uv[DEPTH];
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            av[j] = __builtin_nontemporal_load(reinterpret_cast<const V4 *>(a + (i + j) * n + s));
            uv[j] = __builtin_nontemporal_load(reinterpret_cast<const V4 *>(u + (i + j) * n + s));
        }
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            V4 r = (V4)alpha * (av[j] * vv);
            V4 s2 = (V4)beta * uv[j];
            r = r + s2;
            __builtin_nontemporal_store(r, reinterpret_cast<V4 *>(uo + (i + j) * n + s));
            if (WHAT != 2) acc = acc + av[j] * r;
        }
    }
    if (WHAT != 2) *reinterpret_cast<V4 *>(w + s) = acc;
}

int main(int argc, char **argv)
{
    const int64_t nrow = argc > 1 ? atoll(argv[1]) : 256, edge = argc > 2 ? atoll(argv[2]) : 256;
    const int64_t n = edge * edge * edge;
    float *A, *U, *U2, *V, *W;
    CK(hipSetDevice(0));
    CK(hipMalloc(&A, (size_t)nrow * n * 4));
    CK(hipMalloc(&U, (size_t)nrow * n * 4));
    CK(hipMalloc(&U2, (size_t)nrow * n * 4));
    CK(hipMalloc(&V, (size_t)n * 4));
    CK(hipMalloc(&W, (size_t)n * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, A, nrow * n, 1ull);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, U, nrow * n, 3ull);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, U2, nrow * n, 5ull);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, V, n, 2ull);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double bytes = (3.0 * nrow + 2.0) * n * 4;
    const unsigned grid = (unsigned)(n / 4 / 512);
    printf("== one-pass step, %lld x %lld^3 Float32 (base structure 512 x 1 x 4) ==\n", (long long)nrow, (long long)edge);
    for (int rnd = 0; rnd < 2; rnd++)
        for (int what = 0; what < 3; what++) {
            std::vector<float> ms;
            for (int rep = 0; rep < 7; rep++) {
                float *src = U, *dst = (what == 0) ? U : U2;
                if (what != 0 && (rep & 1)) { src = U2; dst = U; }             // ping-pong like an LSQR loop would
                CK(hipEventRecord(e0, 0));
                if (what == 0) hipLaunchKernelGGL((k_step<4, 512, 0>), dim3(grid), dim3(512), 0, 0, A, src, dst, V, W, n, nrow, 0.5f, 0.5f);
                else if (what == 1) hipLaunchKernelGGL((k_step<4, 512, 1>), dim3(grid), dim3(512), 0, 0, A, src, dst, V, W, n, nrow, 0.5f, 0.5f);
                else hipLaunchKernelGGL((k_step<4, 512, 2>), dim3(grid), dim3(512), 0, 0, A, src, dst, V, W, n, nrow, 0.5f, 0.5f);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float t = 0;
                CK(hipEventElapsedTime(&t, e0, e1));
                if (rep >= 2) ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf("%-34s min %8.3f ms  med %8.3f ms  %7.1f GB/s\n", what == 0 ? "in place" : (what == 1 ? "out of place (ping-pong)" : "out of place, no ordered sum"),
                   ms[0], ms[ms.size() / 2], bytes / ms[0] / 1e6);
        }
    return 0;
}
