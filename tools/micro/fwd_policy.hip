// Experiment: cache-policy bits on the tall forward's stores (and loads).  d_i = a_i .* m, one block row per workgroup (512 lanes x 8
// packs, all rows concurrent -- candidate 6 of the library) and the 16-row sequential sweep (candidate 0's order), stores written as
// inline assembly with every combination of nt / sc0 / sc1 gfx950 accepts.      ./fwd_policy NROW [EDGE]
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

template <int POL> __device__ inline void store16(void *p, V4 v)
{
    if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    if (POL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    if (POL == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

// G rows per workgroup; walk 1: rows fastest (all rows concurrent), walk 0: tiles fastest (sequential sweep)
template <int U, int BLK, int POL, bool LDNT>
__global__ __launch_bounds__(BLK) void k_fwd(const float *__restrict__ a, const float *__restrict__ m, float *d, int64_t n, int64_t nrow, int G,
                                             unsigned ntiles, unsigned ngroups, int walk)
{
    const unsigned tile = walk ? blockIdx.x / ngroups : blockIdx.x % ntiles;
    const unsigned grp = walk ? blockIdx.x % ngroups : blockIdx.x / ntiles;
    int64_t off[U];
    V4 mv[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        off[k] = (((int64_t)tile * U + k) * BLK + threadIdx.x) * 4;
        mv[k] = *reinterpret_cast<const V4 *>(m + off[k]);
    }
    const int64_t i0 = (int64_t)grp * G, i1 = (i0 + G < nrow) ? i0 + G : nrow;
    for (int64_t i = i0; i < i1; i++) {
        V4 av[U];
#pragma unroll
        for (int k = 0; k < U; k++)
            av[k] = LDNT ? __builtin_nontemporal_load(reinterpret_cast<const V4 *>(a + i * n + off[k])) : *reinterpret_cast<const V4 *>(a + i * n + off[k]);
#pragma unroll
        for (int k = 0; k < U; k++) store16<POL>(d + i * n + off[k], av[k] * mv[k]);
    }
}

int main(int argc, char **argv)
{
    const int64_t nrow = argc > 1 ? atoll(argv[1]) : 256, edge = argc > 2 ? atoll(argv[2]) : 256;
    const int64_t n = edge * edge * edge;
    float *A, *D, *M;
    CK(hipSetDevice(0));
    CK(hipMalloc(&A, (size_t)nrow * n * 4));
    CK(hipMalloc(&D, (size_t)nrow * n * 4));
    CK(hipMalloc(&M, (size_t)n * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, A, nrow * n, 1ull);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, M, n, 2ull);
    CK(hipMemset(D, 0, (size_t)nrow * n * 4));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double bytes = (2.0 * nrow + 1.0) * n * 4;
    printf("== tall forward, %lld x %lld^3 Float32: store policy x load policy ==\n", (long long)nrow, (long long)edge);
    const char *pol[8] = {"plain", "nt", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt", "sc0", "sc0 nt"};
    auto run = [&](const char *shape, auto kern, int U, int BLK, int G, int walk, int p, bool ldnt) -> int {
        const unsigned ntiles = (unsigned)(n / 4 / ((int64_t)U * BLK)), ngroups = (unsigned)((nrow + G - 1) / G);
        if ((int64_t)ntiles * U * BLK * 4 != n) return 0;
        std::vector<float> ms;
        for (int rep = 0; rep < 6; rep++) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern, dim3(ntiles * ngroups), dim3(BLK), 0, 0, A, M, D, n, nrow, G, ntiles, ngroups, walk);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t = 0;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (rep >= 2) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-26s store %-11s load %-5s  min %8.3f ms  %7.1f GB/s\n", shape, pol[p], ldnt ? "nt" : "plain", ms[0], bytes / ms[0] / 1e6);
        fflush(stdout);
        return 0;
    };
#define ROW1(P, L) if (run("512x8, 1 row/wg, rows conc", k_fwd<8, 512, P, L>, 8, 512, 1, 1, P, L)) return 1;
#define SEQ16(P, L) if (run("1024x8, 16 rows/wg, seq", k_fwd<8, 1024, P, L>, 8, 1024, 16, 0, P, L)) return 1;
    for (int rnd = 0; rnd < 2; rnd++) {
        ROW1(0, true) ROW1(1, true) ROW1(2, true) ROW1(3, true) ROW1(4, true) ROW1(5, true) ROW1(6, true) ROW1(7, true) ROW1(1, false) ROW1(4, false)
        SEQ16(0, true) SEQ16(1, true) SEQ16(2, true) SEQ16(3, true) SEQ16(4, true) SEQ16(5, true) SEQ16(6, true) SEQ16(7, true) SEQ16(1, false)
    }
    return 0;
}
