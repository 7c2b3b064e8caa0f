// Experiment (round 4): at 128 / 256 block rows the tall forward d_i = a_i .* m runs 10 % below a plain copy BETWEEN THE SAME TWO SLABS
// (profiles/exp_r04_fwd_vs_copy.txt), at 1024 rows it equals it.  What costs the 10 % -- reading the model tile, or workgroups that are born,
// move one tile and die?  Variants of one access order (tile, then all rows concurrently; 256 lanes x 1 pack = the library's candidate 7):
//   copy        d <- a, grid-stride (the library's copy kernel)
//   wg          one workgroup per (tile, row)                                      -- the library's walk
//   wg_nom      the same without the model read (d <- a .* const)                  -- what the model read costs
//   wg_msmall   the same, model tile index masked to 64 KiB (always an L2 hit)     -- ... and whether it is its latency or its traffic
//   pers        persistent workgroups striding over the SAME (tile, row) sequence, ITEMS packs in flight per lane
//   pers_rows   persistent workgroups that own a tile range and walk the rows with the model tile in registers, R rows in flight
//      ./fwd_small_rows NROW [EDGE]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float V4 __attribute__((ext_vector_type(4)));
typedef const V4 __attribute__((address_space(1))) *gcp;
typedef V4 __attribute__((address_space(1))) *gp;

__device__ inline V4 ldnt(const float *p) { return __builtin_nontemporal_load((gcp)p); }
__device__ inline void stnt(float *p, V4 v) { __builtin_nontemporal_store(v, (gp)p); }

__global__ void k_fill(float *p, int64_t n, uint64_t seed)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        uint64_t z = (uint64_t)i + seed * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = (float)((z >> 40) & 0xFFFFFF) * (1.0f / 16777216.0f);
    }
}

__global__ __launch_bounds__(256) void k_copy(const float *__restrict__ a, float *__restrict__ d, int64_t npack)
{
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < npack; v += (int64_t)gridDim.x * 256) stnt(d + v * 4, ldnt(a + v * 4));
}

// MODE 0: model tile from m; 1: no model read; 2: model index masked to 64 KiB
template <int MODE>
__global__ __launch_bounds__(256) void k_wg(const float *__restrict__ a, const float *__restrict__ m, float *__restrict__ d, int64_t n, unsigned nrow)
{
    const unsigned tile = blockIdx.x / nrow, row = blockIdx.x % nrow;
    const int64_t off = ((int64_t)tile * 256 + threadIdx.x) * 4;
    V4 mv;
    if (MODE == 0) mv = *reinterpret_cast<const V4 *>(m + off);
    else if (MODE == 2) mv = *reinterpret_cast<const V4 *>(m + (off & 16383));
    else mv = (V4)1.5f;
    const V4 av = ldnt(a + (int64_t)row * n + off);
    stnt(d + (int64_t)row * n + off, av * mv);
}

// column bands: T consecutive tiles of one row, then the same T tiles of the next row, ... then the next band -- inside a row's share of a band
// the workgroups stream linearly like a copy, the band of the model (T x 4 KiB) is reused by every row from L2
__global__ __launch_bounds__(256) void k_wg_band(const float *__restrict__ a, const float *__restrict__ m, float *__restrict__ d, int64_t n, unsigned nrow,
                                                 unsigned T, int nomodel, unsigned G)
{
    const unsigned ngroups = (nrow + G - 1) / G;              // G rows per workgroup, walked in order (the library does this when the grid would pass 2^32 threads)
    const unsigned per_band = ngroups * T;
    const unsigned band = blockIdx.x / per_band, rem = blockIdx.x % per_band;
    const unsigned grp = rem / T, tile = band * T + rem % T;
    const int64_t off = ((int64_t)tile * 256 + threadIdx.x) * 4;
    const V4 mv = nomodel ? (V4)1.5f : *reinterpret_cast<const V4 *>(m + off);
    for (unsigned row = grp * G; row < grp * G + G && row < nrow; row++) {
        const V4 av = ldnt(a + (int64_t)row * n + off);
        stnt(d + (int64_t)row * n + off, av * mv);
    }
}

// persistent: work item w = (tile, row) in the order of k_wg; workgroup b takes w = b, b + grid, ...; ITEMS items in flight per lane
template <int ITEMS>
__global__ __launch_bounds__(256) void k_pers(const float *__restrict__ a, const float *__restrict__ m, float *__restrict__ d, int64_t n, unsigned nrow,
                                              int64_t nwork)
{
    for (int64_t w0 = blockIdx.x; w0 < nwork; w0 += (int64_t)gridDim.x * ITEMS) {
        V4 av[ITEMS], mv[ITEMS];
        int64_t o[ITEMS];
        bool ok[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            const int64_t w = w0 + (int64_t)j * gridDim.x;
            ok[j] = w < nwork;
            const int64_t ww = ok[j] ? w : 0;
            const int64_t tile = ww / nrow, row = ww % nrow;
            const int64_t off = (tile * 256 + threadIdx.x) * 4;
            o[j] = row * n + off;
            mv[j] = *reinterpret_cast<const V4 *>(m + off);
            av[j] = ldnt(a + o[j]);
        }
#pragma unroll
        for (int j = 0; j < ITEMS; j++)
            if (ok[j]) stnt(d + o[j], av[j] * mv[j]);
    }
}

// persistent, rows walked by the workgroup: tiles t = b, b + grid, ...; for each tile the model pack stays in registers, R rows in flight
template <int R>
__global__ __launch_bounds__(256) void k_pers_rows(const float *__restrict__ a, const float *__restrict__ m, float *__restrict__ d, int64_t n,
                                                   unsigned nrow, int64_t ntiles)
{
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t off = (t * 256 + threadIdx.x) * 4;
        const V4 mv = *reinterpret_cast<const V4 *>(m + off);
        for (unsigned i = 0; i < nrow; i += R) {
            V4 av[R];
#pragma unroll
            for (int j = 0; j < R; j++) av[j] = (i + j < nrow) ? ldnt(a + (int64_t)(i + j) * n + off) : (V4)0.f;
#pragma unroll
            for (int j = 0; j < R; j++)
                if (i + j < nrow) stnt(d + (int64_t)(i + j) * n + off, av[j] * mv);
        }
    }
}

int main(int argc, char **argv)
{
    const int64_t nrow = argc > 1 ? atoll(argv[1]) : 128, edge = argc > 2 ? atoll(argv[2]) : 256;
    const int64_t n = edge * edge * edge, npack = nrow * n / 4, ntiles = n / 4 / 256, nwork = ntiles * nrow;
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
    const double bytes = 2.0 * nrow * n * 4;
    unsigned G = 1;
    while ((double)ntiles * ((nrow + G - 1) / G) * 256.0 >= 4294967296.0) G *= 2;     // grid x block < 2^32 threads
    const bool one_row = (G == 1);
    printf("== %lld x %lld^3 Float32 (%.0f GiB per slab); GB/s over the two slabs ==\n", (long long)nrow, (long long)edge, nrow * n * 4.0 / (1 << 30));
    auto timeit = [&](const char *name, auto launch) -> int {
        std::vector<float> ms;
        for (int rep = 0; rep < 7; rep++) {
            CK(hipEventRecord(e0, 0));
            launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t = 0;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (rep >= 2) ms.push_back(t);
        }
        CK(hipGetLastError());
        std::sort(ms.begin(), ms.end());
        printf("%-44s min %8.3f ms  %7.1f GB/s\n", name, ms[0], bytes / ms[0] / 1e6);
        fflush(stdout);
        return 0;
    };
    for (int rnd = 0; rnd < 2; rnd++) {
        if (timeit("copy, grid-stride, 8192 workgroups", [&] { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, A, D, npack); })) return 1;
        if (one_row && timeit("wg per (tile, row)", [&] { hipLaunchKernelGGL(k_wg<0>, dim3((unsigned)nwork), dim3(256), 0, 0, A, M, D, n, (unsigned)nrow); })) return 1;
        if (one_row && timeit("wg per (tile, row), no model read", [&] { hipLaunchKernelGGL(k_wg<1>, dim3((unsigned)nwork), dim3(256), 0, 0, A, M, D, n, (unsigned)nrow); })) return 1;
        if (one_row && timeit("wg per (tile, row), model within 64 KiB", [&] { hipLaunchKernelGGL(k_wg<2>, dim3((unsigned)nwork), dim3(256), 0, 0, A, M, D, n, (unsigned)nrow); })) return 1;
        for (unsigned T : {1u, 4u, 16u, 32u, 64u, 128u, 256u, 1024u, 4096u, 65536u}) {
            if (ntiles % T) continue;
            char nm[96];
            snprintf(nm, sizeof nm, "column bands of %u tiles, %u rows per wg", T, G);
            if (timeit(nm, [&] { hipLaunchKernelGGL(k_wg_band, dim3((unsigned)(ntiles * ((nrow + G - 1) / G))), dim3(256), 0, 0, A, M, D, n, (unsigned)nrow, T, 0, G); })) return 1;
        }
        if (timeit("column bands of 256 tiles, no model read", [&] { hipLaunchKernelGGL(k_wg_band, dim3((unsigned)(ntiles * ((nrow + G - 1) / G))), dim3(256), 0, 0, A, M, D, n, (unsigned)nrow, 256u, 1, G); })) return 1;
        if (timeit("library-style copy: one wg per 16 KiB, linear", [&] { hipLaunchKernelGGL(k_copy, dim3((unsigned)(npack / 1024)), dim3(256), 0, 0, A, D, npack); })) return 1;
        for (int g : {2048, 4096, 8192}) {
            if (!one_row || rnd > 0) break;
            char nm[96];
            snprintf(nm, sizeof nm, "persistent %d wgs, 1 item in flight", g);
            if (timeit(nm, [&] { hipLaunchKernelGGL(k_pers<1>, dim3(g), dim3(256), 0, 0, A, M, D, n, (unsigned)nrow, nwork); })) return 1;
            snprintf(nm, sizeof nm, "persistent %d wgs, 2 items in flight", g);
            if (timeit(nm, [&] { hipLaunchKernelGGL(k_pers<2>, dim3(g), dim3(256), 0, 0, A, M, D, n, (unsigned)nrow, nwork); })) return 1;
            snprintf(nm, sizeof nm, "persistent %d wgs, 4 items in flight", g);
            if (timeit(nm, [&] { hipLaunchKernelGGL(k_pers<4>, dim3(g), dim3(256), 0, 0, A, M, D, n, (unsigned)nrow, nwork); })) return 1;
        }
        for (int g : {2048, 8192}) {
            if (rnd > 0) break;
            char nm[96];
            snprintf(nm, sizeof nm, "persistent %d wgs walking rows, 4 in flight", g);
            if (timeit(nm, [&] { hipLaunchKernelGGL(k_pers_rows<4>, dim3(g), dim3(256), 0, 0, A, M, D, n, (unsigned)nrow, ntiles); })) return 1;
            snprintf(nm, sizeof nm, "persistent %d wgs walking rows, 8 in flight", g);
            if (timeit(nm, [&] { hipLaunchKernelGGL(k_pers_rows<8>, dim3(g), dim3(256), 0, 0, A, M, D, n, (unsigned)nrow, ntiles); })) return 1;
        }
    }
    return 0;
}
