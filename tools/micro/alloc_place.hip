// Experiment (round 3): does HOW the two 64 GiB slabs of the tall forward are ALLOCATED decide its rate?
// d_i = a_i .* m for NROW rows of EDGE^3 Float32, row-concurrent walk (one row per workgroup, every row in flight: the library's
// candidates 6 / 7, fastest in most processes, slow in some) and sequential sweep (16 rows per workgroup, tile fastest: placement-independent).
// Allocation modes: 0 hipMalloc per slab (what the library does) ; 1 hipExtMallocWithFlags(hipDeviceMallocContiguous) per slab ;
// 2 ONE hipMalloc holding both slabs (+ SKEW bytes between them) ; 3 ONE contiguous allocation holding both.
// For every mode the range slab is freed and allocated again REALLOC times: the spread over re-allocations is the placement lottery.
//   hipcc --offload-arch=gfx950 -O3 -o alloc_place alloc_place.hip && ./alloc_place NROW EDGE MODE [REALLOC] [SKEW_BYTES]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef float V __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ inline V ldnt(const float *p) { typedef const V __attribute__((address_space(1))) *gp; return __builtin_nontemporal_load((gp)p); }
__device__ inline V ldc(const float *p) { typedef const V __attribute__((address_space(1))) *gp; return *(gp)p; }
__device__ inline void stnt(float *p, V v) { typedef V __attribute__((address_space(1))) *gp; __builtin_nontemporal_store(v, (gp)p); }

__global__ void k_init(float *p, int64_t n, uint64_t seed)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 31;
        p[i] = (float)(z >> 40) * (1.0f / 16777216.0f);
    }
}

// ORDER 1: row group fastest (the workgroups of one tile of all row groups are dispatched together); ORDER 0: tile fastest
template <int G, int ORDER>
__global__ __launch_bounds__(256) void k_fwd(const float *__restrict__ a, const float *__restrict__ m, float *__restrict__ d, int64_t n, int64_t nrow,
                                             unsigned ntiles, unsigned ngroups)
{
    unsigned tile, grp;
    if (ORDER) { grp = blockIdx.x % ngroups; tile = blockIdx.x / ngroups; }
    else { tile = blockIdx.x % ntiles; grp = blockIdx.x / ntiles; }
    const int64_t s = ((int64_t)tile * 256 + threadIdx.x) * 4;
    if (s >= n) return;
    const V mv = ldc(m + s);
    const int64_t i0 = (int64_t)grp * G, i1 = i0 + G < nrow ? i0 + G : nrow;
#pragma unroll 4
    for (int64_t i = i0; i < i1; i++) stnt(d + i * n + s, ldnt(a + i * n + s) * mv);
}

template <typename F> float best_of(F f, int reps)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    f();
    float best = 1e9f;
    for (int r = 0; r < reps; r++) {
        (void)hipEventRecord(e0);
        f();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

// mode 4: NSLAB slabs carved from ONE contiguous allocation; the forward for every ordered pair (read slab p, write slab q): which pairs are fast?
int pairs(int64_t nrow, int64_t edge, int nslab)
{
    const int64_t n = edge * edge * edge;
    const size_t slab = (size_t)nrow * n * sizeof(float);
    float *base = nullptr, *m = nullptr;
    CK(hipMalloc((void **)&m, n * sizeof(float)));
    CK(hipExtMallocWithFlags((void **)&base, (size_t)nslab * slab, hipDeviceMallocContiguous));
    hipLaunchKernelGGL(k_init, dim3(32768), dim3(256), 0, 0, base, (int64_t)nslab * nrow * n, 1ull);
    hipLaunchKernelGGL(k_init, dim3(4096), dim3(256), 0, 0, m, n, 2ull);
    CK(hipDeviceSynchronize());
    const unsigned ntiles = (unsigned)((n / 4 + 255) / 256), g16 = (unsigned)((nrow + 15) / 16);
    const double bytes = (2.0 * nrow * n + n) * 4;
    printf("# %d slabs of %lld x %lld^3 Float32 in one contiguous allocation at %p; row-concurrent x16 forward, ms for read slab p (row) -> write slab q (column)\n", nslab,
           (long long)nrow, (long long)edge, (void *)base);
    for (int p = 0; p < nslab; p++) {
        printf("p=%d:", p);
        for (int q = 0; q < nslab; q++) {
            if (p == q) { printf("      -       "); continue; }
            const float *a = (const float *)((const char *)base + (size_t)p * slab);
            float *d = (float *)((char *)base + (size_t)q * slab);
            const float t = best_of([&] { hipLaunchKernelGGL((k_fwd<16, 1>), dim3(ntiles * g16), dim3(256), 0, 0, a, m, d, n, nrow, ntiles, g16); }, 3);
            printf("  %6.2f (%4.2f)", t, bytes / t / 1e9);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}

// mode 5: ONE contiguous allocation of TOTAL GiB; the forward over windows of NROW rows: (a) read window 0, write window w for every w;
// (b) write the last window, read window w for every w: a map of where writes / reads are fast
int scan(int64_t nrow, int64_t edge, int total_gib)
{
    const int64_t n = edge * edge * edge;
    const size_t win = (size_t)nrow * n * sizeof(float);
    const int nwin = (int)(((size_t)total_gib << 30) / win);
    float *base = nullptr, *m = nullptr;
    CK(hipMalloc((void **)&m, n * sizeof(float)));
    CK(hipExtMallocWithFlags((void **)&base, (size_t)nwin * win, hipDeviceMallocContiguous));
    hipLaunchKernelGGL(k_init, dim3(32768), dim3(256), 0, 0, base, (int64_t)nwin * nrow * n, 1ull);
    hipLaunchKernelGGL(k_init, dim3(4096), dim3(256), 0, 0, m, n, 2ull);
    CK(hipDeviceSynchronize());
    const unsigned ntiles = (unsigned)((n / 4 + 255) / 256), g16 = (unsigned)((nrow + 15) / 16);
    const double bytes = (2.0 * nrow * n + n) * 4;
    printf("# %d windows of %.1f GiB (%lld rows of %lld^3 Float32) in one contiguous allocation at %p; row-concurrent x16 forward, TB/s\n", nwin, win / 1073741824.0,
           (long long)nrow, (long long)edge, (void *)base);
    auto at = [&](int w) { return (float *)((char *)base + (size_t)w * win); };
    printf("write window w, read window 0 (w = 1..):");
    for (int w = 1; w < nwin; w++) {
        const float t = best_of([&] { hipLaunchKernelGGL((k_fwd<16, 1>), dim3(ntiles * g16), dim3(256), 0, 0, at(0), m, at(w), n, nrow, ntiles, g16); }, 3);
        printf(" %4.2f", bytes / t / 1e9);
    }
    printf("\nread window w, write the LAST window (w = 0..):");
    for (int w = 0; w < nwin - 1; w++) {
        const float t = best_of([&] { hipLaunchKernelGGL((k_fwd<16, 1>), dim3(ntiles * g16), dim3(256), 0, 0, at(w), m, at(nwin - 1), n, nrow, ntiles, g16); }, 3);
        printf(" %4.2f", bytes / t / 1e9);
    }
    printf("\nwrite window w, read window w+1 (neighbours):");
    for (int w = 0; w < nwin - 1; w++) {
        const float t = best_of([&] { hipLaunchKernelGGL((k_fwd<16, 1>), dim3(ntiles * g16), dim3(256), 0, 0, at(w + 1), m, at(w), n, nrow, ntiles, g16); }, 3);
        printf(" %4.2f", bytes / t / 1e9);
    }
    printf("\n");
    return 0;
}

// 2-D grid form (x = row group or tile, y = the other) so that one-row workgroups fit the launch limits; SKEWED: row group g starts its
// tiles at an offset of g * 61 tiles (the row groups in flight then touch different tile offsets at any moment)
template <int G, int ORDER, int BLK, int U, bool SKEWED>
__global__ __launch_bounds__(BLK) void k_fwd2(const float *__restrict__ a, const float *__restrict__ m, float *__restrict__ d, int64_t n, int64_t nrow, unsigned ntiles)
{
    unsigned tile = ORDER ? blockIdx.y : blockIdx.x;
    const unsigned grp = ORDER ? blockIdx.x : blockIdx.y;
    if (SKEWED) tile = (tile + grp * 61u) % ntiles;
    const int64_t i0 = (int64_t)grp * G, i1 = i0 + G < nrow ? i0 + G : nrow;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int64_t s = (((int64_t)tile * U + u) * BLK + threadIdx.x) * 4;
        if (s >= n) continue;
        const V mv = ldc(m + s);
#pragma unroll 4
        for (int64_t i = i0; i < i1; i++) stnt(d + i * n + s, ldnt(a + i * n + s) * mv);
    }
}

// mode 6: which walk is robust?  8 slabs in one contiguous allocation; pairs (0->1) slow, (0->2) medium, (0->7) fast for the x16 row-concurrent walk
template <int G, int ORDER, int BLK, int U, bool SKEWED>
void try_walk(const char *name, const float *base, const float *m, size_t slab, int64_t n, int64_t nrow, double bytes)
{
    const unsigned ntiles = (unsigned)((n / 4 + (int64_t)BLK * U - 1) / ((int64_t)BLK * U)), ng = (unsigned)((nrow + G - 1) / G);
    printf("%-44s", name);
    const int pq[5][2] = {{0, 1}, {0, 2}, {0, 7}, {7, 0}, {2, 3}};
    for (auto &x : pq) {
        const float *a = (const float *)((const char *)base + (size_t)x[0] * slab);
        float *d = (float *)((char *)base + (size_t)x[1] * slab);
        const dim3 grid = ORDER ? dim3(ng, ntiles) : dim3(ntiles, ng);
        const float t = best_of([&] { hipLaunchKernelGGL((k_fwd2<G, ORDER, BLK, U, SKEWED>), grid, dim3(BLK), 0, 0, a, m, d, n, nrow, ntiles); }, 3);
        printf("  %d->%d %5.2f", x[0], x[1], bytes / t / 1e9);
    }
    printf("\n");
    fflush(stdout);
}

int walks(int64_t nrow, int64_t edge)
{
    const int64_t n = edge * edge * edge;
    const size_t slab = (size_t)nrow * n * sizeof(float);
    float *base = nullptr, *m = nullptr;
    CK(hipMalloc((void **)&m, n * sizeof(float)));
    CK(hipExtMallocWithFlags((void **)&base, 8 * slab, hipDeviceMallocContiguous));
    hipLaunchKernelGGL(k_init, dim3(32768), dim3(256), 0, 0, base, 8 * nrow * n, 1ull);
    hipLaunchKernelGGL(k_init, dim3(4096), dim3(256), 0, 0, m, n, 2ull);
    CK(hipDeviceSynchronize());
    const double bytes = (2.0 * nrow * n + n) * 4;
    printf("# 8 slabs of %lld x %lld^3 in one contiguous allocation; TB/s of the forward per walk for (read slab -> write slab)\n", (long long)nrow, (long long)edge);
    try_walk<16, 1, 256, 1, false>("row-concurrent, 16 rows/wg, 256x1", base, m, slab, n, nrow, bytes);
    try_walk<1, 1, 256, 1, false>("row-concurrent, 1 row/wg, 256x1", base, m, slab, n, nrow, bytes);
    try_walk<1, 1, 512, 8, false>("row-concurrent, 1 row/wg, 512x8", base, m, slab, n, nrow, bytes);
    try_walk<2, 1, 256, 1, false>("row-concurrent, 2 rows/wg, 256x1", base, m, slab, n, nrow, bytes);
    try_walk<4, 1, 256, 4, false>("row-concurrent, 4 rows/wg, 256x4", base, m, slab, n, nrow, bytes);
    try_walk<16, 1, 256, 1, true>("row-concurrent skewed, 16 rows/wg, 256x1", base, m, slab, n, nrow, bytes);
    try_walk<1, 1, 256, 1, true>("row-concurrent skewed, 1 row/wg, 256x1", base, m, slab, n, nrow, bytes);
    try_walk<1, 0, 256, 1, false>("sequential, 1 row/wg, 256x1", base, m, slab, n, nrow, bytes);
    try_walk<1, 0, 1024, 1, false>("sequential, 1 row/wg, 1024x1", base, m, slab, n, nrow, bytes);
    try_walk<4, 0, 256, 4, false>("sequential, 4 rows/wg, 256x4", base, m, slab, n, nrow, bytes);
    try_walk<16, 0, 1024, 8, false>("sequential, 16 rows/wg, 1024x8", base, m, slab, n, nrow, bytes);
    try_walk<16, 0, 256, 1, false>("sequential, 16 rows/wg, 256x1", base, m, slab, n, nrow, bytes);
    return 0;
}

int main(int argc, char **argv)
{
    const int64_t nrow = argc > 1 ? atoll(argv[1]) : 1024, edge = argc > 2 ? atoll(argv[2]) : 256;
    if (argc > 3 && atoi(argv[3]) == 6) return walks(nrow, edge);
    if (argc > 3 && atoi(argv[3]) == 5) return scan(nrow, edge, argc > 4 ? atoi(argv[4]) : 256);
    const int mode = argc > 3 ? atoi(argv[3]) : 0, realloc_n = argc > 4 ? atoi(argv[4]) : 3;
    const int64_t skew = argc > 5 ? atoll(argv[5]) : 0;
    if (mode == 4) return pairs(nrow, edge, realloc_n);
    const int64_t n = edge * edge * edge;
    const size_t slab = (size_t)nrow * n * sizeof(float);
    float *a = nullptr, *d = nullptr, *m = nullptr, *both = nullptr;
    CK(hipMalloc((void **)&m, n * sizeof(float)));
    if (mode == 0) CK(hipMalloc((void **)&a, slab));
    else if (mode == 1) CK(hipExtMallocWithFlags((void **)&a, slab, hipDeviceMallocContiguous));
    else if (mode == 2) { CK(hipMalloc((void **)&both, 2 * slab + (size_t)skew)); a = both; }
    else { CK(hipExtMallocWithFlags((void **)&both, 2 * slab + (size_t)skew, hipDeviceMallocContiguous)); a = both; }
    hipLaunchKernelGGL(k_init, dim3(32768), dim3(256), 0, 0, a, nrow * n, 1ull);
    hipLaunchKernelGGL(k_init, dim3(4096), dim3(256), 0, 0, m, n, 2ull);
    CK(hipDeviceSynchronize());
    const unsigned ntiles = (unsigned)((n / 4 + 255) / 256);
    const double bytes = (2.0 * nrow * n + n) * 4;
    printf("# %lld x %lld^3 Float32, mode %d (%s), skew %lld B\n", (long long)nrow, (long long)edge, mode,
           mode == 0 ? "hipMalloc per slab" : mode == 1 ? "contiguous per slab" : mode == 2 ? "one hipMalloc for both slabs" : "one contiguous allocation for both", (long long)skew);
    for (int r = 0; r < realloc_n; r++) {
        if (mode == 0) CK(hipMalloc((void **)&d, slab));
        else if (mode == 1) CK(hipExtMallocWithFlags((void **)&d, slab, hipDeviceMallocContiguous));
        else d = (float *)((char *)both + slab + skew);
        CK(hipMemset(d, 0, slab));
        CK(hipDeviceSynchronize());
        const unsigned g1 = (unsigned)nrow, g16 = (unsigned)((nrow + 15) / 16);
        const float t_conc = best_of([&] { hipLaunchKernelGGL((k_fwd<1, 1>), dim3(ntiles * g1), dim3(256), 0, 0, a, m, d, n, nrow, ntiles, g1); }, 4);
        const float t_seq = best_of([&] { hipLaunchKernelGGL((k_fwd<16, 0>), dim3(ntiles * g16), dim3(256), 0, 0, a, m, d, n, nrow, ntiles, g16); }, 4);
        const float t_c16 = best_of([&] { hipLaunchKernelGGL((k_fwd<16, 1>), dim3(ntiles * g16), dim3(256), 0, 0, a, m, d, n, nrow, ntiles, g16); }, 4);
        printf("allocation %d of d at %p: row-concurrent x1 %7.3f ms %7.1f GB/s | sequential x16 %7.3f ms %7.1f GB/s | row-concurrent x16 %7.3f ms %7.1f GB/s\n", r,
               (void *)d, t_conc, bytes / t_conc / 1e6, t_seq, bytes / t_seq / 1e6, t_c16, bytes / t_c16 / 1e6);
        fflush(stdout);
        if (mode <= 1) CK(hipFree(d));
        else break;                                                     // one allocation: nothing to re-allocate
    }
    return 0;
}
