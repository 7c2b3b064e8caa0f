// Micro-benchmark (round 3): the FORWARD of an M x K grid of plain diagonal blocks, d_i = d_i + sum_j a_ij .* m_j (src/Jets.jl:1020-1024),
// under different workgroup shapes -- which tiling streams it fastest on MI355X?  Float32, coefficients in one row-major slab.
//   T<R,QQ,U>  register-tiled (the library's k_grid_tile): a workgroup owns R lines x U packs per lane; per summed index the input pack
//              is loaded once for the R lines; QQ steps' loads in flight.  R = 1 is k_grid_diag's shape.
//   H<KH,G,D>  input-holding (the tall forward's shape): a workgroup keeps the K input packs of its tile in registers and streams G block
//              rows, D rows' loads in flight; needs K == KH.
// Every variant adds each output's products in j order, product rounded before the add: a 64-bit sum of the result's bit patterns must agree.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o grid_tile grid_tile.hip && ./grid_tile M K EDGE
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
typedef float V __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool NT> __device__ inline V ld(const float *p)
{
    typedef const V __attribute__((address_space(1))) *gp;
    if (NT) return __builtin_nontemporal_load((gp)p);
    return *(gp)p;
}
__device__ inline void stnt(float *p, V v)
{
    typedef V __attribute__((address_space(1))) *gp;
    __builtin_nontemporal_store(v, (gp)p);
}

__global__ void k_init(float *p, int64_t n, uint64_t seed)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        p[i] = (float)(z >> 40) * (1.0f / 16777216.0f);
    }
}
__global__ void k_bitsum(const uint32_t *p, int64_t n, unsigned long long *out)
{
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) acc += p[i] * (unsigned long long)(i % 1021 + 1);
    atomicAdd(out, acc);
}

// (group, tile) decode: XCD = true: the groups of one tile get ids 8 apart (same XCD, dispatched together); false: tile fastest
template <bool XCD> __device__ inline void decode(unsigned ngroups, unsigned ntiles, int64_t &grp, int64_t &tile)
{
    if (!XCD) {
        grp = blockIdx.x / ntiles;
        tile = blockIdx.x - (unsigned)grp * ntiles;
        return;
    }
    const unsigned per = 8u * ngroups;
    const unsigned g = blockIdx.x / per, rem = blockIdx.x - g * per;
    grp = rem >> 3;
    tile = (int64_t)g * 8 + (rem & 7u);
}

template <int R, int QQ, int U, bool XCD>
__global__ __launch_bounds__(256) void k_tile(const float *__restrict__ coeff, int64_t M, int64_t K, int64_t n, const float *__restrict__ in, float *__restrict__ out,
                                              unsigned ntiles, unsigned ngroups, int64_t bs)
{
    int64_t grp, tile;
    decode<XCD>(ngroups, ntiles, grp, tile);
    if (tile >= ntiles) return;
    int64_t line[R];
#pragma unroll
    for (int r = 0; r < R; r++) line[r] = grp * R + r < M ? grp * R + r : M - 1;
    int64_t s[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        s[u] = ((tile * U + u) * 256 + threadIdx.x) * 4;
        if (s[u] >= n) s[u] = n - 4;                                   // clamped: redundant, harmless
    }
    V acc[R][U];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int u = 0; u < U; u++) acc[r][u] = ld<true>(out + line[r] * bs + s[u]);
    int64_t q0 = 0;
    for (; q0 + QQ <= K; q0 += QQ) {
        V x[QQ][U], c[QQ][R][U];
#pragma unroll
        for (int q = 0; q < QQ; q++)
#pragma unroll
            for (int u = 0; u < U; u++) {
                x[q][u] = ld<false>(in + (q0 + q) * bs + s[u]);
#pragma unroll
                for (int r = 0; r < R; r++) c[q][r][u] = ld<true>(coeff + (line[r] * K + q0 + q) * bs + s[u]);
            }
#pragma unroll
        for (int q = 0; q < QQ; q++)
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int u = 0; u < U; u++) acc[r][u] = acc[r][u] + c[q][r][u] * x[q][u];
    }
    for (int64_t q = q0; q < K; q++) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const V x = ld<false>(in + q * bs + s[u]);
            V c[R];
#pragma unroll
            for (int r = 0; r < R; r++) c[r] = ld<true>(coeff + (line[r] * K + q) * bs + s[u]);
#pragma unroll
            for (int r = 0; r < R; r++) acc[r][u] = acc[r][u] + c[r] * x;
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int u = 0; u < U; u++)
            if (grp * R + r < M) stnt(out + line[r] * bs + s[u], acc[r][u]);
}

// input-holding: K == KH input packs in registers, G rows per workgroup, D rows in flight
template <int KH, int G, int D, int BLK, bool XCD>
__global__ __launch_bounds__(BLK) void k_hold(const float *__restrict__ coeff, int64_t M, int64_t K, int64_t n, const float *__restrict__ in, float *__restrict__ out,
                                              unsigned ntiles, unsigned ngroups, int64_t bs)
{
    int64_t grp, tile;
    decode<XCD>(ngroups, ntiles, grp, tile);
    if (tile >= ntiles) return;
    int64_t s = (tile * BLK + threadIdx.x) * 4;
    if (s >= n) s = n - 4;
    V x[KH];
#pragma unroll
    for (int j = 0; j < KH; j++) x[j] = ld<false>(in + j * bs + s);
    const int64_t r0 = grp * G, r1 = r0 + G < M ? r0 + G : M;
    for (int64_t i0 = r0; i0 < r1; i0 += D) {
        V acc[D], c[D][KH];
#pragma unroll
        for (int dd = 0; dd < D; dd++) {
            const int64_t i = i0 + dd < r1 ? i0 + dd : r1 - 1;
            acc[dd] = ld<true>(out + i * bs + s);
#pragma unroll
            for (int j = 0; j < KH; j++) c[dd][j] = ld<true>(coeff + (i * K + j) * bs + s);
        }
#pragma unroll
        for (int dd = 0; dd < D; dd++) {
#pragma unroll
            for (int j = 0; j < KH; j++) acc[dd] = acc[dd] + c[dd][j] * x[j];
            if (i0 + dd < r1) stnt(out + (i0 + dd) * bs + s, acc[dd]);
        }
    }
}

struct Ctx { float *coeff, *in, *out, *out0; int64_t M, K, n, bs; unsigned long long *sum; int quick; };

template <typename F> float timeit(F f, int reps = 6)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    f();
    float best = 1e9f;
    for (int i = 0; i < reps; i++) {
        hipEventRecord(e0);
        f();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

template <typename F> void report(const char *name, Ctx &c, F f)
{
    hipMemcpy(c.out, c.out0, sizeof(float) * c.M * c.bs, hipMemcpyDeviceToDevice);
    f();
    hipMemset(c.sum, 0, 8);
    hipLaunchKernelGGL(k_bitsum, dim3(4096), dim3(256), 0, 0, (const uint32_t *)c.out, c.M * c.bs, c.sum);
    unsigned long long h = 0;
    hipMemcpy(&h, c.sum, 8, hipMemcpyDeviceToHost);
    const float ms = timeit(f);
    const double bytes = ((double)c.M * c.K + c.K + 2.0 * c.M) * c.n * 4;
    printf("%-34s %8.3f ms %8.1f GB/s  bits %016llx\n", name, ms, bytes / ms / 1e6, h);
    fflush(stdout);
}

template <int R, int QQ, int U, bool XCD> void run_tile(Ctx &c)
{
    const unsigned ngroups = (unsigned)((c.M + R - 1) / R);
    const int64_t want = (c.n / 4 + 256 * U - 1) / (256 * U);
    const unsigned ntiles = (unsigned)want, grid = XCD ? (unsigned)(((want + 7) / 8) * 8 * ngroups) : (unsigned)(want * ngroups);
    char name[96];
    snprintf(name, sizeof(name), "tile R=%d QQ=%d U=%d %s", R, QQ, U, XCD ? "xcd" : "seq");
    report(name, c, [&] { hipLaunchKernelGGL((k_tile<R, QQ, U, XCD>), dim3(grid), dim3(256), 0, 0, c.coeff, c.M, c.K, c.n, c.in, c.out, ntiles, ngroups, c.bs); });
}

template <int KH, int G, int D, int BLK, bool XCD> void run_hold(Ctx &c)
{
    if (c.K != KH) return;
    const unsigned ngroups = (unsigned)((c.M + G - 1) / G);
    const int64_t want = (c.n / 4 + BLK - 1) / BLK;
    const unsigned ntiles = (unsigned)want, grid = XCD ? (unsigned)(((want + 7) / 8) * 8 * ngroups) : (unsigned)(want * ngroups);
    char name[96];
    snprintf(name, sizeof(name), "hold K=%d G=%d D=%d wg=%d %s", KH, G, D, BLK, XCD ? "xcd" : "seq");
    report(name, c, [&] { hipLaunchKernelGGL((k_hold<KH, G, D, BLK, XCD>), dim3(grid), dim3(BLK), 0, 0, c.coeff, c.M, c.K, c.n, c.in, c.out, ntiles, ngroups, c.bs); });
}

int main(int argc, char **argv)
{
    Ctx c;
    c.M = argc > 1 ? atoll(argv[1]) : 64;
    c.K = argc > 2 ? atoll(argv[2]) : 4;
    const int64_t edge = argc > 3 ? atoll(argv[3]) : 256;
    c.n = edge * edge * edge;
    c.bs = c.n + (argc > 4 ? atoll(argv[4]) : 0);                     // block stride: pad elements between consecutive blocks (coefficients and vectors)
    c.quick = argc > 5 ? atoi(argv[5]) : 0;
    CK(hipMalloc((void **)&c.coeff, sizeof(float) * c.M * c.K * c.bs));
    CK(hipMalloc((void **)&c.in, sizeof(float) * c.K * c.bs));
    CK(hipMalloc((void **)&c.out, sizeof(float) * c.M * c.bs));
    CK(hipMalloc((void **)&c.out0, sizeof(float) * c.M * c.bs));
    CK(hipMalloc((void **)&c.sum, 8));
    hipLaunchKernelGGL(k_init, dim3(16384), dim3(256), 0, 0, c.coeff, c.M * c.K * c.bs, 1ull);
    hipLaunchKernelGGL(k_init, dim3(16384), dim3(256), 0, 0, c.in, c.K * c.bs, 2ull);
    hipLaunchKernelGGL(k_init, dim3(16384), dim3(256), 0, 0, c.out0, c.M * c.bs, 3ull);
    CK(hipDeviceSynchronize());
    printf("# %lld x %lld grid of %lld^3 Float32 diagonal blocks, forward; block stride n + %lld elements; unique bytes %.2f GB\n", (long long)c.M, (long long)c.K, (long long)edge, (long long)(c.bs - c.n),
           ((double)c.M * c.K + c.K + 2.0 * c.M) * c.n * 4 / 1e9);
    if (c.quick) {
        for (int pass = 0; pass < 2; pass++) {
            run_tile<1, 4, 1, true>(c);
            run_tile<2, 2, 1, true>(c);
            run_tile<2, 4, 2, true>(c);
            run_tile<4, 2, 2, true>(c);
            run_tile<2, 4, 1, false>(c);
            run_hold<4, 4, 4, 256, true>(c);
            run_hold<4, 16, 4, 256, false>(c);
        }
        return 0;
    }
    for (int pass = 0; pass < 2; pass++) {
        run_tile<1, 4, 1, true>(c);
        run_tile<1, 8, 1, true>(c);
        run_tile<2, 8, 1, true>(c);
        run_tile<2, 4, 1, true>(c);
        run_tile<2, 2, 1, true>(c);
        run_tile<2, 1, 1, true>(c);
        run_tile<2, 4, 2, true>(c);
        run_tile<2, 2, 2, true>(c);
        run_tile<2, 4, 1, false>(c);
        run_tile<3, 4, 1, true>(c);
        run_tile<4, 4, 1, true>(c);
        run_tile<4, 2, 1, true>(c);
        run_tile<4, 2, 2, true>(c);
        run_tile<8, 2, 1, true>(c);
        run_tile<8, 1, 1, true>(c);
        run_hold<4, 4, 4, 256, true>(c);
        run_hold<4, 8, 4, 256, true>(c);
        run_hold<4, 16, 4, 256, false>(c);
        run_hold<4, 16, 4, 1024, false>(c);
        run_hold<4, 64, 4, 256, false>(c);
        run_hold<4, 16, 2, 256, false>(c);
        run_hold<4, 8, 2, 256, true>(c);
    }
    return 0;
}
