// Round 6 microbenchmark: the adjoint-shaped walk over rows OFF the 16-byte grid (n % 4 != 0 Float32 elements per row, rows back to back in one slab).
// A lane owns U packs of the DOMAIN (16-byte grid of the output) and walks all rows, acc += a_i[s] * d_i[s]: both read streams of row i start
// phi_i = (i * n) mod 4 scalars off the grid.
//   variant 0  what the tall kernels do today: one under-aligned global_load_dwordx4 per pack (a wave's 1 KiB request straddles nine 128-byte lines)
//   variant 1  ALIGNED loads in the row's own frame + a one-lane funnel shift (DPP wave_shl:1): lane l loads the aligned pack that holds its first scalar,
//              takes the first phi scalars of lane l + 1's pack, lane 63 loads one extra aligned pack
//   variant 2  the same through ds_bpermute (__shfl_down) instead of DPP
// Prints GB/s over the algorithmic bytes (2 N n 4) and a checksum (the variants must agree bit for bit).
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/micro/adj_shift.hip -o /tmp/adj_shift && /tmp/adj_shift [nrow] [n]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));          \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

typedef float V __attribute__((ext_vector_type(4)));
typedef V __attribute__((aligned(4))) UV;

template <bool NT> __device__ inline V ldu(const float *p)
{
    typedef const UV __attribute__((address_space(1))) *gp;
    if (NT) return __builtin_nontemporal_load((gp)p);
    return *(gp)p;
}
template <bool NT> __device__ inline V lda(const float *p)
{
    typedef const V __attribute__((address_space(1))) *gp;
    if (NT) return __builtin_nontemporal_load((gp)p);
    return *(gp)p;
}

template <int VAR> __device__ inline float from_next_lane(float x)
{
    if (VAR == 1) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
    return __shfl_down(x, 1);
}

// the pack of four scalars that starts at p (4-byte aligned, phi = scalars past the 16-byte boundary below it), from aligned loads
template <int VAR, bool NT> __device__ inline V ld_shift(const float *p, unsigned phi, bool last_lane)
{
    const float *al = p - phi;
    V l = lda<NT>(al);
    if (phi == 0) return l;                                           // wave-uniform: phi is a property of the row
    V n;
    n.x = from_next_lane<VAR>(l.x);
    n.y = from_next_lane<VAR>(l.y);
    n.z = from_next_lane<VAR>(l.z);
    if (last_lane) {                                                  // lane 63: its neighbour is the next wave's lane 0
        const V e = lda<NT>(al + 4);
        n.x = e.x; n.y = e.y; n.z = e.z;
    }
    if (phi == 1) return V{l.y, l.z, l.w, n.x};
    if (phi == 2) return V{l.z, l.w, n.x, n.y};
    return V{l.w, n.x, n.y, n.z};
}

template <int VAR, bool NT, int BLK, int U, int DEPTH>
__global__ __launch_bounds__(BLK) void k_adj(const float *__restrict__ a, const float *__restrict__ d, float *__restrict__ out, int64_t nrow, int64_t n)
{
    const int64_t s0 = ((int64_t)blockIdx.x * U * BLK + threadIdx.x) * 4;
    V acc[U];
    int64_t sk[U];
    bool ok[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        const int64_t s = s0 + (int64_t)k * BLK * 4;
        ok[k] = s + 4 <= n;                                           // (the partial last pack is left out of this experiment)
        sk[k] = ok[k] ? s : 0;
        acc[k] = V{0, 0, 0, 0};
    }
    const bool last_lane = (threadIdx.x & 63) == 63;
    const unsigned base_phi = (unsigned)(((uintptr_t)a >> 2) & 3);   // (a and d are allocated alike)
    for (int64_t i = 0; i + DEPTH <= nrow; i += DEPTH) {
        V av[DEPTH][U], dv[DEPTH][U];
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            const float *ar = a + (i + j) * n, *dr = d + (i + j) * n;
            const unsigned phi = (unsigned)(((i + j) * n + base_phi) & 3);
#pragma unroll
            for (int k = 0; k < U; k++) {
                if (VAR == 0) {
                    av[j][k] = ldu<NT>(ar + sk[k]);
                    dv[j][k] = ldu<NT>(dr + sk[k]);
                } else {
                    av[j][k] = ld_shift<VAR, NT>(ar + sk[k], phi, last_lane);
                    dv[j][k] = ld_shift<VAR, NT>(dr + sk[k], phi, last_lane);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < DEPTH; j++)
#pragma unroll
            for (int k = 0; k < U; k++) acc[k] = acc[k] + av[j][k] * dv[j][k];
    }
#pragma unroll
    for (int k = 0; k < U; k++)
        if (ok[k]) *(V *)(out + s0 + (int64_t)k * BLK * 4) = acc[k];
}

template <int VAR, bool NT> float run(const float *a, const float *d, float *out, int64_t nrow, int64_t n, int reps)
{
    constexpr int BLK = 512, U = 2, DEPTH = 2;
    const int64_t packs = n / 4;
    const unsigned gx = (unsigned)((packs + (int64_t)U * BLK - 1) / ((int64_t)U * BLK));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps + 1; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_adj<VAR, NT, BLK, U, DEPTH>), dim3(gx), dim3(BLK), 0, 0, a, d, out, nrow, n);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    return best;
}

__global__ void k_init(float *p, int64_t n, uint32_t seed)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
    }
}

int main(int argc, char **argv)
{
    const int64_t nrow = argc > 1 ? atoll(argv[1]) : 256;
    const int64_t n = argc > 2 ? atoll(argv[2]) : (int64_t)255 * 255 * 255;
    const int64_t total = nrow * n;
    float *a, *d, *out;
    CK(hipMalloc(&a, (total + 64) * sizeof(float)));
    CK(hipMalloc(&d, (total + 64) * sizeof(float)));
    CK(hipMalloc(&out, (n + 64) * sizeof(float)));
    k_init<<<65536, 256>>>(a, total + 64, 1u);
    k_init<<<65536, 256>>>(d, total + 64, 2u);
    CK(hipDeviceSynchronize());
    const double bytes = 2.0 * (double)total * 4.0;
    std::vector<float> h0((size_t)n), h((size_t)n);
    const char *names[3] = {"under-aligned loads (today)", "aligned loads + DPP wave_shl:1", "aligned loads + ds_bpermute"};
    printf("# %lld rows of %lld Float32 (%s the 16-byte grid), %.2f GiB per stream; 512 x 2 packs x 2 rows in flight\n", (long long)nrow, (long long)n,
           n % 4 ? "OFF" : "on", (double)total * 4 / (1 << 30));
    for (int nt = 1; nt >= 0; nt--)
        for (int var = 0; var < 3; var++) {
            CK(hipMemset(out, 0, n * sizeof(float)));
            float ms;
            if (nt) ms = var == 0 ? run<0, true>(a, d, out, nrow, n, 5) : var == 1 ? run<1, true>(a, d, out, nrow, n, 5) : run<2, true>(a, d, out, nrow, n, 5);
            else ms = var == 0 ? run<0, false>(a, d, out, nrow, n, 5) : var == 1 ? run<1, false>(a, d, out, nrow, n, 5) : run<2, false>(a, d, out, nrow, n, 5);
            CK(hipMemcpy(h.data(), out, n * sizeof(float), hipMemcpyDeviceToHost));
            if (nt == 1 && var == 0) h0 = h;
            size_t bad = 0;
            int64_t first = -1;
            for (int64_t i = 0; i < n / 4 * 4; i++)
                if (h[i] != h0[i]) { if (!bad) first = i; bad++; }
            printf("%-12s %-34s %8.3f ms  %7.1f GB/s   ", nt ? "nontemporal" : "temporal", names[var], ms, bytes / ms / 1e6);
            if (bad) printf("MISMATCH at %zu scalars, first %lld (lane %lld of its wave)\n", bad, (long long)first, (long long)((first / 4) % 64));
            else printf("same bits\n");
        }
    return 0;
}
