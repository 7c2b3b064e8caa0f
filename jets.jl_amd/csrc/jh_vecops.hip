// jh_vecops.hip -- flat slab kernels behind the BlockArray API: fill / rand / broadcast (lincomb,
// hadamard) / dot / norm / extrema.  Because a device BlockArray is ONE contiguous slab
// (src/Jets.jl:742-748 layout), every whole-vector op is a flat, fully coalesced stream:
// 16 B per lane, grid-stride, <= 2048 workgroups.  Reductions: fp64 per-thread accumulators ->
// wave64 shuffle -> LDS across the 4 waves -> one partial per workgroup -> a single-workgroup
// second kernel that folds the partials in index order (deterministic, no float atomics).
#include "jh_internal.h"
#include <cmath>

namespace {

constexpr int WG = 256;

template <typename S, int NS>
struct alignas(sizeof(S) * NS) Pack {
    S v[NS];
};

// streaming (nontemporal) pack load / store: these kernels touch every byte once
template <typename S, int NS> __device__ inline Pack<S, NS> ldnt(const Pack<S, NS> *p)
{
    Pack<S, NS> o;
    if constexpr (NS == 1) {
        o.v[0] = __builtin_nontemporal_load(reinterpret_cast<const S *>(p));
    } else {
        // (round 5, session 3) the pack is addressed through a type aligned like its SCALAR: the same global_load_dwordx4, which gfx950 executes at any
        // dword-aligned address -- a view that starts off a 16-byte boundary (block 1 of a vector of 255^3-element blocks) keeps the 16-byte-per-lane kernels
        typedef S V __attribute__((ext_vector_type(NS)));
        typedef V __attribute__((aligned(alignof(S)))) UV;
        V v = __builtin_nontemporal_load(reinterpret_cast<const UV *>(p));
        __builtin_memcpy(&o, &v, sizeof(o));
    }
    return o;
}
template <typename S, int NS> __device__ inline void stnt(Pack<S, NS> *p, const Pack<S, NS> &o)
{
    if constexpr (NS == 1) {
        __builtin_nontemporal_store(o.v[0], reinterpret_cast<S *>(p));
    } else {
        typedef S V __attribute__((ext_vector_type(NS)));
        typedef V __attribute__((aligned(alignof(S)))) UV;
        V v;
        __builtin_memcpy(&v, &o, sizeof(o));
        __builtin_nontemporal_store(v, reinterpret_cast<UV *>(p));
    }
}

__host__ __device__ inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
constexpr uint64_t GOLDEN = 0x9E3779B97F4A7C15ULL;

template <typename S> __device__ inline S u01_from(uint64_t h);
template <> __device__ inline double u01_from<double>(uint64_t h) { return (double)(h >> 11) * 0x1.0p-53; }


// Elementwise kernels: ONE pack per thread and as many workgroups as that takes (capped at 2^23 workgroups =
// 2^31 threads, HIP's grid x block limit is 2^32; beyond that the grid-stride loop runs a second pass).
// Measured on MI355X (tools/micro/stream.hip, 16 GiB): one-shot fill 6.7 / copy 6.6 / triad 6.5 TB/s against
// 5.2-5.4 TB/s for a 2048-workgroup persistent grid-stride loop.  Reductions keep the persistent grid (few partials).
inline int grid_full(int64_t packs)
{
    int64_t g = (packs + WG - 1) / WG;
    if (g < 1) g = 1;
    if (g > ((int64_t)1 << 23)) g = (int64_t)1 << 23;
    return (int)g;
}

// number of leading scalars to peel so that p + head is 16-byte aligned (NS scalars per pack)
template <typename S>
inline int64_t head_scalars(const void *p, int64_t n)
{
    uintptr_t a = (uintptr_t)p;
    uintptr_t mis = a & 15u;
    if (mis == 0) return 0;
    int64_t h = (int64_t)((16 - mis) / sizeof(S));
    return h < n ? h : n;
}

// ---------------------------------------------------------------- fill ------------------------
// scalar lane k of the range gets (k & 1) ? im : re for complex, re for real (period = PER scalars)
template <typename S, int NS, int PER>
__global__ void k_fill(S *__restrict__ p, int64_t n, int64_t head, S re, S im)
{
    const int64_t nvec = (n - head) / NS;
    const int64_t tail0 = head + nvec * NS;
    const int64_t tid = (int64_t)blockIdx.x * WG + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * WG;
    Pack<S, NS> val;
#pragma unroll
    for (int c = 0; c < NS; c++) val.v[c] = (PER == 2 && ((head + c) & 1)) ? im : re;
    Pack<S, NS> *pv = reinterpret_cast<Pack<S, NS> *>(p + head);
    for (int64_t v = tid; v < nvec; v += stride) stnt(pv + v, val);
    if (tid < head) p[tid] = (PER == 2 && (tid & 1)) ? im : re;
    if (tid < n - tail0) p[tail0 + tid] = (PER == 2 && ((tail0 + tid) & 1)) ? im : re;
}

template <typename S, int PER>
int fill_scalars(S *p, int64_t n, double re, double im)
{
    if (n <= 0) return JH_OK;
    constexpr int NS = 16 / sizeof(S);
    int64_t head = head_scalars<S>(p, n);
    int64_t nvec = (n - head) / NS;
    hipLaunchKernelGGL((k_fill<S, NS, PER>), dim3(grid_full(nvec > 0 ? nvec : 1)), dim3(WG), 0, jh_ctx().stream, p, n,
                       head, (S)re, (S)im);
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

// ---------------------------------------------------------------- uniform random --------------
// Counter-based generator: a pure function of (seed, stream, scalar lane index), so the device fill and a CPU
// regeneration of any slice agree bit for bit.  hash(j) = mix64(key + (j + 1) * GOLDEN), key = mix64(seed * GOLDEN + stream).
//   Float64 lane k:  the top 53 bits of hash(k).
//   Float32 lane k (round 4):  24 bits of hash(k >> 1) -- the top 24 for an even lane, the next 24 (bits 39..16) for an odd one: ONE
//   hash per two values.  The kernel is bound by its 64-bit integer multiplies (two per hash, four quarter-rate 32-bit multiplies each),
//   not by its store: with a hash per value rand(R) ran at 4.8 TB/s (60 % of the roofline, profiles/bench_suite_r03.txt).
__device__ inline float u01_pair(uint64_t h, int odd) { return (float)(odd ? ((h >> 16) & 0xFFFFFFull) : (h >> 40)) * 0x1.0p-24f; }

// zbase = key + (j0 + 1) * GOLDEN for the hash index j0 of PACK 0's first hash (the host forms it: one 64-bit multiply per launch); pack v's
// first hash index is j0 + 2 v for both element widths (a Float32 pack spans two pairs, a Float64 pack two lanes), so its counter is
// zbase + v * (2 GOLDEN): a workgroup-uniform product (scalar unit) plus thread-index-times-constant -- not a 64-bit multiply per pack
template <typename S, int NS>
__global__ void k_uniform(S *__restrict__ p, int64_t n, uint64_t key, int64_t lane_base, uint64_t zbase, int odd)
{
    const int64_t nvec = n / NS;
    const int64_t tid = (int64_t)blockIdx.x * WG + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * WG;
    constexpr uint64_t G2 = 2 * GOLDEN;
    Pack<S, NS> *pv = reinterpret_cast<Pack<S, NS> *>(p);
    uint64_t z0 = zbase + (uint64_t)blockIdx.x * ((uint64_t)WG * G2) + (uint64_t)(unsigned)threadIdx.x * G2;
    const uint64_t zstep = (uint64_t)stride * G2;
    for (int64_t v = tid; v < nvec; v += stride, z0 += zstep) {
        Pack<S, NS> o;
        if constexpr (sizeof(S) == 4) {
            // lanes k0 .. k0 + 3 (k0 = lane_base + 4 v): pairs (k0 >> 1) and the next one -- and a third when k0 is odd
            const uint64_t h0 = mix64(z0), h1 = mix64(z0 + GOLDEN);
            if (!odd) {
                o.v[0] = u01_pair(h0, 0); o.v[1] = u01_pair(h0, 1); o.v[2] = u01_pair(h1, 0); o.v[3] = u01_pair(h1, 1);
            } else {
                const uint64_t h2 = mix64(z0 + 2 * GOLDEN);
                o.v[0] = u01_pair(h0, 1); o.v[1] = u01_pair(h1, 0); o.v[2] = u01_pair(h1, 1); o.v[3] = u01_pair(h2, 0);
            }
        } else {
#pragma unroll
            for (int c = 0; c < NS; c++) o.v[c] = u01_from<S>(mix64(z0 + (uint64_t)c * GOLDEN));
        }
        stnt(pv + v, o);
    }
    const int64_t tail0 = nvec * NS;
    if (tid < n - tail0) {
        const int64_t k = lane_base + tail0 + tid;
        if constexpr (sizeof(S) == 4) p[tail0 + tid] = u01_pair(mix64(key + (uint64_t)((k >> 1) + 1) * GOLDEN), (int)(k & 1));
        else p[tail0 + tid] = u01_from<S>(mix64(key + (uint64_t)(k + 1) * GOLDEN));
    }
}

// standard normal by Box-Muller from two counter-RNG draws per scalar lane: lane k uses draws 2k and 2k+1.
// u1 in (0,1] so log() is finite.  Complex lanes are scaled by 1/sqrt(2) (Julia: randn(ComplexF64) has unit variance).
template <typename S>
__global__ void k_normal(S *__restrict__ p, int64_t n, uint64_t key, int64_t lane_base, S scale)
{
    const int64_t tid = (int64_t)blockIdx.x * WG + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * WG;
    for (int64_t k = tid; k < n; k += stride) {
        const uint64_t h1 = mix64(key + (uint64_t)(2 * (lane_base + k) + 1) * GOLDEN);
        const uint64_t h2 = mix64(key + (uint64_t)(2 * (lane_base + k) + 2) * GOLDEN);
        const double u1 = ((double)(h1 >> 11) + 1.0) * 0x1.0p-53;
        const double u2 = (double)(h2 >> 11) * 0x1.0p-53;
        p[k] = (S)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2)) * scale;
    }
}

// |x| of a complex (or real) vector into a real vector: abs.(x) (test/runtests.jl:545-547)
template <typename S, int E>
__global__ void k_abs(S *__restrict__ dst, const S *__restrict__ x, int64_t n_elems)
{
    const int64_t tid = (int64_t)blockIdx.x * WG + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * WG;
    for (int64_t k = tid; k < n_elems; k += stride) {
        if (E == 1) dst[k] = x[k] < 0 ? -x[k] : x[k];
        else dst[k] = (S)hypot((double)x[2 * k], (double)x[2 * k + 1]);
    }
}

// bitwise copy of a slab (y .= x), 16 B per lane, four packs in flight, nontemporal both ways
__global__ void k_copy16(uint4 *__restrict__ dst, const uint4 *__restrict__ src, int64_t nvec)
{
    typedef unsigned V __attribute__((ext_vector_type(4)));
    const int64_t tid = (int64_t)blockIdx.x * WG + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * WG;
    int64_t v = tid;
    for (; v + 3 * stride < nvec; v += 4 * stride) {
        V t0 = __builtin_nontemporal_load(reinterpret_cast<const V *>(src) + v);
        V t1 = __builtin_nontemporal_load(reinterpret_cast<const V *>(src) + v + stride);
        V t2 = __builtin_nontemporal_load(reinterpret_cast<const V *>(src) + v + 2 * stride);
        V t3 = __builtin_nontemporal_load(reinterpret_cast<const V *>(src) + v + 3 * stride);
        __builtin_nontemporal_store(t0, reinterpret_cast<V *>(dst) + v);
        __builtin_nontemporal_store(t1, reinterpret_cast<V *>(dst) + v + stride);
        __builtin_nontemporal_store(t2, reinterpret_cast<V *>(dst) + v + 2 * stride);
        __builtin_nontemporal_store(t3, reinterpret_cast<V *>(dst) + v + 3 * stride);
    }
    for (; v < nvec; v += stride)
        __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const V *>(src) + v), reinterpret_cast<V *>(dst) + v);
}

// ---------------------------------------------------------------- lincomb ---------------------
constexpr int MAX_TERMS = 8;
struct LincombArgs {
    const void *x[MAX_TERMS];
    double cre[MAX_TERMS], cim[MAX_TERMS];
    unsigned char flags[MAX_TERMS];               // JH_SCALAR_* per coefficient, normalised by lincomb_args_flags
    int k;
};

// the coefficients' types as the kernels read them: COMPLEX when flagged or when the imaginary part is non-zero (an untyped (re, 0) pair
// stands for a Real), WIDE only against 32-bit elements.  Returns whether any coefficient is wide.
static bool lincomb_args_flags(LincombArgs &a, const int32_t *flags, int dtype)
{
    bool any_wide = false;
    for (int j = 0; j < a.k; j++) {
        const int f = flags ? flags[j] : 0;
        a.flags[j] = (unsigned char)((f & JH_SCALAR_COMPLEX) | (a.cim[j] != 0.0 ? JH_SCALAR_COMPLEX : 0));
        if ((f & JH_SCALAR_WIDE) && (dtype == JH_F32 || dtype == JH_C32)) { a.flags[j] |= JH_SCALAR_WIDE; any_wide = true; }
    }
    return any_wide;
}

// dst = c0*x0 + c1*x1 + ... left to right, every product and sum rounded in S (no contraction:
// the translation unit is built with -ffp-contract=off).  E = scalars per element (1 real, 2 complex).
// A REAL coefficient (no JH_SCALAR_COMPLEX) multiplies part by part as Julia's `a::Real * z::Complex` does, so 1.0 * (x + Inf i) keeps
// x where the four-multiplication formula would make it NaN (0 * Inf), and -0.0 parts keep their sign.
// WIDE (S = float, some coefficient is Float64-based): Julia's promotion -- that coefficient's product is a Float64 product, a sum with a
// Float64 operand is a Float64 sum, and the Float32 element is ONE rounding of the final value (`x .= a .* u .+ b .* v` with a::Float64).
// GET(j, part) is operand j's real (0) / imaginary (1) part of the element at hand.
template <typename S, int E, bool WIDE, typename GET>
__device__ inline void lincomb_one(const LincombArgs &a, GET get, S &outr, S &outi)
{
    if constexpr (!WIDE) {
        S accr = 0, acci = 0;
#pragma unroll
        for (int j = 0; j < MAX_TERMS; j++) {
            if (j < a.k) {
                S tr, ti = 0;
                if (E == 1) {
                    tr = (S)a.cre[j] * get(j, 0);
                } else {
                    const S cr = (S)a.cre[j], ci = (S)a.cim[j], xr = get(j, 0), xi = get(j, 1);
                    if (!(a.flags[j] & JH_SCALAR_COMPLEX)) { tr = cr * xr; ti = cr * xi; }
                    else { tr = cr * xr - ci * xi; ti = cr * xi + ci * xr; }       // Julia Base complex.jl `*`
                }
                if (j == 0) { accr = tr; acci = ti; }
                else { accr = accr + tr; acci = acci + ti; }
            }
        }
        outr = accr;
        outi = acci;
    } else {
        double accr = 0, acci = 0;                   // a narrow value is held exactly
        bool acc_wide = false;
#pragma unroll
        for (int j = 0; j < MAX_TERMS; j++) {
            if (j < a.k) {
                double tr, ti = 0;
                const bool tw = (a.flags[j] & JH_SCALAR_WIDE) != 0, tc = (a.flags[j] & JH_SCALAR_COMPLEX) != 0;
                if (tw) {
                    const double xr = (double)get(j, 0), xi = E == 2 ? (double)get(j, 1) : 0.0;
                    if (E == 1) tr = a.cre[j] * xr;
                    else if (!tc) { tr = a.cre[j] * xr; ti = a.cre[j] * xi; }
                    else { tr = a.cre[j] * xr - a.cim[j] * xi; ti = a.cre[j] * xi + a.cim[j] * xr; }
                } else {
                    const S cr = (S)a.cre[j], ci = (S)a.cim[j], xr = get(j, 0), xi = E == 2 ? get(j, 1) : (S)0;
                    S sr, si = 0;
                    if (E == 1) sr = cr * xr;
                    else if (!tc) { sr = cr * xr; si = cr * xi; }
                    else { sr = cr * xr - ci * xi; si = cr * xi + ci * xr; }
                    tr = (double)sr;
                    ti = (double)si;
                }
                if (j == 0) { accr = tr; acci = ti; acc_wide = tw; }
                else if (acc_wide || tw) { accr = accr + tr; acci = acci + ti; acc_wide = true; }
                else { accr = (double)((S)accr + (S)tr); acci = (double)((S)acci + (S)ti); }
            }
        }
        outr = (S)accr;
        outi = (S)acci;
    }
}

template <typename S, int E, bool WIDE>
__device__ inline void lincomb_elem(const LincombArgs &a, int64_t scalar_index, S *out)
{
    S r, i;
    lincomb_one<S, E, WIDE>(a, [&](int j, int part) { return ((const S *)a.x[j])[scalar_index + part]; }, r, i);
    out[0] = r;
    if (E == 2) out[1] = i;
}

template <typename S, int E, int NS, bool WIDE = false>   // NS scalars per pack, NS % E == 0
__global__ void k_lincomb(S *dst, int64_t n_scalars, LincombArgs a)   // dst may alias an operand (x .= a*x .+ b*y): no __restrict__
{
    const int64_t nvec = n_scalars / NS;
    const int64_t tid = (int64_t)blockIdx.x * WG + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * WG;
    for (int64_t v = tid; v < nvec; v += stride) {
        Pack<S, NS> xin[MAX_TERMS];
#pragma unroll
        for (int j = 0; j < MAX_TERMS; j++)
            if (j < a.k) xin[j] = ldnt(reinterpret_cast<const Pack<S, NS> *>(a.x[j]) + v);
        Pack<S, NS> o;
#pragma unroll
        for (int e = 0; e < NS; e += E) {
            S r, i;
            lincomb_one<S, E, WIDE>(a, [&](int j, int part) { return xin[j].v[e + part]; }, r, i);
            o.v[e] = r;
            if (E == 2) o.v[e + 1] = i;
        }
        stnt(reinterpret_cast<Pack<S, NS> *>(dst) + v, o);
    }
    const int64_t tail0 = nvec * NS;
    const int64_t ntail_elems = (n_scalars - tail0) / E;
    if (tid < ntail_elems) lincomb_elem<S, E, WIDE>(a, tail0 + tid * E, dst + tail0 + tid * E);
}

// ---------------------------------------------------------------- hadamard --------------------
template <typename S, int E, int NS>
__global__ void k_hadamard(S *dst, const S *x, const S *y, int64_t n_scalars, int conj_x)   // dst may alias x or y (d .= mask .* d)
{
    const int64_t nvec = n_scalars / NS;
    const int64_t tid = (int64_t)blockIdx.x * WG + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * WG;
    const bool cj = (conj_x & 1) != 0, twice = (conj_x & 2) != 0;     // bit 1: x -> 2 .* x (x + x, exact)
    auto one = [&](const S *xe, const S *ye, S *oe) {
        if (E == 1) {
            const S xr = twice ? xe[0] + xe[0] : xe[0];
            oe[0] = xr * ye[0];
        } else {
            S xr = xe[0], xi = cj ? -xe[1] : xe[1], yr = ye[0], yi = ye[1];
            if (twice) { xr = xr + xr; xi = xi + xi; }
            oe[0] = xr * yr - xi * yi;
            oe[1] = xr * yi + xi * yr;
        }
    };
    int64_t v = tid;
    for (; v + stride < nvec; v += 2 * stride) {                  // two packs per input in flight
        Pack<S, NS> xv0 = ldnt(reinterpret_cast<const Pack<S, NS> *>(x) + v), yv0 = ldnt(reinterpret_cast<const Pack<S, NS> *>(y) + v);
        Pack<S, NS> xv1 = ldnt(reinterpret_cast<const Pack<S, NS> *>(x) + v + stride), yv1 = ldnt(reinterpret_cast<const Pack<S, NS> *>(y) + v + stride);
        Pack<S, NS> o0, o1;
#pragma unroll
        for (int e = 0; e < NS; e += E) { one(&xv0.v[e], &yv0.v[e], &o0.v[e]); one(&xv1.v[e], &yv1.v[e], &o1.v[e]); }
        stnt(reinterpret_cast<Pack<S, NS> *>(dst) + v, o0);
        stnt(reinterpret_cast<Pack<S, NS> *>(dst) + v + stride, o1);
    }
    for (; v < nvec; v += stride) {
        Pack<S, NS> xv = ldnt(reinterpret_cast<const Pack<S, NS> *>(x) + v);
        Pack<S, NS> yv = ldnt(reinterpret_cast<const Pack<S, NS> *>(y) + v);
        Pack<S, NS> o;
#pragma unroll
        for (int e = 0; e < NS; e += E) one(&xv.v[e], &yv.v[e], &o.v[e]);
        stnt(reinterpret_cast<Pack<S, NS> *>(dst) + v, o);
    }
    const int64_t tail0 = nvec * NS;
    const int64_t ntail_elems = (n_scalars - tail0) / E;
    if (tid < ntail_elems) one(x + tail0 + tid * E, y + tail0 + tid * E, dst + tail0 + tid * E);
}

// ---------------------------------------------------------------- reductions ------------------
enum RedOp { RED_DOT = 0, RED_SUMSQ, RED_SUMABS, RED_COUNTNZ, RED_MAXABS, RED_MINABS, RED_SUMPOW, RED_EXTREMA };

template <int OP> __device__ inline void red_init(double &a0, double &a1)
{
    if (OP == RED_MINABS) { a0 = INFINITY; a1 = 0; }
    else if (OP == RED_EXTREMA) { a0 = INFINITY; a1 = -INFINITY; }
    else { a0 = 0; a1 = 0; }
}
// max / min as Julia defines them on floats (the reference folds with `max` / `min`, src/Jets.jl:835-838, and the stdlib's block
// norms and extrema do the same): a NaN is the answer, and stays the answer -- `b > NaN` and `b < NaN` are false for every b
template <int OP> __device__ inline void red_combine(double &a0, double &a1, double b0, double b1)
{
    if (OP == RED_MAXABS) { a0 = (b0 > a0 || b0 != b0) ? b0 : a0; }
    else if (OP == RED_MINABS) { a0 = (b0 < a0 || b0 != b0) ? b0 : a0; }
    else if (OP == RED_EXTREMA) { a0 = (b0 < a0 || b0 != b0) ? b0 : a0; a1 = (b1 > a1 || b1 != b1) ? b1 : a1; }
    else { a0 += b0; a1 += b1; }
}
template <typename S, int E, int OP>
__device__ inline void red_elem(const S *xe, const S *ye, double p, double scale, double &a0, double &a1)
{
    // scale: 1.0, or the power of two jh_norm rescales by when the plain sum of squares / powers left the double range (exact)
    const double xr = (OP == RED_SUMSQ || OP == RED_SUMPOW) ? (double)xe[0] * scale : (double)xe[0];
    const double xi = (E == 2) ? ((OP == RED_SUMSQ || OP == RED_SUMPOW) ? (double)xe[1] * scale : (double)xe[1]) : 0.0;
    if (OP == RED_DOT) {
        const double yr = (double)ye[0];
        const double yi = (E == 2) ? (double)ye[1] : 0.0;
        // conj(x) * y
        a0 += xr * yr + xi * yi;
        if (E == 2) a1 += xr * yi - xi * yr;
    } else if (OP == RED_SUMSQ) {
        a0 += xr * xr + xi * xi;
        a1 += (xr != 0.0 || xi != 0.0) ? 1.0 : 0.0;          // elements that are not zero: an all-zero vector needs no rescaled second pass (jh_norm)
    } else if (OP == RED_EXTREMA) {
        a0 = (xr < a0 || xr != xr) ? xr : a0;
        a1 = (xr > a1 || xr != xr) ? xr : a1;
    } else {
        const double ab = (E == 2) ? hypot(xr, xi) : fabs(xr);
        if (OP == RED_SUMABS) a0 += ab;
        else if (OP == RED_COUNTNZ) a0 += (ab != 0.0) ? 1.0 : 0.0;
        else if (OP == RED_MAXABS) a0 = (ab > a0 || ab != ab) ? ab : a0;
        else if (OP == RED_MINABS) a0 = (ab < a0 || ab != ab) ? ab : a0;
        else if (OP == RED_SUMPOW) {
            const int ip = (int)p;                     // small integer p: repeated multiplication instead of pow()
            if ((double)ip == p && ip >= 1 && ip <= 8) { double t = ab; for (int q = 1; q < ip; q++) t *= ab; a0 += t; }
            else a0 += pow(ab, p);
        }
    }
}

template <int OP>
__device__ inline void block_reduce_store(double a0, double a1, double *partials)
{
    __shared__ double s0[WG / 64], s1[WG / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double b0 = __shfl_down(a0, off, 64);
        double b1 = __shfl_down(a1, off, 64);
        red_combine<OP>(a0, a1, b0, b1);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { s0[wave] = a0; s1[wave] = a1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r0 = s0[0], r1 = s1[0];
#pragma unroll
        for (int w = 1; w < WG / 64; w++) red_combine<OP>(r0, r1, s0[w], s1[w]);
        partials[2 * blockIdx.x + 0] = r0;
        partials[2 * blockIdx.x + 1] = r1;
    }
}

template <typename S, int E, int NS, int OP>
__global__ void k_reduce(const S *__restrict__ x, const S *__restrict__ y, int64_t n_scalars, double p, double scale, double *__restrict__ partials)
{
    double a0, a1;
    red_init<OP>(a0, a1);
    const int64_t nvec = n_scalars / NS;
    const int64_t tid = (int64_t)blockIdx.x * WG + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * WG;
    constexpr int UN = 4;
    int64_t v = tid;
    for (; v + (UN - 1) * stride < nvec; v += UN * stride) {
        Pack<S, NS> xv[UN], yv[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            xv[u] = ldnt(reinterpret_cast<const Pack<S, NS> *>(x) + v + u * stride);
            if (OP == RED_DOT) yv[u] = ldnt(reinterpret_cast<const Pack<S, NS> *>(y) + v + u * stride);
        }
#pragma unroll
        for (int u = 0; u < UN; u++)
#pragma unroll
            for (int e = 0; e < NS; e += E) red_elem<S, E, OP>(&xv[u].v[e], &yv[u].v[e], p, scale, a0, a1);
    }
    for (; v < nvec; v += stride) {
        Pack<S, NS> xv = ldnt(reinterpret_cast<const Pack<S, NS> *>(x) + v), yv;
        if (OP == RED_DOT) yv = ldnt(reinterpret_cast<const Pack<S, NS> *>(y) + v);
#pragma unroll
        for (int e = 0; e < NS; e += E) red_elem<S, E, OP>(&xv.v[e], &yv.v[e], p, scale, a0, a1);
    }
    const int64_t tail0 = nvec * NS;
    const int64_t ntail_elems = (n_scalars - tail0) / E;
    if (tid < ntail_elems) red_elem<S, E, OP>(x + tail0 + tid * E, OP == RED_DOT ? y + tail0 + tid * E : x, p, scale, a0, a1);
    block_reduce_store<OP>(a0, a1, partials);
}

template <int OP>
__global__ void k_reduce_final(const double *__restrict__ partials, int nparts, double *__restrict__ out)
{
    // one workgroup; thread t folds partials t, t+WG, ... in index order, then the fixed tree
    double a0, a1;
    red_init<OP>(a0, a1);
    for (int i = threadIdx.x; i < nparts; i += WG) red_combine<OP>(a0, a1, partials[2 * i], partials[2 * i + 1]);
    __shared__ double s0[WG], s1[WG];
    s0[threadIdx.x] = a0;
    s1[threadIdx.x] = a1;
    __syncthreads();
    for (int off = WG / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            double r0 = s0[threadIdx.x], r1 = s1[threadIdx.x];
            red_combine<OP>(r0, r1, s0[threadIdx.x + off], s1[threadIdx.x + off]);
            s0[threadIdx.x] = r0;
            s1[threadIdx.x] = r1;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = s0[0]; out[1] = s1[0]; }
}

// ---- per-BLOCK reductions in one launch (round 6): norm(x_i, p) / dot(x_i, y_i) for every block i of a block vector -- what the reference computes block by
// block anyway (src/Jets.jl:836-846, 850-856) and what per-shot residual norms ask for.  One reduction per block through the whole-vector entry points costs a
// launch + a host round trip each (33 us per 64 MiB block = 2 TB/s: 34 ms for 1024 blocks); here workgroup (b, c) reduces chunk c of block b -- `cpb` chunks per
// block, each a contiguous run of whole packs, the block's odd tail scalars with chunk 0 --, a second launch folds every block's chunk partials in chunk order
// (deterministic), one copy brings the nblocks results back.  Blocks may start off the 16-byte grid (odd lengths in one slab): under-aligned packs (ldnt).
template <typename S, int E, int NS, int OP>
__global__ void k_reduce_blocks(const S *__restrict__ x, const S *__restrict__ y, const int64_t *__restrict__ off, int cpb, double p, double *__restrict__ partials)
{
    const int64_t b = blockIdx.x / (unsigned)cpb;
    const int c = (int)(blockIdx.x - b * (unsigned)cpb);
    const int64_t e0 = off[b], e1 = off[b + 1];
    const S *xb = x + e0 * E, *yb = (OP == RED_DOT) ? y + e0 * E : x;
    const int64_t n_scalars = (e1 - e0) * E, nvec = n_scalars / NS;
    const int64_t v0 = nvec / cpb * c + (c < nvec % cpb ? c : nvec % cpb), v1 = v0 + nvec / cpb + (c < nvec % cpb ? 1 : 0);
    double a0, a1;
    red_init<OP>(a0, a1);
    constexpr int UN = 4;
    int64_t v = v0 + threadIdx.x;
    for (; v + (UN - 1) * WG < v1; v += UN * WG) {
        Pack<S, NS> xv[UN], yv[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            xv[u] = ldnt(reinterpret_cast<const Pack<S, NS> *>(xb) + v + u * WG);
            if (OP == RED_DOT) yv[u] = ldnt(reinterpret_cast<const Pack<S, NS> *>(yb) + v + u * WG);
        }
#pragma unroll
        for (int u = 0; u < UN; u++)
#pragma unroll
            for (int e = 0; e < NS; e += E) red_elem<S, E, OP>(&xv[u].v[e], &yv[u].v[e], p, 1.0, a0, a1);
    }
    for (; v < v1; v += WG) {
        Pack<S, NS> xv = ldnt(reinterpret_cast<const Pack<S, NS> *>(xb) + v), yv;
        if (OP == RED_DOT) yv = ldnt(reinterpret_cast<const Pack<S, NS> *>(yb) + v);
#pragma unroll
        for (int e = 0; e < NS; e += E) red_elem<S, E, OP>(&xv.v[e], &yv.v[e], p, 1.0, a0, a1);
    }
    const int64_t tail0 = nvec * NS, ntail_elems = (n_scalars - tail0) / E;
    if (c == 0 && (int64_t)threadIdx.x < ntail_elems)
        red_elem<S, E, OP>(xb + tail0 + threadIdx.x * E, yb + tail0 + threadIdx.x * E, p, 1.0, a0, a1);
    block_reduce_store<OP>(a0, a1, partials);
}

// Many SHORT blocks (traces rather than volumes: per-trace norms): a workgroup per block is a launch slot, a tree and a barrier for a few hundred bytes --
// 262144 blocks of 2 KiB took 6.5 ms where the whole-vector norm takes 0.12.  Here a WAVE owns a block (four blocks per workgroup): lanes stride over its
// packs, the odd tail scalars ride with the first lanes, a shuffle tree, lane 0 writes the block's result -- no partials, no second launch, no barrier.
template <typename S, int E, int NS, int OP>
__global__ __launch_bounds__(WG) void k_reduce_blocks_wave(const S *__restrict__ x, const S *__restrict__ y, const int64_t *__restrict__ off, int64_t nblocks, double p,
                                                           double *__restrict__ out)
{
    const int64_t b = (int64_t)blockIdx.x * (WG / 64) + (threadIdx.x >> 6);
    if (b >= nblocks) return;                                                 // (whole waves leave: no barrier below)
    const int lane = threadIdx.x & 63;
    const int64_t e0 = off[b], e1 = off[b + 1];
    const S *xb = x + e0 * E, *yb = (OP == RED_DOT) ? y + e0 * E : x;
    const int64_t n_scalars = (e1 - e0) * E, nvec = n_scalars / NS;
    double a0, a1;
    red_init<OP>(a0, a1);
    for (int64_t v = lane; v < nvec; v += 64) {
        Pack<S, NS> xv = ldnt(reinterpret_cast<const Pack<S, NS> *>(xb) + v), yv;
        if (OP == RED_DOT) yv = ldnt(reinterpret_cast<const Pack<S, NS> *>(yb) + v);
#pragma unroll
        for (int e = 0; e < NS; e += E) red_elem<S, E, OP>(&xv.v[e], &yv.v[e], p, 1.0, a0, a1);
    }
    const int64_t tail0 = nvec * NS, ntail_elems = (n_scalars - tail0) / E;
    if (lane < ntail_elems) red_elem<S, E, OP>(xb + tail0 + lane * E, yb + tail0 + lane * E, p, 1.0, a0, a1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double b0 = __shfl_down(a0, o, 64);
        double b1 = __shfl_down(a1, o, 64);
        red_combine<OP>(a0, a1, b0, b1);
    }
    if (lane == 0) { out[2 * b] = a0; out[2 * b + 1] = a1; }
}

template <int OP>
__global__ void k_reduce_blocks_final(const double *__restrict__ partials, int cpb, int64_t nblocks, double *__restrict__ out)
{
    const int64_t b = (int64_t)blockIdx.x * WG + threadIdx.x;
    if (b >= nblocks) return;
    double a0, a1;
    red_init<OP>(a0, a1);
    for (int c = 0; c < cpb; c++) red_combine<OP>(a0, a1, partials[2 * (b * cpb + c)], partials[2 * (b * cpb + c) + 1]);
    out[2 * b] = a0;
    out[2 * b + 1] = a1;
}

template <typename S, int E, int OP>
int reduce_blocks_launch(const jh_bvec *x, const jh_bvec *y, double p, double *host_out /* 2 * nblocks */)
{
    jh_context &c = jh_ctx();
    constexpr int NSV = (16 / sizeof(S)) >= E ? (16 / sizeof(S)) : E;
    const int64_t nb = x->nblocks;
    int64_t longest = 0;
    for (int64_t i = 0; i < nb; i++) longest = x->len(i) > longest ? x->len(i) : longest;
    // chunks per block: about 16 K workgroups over the whole vector (a chunk is at least 4 packs per lane), 64 at most
    int64_t cpb = (16384 + nb - 1) / nb;
    const int64_t by_size = (longest * E / NSV + (int64_t)WG * 4 - 1) / ((int64_t)WG * 4);
    if (cpb > by_size) cpb = by_size;
    if (cpb > 64) cpb = 64;
    if (cpb < 1) cpb = 1;
    JH_REQUIRE(nb * cpb < ((int64_t)1 << 31), "per-block reduction: %lld blocks are too many for one launch", (long long)nb);
    JH_TRY(jh_ensure_partials(2 * nb * cpb + 2 * nb));
    void *offs = nullptr;
    JH_TRY(jh_ensure_scratch((size_t)(nb + 1) * sizeof(int64_t), &offs));
    JH_CHECK_HIP(hipMemcpyAsync(offs, x->off.data(), (size_t)(nb + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c.stream));
    double *partials = c.part_dev, *results = c.part_dev + 2 * nb * cpb;
    // many short blocks (at most 16 packs per lane of a wave: 16 KiB) and enough of them to fill the chip with waves: a wave per block, one launch
    if (cpb == 1 && longest * E / NSV <= 64 * 16 && nb >= 4 * (int64_t)c.cu_count && c.red_blocks_wave) {
        hipLaunchKernelGGL((k_reduce_blocks_wave<S, E, NSV, OP>), dim3((unsigned)((nb + WG / 64 - 1) / (WG / 64))), dim3(WG), 0, c.stream, (const S *)x->data,
                           (const S *)(y ? y->data : x->data), (const int64_t *)offs, nb, p, results);
        JH_CHECK_HIP(hipGetLastError());
        JH_CHECK_HIP(hipMemcpyAsync(host_out, results, (size_t)(2 * nb) * sizeof(double), hipMemcpyDeviceToHost, c.stream));
        JH_CHECK_HIP(hipStreamSynchronize(c.stream));
        return JH_OK;
    }
    hipLaunchKernelGGL((k_reduce_blocks<S, E, NSV, OP>), dim3((unsigned)(nb * cpb)), dim3(WG), 0, c.stream, (const S *)x->data,
                       (const S *)(y ? y->data : x->data), (const int64_t *)offs, (int)cpb, p, partials);
    JH_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL((k_reduce_blocks_final<OP>), dim3((unsigned)((nb + WG - 1) / WG)), dim3(WG), 0, c.stream, partials, (int)cpb, nb, results);
    JH_CHECK_HIP(hipGetLastError());
    JH_CHECK_HIP(hipMemcpyAsync(host_out, results, (size_t)(2 * nb) * sizeof(double), hipMemcpyDeviceToHost, c.stream));
    JH_CHECK_HIP(hipStreamSynchronize(c.stream));
    return JH_OK;
}

template <int OP>
int reduce_blocks_dispatch(const jh_bvec *x, const jh_bvec *y, double p, double *host_out)
{
    switch (x->dtype) {
    case JH_F32: return reduce_blocks_launch<float, 1, OP>(x, y, p, host_out);
    case JH_F64: return reduce_blocks_launch<double, 1, OP>(x, y, p, host_out);
    case JH_C32: return reduce_blocks_launch<float, 2, OP>(x, y, p, host_out);
    case JH_C64: return reduce_blocks_launch<double, 2, OP>(x, y, p, host_out);
    }
    return jh_fail(JH_ERR_INVALID, "unknown dtype %d", x->dtype);
}

template <typename S, int E, int OP>
int reduce_launch(const void *x, const void *y, int64_t n_elems, double p, double *r0, double *r1, double scale)
{
    jh_context &c = jh_ctx();
    const int64_t n_scalars = n_elems * E;
    constexpr int NSV = (16 / sizeof(S)) >= E ? (16 / sizeof(S)) : E;
    const bool aligned = (((uintptr_t)x | (uintptr_t)(y ? y : x)) & (sizeof(S) - 1)) == 0;   // (like the scalar: ldnt addresses its pack under-aligned)
    int grid;
    const int64_t cap = c.red_wgs > 0 ? c.red_wgs : 16384;
    auto clamp_grid = [&](int64_t packs) {
        int64_t g = (packs + (int64_t)WG * 4 - 1) / ((int64_t)WG * 4);
        return (int)(g < 1 ? 1 : (g > cap ? cap : g));
    };
    JH_TRY(jh_ensure_partials(2 * cap));
    double *partials = c.part_dev;      // one (a0, a1) pair per workgroup
    if (aligned) {
        grid = clamp_grid(n_scalars / NSV + 1);
        hipLaunchKernelGGL((k_reduce<S, E, NSV, OP>), dim3(grid), dim3(WG), 0, c.stream, (const S *)x, (const S *)y,
                           n_scalars, p, scale, partials);
    } else {
        grid = clamp_grid(n_elems + 1);
        hipLaunchKernelGGL((k_reduce<S, E, E, OP>), dim3(grid), dim3(WG), 0, c.stream, (const S *)x, (const S *)y, n_scalars,
                           p, scale, partials);
    }
    JH_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL((k_reduce_final<OP>), dim3(1), dim3(WG), 0, c.stream, partials, grid, c.red_dev);
    JH_CHECK_HIP(hipGetLastError());
    JH_CHECK_HIP(hipMemcpyAsync(c.red_host, c.red_dev, 2 * sizeof(double), hipMemcpyDeviceToHost, c.stream));
    if (c.red_defer) { *r0 = *r1 = 0.0; return JH_OK; }                      // jh_dot_begin: the caller reads red_host after its own wait (jh_dot_end)
    JH_CHECK_HIP(hipStreamSynchronize(c.stream));
    *r0 = c.red_host[0];
    *r1 = c.red_host[1];
    return JH_OK;
}

template <int OP>
int reduce_dispatch(int dtype, const void *x, const void *y, int64_t n, double p, double *r0, double *r1, double scale = 1.0)
{
    switch (dtype) {
    case JH_F32: return reduce_launch<float, 1, OP>(x, y, n, p, r0, r1, scale);
    case JH_F64: return reduce_launch<double, 1, OP>(x, y, n, p, r0, r1, scale);
    case JH_C32: return reduce_launch<float, 2, OP>(x, y, n, p, r0, r1, scale);
    case JH_C64: return reduce_launch<double, 2, OP>(x, y, n, p, r0, r1, scale);
    }
    return jh_fail(JH_ERR_INVALID, "unknown dtype %d", dtype);
}

template <typename S, int E, bool WIDE>
void lincomb_launch_w(void *dst, int64_t n_elems, const LincombArgs &a)
{
    const int64_t n_scalars = n_elems * E;
    constexpr int NSV = (16 / sizeof(S)) >= E ? (16 / sizeof(S)) : E;
    uintptr_t bits = (uintptr_t)dst;
    for (int j = 0; j < a.k; j++) bits |= (uintptr_t)a.x[j];
    if ((bits & (sizeof(S) - 1)) == 0)                                     // aligned like the scalar: ldnt / stnt address their packs under-aligned
        hipLaunchKernelGGL((k_lincomb<S, E, NSV, WIDE>), dim3(grid_full(n_scalars / NSV + 1)), dim3(WG), 0, jh_ctx().stream, (S *)dst,
                           n_scalars, a);
    else
        hipLaunchKernelGGL((k_lincomb<S, E, E, WIDE>), dim3(grid_full(n_elems + 1)), dim3(WG), 0, jh_ctx().stream, (S *)dst, n_scalars, a);
}

template <typename S, int E>
int lincomb_launch(void *dst, int64_t n_elems, const LincombArgs &a, bool any_wide = false)
{
    if (n_elems * E == 0) return JH_OK;
    if constexpr (sizeof(S) == 4) {
        if (any_wide) lincomb_launch_w<S, E, true>(dst, n_elems, a);
        else lincomb_launch_w<S, E, false>(dst, n_elems, a);
    } else {
        lincomb_launch_w<S, E, false>(dst, n_elems, a);
    }
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

template <typename S, int E>
int hadamard_launch(void *dst, const void *x, const void *y, int64_t n_elems, int conj_x)
{
    const int64_t n_scalars = n_elems * E;
    if (n_scalars == 0) return JH_OK;
    constexpr int NSV = (16 / sizeof(S)) >= E ? (16 / sizeof(S)) : E;
    uintptr_t bits = (uintptr_t)dst | (uintptr_t)x | (uintptr_t)y;
    if ((bits & (sizeof(S) - 1)) == 0)
        hipLaunchKernelGGL((k_hadamard<S, E, NSV>), dim3(grid_full(n_scalars / NSV + 1)), dim3(WG), 0, jh_ctx().stream, (S *)dst,
                           (const S *)x, (const S *)y, n_scalars, conj_x);
    else
        hipLaunchKernelGGL((k_hadamard<S, E, E>), dim3(grid_full(n_elems + 1)), dim3(WG), 0, jh_ctx().stream, (S *)dst,
                           (const S *)x, (const S *)y, n_scalars, conj_x);
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

}  // namespace

int jh_launch_hadamard_raw(void *dst, const void *x, const void *y, int dtype, int64_t count, int conj_x)
{
    switch (dtype) {
    case JH_F32: return hadamard_launch<float, 1>(dst, x, y, count, conj_x);
    case JH_F64: return hadamard_launch<double, 1>(dst, x, y, count, conj_x);
    case JH_C32: return hadamard_launch<float, 2>(dst, x, y, count, conj_x);
    case JH_C64: return hadamard_launch<double, 2>(dst, x, y, count, conj_x);
    }
    return jh_fail(JH_ERR_INVALID, "hadamard: unknown dtype %d", dtype);
}

int jh_launch_square_jvp_raw(void *dst, const void *mo, const void *x, int dtype, int64_t count, int conj_mo)
{
    return jh_launch_hadamard_raw(dst, mo, x, dtype, count, 2 | (conj_mo ? 1 : 0));
}

// dst = c0*x0 (+ c1*x1): k in {1, 2}; flags: the coefficients' JH_SCALAR_* (nullptr: untyped)
int jh_launch_lincomb_raw(void *dst, int dtype, int64_t count, int k, const double *cre, const double *cim, const void *const *x, const int32_t *flags)
{
    LincombArgs a;
    a.k = k;
    for (int j = 0; j < k; j++) { a.x[j] = x[j]; a.cre[j] = cre[j]; a.cim[j] = cim[j]; }
    const bool wide = lincomb_args_flags(a, flags, dtype);
    switch (dtype) {
    case JH_F32: return lincomb_launch<float, 1>(dst, count, a, wide);
    case JH_F64: return lincomb_launch<double, 1>(dst, count, a);
    case JH_C32: return lincomb_launch<float, 2>(dst, count, a, wide);
    case JH_C64: return lincomb_launch<double, 2>(dst, count, a);
    }
    return jh_fail(JH_ERR_INVALID, "lincomb: unknown dtype %d", dtype);
}

// device-to-device copy of `bytes` bytes: own streaming kernel when both ends are 16-byte aligned and large,
// the runtime's copy otherwise
int jh_launch_copy_bytes(void *dst, const void *src, size_t bytes)
{
    hipStream_t st = jh_ctx().stream;
    if (bytes == 0 || dst == src) return JH_OK;
    if (bytes >= ((size_t)1 << 20) && ((((uintptr_t)dst) | ((uintptr_t)src)) & 15u) == 0) {
        const int64_t nvec = (int64_t)(bytes / 16);
        hipLaunchKernelGGL(k_copy16, dim3(grid_full(nvec)), dim3(WG), 0, st, (uint4 *)dst, (const uint4 *)src, nvec);
        JH_CHECK_HIP(hipGetLastError());
        const size_t done = (size_t)nvec * 16;
        if (done < bytes) JH_CHECK_HIP(hipMemcpyAsync((char *)dst + done, (const char *)src + done, bytes - done, hipMemcpyDeviceToDevice, st));
        return JH_OK;
    }
    JH_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
    return JH_OK;
}

int jh_launch_fill_range(void *ptr, int dtype, int64_t count, double re, double im)
{
    switch (dtype) {
    case JH_F32: return fill_scalars<float, 1>((float *)ptr, count, re, 0);
    case JH_F64: return fill_scalars<double, 1>((double *)ptr, count, re, 0);
    case JH_C32: return fill_scalars<float, 2>((float *)ptr, 2 * count, re, im);
    case JH_C64: return fill_scalars<double, 2>((double *)ptr, 2 * count, re, im);
    }
    return jh_fail(JH_ERR_INVALID, "fill: unknown dtype %d", dtype);
}

extern "C" {

int jh_fill_uniform(jh_bvec *v, uint64_t seed, uint64_t stream, int64_t index_base)
{
    JH_TRY(jh_enter(v));
    JH_REQUIRE(v, "jh_fill_uniform: null vector");
    JH_REQUIRE(index_base >= 0, "jh_fill_uniform: negative index_base");
    JH_REQUIRE((((uintptr_t)v->data) & 15u) == 0, "jh_fill_uniform: slab must be 16-byte aligned");
    if (v->length == 0) return JH_OK;
    const uint64_t key = mix64(seed * GOLDEN + stream);
    const int lanes = jh_dtype_complex(v->dtype) ? 2 : 1;
    const int64_t n = v->length * lanes, base = index_base * lanes;
    hipStream_t st = jh_ctx().stream;
    if (v->dtype == JH_F32 || v->dtype == JH_C32)          // pack 0 starts at lane `base`: its first hash is that of pair base >> 1
        hipLaunchKernelGGL((k_uniform<float, 4>), dim3(grid_full(n / 4 + 1)), dim3(WG), 0, st, (float *)v->data, n, key, base,
                           key + (uint64_t)((base >> 1) + 1) * GOLDEN, (int)(base & 1));
    else
        hipLaunchKernelGGL((k_uniform<double, 2>), dim3(grid_full(n / 2 + 1)), dim3(WG), 0, st, (double *)v->data, n, key, base,
                           key + (uint64_t)(base + 1) * GOLDEN, 0);
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

int jh_fill_normal(jh_bvec *v, uint64_t seed, uint64_t stream, int64_t index_base)
{
    JH_TRY(jh_enter(v));
    JH_REQUIRE(v, "jh_fill_normal: null vector");
    JH_REQUIRE(index_base >= 0, "jh_fill_normal: negative index_base");
    if (v->length == 0) return JH_OK;
    const uint64_t key = mix64(seed * GOLDEN + stream);
    const bool cplx = jh_dtype_complex(v->dtype);
    const int lanes = cplx ? 2 : 1;
    const int64_t n = v->length * lanes, base = index_base * lanes;
    const double scale = cplx ? 0.7071067811865476 : 1.0;
    hipStream_t st = jh_ctx().stream;
    if (v->dtype == JH_F32 || v->dtype == JH_C32)
        hipLaunchKernelGGL((k_normal<float>), dim3(grid_full(n)), dim3(WG), 0, st, (float *)v->data, n, key, base, (float)scale);
    else
        hipLaunchKernelGGL((k_normal<double>), dim3(grid_full(n)), dim3(WG), 0, st, (double *)v->data, n, key, base, scale);
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

int jh_abs(jh_bvec *dst, const jh_bvec *x)
{
    JH_TRY(jh_enter(dst, x));
    JH_REQUIRE(dst && x, "jh_abs: null argument");
    JH_REQUIRE(dst->length == x->length, "jh_abs: length mismatch (%lld vs %lld)", (long long)dst->length, (long long)x->length);
    const int want = (x->dtype == JH_F32 || x->dtype == JH_C32) ? JH_F32 : JH_F64;
    JH_REQUIRE(dst->dtype == want, "jh_abs: destination must be the real type of the source (dtype %d), got %d", want, dst->dtype);
    if (x->length == 0) return JH_OK;
    hipStream_t st = jh_ctx().stream;
    const int g = grid_full(x->length);
    switch (x->dtype) {
    case JH_F32: hipLaunchKernelGGL((k_abs<float, 1>), dim3(g), dim3(WG), 0, st, (float *)dst->data, (const float *)x->data, x->length); break;
    case JH_F64: hipLaunchKernelGGL((k_abs<double, 1>), dim3(g), dim3(WG), 0, st, (double *)dst->data, (const double *)x->data, x->length); break;
    case JH_C32: hipLaunchKernelGGL((k_abs<float, 2>), dim3(g), dim3(WG), 0, st, (float *)dst->data, (const float *)x->data, x->length); break;
    case JH_C64: hipLaunchKernelGGL((k_abs<double, 2>), dim3(g), dim3(WG), 0, st, (double *)dst->data, (const double *)x->data, x->length); break;
    }
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

int jh_lincomb_typed(jh_bvec *dst, int k, const double *coef, const int32_t *coef_flags, const jh_bvec *const *x)
{
    JH_TRY(jh_enter(dst));
    JH_REQUIRE(dst && coef && x, "jh_lincomb: null argument");
    JH_REQUIRE(k >= 1 && k <= MAX_TERMS, "jh_lincomb: k = %d outside 1..%d", k, MAX_TERMS);
    LincombArgs a;
    a.k = k;
    for (int j = 0; j < k; j++) {
        JH_REQUIRE(x[j], "jh_lincomb: null operand %d", j);
        JH_REQUIRE(x[j]->ctx == dst->ctx, "jh_lincomb: operand %d lives in context %d, the destination in %d", j, x[j]->ctx, dst->ctx);
        JH_REQUIRE(x[j]->dtype == dst->dtype, "jh_lincomb: dtype mismatch on operand %d", j);
        JH_REQUIRE(x[j]->length == dst->length, "jh_lincomb: length mismatch on operand %d (%lld vs %lld)", j,
                   (long long)x[j]->length, (long long)dst->length);
        JH_REQUIRE(jh_dtype_complex(dst->dtype) || (coef[2 * j + 1] == 0.0 && !(coef_flags && (coef_flags[j] & JH_SCALAR_COMPLEX))),
                   "jh_lincomb: complex coefficient %d on a real vector", j);
        JH_REQUIRE(!coef_flags || (coef_flags[j] & ~(JH_SCALAR_COMPLEX | JH_SCALAR_WIDE)) == 0, "jh_lincomb_typed: unknown flags %d on coefficient %d",
                   coef_flags ? coef_flags[j] : 0, j);
        a.x[j] = x[j]->data;
        a.cre[j] = coef[2 * j];
        a.cim[j] = coef[2 * j + 1];
    }
    const bool wide = lincomb_args_flags(a, coef_flags, dst->dtype);
    switch (dst->dtype) {
    case JH_F32: return lincomb_launch<float, 1>(dst->data, dst->length, a, wide);
    case JH_F64: return lincomb_launch<double, 1>(dst->data, dst->length, a);
    case JH_C32: return lincomb_launch<float, 2>(dst->data, dst->length, a, wide);
    case JH_C64: return lincomb_launch<double, 2>(dst->data, dst->length, a);
    }
    return jh_fail(JH_ERR_INVALID, "jh_lincomb: unknown dtype %d", dst->dtype);
}

int jh_lincomb(jh_bvec *dst, int k, const double *coef, const jh_bvec *const *x) { return jh_lincomb_typed(dst, k, coef, nullptr, x); }

int jh_hadamard(jh_bvec *dst, const jh_bvec *x, const jh_bvec *y, int conj_x)
{
    JH_TRY(jh_enter(dst, x, y));
    JH_REQUIRE(dst && x && y, "jh_hadamard: null argument");
    JH_REQUIRE(dst->dtype == x->dtype && dst->dtype == y->dtype, "jh_hadamard: dtype mismatch");
    JH_REQUIRE(conj_x >= 0 && conj_x <= 3, "jh_hadamard: flags must be 0..3 (got %d)", conj_x);
    JH_REQUIRE(dst->length == x->length && dst->length == y->length, "jh_hadamard: length mismatch (%lld, %lld, %lld)",
               (long long)dst->length, (long long)x->length, (long long)y->length);
    switch (dst->dtype) {
    case JH_F32: return hadamard_launch<float, 1>(dst->data, x->data, y->data, dst->length, conj_x);
    case JH_F64: return hadamard_launch<double, 1>(dst->data, x->data, y->data, dst->length, conj_x);
    case JH_C32: return hadamard_launch<float, 2>(dst->data, x->data, y->data, dst->length, conj_x);
    case JH_C64: return hadamard_launch<double, 2>(dst->data, x->data, y->data, dst->length, conj_x);
    }
    return jh_fail(JH_ERR_INVALID, "jh_hadamard: unknown dtype %d", dst->dtype);
}

int jh_dot(const jh_bvec *x, const jh_bvec *y, double *re, double *im)
{
    JH_TRY(jh_enter(x, y));
    JH_REQUIRE(x && y && re, "jh_dot: null argument");
    JH_REQUIRE(x->dtype == y->dtype, "jh_dot: dtype mismatch (%d vs %d)", x->dtype, y->dtype);
    JH_REQUIRE(x->length == y->length, "jh_dot: length mismatch (%lld vs %lld)", (long long)x->length, (long long)y->length);
    double r0 = 0, r1 = 0;
    if (x->length > 0) JH_TRY((reduce_dispatch<RED_DOT>(x->dtype, x->data, y->data, x->length, 0.0, &r0, &r1)));
    *re = r0;
    if (im) *im = r1;
    return JH_OK;
}

// The two halves of jh_dot, for loops over the members of a team (jh_lsqr.hip: cgls_impl): jh_dot_begin enqueues the reduction and the
// copy of its result into the context's pinned landing zone and returns; jh_dot_end waits for x's context and reads it.  Nothing else
// that returns a scalar may run in that context in between (the landing zone is per context).
int jh_dot_begin(const jh_bvec *x, const jh_bvec *y)
{
    JH_TRY(jh_enter(x, y));
    JH_REQUIRE(x && y, "jh_dot_begin: null argument");
    JH_REQUIRE(x->dtype == y->dtype && x->length == y->length && x->length > 0, "jh_dot_begin: vectors differ in element type or length (or are empty)");
    jh_context &c = jh_ctx();
    double r0 = 0, r1 = 0;
    c.red_defer = 1;
    const int st = reduce_dispatch<RED_DOT>(x->dtype, x->data, y->data, x->length, 0.0, &r0, &r1);
    c.red_defer = 0;
    return st;
}

int jh_dot_end(const jh_bvec *x, double *re, double *im)
{
    JH_TRY(jh_enter(x));
    JH_REQUIRE(x && re, "jh_dot_end: null argument");
    jh_context &c = jh_ctx();
    JH_CHECK_HIP(hipStreamSynchronize(c.stream));
    *re = c.red_host[0];
    if (im) *im = c.red_host[1];
    return JH_OK;
}

int jh_norm(const jh_bvec *x, double p, double *out)
{
    JH_TRY(jh_enter(x));
    JH_REQUIRE(x && out, "jh_norm: null argument");
    JH_REQUIRE(!std::isnan(p), "jh_norm: p is NaN");
    double r0 = 0, r1 = 0;
    if (x->length == 0) { *out = 0.0; return JH_OK; }
    if (p == INFINITY) {            // src/Jets.jl:835-836
        JH_TRY((reduce_dispatch<RED_MAXABS>(x->dtype, x->data, nullptr, x->length, p, &r0, &r1)));
        *out = r0;
    } else if (p == -INFINITY) {    // :837-838
        JH_TRY((reduce_dispatch<RED_MINABS>(x->dtype, x->data, nullptr, x->length, p, &r0, &r1)));
        *out = r0;
    } else if (p == 1.0) {          // :839-840
        JH_TRY((reduce_dispatch<RED_SUMABS>(x->dtype, x->data, nullptr, x->length, p, &r0, &r1)));
        *out = r0;
    } else if (p == 0.0) {          // :841-842
        JH_TRY((reduce_dispatch<RED_COUNTNZ>(x->dtype, x->data, nullptr, x->length, p, &r0, &r1)));
        *out = r0;
    } else {                        // :843-846; p = 2: (sum_i norm(x_i)^2)^(1/2)
        const bool two = (p == 2.0);
        if (two) JH_TRY((reduce_dispatch<RED_SUMSQ>(x->dtype, x->data, nullptr, x->length, p, &r0, &r1)));
        else JH_TRY((reduce_dispatch<RED_SUMPOW>(x->dtype, x->data, nullptr, x->length, p, &r0, &r1)));
        *out = two ? sqrt(r0) : pow(r0, 1.0 / p);
        // The stdlib's block norms rescale (BLAS nrm2 / generic_normp), so they neither overflow on 1e200 nor lose 1e-200.  The
        // plain sum above does when the powers leave the double range (only Float64 data can do that for p = 2): seen as an
        // infinite or vanishing sum, in which case the pass is repeated on x / 2^k with 2^k ~ max|x| (a power of two: exact).
        // (An exactly zero vector -- x0 of a solver, a fresh zeros(R) -- also has a vanishing sum: the SUMSQ pass counts the elements
        // that are not zero beside the sum, and zero of them means the answer is 0 without a second pass and its host round trip.)
        if (p > 0 && (std::isinf(r0) || r0 < 1e-290) && !(two && r0 == 0.0 && r1 == 0.0)) {
            double big = 0, unused = 0;
            JH_TRY((reduce_dispatch<RED_MAXABS>(x->dtype, x->data, nullptr, x->length, p, &big, &unused)));
            if (big > 0 && std::isfinite(big)) {
                int ex = 0;
                (void)frexp(big, &ex);
                const double down = ldexp(1.0, -ex), up = ldexp(1.0, ex);
                if (two) JH_TRY((reduce_dispatch<RED_SUMSQ>(x->dtype, x->data, nullptr, x->length, p, &r0, &r1, down)));
                else JH_TRY((reduce_dispatch<RED_SUMPOW>(x->dtype, x->data, nullptr, x->length, p, &r0, &r1, down)));
                *out = (two ? sqrt(r0) : pow(r0, 1.0 / p)) * up;
            }
        }
    }
    return JH_OK;
}

// norm(x_i, p) of EVERY block in one pass (src/Jets.jl:836-846 computes them block by block before combining): out[i], i = 0 .. nblocks - 1
int jh_norm_blocks(const jh_bvec *x, double p, double *out)
{
    JH_TRY(jh_enter(x));
    JH_REQUIRE(x && out, "jh_norm_blocks: null argument");
    JH_REQUIRE(!std::isnan(p), "jh_norm_blocks: p is NaN");
    const int64_t nb = x->nblocks;
    if (nb == 0) return JH_OK;
    if (x->length == 0) { for (int64_t i = 0; i < nb; i++) out[i] = 0.0; return JH_OK; }
    JH_REQUIRE((((uintptr_t)x->data) & ((jh_dtype_complex(x->dtype) ? jh_dtype_size(x->dtype) / 2 : jh_dtype_size(x->dtype)) - 1)) == 0,
               "jh_norm_blocks: the vector is not aligned like its scalar");
    std::vector<double> r((size_t)(2 * nb));
    const bool two = (p == 2.0);
    if (p == INFINITY) JH_TRY((reduce_blocks_dispatch<RED_MAXABS>(x, nullptr, p, r.data())));
    else if (p == -INFINITY) JH_TRY((reduce_blocks_dispatch<RED_MINABS>(x, nullptr, p, r.data())));
    else if (p == 1.0) JH_TRY((reduce_blocks_dispatch<RED_SUMABS>(x, nullptr, p, r.data())));
    else if (p == 0.0) JH_TRY((reduce_blocks_dispatch<RED_COUNTNZ>(x, nullptr, p, r.data())));
    else if (two) JH_TRY((reduce_blocks_dispatch<RED_SUMSQ>(x, nullptr, p, r.data())));
    else JH_TRY((reduce_blocks_dispatch<RED_SUMPOW>(x, nullptr, p, r.data())));
    for (int64_t i = 0; i < nb; i++) {
        const double r0 = r[(size_t)(2 * i)], r1 = r[(size_t)(2 * i + 1)];
        if (x->len(i) == 0) { out[i] = 0.0; continue; }                       // (norm of an empty block: 0, also for p = -Inf where the fold starts at Inf)
        if (p == INFINITY || p == -INFINITY || p == 1.0 || p == 0.0) { out[i] = r0; continue; }
        out[i] = two ? sqrt(r0) : pow(r0, 1.0 / p);
        // (a block whose powers left the double range: the whole-vector entry point's rescaled second pass, on that block alone -- jh_norm)
        if (p > 0 && (std::isinf(r0) || r0 < 1e-290) && !(two && r0 == 0.0 && r1 == 0.0)) {
            jh_bvec v;
            v.ctx = x->ctx; v.dtype = x->dtype; v.nblocks = 1; v.length = x->len(i); v.off = {0, x->len(i)}; v.data = x->ptr(x->off[(size_t)i]); v.uniform = true;
            JH_TRY(jh_norm(&v, p, &out[i]));
        }
    }
    return JH_OK;
}

// dot(x_i, y_i) of every block pair in one pass (850-856: `a += dot(x_i, y_i)` block by block); conj on the first argument; im may be NULL
int jh_dot_blocks(const jh_bvec *x, const jh_bvec *y, double *re, double *im)
{
    JH_TRY(jh_enter(x, y));
    JH_REQUIRE(x && y && re, "jh_dot_blocks: null argument");
    JH_REQUIRE(x->dtype == y->dtype, "jh_dot_blocks: dtype mismatch (%d vs %d)", x->dtype, y->dtype);
    JH_REQUIRE(x->nblocks == y->nblocks && x->off == y->off, "jh_dot_blocks: the vectors' blocks differ");
    const int64_t nb = x->nblocks;
    if (nb == 0) return JH_OK;
    std::vector<double> r((size_t)(2 * nb), 0.0);
    if (x->length > 0) {
        const size_t sa = jh_dtype_complex(x->dtype) ? jh_dtype_size(x->dtype) / 2 : jh_dtype_size(x->dtype);
        JH_REQUIRE(((((uintptr_t)x->data) | ((uintptr_t)y->data)) & (sa - 1)) == 0, "jh_dot_blocks: a vector is not aligned like its scalar");
        JH_TRY((reduce_blocks_dispatch<RED_DOT>(x, y, 0.0, r.data())));
    }
    for (int64_t i = 0; i < nb; i++) {
        re[i] = r[(size_t)(2 * i)];
        if (im) im[i] = r[(size_t)(2 * i + 1)];
    }
    return JH_OK;
}

int jh_extrema(const jh_bvec *x, double *mn, double *mx)
{
    JH_TRY(jh_enter(x));
    JH_REQUIRE(x && mn && mx, "jh_extrema: null argument");
    JH_REQUIRE(!jh_dtype_complex(x->dtype), "jh_extrema: complex values are not ordered");
    JH_REQUIRE(x->length > 0, "jh_extrema: empty vector");
    double r0 = 0, r1 = 0;
    JH_TRY((reduce_dispatch<RED_EXTREMA>(x->dtype, x->data, nullptr, x->length, 0.0, &r0, &r1)));
    if (r0 != r0 || r1 != r1) {
        // a NaN somewhere.  The reference folds the BLOCKS' extrema with `_mn < mn && (mn = _mn)` (src/Jets.jl:870-878), and the
        // stdlib's extrema of a block holding a NaN is (NaN, NaN) (Julia >= 1.8): a NaN in the first block is the answer; a later
        // block holding one compares false both ways and drops out, finite values and all.  One reduction per block, in order.
        bool first = true;
        for (int64_t i = 0; i < x->nblocks; i++) {
            if (x->len(i) == 0) continue;
            double b0 = 0, b1 = 0;
            JH_TRY((reduce_dispatch<RED_EXTREMA>(x->dtype, x->ptr(x->off[i]), nullptr, x->len(i), 0.0, &b0, &b1)));
            if (b0 != b0 || b1 != b1) b0 = b1 = NAN;
            if (first) { r0 = b0; r1 = b1; first = false; continue; }
            if (b0 < r0) r0 = b0;
            if (b1 > r1) r1 = b1;
        }
    }
    *mn = r0;
    *mx = r1;
    return JH_OK;
}

}  // extern "C"
