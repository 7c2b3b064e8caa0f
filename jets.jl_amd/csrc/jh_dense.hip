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

// every operand of these kernels lives in HBM: load / store through address_space(1) pointers, so that the compiler emits global_load /
// global_store (never flat_*: a flat access also waits on the LDS counter and costs an address-space check) -- in particular for the
// matrix pointers READ FROM THE BLOCK TABLE, whose address space the compiler cannot know (round 5; the block-operator kernels -- jh_blockop_common.h: ld / st -- have done so since
// round 1; tests/test_kernel_resources.py disassembles this code object and finds no flat access)
template <typename V> __device__ inline V ldg_nt(const V *p)
{
    typedef const V __attribute__((address_space(1))) *gp;
    return __builtin_nontemporal_load((gp)p);
}
template <typename V> __device__ inline V ldg(const V *p)
{
    typedef const V __attribute__((address_space(1))) *gp;
    return *(gp)p;
}
template <typename V, typename W> __device__ inline void stg(V *p, W v)
{
    typedef V __attribute__((address_space(1))) *gp;
    *(gp)p = (V)v;
}
// UNDER-ALIGNED packs (round 5, session 3; the block-operator kernels' ldu, jh_blockop_common.h): a pack addressed through a type aligned like its scalar -- the
// same global_load_dwordx4, at any dword-aligned address.  The adjoint (column) kernels use them: a k x k child with k odd has every column off the 16-byte
// grid and a partial last pack per column; that pack is loaded from the column's end minus NS and only its scalars from e0 on are summed (cols_accumulate).
template <typename S, int NS> __device__ inline typename vec_of<S, NS>::type ldgu_nt(const S *p)
{
    typedef typename vec_of<S, NS>::type V;
    if constexpr (NS == 1) {
        return ldg_nt(p);
    } else {
        typedef V __attribute__((aligned(alignof(S)))) UV;
        typedef const UV __attribute__((address_space(1))) *gp;
        return __builtin_nontemporal_load((gp)p);
    }
}
template <typename S, int NS> __device__ inline typename vec_of<S, NS>::type ldgu(const S *p)
{
    typedef typename vec_of<S, NS>::type V;
    if constexpr (NS == 1) {
        return ldg(p);
    } else {
        typedef V __attribute__((aligned(alignof(S)))) UV;
        typedef const UV __attribute__((address_space(1))) *gp;
        return *(gp)p;
    }
}
// store the pack loaded from sc for the nominal start s (jh_blockop_common.h: st_pack): all of it, or the scalars from s on
template <typename S, int NS> __device__ inline void stgu_pack(S *row, int64_t s, int64_t sc, typename vec_of<S, NS>::type v)
{
    typedef typename vec_of<S, NS>::type V;
    if constexpr (NS == 1) {
        stg(row + s, v);
    } else {
        if (sc == s) {
            typedef V __attribute__((aligned(alignof(S)))) UV;
            typedef UV __attribute__((address_space(1))) *gp;
            *(gp)(row + s) = v;
        } else {
#pragma unroll
            for (int e = 0; e < NS; e++)
                if (sc + e >= s) stg(row + sc + e, (S)v[e]);
        }
    }
}
// sr (+ i si) += conj(a) . x over the scalars e >= e0 of one pack (e0 = 0: the whole pack)
template <typename S, int E, int NS, typename V> __device__ inline void cols_accumulate(const V &a, const V &xv, int e0, double &sr, double &si)
{
    if constexpr (E == 1) {
        if constexpr (NS == 1) { if (e0 <= 0) sr += (double)a * (double)xv; }
        else {
#pragma unroll
            for (int e = 0; e < NS; e++) sr += e >= e0 ? (double)a[e] * (double)xv[e] : 0.0;
        }
    } else {
#pragma unroll
        for (int e = 0; e < NS; e += 2) {
            const double ar = a[e], ai = -(double)a[e + 1], xr = xv[e], xi = xv[e + 1];
            if (e >= e0) {
                sr += ar * xr - ai * xi;
                si += ar * xi + ai * xr;
            }
        }
    }
}

// ---- y = A x ---------------------------------------------------------------------------------------
// acc += A[:, c0:c1] x[c0:c1] for this lane's NS scalar rows: columns in order, product rounded then added (the sequential loop's bits);
// sixteen columns' loads in flight (the adds are serial by definition, the loads need not be).  Round 4 A/B against four in flight:
// no difference at any size (4 x 8192^2 children 4.8, 16 x 4096^2 5.2, smaller ones 6.2-6.5 TB/s either way,
// profiles/bench_dense_blocks_r04.txt) -- the forward of FEW HUGE children is not held back by memory-level parallelism
template <typename S, int E, int NS, typename V>
__device__ inline V gemv_rows_walk(const S *__restrict__ col, int64_t ns, const S *__restrict__ x, int64_t c0, int64_t c1)
{
    V acc = (V)(S)0;
    int64_t c = c0;
    for (; c + 16 <= c1; c += 16) {
        V a[16];
#pragma unroll
        for (int k = 0; k < 16; k++) a[k] = ldg_nt(reinterpret_cast<const V *>(col + (int64_t)k * ns));
        col += 16 * ns;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if constexpr (E == 1) {
                acc = acc + a[k] * (V)x[c + k];
            } else {
                const S xr = x[2 * (c + k)], xi = x[2 * (c + k) + 1];
                V p;
#pragma unroll
                for (int e = 0; e < NS; e += 2) {
                    p[e] = a[k][e] * xr - a[k][e + 1] * xi;
                    p[e + 1] = a[k][e] * xi + a[k][e + 1] * xr;
                }
                acc = acc + p;
            }
        }
    }
    for (; c < c1; c++, col += ns) {
        V a = ldg_nt(reinterpret_cast<const V *>(col));
        if constexpr (E == 1) {
            acc = acc + a * (V)x[c];
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
    return acc;
}


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
    const V acc = gemv_rows_walk<S, E, NS, V>(A + c0 * ns + s, ns, x, c0, c1);  // product rounded, then added (no FMA: -ffp-contract=off)
    stg(reinterpret_cast<V *>(out + (int64_t)blockIdx.y * ns + s), acc);          // chunk 0 of a one-chunk launch is y itself
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
    // (columns need not be whole, 16-byte aligned packs: under-aligned loads, the chunk's last pack from s1 - NS -- the host guarantees ns >= NS)
    for (int64_t s = s0 + (int64_t)lane * NS; s < s1; s += 64 * NS) {
        const int64_t sc = s + NS <= s1 ? s : s1 - NS;
        V a = ldgu_nt<S, NS>(col + sc);
        V xv = ldgu<S, NS>(x + sc);
        cols_accumulate<S, E, NS, V>(a, xv, (int)(s - sc), sr, si);              // conj(a) * x
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

// ---- the same two kernels BATCHED over the children of a tall block operator of dense blocks (blockIdx.z = child) -----------
// A tall operator of N dense children run child by child is N tiny launches in a row -- a 256 x 256 child keeps ONE workgroup
// busy for ~100 us, so 4096 of them (1 GiB) take 413 ms forward (profiles/bench_dense_blocks_r01.txt).  Batched, the chip sees
// all children at once.  Uniform children only (same nr x nc, none adjointed); child z's matrix is blocks[z].coeff, its range
// block sits at y + z*nr*E (the slab layout), the domain vector x is shared.
template <typename S, int E, int NS>
__global__ __launch_bounds__(256) void k_gemv_rows_batched(const jh_dev_block *__restrict__ blocks, int64_t z0, int64_t nr, int64_t nc,
                                                           const S *__restrict__ x, int64_t x_stride, S *__restrict__ out,
                                                           int64_t child_stride, int64_t chunk_stride, int64_t cols_per_chunk,
                                                           const int64_t *__restrict__ row_off)
{
    typedef typename vec_of<S, NS>::type V;
    const int64_t s = ((int64_t)blockIdx.x * 256 + threadIdx.x) * NS;
    const int64_t z = z0 + blockIdx.z;
    int64_t ns = nr * E, obase = z * child_stride;
    if (row_off) {                                                             // ragged children (shots with different trace counts): child z
        ns = (row_off[z + 1] - row_off[z]) * E;                                // has its own row count (= its leading dimension) and writes
        obase = row_off[z] * E;                                                // its block of the slab directly (one column chunk only)
    }
    if (s >= ns) return;
    const S *A = (const S *)blocks[z].coeff;
    x += z * x_stride;                                                         // tall: every child reads m (stride 0); wide: child z reads m_z
    const int64_t c0 = (int64_t)blockIdx.y * cols_per_chunk;
    const int64_t c1 = c0 + cols_per_chunk < nc ? c0 + cols_per_chunk : nc;
    const V acc = gemv_rows_walk<S, E, NS, V>(A + c0 * ns + s, ns, x, c0, c1);  // columns in order, product rounded then added
    stg(reinterpret_cast<V *>(out + obase + (int64_t)blockIdx.y * chunk_stride + s), acc);
}

// y[z][k] = sum over column chunks (in order) of partial[z][chunk][k]
template <typename S>
__global__ void k_sum_chunks_batched(const S *__restrict__ partial, int64_t ns, int nchunks, S *__restrict__ y)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= ns) return;
    const int64_t z = blockIdx.y;
    const S *p = partial + z * nchunks * ns + k;
    S acc = p[0];
    for (int c = 1; c < nchunks; c++) acc = acc + p[(int64_t)c * ns];
    y[z * ns + k] = acc;
}

// partial[z][chunk][c] = sum over the chunk's rows of conj(A_z[r,c]) * d_z[r]   (fp64, one wave per column)
template <typename S, int E, int NS>
__global__ __launch_bounds__(256) void k_gemv_cols_batched(const jh_dev_block *__restrict__ blocks, int64_t z0, int64_t nr, int64_t nc,
                                                           const S *__restrict__ d, int64_t d_stride, double *__restrict__ partial,
                                                           int64_t rows_per_chunk, const int64_t *__restrict__ row_off)
{
    typedef typename vec_of<S, NS>::type V;
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= nc) return;
    const int lane = threadIdx.x & 63;
    const int64_t z = z0 + blockIdx.z;
    int64_t ns = nr * E, dbase = z * d_stride;
    if (row_off) {                                                             // ragged children: own row count, own block of d
        ns = (row_off[z + 1] - row_off[z]) * E;
        dbase = row_off[z] * E;
    }
    const int64_t s0 = (int64_t)blockIdx.y * rows_per_chunk * E;
    const int64_t s1 = s0 + rows_per_chunk * E < ns ? s0 + rows_per_chunk * E : ns;
    const S *col = (const S *)blocks[z].coeff + c * ns;
    const S *x = d + dbase;                                                    // tall: child z reads d_z; wide: every child reads d (stride 0)
    double sr = 0.0, si = 0.0;
    // (columns need not be whole, 16-byte aligned packs: under-aligned loads, the chunk's last pack from s1 - NS -- the host guarantees ns >= NS)
    for (int64_t s = s0 + (int64_t)lane * NS; s < s1; s += 64 * NS) {
        const int64_t sc = s + NS <= s1 ? s : s1 - NS;
        V a = ldgu_nt<S, NS>(col + sc);
        V xv = ldgu<S, NS>(x + sc);
        cols_accumulate<S, E, NS, V>(a, xv, (int)(s - sc), sr, si);              // conj(a) * x
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sr += __shfl_down(sr, off, 64);
        if (E == 2) si += __shfl_down(si, off, 64);
    }
    if (lane == 0) {
        double *p = partial + (((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * nc + c) * 2;
        p[0] = sr;
        p[1] = si;
    }
}

// The same for SMALL uniform children whose column fits one pass of (part of) a wave (<= 64 packs): `lpc` lanes per column (a power
// of two), 64/lpc columns per wave, and CH children per wave with all their loads in flight before the reductions -- a quarter of
// the waves, each with real memory-level parallelism (4096 children of 256^2: 0.29 -> see profiles/bench_dense_blocks_r01.txt).
template <typename S, int E, int NS, int CH>
__global__ __launch_bounds__(256) void k_gemv_cols_small(const jh_dev_block *__restrict__ blocks, int64_t z0, int nchild, int64_t nr, int64_t nc,
                                                         const S *__restrict__ d, int64_t d_stride, double *__restrict__ partial, int lpc)
{
    typedef typename vec_of<S, NS>::type V;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cpw = 64 / lpc, sub = lane % lpc, which = lane / lpc;
    const int64_t c = ((int64_t)blockIdx.x * 4 + wave) * cpw + which;
    const int64_t ns = nr * E, s = (int64_t)sub * NS;
    const bool live = c < nc && s < ns;
    const int zl0 = (int)blockIdx.z * CH;
    V a[CH], xv[CH];
#pragma unroll
    for (int ch = 0; ch < CH; ch++) {
        a[ch] = (V)(S)0;
        xv[ch] = (V)(S)0;
        if (live && zl0 + ch < nchild) {
            const int64_t z = z0 + zl0 + ch;
            a[ch] = ldg_nt(reinterpret_cast<const V *>((const S *)blocks[z].coeff + c * ns + s));
            xv[ch] = ldg(reinterpret_cast<const V *>(d + z * d_stride + s));
        }
    }
#pragma unroll
    for (int ch = 0; ch < CH; ch++) {
        double sr = 0.0, si = 0.0;
        if constexpr (E == 1) {
#pragma unroll
            for (int e = 0; e < NS; e++) {
                if constexpr (NS == 1) sr += (double)a[ch] * (double)xv[ch];
                else sr += (double)a[ch][e] * (double)xv[ch][e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < NS; e += 2) {
                const double ar = a[ch][e], ai = -(double)a[ch][e + 1], xr = xv[ch][e], xi = xv[ch][e + 1];
                sr += ar * xr - ai * xi;
                si += ar * xi + ai * xr;
            }
        }
        for (int off = lpc >> 1; off > 0; off >>= 1) {
            sr += __shfl_down(sr, off, lpc);
            if (E == 2) si += __shfl_down(si, off, lpc);
        }
        if (sub == 0 && c < nc && zl0 + ch < nchild) {
            double *p = partial + ((int64_t)(zl0 + ch) * nc + c) * 2;
            p[0] = sr;
            p[1] = si;
        }
    }
}

template <typename S, int E, int NS>
static bool launch_cols_small(const jh_dev_block *dev_blocks, int64_t z0, int64_t gz, int64_t nr, int64_t nc, const S *d, int64_t d_stride,
                              double *partial, hipStream_t st)
{
    const int64_t packs = (nr * E + NS - 1) / NS;
    if (packs > 64 || (nr * E) % NS != 0) return false;
    int lpc = 1;
    while (lpc < packs) lpc *= 2;
    const int cpw = 64 / lpc;
    constexpr int CH = 4;
    const int64_t gx = (nc + 4 * cpw - 1) / (4 * cpw), gzz = (gz + CH - 1) / CH;
    if (gx > 65535 * 32768ll || gzz > 65535) return false;
    hipLaunchKernelGGL((k_gemv_cols_small<S, E, NS, CH>), dim3((unsigned)gx, 1, (unsigned)gzz), dim3(256), 0, st, dev_blocks, z0, (int)gz, nr, nc, d,
                       d_stride, partial, lpc);
    return true;
}

// group sums: group g = children [g*per_group, (g+1)*per_group) of this launch: mtmp_z = A_z' d_z (its row chunks added, then
// rounded to the element type like the per-child kernel does, 1049) summed over the group's children in fp64 -- 64 column lanes
// x 4 child lanes per workgroup, fixed order; k_fold_groups adds the groups.  Two stages so that thousands of small children
// do not queue behind nc/64 workgroups.
template <typename S, int E>
__global__ __launch_bounds__(256) void k_fold_children(const double *__restrict__ partial, int64_t nc, int nchunks, int nchild, int per_group,
                                                       double *__restrict__ group_sums)
{
    __shared__ double sm[4][64][2];
    const int v = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + v;
    const int zlo = (int)blockIdx.y * per_group, zhi = zlo + per_group < nchild ? zlo + per_group : nchild;
    double tr = 0.0, ti = 0.0;
    if (c < nc) {
        for (int z = zlo + q; z < zhi; z += 4) {
            double sr = 0.0, si = 0.0;
            const double *p = partial + ((int64_t)z * nchunks * nc + c) * 2;
            for (int k = 0; k < nchunks; k++) { sr += p[(int64_t)k * nc * 2]; si += p[(int64_t)k * nc * 2 + 1]; }
            tr += (double)(S)sr;
            if (E == 2) ti += (double)(S)si;
        }
    }
    sm[q][v][0] = tr;
    sm[q][v][1] = ti;
    __syncthreads();
    if (q == 0 && c < nc) {
        double r = 0.0, i = 0.0;
        for (int qq = 0; qq < 4; qq++) { r += sm[qq][v][0]; i += sm[qq][v][1]; }
        group_sums[((int64_t)blockIdx.y * nc + c) * 2] = r;
        group_sums[((int64_t)blockIdx.y * nc + c) * 2 + 1] = i;
    }
}

template <typename S, int E>
__global__ void k_fold_groups(const double *__restrict__ group_sums, int64_t nc, int ngroups, S *__restrict__ m, int accumulate)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= nc) return;
    double r = accumulate ? (double)m[c * E] : 0.0, i = (E == 2 && accumulate) ? (double)m[c * E + 1] : 0.0;
    for (int g = 0; g < ngroups; g++) { r += group_sums[((int64_t)g * nc + c) * 2]; i += group_sums[((int64_t)g * nc + c) * 2 + 1]; }
    m[c * E] = (S)r;
    if (E == 2) m[c * E + 1] = (S)i;
}

// ---- the adjoint of MANY SMALL uniform children in ONE streaming kernel + a small fold (round 4) -----------------------------------
// m = sum_z A_z' d_z over thousands of 128^2 ... 2048^2 children used to take three launches (k_gemv_cols_small / _batched, then
// k_fold_children, then k_fold_groups): a wave lived for 4 KiB of matrix, every (child, column) sum went through six dependent
// fp64 shuffles and left the kernel as a 16-byte partial, and the chip ran at 41 % of its HBM rate on 4096 x 256^2.  Here
//   * a child's matrix is one flat stream of 16-byte packs (column c = packs [c P, (c+1) P), P = nr E / NS); a UNIT is one wave
//     load (64 packs), a BATCH is B units = B KiB of one child = 64 B / L whole columns (L = P rounded up to a power of two; for
//     P = L a batch is contiguous memory);
//   * a WAVE owns one column batch of Gw consecutive children: per child it has B matrix loads in flight (plus the child's pack
//     of d), forms each lane's exact fp64 products, and reduces the B x 64 lane sums through a padded LDS image read TRANSPOSED
//     (lane r adds the B consecutive entries [r B, (r+1) B): two LDS instructions per unit instead of twelve shuffles), then a
//     log2(L / B)-step butterfly; the child's column sum is rounded to the element type where the reference holds it in `mtmp`
//     (1049) and added to the wave's fp64 running sum -- the children of a wave in order;
//   * the four waves of a workgroup own four consecutive child quarters of one group: their sums are added in wave order through
//     LDS and ONE fp64 partial per (group, column) leaves the kernel, stored [column][group] so that the fold (one wave per
//     column, coalesced, fixed tree) reads it contiguously.
// Deterministic (no atomics; the order is a function of the shape alone), tolerance parity like every dense adjoint (1e-6 / 1e-14).
// DIRECT = the wide operator's m_j = A_j' d (1051): no sum over children, a wave owns (child, column batch) and stores the
// rounded sums itself.  Workgroup ids are decoded XCD-aware: the workgroups of one group (they read the same d_z) share an XCD.
template <typename S, int E, int NS, int B, int NPX>
__global__ __launch_bounds__(256) void k_gemv_cols_fused(const jh_dev_block *__restrict__ blocks, int64_t nchild, int G, int P, int lsh, int64_t nc,
                                                         int nb, const S *__restrict__ d, int64_t d_stride, double *__restrict__ partial,
                                                         int64_t ngroups, S *__restrict__ out, int direct)
{
    typedef typename vec_of<S, NS>::type V;
    constexpr int LB = B == 16 ? 4 : 3;                                        // log2 B
    constexpr int ROW = 64 + 64 / B;                                           // doubles per unit in the padded image (one pad per B entries)
    constexpr int PLANE = B * ROW;                                             // doubles per wave (and per real / imaginary plane)
    __shared__ double scr[4 * E * PLANE];
    __shared__ double comb[4][64][2];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int L = 1 << lsh, cpb = (B * 64) >> lsh;                             // columns per batch
    int64_t cb, g = 0, zlo, zhi;
    if (direct) {
        const int64_t item = (int64_t)blockIdx.x * 4 + wave;
        cb = item % nb;
        zlo = item / nb;
        zhi = zlo + 1 < nchild ? zlo + 1 : nchild;                             // an item past the end walks no child
    } else {
        const int64_t ng8 = ngroups & ~(int64_t)7, bid = blockIdx.x;           // whole octets of groups: group g on XCD g mod 8 (ids 8 apart share an XCD);
        if (bid < ng8 * nb) {                                                  // the last few groups (and operators of fewer than 8) in plain order, over all XCDs
            const int64_t t = bid >> 3;
            cb = t % nb;
            g = (t / nb) * 8 + (bid & 7);
        } else {
            const int64_t r = bid - ng8 * nb;
            cb = r % nb;
            g = ng8 + r / nb;
        }
        const int Gw = (G + 3) >> 2;
        zlo = g * G + (int64_t)wave * Gw;
        zhi = zlo + Gw;
        if (zhi > (g + 1) * G) zhi = (g + 1) * G;
        if (zhi > nchild) zhi = nchild;
    }
    // what this lane loads in unit u: entry e = 64 u + lane of the batch = pack (e mod L) of column (e / L); the same for every child
    uint32_t off[B];
    uint32_t livemask = 0;
#pragma unroll
    for (int u = 0; u < B; u++) {
        const int e = u * 64 + lane, pe = e & (L - 1);
        const int64_t c = cb * cpb + (e >> lsh);
        const bool live = pe < P && c < nc;
        off[u] = live ? (uint32_t)((c * P + pe) * NS) : 0u;                    // a dead lane re-reads the first pack (branch-free load section)
        livemask |= (live ? 1u : 0u) << u;
    }
    uint32_t xo[NPX];
#pragma unroll
    for (int i = 0; i < NPX; i++) {
        const int pe = (i * 64 + lane) & (L - 1);
        xo[i] = pe < P ? (uint32_t)(pe * NS) : 0u;
    }
    double *img = scr + wave * (E * PLANE);
    const int wr = lane + (lane >> LB);                                        // this lane's slot in a unit's row of the image
    const int rd = lane * (B + 1);                                             // first of the B consecutive entries this lane adds
    const int lp = lsh - LB;                                                   // log2 (lanes holding one column's run sums)
    double accr = 0.0, acci = 0.0;
    for (int64_t z = zlo; z < zhi; z++) {
        const S *Az = (const S *)blocks[z].coeff;
        const S *xz = d + z * d_stride;
        V a[B];
#pragma unroll
        for (int u = 0; u < B; u++) a[u] = ldg_nt(reinterpret_cast<const V *>(Az + off[u]));
        V xv[NPX];
#pragma unroll
        for (int i = 0; i < NPX; i++) xv[i] = ldg(reinterpret_cast<const V *>(xz + xo[i]));
#pragma unroll
        for (int u = 0; u < B; u++) {
            const V x = xv[u % NPX];                                           // unit u holds piece u mod (L / 64) of its column
            double pr = 0.0, pi = 0.0;
            if constexpr (E == 1) {
#pragma unroll
                for (int e = 0; e < NS; e++) {
                    if constexpr (NS == 1) pr += (double)a[u] * (double)x;
                    else if constexpr (sizeof(S) == 4) pr = __builtin_fma((double)a[u][e], (double)x[e], pr);   // the product of two floats is exact in fp64: the same value as multiply-then-add
                    else pr += (double)a[u][e] * (double)x[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < NS; e += 2) {                               // conj(a) * x
                    const double ar = a[u][e], ai = -(double)a[u][e + 1], xr = x[e], xi = x[e + 1];
                    pr += ar * xr - ai * xi;
                    pi += ar * xi + ai * xr;
                }
            }
            if (!((livemask >> u) & 1u)) { pr = 0.0; pi = 0.0; }
            img[u * ROW + wr] = pr;
            if constexpr (E == 2) img[PLANE + u * ROW + wr] = pi;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                 // the image is this wave's own: LDS operations of one wave complete in order
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double rr = 0.0, ri = 0.0;
#pragma unroll
        for (int k = 0; k < B; k++) {
            rr += img[rd + k];
            if constexpr (E == 2) ri += img[PLANE + rd + k];
        }
        for (int s = 0; s < lp; s++) {                                          // the L / B lanes of one column: fixed butterfly
            rr += __shfl_xor(rr, 1 << s, 64);
            if constexpr (E == 2) ri += __shfl_xor(ri, 1 << s, 64);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                 // the next child's stores come after these reads
        __builtin_amdgcn_wave_barrier();
        if (direct) {
            const int64_t c = cb * cpb + (lane >> lp);
            if ((lane & ((1 << lp) - 1)) == 0 && c < nc) {
                out[(z * nc + c) * E] = (S)rr;
                if constexpr (E == 2) out[(z * nc + c) * E + 1] = (S)ri;
            }
        } else {
            accr += (double)(S)rr;                                             // mtmp is an array of the element type (1049)
            if constexpr (E == 2) acci += (double)(S)ri;
        }
    }
    if (direct) return;
    if ((lane & ((1 << lp) - 1)) == 0) {
        comb[wave][lane >> lp][0] = accr;
        comb[wave][lane >> lp][1] = acci;
    }
    __syncthreads();
    if (wave == 0 && lane < cpb) {
        const int64_t c = cb * cpb + lane;
        if (c < nc) {
            double r = comb[0][lane][0], i = comb[0][lane][1];
#pragma unroll
            for (int w = 1; w < 4; w++) { r += comb[w][lane][0]; i += comb[w][lane][1]; }
            double *p = partial + (c * ngroups + g) * E;
            p[0] = r;
            if constexpr (E == 2) p[1] = i;
        }
    }
}

// m[c] = sum over the groups (fixed order: 64 interleaved lane sums, then the wave's butterfly) of partial[c][g]; one wave per column
template <typename S, int E>
__global__ __launch_bounds__(256) void k_fold_fused(const double *__restrict__ partial, int64_t nc, int64_t ngroups, S *__restrict__ m, int add_found = 0)
{
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= nc) return;
    const double *p = partial + c * ngroups * E;
    double r = 0.0, i = 0.0;
    for (int64_t g = lane; g < ngroups; g += 64) {
        r += p[g * E];
        if constexpr (E == 2) i += p[g * E + 1];
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        r += __shfl_xor(r, s, 64);
        if constexpr (E == 2) i += __shfl_xor(i, s, 64);
    }
    if (lane == 0) {
        if (add_found) {                                                       // the wide forward: `_d .+=` into d as found (1024)
            r += (double)m[c * E];
            if constexpr (E == 2) i += (double)m[c * E + 1];
        }
        m[c * E] = (S)r;
        if constexpr (E == 2) m[c * E + 1] = (S)i;
    }
}

// ---- the FORWARD of a wide operator of MANY SMALL children in one streaming kernel + the same fold (round 4) -------------------------------
// d = d_found + sum_j A_j m_j over thousands of small children (1 x 16384 of 128^2): three launches and a child-sized slab per child before
// (k_gemv_rows_batched into T, k_fold_wide_groups, k_fold_wide_final: 5.0-5.1 TB/s).  Here a lane owns one 16-byte pack of ROWS of one child
// (lpc lanes per child, 64 / lpc children per wave side by side), walks that child's columns in order with sixteen loads in flight -- the
// sequential loop's bits for every child's product, rounded to the element type like the reference's dtmp (1024) -- and adds it to its
// fp64 running sum over the wave's Gw children; the lane groups of a wave, then the four waves of a workgroup are combined in a fixed order
// and ONE fp64 partial per (group, scalar row) leaves the kernel, [row][group], for k_fold_fused (which adds d as found).  Tolerance parity,
// as the fp64 group sums of the three-launch path were (wide operators of up to 64 children keep the reference's order: k_fold_wide_ordered).
template <typename S, int E, int NS>
__global__ __launch_bounds__(256) void k_gemv_rows_wide_fused(const jh_dev_block *__restrict__ blocks, int64_t nchild, int G, int P, int lsh, int64_t nc,
                                                              const S *__restrict__ x, double *__restrict__ partial, int64_t ngroups)
{
    typedef typename vec_of<S, NS>::type V;
    __shared__ double comb[4][64 * NS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lpc = 1 << lsh, cpw = 64 >> lsh, sub = lane & (lpc - 1), which = lane >> lsh;
    const int64_t g = blockIdx.x;
    const int Gw = (G + 3) >> 2;
    const int64_t zlo = g * G + (int64_t)wave * Gw;
    int64_t zhi = zlo + Gw;
    if (zhi > (g + 1) * G) zhi = (g + 1) * G;
    if (zhi > nchild) zhi = nchild;
    const int64_t ns = (int64_t)P * NS;                                        // scalar rows of a child
    double gacc[NS];
#pragma unroll
    for (int e = 0; e < NS; e++) gacc[e] = 0.0;
    for (int64_t zb = zlo; zb < zhi; zb += cpw) {
        const int64_t z = zb + which;
        const bool ok = z < zhi && sub < P;
        const int64_t zz = ok ? z : zlo;                                       // an idle lane walks the wave's first child again (its sum is dropped)
        const S *col = (const S *)blocks[zz].coeff + (ok ? sub * NS : 0);
        const V acc = gemv_rows_walk<S, E, NS, V>(col, ns, x + zz * nc * E, 0, nc);
        if (ok) {
#pragma unroll
            for (int e = 0; e < NS; e++) gacc[e] += (double)acc[e];            // mul!(dtmp, A_j, m_j) is an array of the element type (1024)
        }
    }
#pragma unroll
    for (int e = 0; e < NS; e++)
        for (int s = lpc; s < 64; s <<= 1) gacc[e] += __shfl_xor(gacc[e], s, 64);   // the children a wave walked side by side
    if (which == 0 && sub < P) {
#pragma unroll
        for (int e = 0; e < NS; e++) comb[wave][sub * NS + e] = gacc[e];
    }
    __syncthreads();
    if (wave == 0 && which == 0 && sub < P) {
#pragma unroll
        for (int e = 0; e < NS; e++) {
            const int k = sub * NS + e;
            partial[(int64_t)k * ngroups + g] = ((comb[0][k] + comb[1][k]) + comb[2][k]) + comb[3][k];
        }
    }
}

// *launched = false: not a shape for this kernel (tiny or long columns) -- the caller's older path runs.
template <typename S, int E, int NS>
static int launch_cols_fused(const jh_dev_block *dev_blocks, int64_t nchild, int64_t nr, int64_t nc, const S *d, int64_t d_stride, S *y, bool direct,
                             bool *launched)
{
    jh_context &c = jh_ctx();
    hipStream_t st = c.stream;
    *launched = false;
    constexpr int B = E == 2 ? 8 : 16;
    if ((nr * E) % NS != 0) return JH_OK;
    const int64_t P = nr * E / NS;
    // columns of B .. 256 packs (Float32: 64 .. 1024 rows).  Longer columns stream as well on the wave-per-column kernels (measured,
    // profiles/bench_dense_blocks_r04.txt: 2048^2 and 4096^2 children 5.5 against 6.6 TB/s), shorter ones are left to k_gemv_cols_small
    if (P < B || P > 256 || nc < 1 || nchild < 1) return JH_OK;
    if ((double)nr * (double)nc * E * sizeof(S) >= 2147483648.0) return JH_OK;   // 32-bit offsets inside a child
    int lsh = 0;
    while ((1 << lsh) < P) lsh++;
    const int64_t L = (int64_t)1 << lsh, cpb = (B * 64) / L, nb = (nc + cpb - 1) / cpb;
    if (nb > (1 << 30)) return JH_OK;
    const int npx = L <= 64 ? 1 : (int)(L / 64);                               // packs of d_z a lane keeps per child: 1, 2 or 4
    // children per wave: enough waves to fill the chip several times over, short-lived (workgroups that move one batch and exit
    // stream best on this chip), but not so many groups that the fold matters
    int64_t gw = c.dense_gw > 0 ? c.dense_gw : nchild / 1024;                  // about 256 groups (measured: profiles/bench_dense_blocks_r04.txt)
    if (gw < 1) gw = 1;
    if (gw * 4 > nchild) gw = (nchild + 3) / 4;
    int64_t G = direct ? 1 : gw * 4, ngroups = direct ? 0 : (nchild + G - 1) / G;
    int64_t wgs;
    if (direct) wgs = (nchild * nb + 3) / 4;
    else {
        JH_TRY(jh_ensure_partials(nc * ngroups * E + 2));
        wgs = ngroups * nb;
    }
    if (wgs > 2000000000ll) return JH_OK;
#define JH_FUSED(NPXV)                                                                                                                              \
    hipLaunchKernelGGL((k_gemv_cols_fused<S, E, NS, B, NPXV>), dim3((unsigned)wgs), dim3(256), 0, st, dev_blocks, nchild, (int)G, (int)P, lsh, nc, \
                       (int)nb, d, d_stride, c.part_dev, ngroups, y, direct ? 1 : 0)
    switch (npx) {
    case 1: JH_FUSED(1); break;
    case 2: JH_FUSED(2); break;
    default: JH_FUSED(4); break;
    }
#undef JH_FUSED
    JH_CHECK_HIP(hipGetLastError());
    if (!direct) {
        hipLaunchKernelGGL((k_fold_fused<S, E>), dim3((unsigned)((nc + 3) / 4)), dim3(256), 0, st, c.part_dev, nc, ngroups, y);
        JH_CHECK_HIP(hipGetLastError());
    }
    *launched = true;
    return JH_OK;
}

// ---- WIDE operator (one block row of K dense children): d (+)= sum_j A_j m_j and m_j = A_j' d -----------------------------
// forward fold, few children: the reference's order and rounding -- `_d .+= mul!(dtmp, A_j, m_j)` (1024): dtmp_j is the ordered
// sum of its column chunks in the element type, added to d as found, j in order
template <typename S>
__global__ void k_fold_wide_ordered(const S *__restrict__ T, int64_t ns, int nchunks, int nchild, S *__restrict__ d)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= ns) return;
    S acc = d[k];
    for (int j = 0; j < nchild; j++) {
        const S *p = T + (int64_t)j * nchunks * ns + k;
        S t = p[0];
        for (int c = 1; c < nchunks; c++) t = t + p[(int64_t)c * ns];
        acc = acc + t;
    }
    d[k] = acc;
}

// forward fold, many children: group sums in fp64 (64 scalar lanes x 4 child lanes), then d = found + groups (tolerance parity)
template <typename S>
__global__ __launch_bounds__(256) void k_fold_wide_groups(const S *__restrict__ T, int64_t ns, int nchunks, int nchild, int per_group,
                                                          double *__restrict__ group_sums)
{
    __shared__ double sm[4][64];
    const int v = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t k = (int64_t)blockIdx.x * 64 + v;
    const int jlo = (int)blockIdx.y * per_group, jhi = jlo + per_group < nchild ? jlo + per_group : nchild;
    double acc = 0.0;
    if (k < ns)
        for (int j = jlo + q; j < jhi; j += 4) {
            const S *p = T + (int64_t)j * nchunks * ns + k;
            S t = p[0];
            for (int c = 1; c < nchunks; c++) t = t + p[(int64_t)c * ns];
            acc += (double)t;
        }
    sm[q][v] = acc;
    __syncthreads();
    if (q == 0 && k < ns) group_sums[(int64_t)blockIdx.y * ns + k] = sm[0][v] + sm[1][v] + sm[2][v] + sm[3][v];
}

template <typename S>
__global__ void k_fold_wide_final(const double *__restrict__ group_sums, int64_t ns, int ngroups, S *__restrict__ d)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= ns) return;
    double acc = (double)d[k];
    for (int g = 0; g < ngroups; g++) acc += group_sums[(int64_t)g * ns + k];
    d[k] = (S)acc;
}

// adjoint: m_z[c] = sum over the row chunks of partial[z][chunk][c]  (one block row: written directly, 1051)
template <typename S, int E>
__global__ void k_store_col_sums(const double *__restrict__ partial, int64_t nc, int nchunks, S *__restrict__ m)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= nc) return;
    const int64_t z = blockIdx.y;
    double sr = 0.0, si = 0.0;
    const double *p = partial + (z * nchunks * nc + c) * 2;
    for (int k = 0; k < nchunks; k++) { sr += p[(int64_t)k * nc * 2]; si += p[(int64_t)k * nc * 2 + 1]; }
    m[(z * nc + c) * E] = (S)sr;
    if (E == 2) m[(z * nc + c) * E + 1] = (S)si;
}

template <typename S, int E>
int gemv_batched_wide(const jh_dev_block *dev_blocks, int64_t nchild, int64_t nr, int64_t nc, void *y, const void *x, int adjoint, bool aligned)
{
    jh_context &c = jh_ctx();
    hipStream_t st = c.stream;
    constexpr int NSV = (16 / sizeof(S)) >= E ? (16 / sizeof(S)) : E;
    const int64_t ns = nr * E;
    const bool vec_ok = aligned && ((ns * (int64_t)sizeof(S)) % 16 == 0) && ((((uintptr_t)x) | ((uintptr_t)y)) & 15u) == 0;
    const double child_bytes = (double)nr * (double)nc * sizeof(S) * E;
    JH_REQUIRE(nchild <= 32768, "batched wide dense operator: too many children");
    if (!adjoint) {                                                            // d (+)= sum_j A_j m_j
        c.last_dense_fused = 0;
        if (c.dense_fused && vec_ok && nchild > 64 && ns % NSV == 0 && ns / NSV <= 64 && nc >= 1 && child_bytes < (double)(1 << 20)) {
            // many small children: one streaming kernel + the fold (see k_gemv_rows_wide_fused)
            const int P = (int)(ns / NSV);
            int lsh = 0;
            while ((1 << lsh) < P) lsh++;
            const int cpw = 64 >> lsh;
            int64_t gw = c.dense_gw > 0 ? c.dense_gw : nchild / 4096;            // about 1024 groups
            if (gw < cpw) gw = cpw;
            gw = (gw + cpw - 1) / cpw * cpw;                                     // whole wave loads of children
            const int64_t G = gw * 4, ngroups = (nchild + G - 1) / G;
            JH_TRY(jh_ensure_partials(ns * ngroups + 2));
            hipLaunchKernelGGL((k_gemv_rows_wide_fused<S, E, NSV>), dim3((unsigned)ngroups), dim3(256), 0, st, dev_blocks, nchild, (int)G, P, lsh, nc,
                               (const S *)x, c.part_dev, ngroups);
            JH_CHECK_HIP(hipGetLastError());
            hipLaunchKernelGGL((k_fold_fused<S, 1>), dim3((unsigned)((ns + 3) / 4)), dim3(256), 0, st, (const double *)c.part_dev, ns, ngroups, (S *)y, 1);
            JH_CHECK_HIP(hipGetLastError());
            c.last_dense_fused = 1;
            return JH_OK;
        }
        const int NS = vec_ok ? NSV : E;
        const int64_t row_wgs = (ns / NS + 255) / 256;
        int64_t nchunks = 1;
        const int64_t want_wgs = c.dense_fwd_wgs > 0 ? c.dense_fwd_wgs : 512;   // (see gemv_batched)
        if (child_bytes >= (double)(1 << 20) && row_wgs * nchild < want_wgs) {
            nchunks = (want_wgs + row_wgs * nchild - 1) / (row_wgs * nchild);
            const int64_t maxc = (nc + 31) / 32;
            if (nchunks > maxc) nchunks = maxc;
        }
        const int64_t cpc = (nc + nchunks - 1) / nchunks;
        nchunks = (nc + cpc - 1) / cpc;
        const int64_t t_doubles = ((int64_t)nchild * nchunks * ns * (int64_t)sizeof(S) + 7) / 8 + 2;
        const int64_t fold_wgs = (ns + 63) / 64;
        // once the columns are split the sum is tolerance-level anyway: fold all (child, chunk) partial rows alike, in fp64
        const bool ordered = nchild <= 64 && nchunks == 1;
        const int64_t nitems = nchild * nchunks;
        int64_t ngroups = (2048 + fold_wgs - 1) / fold_wgs;
        if (ngroups > nitems / 8) ngroups = nitems / 8;
        { int64_t r = 1; while ((r + 1) * (r + 1) <= nitems) r++; if (ngroups > r) ngroups = r; }
        if (ngroups < 1) ngroups = 1;
        const int64_t per_group = (nitems + ngroups - 1) / ngroups;
        ngroups = (nitems + per_group - 1) / per_group;
        JH_TRY(jh_ensure_partials(t_doubles + ngroups * ns));
        S *T = (S *)c.part_dev;
        double *group_sums = c.part_dev + t_doubles;
        if (vec_ok)
            hipLaunchKernelGGL((k_gemv_rows_batched<S, E, NSV>), dim3((unsigned)row_wgs, (unsigned)nchunks, (unsigned)nchild), dim3(256), 0, st, dev_blocks,
                               (int64_t)0, nr, nc, (const S *)x, nc * E, T, nchunks * ns, ns, cpc, (const int64_t *)nullptr);
        else
            hipLaunchKernelGGL((k_gemv_rows_batched<S, E, E>), dim3((unsigned)row_wgs, (unsigned)nchunks, (unsigned)nchild), dim3(256), 0, st, dev_blocks,
                               (int64_t)0, nr, nc, (const S *)x, nc * E, T, nchunks * ns, ns, cpc, (const int64_t *)nullptr);
        JH_CHECK_HIP(hipGetLastError());
        if (ordered) {
            hipLaunchKernelGGL((k_fold_wide_ordered<S>), dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, st, T, ns, (int)nchunks, (int)nchild, (S *)y);
        } else {
            hipLaunchKernelGGL((k_fold_wide_groups<S>), dim3((unsigned)fold_wgs, (unsigned)ngroups), dim3(256), 0, st, T, ns, 1, (int)nitems,
                               (int)per_group, group_sums);
            JH_CHECK_HIP(hipGetLastError());
            hipLaunchKernelGGL((k_fold_wide_final<S>), dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, st, group_sums, ns, (int)ngroups, (S *)y);
        }
        JH_CHECK_HIP(hipGetLastError());
        return JH_OK;
    }
    // m_j = A_j' d for every child
    c.last_dense_fused = 0;
    if (c.dense_fused) {
        bool launched = false;
        if (vec_ok) JH_TRY((launch_cols_fused<S, E, NSV>(dev_blocks, nchild, nr, nc, (const S *)x, 0, (S *)y, true, &launched)));
        else JH_TRY((launch_cols_fused<S, E, E>(dev_blocks, nchild, nr, nc, (const S *)x, 0, (S *)y, true, &launched)));
        if (launched) { c.last_dense_fused = 1; return JH_OK; }
    }
    const int64_t col_wgs = (nc + 3) / 4;
    int64_t nchunks = 1;
    if (child_bytes >= (double)(1 << 20) && col_wgs * nchild < 2048) {
        nchunks = (2048 + col_wgs * nchild - 1) / (col_wgs * nchild);
        const int64_t maxc = (nr + 4095) / 4096;
        if (nchunks > maxc) nchunks = maxc;
        if (nchunks < 1) nchunks = 1;
    }
    int64_t rpc = (nr + nchunks - 1) / nchunks;
    rpc = (rpc + 3) / 4 * 4;
    if (rpc < 4) rpc = 4;
    nchunks = nr ? (nr + rpc - 1) / rpc : 1;
    JH_TRY(jh_ensure_partials(2 * nchild * nchunks * nc));
    if (nchunks == 1 && (vec_ok ? launch_cols_small<S, E, NSV>(dev_blocks, 0, nchild, nr, nc, (const S *)x, 0, c.part_dev, st)
                                : launch_cols_small<S, E, E>(dev_blocks, 0, nchild, nr, nc, (const S *)x, 0, c.part_dev, st))) {
    } else if (vec_ok || (c.tall_unaligned != 0 && ns >= NSV && ((((uintptr_t)x) | ((uintptr_t)y)) & (sizeof(S) - 1)) == 0))   // (columns off the 16-byte grid: under-aligned packs)
        hipLaunchKernelGGL((k_gemv_cols_batched<S, E, NSV>), dim3((unsigned)col_wgs, (unsigned)nchunks, (unsigned)nchild), dim3(256), 0, st, dev_blocks,
                           (int64_t)0, nr, nc, (const S *)x, (int64_t)0, c.part_dev, rpc, (const int64_t *)nullptr);
    else
        hipLaunchKernelGGL((k_gemv_cols_batched<S, E, E>), dim3((unsigned)col_wgs, (unsigned)nchunks, (unsigned)nchild), dim3(256), 0, st, dev_blocks,
                           (int64_t)0, nr, nc, (const S *)x, (int64_t)0, c.part_dev, rpc, (const int64_t *)nullptr);
    JH_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL((k_store_col_sums<S, E>), dim3((unsigned)((nc + 255) / 256), (unsigned)nchild), dim3(256), 0, st, c.part_dev, nc, (int)nchunks, (S *)y);
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

// row_off != nullptr: ragged children -- `nr` is then the LARGEST row count (grid sizing), every child's own count comes from the table
template <typename S, int E>
int gemv_batched(const jh_dev_block *dev_blocks, int64_t nchild, int64_t nr, int64_t nc, void *y, const void *x, int adjoint, bool aligned,
                 const int64_t *row_off = nullptr)
{
    jh_context &c = jh_ctx();
    hipStream_t st = c.stream;
    constexpr int NSV = (16 / sizeof(S)) >= E ? (16 / sizeof(S)) : E;
    const int64_t ns = nr * E;
    const bool vec_ok = aligned && ((ns * (int64_t)sizeof(S)) % 16 == 0) && ((((uintptr_t)x) | ((uintptr_t)y)) & 15u) == 0;
    const double child_bytes = (double)nr * (double)nc * sizeof(S) * E;
    const int64_t zmax = 32768;                                                // gridDim.z
    if (!adjoint) {                                                            // d_z = A_z m for every child
        const int NS = vec_ok ? NSV : E;
        const int64_t row_wgs = (ns / NS + 255) / 256;
        int64_t nchunks = 1;                                                   // split the columns only while the chip is not full
        // columns are split over grid.y only until about 512 workgroups exist: with sixteen loads in flight per lane that fills the chip,
        // and every further chunk is another partial row to write and to fold (round 4 sweep, profiles/bench_dense_blocks_r04.txt:
        // 4 x 8192^2 children 4.6 -> 5.8 TB/s, 16 x 4096^2 5.3 -> 6.1 against the 2048 of rounds 1-3); knob dense_fwd_wgs
        const int64_t want_wgs = c.dense_fwd_wgs > 0 ? c.dense_fwd_wgs : 512;
        if (!row_off && child_bytes >= (double)(1 << 20) && row_wgs * nchild < want_wgs) {
            nchunks = (want_wgs + row_wgs * nchild - 1) / (row_wgs * nchild);
            const int64_t maxc = (nc + 31) / 32;
            if (nchunks > maxc) nchunks = maxc;
        }
        const int64_t cpc = (nc + nchunks - 1) / nchunks;
        nchunks = (nc + cpc - 1) / cpc;
        S *out = (S *)y;
        int64_t child_stride = ns, chunk_stride = 0;
        if (nchunks > 1) {
            JH_TRY(jh_ensure_partials(((int64_t)nchild * nchunks * ns * (int64_t)sizeof(S) + 7) / 8 + 2));
            out = (S *)c.part_dev;
            child_stride = nchunks * ns;
            chunk_stride = ns;
        }
        for (int64_t z0 = 0; z0 < nchild; z0 += zmax) {
            const unsigned gz = (unsigned)(nchild - z0 < zmax ? nchild - z0 : zmax);
            if (vec_ok)
                hipLaunchKernelGGL((k_gemv_rows_batched<S, E, NSV>), dim3((unsigned)row_wgs, (unsigned)nchunks, gz), dim3(256), 0, st, dev_blocks, z0,
                                   nr, nc, (const S *)x, (int64_t)0, out, child_stride, chunk_stride, cpc, row_off);
            else
                hipLaunchKernelGGL((k_gemv_rows_batched<S, E, E>), dim3((unsigned)row_wgs, (unsigned)nchunks, gz), dim3(256), 0, st, dev_blocks, z0,
                                   nr, nc, (const S *)x, (int64_t)0, out, child_stride, chunk_stride, cpc, row_off);
            JH_CHECK_HIP(hipGetLastError());
        }
        if (nchunks > 1) {
            JH_REQUIRE(nchild <= 65535, "batched dense forward: too many children for the chunk fold");
            hipLaunchKernelGGL((k_sum_chunks_batched<S>), dim3((unsigned)((ns + 255) / 256), (unsigned)nchild), dim3(256), 0, st, out, ns,
                               (int)nchunks, (S *)y);
            JH_CHECK_HIP(hipGetLastError());
        }
        return JH_OK;
    }
    // m = sum_z A_z' d_z  (the caller's m is overwritten: `_m .= 0` then `_m .+= mtmp`, 1042-1049)
    // round 5, session 3: children whose columns are not whole, 16-byte aligned packs (k x k with k odd) keep the 16-byte-per-lane column kernel
    // on under-aligned packs (k_gemv_cols_batched): 128 children of 1023^2 3.1 -> see profiles/bench_unaligned_r05.txt; knob tall_unaligned = 0: as before
    const bool vec_cols_ua = !vec_ok && c.tall_unaligned != 0 && !row_off && ns >= NSV && ((((uintptr_t)x) | ((uintptr_t)y)) & (sizeof(S) - 1)) == 0;
    c.last_dense_fused = 0;
    if (!row_off && c.dense_fused) {                                            // many small children: one streaming kernel + a fold (round 4)
        bool launched = false;
        if (vec_ok) JH_TRY((launch_cols_fused<S, E, NSV>(dev_blocks, nchild, nr, nc, (const S *)x, ns, (S *)y, false, &launched)));
        else JH_TRY((launch_cols_fused<S, E, E>(dev_blocks, nchild, nr, nc, (const S *)x, ns, (S *)y, false, &launched)));
        if (launched) { c.last_dense_fused = 1; return JH_OK; }
    }
    const int64_t col_wgs = (nc + 3) / 4;
    int64_t nchunks = 1;
    if (child_bytes >= (double)(1 << 20) && col_wgs * nchild < 2048) {
        nchunks = (2048 + col_wgs * nchild - 1) / (col_wgs * nchild);
        const int64_t maxc = (nr + 4095) / 4096;
        if (nchunks > maxc) nchunks = maxc;
        if (nchunks < 1) nchunks = 1;
    }
    int64_t rpc = (nr + nchunks - 1) / nchunks;
    rpc = (rpc + 3) / 4 * 4;
    if (rpc < 4) rpc = 4;
    nchunks = nr ? (nr + rpc - 1) / rpc : 1;
    int64_t zstep = zmax;                                                      // bound the fp64 partials to 64 MiB per launch
    while (zstep > 1 && (double)zstep * (double)nchunks * (double)nc * 16.0 > 64.0 * (double)(1 << 20)) zstep /= 2;
    for (int64_t z0 = 0; z0 < nchild; z0 += zstep) {
        const int64_t gz = nchild - z0 < zstep ? nchild - z0 : zstep;
        const int64_t fold_wgs = (nc + 63) / 64;
        int64_t ngroups = (2048 + fold_wgs - 1) / fold_wgs;                    // enough workgroups for the fold of many small children,
        if (ngroups > gz / 8) ngroups = gz / 8;
        { int64_t r = 1; while ((r + 1) * (r + 1) <= gz) r++; if (ngroups > r) ngroups = r; }   // but the second stage walks the groups serially: ~sqrt
        if (ngroups < 1) ngroups = 1;
        const int64_t per_group = (gz + ngroups - 1) / ngroups;
        ngroups = (gz + per_group - 1) / per_group;
        const int64_t npart = 2 * gz * nchunks * nc;
        JH_TRY(jh_ensure_partials(npart + 2 * ngroups * nc));
        double *group_sums = c.part_dev + npart;
        const S *d0 = (const S *)x;
        if (!row_off && nchunks == 1 && (vec_ok ? launch_cols_small<S, E, NSV>(dev_blocks, z0, gz, nr, nc, d0, ns, c.part_dev, st)
                                                : launch_cols_small<S, E, E>(dev_blocks, z0, gz, nr, nc, d0, ns, c.part_dev, st))) {
        } else if (vec_ok || vec_cols_ua)
            hipLaunchKernelGGL((k_gemv_cols_batched<S, E, NSV>), dim3((unsigned)col_wgs, (unsigned)nchunks, (unsigned)gz), dim3(256), 0, st, dev_blocks,
                               z0, nr, nc, d0, ns, c.part_dev, rpc, row_off);
        else
            hipLaunchKernelGGL((k_gemv_cols_batched<S, E, E>), dim3((unsigned)col_wgs, (unsigned)nchunks, (unsigned)gz), dim3(256), 0, st, dev_blocks,
                               z0, nr, nc, d0, ns, c.part_dev, rpc, row_off);
        JH_CHECK_HIP(hipGetLastError());
        hipLaunchKernelGGL((k_fold_children<S, E>), dim3((unsigned)fold_wgs, (unsigned)ngroups), dim3(256), 0, st, c.part_dev, nc, (int)nchunks,
                           (int)gz, (int)per_group, group_sums);
        JH_CHECK_HIP(hipGetLastError());
        hipLaunchKernelGGL((k_fold_groups<S, E>), dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, st, group_sums, nc, (int)ngroups, (S *)y,
                           z0 > 0 ? 1 : 0);
        JH_CHECK_HIP(hipGetLastError());
    }
    return JH_OK;
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
        const int64_t want_wgs = c.dense_fwd_wgs > 0 ? c.dense_fwd_wgs : 512;   // (see gemv_batched)
        if (bytes >= (double)(1 << 20) && row_wgs < want_wgs) {                // few rows: split the columns (tiny matrices stay one ordered sum)
            nchunks = (want_wgs + row_wgs - 1) / row_wgs;
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
    // (columns off the 16-byte grid -- k x k with k odd -- keep the 16-byte-per-lane kernel on under-aligned packs: k_gemv_cols)
    const bool vec_cols_ua = !vec_ok && c.tall_unaligned != 0 && ns >= NSV && ((((uintptr_t)A) | ((uintptr_t)x) | ((uintptr_t)y)) & (sizeof(S) - 1)) == 0;
    if (vec_ok || vec_cols_ua)
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


// ---- ALL dense children of an operator that mixes them with other kinds (round 3) ------------------------------------------------
// Operators with DENSE children that fit none of the uniform batches above -- dense next to diagonal / identity / zero blocks,
// children of different shapes -- and are too big for the one-launch loop used to run the reference's loop literally: one child
// launch + one accumulate launch per non-zero block (jh_general.hip: loop_fwd / loop_adj).  Here ONE launch runs every dense child
// (blockIdx.y = block column j, blockIdx.z = block row i; a block that is not an un-adjointed dense matrix returns at once): child
// (i, j) is row_len[i] x col_len[j], reads m_j (forward) / d_i (adjoint), and leaves its product, rounded to the element type like
// the reference's dtmp / mtmp, in a slab (forward: slab j at row i's elements of the range; adjoint: slab i at column j's elements
// of the domain).  ONE launch of the general kernels then walks every output line in the reference's order, taking a dense block's
// term from its slab (jh_general.hip: dense_mixed_apply): two launches per mul! instead of up to 2 M K.
// y = B x for the children this pass owns (B column-major, its own leading dimension).  `transposed` = the pass belongs to the
// operator's ADJOINT: then it takes the ADJOINTED children -- block (i, j) = B', whose adjoint is B: B is col_len[j] x row_len[i],
// reads d_i, writes slab i at column j's elements -- while in the forward it takes the un-adjointed ones (B = row_len[i] x col_len[j],
// reads m_j, writes slab j at row i's elements).  Columns in order, product rounded then added: the sequential loop's bits.
template <typename S, int E, int NS>
__global__ __launch_bounds__(256) void k_gemv_rows_mixed(const jh_dev_block *__restrict__ blocks, int64_t nrow, const S *__restrict__ in,
                                                         S *__restrict__ slabs, int64_t slab_stride, const int64_t *__restrict__ row_off,
                                                         const int64_t *__restrict__ col_off, int transposed)
{
    typedef typename vec_of<S, NS>::type V;
    const int64_t i = blockIdx.z, j = blockIdx.y;
    const jh_dev_block b = blocks[i + j * nrow];
    if (b.kind != JH_OP_DENSE || (b.adjoint != 0) != (transposed != 0)) return;
    const int64_t rl = row_off[i + 1] - row_off[i], cl = col_off[j + 1] - col_off[j];
    const int64_t ns = (transposed ? cl : rl) * E, nc = transposed ? rl : cl;     // B's rows (as scalars) and columns
    const int64_t s = ((int64_t)blockIdx.x * 256 + threadIdx.x) * NS;
    if (s >= ns) return;
    const S *x = in + (transposed ? row_off[i] : col_off[j]) * E;
    S *out = slabs + jh_dev_block_prod_off(b, transposed != 0) * E;
    V acc = (V)(S)0;
    const S *col = (const S *)b.coeff + s;
    // sixteen columns' loads in flight (the adds are serial by definition, the loads need not be: a 384-row child has 96 active
    // lanes, latency is all there is to hide)
    int64_t c = 0;
    for (; c + 16 <= nc; c += 16) {
        V a[16];
#pragma unroll
        for (int k = 0; k < 16; k++) a[k] = ldg_nt(reinterpret_cast<const V *>(col + (int64_t)k * ns));
        col += 16 * ns;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if constexpr (E == 1) {
                acc = acc + a[k] * (V)x[c + k];
            } else {
                const S xr = x[2 * (c + k)], xi = x[2 * (c + k) + 1];
                V p;
#pragma unroll
                for (int e = 0; e < NS; e += 2) {
                    p[e] = a[k][e] * xr - a[k][e + 1] * xi;
                    p[e + 1] = a[k][e] * xi + a[k][e + 1] * xr;
                }
                acc = acc + p;
            }
        }
    }
    for (; c < nc; c++, col += ns) {
        V a = ldg_nt(reinterpret_cast<const V *>(col));
        if constexpr (E == 1) {
            acc = acc + a * (V)x[c];
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
    stg(reinterpret_cast<V *>(out + s), acc);
}

// y = B' x for the children this pass owns: one wave per column of B, fp64 wave reduction, rounded and stored.  In the operator's
// adjoint (transposed) these are the UN-adjointed children (B = row_len[i] x col_len[j], reads d_i, slab i at column j's elements); in
// the forward the ADJOINTED ones (block = B', B = col_len[j] x row_len[i], reads m_j, slab j at row i's elements).
template <typename S, int E, int NS>
__global__ __launch_bounds__(256) void k_gemv_cols_mixed(const jh_dev_block *__restrict__ blocks, int64_t nrow, const S *__restrict__ in,
                                                         S *__restrict__ slabs, int64_t slab_stride, const int64_t *__restrict__ row_off,
                                                         const int64_t *__restrict__ col_off, int transposed)
{
    typedef typename vec_of<S, NS>::type V;
    const int64_t i = blockIdx.z, j = blockIdx.y;
    const jh_dev_block b = blocks[i + j * nrow];
    if (b.kind != JH_OP_DENSE || (b.adjoint != 0) == (transposed != 0)) return;
    const int64_t rl = row_off[i + 1] - row_off[i], cl = col_off[j + 1] - col_off[j];
    const int64_t ns = (transposed ? rl : cl) * E, nc = transposed ? cl : rl;     // B's rows (as scalars) and columns
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= nc) return;
    const int lane = threadIdx.x & 63;
    const S *col = (const S *)b.coeff + c * ns;
    const S *x = in + (transposed ? row_off[i] : col_off[j]) * E;
    double sr = 0.0, si = 0.0;
    for (int64_t s = (int64_t)lane * NS; s < ns; s += 64 * NS) {
        V a = ldg_nt(reinterpret_cast<const V *>(col + s));
        V xv = ldg(reinterpret_cast<const V *>(x + s));
        if constexpr (E == 1) {
#pragma unroll
            for (int e = 0; e < NS; e++) {
                if constexpr (NS == 1) sr += (double)a * (double)xv;
                else sr += (double)a[e] * (double)xv[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < NS; e += 2) {
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
        S *o = slabs + (jh_dev_block_prod_off(b, transposed != 0) + c) * E;
        o[0] = (S)sr;
        if (E == 2) o[1] = (S)si;
    }
}

// rows_pass / cols_pass: which of the two kernels have children to process in this direction (jh_general.hip: dense_mixed_apply knows); max_out / max_in:
// the largest output / input length over those children (grid sizing)
template <typename S, int E>
int gemv_mixed_all(const jh_dev_block *blocks, int64_t nrow, int64_t ncol, int64_t rows_max_out, int64_t cols_max_out, void *slabs, int64_t slab_stride,
                   const void *x, int transposed, bool aligned, const int64_t *dev_row_off, const int64_t *dev_col_off)
{
    hipStream_t st = jh_ctx().stream;
    constexpr int NSV = (16 / sizeof(S)) >= E ? (16 / sizeof(S)) : E;
    const bool vec_ok = aligned && ((((uintptr_t)x) | ((uintptr_t)slabs)) & 15u) == 0;
    if (rows_max_out > 0) {
        const int NS = vec_ok ? NSV : E;
        const int64_t row_wgs = (rows_max_out * E / NS + 255) / 256;
        if (vec_ok)
            hipLaunchKernelGGL((k_gemv_rows_mixed<S, E, NSV>), dim3((unsigned)row_wgs, (unsigned)ncol, (unsigned)nrow), dim3(256), 0, st, blocks, nrow,
                               (const S *)x, (S *)slabs, slab_stride, dev_row_off, dev_col_off, transposed);
        else
            hipLaunchKernelGGL((k_gemv_rows_mixed<S, E, E>), dim3((unsigned)row_wgs, (unsigned)ncol, (unsigned)nrow), dim3(256), 0, st, blocks, nrow,
                               (const S *)x, (S *)slabs, slab_stride, dev_row_off, dev_col_off, transposed);
        JH_CHECK_HIP(hipGetLastError());
    }
    if (cols_max_out > 0) {
        const int64_t col_wgs = (cols_max_out + 3) / 4;
        if (vec_ok)
            hipLaunchKernelGGL((k_gemv_cols_mixed<S, E, NSV>), dim3((unsigned)col_wgs, (unsigned)ncol, (unsigned)nrow), dim3(256), 0, st, blocks, nrow,
                               (const S *)x, (S *)slabs, slab_stride, dev_row_off, dev_col_off, transposed);
        else
            hipLaunchKernelGGL((k_gemv_cols_mixed<S, E, E>), dim3((unsigned)col_wgs, (unsigned)ncol, (unsigned)nrow), dim3(256), 0, st, blocks, nrow,
                               (const S *)x, (S *)slabs, slab_stride, dev_row_off, dev_col_off, transposed);
        JH_CHECK_HIP(hipGetLastError());
    }
    return JH_OK;
}


// ---- the dense children of a SPARSE / mixed operator as a LIST (late round 5) ---------------------------------------------------------
// k_gemv_rows_mixed / k_gemv_cols_mixed above launch a grid over EVERY (block row, block column) pair and let the pairs that hold no dense
// child return at once: fine for a 3 x 4 operator, ruinous for a block-diagonal one -- 256 x 256 blocks with 511 dense children launched
// 8.4 M workgroups of which 65 k had work (adjoint 1.86 ms = 0.29 TB/s), 64 children of 1024^2 ran their forward on 64 workgroups.
// jh_blockop_create now lists the dense children of each direction and pass (jh_dense_item: matrix, shape, where its input block and
// its slab piece start), and these two kernels walk the list:
//  * rows (y = B x): 256 lanes = RL row lanes x CG column groups (RL = 2^rl_shift chosen at launch: the rows of the children, shrunk
//    while the launch would leave the chip empty); a lane owns NS rows and walks every CG-th column, sixteen loads in flight; the CG partial
//    rows meet in LDS and are added in group order.  CG = 1 (knob dense_list_split = 0, or children of >= 256 * NS rows that fill the
//    chip anyway): columns in order, product rounded then added -- the sequential loop's bits, k_gemv_rows_mixed's; CG > 1:
//    deterministic, tolerance parity (like the column chunks of k_gemv_rows and like any BLAS).
//  * cols (y = B' x): a group of SUB lanes (a wave, or half / a quarter of one for short columns) owns CPW = 4 columns at once -- the input
//    pack is loaded once for the four, four matrix loads in flight per step --, fp64 partials, xor-butterfly inside the group.
// SHARED (round 6; children whose columns start off the 16-byte grid): neighbouring row chunks of a child share a 128-byte line at BOTH ends of every column run.
// The workgroups are then numbered so that consecutive chunks land on the same XCD (hardware hands consecutive workgroup ids to the eight XCDs in turn), and
// the matrix loads are temporal: the shared line is an L2 hit for the second chunk instead of a second trip to HBM.
template <typename S, int E, int NS, bool SHARED = false>
__global__ __launch_bounds__(256) void k_gemv_rows_list(const jh_dense_item *__restrict__ items, unsigned chunks, int rl_shift, const S *__restrict__ in,
                                                        S *__restrict__ slabs, S *__restrict__ direct_out, int add_found)
{
    typedef typename vec_of<S, NS>::type V;
    __shared__ V sm[256];
    unsigned bid = blockIdx.x;
    if constexpr (SHARED) {
        const unsigned per = gridDim.x / 8u;                                      // logical ids [x * per, x * per + per) run on XCD x; the remainder keeps its ids
        if (bid < per * 8u) bid = (bid & 7u) * per + (bid >> 3);
    }
    const unsigned item = bid / chunks, chunk = bid - item * chunks;
    const jh_dense_item it = items[item];
    const int RL = 1 << rl_shift, CG = 256 >> rl_shift;
    const int rowlane = threadIdx.x & (RL - 1), cg = threadIdx.x >> rl_shift;
    const int64_t ns = it.nr * E, nc = it.nc;
    if ((int64_t)chunk * RL * NS >= ns) return;                               // (the whole workgroup: this child has fewer row chunks than the longest)
    const int64_t s = ((int64_t)chunk * RL + rowlane) * NS;
    const bool live = s < ns;
    // (round 5, last session) the child's rows need not be whole, 16-byte aligned packs: under-aligned loads, the last pack of the rows from ns - NS and
    // stored from its own first scalar on (the host guarantees ns >= NS for every child of the list when NS > E)
    const int64_t sc = (NS > E && live && s + NS > ns) ? ns - NS : s;
    const S *x = in + it.x_off * E;
    V acc = (V)(S)0;
    if (live) {
        const S *col = (const S *)it.A + sc + (int64_t)cg * ns;
        const int64_t step = (int64_t)CG * ns;
        int64_t c = cg;
        for (; c + 15 * (int64_t)CG < nc; c += 16 * (int64_t)CG) {
            V a[16];
#pragma unroll
            for (int k = 0; k < 16; k++) a[k] = SHARED ? ldgu<S, NS>(col + (int64_t)k * step) : ldgu_nt<S, NS>(col + (int64_t)k * step);
            col += 16 * step;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int64_t ck = c + (int64_t)k * CG;
                if constexpr (E == 1) {
                    acc = acc + a[k] * (V)x[ck];
                } else {
                    const S xr = x[2 * ck], xi = x[2 * ck + 1];
                    V p;
#pragma unroll
                    for (int e = 0; e < NS; e += 2) {
                        p[e] = a[k][e] * xr - a[k][e + 1] * xi;
                        p[e + 1] = a[k][e] * xi + a[k][e + 1] * xr;
                    }
                    acc = acc + p;
                }
            }
        }
        for (; c < nc; c += CG, col += step) {
            V a = SHARED ? ldgu<S, NS>(col) : ldgu_nt<S, NS>(col);
            if constexpr (E == 1) {
                acc = acc + a * (V)x[c];
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
    }
    // direct_out (the operator's output vector): every output line has this ONE block -- the product goes where the combine launch would have put it,
    // rounded first like dtmp, then added to d as found (1024) when the operator has more than one block column
    S *out = direct_out ? direct_out + it.line_off * E : slabs + it.out_off * E;
    if (CG == 1) {
        if (live) {
            if (add_found) acc = ldgu<S, NS>(out + sc) + acc;
            stgu_pack<S, NS>(out, s, sc, acc);
        }
        return;
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (cg == 0 && live) {
        V t = sm[rowlane];
        for (int g = 1; g < CG; g++) t = t + sm[g * RL + rowlane];
        if (add_found) t = ldgu<S, NS>(out + sc) + t;
        stgu_pack<S, NS>(out, s, sc, t);
    }
}

template <typename S, int E, int NS, int CPW>
__global__ __launch_bounds__(256) void k_gemv_cols_list(const jh_dense_item *__restrict__ items, unsigned chunks, int sub_shift, const S *__restrict__ in,
                                                        S *__restrict__ slabs, S *__restrict__ direct_out, int add_found)
{
    typedef typename vec_of<S, NS>::type V;
    const unsigned item = blockIdx.x / chunks, chunk = blockIdx.x - item * chunks;
    const jh_dense_item it = items[item];
    const int SUB = 1 << sub_shift, G = 256 >> sub_shift;
    const int g = threadIdx.x >> sub_shift, l = threadIdx.x & (SUB - 1);
    const int64_t ns = it.nr * E, nc = it.nc;
    const int64_t c0 = ((int64_t)chunk * G + g) * CPW;
    if (c0 >= nc) return;
    const S *x = in + it.x_off * E;
    const S *colp[CPW];
#pragma unroll
    for (int k = 0; k < CPW; k++) colp[k] = (const S *)it.A + (c0 + k < nc ? c0 + k : nc - 1) * ns;   // (columns past the end re-read the last one: not stored)
    double sr[CPW], si[CPW];
#pragma unroll
    for (int k = 0; k < CPW; k++) { sr[k] = 0.0; si[k] = 0.0; }
    constexpr int UNR = CPW == 1 ? 4 : (CPW == 2 ? 2 : 1);
    // (columns need not be whole, 16-byte aligned packs: under-aligned loads, the column's last pack from ns - NS; a column shorter than one pack -- a
    // child of a list whose other children are longer -- is summed scalar by scalar)
    if (NS > E && ns < NS) {
        for (int64_t s = (int64_t)l * E; s < ns; s += (int64_t)SUB * E) {
            typedef typename vec_of<S, E>::type VE;
            const VE xv = ldgu<S, E>(x + s);
#pragma unroll
            for (int k = 0; k < CPW; k++) cols_accumulate<S, E, E, VE>(ldgu_nt<S, E>(colp[k] + s), xv, 0, sr[k], si[k]);
        }
    } else {
#pragma unroll UNR
        for (int64_t s = (int64_t)l * NS; s < ns; s += (int64_t)SUB * NS) {
            const int64_t sc = s + NS <= ns ? s : ns - NS;
            const V xv = ldgu<S, NS>(x + sc);
            V a[CPW];
#pragma unroll
            for (int k = 0; k < CPW; k++) a[k] = ldgu_nt<S, NS>(colp[k] + sc);
#pragma unroll
            for (int k = 0; k < CPW; k++) cols_accumulate<S, E, NS, V>(a[k], xv, (int)(s - sc), sr[k], si[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < CPW; k++)
        for (int off = SUB >> 1; off > 0; off >>= 1) {
            sr[k] += __shfl_xor(sr[k], off, 64);
            if (E == 2) si[k] += __shfl_xor(si[k], off, 64);
        }
    if (l == 0) {
        S *o = direct_out ? direct_out + (it.line_off + c0) * E : slabs + (it.out_off + c0) * E;
#pragma unroll
        for (int k = 0; k < CPW; k++)
            if (c0 + k < nc) {
                S vr = (S)sr[k], vi = (S)si[k];                                  // rounded like dtmp / mtmp, then (direct mode of a forward) added to d as found
                if (add_found) { vr = o[(int64_t)k * E] + vr; if (E == 2) vi = o[(int64_t)k * E + 1] + vi; }
                o[(int64_t)k * E] = vr;
                if (E == 2) o[(int64_t)k * E + 1] = vi;
            }
    }
}

static inline int pow2_shift_at_least(int64_t v)                             // smallest k with 2^k >= v
{
    int k = 0;
    while (((int64_t)1 << k) < v) k++;
    return k;
}

// pass 0: y = B x for every item (max_out = the longest output, B's rows); pass 1: y = B' x (max_out = the most columns, max_in = the longest column)
template <typename S, int E>
int gemv_list(const jh_dense_item *items, int64_t nitems, int64_t max_out, int64_t max_in, int pass, void *slabs, const void *x, int aligned, void *direct_out,
              int add_found)
{
    if (nitems == 0 || max_out == 0) return JH_OK;
    jh_context &c = jh_ctx();
    hipStream_t st = c.stream;
    constexpr int NSV = (16 / sizeof(S)) >= E ? (16 / sizeof(S)) : E;
    // aligned: 2 every matrix, dimension and vector on the 16-byte grid; 1 not, but every dimension holds a pack (children of odd dimensions: under-aligned packs); 0 neither
    const bool vec_ok = aligned == 2 && ((((uintptr_t)x) | ((uintptr_t)slabs) | ((uintptr_t)direct_out)) & 15u) == 0;
    const bool scalar_aligned = ((((uintptr_t)x) | ((uintptr_t)slabs) | ((uintptr_t)direct_out)) & (sizeof(S) - 1)) == 0;
    // (the column kernel sums a column shorter than a pack scalar by scalar itself; the row kernel needs every child's rows to hold one)
    const bool vec_cols = vec_ok || (pass == 1 && c.tall_unaligned != 0 && scalar_aligned);
    const bool vec_rows = vec_ok || (pass == 0 && aligned >= 1 && c.tall_unaligned != 0 && scalar_aligned);
    const int NS = (pass == 1 ? vec_cols : vec_rows) ? NSV : E;
    if (pass == 0) {
        const int64_t lanes = (max_out * E + NS - 1) / NS;                   // row lanes of the longest child
        int sh = 8;                                                          // RL = 256: every lane a row lane, columns in order (the sequential loop's bits)
        if (c.dense_list_split != 0) {
            sh = pow2_shift_at_least(lanes);
            if (sh > 8) sh = 8;
            if (sh < 4) sh = 4;
            // few workgroups: narrower row sets, more column groups per workgroup and more workgroups per child
            // (children whose columns do not start on 16-byte boundaries -- odd row counts --: a run of 2^sh packs per column and wave set touches one 128-byte line more
            // than it fills, and with streaming loads that line comes from HBM again for the neighbouring run: 16 row lanes = 256-byte runs fetched x1.49 of their
            // bytes under rocprofv3 (8 x 4095^2: forward 4.0 TB/s against 5.7 for 4096^2; profiles/rocprof_r06_dense_odd_summary.md).  Such lists run the SHARED
            // instantiation -- neighbouring chunks on one XCD, temporal loads: the shared line is an L2 hit -- and keep at least 2^5 row lanes (knob
            // dense_list_rl_min; without SHARED 2^6 was the best: 4.9 / 5.0 TB/s where SHARED reaches 5.7 / 5.9 on 8 x 4095^2 / 64 x 1023^2)
            const int sh_min = (aligned != 2 && vec_rows) ? (int)(c.dense_list_rl_min > 0 ? c.dense_list_rl_min : (c.dense_list_shared ? 5 : 6)) : 4;
            while (sh > sh_min && nitems * ((lanes + ((int64_t)1 << sh) - 1) >> sh) < 4 * (int64_t)c.cu_count) sh--;
        }
        const int64_t chunks = (lanes + ((int64_t)1 << sh) - 1) >> sh;
        JH_REQUIRE(nitems * chunks < ((int64_t)1 << 31), "dense child list: %lld x %lld workgroups exceed the grid", (long long)nitems, (long long)chunks);
        c.last_dense_rl = (int64_t)1 << sh;
        if (vec_rows && aligned != 2 && chunks > 1 && c.dense_list_shared)
            hipLaunchKernelGGL((k_gemv_rows_list<S, E, NSV, true>), dim3((unsigned)(nitems * chunks)), dim3(256), 0, st, items, (unsigned)chunks, sh, (const S *)x, (S *)slabs, (S *)direct_out, add_found);
        else if (vec_rows)
            hipLaunchKernelGGL((k_gemv_rows_list<S, E, NSV>), dim3((unsigned)(nitems * chunks)), dim3(256), 0, st, items, (unsigned)chunks, sh, (const S *)x, (S *)slabs, (S *)direct_out, add_found);
        else
            hipLaunchKernelGGL((k_gemv_rows_list<S, E, E>), dim3((unsigned)(nitems * chunks)), dim3(256), 0, st, items, (unsigned)chunks, sh, (const S *)x, (S *)slabs, (S *)direct_out, add_found);
    } else {
        const int64_t lanes = (max_in * E + NS - 1) / NS;                    // lanes one column keeps busy
        int sh = pow2_shift_at_least(lanes);
        if (sh > 6) sh = 6;
        if (sh < 4) sh = 4;
        // columns per lane group: four share one load of the input pack -- right for short columns (a group is done after a step or two); long columns
        // (>= 8 steps of a full wave) run one column per wave like k_gemv_cols_mixed: more waves, the same loads in flight (knob dense_list_cpw: 0 this rule)
        int cpw = c.dense_list_cpw ? (int)c.dense_list_cpw : (lanes >= 512 ? 1 : (lanes >= 128 ? 2 : 4));   // (adjoint of 16 x 4096^2: 6.66 / 6.69 / 5.97 TB/s with 1 / 2 / 4; 256 x 512^2: 3.69 / 6.44 / 6.17; 256 x 256^2: 2.10 / 3.11 / 3.47)
        const int64_t per_wg = (int64_t)(256 >> sh) * cpw;                   // columns per workgroup
        const int64_t chunks = (max_out + per_wg - 1) / per_wg;
        JH_REQUIRE(nitems * chunks < ((int64_t)1 << 31), "dense child list: %lld x %lld workgroups exceed the grid", (long long)nitems, (long long)chunks);
#define JH_COLS(NSX, CPWX) hipLaunchKernelGGL((k_gemv_cols_list<S, E, NSX, CPWX>), dim3((unsigned)(nitems * chunks)), dim3(256), 0, st, items, (unsigned)chunks, sh, (const S *)x, (S *)slabs, (S *)direct_out, add_found)
        if (vec_cols) { if (cpw == 1) JH_COLS(NSV, 1); else if (cpw == 2) JH_COLS(NSV, 2); else JH_COLS(NSV, 4); }
        else { if (cpw == 1) JH_COLS(E, 1); else if (cpw == 2) JH_COLS(E, 2); else JH_COLS(E, 4); }
#undef JH_COLS
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

// all children of a tall operator of uniform dense blocks in one go (jh_blockop.hip); `aligned`: every matrix pointer on 16 bytes
int jh_launch_gemv_batched(const jh_dev_block *dev_blocks, int64_t nchild, int64_t nr, int64_t nc, int dtype, void *y, const void *x,
                           int adjoint, bool aligned, bool wide, const int64_t *dev_row_off)
{
    if (wide)
        switch (dtype) {
        case JH_F32: return gemv_batched_wide<float, 1>(dev_blocks, nchild, nr, nc, y, x, adjoint, aligned);
        case JH_F64: return gemv_batched_wide<double, 1>(dev_blocks, nchild, nr, nc, y, x, adjoint, aligned);
        case JH_C32: return gemv_batched_wide<float, 2>(dev_blocks, nchild, nr, nc, y, x, adjoint, aligned);
        case JH_C64: return gemv_batched_wide<double, 2>(dev_blocks, nchild, nr, nc, y, x, adjoint, aligned);
        }
    switch (dtype) {
    case JH_F32: return gemv_batched<float, 1>(dev_blocks, nchild, nr, nc, y, x, adjoint, aligned, dev_row_off);
    case JH_F64: return gemv_batched<double, 1>(dev_blocks, nchild, nr, nc, y, x, adjoint, aligned, dev_row_off);
    case JH_C32: return gemv_batched<float, 2>(dev_blocks, nchild, nr, nc, y, x, adjoint, aligned, dev_row_off);
    case JH_C64: return gemv_batched<double, 2>(dev_blocks, nchild, nr, nc, y, x, adjoint, aligned, dev_row_off);
    }
    return jh_fail(JH_ERR_INVALID, "gemv_batched: unknown dtype %d", dtype);
}

// every dense child of a mixed operator: one launch of the sequential (rows) kernel for the children whose product in this direction is
// B x, one of the wave-reduction (cols) kernel for those where it is B' x (see k_gemv_rows_mixed); x: forward the domain vector, adjoint the
// range vector; rows_max_out / cols_max_out = 0: that pass has no children
int jh_launch_gemv_mixed_all(const jh_dev_block *blocks, int64_t nrow, int64_t ncol, int64_t rows_max_out, int64_t cols_max_out, int dtype, void *slabs,
                             int64_t slab_stride, const void *x, int transposed, bool aligned, const int64_t *dev_row_off, const int64_t *dev_col_off)
{
    switch (dtype) {
    case JH_F32: return gemv_mixed_all<float, 1>(blocks, nrow, ncol, rows_max_out, cols_max_out, slabs, slab_stride, x, transposed, aligned, dev_row_off, dev_col_off);
    case JH_F64: return gemv_mixed_all<double, 1>(blocks, nrow, ncol, rows_max_out, cols_max_out, slabs, slab_stride, x, transposed, aligned, dev_row_off, dev_col_off);
    case JH_C32: return gemv_mixed_all<float, 2>(blocks, nrow, ncol, rows_max_out, cols_max_out, slabs, slab_stride, x, transposed, aligned, dev_row_off, dev_col_off);
    case JH_C64: return gemv_mixed_all<double, 2>(blocks, nrow, ncol, rows_max_out, cols_max_out, slabs, slab_stride, x, transposed, aligned, dev_row_off, dev_col_off);
    }
    return jh_fail(JH_ERR_INVALID, "gemv_mixed_all: unknown dtype %d", dtype);
}

// the dense children of one direction and pass of a sparse / mixed operator from their list (jh_blockop_create builds it)
int jh_launch_gemv_list(const jh_dense_item *items, int64_t nitems, int64_t max_out, int64_t max_in, int pass, int dtype, void *slabs, const void *x, int aligned,
                        void *direct_out, int add_found)
{
    switch (dtype) {
    case JH_F32: return gemv_list<float, 1>(items, nitems, max_out, max_in, pass, slabs, x, aligned, direct_out, add_found);
    case JH_F64: return gemv_list<double, 1>(items, nitems, max_out, max_in, pass, slabs, x, aligned, direct_out, add_found);
    case JH_C32: return gemv_list<float, 2>(items, nitems, max_out, max_in, pass, slabs, x, aligned, direct_out, add_found);
    case JH_C64: return gemv_list<double, 2>(items, nitems, max_out, max_in, pass, slabs, x, aligned, direct_out, add_found);
    }
    return jh_fail(JH_ERR_INVALID, "gemv_list: unknown dtype %d", dtype);
}

extern "C" int jh_gemv(const void *A_device, int64_t nr, int64_t nc, int dtype, jh_bvec *y, const jh_bvec *x, int adjoint)
{
    JH_TRY(jh_enter(y, x));
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
