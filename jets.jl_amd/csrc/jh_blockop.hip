// jh_blockop.hip -- JetBlock_df! / JetBlock_df'! (src/Jets.jl:1010-1057) and the fused A'oA
// (src/Jets.jl:530-534 over (A', A)) as hand-written gfx950 kernels.
//
// Data layout in HBM: the range vector d is one slab, block i at element offset row_off[i];
// a DIAG block's coefficients are a device array of the block's length; the domain vector m of a
// tall (one-column) operator is a plain array (src/Jets.jl:927).
//
// Two kernel families:
//  * tall fast path (ncol == 1, every block DIAG, equal block length, 16-byte aligned): the
//    BASELINE.json workload.  Forward: a workgroup owns an element tile, keeps its m tile in
//    registers and streams `fwd_group` blocks through it (a read once, d written once, m
//    re-read nrow/fwd_group times, from L2/MALL).  Adjoint: a thread owns 16-byte element
//    vectors and walks the rows IN ORDER, product rounded then added -- the reference's
//    `_m .+= mul!(mtmp, op', _d)` (1049) without the mtmp round trip -- so the result is
//    bit-identical to the sequential CPU loop.  HBM-bound: 16 B/lane loads, `adj_depth` rows in
//    flight per thread, nontemporal on the streamed operands.
//  * general path: any nrow x ncol mix of ZERO / IDENTITY / SCALE / DIAG blocks with ragged block
//    lengths; one thread per element walks a block row (forward) or block column (adjoint) in
//    the reference's loop order with the same rounding sequence.
#include "jh_internal.h"
#include <type_traits>
#include <tuple>
#include <mutex>
#include <map>

namespace {

template <typename S, int NS> struct vec_of { typedef S type __attribute__((ext_vector_type(NS))); };
template <typename S> struct vec_of<S, 1> { typedef S type; };

// every operand of these kernels lives in HBM: load/store through address_space(1) pointers so the
// compiler emits global_load/global_store (never flat_*), also for pointers read from the block table
template <bool NT, typename V> __device__ inline V ld(const V *p)
{
    typedef const V __attribute__((address_space(1))) *gp;
    if (NT) return __builtin_nontemporal_load((gp)p);
    return *(gp)p;
}
template <bool NT, typename V> __device__ inline void st(V *p, V v)
{
    typedef V __attribute__((address_space(1))) *gp;
    if (NT) __builtin_nontemporal_store(v, (gp)p);
    else *(gp)p = v;
}

// a (conj?) * b on a vector of NS scalars holding NS/E elements; every product/sum rounded.
template <typename S, int E, int NS, typename V> __device__ inline V vmul(V a, V b, bool conj_a)
{
    if constexpr (E == 1) {
        return a * b;
    } else {
        V o;
#pragma unroll
        for (int e = 0; e < NS; e += 2) {
            S ar = a[e], ai = conj_a ? -a[e + 1] : a[e + 1], br = b[e], bi = b[e + 1];
            o[e] = ar * br - ai * bi;
            o[e + 1] = ar * bi + ai * br;
        }
        return o;
    }
}

// does this block read a coefficient pack (DIAG: its diagonal; SQUARE as a Jacobian: its linearisation point)?
__device__ inline bool block_reads_coeff(const jh_dev_block &b, bool fmode)
{
    return b.kind == JH_OP_DIAG || (b.kind == JH_OP_SQUARE && !(fmode && !b.adjoint));
}

// child mul! of an elementwise block on a 16-byte pack, coefficient pack already loaded (the kernels below issue the loads of
// GENERAL_Q blocks before combining them)
template <typename S, int E, int NS, typename V>
__device__ inline V apply_block_loaded(const jh_dev_block &b, V x, V c, bool transposed, bool fmode)
{
    const bool cj = (b.adjoint != 0) != transposed;
    switch (b.kind) {
    case JH_OP_IDENTITY: return x;
    case JH_OP_SQUARE:
        if (fmode && !b.adjoint) return vmul<S, E, NS, V>(x, x, false);
        return vmul<S, E, NS, V>(c + c, x, cj);
    case JH_OP_SCALE: {
        if (E == 1 || b.real_scale) {                   // a REAL scalar (jh_dev_block_of) multiplies part by part (Julia's a::Real * z)
            return (V)(S)b.sre * x;
        } else {                                        // a Complex one: the full product, also when its imaginary part is zero
            V a;
#pragma unroll
            for (int e = 0; e < NS; e += 2) { a[e] = (S)b.sre; a[e + 1] = (S)b.sim; }
            return vmul<S, E, NS, V>(a, x, cj);
        }
    }
    case JH_OP_DIAG: return vmul<S, E, NS, V>(c, x, cj);
    default: return (V)(S)0;
    }
}

// ------------------------------------------------------------------ tall fast path ------------
// 1-D grid of ntiles * ngroups workgroups, walked in BANDS of `band` row groups: inside a band the row
// group is the fastest index (workgroups sharing an m tile are dispatched together, so the tile is
// re-read from L2/MALL, not HBM), bands follow one another.  band = 1 is the fully sequential sweep
// (one block row at a time); band = ngroups touches every row concurrently.  n_scalars % NS == 0.
// MIXED: the rows are not all plain diagonals -- a row may be IDENTITY, SCALE, a (conjugated) DIAG, the Jacobian of a SQUARE
// child, or a ZERO block, which the linear loop SKIPS (src/Jets.jl:1022): its d_i stays as found.  The kind is read from the
// row table (uniform per workgroup: scalar branches), the arithmetic is the general kernels' apply_block_loaded.
template <typename S, int E, int NS, int U, bool NT, int BLK, bool MIXED = false>
__global__ __launch_bounds__(BLK) void k_tall_diag_fwd(const jh_dev_block *__restrict__ blocks, int64_t nrow, int rows_per_wg,
                                                       const S *__restrict__ a_base, int64_t a_stride,
                                                       const S *__restrict__ m, S *__restrict__ d, int64_t n_scalars,
                                                       unsigned ntiles, unsigned ngroups, unsigned band, unsigned ctiles)
{
    typedef typename vec_of<S, NS>::type V;
    unsigned tile, grp;
    if (ctiles) {
        // COLUMN bands (late round 4): `ctiles` consecutive tiles of one row group, then the same tiles of the next group, ... then the next
        // band of tiles.  Inside a group's share of a band the workgroups stream linearly like a copy (32-64 tiles = 128-256 KiB), the band
        // of m is reused by every row from L2: at 128-512 rows +3 ... +10 % over the row-concurrent walk, whose consecutive workgroups are a
        // whole block apart (tools/micro/fwd_small_rows.hip, profiles/exp_r04_fwd_small_rows.txt); at 1024 rows the row-concurrent walk wins
        const unsigned per_c = ctiles * ngroups;              // workgroups in a full column band
        const unsigned cb = blockIdx.x / per_c;
        const unsigned r = blockIdx.x - cb * per_c;
        const unsigned cw = (cb * ctiles + ctiles <= ntiles) ? ctiles : ntiles - cb * ctiles;   // last band may be narrower
        grp = r / cw;
        tile = cb * ctiles + r % cw;
    } else {
        const unsigned per_band = ntiles * band;              // workgroups in a full band
        const unsigned b = blockIdx.x / per_band;
        const unsigned r = blockIdx.x - b * per_band;
        const unsigned width = (b * band + band <= ngroups) ? band : ngroups - b * band;   // last band may be narrower
        tile = r / width;
        grp = b * band + r % width;
    }
    const int64_t s0 = ((int64_t)tile * U * BLK + threadIdx.x) * NS;
    const int64_t i0 = (int64_t)grp * rows_per_wg;
    const int64_t i1 = (i0 + rows_per_wg < nrow) ? i0 + rows_per_wg : nrow;
    const bool full = ((int64_t)(tile + 1) * U * BLK * NS) <= n_scalars;
    V mv[U];
    if constexpr (MIXED) {
        bool ok[U];
        int64_t sk[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            ok[k] = (s0 + (int64_t)k * BLK * NS) < n_scalars;
            sk[k] = ok[k] ? s0 + (int64_t)k * BLK * NS : 0;
            mv[k] = ld<false>(reinterpret_cast<const V *>(m + sk[k]));
        }
        jh_dev_block nxt;                                                                  // the row table one row ahead (scalar loads)
        if (i0 < i1) nxt = blocks[i0];
        for (int64_t i = i0; i < i1; i++) {
            const jh_dev_block blk = nxt;
            if (i + 1 < i1) nxt = blocks[i + 1];
            if (blk.kind == JH_OP_ZERO) continue;                                          // (1022)
            const bool rc = block_reads_coeff(blk, false);
            S *di = d + i * n_scalars;
#pragma unroll
            for (int k = 0; k < U; k++) {
                const V c = rc ? ld<NT>(reinterpret_cast<const V *>((const S *)blk.coeff + sk[k])) : (V)(S)0;
                if (ok[k]) st<NT>(reinterpret_cast<V *>(di + sk[k]), apply_block_loaded<S, E, NS, V>(blk, mv[k], c, false, false));   // (1026)
            }
        }
        return;
    }
    if (full) {
#pragma unroll
        for (int k = 0; k < U; k++) mv[k] = ld<false>(reinterpret_cast<const V *>(m + s0 + (int64_t)k * BLK * NS));
#pragma unroll 2
        for (int64_t i = i0; i < i1; i++) {
            const S *a = a_base ? a_base + i * a_stride : (const S *)blocks[i].coeff;
            S *di = d + i * n_scalars;
            V av[U];
#pragma unroll
            for (int k = 0; k < U; k++) av[k] = ld<NT>(reinterpret_cast<const V *>(a + s0 + (int64_t)k * BLK * NS));
#pragma unroll
            for (int k = 0; k < U; k++)
                st<NT>(reinterpret_cast<V *>(di + s0 + (int64_t)k * BLK * NS), vmul<S, E, NS, V>(av[k], mv[k], false));
        }
    } else {
        // the last tile of a row: a pack past the end re-reads pack 0 and stores nothing (every mv[k] is defined on every lane:
        // conditionally loaded ones made the compiler keep the tile in scratch, 400 bytes per lane at 8 packs x 1024 threads --
        // tools/kernel_resources.py; tests/test_kernel_resources.py keeps every kernel of the library at 0 bytes of scratch)
        bool ok[U];
        int64_t sk[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            ok[k] = (s0 + (int64_t)k * BLK * NS) < n_scalars;
            sk[k] = ok[k] ? s0 + (int64_t)k * BLK * NS : 0;
            mv[k] = ld<false>(reinterpret_cast<const V *>(m + sk[k]));
        }
        for (int64_t i = i0; i < i1; i++) {
            const S *a = a_base ? a_base + i * a_stride : (const S *)blocks[i].coeff;
            S *di = d + i * n_scalars;
#pragma unroll
            for (int k = 0; k < U; k++) {
                const V av = ld<NT>(reinterpret_cast<const V *>(a + sk[k]));
                if (ok[k]) st<NT>(reinterpret_cast<V *>(di + sk[k]), vmul<S, E, NS, V>(av, mv[k], false));
            }
        }
    }
}

// one thread: U vectors of the domain, all rows in order.  MODE 0: adjoint (reads a_i, d_i);
// MODE 1: fused normal equations y = sum_i conj(a_i) .* (a_i .* m) (reads a_i only).
template <typename S, int E, int NS, int U, int DEPTH, bool NT, int MODE, int BLK, bool MIXED = false>
__global__ __launch_bounds__(BLK) void k_tall_diag_adj(const jh_dev_block *__restrict__ blocks, int64_t nrow,
                                                       const S *__restrict__ a_base, int64_t a_stride, S *__restrict__ out,
                                                       const S *__restrict__ in, int64_t n_scalars, int direct,
                                                       int64_t s_begin, int64_t s_end, int64_t row0, int64_t row1, int accumulate,
                                                       int64_t rows_per_part, S *__restrict__ part_out, int64_t part_stride)
{
    // rows [row0, row1) of the operator; accumulate != 0 continues the ordered sum from what `out` holds (a long operator
    // can be walked in several launches with the bits of one: ((0 + p_0) + p_1) + ... is the same sequence)
    // the launch covers the scalar range [s_begin, s_end) of the domain vector (the whole vector, or one chunk
    // when the multi-GPU exchange is pipelined chunk by chunk against this kernel)
    // rows_per_part > 0: split-row walk (many rows of small blocks, where one workgroup per element tile would leave the
    // chip idle): workgroup row blockIdx.y sums its own rows in order into slab blockIdx.y of `part_out`; k_fold_parts
    // adds the slabs in part order afterwards (deterministic; not the bits of the single ordered sum)
    typedef typename vec_of<S, NS>::type V;
    if (rows_per_part > 0) {
        row0 += (int64_t)blockIdx.y * rows_per_part;
        if (row0 + rows_per_part < row1) row1 = row0 + rows_per_part;
        out = part_out + (int64_t)blockIdx.y * part_stride - s_begin;
        accumulate = 0;
    }
    const int64_t s0 = s_begin + ((int64_t)blockIdx.x * U * BLK + threadIdx.x) * NS;
    bool ok[U];
    V acc[U], mv[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        ok[k] = (s0 + (int64_t)k * BLK * NS) < s_end;
        acc[k] = (accumulate && ok[k]) ? ld<false>(reinterpret_cast<const V *>(out + s0 + (int64_t)k * BLK * NS)) : (V)(S)0;
        if (MODE == 1) mv[k] = ok[k] ? ld<false>(reinterpret_cast<const V *>(in + s0 + (int64_t)k * BLK * NS)) : (V)(S)0;
    }
    // clamp out-of-range vectors onto a valid address so the main loop is branch-free
    int64_t sk[U];
#pragma unroll
    for (int k = 0; k < U; k++) sk[k] = ok[k] ? s0 + (int64_t)k * BLK * NS : s_begin;

    int64_t i = row0;
    if constexpr (MIXED) {                 // rows of any elementwise kind (see k_tall_diag_fwd); zero blocks are skipped (1047)
        jh_dev_block blk[DEPTH], nxt[DEPTH];                               // the row table one batch ahead (scalar loads)
        if (i + DEPTH <= row1) {
#pragma unroll
            for (int j = 0; j < DEPTH; j++) nxt[j] = blocks[i + j];
        }
        for (; i + DEPTH <= row1; i += DEPTH) {
            V av[DEPTH][U], dv[DEPTH][U];
            const int64_t ahead = (i + 2 * DEPTH <= row1) ? i + DEPTH : i;
#pragma unroll
            for (int j = 0; j < DEPTH; j++) {
                blk[j] = nxt[j];
                nxt[j] = blocks[ahead + j];
            }
#pragma unroll
            for (int j = 0; j < DEPTH; j++) {
                const bool on = blk[j].kind != JH_OP_ZERO, rc = block_reads_coeff(blk[j], false);
#pragma unroll
                for (int k = 0; k < U; k++) {
                    av[j][k] = rc ? ld<NT>(reinterpret_cast<const V *>((const S *)blk[j].coeff + sk[k])) : (V)(S)0;
                    dv[j][k] = (MODE == 0 && on) ? ld<NT>(reinterpret_cast<const V *>(in + (i + j) * n_scalars + sk[k])) : (V)(S)0;
                }
            }
#pragma unroll
            for (int j = 0; j < DEPTH; j++)
                if (blk[j].kind != JH_OP_ZERO) {
#pragma unroll
                    for (int k = 0; k < U; k++) {
                        const V t = (MODE == 0) ? dv[j][k] : apply_block_loaded<S, E, NS, V>(blk[j], mv[k], av[j][k], false, false);
                        acc[k] = acc[k] + apply_block_loaded<S, E, NS, V>(blk[j], t, av[j][k], true, false);   // _m .+= mul!(mtmp, op', _d)
                    }
                }
        }
        for (; i < row1; i++) {
            const jh_dev_block blk = blocks[i];
            if (blk.kind == JH_OP_ZERO) continue;
            const bool rc = block_reads_coeff(blk, false);
#pragma unroll
            for (int k = 0; k < U; k++) {
                const V c = rc ? ld<NT>(reinterpret_cast<const V *>((const S *)blk.coeff + sk[k])) : (V)(S)0;
                const V t = (MODE == 0) ? ld<NT>(reinterpret_cast<const V *>(in + i * n_scalars + sk[k])) : apply_block_loaded<S, E, NS, V>(blk, mv[k], c, false, false);
                acc[k] = acc[k] + apply_block_loaded<S, E, NS, V>(blk, t, c, true, false);
            }
        }
    }
    for (; !MIXED && !direct && i + DEPTH <= row1; i += DEPTH) {
        V av[DEPTH][U], dv[DEPTH][U];
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            const S *a = a_base ? a_base + (i + j) * a_stride : (const S *)blocks[i + j].coeff;
#pragma unroll
            for (int k = 0; k < U; k++) {
                av[j][k] = ld<NT>(reinterpret_cast<const V *>(a + sk[k]));
                if (MODE == 0) dv[j][k] = ld<NT>(reinterpret_cast<const V *>(in + (i + j) * n_scalars + sk[k]));
            }
        }
#pragma unroll
        for (int j = 0; j < DEPTH; j++)
#pragma unroll
            for (int k = 0; k < U; k++) {
                V t = (MODE == 0) ? dv[j][k] : vmul<S, E, NS, V>(av[j][k], mv[k], false);   // d_i = a_i .* m   (1026)
                V p = vmul<S, E, NS, V>(av[j][k], t, true);                                 // mtmp = conj(a_i) .* d_i
                acc[k] = acc[k] + p;                                                         // _m .+= mtmp   (1049)
            }
    }
    for (; i < row1; i++) {
        const S *a = a_base ? a_base + i * a_stride : (const S *)blocks[i].coeff;
#pragma unroll
        for (int k = 0; k < U; k++) {
            V av = ld<NT>(reinterpret_cast<const V *>(a + sk[k]));
            V t = (MODE == 0) ? ld<NT>(reinterpret_cast<const V *>(in + i * n_scalars + sk[k])) : vmul<S, E, NS, V>(av, mv[k], false);
            V p = vmul<S, E, NS, V>(av, t, true);
            acc[k] = direct ? p : acc[k] + p;     // nrow == 1: mul!(_m, op', _d) writes directly (1051)
        }
    }
#pragma unroll
    for (int k = 0; k < U; k++)
        if (ok[k]) st<false>(reinterpret_cast<V *>(out + s0 + (int64_t)k * BLK * NS), acc[k]);
}

// ------------------------------------------------------------------ fused solver updates ------
// y = alpha * (A x) + beta * y with ||y||^2 in the same pass: the two halves of an LSQR/CGLS iteration
// (u <- A v - alpha u ; v <- A'u - beta v, each followed by a norm) without a temporary range vector,
// a separate axpby pass or a separate norm pass.  Rounding sequence == the unfused chain
// (mul! into a temporary, then `y .= alpha*tmp .+ beta*y`): product, scale, scale, add, each rounded.
template <int BLK> __device__ inline void wg_sum_store(double v, double *slot)
{
    __shared__ double sm[BLK / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sm[0];
#pragma unroll
        for (int w = 1; w < BLK / 64; w++) r += sm[w];
        *slot = r;
    }
}

// ---- the normal-equations pass of the device-resident CG loops (jh_lsqr.hip: cg_graph_impl; round 4) ---------------------------------
// ONE launch per iteration where the host-driven loop makes four (p <- s + bk p ; y = A'A p ; y += damp^2 p ; <p, y>): a thread owns one
// 16-byte pack of the domain -- it updates its pack of p (nobody else reads it in this launch: the rows below read coefficients only),
// walks all rows in order with DEPTH rows in flight exactly as k_tall_diag_adj MODE 1 does (product, product, add, each rounded: the bits
// of jh_blockop_normal_mul), adds the damping term with the lincomb's rounding, stores y and leaves its share of <p, y> (fp64) to the
// workgroup's partial.  Coefficients (bk, damp^2, the flags) come from device memory, so the launch is the same every iteration.
template <typename S, int E, int NS, int DEPTH, int BLK = 256, int U = 1, bool NT = true>
__global__ __launch_bounds__(BLK) void k_cg_normal(const jh_dev_block *__restrict__ blocks, int64_t nrow, const S *__restrict__ a_base, int64_t a_stride,
                                                   S *__restrict__ p, const S *__restrict__ sres, S *__restrict__ y, int64_t n_scalars,
                                                   const jh_cg_dev *__restrict__ stt, double *__restrict__ partials)
{
    typedef typename vec_of<S, NS>::type V;
    // a thread owns U packs, BLK packs apart (the shapes of the fused normal operator, launch_tall_adj_mixed: fat workgroups once the
    // blocks are big -- 64 x 128^3 with 256 x 1 x 8: 120 us per pass, with 512 x 2 x 2: 80)
    int64_t sk[U];
    bool ok[U];
    V pv[U], sv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int64_t s0 = (((int64_t)blockIdx.x * U + u) * BLK + threadIdx.x) * NS;
        ok[u] = s0 < n_scalars;
        sk[u] = ok[u] ? s0 : 0;
    }
    // the state and this lane's packs of p and s are requested together, before the first decision (a launch of this size is paced by
    // round trips, not by bytes).  Folding the previous vector update's ||s||^2 partials and applying the second scalar update HERE, in
    // every workgroup (one more graph node less), was tried and lost: the whole grid then waits for a fold, a barrier and an fp64 chain
    // before its first coefficient load -- 28.7 us per iteration against 20.2 at 64 x 64^3 (profiles/bench_cgnr_sizes_r04.txt).
#pragma unroll
    for (int u = 0; u < U; u++) {
        pv[u] = ld<false>(reinterpret_cast<const V *>(p + sk[u]));
        sv[u] = ld<false>(reinterpret_cast<const V *>(sres + sk[u]));
    }
    const int done = stt->done, skip_p = stt->skip_p;
    const double bk = stt->bk, damp2 = stt->damp2;
#pragma unroll
    for (int u = 0; u < U; u++) asm volatile("" : "+v"(pv[u]), "+v"(sv[u]));
    if (done) return;
    if (!skip_p) {                                                           // p = 1*s + bk*p  (jh_lincomb's sequence: bk*p rounded, then the sum)
#pragma unroll
        for (int u = 0; u < U; u++) {
            const V bp = (V)(S)bk * pv[u];
            pv[u] = sv[u] + bp;
            if (ok[u]) st<false>(reinterpret_cast<V *>(p + sk[u]), pv[u]);
        }
    }
    V acc[U];
#pragma unroll
    for (int u = 0; u < U; u++) acc[u] = (V)(S)0;
    int64_t i = 0;
    for (; i + DEPTH <= nrow; i += DEPTH) {
        V av[DEPTH][U];
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            const S *a = a_base ? a_base + (i + j) * a_stride : (const S *)blocks[i + j].coeff;
#pragma unroll
            for (int u = 0; u < U; u++) av[j][u] = ld<NT>(reinterpret_cast<const V *>(a + sk[u]));
        }
#pragma unroll
        for (int j = 0; j < DEPTH; j++)
#pragma unroll
            for (int u = 0; u < U; u++) {
                const V t = vmul<S, E, NS, V>(av[j][u], pv[u], false);        // d_i = a_i .* p
                acc[u] = acc[u] + vmul<S, E, NS, V>(av[j][u], t, true);       // y .+= conj(a_i) .* d_i, rows in order
            }
    }
    for (; i < nrow; i++) {
        const S *a = a_base ? a_base + i * a_stride : (const S *)blocks[i].coeff;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const V av = ld<NT>(reinterpret_cast<const V *>(a + sk[u]));
            const V t = vmul<S, E, NS, V>(av, pv[u], false);
            acc[u] = acc[u] + vmul<S, E, NS, V>(av, t, true);
        }
    }
    double part = 0.0;
#pragma unroll
    for (int u = 0; u < U; u++) {
        if (damp2 != 0.0) {                                                  // y = 1*y + damp^2*p
            const V dp = (V)(S)damp2 * pv[u];
            acc[u] = acc[u] + dp;
        }
        if (ok[u]) {
            st<false>(reinterpret_cast<V *>(y + sk[u]), acc[u]);
#pragma unroll
            for (int e = 0; e < NS; e++) part += (double)pv[u][e] * (double)acc[u][e];   // Re <p, y>: over the scalars (a complex vector is 2n reals here)
        }
    }
    wg_sum_store<BLK>(part, partials + blockIdx.x);
}


template <typename S, int NS, typename V> __device__ inline double vnorm2(V r)
{
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < NS; e++) acc += (double)r[e] * (double)r[e];
    return acc;
}

// forward: d_i = alpha * (a_i .* m) + beta * d_i ; sequential row sweep (tile index fastest)
// WIDE (S = float, beta == 0): the scalar is Julia's Float64 (JH_SCALAR_WIDE) -- d_i = Float32(wscal * Float64(a_i .* m)), the promoted
// product of `d .= a * tmp` (src/Jets.jl:1159) rounded once on the store
template <typename S, int E, int NS, int U, int BLK, bool MIXED = false, bool WIDE = false>
__global__ __launch_bounds__(BLK) void k_tall_diag_fwd_update(const jh_dev_block *__restrict__ blocks, int64_t nrow, int rows_per_wg,
                                                              const S *__restrict__ a_base, int64_t a_stride,
                                                              const S *__restrict__ m, S *__restrict__ d, int64_t n_scalars,
                                                              unsigned ntiles, unsigned ngroups, int walk, S alpha, S beta,
                                                              double *__restrict__ partials, double wscal = 0.0)
{
    typedef typename vec_of<S, NS>::type V;
    // walk 0: tile index fastest (one block row at a time); walk 1: row group fastest (all rows concurrently); walk >= 2: COLUMN bands of
    // `walk` tiles (that many consecutive tiles of one row group, then the same tiles of the next group, ... then the next band: k_tall_diag_fwd)
    unsigned tile, grp;
    if (walk >= 2) {
        const unsigned ct = (unsigned)walk, per_c = ct * ngroups, cb = blockIdx.x / per_c, r = blockIdx.x - cb * per_c;
        const unsigned cw = (cb * ct + ct <= ntiles) ? ct : ntiles - cb * ct;
        grp = r / cw;
        tile = cb * ct + r % cw;
    } else {
        tile = walk ? blockIdx.x / ngroups : blockIdx.x % ntiles;
        grp = walk ? blockIdx.x % ngroups : blockIdx.x / ntiles;
    }
    const int64_t s0 = ((int64_t)tile * U * BLK + threadIdx.x) * NS;
    const int64_t i0 = (int64_t)grp * rows_per_wg;
    const int64_t i1 = (i0 + rows_per_wg < nrow) ? i0 + rows_per_wg : nrow;
    bool ok[U];
    int64_t sk[U];
    V mv[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        ok[k] = (s0 + (int64_t)k * BLK * NS) < n_scalars;
        sk[k] = ok[k] ? s0 + (int64_t)k * BLK * NS : 0;
        mv[k] = ld<false>(reinterpret_cast<const V *>(m + sk[k]));
    }
    const bool use_old = (beta != (S)0);
    double nrm = 0.0;
    jh_dev_block nxt;                                                       // MIXED: the row table one row ahead
    if (MIXED && i0 < i1) nxt = blocks[i0];
    for (int64_t i = i0; i < i1; i++) {
        jh_dev_block blk;
        bool rc = true;
        if constexpr (MIXED) {                                              // any elementwise row kind (see k_tall_diag_fwd)
            blk = nxt;
            if (i + 1 < i1) nxt = blocks[i + 1];
            rc = block_reads_coeff(blk, false);
        }
        const S *a = MIXED ? (const S *)blk.coeff : (!a_base ? (const S *)blocks[i].coeff : a_base + i * a_stride);
        S *di = d + i * n_scalars;
        V av[U], dv[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            av[k] = rc ? ld<true>(reinterpret_cast<const V *>(a + sk[k])) : (V)(S)0;
            dv[k] = use_old ? ld<true>(reinterpret_cast<const V *>(di + sk[k])) : (V)(S)0;   // beta == 0: d is write-only
        }
#pragma unroll
        for (int k = 0; k < U; k++) {
            V t;
            if constexpr (MIXED) t = (blk.kind != JH_OP_ZERO) ? apply_block_loaded<S, E, NS, V>(blk, mv[k], av[k], false, false) : (V)(S)0;   // a zero row of the zeros() temporary
            else t = vmul<S, E, NS, V>(av[k], mv[k], false);      // mul!(tmp, A_i, m)
            V s1;
            if constexpr (WIDE) {
#pragma unroll
                for (int e = 0; e < NS; e++) s1[e] = (S)(wscal * (double)t[e]);
            } else {
                s1 = (V)alpha * t;
            }
            V r = s1;
            if (use_old) { V s2 = (V)beta * dv[k]; r = s1 + s2; }   // d_i .= alpha*tmp .+ beta*d_i
            if (ok[k]) {
                st<true>(reinterpret_cast<V *>(di + sk[k]), r);
                nrm += vnorm2<S, NS, V>(r);
            }
        }
    }
    wg_sum_store<BLK>(nrm, partials + blockIdx.x);
}

// adjoint: out = alpha * (sum_i conj(a_i) .* (gamma * d_i), rows in order) + beta * out
// WIDE (S = float): gamma is Julia's Float64 -- every d_i is scaled as Float32(wscal * Float64(d_i)), the `m .= conj(a) * d` stage (1160)
template <typename S, int E, int NS, int U, int DEPTH, int BLK, bool WIDE = false>
__global__ __launch_bounds__(BLK) void k_tall_diag_adj_update(const jh_dev_block *__restrict__ blocks, int64_t nrow,
                                                              const S *__restrict__ a_base, int64_t a_stride, S *__restrict__ out,
                                                              const S *__restrict__ in, int64_t n_scalars, int direct, S alpha, S beta,
                                                              S gamma, double *__restrict__ partials, double wscal = 0.0)
{
    typedef typename vec_of<S, NS>::type V;
    auto scaled = [&](V x) -> V {                          // gamma * d_i, rounded to the element type
        if constexpr (WIDE) {
            V r;
#pragma unroll
            for (int e = 0; e < NS; e++) r[e] = (S)(wscal * (double)x[e]);
            return r;
        } else {
            return (V)gamma * x;                           // gamma = 1: exact
        }
    };
    const int64_t s0 = ((int64_t)blockIdx.x * U * BLK + threadIdx.x) * NS;
    bool ok[U];
    int64_t sk[U];
    V acc[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        ok[k] = (s0 + (int64_t)k * BLK * NS) < n_scalars;
        sk[k] = ok[k] ? s0 + (int64_t)k * BLK * NS : 0;
        acc[k] = (V)(S)0;
    }
    int64_t i = 0;
    for (; !direct && i + DEPTH <= nrow; i += DEPTH) {
        V av[DEPTH][U], dv[DEPTH][U];
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            const S *a = a_base ? a_base + (i + j) * a_stride : (const S *)blocks[i + j].coeff;
#pragma unroll
            for (int k = 0; k < U; k++) {
                av[j][k] = ld<true>(reinterpret_cast<const V *>(a + sk[k]));
                dv[j][k] = ld<true>(reinterpret_cast<const V *>(in + (i + j) * n_scalars + sk[k]));
            }
        }
#pragma unroll
        for (int j = 0; j < DEPTH; j++)
#pragma unroll
            for (int k = 0; k < U; k++) acc[k] = acc[k] + vmul<S, E, NS, V>(av[j][k], scaled(dv[j][k]), true);
    }
    for (; i < nrow; i++) {
        const S *a = a_base ? a_base + i * a_stride : (const S *)blocks[i].coeff;
#pragma unroll
        for (int k = 0; k < U; k++) {
            V p = vmul<S, E, NS, V>(ld<true>(reinterpret_cast<const V *>(a + sk[k])),
                                    scaled(ld<true>(reinterpret_cast<const V *>(in + i * n_scalars + sk[k]))), true);
            acc[k] = direct ? p : acc[k] + p;
        }
    }
    double nrm = 0.0;
#pragma unroll
    for (int k = 0; k < U; k++) {
        V s1 = (V)alpha * acc[k];
        V r = s1;
        if (beta != (S)0) { V s2 = (V)beta * ld<false>(reinterpret_cast<const V *>(out + sk[k])); r = s1 + s2; }
        if (ok[k]) {
            st<false>(reinterpret_cast<V *>(out + sk[k]), r);
            nrm += vnorm2<S, NS, V>(r);
        }
    }
    wg_sum_store<BLK>(nrm, partials + blockIdx.x);
}

// One Golub-Kahan (LSQR) step in ONE pass over the operator and the range vector:
//   u_i <- alpha * (a_i .* v) + beta * u_i        (the forward half: jh_blockop_mul_axpby)
//   w   <- sum_i conj(a_i) .* u_i  (new u, rows in order, product rounded then added: jh_blockop_mul_adj)
//   partial ||u||^2
// A thread owns U 16-byte vectors of the DOMAIN (v and the accumulator stay in registers) and walks all rows with DEPTH rows
// in flight; every coefficient and every element of u is read once, u is written once: (3*N*n + 2*n)*s bytes where the two
// separate halves move (5*N*n + 3*n)*s.  u and w come out bit-identical to the two-kernel sequence.
template <typename S, int E, int NS, int U, int DEPTH, int BLK, bool MIXED = false, bool NT = true>
__global__ __launch_bounds__(BLK) void k_tall_diag_bidiag(const jh_dev_block *__restrict__ blocks, int64_t nrow,
                                                          const S *__restrict__ a_base, int64_t a_stride, S *__restrict__ u,
                                                          const S *__restrict__ v, S *__restrict__ w, int64_t n_scalars, int direct,
                                                          S alpha, S beta, double *__restrict__ partials, int64_t s_begin, int64_t s_end,
                                                          int64_t row0, int64_t row1, int accumulate, int64_t rows_per_part,
                                                          S *__restrict__ part_out, int64_t part_stride, int remap,
                                                          const double *__restrict__ coef_dev, const int *__restrict__ done_dev)
{
    // coef_dev / done_dev (the graph-captured LSQR loop of small operators, jh_lsqr.hip): (alpha, beta) come from device memory --
    // the previous iteration's scalar kernel wrote them -- and a finished solve turns the launch into a no-op
    if (done_dev && *done_dev) return;
    if (coef_dev) { alpha = (S)coef_dev[0]; beta = (S)coef_dev[1]; }
    // rows [row0, row1); accumulate != 0 continues w's ordered sum from what it holds (several launches, the bits of one)
    // remap != 0 (gridDim.x % 8 == 0): workgroups are dealt round-robin over the 8 XCDs, so id % 8 names the XCD; XCD x then owns
    // one CONTIGUOUS eighth of the tiles instead of every eighth tile.  +4 % on this kernel at 128-256 rows of 64 MiB blocks when
    // the rows sit at power-of-two strides, neutral or worse on other layouts (profiles/exp_r02_step_structure.txt), so it is
    // chosen per operator by timing the first real calls (launch_bidiag).  Same values either way: only WHO computes a tile changes.
    // the launch covers the scalar range [s_begin, s_end) of the domain (the whole vector, or one chunk when a multi-GPU
    // host pipelines the exchange of w chunk by chunk against this kernel)
    // rows_per_part > 0: split-row walk, as in k_tall_diag_adj (u is updated row by row either way: same bits; w's sum is
    // formed per part and folded by k_fold_parts)
    typedef typename vec_of<S, NS>::type V;
    if (rows_per_part > 0) {
        row0 += (int64_t)blockIdx.y * rows_per_part;
        if (row0 + rows_per_part < row1) row1 = row0 + rows_per_part;
        w = part_out + (int64_t)blockIdx.y * part_stride - s_begin;
        accumulate = 0;
    }
    const unsigned tile = remap ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int64_t s0 = s_begin + ((int64_t)tile * U * BLK + threadIdx.x) * NS;
    bool ok[U];
    int64_t sk[U];
    V acc[U], vv[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        ok[k] = (s0 + (int64_t)k * BLK * NS) < s_end;
        sk[k] = ok[k] ? s0 + (int64_t)k * BLK * NS : s_begin;
        acc[k] = (accumulate && ok[k]) ? ld<false>(reinterpret_cast<const V *>(w + sk[k])) : (V)(S)0;
        vv[k] = ld<false>(reinterpret_cast<const V *>(v + sk[k]));
    }
    const bool use_old = (beta != (S)0);
    double nrm = 0.0;
    int64_t i = row0;
    if constexpr (MIXED) {
        // rows of any elementwise kind.  A ZERO row: mul!(tmp, A, v) into the zeros() temporary leaves tmp_i = 0 (1022), so
        // u_i <- alpha*0 + beta*u_i, and the row adds nothing to w (1047)
        // the row table is read one batch AHEAD (scalar loads): a batch's coefficient loads need its descriptors, and waiting for
        // them row by row cost 12 % at 1024 rows of 8 MiB (profiles/exp_r02_mixed_step_shapes.txt)
        // Round 4: the LOAD section of a batch is straight-line code -- a row without a coefficient array (identity, scalar, zero) loads
        // v's pack again (an L1 hit, unused) instead of branching around the load, and "beta == 0: u is write-only" is decided once
        // outside the row loop (two instantiations of the walk) instead of around every load of u.  With a branch per load (what the
        // first version compiled to) the waves drained their outstanding loads at every row, and a tall operator with ONE
        // regularisation row ran its step 13 % below the all-diagonal one (profiles/bench_mixed_rows_r02.txt; now bench_mixed_rows_r04.txt).
        auto walk = [&](auto old_tag) {
            constexpr bool OLD = decltype(old_tag)::value;
            jh_dev_block blk[DEPTH], nxt[DEPTH];
            if (i + DEPTH <= row1) {
#pragma unroll
                for (int j = 0; j < DEPTH; j++) nxt[j] = blocks[i + j];
            }
            for (; i + DEPTH <= row1; i += DEPTH) {
                V av[DEPTH][U], uv[DEPTH][U];
                const int64_t ahead = (i + 2 * DEPTH <= row1) ? i + DEPTH : i;
#pragma unroll
                for (int j = 0; j < DEPTH; j++) {
                    blk[j] = nxt[j];
                    nxt[j] = blocks[ahead + j];
                }
#pragma unroll
                for (int j = 0; j < DEPTH; j++) {
                    const S *ap = block_reads_coeff(blk[j], false) ? (const S *)blk[j].coeff : v;   // no coefficient array: v's pack again (unused)
#pragma unroll
                    for (int k = 0; k < U; k++) {
                        av[j][k] = ld<true>(reinterpret_cast<const V *>(ap + sk[k]));
                        if constexpr (OLD) uv[j][k] = ld<true>(reinterpret_cast<const V *>(u + (i + j) * n_scalars + sk[k]));
                        else uv[j][k] = (V)(S)0;
                    }
                }
#pragma unroll
                for (int j = 0; j < DEPTH; j++) {
                    if (blk[j].kind == JH_OP_DIAG) {                 // the common row: ONE branch per row, then the all-diagonal kernel's straight line
                        const bool cj = blk[j].adjoint != 0;
#pragma unroll
                        for (int k = 0; k < U; k++) {
                            const V t = vmul<S, E, NS, V>(av[j][k], vv[k], cj);
                            V r = (V)alpha * t;
                            if constexpr (OLD) { V s2 = (V)beta * uv[j][k]; r = r + s2; }
                            if (ok[k]) {
                                st<true>(reinterpret_cast<V *>(u + (i + j) * n_scalars + sk[k]), r);
                                nrm += vnorm2<S, NS, V>(r);
                            }
                            acc[k] = acc[k] + vmul<S, E, NS, V>(av[j][k], r, !cj);
                        }
                        continue;
                    }
                    const bool on = blk[j].kind != JH_OP_ZERO;
#pragma unroll
                    for (int k = 0; k < U; k++) {
                        const V t = on ? apply_block_loaded<S, E, NS, V>(blk[j], vv[k], av[j][k], false, false) : (V)(S)0;
                        V r = (V)alpha * t;
                        if constexpr (OLD) { V s2 = (V)beta * uv[j][k]; r = r + s2; }
                        if (ok[k]) {
                            st<true>(reinterpret_cast<V *>(u + (i + j) * n_scalars + sk[k]), r);
                            nrm += vnorm2<S, NS, V>(r);
                        }
                        if (on) acc[k] = acc[k] + apply_block_loaded<S, E, NS, V>(blk[j], r, av[j][k], true, false);
                    }
                }
            }
        };
        if (use_old) walk(std::true_type{});
        else walk(std::false_type{});
        for (; i < row1; i++) {
            const jh_dev_block blk = blocks[i];
            const bool on = blk.kind != JH_OP_ZERO;
            const S *ap = block_reads_coeff(blk, false) ? (const S *)blk.coeff : v;
#pragma unroll
            for (int k = 0; k < U; k++) {
                const V c = ld<true>(reinterpret_cast<const V *>(ap + sk[k]));
                const V t = on ? apply_block_loaded<S, E, NS, V>(blk, vv[k], c, false, false) : (V)(S)0;
                V r = (V)alpha * t;
                if (use_old) { V s2 = (V)beta * ld<true>(reinterpret_cast<const V *>(u + i * n_scalars + sk[k])); r = r + s2; }
                if (ok[k]) {
                    st<true>(reinterpret_cast<V *>(u + i * n_scalars + sk[k]), r);
                    nrm += vnorm2<S, NS, V>(r);
                }
                if (on) acc[k] = acc[k] + apply_block_loaded<S, E, NS, V>(blk, r, c, true, false);
            }
        }
    }
    for (; !MIXED && !direct && i + DEPTH <= row1; i += DEPTH) {
        V av[DEPTH][U], uv[DEPTH][U];
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            const S *a = a_base ? a_base + (i + j) * a_stride : (const S *)blocks[i + j].coeff;
#pragma unroll
            for (int k = 0; k < U; k++) {
                av[j][k] = ld<NT>(reinterpret_cast<const V *>(a + sk[k]));
                uv[j][k] = use_old ? ld<NT>(reinterpret_cast<const V *>(u + (i + j) * n_scalars + sk[k])) : (V)(S)0;
            }
        }
#pragma unroll
        for (int j = 0; j < DEPTH; j++)
#pragma unroll
            for (int k = 0; k < U; k++) {
                V t = vmul<S, E, NS, V>(av[j][k], vv[k], false);       // mul!(tmp, A_i, v)
                V r = (V)alpha * t;
                if (use_old) { V s2 = (V)beta * uv[j][k]; r = r + s2; }  // u_i .= alpha*tmp .+ beta*u_i
                if (ok[k]) {
                    st<NT>(reinterpret_cast<V *>(u + (i + j) * n_scalars + sk[k]), r);
                    nrm += vnorm2<S, NS, V>(r);
                }
                acc[k] = acc[k] + vmul<S, E, NS, V>(av[j][k], r, true);  // _m .+= conj(a_i) .* u_i   (1049)
            }
    }
    for (; i < row1; i++) {
        const S *a = a_base ? a_base + i * a_stride : (const S *)blocks[i].coeff;
#pragma unroll
        for (int k = 0; k < U; k++) {
            V av = ld<NT>(reinterpret_cast<const V *>(a + sk[k]));
            V t = vmul<S, E, NS, V>(av, vv[k], false);
            V r = (V)alpha * t;
            if (use_old) { V s2 = (V)beta * ld<NT>(reinterpret_cast<const V *>(u + i * n_scalars + sk[k])); r = r + s2; }
            if (ok[k]) {
                st<NT>(reinterpret_cast<V *>(u + i * n_scalars + sk[k]), r);
                nrm += vnorm2<S, NS, V>(r);
            }
            V p = vmul<S, E, NS, V>(av, r, true);
            acc[k] = direct ? p : acc[k] + p;                            // nrow == 1 writes directly (1051)
        }
    }
#pragma unroll
    for (int k = 0; k < U; k++)
        if (ok[k]) st<false>(reinterpret_cast<V *>(w + sk[k]), acc[k]);
    wg_sum_store<BLK>(nrm, partials + tile + (size_t)blockIdx.y * gridDim.x);      // by tile: the fold's order does not depend on remap
}

// ---- the one-pass step as CHAINED ROW CHUNKS: one batch of DEPTH rows per workgroup ---------------------------------------
// A workgroup of k_tall_diag_bidiag lives for all rows of its tile.  Kernels that read AND write like that run 5-20 % below
// what the same chip does for workgroups that are born, move one batch and die in dispatch order (profiles/
// exp_r02_step_chain.txt: 5.2-5.3 TB/s at 64-512 rows of 64 MiB, 5.8-6.1 at 1024, against 6.1-6.2 for every row count here).
// So the rows are cut into chunks of DEPTH rows and workgroup (chunk c, tile t) CONTINUES the ordered sum of (c-1, t):
//   w_t = ((((0 + p_0) + p_1) + ... ) + p_{8c-1})  |  + p_{8c} + ... + p_{8c+7}   -- the same additions in the same order, so w
// keeps the bits of the single ordered walk (u is row-wise work anyway).  The partial sum travels through memory in the form
// MI355X_MICROARCH.md validates for inter-workgroup hand-offs: every wave stores its piece write-through (sc1), drains
// (s_waitcnt vmcnt(0)), the workgroup barriers, ONE lane raises flag[t] with an agent-scope store; the consumer polls flag[t] with
// agent-scope loads from ONE lane, barriers, then loads the partial with sc1 loads.  Two alternating partial buffers.
// No deadlock, whatever order the hardware starts workgroups in: logical ids are TICKETS taken at start, and (c, t) only waits for
// (c-1, t), whose ticket is smaller -- it has started and depends only on still smaller tickets.  The poll is bounded all the
// same: on its (never observed) expiry the sticky word *err is set and the caller reports it where ||u||^2 is read back.
__device__ inline void st_sc1_16(void *p, unsigned __attribute__((ext_vector_type(4))) v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
// streaming store of one 16-byte pack, spelled out: in this kernel the compiler dropped the `nt` of __builtin_nontemporal_store on
// the batch's stores (plain write-back stores cost 20 % here)
__device__ inline void st_nt_16(void *p, unsigned __attribute__((ext_vector_type(4))) v)
{
    // s_nop 1: a VMEM store of more than 8 bytes reads its data VGPRs a wait state after issue; the compiler pads that hazard
    // for its own stores, not behind inline assembly
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ inline unsigned __attribute__((ext_vector_type(4))) ld_sc1_16(const void *p)
{
    unsigned __attribute__((ext_vector_type(4))) v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <typename S, int E, int NS, int U, int DEPTH, int BLK, bool MIXED = false>
__global__ __launch_bounds__(BLK) void k_tall_diag_bidiag_chain(const jh_dev_block *__restrict__ blocks, int64_t nrow,
                                                                const S *__restrict__ a_base, int64_t a_stride, S *__restrict__ u,
                                                                const S *__restrict__ v, S *__restrict__ w, int64_t n_scalars, S alpha,
                                                                S beta, double *__restrict__ partials, int64_t s_begin, int64_t s_end,
                                                                unsigned ntiles, unsigned nchunks, unsigned *__restrict__ sync,
                                                                S *__restrict__ wpart, unsigned *__restrict__ err, unsigned ctiles)
{
    typedef typename vec_of<S, NS>::type V;
    typedef unsigned U4 __attribute__((ext_vector_type(4)));
    static_assert(sizeof(V) == 16, "one 16-byte pack per lane");
    __shared__ unsigned s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(&sync[0], 1u);              // logical id = order of arrival
    __syncthreads();
    const unsigned ticket = s_ticket;
    // ctiles == 0: tiles fastest over the whole row -- every tile of chunk 0, then every tile of chunk 1, ...
    // ctiles  > 0 (round 5): COLUMN bands -- `ctiles` consecutive tiles of chunk 0, the same tiles of chunk 1, ... of the last chunk, then the
    // next band (k_tall_diag_fwd's walk).  The band of v is then re-read by every chunk from L2 instead of the Infinity Cache, and (c, t) still
    // only waits for (c - 1, t), whose ticket is smaller by the band's width: started, and depending on smaller tickets only.
    unsigned chunk, tile;
    if (ctiles) {
        const unsigned per_band = ctiles * nchunks, b = ticket / per_band, r = ticket - b * per_band;
        const unsigned cw = (b * ctiles + ctiles <= ntiles) ? ctiles : ntiles - b * ctiles;     // the last band may be narrower
        chunk = r / cw;
        tile = b * ctiles + r % cw;
    } else {
        chunk = ticket / ntiles;
        tile = ticket - chunk * ntiles;
    }
    const int64_t row0 = (int64_t)chunk * DEPTH, row1 = (row0 + DEPTH < nrow) ? row0 + DEPTH : nrow;
    const int64_t span = s_end - s_begin;                                   // the host guarantees span % (U * BLK * NS) == 0: full tiles only
    int64_t sk[U];
    V acc[U], vv[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        sk[k] = s_begin + (((int64_t)tile * U + k) * BLK + threadIdx.x) * NS;
        acc[k] = (V)(S)0;
        vv[k] = ld<false>(reinterpret_cast<const V *>(v + sk[k]));
    }
    const bool use_old = (beta != (S)0);
    const bool full = row0 + DEPTH <= nrow;
    V av[DEPTH][U], uv[DEPTH][U];
    jh_dev_block blk[MIXED ? DEPTH : 1];                                    // MIXED: rows of any elementwise kind (as in k_tall_diag_bidiag)
    if (full) {                                                             // the batch's loads go out BEFORE the wait for the predecessor
        // u first: its addresses are arithmetic, so these loads are in flight while the row table (separate coefficient arrays,
        // rows of several kinds) is still being fetched -- a workgroup that lives for one batch cannot hide that round trip
        // otherwise (256 x 256^3 over separate arrays: chained step 5.7 TB/s against 6.1-6.3 over one slab)
#pragma unroll
        for (int j = 0; j < DEPTH; j++)
#pragma unroll
            for (int k = 0; k < U; k++) uv[j][k] = use_old ? ld<true>(reinterpret_cast<const V *>(u + (row0 + j) * n_scalars + sk[k])) : (V)(S)0;
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            const S *a;
            bool rc = true;
            if constexpr (MIXED) {
                blk[j] = blocks[row0 + j];
                a = (const S *)blk[j].coeff;
                rc = block_reads_coeff(blk[j], false);
            } else
                a = a_base ? a_base + (row0 + j) * a_stride : (const S *)blocks[row0 + j].coeff;
#pragma unroll
            for (int k = 0; k < U; k++) av[j][k] = rc ? ld<true>(reinterpret_cast<const V *>(a + sk[k])) : (V)(S)0;
        }
    }
    if (chunk > 0) {
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(&sync[2 + tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < chunk) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 22)) { atomicOr(err, 1u); break; }    // never hang: flag it and go on
            }
        }
        __syncthreads();
        const S *src = wpart + (int64_t)((chunk - 1) & 1u) * span - s_begin;
#pragma unroll
        for (int k = 0; k < U; k++) acc[k] = __builtin_bit_cast(V, ld_sc1_16(src + sk[k]));
    }
    double nrm = 0.0;
    if (full) {
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            bool diag = true, cj = false, on = true;                        // MIXED: ONE branch per row; a DIAG row takes the straight line
            if constexpr (MIXED) {
                diag = blk[j].kind == JH_OP_DIAG;
                cj = blk[j].adjoint != 0;
                on = blk[j].kind != JH_OP_ZERO;                             // a ZERO row: tmp_i stays 0 (1022) and adds nothing to w (1047)
            }
            if (diag) {
#pragma unroll
                for (int k = 0; k < U; k++) {
                    V t = vmul<S, E, NS, V>(av[j][k], vv[k], cj);           // mul!(tmp, A_i, v)
                    V r = (V)alpha * t;
                    if (use_old) { V s2 = (V)beta * uv[j][k]; r = r + s2; } // u_i .= alpha*tmp .+ beta*u_i
                    st_nt_16(u + (row0 + j) * n_scalars + sk[k], __builtin_bit_cast(U4, r));
                    nrm += vnorm2<S, NS, V>(r);
                    acc[k] = acc[k] + vmul<S, E, NS, V>(av[j][k], r, !cj);  // _m .+= conj(a_i) .* u_i   (1049)
                }
            } else if constexpr (MIXED) {
#pragma unroll
                for (int k = 0; k < U; k++) {
                    V t = on ? apply_block_loaded<S, E, NS, V>(blk[j], vv[k], av[j][k], false, false) : (V)(S)0;
                    V r = (V)alpha * t;
                    if (use_old) { V s2 = (V)beta * uv[j][k]; r = r + s2; }
                    st_nt_16(u + (row0 + j) * n_scalars + sk[k], __builtin_bit_cast(U4, r));
                    nrm += vnorm2<S, NS, V>(r);
                    if (on) acc[k] = acc[k] + apply_block_loaded<S, E, NS, V>(blk[j], r, av[j][k], true, false);
                }
            }
        }
    } else {
        for (int64_t i = row0; i < row1; i++) {
            jh_dev_block b1;
            if (MIXED || !a_base) b1 = blocks[i];
            const S *a = (!MIXED && a_base) ? a_base + i * a_stride : (const S *)b1.coeff;
            const bool on = !MIXED || b1.kind != JH_OP_ZERO, rc = !MIXED || block_reads_coeff(b1, false);
#pragma unroll
            for (int k = 0; k < U; k++) {
                V a1 = rc ? ld<true>(reinterpret_cast<const V *>(a + sk[k])) : (V)(S)0;
                V t;
                if constexpr (MIXED) t = on ? apply_block_loaded<S, E, NS, V>(b1, vv[k], a1, false, false) : (V)(S)0;
                else t = vmul<S, E, NS, V>(a1, vv[k], false);
                V r = (V)alpha * t;
                if (use_old) { V s2 = (V)beta * ld<true>(reinterpret_cast<const V *>(u + i * n_scalars + sk[k])); r = r + s2; }
                st_nt_16(u + i * n_scalars + sk[k], __builtin_bit_cast(U4, r));
                nrm += vnorm2<S, NS, V>(r);
                if constexpr (MIXED) { if (on) acc[k] = acc[k] + apply_block_loaded<S, E, NS, V>(b1, r, a1, true, false); }
                else acc[k] = acc[k] + vmul<S, E, NS, V>(a1, r, true);
            }
        }
    }
    if (chunk + 1 < nchunks) {                                              // hand the ordered partial sum on
        S *dst = wpart + (int64_t)(chunk & 1u) * span - s_begin;
#pragma unroll
        for (int k = 0; k < U; k++) st_sc1_16(dst + sk[k], __builtin_bit_cast(U4, acc[k]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&sync[2 + tile], chunk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
#pragma unroll
        for (int k = 0; k < U; k++) st<false>(reinterpret_cast<V *>(w + sk[k]), acc[k]);
    }
    wg_sum_store<BLK>(nrm, partials + (size_t)chunk * ntiles + tile);     // by (chunk, tile): the fold's order does not depend on the walk
}

// out[s] = sum over parts p = 0..nparts-1 (in that order within a part lane, part lanes in order) of parts[p][s - s_begin]:
// the second stage of the split-row walk.  64 vector lanes x 16 part lanes per workgroup; fp64 accumulation (exact
// conversions of S, so the fold adds no rounding of its own until the final cast); fixed order => deterministic.
template <typename S, int NS>
__global__ __launch_bounds__(1024) void k_fold_parts(const S *__restrict__ parts, int64_t part_stride, int nparts, S *__restrict__ out,
                                                     int64_t s_begin, int64_t s_end)
{
    typedef typename vec_of<S, NS>::type V;
    __shared__ double sm[16][NS][64];
    const int v = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t s = s_begin + ((int64_t)blockIdx.x * 64 + v) * NS;
    const bool ok = s < s_end;
    double acc[NS];
#pragma unroll
    for (int e = 0; e < NS; e++) acc[e] = 0.0;
    if (ok) {
        const S *src = parts + (s - s_begin);
#pragma unroll 4
        for (int p = q; p < nparts; p += 16) {
            const V x = ld<false>(reinterpret_cast<const V *>(src + (int64_t)p * part_stride));
#pragma unroll
            for (int e = 0; e < NS; e++) acc[e] += (double)x[e];
        }
    }
#pragma unroll
    for (int e = 0; e < NS; e++) sm[q][e][v] = acc[e];
    __syncthreads();
    if (q == 0 && ok) {
        V r;
#pragma unroll
        for (int e = 0; e < NS; e++) {
            double t = acc[e];
#pragma unroll
            for (int qq = 1; qq < 16; qq++) t += sm[qq][e][v];
            r[e] = (S)t;
        }
        st<false>(reinterpret_cast<V *>(out + s), r);
    }
}

// out = c0 * t + c1 * out (c1 == 0: out = c0 * t) with partial ||out||^2: the epilogue of the fused adjoint update when the
// row sum itself went through the split walk (real coefficients: a complex vector is 2n reals here)
template <typename S>
__global__ __launch_bounds__(256) void k_axpby_norm(S *__restrict__ out, const S *__restrict__ t, int64_t n_scalars, S c0, S c1,
                                                    double *__restrict__ partials)
{
    double nrm = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_scalars; i += (int64_t)gridDim.x * 256) {
        S r = c0 * t[i];
        if (c1 != (S)0) { const S s2 = c1 * out[i]; r = r + s2; }
        out[i] = r;
        nrm += (double)r * (double)r;
    }
    wg_sum_store<256>(nrm, partials + blockIdx.x);
}

// fold the per-workgroup partials deterministically: workgroup b sums the contiguous chunk
// [b*chunk, (b+1)*chunk) in a fixed order and writes out[b]; launched twice for large counts (1M -> 1024 -> 1)
// accum != 0 (single-workgroup launches only): the sum is ADDED to what out[0] holds -- the deferred ||u||^2 of a step that
// is enqueued range by range (jh_blockop_bidiag_step_range with normsq == NULL); stream order makes the additions sequential
__global__ void k_sum_partials(const double *__restrict__ partials, int64_t n, int64_t chunk, double *__restrict__ out, int accum)
{
    const int64_t lo = (int64_t)blockIdx.x * chunk;
    const int64_t hi = lo + chunk < n ? lo + chunk : n;
    double v = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) v += partials[i];
    double *slot = out + blockIdx.x;
    const double prev = (accum && threadIdx.x == 0) ? *slot : 0.0;
    wg_sum_store<256>(v, slot);
    if (accum && threadIdx.x == 0) *slot += prev;
}

// ------------------------------------------------------------------ fused JetSum of tall operators ---------
// d_i = sum_k sign_k * (scale_k * (a_{k,i} .* m))      (JetSum_df!, src/Jets.jl:639-646, of terms A_k or s_k*A_k)
// m   = sum_k sign_k * (sum_i conj(a_{k,i}) .* (scale_k * d_i))                 (JetSum_df'!, 648-655)
// for up to JH_SUM_MAX tall all-DIAG operators of identical shape, in ONE pass over the range vector: every
// coefficient slab is read once, d is written (forward) or read (adjoint) once -- the unfused chain moves 5 range-sized
// streams per term.  Rounding sequence == the unfused chain: product, scale (exact when 1), signed add, terms in order;
// in the adjoint each term's rows are summed in order into its own accumulator before the terms are combined.
// Up to four terms run on the KM = 4 instantiations (two packs per lane); five to eight on KM = 8 (one pack per lane, to stay within
// the registers of four waves per SIMD) -- round 3: eight terms used to be two launches, the second re-reading and re-writing d.
constexpr int JH_SUM_MAX = 16;           // coefficient streams per launch (forward since round 4, adjoint: sixteen accumulators, knob sum_adj_group)
constexpr int JH_SUM_ADJ_MAX = 8;
// Round 5: the load section of both kernels is STRAIGHT-LINE code.  The round-4 kernels decided per term and row whether the term exists
// (t < k) and how its row is addressed (a slab's stride or the block table) -- two scalar branches and an s_waitcnt lgkmcnt(0) in front
// of every one of up to sixteen loads -- and held sixteen bases, strides, scales and signs in SGPRs: the sixteen-term forward was out of
// SGPRs (106 of 106) and spilled them into VGPR lanes (138 v_readlane + 108 v_writelane in its ISA, 148 VGPRs = three waves per SIMD).
// Now: STRIDED (every term's diagonals in one slab, all with the same row stride: one 64-bit row offset per row, a base per term) or the
// block TABLES (every term's row pointer is read, all scalar loads issued together) is a template parameter; terms beyond k are filled
// with term 0's addresses on the host -- their loads are L1 hits, their arithmetic is computed and dropped by a wave-uniform select --
// and sign * scale arrives as ONE factor per term.
struct SumArgs {
    const void *a0[JH_SUM_MAX];          // STRIDED: row 0 of term t's coefficients; else term t's device block table (jh_dev_block *)
    int64_t stride;                      // STRIDED: scalars from one row to the next (the same for every term)
    double coef[JH_SUM_MAX];             // forward: sign_t * scale_t (-(s*x) == (-s)*x exactly); adjoint: scale_t
    double sign[JH_SUM_MAX];             // adjoint: the sign of term t's ordered row sum in the final combination
    float coef32[JH_SUM_MAX], sign32[JH_SUM_MAX];   // the same in Float32, for 32-bit elements: read straight into SGPRs (a double converted in the
                                                    // kernel lands in a VGPR, and the compiler then keeps sixteen splatted packs live: 64 registers)
    int k;
};
template <typename S> __device__ inline S sum_coef(const SumArgs &a, int t) { if constexpr (sizeof(S) == 4) return a.coef32[t]; else return a.coef[t]; }
template <typename S> __device__ inline S sum_sign(const SumArgs &a, int t) { if constexpr (sizeof(S) == 4) return a.sign32[t]; else return a.sign[t]; }

// WIDE (S = float; round 5): some scale_k is Julia's Float64 (JH_SCALAR_WIDE: `1.0*A1 - 2.0*A2 + 3.0*A3` on Float32 operators, the reference's
// own docstring example, src/Jets.jl:686) -- the scalar stage `_d .= a * tmp` (1159) is then the promoted product rounded once,
// Float32(a * Float64(tmp)); the signed add stays a Float32 add (`broadcast!(sgn, d, d, _d)`, 644).  EVERY term of such a launch is
// computed that way with scale_k as a double: for a scalar that is T(a) in Julia the host passes double(T(a)), and
// Float32(double(a32) * Float64(x)) == a32 * x in Float32 arithmetic (the Float64 product of two Float32 values is exact, so both
// round the exact product once) -- one instantiation serves sums that mix wide and narrow scalars, with the chain's bits.
template <typename S, int E, int NS, int U, int BLK, int KM, bool STRIDED, bool WIDE = false>
__global__ __launch_bounds__(BLK) void k_tall_sum_fwd(SumArgs args, int64_t nrow, int rows_per_wg, const S *__restrict__ m,
                                                      S *__restrict__ d, int64_t n_scalars, unsigned ntiles, int accumulate)
{
    // accumulate != 0: continue the left-to-right sum from what d holds (terms 17..32, ... of a long JetSum: same sequence)
    typedef typename vec_of<S, NS>::type V;
    const unsigned tile = blockIdx.x % ntiles, grp = blockIdx.x / ntiles;
    const int64_t s0 = ((int64_t)tile * U * BLK + threadIdx.x) * NS;
    const int64_t i0 = (int64_t)grp * rows_per_wg;
    const int64_t i1 = (i0 + rows_per_wg < nrow) ? i0 + rows_per_wg : nrow;
    bool ok[U];
    int64_t sk[U];
    V mv[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        ok[k] = (s0 + (int64_t)k * BLK * NS) < n_scalars;
        sk[k] = ok[k] ? s0 + (int64_t)k * BLK * NS : 0;
        mv[k] = ld<false>(reinterpret_cast<const V *>(m + sk[k]));
    }
    // one row per iteration, its KM coefficient packs (x U) in flight.  The row loop is kept rolled: left to itself the compiler unrolls
    // the eight-stream shape to 241 VGPRs (one wave per SIMD: 1.6 TB/s)
#pragma unroll 1
    for (int64_t i = i0; i < i1; i++) {
        V av[KM][U], dv[U];
        if constexpr (STRIDED) {
            const int64_t roff = i * args.stride;
#pragma unroll
            for (int t = 0; t < KM; t++) {
                const S *a = (const S *)args.a0[t] + roff;
#pragma unroll
                for (int k = 0; k < U; k++) av[t][k] = ld<true>(reinterpret_cast<const V *>(a + sk[k]));
            }
        } else {
            const S *ap[KM];
#pragma unroll
            for (int t = 0; t < KM; t++) ap[t] = (const S *)((const jh_dev_block *)args.a0[t])[i].coeff;     // KM scalar loads, one wait
#pragma unroll
            for (int t = 0; t < KM; t++)
#pragma unroll
                for (int k = 0; k < U; k++) av[t][k] = ld<true>(reinterpret_cast<const V *>(ap[t] + sk[k]));
        }
#pragma unroll
        for (int k = 0; k < U; k++)
            dv[k] = accumulate ? ld<true>(reinterpret_cast<const V *>(d + i * n_scalars + sk[k])) : (V)(S)0;   // d .= 0  (639-640)
#pragma unroll
        for (int k = 0; k < U; k++) {
            V acc = dv[k];
#pragma unroll
            for (int t = 0; t < KM; t++) {
                const V prod = vmul<S, E, NS, V>(av[t][k], mv[k], false);            // mul!(_d, A_t, m)
                V term;
                if constexpr (WIDE) {
#pragma unroll
                    for (int e = 0; e < NS; e++) term[e] = (S)(args.coef[t] * (double)prod[e]);
                } else {
                    term = sum_coef<S>(args, t) * prod;                              // (s_t * .) then the sign: -(s*x) == (-s)*x exactly
                }
                const V sum = acc + term;                                            // broadcast!(sgn, d, d, _d)
                acc = (t < args.k) ? sum : acc;                                      // (a term beyond k: dropped, wave-uniform)
            }
            if (ok[k]) st<true>(reinterpret_cast<V *>(d + i * n_scalars + sk[k]), acc);
        }
    }
}

// WIDE: as in the forward -- the adjoint's scalar stage `tmp .= conj(a) * d` (1160) is Float32(a * Float64(d_i)) per element
template <typename S, int E, int NS, int U, int DEPTH, int BLK, int KM, bool STRIDED, bool WIDE = false>
__global__ __launch_bounds__(BLK) void k_tall_sum_adj(SumArgs args, int64_t nrow, S *__restrict__ out, const S *__restrict__ in,
                                                      int64_t n_scalars, int accumulate)
{
    typedef typename vec_of<S, NS>::type V;
    const int64_t s0 = ((int64_t)blockIdx.x * U * BLK + threadIdx.x) * NS;
    bool ok[U];
    int64_t sk[U];
    V acc[KM][U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        ok[k] = (s0 + (int64_t)k * BLK * NS) < n_scalars;
        sk[k] = ok[k] ? s0 + (int64_t)k * BLK * NS : 0;
#pragma unroll
        for (int t = 0; t < KM; t++) acc[t][k] = (V)(S)0;
    }
    // nrow == 1: mul!(_m, op', _d) writes directly (1051) -- the product itself, not 0 + product, which differs for a product of -0.
    // Starting the accumulators at -0 instead gives exactly that ((-0) + p == p for every p, signed zeros included) without a select per add.
    if (nrow == 1) {
#pragma unroll
        for (int k = 0; k < U; k++)
#pragma unroll
            for (int t = 0; t < KM; t++) acc[t][k] = (V)(S)(-0.0);
    }
    auto batch = [&](int64_t i, auto depth_tag) {                             // rows [i, i + D): all loads, then the arithmetic, rows in order
        constexpr int D = decltype(depth_tag)::value;
        V dv[D][U], av[D][KM][U];
#pragma unroll
        for (int j = 0; j < D; j++) {
#pragma unroll
            for (int k = 0; k < U; k++) dv[j][k] = ld<true>(reinterpret_cast<const V *>(in + (i + j) * n_scalars + sk[k]));
            if constexpr (STRIDED) {
                const int64_t roff = (i + j) * args.stride;
#pragma unroll
                for (int t = 0; t < KM; t++) {
                    const S *a = (const S *)args.a0[t] + roff;
#pragma unroll
                    for (int k = 0; k < U; k++) av[j][t][k] = ld<true>(reinterpret_cast<const V *>(a + sk[k]));
                }
            } else {
                const S *ap[KM];
#pragma unroll
                for (int t = 0; t < KM; t++) ap[t] = (const S *)((const jh_dev_block *)args.a0[t])[i + j].coeff;   // KM scalar loads, one wait
#pragma unroll
                for (int t = 0; t < KM; t++)
#pragma unroll
                    for (int k = 0; k < U; k++) av[j][t][k] = ld<true>(reinterpret_cast<const V *>(ap[t] + sk[k]));
            }
        }
#pragma unroll
        for (int j = 0; j < D; j++)
#pragma unroll
            for (int t = 0; t < KM; t++)
#pragma unroll
                for (int k = 0; k < U; k++) {
                    V sd;
                    if constexpr (WIDE) {
#pragma unroll
                        for (int e = 0; e < NS; e++) sd[e] = (S)(args.coef[t] * (double)dv[j][k][e]);
                    } else {
                        sd = sum_coef<S>(args, t) * dv[j][k];
                    }
                    acc[t][k] = acc[t][k] + vmul<S, E, NS, V>(av[j][t][k], sd, true);                      // conj(a_i) .* (s_t * d_i), rows in order
                }
    };
    int64_t i = 0;
    for (; i + DEPTH <= nrow; i += DEPTH) batch(i, std::integral_constant<int, DEPTH>{});
    for (; i < nrow; i++) batch(i, std::integral_constant<int, 1>{});
#pragma unroll
    for (int k = 0; k < U; k++) {
        V r = accumulate ? ld<false>(reinterpret_cast<const V *>(out + sk[k])) : (V)(S)0;   // m .= 0  (648-649), or the sum so far
#pragma unroll
        for (int t = 0; t < KM; t++) {
            const V sum = r + sum_sign<S>(args, t) * acc[t][k];                  // broadcast!(sgn, m, m, _m)
            r = (t < args.k) ? sum : r;
        }
        if (ok[k]) st<false>(reinterpret_cast<V *>(out + sk[k]), r);
    }
}

// ---- sums of up to EIGHT terms keep round 4's kernels: their loads sit behind `t < k` (a sum of three terms on the four-stream shape loads three
// streams, not four), which is what such short sums want -- same box, 3 terms: forward 5.9-6.1 against 5.5-5.6 TB/s for the padded straight-line
// form below, adjoint 5.9-6.6 against 5.1; 8 terms within noise (profiles/ab_r05_jetsum.txt).  With few streams there is no SGPR pressure
// either.  Addressing and coefficients as in the straight-line kernels (STRIDED is a template parameter, sign * scale one factor).
template <typename S, int E, int NS, int U, int BLK, int KM, int D, bool STRIDED, bool WIDE = false>
__global__ __launch_bounds__(BLK) void k_tall_sum_fwd_few(SumArgs args, int64_t nrow, int rows_per_wg, const S *__restrict__ m,
                                                      S *__restrict__ d, int64_t n_scalars, unsigned ntiles, int accumulate)
{
    // accumulate != 0: continue the left-to-right sum from what d holds (terms 5..8, 9..12, ... of a long JetSum: same sequence)
    typedef typename vec_of<S, NS>::type V;
    const unsigned tile = blockIdx.x % ntiles, grp = blockIdx.x / ntiles;
    const int64_t s0 = ((int64_t)tile * U * BLK + threadIdx.x) * NS;
    const int64_t i0 = (int64_t)grp * rows_per_wg;
    const int64_t i1 = (i0 + rows_per_wg < nrow) ? i0 + rows_per_wg : nrow;
    bool ok[U];
    int64_t sk[U];
    V mv[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        ok[k] = (s0 + (int64_t)k * BLK * NS) < n_scalars;
        sk[k] = ok[k] ? s0 + (int64_t)k * BLK * NS : 0;
        mv[k] = ld<false>(reinterpret_cast<const V *>(m + sk[k]));
    }
    // D rows' loads in flight.  The row loop is kept rolled: left to itself the compiler unrolls the eight-stream shape to 241 VGPRs
    // (one wave per SIMD: 1.6 TB/s)
#pragma unroll 1
    for (int64_t i = i0; i < i1; i += D) {
        V av[D][KM][U], dv[D][U];
#pragma unroll
        for (int j = 0; j < D; j++) {
            const int64_t ij = i + j < i1 ? i + j : i1 - 1;                    // clamped: branch-free loads of valid memory
#pragma unroll
            for (int t = 0; t < KM; t++)
                if (t < args.k) {
                    const S *a = STRIDED ? (const S *)args.a0[t] + ij * args.stride : (const S *)((const jh_dev_block *)args.a0[t])[ij].coeff;
#pragma unroll
                    for (int k = 0; k < U; k++) av[j][t][k] = ld<true>(reinterpret_cast<const V *>(a + sk[k]));
                }
#pragma unroll
            for (int k = 0; k < U; k++)
                dv[j][k] = accumulate ? ld<true>(reinterpret_cast<const V *>(d + ij * n_scalars + sk[k])) : (V)(S)0;   // d .= 0  (639-640)
        }
#pragma unroll
        for (int j = 0; j < D; j++)
#pragma unroll
            for (int k = 0; k < U; k++) {
                V acc = dv[j][k];
#pragma unroll
                for (int t = 0; t < KM; t++)
                    if (t < args.k) {
                        V prod = vmul<S, E, NS, V>(av[j][t][k], mv[k], false);       // mul!(_d, A_t, m)
                        V term;
                        if constexpr (WIDE) {
                            const double sd = args.coef[t];                           // sign * scale: -(Float32(s*x)) == Float32((-s)*x) exactly
#pragma unroll
                            for (int e = 0; e < NS; e++) term[e] = (S)(sd * (double)prod[e]);
                        } else {
                            term = sum_coef<S>(args, t) * prod;                       // (s_t * .) then the sign: -(s*x) == (-s)*x exactly
                        }
                        acc = acc + term;                                            // broadcast!(sgn, d, d, _d)
                    }
                if (ok[k] && i + j < i1) st<true>(reinterpret_cast<V *>(d + (i + j) * n_scalars + sk[k]), acc);
            }
    }
}

// WIDE: as in the forward -- the adjoint's scalar stage `tmp .= conj(a) * d` (1160) is Float32(a * Float64(d_i)) per element
template <typename S, int E, int NS, int U, int DEPTH, int BLK, int KM, bool STRIDED, bool WIDE = false>
__global__ __launch_bounds__(BLK) void k_tall_sum_adj_few(SumArgs args, int64_t nrow, S *__restrict__ out, const S *__restrict__ in,
                                                      int64_t n_scalars, int accumulate)
{
    typedef typename vec_of<S, NS>::type V;
    const int64_t s0 = ((int64_t)blockIdx.x * U * BLK + threadIdx.x) * NS;
    bool ok[U];
    int64_t sk[U];
    V acc[KM][U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        ok[k] = (s0 + (int64_t)k * BLK * NS) < n_scalars;
        sk[k] = ok[k] ? s0 + (int64_t)k * BLK * NS : 0;
#pragma unroll
        for (int t = 0; t < KM; t++) acc[t][k] = (V)(S)0;
    }
    const bool direct = (nrow == 1);                                            // mul!(_m, op', _d) writes directly (1051)
    for (int64_t i = 0; i < nrow; i += DEPTH) {
        V dv[DEPTH][U], av[DEPTH][KM][U];
#pragma unroll
        for (int j = 0; j < DEPTH; j++)
            if (i + j < nrow) {
#pragma unroll
                for (int k = 0; k < U; k++) dv[j][k] = ld<true>(reinterpret_cast<const V *>(in + (i + j) * n_scalars + sk[k]));
#pragma unroll
                for (int t = 0; t < KM; t++)
                    if (t < args.k) {
                        const S *a = STRIDED ? (const S *)args.a0[t] + (i + j) * args.stride : (const S *)((const jh_dev_block *)args.a0[t])[i + j].coeff;
#pragma unroll
                        for (int k = 0; k < U; k++) av[j][t][k] = ld<true>(reinterpret_cast<const V *>(a + sk[k]));
                    }
            }
#pragma unroll
        for (int j = 0; j < DEPTH; j++)
            if (i + j < nrow) {
#pragma unroll
                for (int t = 0; t < KM; t++)
                    if (t < args.k) {
#pragma unroll
                        for (int k = 0; k < U; k++) {
                            V sd;
                            if constexpr (WIDE) {
#pragma unroll
                                for (int e = 0; e < NS; e++) sd[e] = (S)(args.coef[t] * (double)dv[j][k][e]);
                            } else {
                                sd = sum_coef<S>(args, t) * dv[j][k];
                            }
                            V p = vmul<S, E, NS, V>(av[j][t][k], sd, true);                                // conj(a_i) .* (s_t * d_i)
                            acc[t][k] = direct ? p : acc[t][k] + p;
                        }
                    }
            }
    }
#pragma unroll
    for (int k = 0; k < U; k++) {
        V r = accumulate ? ld<false>(reinterpret_cast<const V *>(out + sk[k])) : (V)(S)0;   // m .= 0  (648-649), or the sum so far
#pragma unroll
        for (int t = 0; t < KM; t++)
            if (t < args.k) r = r + sum_sign<S>(args, t) * acc[t][k];            // broadcast!(sgn, m, m, _m)
        if (ok[k]) st<false>(reinterpret_cast<V *>(out + sk[k]), r);
    }
}


// ------------------------------------------------------------------ general path --------------
template <typename S, int E> struct elem {
    S re, im;
};
template <typename S, int E> __device__ inline elem<S, E> eload(const S *p, int64_t idx)
{
    typedef const S __attribute__((address_space(1))) *gp;
    elem<S, E> r;
    r.re = ((gp)p)[idx * E];
    r.im = (E == 2) ? ((gp)p)[idx * E + 1] : (S)0;
    return r;
}
template <typename S, int E> __device__ inline void estore(S *p, int64_t idx, elem<S, E> v)
{
    p[idx * E] = v.re;
    if (E == 2) p[idx * E + 1] = v.im;
}
template <typename S, int E> __device__ inline elem<S, E> emul(elem<S, E> a, elem<S, E> b)
{
    elem<S, E> r;
    if (E == 1) { r.re = a.re * b.re; r.im = 0; }
    else { r.re = a.re * b.re - a.im * b.im; r.im = a.re * b.im + a.im * b.re; }
    return r;
}
template <typename S, int E> __device__ inline elem<S, E> eadd(elem<S, E> a, elem<S, E> b)
{
    elem<S, E> r;
    r.re = a.re + b.re;
    r.im = (E == 2) ? a.im + b.im : (S)0;
    return r;
}

// child mul! of an elementwise block applied to one element x at local index e.
// `transposed` = we are inside df'! (so the child is op').  Effective conjugation = adjoint XOR transposed.
// `fmode` = we are inside f! (JetBlock_f!, 988-1008): a SQUARE child squares its input instead of applying its Jacobian.
template <typename S, int E>
__device__ inline elem<S, E> apply_block(const jh_dev_block &b, elem<S, E> x, int64_t e, bool transposed, bool fmode = false)
{
    const bool cj = (b.adjoint != 0) != transposed;
    switch (b.kind) {
    case JH_OP_IDENTITY: return x;
    case JH_OP_SQUARE: {
        if (fmode && !b.adjoint) return emul<S, E>(x, x);   // d .= m.^2   (test/runtests.jl:19)
        elem<S, E> a = eload<S, E>((const S *)b.coeff, e);  // mo
        a.re = a.re + a.re;                                  // 2 .* mo (exact)
        a.im = (E == 2) ? a.im + a.im : (S)0;
        if (E == 2 && cj) a.im = -a.im;
        return emul<S, E>(a, x);                             // (2 .* mo) .* dm / conj.(2 .* mo) .* dd   (test/runtests.jl:20)
    }
    case JH_OP_SCALE: {
        elem<S, E> a;
        a.re = (S)b.sre;
        a.im = (E == 2) ? (cj ? -(S)b.sim : (S)b.sim) : (S)0;
        if (E == 2 && b.real_scale) {                  // a REAL scalar multiplies part by part (Julia's a::Real * z)
            x.re = a.re * x.re;
            x.im = a.re * x.im;
            return x;
        }
        return emul<S, E>(a, x);                       // d .= a*m / m .= conj(a)*d   (1159-1160)
    }
    case JH_OP_DIAG: {
        elem<S, E> a = eload<S, E>((const S *)b.coeff, e);
        if (E == 2 && cj) a.im = -a.im;
        return emul<S, E>(a, x);                       // diagonal .* m / conj.(diagonal) .* d
    }
    default: {
        elem<S, E> z;
        z.re = 0; z.im = 0;
        return z;
    }
    }
}

// Grid of the general kernels: 1-D, (line, tile) decoded XCD-aware.  A "line" is a block row (forward) or a block column
// (adjoint); a tile is 256 lanes' worth of its elements.  The lines of one tile read the SAME input elements (block j of
// m is used by every block row; block i of d by every block column).  Workgroups are dispatched round-robin over the 8
// XCDs, each with its own L2, so the lines of a tile get workgroup ids 8 apart: same XCD, dispatched together -- the shared
// input comes from HBM once and from that L2 afterwards (without this an M x K operator with big blocks re-reads every
// input block once per line: profiles/bench_blocks_nl_r01.txt).
// Late round 4: the group of tiles that every line walks before the next group starts is 8 << k tiles (k in bits 28..30 of `ntiles`, knob
// general_band): bands of 32 tiles stream 128 KiB of every block linearly where 8 tiles made every workgroup jump a whole block after 32 KiB
// (the tall forward's column bands); any multiple of 8 keeps the lines of a tile on one XCD.
__device__ inline void general_line_tile(unsigned nlines, unsigned ntiles, int64_t &line, int64_t &tile)
{
    const unsigned k = (ntiles >> 28) & 7u, T = 8u << k;
    if (ntiles & 0x80000000u) {                              // knob general_xcd = 0 (A/B measurements): tile fastest, line by line
        const unsigned padded = ((ntiles & 0x0fffffffu) + T - 1u) / T * T;
        line = blockIdx.x / padded;
        tile = blockIdx.x - (unsigned)line * padded;
        return;
    }
    const unsigned per = T * nlines;
    const unsigned grp = blockIdx.x / per, rem = blockIdx.x - grp * per;
    line = rem >> (3u + k);
    tile = (int64_t)grp * T + (rem & (T - 1u));
}

// JetBlock_df! (1010-1032): one line per block row, threads over the row's elements.
template <typename S, int E>
__global__ void k_block_fwd_general(const jh_dev_block *__restrict__ blocks, int64_t nrow, int64_t ncol,
                                    const int64_t *__restrict__ row_off, const int64_t *__restrict__ col_off,
                                    const S *__restrict__ m, S *__restrict__ d, int fmode, unsigned ntiles,
                                    int64_t q_per_part, S *__restrict__ slabs, int64_t slab_stride,
                                    const S *__restrict__ dense_prod = nullptr, int64_t prod_stride = 0)
{
    // dense_prod != null (dense_mixed_fwd): block (i, j) of kind DENSE contributes the product A_ij m_j a batched GEMV launch has
    // left, rounded like the reference's dtmp (1024), at the row's elements of slab j
    int64_t i, tile;                                                       // block row, tile
    general_line_tile((unsigned)nrow, ntiles, i, tile);
    ntiles &= 0x0fffffffu;
    if (tile >= ntiles) return;
    // q_per_part > 0: split walk (many block columns of small blocks, general_parts): workgroup row blockIdx.y sums its own
    // columns, in order, from zero into slab blockIdx.y; k_fold_general adds d as found and the slabs afterwards
    const bool split = q_per_part > 0;
    int64_t j_lo = 0, j_hi = ncol;
    if (split) {
        j_lo = (int64_t)blockIdx.y * q_per_part;
        j_hi = j_lo + q_per_part < ncol ? j_lo + q_per_part : ncol;
        d = slabs + (int64_t)blockIdx.y * slab_stride;
    }
    const int64_t n = row_off[i + 1] - row_off[i];
    for (int64_t e = tile * 256 + threadIdx.x; e < n; e += (int64_t)ntiles * 256) {
        elem<S, E> acc;
        bool touched = split;
        if (ncol > 1 && !split) { acc = eload<S, E>(d, row_off[i] + e); }   // `_d .+=` accumulates into d as found (1024 / 1001)
        else { acc.re = 0; acc.im = 0; }
        for (int64_t j = j_lo; j < j_hi; j++) {                            // (1020)
            const jh_dev_block b = blocks[i + j * nrow];
            if (b.kind == JH_OP_ZERO && !fmode) continue;                  // (1022); JetBlock_f! has no such test
            elem<S, E> p;
            p.re = 0; p.im = 0;                                            // a zero block's `d .= 0` (942): no load -- its column may be shorter than this row
            if (b.kind == JH_OP_DENSE) {
                p = eload<S, E>(dense_prod, j * prod_stride + row_off[i] + e);   // mul!(dtmp, op, _m), computed by the column's batch
            } else if (b.kind != JH_OP_ZERO) {
                elem<S, E> x = eload<S, E>(m, col_off[j] + e);
                p = apply_block<S, E>(b, x, e, false, fmode != 0);         // mul!(dtmp, op, _m)
            }
            acc = (ncol > 1) ? eadd<S, E>(acc, p) : p;                     // (1024) / (1026)
            touched = true;
        }
        if (touched) estore<S, E>(d, row_off[i] + e, acc);
    }
}

// JetBlock_df'! (1034-1057): grid.y = block column, threads over the column's elements.
template <typename S, int E>
__global__ void k_block_adj_general(const jh_dev_block *__restrict__ blocks, int64_t nrow, int64_t ncol,
                                    const int64_t *__restrict__ row_off, const int64_t *__restrict__ col_off,
                                    S *__restrict__ m, const S *__restrict__ d, unsigned ntiles,
                                    int64_t q_per_part, S *__restrict__ slabs, int64_t slab_stride,
                                    const S *__restrict__ dense_prod = nullptr, int64_t prod_stride = 0)
{
    // dense_prod != null (dense_mixed_adj): block (i, j) of kind DENSE contributes A_ij' d_i, left by the column's batch, rounded
    // like the reference's mtmp (1049), at column j's elements of slab i
    int64_t j, tile;                                                       // block column, tile
    general_line_tile((unsigned)ncol, ntiles, j, tile);
    ntiles &= 0x0fffffffu;
    if (tile >= ntiles) return;
    int64_t i_lo = 0, i_hi = nrow;                                         // q_per_part > 0: split walk over the block rows
    if (q_per_part > 0) {
        i_lo = (int64_t)blockIdx.y * q_per_part;
        i_hi = i_lo + q_per_part < nrow ? i_lo + q_per_part : nrow;
        m = slabs + (int64_t)blockIdx.y * slab_stride;
    }
    const int64_t n = col_off[j + 1] - col_off[j];
    for (int64_t e = tile * 256 + threadIdx.x; e < n; e += (int64_t)ntiles * 256) {
        elem<S, E> acc;
        acc.re = 0; acc.im = 0;                                            // `_m .= 0` when nrow > 1 (1042)
        bool touched = (nrow > 1);
        for (int64_t i = i_lo; i < i_hi; i++) {                            // (1045)
            const jh_dev_block b = blocks[i + j * nrow];
            if (b.kind == JH_OP_ZERO) continue;                            // (1047)
            elem<S, E> p;
            if (b.kind == JH_OP_DENSE) {
                p = eload<S, E>(dense_prod, i * prod_stride + col_off[j] + e);   // mul!(mtmp, op', _d), computed by the column's batch
            } else {
                elem<S, E> x = eload<S, E>(d, row_off[i] + e);
                p = apply_block<S, E>(b, x, e, true);                      // mul!(mtmp, op', _d)
            }
            acc = (nrow > 1) ? eadd<S, E>(acc, p) : p;                     // (1049) / (1051)
            touched = true;
        }
        if (touched) estore<S, E>(m, col_off[j] + e, acc);
    }
}

// ------------------------------------------------------------------ general path with SMALL dense children ----
// Operators that mix dense matrices (adjointed or not) with the elementwise kinds -- the reference's own 3 x 4 test operator
// (test/runtests.jl:622-695: JopBaz children, one of them adjointed, Jacobians of JopBar, zero blocks) -- used to run the
// reference's loop literally: one child launch + one accumulate launch per non-zero block.  For SMALL children that is pure
// launch overhead.  Here ONE launch does the whole loop: a thread owns one element of an output line (block row of d, or block
// column of m), walks the line's blocks in the reference's order and forms a dense child's dot product itself, sequentially
// from zero, product rounded then added -- the oracle's loop, so forward AND adjoint are bit-identical to it (the per-child
// kernels reduce the adjoint's dot in fp64 across a wave: tolerance parity).  Used while every matrix is at most 256 KiB.
template <typename S, int E>
__device__ inline elem<S, E> dense_child_dot(const jh_dev_block &b, int64_t nr, int64_t nc, const S *__restrict__ x, int64_t e, bool transposed)
{
    const S *A = (const S *)b.coeff;                                       // column-major nr x nc
    const bool adj = (b.adjoint != 0) != transposed;                       // (op')' = op
    elem<S, E> s;
    s.re = 0; s.im = 0;
    if (!adj) {                                                            // d[e] = sum_c A[e, c] x[c]        (test/runtests.jl:27)
        for (int64_t c = 0; c < nc; c++) s = eadd<S, E>(s, emul<S, E>(eload<S, E>(A, e + c * nr), eload<S, E>(x, c)));
    } else {                                                               // m[e] = sum_r conj(A[r, e]) x[r]  (test/runtests.jl:28)
        for (int64_t r = 0; r < nr; r++) {
            elem<S, E> a = eload<S, E>(A, r + e * nr);
            if (E == 2) a.im = -a.im;
            s = eadd<S, E>(s, emul<S, E>(a, eload<S, E>(x, r)));
        }
    }
    return s;
}

// transposed == 0: JetBlock_df! / JetBlock_f! (fmode), line = block row; transposed == 1: JetBlock_df'!, line = block column.
// dims: per block (column-major like `blocks`) the matrix shape {nr, nc} of a DENSE child (unused for the other kinds).
template <typename S, int E>
__global__ __launch_bounds__(256) void k_block_loop_small(const jh_dev_block *__restrict__ blocks, const int64_t *__restrict__ dims,
                                                          int64_t nrow, int64_t ncol, const int64_t *__restrict__ row_off,
                                                          const int64_t *__restrict__ col_off, S *__restrict__ out,
                                                          const S *__restrict__ in, int transposed, int fmode)
{
    const int64_t line = blockIdx.y;
    const int64_t *out_off = transposed ? col_off : row_off, *in_off = transposed ? row_off : col_off;
    const int64_t n = out_off[line + 1] - out_off[line];
    const int64_t nsum = transposed ? nrow : ncol;                         // blocks walked per line
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        elem<S, E> acc;
        acc.re = 0; acc.im = 0;
        bool touched = transposed ? (nrow > 1) : false;                    // `_m .= 0` (1042)
        if (!transposed && ncol > 1) acc = eload<S, E>(out, out_off[line] + e);   // `_d .+=` into d as found (1024 / 1001)
        for (int64_t q = 0; q < nsum; q++) {
            const int64_t bi = transposed ? q + line * nrow : line + q * nrow;
            const jh_dev_block b = blocks[bi];
            if (b.kind == JH_OP_ZERO && !fmode) continue;                  // (1022 / 1047); JetBlock_f! applies the zero block (adds 0)
            elem<S, E> p;
            p.re = 0; p.im = 0;
            if (b.kind == JH_OP_DENSE) p = dense_child_dot<S, E>(b, dims[2 * bi], dims[2 * bi + 1], in + in_off[q] * E, e, transposed != 0);
            else if (b.kind != JH_OP_ZERO) p = apply_block<S, E>(b, eload<S, E>(in, in_off[q] + e), e, transposed != 0, fmode != 0);
            acc = (nsum > 1) ? eadd<S, E>(acc, p) : p;                     // (1024 / 1049) accumulate, (1026 / 1051) direct
            touched = true;
        }
        if (touched) estore<S, E>(out, out_off[line] + e, acc);
    }
}

// 16-byte-per-lane variants of the two general kernels, used when every block offset, block length and
// coefficient pointer is a multiple of 16 bytes.  Same loop order and rounding as the scalar versions.
// Blocks whose loads are issued together per thread.  4 was measured no faster than 1 (0-10 % slower, within run-to-run spread) on every M x K shape but the tall
// mixed adjoint (profiles/exp_r01_general_prefetch.txt): with one pack per lane and a full-size grid the chip already has
// enough loads in flight, the extra registers only cost occupancy.
constexpr int GENERAL_Q = 1;

template <typename S, int E, int NS>
__global__ void k_block_fwd_general_vec(const jh_dev_block *__restrict__ blocks, int64_t nrow, int64_t ncol,
                                        const int64_t *__restrict__ row_off, const int64_t *__restrict__ col_off,
                                        const S *__restrict__ m, S *__restrict__ d, int fmode, unsigned ntiles,
                                        int64_t q_per_part, S *__restrict__ slabs, int64_t slab_stride)
{
    typedef typename vec_of<S, NS>::type V;
    int64_t i, tile;                                                       // block row, tile
    general_line_tile((unsigned)nrow, ntiles, i, tile);
    ntiles &= 0x0fffffffu;
    if (tile >= ntiles) return;
    const bool split = q_per_part > 0;                                     // split walk over the block columns (see the scalar kernel)
    int64_t j_lo = 0, j_hi = ncol;
    if (split) {
        j_lo = (int64_t)blockIdx.y * q_per_part;
        j_hi = j_lo + q_per_part < ncol ? j_lo + q_per_part : ncol;
        d = slabs + (int64_t)blockIdx.y * slab_stride;
    }
    const int64_t ns = (row_off[i + 1] - row_off[i]) * E;                 // scalars in this block row
    for (int64_t s = (tile * 256 + threadIdx.x) * NS; s < ns; s += (int64_t)ntiles * 256 * NS) {
        V acc = (V)(S)0;
        bool touched = split;
        if (ncol > 1 && !split) acc = ld<false>(reinterpret_cast<const V *>(d + row_off[i] * E + s));
        // the block table and the column offsets are read ONE GROUP AHEAD (scalar loads): a block's vector loads need them, and
        // waiting for them block by block serialises two latencies per block (the mixed one-pass step lost 12 % to that)
        jh_dev_block nb[GENERAL_Q];
        int64_t noff[GENERAL_Q];
#pragma unroll
        for (int q = 0; q < GENERAL_Q; q++)
            if (j_lo + q < j_hi) { nb[q] = blocks[i + (j_lo + q) * nrow]; noff[q] = col_off[j_lo + q]; }
        for (int64_t j0 = j_lo; j0 < j_hi; j0 += GENERAL_Q) {              // (1020), GENERAL_Q columns' loads in flight
            jh_dev_block b[GENERAL_Q];
            int64_t off[GENERAL_Q];
            V x[GENERAL_Q], c[GENERAL_Q];
            bool on[GENERAL_Q];
#pragma unroll
            for (int q = 0; q < GENERAL_Q; q++) {
                b[q] = nb[q];
                off[q] = noff[q];
                const int64_t jn = j0 + GENERAL_Q + q;
                if (jn < j_hi) { nb[q] = blocks[i + jn * nrow]; noff[q] = col_off[jn]; }
            }
#pragma unroll
            for (int q = 0; q < GENERAL_Q; q++) {
                const int64_t j = j0 + q;
                on[q] = j < j_hi;
                x[q] = (V)(S)0;
                c[q] = (V)(S)0;
                if (on[q]) {
                    if (b[q].kind == JH_OP_ZERO) on[q] = (fmode != 0);     // (1022) skipped; f! keeps it as +0 -- and never loads for it
                    else {
                        x[q] = ld<false>(reinterpret_cast<const V *>(m + off[q] * E + s));
                        if (block_reads_coeff(b[q], fmode != 0)) c[q] = ld<true>(reinterpret_cast<const V *>((const S *)b[q].coeff + s));   // streamed once
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < GENERAL_Q; q++)
                if (on[q]) {
                    const V p = apply_block_loaded<S, E, NS, V>(b[q], x[q], c[q], false, fmode != 0);   // mul!(dtmp, op, _m)
                    acc = (ncol > 1) ? acc + p : p;                        // (1024) / (1026), columns in order
                    touched = true;
                }
        }
        if (touched) {
            if (split) st<false>(reinterpret_cast<V *>(d + row_off[i] * E + s), acc);            // a slab: the fold reads it back
            else st<true>(reinterpret_cast<V *>(d + row_off[i] * E + s), acc);                   // the result row: written once
        }
    }
}

template <typename S, int E, int NS>
__global__ void k_block_adj_general_vec(const jh_dev_block *__restrict__ blocks, int64_t nrow, int64_t ncol,
                                        const int64_t *__restrict__ row_off, const int64_t *__restrict__ col_off,
                                        S *__restrict__ m, const S *__restrict__ d, unsigned ntiles,
                                        int64_t q_per_part, S *__restrict__ slabs, int64_t slab_stride, int nt_out)
{
    typedef typename vec_of<S, NS>::type V;
    int64_t j, tile;                                                       // block column, tile
    general_line_tile((unsigned)ncol, ntiles, j, tile);
    ntiles &= 0x0fffffffu;
    if (tile >= ntiles) return;
    int64_t i_lo = 0, i_hi = nrow;                                         // q_per_part > 0: split walk over the block rows
    if (q_per_part > 0) {
        i_lo = (int64_t)blockIdx.y * q_per_part;
        i_hi = i_lo + q_per_part < nrow ? i_lo + q_per_part : nrow;
        m = slabs + (int64_t)blockIdx.y * slab_stride;
    }
    const int64_t ns = (col_off[j + 1] - col_off[j]) * E;
    for (int64_t s = (tile * 256 + threadIdx.x) * NS; s < ns; s += (int64_t)ntiles * 256 * NS) {
        V acc = (V)(S)0;
        bool touched = (nrow > 1);
        jh_dev_block nb[GENERAL_Q];                                        // block table and row offsets one group ahead (see the forward)
        int64_t noff[GENERAL_Q];
#pragma unroll
        for (int q = 0; q < GENERAL_Q; q++)
            if (i_lo + q < i_hi) { nb[q] = blocks[(i_lo + q) + j * nrow]; noff[q] = row_off[i_lo + q]; }
        for (int64_t i0 = i_lo; i0 < i_hi; i0 += GENERAL_Q) {              // (1045), GENERAL_Q rows' loads in flight
            jh_dev_block b[GENERAL_Q];
            int64_t off[GENERAL_Q];
            V x[GENERAL_Q], c[GENERAL_Q];
            bool on[GENERAL_Q];
#pragma unroll
            for (int q = 0; q < GENERAL_Q; q++) {
                b[q] = nb[q];
                off[q] = noff[q];
                const int64_t in = i0 + GENERAL_Q + q;
                if (in < i_hi) { nb[q] = blocks[in + j * nrow]; noff[q] = row_off[in]; }
            }
#pragma unroll
            for (int q = 0; q < GENERAL_Q; q++) {
                const int64_t i = i0 + q;
                on[q] = i < i_hi;
                x[q] = (V)(S)0;
                c[q] = (V)(S)0;
                if (on[q]) {
                    if (b[q].kind == JH_OP_ZERO) on[q] = false;            // (1047)
                    else {
                        x[q] = ld<false>(reinterpret_cast<const V *>(d + off[q] * E + s));
                        if (block_reads_coeff(b[q], false)) c[q] = ld<true>(reinterpret_cast<const V *>((const S *)b[q].coeff + s));   // streamed once
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < GENERAL_Q; q++)
                if (on[q]) {
                    const V p = apply_block_loaded<S, E, NS, V>(b[q], x[q], c[q], true, false);         // mul!(mtmp, op', _d)
                    acc = (nrow > 1) ? acc + p : p;                        // (1049) / (1051), rows in order
                    touched = true;
                }
        }
        if (touched) {
            if (nt_out) st<true>(reinterpret_cast<V *>(m + col_off[j] * E + s), acc);   // a large result written once (a wide operator's adjoint)
            else st<false>(reinterpret_cast<V *>(m + col_off[j] * E + s), acc);         // a slab of the split walk / a small domain vector: read again soon
        }
    }
}

// ---- M x K grids whose blocks are ALL plain diagonals (>= 2 x 2, one block length, everything 16-byte aligned) --------------
// The general kernels above decide per block what to do (kind switch, zero-block skip); those branches make the compiler wait for
// ALL outstanding loads at every join, so more than one block's loads in flight per lane buys nothing there (GENERAL_Q).  A grid
// of diagonals needs no decision: this kernel issues the loads of Q blocks of a line back to back -- coefficient pointers one
// group ahead, like the general kernels -- and combines them in the reference's order, product rounded then added
// (forward: d_i = d_i as found + a_i1 .* m_1 + a_i2 .* m_2 + ..., 1020-1024; adjoint: m_j = 0 + conj(a_1j) .* d_1 + ..., 1042-1049).
// Same (line, tile) decode as the general kernels: the workgroups of one tile of every line run together, so the shared input
// tile comes from HBM once.  TRANSPOSED = false: line = block row; true: line = block column.
template <typename S, int E, int NS, int Q, bool TRANSPOSED, int U = 1>
__global__ __launch_bounds__(256) void k_grid_diag(const jh_dev_block *__restrict__ blocks, int64_t nrow, int64_t ncol, int64_t n_scalars,
                                                   const S *__restrict__ in, S *__restrict__ out, unsigned ntiles)
{
    typedef typename vec_of<S, NS>::type V;
    int64_t line, tile;
    general_line_tile((unsigned)(TRANSPOSED ? ncol : nrow), ntiles, line, tile);
    ntiles &= 0x0fffffffu;
    if (tile >= ntiles) return;
    const int64_t nsum = TRANSPOSED ? nrow : ncol;                          // blocks walked per line
    const int64_t step = TRANSPOSED ? 1 : nrow, first = TRANSPOSED ? line * nrow : line;   // block (q) of the line = blocks[first + q * step]
    S *o = out + line * n_scalars;
    for (int64_t s0 = (tile * 256 * U + threadIdx.x) * NS; s0 < n_scalars; s0 += (int64_t)ntiles * 256 * U * NS) {
        int64_t s[U];
        bool ok[U];
        V acc[U];
#pragma unroll
        for (int u = 0; u < U; u++) {                                       // U packs per lane, 256 lanes apart (clamped: branch-free loads)
            ok[u] = s0 + (int64_t)u * 256 * NS < n_scalars;
            s[u] = ok[u] ? s0 + (int64_t)u * 256 * NS : s0;
            acc[u] = TRANSPOSED ? (V)(S)0 : ld<false>(reinterpret_cast<const V *>(o + s[u]));   // `_m .= 0` (1042) / d as found (1024)
        }
        const S *na[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) na[q] = (const S *)blocks[first + (q < nsum ? q : 0) * step].coeff;
        int64_t q0 = 0;
        for (; q0 + Q <= nsum; q0 += Q) {
            const S *a[Q];
            V x[Q][U], c[Q][U];
#pragma unroll
            for (int q = 0; q < Q; q++) {
                a[q] = na[q];
                const int64_t qn = q0 + Q + q;
                na[q] = (const S *)blocks[first + (qn < nsum ? qn : 0) * step].coeff;
            }
#pragma unroll
            for (int q = 0; q < Q; q++)
#pragma unroll
                for (int u = 0; u < U; u++) {
                    c[q][u] = ld<true>(reinterpret_cast<const V *>(a[q] + s[u]));                       // streamed once
                    x[q][u] = ld<false>(reinterpret_cast<const V *>(in + (q0 + q) * n_scalars + s[u]));  // shared by every line: through the caches
                }
#pragma unroll
            for (int q = 0; q < Q; q++)
#pragma unroll
                for (int u = 0; u < U; u++) acc[u] = acc[u] + vmul<S, E, NS, V>(c[q][u], x[q][u], TRANSPOSED);
        }
        for (int64_t q = q0; q < nsum; q++) {
            const S *aq = (const S *)blocks[first + q * step].coeff;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const V c = ld<true>(reinterpret_cast<const V *>(aq + s[u]));
                const V x = ld<false>(reinterpret_cast<const V *>(in + q * n_scalars + s[u]));
                acc[u] = acc[u] + vmul<S, E, NS, V>(c, x, TRANSPOSED);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (ok[u]) st<true>(reinterpret_cast<V *>(o + s[u]), acc[u]);
    }
}

// ---- the same grids, REGISTER-TILED (round 3) -----------------------------------------------------------------------------
// k_grid_diag gives every (line, tile) its own workgroup, so a workgroup issues TWO loads per product -- its coefficient pack
// (from HBM) and the input pack every other line reads too (from L2) -- and runs at 59-65 % of the HBM roofline on big grids
// although its HBM traffic is exactly the unique bytes.  Here a workgroup owns R LINES x one element tile: R accumulators stay
// in registers, and for every summed block index q the input pack is loaded ONCE and used for the R lines, the R coefficient
// packs next to it -- (R + 1) loads for R products, QQ such steps' loads issued back to back before any arithmetic, U packs per lane.  Every accumulator still adds
// its products in the reference's order, q = 0, 1, 2, ..., each product rounded before its add (forward 1020-1024: d_i as found
// + a_i1 .* m_1 + a_i2 .* m_2 + ...; adjoint 1042-1049: 0 + conj(a_1j) .* d_1 + ...) => the bits of k_grid_diag and of the oracle.
// Line GROUPS take the place of lines in the XCD-aware decode: the groups that read one input tile are dispatched together on one
// XCD, so that tile still comes from HBM once.  Lines beyond the last group are clamped to the last line (branch-free loads of
// valid memory) and not stored.
template <typename S, int E, int NS, int R, int QQ, int U, bool TRANSPOSED>
__global__ __launch_bounds__(256) void k_grid_tile(const jh_dev_block *__restrict__ blocks, int64_t nrow, int64_t ncol, int64_t n_scalars,
                                                   const S *__restrict__ in, S *__restrict__ out, unsigned ntiles, unsigned ngroups)
{
    typedef typename vec_of<S, NS>::type V;
    int64_t grp, tile;
    general_line_tile(ngroups, ntiles, grp, tile);
    ntiles &= 0x0fffffffu;
    if (tile >= ntiles) return;
    const int64_t nlines = TRANSPOSED ? ncol : nrow, nsum = TRANSPOSED ? nrow : ncol;
    const int64_t qstep = TRANSPOSED ? 1 : nrow, lstep = TRANSPOSED ? nrow : 1;   // block (line l, q) = blocks[l * lstep + q * qstep]
    int64_t line[R];
#pragma unroll
    for (int r = 0; r < R; r++) line[r] = grp * R + r < nlines ? grp * R + r : nlines - 1;
    for (int64_t s0 = (tile * 256 * U + threadIdx.x) * NS; s0 < n_scalars; s0 += (int64_t)ntiles * 256 * U * NS) {
        int64_t s[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; u++) {                                       // U packs per lane, 256 lanes apart (clamped: branch-free loads)
            ok[u] = s0 + (int64_t)u * 256 * NS < n_scalars;
            s[u] = ok[u] ? s0 + (int64_t)u * 256 * NS : s0;
        }
        V acc[R][U];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int u = 0; u < U; u++)
                acc[r][u] = TRANSPOSED ? (V)(S)0 : ld<true>(reinterpret_cast<const V *>(out + line[r] * n_scalars + s[u]));   // `_m .= 0` (1042) / d as found (1024)
        const S *na[QQ][R];                                                   // coefficient pointers, one group of QQ steps ahead
#pragma unroll
        for (int q = 0; q < QQ; q++)
#pragma unroll
            for (int r = 0; r < R; r++) na[q][r] = (const S *)blocks[line[r] * lstep + (q < nsum ? q : 0) * qstep].coeff;
        int64_t q0 = 0;
        for (; q0 + QQ <= nsum; q0 += QQ) {
            const S *a[QQ][R];
#pragma unroll
            for (int q = 0; q < QQ; q++) {
                const int64_t qn = q0 + QQ + q;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    a[q][r] = na[q][r];
                    na[q][r] = (const S *)blocks[line[r] * lstep + (qn < nsum ? qn : 0) * qstep].coeff;
                }
            }
            V x[QQ][U], c[QQ][R][U];
#pragma unroll
            for (int q = 0; q < QQ; q++)
#pragma unroll
                for (int u = 0; u < U; u++) {
                    x[q][u] = ld<false>(reinterpret_cast<const V *>(in + (q0 + q) * n_scalars + s[u]));          // shared by every line group: through the caches
#pragma unroll
                    for (int r = 0; r < R; r++) c[q][r][u] = ld<true>(reinterpret_cast<const V *>(a[q][r] + s[u]));   // streamed once
                }
#pragma unroll
            for (int q = 0; q < QQ; q++)
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int u = 0; u < U; u++) acc[r][u] = acc[r][u] + vmul<S, E, NS, V>(c[q][r][u], x[q][u], TRANSPOSED);
        }
        for (int64_t q = q0; q < nsum; q++) {
            const S *aq[R];
#pragma unroll
            for (int r = 0; r < R; r++) aq[r] = (const S *)blocks[line[r] * lstep + q * qstep].coeff;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const V x = ld<false>(reinterpret_cast<const V *>(in + q * n_scalars + s[u]));
                V c[R];
#pragma unroll
                for (int r = 0; r < R; r++) c[r] = ld<true>(reinterpret_cast<const V *>(aq[r] + s[u]));
#pragma unroll
                for (int r = 0; r < R; r++) acc[r][u] = acc[r][u] + vmul<S, E, NS, V>(c[r], x, TRANSPOSED);
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int u = 0; u < U; u++)
                if (ok[u] && grp * R + r < nlines) st<true>(reinterpret_cast<V *>(out + line[r] * n_scalars + s[u]), acc[r][u]);
    }
}

// ---- general M x K grids of EQUAL blocks of any elementwise kinds, register-tiled (round 3) ---------------------------------------
// The same idea as k_grid_tile for grids that are not all plain diagonals (zero blocks, identity / scalar blocks, adjointed
// diagonals, SQUARE Jacobians): a workgroup owns TWO lines x one element tile; per summed block index the input pack is loaded once
// for both lines, the coefficient packs of the blocks that have one next to it, two steps' loads issued back to back.  The LOAD
// section is branch-free -- a block without a coefficient array (or a zero block, which contributes nothing: 1022 / 1047) loads the
// input pack's address again, an L1 hit -- so that the compiler does not drain the outstanding loads at every kind switch, which
// is what holds k_block_*_general_vec to one block in flight (GENERAL_Q); the kind switches come afterwards, wave-uniform.
// Each accumulator adds its non-zero blocks' terms in the reference's order, product rounded before the add: the bits of the general
// kernels.  A block row of zero blocks only is left as found (forward, 1022); the adjoint of a grid (nrow > 1) always writes (1042).
template <typename S, int E, int NS, int QQ, int U, bool TRANSPOSED, int R = 2>
__global__ __launch_bounds__(256) void k_general_tile(const jh_dev_block *__restrict__ blocks, int64_t nrow, int64_t ncol, int64_t n_scalars,
                                                      const S *__restrict__ in, S *__restrict__ out, unsigned ntiles, unsigned ngroups)
{
    typedef typename vec_of<S, NS>::type V;
    int64_t grp, tile;
    general_line_tile(ngroups, ntiles, grp, tile);
    ntiles &= 0x0fffffffu;
    if (tile >= ntiles) return;
    const int64_t nlines = TRANSPOSED ? ncol : nrow, nsum = TRANSPOSED ? nrow : ncol;
    const int64_t qstep = TRANSPOSED ? 1 : nrow, lstep = TRANSPOSED ? nrow : 1;   // block (line l, q) = blocks[l * lstep + q * qstep]
    int64_t line[R];
#pragma unroll
    for (int r = 0; r < R; r++) line[r] = grp * R + r < nlines ? grp * R + r : nlines - 1;
    for (int64_t s0 = (tile * 256 * U + threadIdx.x) * NS; s0 < n_scalars; s0 += (int64_t)ntiles * 256 * U * NS) {
        int64_t s[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; u++) {                                       // U packs per lane, 256 lanes apart (clamped: branch-free loads)
            ok[u] = s0 + (int64_t)u * 256 * NS < n_scalars;
            s[u] = ok[u] ? s0 + (int64_t)u * 256 * NS : s0;
        }
        V acc[R][U];
        bool touched[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
#pragma unroll
            for (int u = 0; u < U; u++)
                acc[r][u] = TRANSPOSED ? (V)(S)0 : ld<true>(reinterpret_cast<const V *>(out + line[r] * n_scalars + s[u]));   // `_m .= 0` (1042) / d as found (1024)
            touched[r] = TRANSPOSED;
        }
        jh_dev_block nb[QQ][R];                                               // block table entries one group of steps ahead
#pragma unroll
        for (int q = 0; q < QQ; q++)
#pragma unroll
            for (int r = 0; r < R; r++) nb[q][r] = blocks[line[r] * lstep + (q < nsum ? q : 0) * qstep];
        for (int64_t q0 = 0; q0 < nsum; q0 += QQ) {
            jh_dev_block b[QQ][R];
#pragma unroll
            for (int q = 0; q < QQ; q++) {
                const int64_t qn = q0 + QQ + q;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    b[q][r] = nb[q][r];
                    nb[q][r] = blocks[line[r] * lstep + (qn < nsum ? qn : 0) * qstep];
                }
            }
            V x[QQ][U], c[QQ][R][U];
#pragma unroll
            for (int q = 0; q < QQ; q++) {
                const S *xb = in + (q0 + q < nsum ? q0 + q : 0) * n_scalars;                              // (a step beyond the end re-reads block 0: unused)
#pragma unroll
                for (int u = 0; u < U; u++) x[q][u] = ld<false>(reinterpret_cast<const V *>(xb + s[u]));  // shared by every line group: through the caches
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const bool has = block_reads_coeff(b[q][r], false);
                    const S *cb = has ? (const S *)b[q][r].coeff : xb;                                    // no coefficient array: the input pack again (L1)
#pragma unroll
                    for (int u = 0; u < U; u++)
                        c[q][r][u] = has ? ld<true>(reinterpret_cast<const V *>(cb + s[u])) : ld<false>(reinterpret_cast<const V *>(cb + s[u]));
                }
            }
#pragma unroll
            for (int q = 0; q < QQ; q++)
                if (q0 + q < nsum) {
#pragma unroll
                    for (int r = 0; r < R; r++)
                        if (b[q][r].kind != JH_OP_ZERO) {                                                 // (1022) / (1047): skipped
#pragma unroll
                            for (int u = 0; u < U; u++)
                                acc[r][u] = acc[r][u] + apply_block_loaded<S, E, NS, V>(b[q][r], x[q][u], c[q][r][u], TRANSPOSED, false);
                            touched[r] = true;
                        }
                }
        }
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int u = 0; u < U; u++)
                if (touched[r] && ok[u] && grp * R + r < nlines) st<true>(reinterpret_cast<V *>(out + line[r] * n_scalars + s[u]), acc[r][u]);
    }
}

// is `op` such a grid?  (every block an un-adjointed diagonal -- for a real element type the adjoint flag is immaterial and
// all_diag already says so --, >= 2 x 2, aligned)
bool grid_diag_ok(const jh_blockop *op, const void *rng_ptr, const void *dom_ptr)
{
    if (!(op->all_diag && op->nrow >= 2 && op->ncol >= 2)) return false;
    const int64_t n = op->row_len[0];
    if (n == 0 || (n * (int64_t)jh_dtype_size(op->dtype)) % 16 != 0) return false;
    if ((((uintptr_t)rng_ptr) | ((uintptr_t)dom_ptr)) & 15u) return false;
    for (const auto &b : op->blocks)
        if (((uintptr_t)b.coeff) & 15u) return false;
    return true;
}

// second stage of the general kernels' split walk: out[line] = (add_found ? out as found : 0) + slab 0 + slab 1 + ... for every
// line (block row of the range / block column of the domain) that the operator touches; 64 scalar lanes x 4 part lanes per
// workgroup, fp64 accumulation, fixed order => deterministic (tolerance parity with the single ordered sum)
template <typename S>
__global__ __launch_bounds__(256) void k_fold_general(const S *__restrict__ slabs, int64_t slab_stride, int nparts, S *__restrict__ out,
                                                      const int64_t *__restrict__ off, int E, const unsigned char *__restrict__ touched,
                                                      int add_found)
{
    __shared__ double sm[4][64];
    const int64_t line = blockIdx.y;
    if (touched && !touched[line]) return;                                 // a block row of zero blocks only: d stays as found (1022)
    const int64_t base = off[line] * E, ns = (off[line + 1] - off[line]) * E;
    const int v = threadIdx.x & 63, q = threadIdx.x >> 6;
    for (int64_t s0 = (int64_t)blockIdx.x * 64; s0 < ns; s0 += (int64_t)gridDim.x * 64) {
        const int64_t s = s0 + v;
        const bool ok = s < ns;
        double acc = 0.0;
        if (ok) {
#pragma unroll 4
            for (int p = q; p < nparts; p += 4) acc += (double)slabs[(int64_t)p * slab_stride + base + s];
        }
        sm[q][v] = acc;
        __syncthreads();
        if (q == 0 && ok) {
            double t = add_found ? (double)out[base + s] : 0.0;
            t += acc;
            t += sm[1][v];
            t += sm[2][v];
            t += sm[3][v];
            out[base + s] = (S)t;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ launch helpers ------------
// Kernel shapes, fitted to interleaved sweeps on MI355X (profiles/sweep_r01*.txt, profiles/repeat_r01.txt; Float32):
//   forward  1024 x 256^3 (128 GiB): sequential row sweep, 1024 threads x 8 vectors x 16 rows: 6.04 TB/s in
//            every process.  Walking all rows concurrently (order 1) reaches 6.4-6.5 TB/s in some processes
//            and 5.2-5.6 TB/s in others (same binary, same box: physical placement luck), banded walks sit
//            in between -- so the sequential sweep is the default and the other orders stay behind the knob.
//            256 x 256^3 ... 16 x 256^3: 256 threads x 4 vectors x 4 rows (5.8-6.3 TB/s)
//            64 x 128^3, 1024 x 64^3 (1-2 GiB): 256 threads x 1 vector x 2 rows (6.3-6.4 TB/s)
//   adjoint  wants FEW, FAT workgroups: 4 vectors per thread as long as >= 256 workgroups remain
//            (1024 x 256^3: 6.6-6.7 TB/s; 64 x 128^3: 6.8-7.1 TB/s; 1024 x 64^3: 1 vector, 7.1 TB/s)
//   fused A'A reads one stream, so it keeps twice the rows in flight.
struct TallShape { int wg, unroll, aux, order, ctiles = 0; };   // aux = rows per workgroup (forward) / rows in flight (adjoint); ctiles: forward column bands (tiles per band, 0: none)

TallShape pick_fwd_shape(int64_t nvec, int64_t nrow, size_t vec_bytes)
{
    jh_context &c = jh_ctx();
    (void)vec_bytes;
    TallShape s;
    if (nvec >= ((int64_t)1 << 21) && nrow >= 512) s = TallShape{1024, 8, 16, 0};
    else if (nvec >= ((int64_t)1 << 21)) s = TallShape{256, 1, 1, 1, 32};   // blocks of >= 32 MiB, fewer than 512 rows: column bands (16-32 x 256^3: 6.05-6.19 -> 6.67 TB/s with
                                                                            // 256 x 4 x 4 rows sequential before; profiles/exp_r04_small_fwd.txt)
    else s = TallShape{256, 1, 2, 0};
    if (c.fwd_wg) s.wg = (int)c.fwd_wg;
    if (c.fwd_unroll) s.unroll = (int)c.fwd_unroll;
    if (c.fwd_group) s.aux = (int)c.fwd_group;
    if (c.fwd_order >= 0) s.order = (int)c.fwd_order;
    if (c.fwd_ctiles >= 0) s.ctiles = (int)c.fwd_ctiles;
    return s;
}

TallShape pick_adj_shape(int64_t nvec, int64_t nrow, int mode)
{
    jh_context &c = jh_ctx();
    TallShape s{256, 1, 4, 0};
    if (nvec >= 4 * 256 * 256) s.unroll = 4;
    else if (nvec >= 2 * 256 * 256) s.unroll = 2;
    if (s.unroll == 4) s.aux = (nrow >= 256) ? 4 : 2;
    if (nvec >= ((int64_t)1 << 22)) s.wg = 512;
    if (mode == 0 && nvec >= ((int64_t)1 << 22) && nrow < 512) { s.wg = 1024; s.aux = 2; }   // 128..256 x 256^3: +2..7 % (sweep_r01_pair_*)
    if (mode == 1) {                                   // fused normal operator: one input stream
        s.aux = (s.unroll == 4) ? 4 : 8;
        if (nvec >= ((int64_t)1 << 22)) s.wg = 1024;
    }
    if (c.adj_wg) s.wg = (int)c.adj_wg;
    if (c.adj_unroll) s.unroll = (int)c.adj_unroll;
    if (c.adj_depth) s.aux = (int)c.adj_depth;
    if (s.unroll == 4 && s.aux == 8) s.aux = 4;        // 4 x 8 is not instantiated (register budget)
    // a 1024-thread workgroup has 128 VGPRs per lane: the two-stream adjoint keeps at most 8 packs per stream in flight there
    // (2 x 8 and 4 x 4 spilled 108-176 bytes per lane to scratch; same bits with fewer rows in flight)
    if (mode == 0 && s.wg == 1024 && s.unroll * s.aux > 8) s.aux = 8 / s.unroll;
    return s;
}

// Split-row walk of the adjoint-shaped kernels.  The ordered walk gives one thread a 16-byte vector of the DOMAIN and
// all rows: with n elements per block that is n/4 threads, so a tall operator of many SMALL blocks (seismic traces
// rather than volumes) leaves most of the chip idle -- 1 GiB of 4096-element Float32 rows: 10.7 ms, 200 GB/s
// (profiles/exp_r01_small_blocks.txt).  When the ordered walk would launch fewer workgroups than the chip has CUs, the
// rows are cut into `parts` contiguous ranges, workgroup row y sums range y in order into its own slab, and k_fold_parts
// adds the slabs in a fixed order.  Deterministic, but not the bits of the single ordered sum (tolerance parity, like the
// multi-GPU sum).  Knob adj_split: -1 automatic, 0 never (always the ordered, bit-exact walk), k > 1 that many parts.
int64_t pick_adj_parts(int64_t gx, int64_t nrow)
{
    jh_context &c = jh_ctx();
    if (c.adj_split == 0 || nrow < 4) return 1;
    int64_t parts;
    if (c.adj_split > 0) parts = c.adj_split;
    else {
        if (gx >= c.cu_count || nrow < 256) return 1;                     // small operators keep the ordered, bit-exact walk
        parts = (8 * (int64_t)c.cu_count + gx - 1) / gx;                 // ~8 workgroups per CU
        if (parts > nrow / 16) parts = nrow / 16;                       // at least 16 rows per part
    }
    if (parts > nrow / 2) parts = nrow / 2;
    if (parts > 65535) parts = 65535;                                    // gridDim.y
    return parts < 2 ? 1 : parts;
}

template <typename S, int NS>
int launch_fold_parts(const void *parts, int64_t part_stride, int64_t nparts, void *out, int64_t s_begin, int64_t s_end)
{
    jh_context &c = jh_ctx();
    const int64_t gx = ((s_end - s_begin) / NS + 63) / 64;
    hipLaunchKernelGGL((k_fold_parts<S, NS>), dim3((unsigned)gx), dim3(1024), 0, c.stream, (const S *)parts, part_stride, (int)nparts,
                       (S *)out, s_begin, s_end);
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

template <typename S, int E, int NS, bool NT, int BLK>
int launch_tall_fwd_u(const jh_blockop *op, void *d, const void *m, int64_t n_scalars, const TallShape &sh)
{
    jh_context &c = jh_ctx();
    const S *a_base = op->diag_strided ? (const S *)op->blocks[0].coeff : nullptr;
    const int64_t a_stride = op->diag_stride_elems * E;
    int64_t G = sh.aux;
    if (G > op->nrow) G = op->nrow;
    int64_t gy = (op->nrow + G - 1) / G;
    {   // HIP: grid x block must stay below 2^32 threads
        const int64_t gx0 = (n_scalars / NS + (int64_t)sh.unroll * BLK - 1) / ((int64_t)sh.unroll * BLK);
        while (gx0 * gy * BLK >= ((int64_t)1 << 32) && G < op->nrow) { G *= 2; gy = (op->nrow + G - 1) / G; }
    }
    int64_t band = sh.order <= 0 ? 1 : (sh.order == 1 ? gy : sh.order);   // order: 0 sequential, 1 all rows, k>1 = k groups per band
    if (band > gy) band = gy;
    c.last_fwd_walk = sh.ctiles ? 2 : sh.order;
    c.last_fwd_rows_per_wg = G;
#define JH_FWD_CASE(U)                                                                                               \
    case U: {                                                                                                         \
        int64_t gx = (n_scalars + (int64_t)U * BLK * NS - 1) / ((int64_t)U * BLK * NS);                               \
        JH_REQUIRE(gx * gy * BLK < (int64_t)1 << 32, "tall forward: grid of %lld workgroups is too large", (long long)(gx * gy)); \
        hipLaunchKernelGGL((k_tall_diag_fwd<S, E, NS, U, NT, BLK>), dim3((unsigned)(gx * gy)), dim3(BLK), 0, c.stream, \
                           op->dev_blocks, op->nrow, (int)G, a_base, a_stride, (const S *)m, (S *)d, n_scalars,      \
                           (unsigned)gx, (unsigned)gy, (unsigned)band, (unsigned)sh.ctiles);                       \
    } break;
    switch (sh.unroll) {
        JH_FWD_CASE(1)
        JH_FWD_CASE(2)
        JH_FWD_CASE(4)
        JH_FWD_CASE(8)
    default: return jh_fail(JH_ERR_INVALID, "fwd_unroll %d unsupported", sh.unroll);
    }
#undef JH_FWD_CASE
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

template <typename S, int E, int NS, bool NT, int MODE, int BLK>
int launch_tall_adj_u(const jh_blockop *op, void *out, const void *in, int64_t n_scalars, const TallShape &sh, int64_t s_begin,
                      int64_t s_end)
{
    jh_context &c = jh_ctx();
    const S *a_base = op->diag_strided ? (const S *)op->blocks[0].coeff : nullptr;
    const int64_t a_stride = op->diag_stride_elems * E;
    const int direct = (op->nrow == 1 && MODE == 0) ? 1 : 0;
    // A 128 GiB walk runs 2-3 % faster as two launches over 512 rows each than as one (profiles/exp_r01_adj_row_chunks.txt,
    // exp_r01_adj_by_rows.txt); the second launch continues the ordered sum, so the bits do not change.
    int64_t rows_per_launch = op->nrow;
    if (c.adj_rows_per_launch > 0) rows_per_launch = c.adj_rows_per_launch < op->nrow ? c.adj_rows_per_launch : op->nrow;
    else if (op->nrow >= 768 && (double)op->nrow * (double)n_scalars * sizeof(S) >= 48.0 * (double)(1ull << 30)) rows_per_launch = 512;
    c.last_adj_launches = (op->nrow + rows_per_launch - 1) / rows_per_launch;
    // many rows of small blocks: split-row walk (pick_adj_parts) -- one launch over (tiles, parts), then the fold
    const int64_t gx0 = (s_end - s_begin + (int64_t)sh.unroll * BLK * NS - 1) / ((int64_t)sh.unroll * BLK * NS);
    const int from_found = (MODE == 0) ? c.adj_from_found : 0;              // continue from what `out` holds (a wide operator's forward)
    int64_t parts = (direct || from_found) ? 1 : pick_adj_parts(gx0, op->nrow);
    int64_t rows_per_part = 0;
    const int64_t part_stride = s_end - s_begin;
    void *slabs = nullptr;
    if (parts > 1) {
        rows_per_part = (op->nrow + parts - 1) / parts;
        parts = (op->nrow + rows_per_part - 1) / rows_per_part;                // no empty part
        JH_TRY(jh_ensure_scratch((size_t)parts * (size_t)part_stride * sizeof(S), &slabs));
        rows_per_launch = op->nrow;
        c.last_adj_launches = 1;
    }
    c.last_adj_parts = parts;
#define JH_ADJ_CASE(U, DEPTH)                                                                                          \
    if constexpr (!(BLK == 1024 && MODE == 0 && U * DEPTH > 8))                                                        \
    if (sh.unroll == U && sh.aux == DEPTH) {                                                                           \
        int64_t gx = (s_end - s_begin + (int64_t)U * BLK * NS - 1) / ((int64_t)U * BLK * NS);                          \
        for (int64_t r0 = 0; r0 < op->nrow; r0 += rows_per_launch) {                                                   \
            const int64_t r1 = r0 + rows_per_launch < op->nrow ? r0 + rows_per_launch : op->nrow;                        \
            hipLaunchKernelGGL((k_tall_diag_adj<S, E, NS, U, DEPTH, NT, MODE, BLK>), dim3((unsigned)gx, (unsigned)parts), \
                               dim3(BLK), 0,                                                                           \
                               c.stream, op->dev_blocks, op->nrow, a_base, a_stride, (S *)out, (const S *)in, n_scalars,   \
                               direct, s_begin, s_end, r0, r1, (r0 > 0 || from_found) ? 1 : 0, rows_per_part, (S *)slabs, part_stride); \
            JH_CHECK_HIP(hipGetLastError());                                                                           \
        }                                                                                                              \
        if (parts > 1) return launch_fold_parts<S, NS>(slabs, part_stride, parts, out, s_begin, s_end);                \
        return JH_OK;                                                                                                  \
    }
    JH_ADJ_CASE(1, 1) JH_ADJ_CASE(1, 2) JH_ADJ_CASE(1, 4) JH_ADJ_CASE(1, 8)
    JH_ADJ_CASE(2, 1) JH_ADJ_CASE(2, 2) JH_ADJ_CASE(2, 4) JH_ADJ_CASE(2, 8)
    JH_ADJ_CASE(4, 1) JH_ADJ_CASE(4, 2) JH_ADJ_CASE(4, 4)
#undef JH_ADJ_CASE
    return jh_fail(JH_ERR_INVALID, "adj_unroll %d x adj_depth %d unsupported", sh.unroll, sh.aux);
}

template <typename S, int E, int NS>
int launch_tall_fwd_shape(const jh_blockop *op, void *d, const void *m, int64_t n_scalars, const TallShape &sh);

// the shapes the first forward of a large operator is timed with: which one wins differs from process to process
// (profiles/repeat_r01*.txt, sweep_r01_order.txt): sequential sweeps, and walks that touch every row group concurrently
// -- and the column-persistent walk (rows per workgroup = all rows: a workgroup keeps its m tile and streams every block row
// through it, no m re-reads), which is the best of the placement-independent shapes (22.0 vs 23.0 ms for the 16-row sweep,
// profiles/sweep_r01_fwd_persistent_1024x256.txt)
// Candidates 6 and 7 (late round 2) are ONE block row per workgroup with all rows concurrent -- workgroups that are born, move
// one tile of one row and die: the fastest shapes at the row counts a rank owns on 2 and 4 GPUs (512 rows: 6.32 TB/s against
// 5.98 for the best of the first six, 256 rows: 6.2 against 6.06; profiles/sweep_r02_fwd_rows.txt), equal to the others at 1024.
// Candidates 8 and 9 (late round 4): one block row per workgroup in COLUMN bands of 32 / 64 tiles (k_tall_diag_fwd's ctiles decode) -- tried by
// operators of fewer than 1024 rows only (the row counts a rank owns on 2 / 4 / 8 GPUs): 128 rows 5.8 -> 6.4 TB/s, 256 rows 5.8 -> 6.1, 512 rows
// 6.05 -> 6.4 where a copy between the same slabs runs at 6.5; at 1024 rows the row-concurrent walk equals the copy and the bands lose 2 %.
constexpr int K_FWD_CANDIDATES = 10, K_FWD_CANDIDATES_TALL = 8;          // (operators of >= 1024 rows of blocks >= 64 MiB try the first eight)
static_assert(2 * K_FWD_CANDIDATES + 4 <= jh_blockop::LazyTune::SLOTS, "two passes per candidate and the play-off must fit the trial slots");
static_assert(K_FWD_CANDIDATES <= jh_blockop::LazyTune::MAXC, "the candidates' records");
const TallShape k_fwd_candidates[K_FWD_CANDIDATES] = {TallShape{1024, 8, 16, 0}, TallShape{512, 1, 2, 1}, TallShape{256, 4, 4, 0},
                                                      TallShape{256, 4, 16, 1}, TallShape{512, 4, 8, 1}, TallShape{1024, 8, 1 << 20, 0},
                                                      TallShape{512, 8, 1, 1},  TallShape{256, 1, 1, 1},
                                                      TallShape{256, 1, 1, 1, 32}, TallShape{256, 1, 1, 1, 64}};

// the order in which an untuned operator tries them: the one-row-per-workgroup walks first (the winners on most boxes and pairings,
// profiles/repeat_r03_boxes.txt), the sequential sweeps last
const int k_fwd_trial_order[K_FWD_CANDIDATES_TALL] = {7, 6, 1, 4, 3, 2, 0, 5};
const int k_fwd_trial_order_few[K_FWD_CANDIDATES] = {8, 9, 7, 6, 1, 4, 3, 2, 0, 5};
// (the bands also for >= 1024 rows of blocks below 64 MiB: 1024 x 128^3 with a non-diagonal row runs its banded forward + adjoint pair at 6.55 TB/s
// where the all-diagonal operator's best of eight gave 6.31)
// The shape candidate k RUNS on rows of `nvec` 16-byte packs -- in a trial, once chosen, in the periodic re-check and when an operator inherits
// the choice (walk memory): the column-persistent walk (candidate 5) has one workgroup per 128 KiB of a ROW, so with small blocks it is a
// handful of workgroups walking thousands of rows (4096 x 64^3: 17.8 ms where the others take 1.4-1.7) -- there candidate 5 IS candidate 0's
// shape, everywhere, so a timing of "5" is always a timing of what a choice of 5 would run (round-4 advisor finding: the trial alone was
// substituted, and a tie or play-off won by 5 then ran the real column-persistent walk for ~192 calls until the re-check rotated it out)
static inline TallShape fwd_candidate_shape(int k, int64_t nvec)
{
    if (k == 5 && nvec < (int64_t)512 * 1024 * 8) return k_fwd_candidates[0];
    return k_fwd_candidates[k];
}

static inline int fwd_candidates_of(const jh_blockop *op)
{
    const bool small_blocks = (double)op->row_len[0] * (double)jh_dtype_size(op->dtype) < (double)(64u << 20);
    return (op->nrow < 1024 || small_blocks) ? K_FWD_CANDIDATES : K_FWD_CANDIDATES_TALL;
}

// For operators far larger than the caches the row-concurrent walk is 5-7 % faster than the sequential sweep in some
// processes and 10-15 % slower in others (profiles/repeat_r01.txt: same binary, same box; it depends on where the slabs landed
// physically), so the shape is chosen by measurement -- LAZILY: while an operator is untuned, each real forward call runs the
// next candidate shape between two events (every candidate computes the same bits), nothing is launched that the caller did
// not ask for and the host never waits; a later call harvests the finished timings with hipEventQuery and, once every
// candidate has been measured twice (the first pass also warms caches and TLBs), keeps the fastest.  jh_blockop_mul returns
// after enqueue, always.  Skipped while the stream is being captured.  jh_blockop_tune_get/set export / import the choice.
void lazy_release(jh_blockop::LazyTune &t)
{
    for (auto &pair : t.ev)
        for (auto &e : pair)
            if (e) { (void)hipEventDestroy(e); e = nullptr; }
    for (auto &e : t.rc_ev)                                                 // (the re-check's pair exists only after a choice was made)
        if (e) { (void)hipEventDestroy(e); e = nullptr; }
    t.rc_in_flight = false;
}

void lazy_reset(jh_blockop::LazyTune &t)
{
    lazy_release(t);
    for (auto &st : t.state) st = 0;
    for (auto &m : t.ms) m = 0.f;
    for (auto &m : t.best_ms) m = 0.f;
    t.launched = 0;
    t.playoff[0] = t.playoff[1] = -1;
    t.calls = 0;
    t.rc_in_flight = false;
    t.rc_slow = 0;
}

// Which candidate should THIS call run?  Trial slots are laid out pass-major after `warm` untimed-in-effect slots (their
// timings are discarded): slot = warm + pass * ncand + candidate.  Returns the candidate and, when the call is a trial, its
// slot (else -1).  Once every slot has a timing, *choice = the candidate with the best time over its passes -- candidate 0
// unless another one beats it by `margin` -- and the events are released.
// playoff (round 3): two timings per candidate decide between shapes that often differ by 1-2 %, less than the timings scatter; so
// when the runner-up is within 3 % of the winner the two run a play-off -- four more of the caller's own calls, alternating
// winner / runner-up / winner / runner-up, each between two events like the trials -- and the best time over ALL of a
// candidate's samples decides.
// `order` (optional, ncand entries): the candidate that trial j of a pass runs -- the likely winners first, so that an operator that
// lives for a handful of calls only (a caller in the reference's style builds operators all the time) spends them on good shapes
int lazy_next(jh_blockop::LazyTune &t, int ncand, int npass, int warm, float margin, int *choice, int *slot, bool playoff = false,
              const int *order = nullptr)
{
    *slot = -1;
    const int regular = warm + ncand * npass;
    const int total = regular + (t.playoff[0] >= 0 ? 4 : 0);
    int measured = 0;
    for (int k = 0; k < t.launched; k++) {                                  // harvest what has finished (non-blocking)
        if (t.state[k] == 1 && hipEventQuery(t.ev[k][1]) == hipSuccess) {
            float ms = 0.f;
            t.state[k] = (hipEventElapsedTime(&ms, t.ev[k][0], t.ev[k][1]) == hipSuccess && ms > 0.f) ? 2 : 3;
            t.ms[k] = ms;
        }
        if (t.state[k] >= 2) measured++;
    }
    (void)hipGetLastError();                                                // hipEventQuery's hipErrorNotReady is not an error
    if (measured == total) {
        float best[jh_blockop::LazyTune::MAXC] = {};
        auto take = [&](int c, int k) { if (t.state[k] == 2 && (best[c] == 0.f || t.ms[k] < best[c])) best[c] = t.ms[k]; };
        for (int j = 0; j < ncand; j++)
            for (int p = 0; p < npass; p++) take(order ? order[j] : j, warm + p * ncand + j);
        if (t.playoff[0] >= 0)
            for (int k = 0; k < 4; k++) take(t.playoff[k & 1], regular + k);
        int pick = 0, runner = -1;
        for (int c = 1; c < ncand; c++)
            if (best[c] > 0.f && (best[pick] == 0.f || best[c] < (1.f - margin) * best[pick])) pick = c;
        for (int c = 0; c < ncand; c++)
            if (c != pick && best[c] > 0.f && (runner < 0 || best[c] < best[runner])) runner = c;
        if (playoff && t.playoff[0] < 0 && runner >= 0 && best[pick] > 0.f && best[runner] <= 1.03f * best[pick] &&
            regular + 4 <= jh_blockop::LazyTune::SLOTS) {
            t.playoff[0] = pick;                                            // four more trials; the choice waits for them
            t.playoff[1] = runner;
            *slot = t.launched;
            return pick;
        }
        for (int c = 0; c < ncand && c < jh_blockop::LazyTune::MAXC; c++) t.best_ms[c] = best[c];
        *choice = pick;
        lazy_release(t);
        return pick;
    }
    if (t.launched < total) {
        *slot = t.launched;
        if (*slot >= regular) return t.playoff[(*slot - regular) & 1];
        if (*slot < warm) return order ? order[0] : 0;
        return order ? order[(*slot - warm) % ncand] : (*slot - warm) % ncand;
    }
    return t.playoff[0] >= 0 ? t.playoff[0] : (order ? order[0] : 0);       // every trial is in flight: the (provisional) default meanwhile
}

// Periodic re-check of a choice already made (round 3): every 64th call of the chosen shape is timed between two events (again the
// caller's own launch, harvested later without waiting).  Three such samples in a row that are more than 3 % slower than the best
// time ANOTHER candidate recorded during the search rotate that candidate in; the dethroned one's record is replaced by what it
// has just shown, so the two cannot flip back and forth on stale numbers.  Returns true when THIS call should be timed.
bool recheck_should_time(jh_blockop::LazyTune &t, int ncand, int *choice)
{
    if (t.rc_in_flight && hipEventQuery(t.rc_ev[1]) == hipSuccess) {
        float ms = 0.f;
        t.rc_in_flight = false;
        if (hipEventElapsedTime(&ms, t.rc_ev[0], t.rc_ev[1]) == hipSuccess && ms > 0.f && *choice >= 0 && *choice < jh_blockop::LazyTune::MAXC) {
            int other = -1;
            for (int c = 0; c < ncand && c < jh_blockop::LazyTune::MAXC; c++)
                if (c != *choice && t.best_ms[c] > 0.f && (other < 0 || t.best_ms[c] < t.best_ms[other])) other = c;
            if (other >= 0 && ms > 1.03f * t.best_ms[other]) {
                if (++t.rc_slow >= 3) {
                    t.best_ms[*choice] = ms;
                    *choice = other;
                    t.rc_slow = 0;
                    t.switches++;
                }
            } else {
                t.rc_slow = 0;
                if (ms < t.best_ms[*choice] || t.best_ms[*choice] == 0.f) t.best_ms[*choice] = ms;
            }
        }
    }
    (void)hipGetLastError();
    t.calls++;
    // every 64th call -- and the three calls after a slow sample, so that a real slowdown is confirmed (or dismissed) at once
    return !t.rc_in_flight && (t.calls % 64 == 0 || t.rc_slow > 0);
}

bool recheck_begin(jh_blockop::LazyTune &t, hipStream_t st)
{
    for (auto &e : t.rc_ev)
        if (!e && hipEventCreate(&e) != hipSuccess) return false;
    return hipEventRecord(t.rc_ev[0], st) == hipSuccess;
}

void recheck_end(jh_blockop::LazyTune &t, hipStream_t st, bool ok) { t.rc_in_flight = ok && hipEventRecord(t.rc_ev[1], st) == hipSuccess; }

bool lazy_begin(jh_blockop::LazyTune &t, int slot, hipStream_t st)
{
    return hipEventCreate(&t.ev[slot][0]) == hipSuccess && hipEventCreate(&t.ev[slot][1]) == hipSuccess &&
           hipEventRecord(t.ev[slot][0], st) == hipSuccess;
}

void lazy_end(jh_blockop::LazyTune &t, int slot, hipStream_t st, bool ok)
{
    ok = ok && hipEventRecord(t.ev[slot][1], st) == hipSuccess;
    t.state[slot] = ok ? 1 : 3;
    t.launched = slot + 1;
}

bool stream_is_capturing(hipStream_t st)
{
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    return !(hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone);
}

// Round 4: what an operator of this SHAPE chose last time.  A caller in the reference's style builds operators again and again (a new
// JopBlock per outer iteration, per shot set): each would spend its first 16-20 forwards on the candidate walks, the slow ones included --
// the first `d = A*m` of a new operator ran the 16-row sweep (23-24 ms at the headline size) whatever the last operator had found.
// A new operator of a shape (device, element type, rows, block length, strided or table addressing) seen before starts with that
// choice and its records; the periodic re-check (every 64th call) still corrects it.  Knob walk_memory (0: every operator measures);
// jh_blockop_tune_set(op, "fwd_walk", -1) makes that operator measure for itself.
struct WalkKey {
    int device, dtype, strided;
    int64_t nrow, n_scalars;
    bool operator<(const WalkKey &o) const
    {
        return std::tie(device, dtype, strided, nrow, n_scalars) < std::tie(o.device, o.dtype, o.strided, o.nrow, o.n_scalars);
    }
};
struct WalkRecord { int walk; float best_ms[jh_blockop::LazyTune::MAXC]; };
std::mutex g_walk_mutex;
std::map<WalkKey, WalkRecord> g_walk_memory;

WalkKey walk_key(const jh_blockop *op, int64_t n_scalars) { return WalkKey{jh_ctx().device, op->dtype, op->diag_strided ? 1 : 0, op->nrow, n_scalars}; }

void walk_remember(const jh_blockop *op, int64_t n_scalars)
{
    if (op->fwd_walk < 0 || op->fwd_walk >= K_FWD_CANDIDATES) return;
    WalkRecord r{op->fwd_walk, {}};
    for (int k = 0; k < jh_blockop::LazyTune::MAXC; k++) r.best_ms[k] = op->fwd_tune.best_ms[k];
    std::lock_guard<std::mutex> lock(g_walk_mutex);
    g_walk_memory[walk_key(op, n_scalars)] = r;
}

bool walk_recall(const jh_blockop *op, int64_t n_scalars)
{
    std::lock_guard<std::mutex> lock(g_walk_mutex);
    auto it = g_walk_memory.find(walk_key(op, n_scalars));
    if (it == g_walk_memory.end()) return false;
    op->fwd_walk = it->second.walk;
    for (int k = 0; k < jh_blockop::LazyTune::MAXC; k++) op->fwd_tune.best_ms[k] = it->second.best_ms[k];
    op->walk_inherited = true;
    return true;
}

template <typename S, int E, int NS>
int launch_tall_fwd(const jh_blockop *op, void *d, const void *m, int64_t n_scalars)
{
    jh_context &c = jh_ctx();
    TallShape sh = pick_fwd_shape(n_scalars / NS, op->nrow, sizeof(S) * NS);
    const bool knobs_free = !c.fwd_wg && !c.fwd_unroll && !c.fwd_group && c.fwd_order < 0;
    const double stream_bytes = 2.0 * (double)op->nrow * (double)n_scalars * sizeof(S);
    if (c.autotune && knobs_free && stream_bytes >= 8.0 * (double)(1ull << 30) && op->nrow >= 64) {
        int slot = -1;
        if (op->fwd_walk < 0 && op->fwd_tune.launched == 0 && !op->walk_measure_again && c.walk_memory) (void)walk_recall(op, n_scalars);
        if (op->fwd_walk < 0) {
            if (!stream_is_capturing(c.stream)) {
                const int nc = fwd_candidates_of(op);
                const int k = lazy_next(op->fwd_tune, nc, 2, 0, 0.f, &op->fwd_walk, &slot, true, nc == K_FWD_CANDIDATES ? k_fwd_trial_order_few : k_fwd_trial_order);
                if (k >= 0 && k < K_FWD_CANDIDATES) sh = fwd_candidate_shape(k, n_scalars / NS);
                if (op->fwd_walk >= 0) walk_remember(op, n_scalars);       // the choice has just been made
            }
        } else if (op->fwd_walk < K_FWD_CANDIDATES) {
            if (!stream_is_capturing(c.stream) && recheck_should_time(op->fwd_tune, fwd_candidates_of(op), &op->fwd_walk)) {
                walk_remember(op, n_scalars);                              // (the re-check may have rotated another candidate in)
                sh = fwd_candidate_shape(op->fwd_walk, n_scalars / NS);
                const bool ok = recheck_begin(op->fwd_tune, c.stream);
                const int st = launch_tall_fwd_shape<S, E, NS>(op, d, m, n_scalars, sh);
                recheck_end(op->fwd_tune, c.stream, ok && st == JH_OK);
                return st;
            }
            sh = fwd_candidate_shape(op->fwd_walk, n_scalars / NS);
        }
        if (slot >= 0) {                                                   // a timed trial: the caller's own launch between two events
            const bool ok = lazy_begin(op->fwd_tune, slot, c.stream);
            const int st = launch_tall_fwd_shape<S, E, NS>(op, d, m, n_scalars, sh);
            lazy_end(op->fwd_tune, slot, c.stream, ok && st == JH_OK);
            return st;
        }
    }
    return launch_tall_fwd_shape<S, E, NS>(op, d, m, n_scalars, sh);
}

template <typename S, int E, int NS>
int launch_tall_fwd_shape(const jh_blockop *op, void *d, const void *m, int64_t n_scalars, const TallShape &sh)
{
    const bool nt = jh_stream_nt(2.0 * (double)op->nrow * (double)n_scalars * sizeof(S));     // coefficients read + range vector written
    if (sh.wg == 256) return nt ? launch_tall_fwd_u<S, E, NS, true, 256>(op, d, m, n_scalars, sh) : launch_tall_fwd_u<S, E, NS, false, 256>(op, d, m, n_scalars, sh);
    if (sh.wg == 512) return nt ? launch_tall_fwd_u<S, E, NS, true, 512>(op, d, m, n_scalars, sh) : launch_tall_fwd_u<S, E, NS, false, 512>(op, d, m, n_scalars, sh);
    return nt ? launch_tall_fwd_u<S, E, NS, true, 1024>(op, d, m, n_scalars, sh) : launch_tall_fwd_u<S, E, NS, false, 1024>(op, d, m, n_scalars, sh);
}
template <typename S, int E, int NS, int MODE>
int launch_tall_adj(const jh_blockop *op, void *out, const void *in, int64_t n_scalars, int64_t s_begin = 0, int64_t s_end = -1)
{
    if (s_end < 0) s_end = n_scalars;
    if (s_end <= s_begin) return JH_OK;
    const TallShape sh = pick_adj_shape(n_scalars / NS, op->nrow, MODE);
    // coefficients + the range vector; the fused A'A reads the coefficients alone, but at 256 MiB of them nontemporal loads are already 3 % ahead
    // (profiles/exp_r05_nt_small.txt), so it counts them twice as well
    const bool nt = jh_stream_nt(2.0 * (double)op->nrow * (double)n_scalars * sizeof(S));
    if (sh.wg == 256) return nt ? launch_tall_adj_u<S, E, NS, true, MODE, 256>(op, out, in, n_scalars, sh, s_begin, s_end) : launch_tall_adj_u<S, E, NS, false, MODE, 256>(op, out, in, n_scalars, sh, s_begin, s_end);
    if (sh.wg == 512) return nt ? launch_tall_adj_u<S, E, NS, true, MODE, 512>(op, out, in, n_scalars, sh, s_begin, s_end) : launch_tall_adj_u<S, E, NS, false, MODE, 512>(op, out, in, n_scalars, sh, s_begin, s_end);
    return nt ? launch_tall_adj_u<S, E, NS, true, MODE, 1024>(op, out, in, n_scalars, sh, s_begin, s_end) : launch_tall_adj_u<S, E, NS, false, MODE, 1024>(op, out, in, n_scalars, sh, s_begin, s_end);
}

// Does the plain tall adjoint (MODE 0) take the split walk for this operator?  If so, reserve scratch for its slabs PLUS one
// domain-sized temporary behind them and return that temporary: the kernels that have no split variant of their own
// (fused adjoint update, JetSum adjoint) then run "split adjoint into the temporary + a small epilogue" instead of crawling.
template <typename S, int NS>
int split_adjoint_tmp(const jh_blockop *op, int64_t n_scalars, void **tmp)
{
    *tmp = nullptr;
    if (op->nrow == 1) return JH_OK;
    const TallShape sh = pick_adj_shape(n_scalars / NS, op->nrow, 0);
    const int64_t gx0 = (n_scalars + (int64_t)sh.unroll * sh.wg * NS - 1) / ((int64_t)sh.unroll * sh.wg * NS);
    int64_t parts = pick_adj_parts(gx0, op->nrow);
    if (parts <= 1) return JH_OK;
    const int64_t rows_per_part = (op->nrow + parts - 1) / parts;
    parts = (op->nrow + rows_per_part - 1) / rows_per_part;
    const size_t slab_bytes = ((size_t)parts * (size_t)n_scalars * sizeof(S) + 255) / 256 * 256;
    void *base = nullptr;
    JH_TRY(jh_ensure_scratch(slab_bytes + (size_t)n_scalars * sizeof(S), &base));
    *tmp = (char *)base + slab_bytes;
    return JH_OK;
}

// fast path usable?  (tall, all DIAG, uniform rows, 16-byte aligned everything, no conj flags on complex)
bool tall_fast_ok(const jh_blockop *op, const void *rng_ptr, const void *dom_ptr)
{
    if (!(op->tall && op->all_diag && op->uniform_rows)) return false;
    const size_t es = jh_dtype_size(op->dtype);
    const int64_t n = op->row_len[0];
    if (n == 0 || (n * (int64_t)es) % 16 != 0) return false;
    if ((((uintptr_t)rng_ptr) | ((uintptr_t)dom_ptr)) & 15u) return false;
    for (const auto &b : op->blocks)
        if (((uintptr_t)b.coeff) & 15u) return false;
    return true;
}

// mixed tall path usable?  Tall with >= 2 equal rows of ANY elementwise kind (ZERO / IDENTITY / SCALE / DIAG, adjointed or
// not / the Jacobian of SQUARE), everything 16-byte aligned -- the rows a pure-diagonal operator gains when a regularisation
// row (identity, scalar) or a muted shot (zero block) joins it.  Such operators keep the tall kernels' tiling, the fused A'A
// and the one-pass LSQR step instead of dropping to the general M x K kernels.
bool tall_mixed_ok(const jh_blockop *op, const void *rng_ptr, const void *dom_ptr)
{
    if (!(op->tall && op->uniform_rows && op->elementwise) || op->nrow < 2 || op->all_diag) return false;
    const size_t es = jh_dtype_size(op->dtype);
    const int64_t n = op->row_len[0];
    if (n == 0 || (n * (int64_t)es) % 16 != 0) return false;
    if ((((uintptr_t)rng_ptr) | ((uintptr_t)dom_ptr)) & 15u) return false;
    for (const auto &b : op->blocks)
        if ((b.kind == JH_OP_DIAG || b.kind == JH_OP_SQUARE) && (((uintptr_t)b.coeff) & 15u)) return false;
    return true;
}

// one shape per kernel for the mixed rows (they are the exception; the all-DIAG instantiations keep their tuned shapes)
template <typename S, int E, int NS>
int launch_tall_fwd_mixed(const jh_blockop *op, void *d, const void *m, int64_t n_scalars)
{
    jh_context &c = jh_ctx();
    // late round 4: one pack per lane, two rows per workgroup, COLUMN bands of 32 tiles (128 KiB of each row, then the same tiles of the next
    // row group: k_tall_diag_fwd's ctiles decode) -- against round 3's 256 x 4 packs x 4 rows in a sequential sweep: 1024 x 64^3 5.0 -> 6.1 TB/s,
    // 4096 x 64^3 5.3 -> 6.3, 16384 x 32^3 5.5 -> 6.4, 64 x 128^3 5.6 -> 6.4, 2048 x 128^3 5.6 -> 6.35, 256 x 256^3 5.8 -> 6.4
    // (profiles/exp_r04_mixed_fwd.txt).  Knobs fwd_group / fwd_ctiles override rows per workgroup / tiles per band (0: sequential sweep).
    constexpr int BLK = 256, U = 1;
    int64_t G = c.fwd_group > 0 ? c.fwd_group : 2;
    if (G > op->nrow) G = op->nrow;
    const int64_t gx = (n_scalars + (int64_t)U * BLK * NS - 1) / ((int64_t)U * BLK * NS);
    int64_t gy = (op->nrow + G - 1) / G;
    while (gx * gy * BLK >= ((int64_t)1 << 32) && G < op->nrow) { G *= 2; gy = (op->nrow + G - 1) / G; }
    JH_REQUIRE(gx * gy * BLK < (int64_t)1 << 32, "tall forward: grid of %lld workgroups is too large", (long long)(gx * gy));
    int64_t ctiles = c.fwd_ctiles >= 0 ? c.fwd_ctiles : 32;
    if (ctiles > gx) ctiles = gx;
    c.last_fwd_walk = ctiles ? 2 : 0;
    c.last_fwd_rows_per_wg = G;
    hipLaunchKernelGGL((k_tall_diag_fwd<S, E, NS, U, true, BLK, true>), dim3((unsigned)(gx * gy)), dim3(BLK), 0, c.stream, op->dev_blocks,
                       op->nrow, (int)G, (const S *)nullptr, (int64_t)0, (const S *)m, (S *)d, n_scalars, (unsigned)gx, (unsigned)gy, 1u, (unsigned)ctiles);
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

template <typename S, int E, int NS, int MODE, int BLK, int U, int DEPTH>
int launch_tall_adj_mixed_u(const jh_blockop *op, void *out, const void *in, int64_t n_scalars, int64_t s_begin, int64_t s_end)
{
    jh_context &c = jh_ctx();
    const int64_t gx = (s_end - s_begin + (int64_t)U * BLK * NS - 1) / ((int64_t)U * BLK * NS);
    const int from_found = (MODE == 0) ? c.adj_from_found : 0;                        // continue from what `out` holds (a wide operator's forward)
    int64_t parts = from_found ? 1 : pick_adj_parts(gx, op->nrow), rows_per_part = 0;   // many rows of small blocks: split-row walk
    const int64_t part_stride = s_end - s_begin;
    void *slabs = nullptr;
    if (parts > 1) {
        rows_per_part = (op->nrow + parts - 1) / parts;
        parts = (op->nrow + rows_per_part - 1) / rows_per_part;
        JH_TRY(jh_ensure_scratch((size_t)parts * (size_t)part_stride * sizeof(S), &slabs));
    }
    c.last_adj_parts = parts;
    c.last_adj_launches = 1;
    hipLaunchKernelGGL((k_tall_diag_adj<S, E, NS, U, DEPTH, true, MODE, BLK, true>), dim3((unsigned)gx, (unsigned)parts), dim3(BLK), 0, c.stream,
                       op->dev_blocks, op->nrow, (const S *)nullptr, (int64_t)0, (S *)out, (const S *)in, n_scalars, 0, s_begin, s_end,
                       (int64_t)0, op->nrow, from_found, rows_per_part, (S *)slabs, part_stride);
    JH_CHECK_HIP(hipGetLastError());
    if (parts > 1) return launch_fold_parts<S, NS>(slabs, part_stride, parts, out, s_begin, s_end);
    return JH_OK;
}

template <typename S, int E, int NS, int MODE>
int launch_tall_adj_mixed(const jh_blockop *op, void *out, const void *in, int64_t n_scalars, int64_t s_begin = 0, int64_t s_end = -1)
{
    if (s_end < 0) s_end = n_scalars;
    if (s_end <= s_begin) return JH_OK;
    // the fused normal operator reads ONE stream: fat workgroups with more rows in flight once the blocks are big (like the
    // all-DIAG shapes of pick_adj_shape); everything else 512 x 2 x 2
    // (ComplexF32 with its per-row kind switch: two rows in flight, four spilled 20 bytes per lane)
    if constexpr (MODE == 1) {
        constexpr int DEPTH = (E == 2 && sizeof(S) == 4) ? 2 : 4;
        if (n_scalars / NS >= ((int64_t)1 << 22)) return launch_tall_adj_mixed_u<S, E, NS, MODE, 1024, 4, DEPTH>(op, out, in, n_scalars, s_begin, s_end);
    }
    return launch_tall_adj_mixed_u<S, E, NS, MODE, 512, 2, 2>(op, out, in, n_scalars, s_begin, s_end);
}

// every block boundary / coefficient pointer / vector base on a 16-byte boundary?
bool general_vec_ok(const jh_blockop *op, const void *rng_ptr, const void *dom_ptr)
{
    const int64_t es = (int64_t)jh_dtype_size(op->dtype);
    if ((((uintptr_t)rng_ptr) | ((uintptr_t)dom_ptr)) & 15u) return false;
    for (int64_t v : op->row_len) if ((v * es) % 16) return false;
    for (int64_t v : op->col_len) if ((v * es) % 16) return false;
    for (const auto &b : op->blocks)
        if ((b.kind == JH_OP_DIAG || b.kind == JH_OP_SQUARE) && (((uintptr_t)b.coeff) & 15u)) return false;
    return true;
}

// tiles per line and the 1-D grid of the general kernels: ceil(ntiles / 8) * 8 * nlines workgroups of 256 lanes, < 2^24
// XCD-aware decode or line by line?  The XCD-aware order exists so that a shared input block comes from HBM once; it also
// makes the workgroups dispatched together write (forward) or read (adjoint) one tile of EVERY line at once -- hundreds of
// concurrent streams.  When the whole input vector is small enough to stay in L2 / Infinity Cache between lines anyway, the
// line-by-line order is faster: tall mixed 1024 x 1 of 4 MiB blocks forward 1.54 -> 1.02 ms, 512 x 2 0.40 -> 0.36 ms, while
// 8 x 8 of 16 MiB blocks (128 MiB of input) wants the XCD-aware order, 0.35 -> 0.28 ms (profiles/exp_r01_cliffs.txt).
// Knob general_xcd: 1 automatic (XCD-aware from 32 MiB of input on), 0 never, 2 always.
static inline bool general_use_xcd(int64_t input_bytes)
{
    const int64_t k = jh_ctx().general_xcd;
    return k == 2 || (k == 1 && input_bytes >= ((int64_t)32 << 20));
}

static inline void general_grid(int64_t want_tiles, int64_t nlines, unsigned &ntiles, unsigned &grid, bool xcd)
{
    const int64_t band = jh_ctx().general_band;                            // tiles per band: 8, 16, 32 or 64 (knob general_band)
    const unsigned k = band >= 64 ? 3u : (band >= 32 ? 2u : (band >= 16 ? 1u : 0u));
    const int64_t T = (int64_t)8 << k;
    int64_t cap = (((int64_t)1 << 24) / nlines) / T * T - T;               // grid * 256 threads < 2^32
    if (cap < T) cap = T;
    if (want_tiles > cap) want_tiles = cap;                                // the kernels stride over the rest
    if (want_tiles < 1) want_tiles = 1;
    ntiles = (unsigned)want_tiles | (k << 28);
    grid = (unsigned)(((want_tiles + T - 1) / T) * T * nlines);
    if (!xcd) ntiles |= 0x80000000u;                                       // flag for the kernels' decode: tile fastest, line by line
}

// Split walk of the general kernels.  One line (block row of the forward, block column of the adjoint) is summed by the
// threads that own its elements, over ALL blocks of the line: a wide operator of many small blocks (or a tall one, in the
// adjoint) launches a handful of workgroups that each walk thousands of blocks -- 1 x 16384 blocks of 16384 Float32: forward
// 12.4 ms, 173 GB/s (profiles/exp_r01_cliffs.txt).  When the summed dimension has >= 256 blocks and the launch would have
// fewer workgroups than the chip has CUs, it is cut into `parts` ranges (grid.y), each summed in order into its own slab,
// and k_fold_general adds the output as found (forward: `_d .+=`, 1024) and the slabs.  Deterministic; tolerance parity.
// Same knob as the tall kernels: adj_split (-1 automatic, 0 never, k parts).
int64_t general_parts(int64_t wgs, int64_t nsum, int64_t out_bytes)
{
    jh_context &c = jh_ctx();
    if (c.adj_split == 0 || nsum < 4) return 1;
    int64_t parts;
    if (c.adj_split > 0) parts = c.adj_split;
    else {
        // these kernels keep ONE block's loads in flight per thread (GENERAL_Q), so they want more workgroups than the tall walk
        if (wgs >= 4 * (int64_t)c.cu_count || nsum < 256) return 1;
        parts = (16 * (int64_t)c.cu_count + wgs - 1) / wgs;
        if (parts > nsum / 16) parts = nsum / 16;
    }
    if (parts > nsum / 2) parts = nsum / 2;
    if (parts > 65535) parts = 65535;
    while (parts > 1 && (double)parts * (double)out_bytes > 256.0 * (double)(1 << 20)) parts /= 2;   // scratch for the slabs
    return parts < 2 ? 1 : parts;
}

template <typename S>
int launch_fold_general(const void *slabs, int64_t slab_stride, int64_t parts, void *out, const int64_t *dev_off, int E, int64_t nlines,
                        int64_t max_scalars, const unsigned char *touched, int add_found)
{
    int64_t gx = (max_scalars + 63) / 64;
    if (gx > 4096) gx = 4096;
    if (gx < 1) gx = 1;
    JH_REQUIRE(nlines <= 65535, "split walk: %lld lines exceed the grid", (long long)nlines);
    hipLaunchKernelGGL((k_fold_general<S>), dim3((unsigned)gx, (unsigned)nlines), dim3(256), 0, jh_ctx().stream, (const S *)slabs, slab_stride,
                       (int)parts, (S *)out, dev_off, E, touched, add_found);
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

// launch of the register-tiled grid kernel.  Shapes from tools/micro/grid_tile.hip (profiles/exp_r03_grid_tile.txt; same-box sweeps):
// blocks of >= 16 MiB: 2 lines x 1 pack per lane, two steps' loads in flight -- small, short-lived workgroups win there (4 or 8 lines per
// workgroup are 1-5 % slower, more packs per lane too); smaller blocks: 4 lines x 2 packs (32 x 32 of 128^3: 6.2 against 5.4 TB/s).
template <typename S, int E, int NS, bool TRANSPOSED>
int launch_grid_tile(const jh_blockop *op, const S *in, S *out, int64_t in_bytes)
{
    jh_context &c = jh_ctx();
    const int64_t nlines = TRANSPOSED ? op->ncol : op->nrow, n_scalars = op->row_len[0] * E;
    int R = c.grid_tile > 1 ? (int)c.grid_tile : ((n_scalars * (int64_t)sizeof(S) >= ((int64_t)16 << 20) || nlines < 4) ? 2 : 4);
    const int U = R == 4 ? 2 : 1;
    const int64_t ngroups = (nlines + R - 1) / R;
    unsigned ntiles, grid;
    general_grid((n_scalars / NS + 256 * U - 1) / (256 * U), ngroups, ntiles, grid, general_use_xcd(in_bytes));
#define JH_TILE(RR, QQ, UU) hipLaunchKernelGGL((k_grid_tile<S, E, NS, RR, QQ, UU, TRANSPOSED>), dim3(grid), dim3(256), 0, c.stream, op->dev_blocks, \
                                               op->nrow, op->ncol, n_scalars, in, out, ntiles, (unsigned)ngroups)
    if (R == 8) JH_TILE(8, 2, 1);
    else if (R == 4) JH_TILE(4, 2, 2);
    else JH_TILE(2, 2, 1);
#undef JH_TILE
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

// the register-tiled general kernel applies to grids (>= 2 x 2) of EQUAL, 16-byte aligned elementwise blocks
bool general_tile_ok(const jh_blockop *op, const void *rng_ptr, const void *dom_ptr)
{
    if (!(op->elementwise && op->uniform_rows && op->nrow >= 2 && op->ncol >= 2)) return false;
    const int64_t n = op->row_len[0];
    if (n == 0) return false;
    for (int64_t v : op->col_len) if (v != n) return false;
    return general_vec_ok(op, rng_ptr, dom_ptr);
}

template <typename S, int E, int NS, bool TRANSPOSED>
int launch_general_tile(const jh_blockop *op, const S *in, S *out, int64_t in_bytes)
{
    jh_context &c = jh_ctx();
    const int64_t nlines = TRANSPOSED ? op->ncol : op->nrow, n_scalars = op->row_len[0] * E;
    // round 4: FOUR lines per workgroup, one step in flight (the input pack is loaded once for four lines), whenever there are four lines:
    // same box against two lines x two steps, forward | adjoint: 32 x 32 of 128^3 4.88 -> 5.28 | 4.78 -> 5.36 TB/s, 16 x 16 of 256^3
    // 5.43 -> 5.90 | 5.28 -> 5.87, 64 x 64 of 64^3 5.1 -> 5.7 | 5.1 -> 5.7, 8 x 8 and 64 x 4 +2 % (profiles/bench_grid_mixed_r04.txt).
    // Knob general_tile: 1 this rule, 2 / 4 that many lines always, 0 the one-line kernels
    const bool four = c.general_tile == 4 || (c.general_tile == 1 && nlines >= 4);
    const int exp_r = c.general_tile == 42 ? 4 : (c.general_tile == 8 ? 8 : 0);     // round-5 experiment shapes: 42 = 4 lines x 2 steps, 8 = 8 lines x 1 step (8 x 2 needs more than 512 registers per lane)
    const int64_t ngroups = exp_r ? (nlines + exp_r - 1) / exp_r : (four ? (nlines + 3) / 4 : (nlines + 1) / 2);
    const int U = (c.fwd_unroll == 2) ? 2 : 1;                           // two packs per lane did not pay here (knob fwd_unroll = 2: measurements)
    unsigned ntiles, grid;
    general_grid((n_scalars / NS + 256 * U - 1) / (256 * U), ngroups, ntiles, grid, general_use_xcd(in_bytes));
    if (c.general_tile == 42)
        hipLaunchKernelGGL((k_general_tile<S, E, NS, 2, 1, TRANSPOSED, 4>), dim3(grid), dim3(256), 0, c.stream, op->dev_blocks, op->nrow, op->ncol, n_scalars, in, out,
                           ntiles, (unsigned)ngroups);
    else if (c.general_tile == 8)
        hipLaunchKernelGGL((k_general_tile<S, E, NS, 1, 1, TRANSPOSED, 8>), dim3(grid), dim3(256), 0, c.stream, op->dev_blocks, op->nrow, op->ncol, n_scalars, in, out,
                           ntiles, (unsigned)ngroups);
    else if (four)
        hipLaunchKernelGGL((k_general_tile<S, E, NS, 1, 1, TRANSPOSED, 4>), dim3(grid), dim3(256), 0, c.stream, op->dev_blocks, op->nrow, op->ncol, n_scalars, in, out,
                           ntiles, (unsigned)ngroups);
    else if (U == 2)
        hipLaunchKernelGGL((k_general_tile<S, E, NS, 2, 2, TRANSPOSED>), dim3(grid), dim3(256), 0, c.stream, op->dev_blocks, op->nrow, op->ncol, n_scalars, in, out,
                           ntiles, (unsigned)ngroups);
    else
        hipLaunchKernelGGL((k_general_tile<S, E, NS, 2, 1, TRANSPOSED>), dim3(grid), dim3(256), 0, c.stream, op->dev_blocks, op->nrow, op->ncol, n_scalars, in, out,
                           ntiles, (unsigned)ngroups);
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

template <typename S, int E>
int general_fwd(const jh_blockop *op, void *d, const void *m, int fmode = 0)
{
    jh_context &c = jh_ctx();
    int64_t maxn = 0;
    for (int64_t i = 0; i < op->nrow; i++) maxn = op->row_len[i] > maxn ? op->row_len[i] : maxn;
    if (maxn == 0) return JH_OK;
    JH_REQUIRE(op->nrow < ((int64_t)1 << 20), "general block forward supports fewer than 2^20 block rows (got %lld)", (long long)op->nrow);
    unsigned ntiles, grid;
    const bool vec = general_vec_ok(op, d, m);
    constexpr int NS = 16 / sizeof(S);
    int64_t want = vec ? (maxn * E / NS + 255) / 256 : (maxn + 255) / 256;     // vec: one pack per thread (see jh_vecops.hip: grid_full)
    if (!vec && want > 4096) want = 4096;
    const bool gdiag = vec && !fmode && c.grid_diag && grid_diag_ok(op, d, m);  // a grid of plain diagonals: the branch-free kernel,
    const int gu = gdiag ? (c.grid_diag >= 4 ? 4 : (c.grid_diag >= 2 ? 2 : 1)) : 1;   // gu packs per lane
    general_grid(want, op->nrow, ntiles, grid, general_use_xcd(op->col_off[(size_t)op->ncol] * (int64_t)(sizeof(S) * E)));
    // split walk over the block columns (general_parts)
    const int64_t out_scalars = op->row_off[(size_t)op->nrow] * E;
    int64_t parts = (op->nrow <= 65535) ? general_parts((int64_t)grid, op->ncol, out_scalars * (int64_t)sizeof(S)) : 1;
    int64_t per = 0;
    void *slabs = nullptr;
    if (parts > 1) {
        per = (op->ncol + parts - 1) / parts;
        parts = (op->ncol + per - 1) / per;
        JH_TRY(jh_ensure_scratch((size_t)parts * (size_t)out_scalars * sizeof(S), &slabs));
    }
    c.last_adj_parts = parts;
    if (gdiag && parts == 1 && c.grid_tile)
        return launch_grid_tile<S, E, NS, false>(op, (const S *)m, (S *)d, op->col_off[(size_t)op->ncol] * (int64_t)(sizeof(S) * E));
    if (vec && !fmode && parts == 1 && c.general_tile && general_tile_ok(op, d, m))
        return launch_general_tile<S, E, NS, false>(op, (const S *)m, (S *)d, op->col_off[(size_t)op->ncol] * (int64_t)(sizeof(S) * E));
    if (gdiag && parts == 1) {
        if (gu > 1) general_grid((want + gu - 1) / gu, op->nrow, ntiles, grid, general_use_xcd(op->col_off[(size_t)op->ncol] * (int64_t)(sizeof(S) * E)));
#define JH_GRID(UU) hipLaunchKernelGGL((k_grid_diag<S, E, NS, 4, false, UU>), dim3(grid), dim3(256), 0, c.stream, op->dev_blocks, op->nrow, op->ncol, \
                                       op->row_len[0] * E, (const S *)m, (S *)d, ntiles)
        if (gu == 4) JH_GRID(4); else if (gu == 2) JH_GRID(2); else JH_GRID(1);
#undef JH_GRID
        JH_CHECK_HIP(hipGetLastError());
        return JH_OK;
    }
    if (vec)
        hipLaunchKernelGGL((k_block_fwd_general_vec<S, E, NS>), dim3(grid, (unsigned)parts), dim3(256), 0, c.stream,
                           op->dev_blocks, op->nrow, op->ncol, op->dev_row_off, op->dev_col_off, (const S *)m, (S *)d, fmode, ntiles,
                           per, (S *)slabs, out_scalars);
    else
        hipLaunchKernelGGL((k_block_fwd_general<S, E>), dim3(grid, (unsigned)parts), dim3(256), 0, c.stream,
                           op->dev_blocks, op->nrow, op->ncol, op->dev_row_off, op->dev_col_off, (const S *)m, (S *)d, fmode, ntiles,
                           per, (S *)slabs, out_scalars);
    JH_CHECK_HIP(hipGetLastError());
    if (parts > 1)      // JetBlock_f! touches every row (1001); the linear loop leaves a row of zero blocks as found (1022)
        return launch_fold_general<S>(slabs, out_scalars, parts, d, op->dev_row_off, E, op->nrow, maxn * E, fmode ? nullptr : op->dev_row_touched, 1);
    return JH_OK;
}

template <typename S, int E>
int general_adj(const jh_blockop *op, void *m, const void *d)
{
    jh_context &c = jh_ctx();
    int64_t maxn = 0;
    for (int64_t j = 0; j < op->ncol; j++) maxn = op->col_len[j] > maxn ? op->col_len[j] : maxn;
    if (maxn == 0) return JH_OK;
    JH_REQUIRE(op->ncol < ((int64_t)1 << 20), "general block adjoint supports fewer than 2^20 block columns (got %lld)", (long long)op->ncol);
    unsigned ntiles, grid;
    const bool vec = general_vec_ok(op, d, m);
    constexpr int NS = 16 / sizeof(S);
    int64_t want = vec ? (maxn * E / NS + 255) / 256 : (maxn + 255) / 256;
    if (!vec && want > 4096) want = 4096;
    general_grid(want, op->ncol, ntiles, grid, general_use_xcd(op->row_off[(size_t)op->nrow] * (int64_t)(sizeof(S) * E)));
    // split walk over the block rows (general_parts); nrow >= 4 there, so every column is zeroed first (1042): all lines touched
    const int64_t out_scalars = op->col_off[(size_t)op->ncol] * E;
    int64_t parts = (op->ncol <= 65535) ? general_parts((int64_t)grid, op->nrow, out_scalars * (int64_t)sizeof(S)) : 1;
    int64_t per = 0;
    void *slabs = nullptr;
    if (parts > 1) {
        per = (op->nrow + parts - 1) / parts;
        parts = (op->nrow + per - 1) / per;
        JH_TRY(jh_ensure_scratch((size_t)parts * (size_t)out_scalars * sizeof(S), &slabs));
    }
    c.last_adj_parts = parts;
    if (vec && parts == 1 && c.grid_diag && c.grid_tile && grid_diag_ok(op, d, m))
        return launch_grid_tile<S, E, NS, true>(op, (const S *)d, (S *)m, op->row_off[(size_t)op->nrow] * (int64_t)(sizeof(S) * E));
    if (vec && parts == 1 && c.general_tile && general_tile_ok(op, d, m))
        return launch_general_tile<S, E, NS, true>(op, (const S *)d, (S *)m, op->row_off[(size_t)op->nrow] * (int64_t)(sizeof(S) * E));
    if (vec && parts == 1 && c.grid_diag && grid_diag_ok(op, d, m)) {
        const int gu = c.grid_diag >= 4 ? 4 : (c.grid_diag >= 2 ? 2 : 1);
        if (gu > 1) general_grid((want + gu - 1) / gu, op->ncol, ntiles, grid, general_use_xcd(op->row_off[(size_t)op->nrow] * (int64_t)(sizeof(S) * E)));
#define JH_GRID(UU) hipLaunchKernelGGL((k_grid_diag<S, E, NS, 4, true, UU>), dim3(grid), dim3(256), 0, c.stream, op->dev_blocks, op->nrow, op->ncol, \
                                       op->row_len[0] * E, (const S *)d, (S *)m, ntiles)
        if (gu == 4) JH_GRID(4); else if (gu == 2) JH_GRID(2); else JH_GRID(1);
#undef JH_GRID
        JH_CHECK_HIP(hipGetLastError());
        return JH_OK;
    }
    if (vec)
        hipLaunchKernelGGL((k_block_adj_general_vec<S, E, NS>), dim3(grid, (unsigned)parts), dim3(256), 0, c.stream,
                           op->dev_blocks, op->nrow, op->ncol, op->dev_row_off, op->dev_col_off, (S *)m, (const S *)d, ntiles,
                           per, (S *)slabs, out_scalars, (parts == 1 && c.nt && out_scalars * (int64_t)sizeof(S) >= ((int64_t)64 << 20)) ? 1 : 0);
    else
        hipLaunchKernelGGL((k_block_adj_general<S, E>), dim3(grid, (unsigned)parts), dim3(256), 0, c.stream,
                           op->dev_blocks, op->nrow, op->ncol, op->dev_row_off, op->dev_col_off, (S *)m, (const S *)d, ntiles,
                           per, (S *)slabs, out_scalars);
    JH_CHECK_HIP(hipGetLastError());
    if (parts > 1) return launch_fold_general<S>(slabs, out_scalars, parts, m, op->dev_col_off, E, op->ncol, maxn * E, nullptr, 0);
    return JH_OK;
}

// ---- fused solver updates: launch + partial fold ---------------------------------------------------
// normsq != NULL: read the folded sum back (synchronises).  normsq == NULL and defer: add it to the device-side accumulator
// instead (no host synchronisation at all).  normsq == NULL and !defer: the caller does not want the norm.
int finish_normsq(int64_t nparts, double *normsq, bool defer = false, int private_slot = -1)
{
    // private_slot >= 0: add the folded sum to red_dev[private_slot] and return without reading anything back (a walk in
    // several row launches sums its launches on the device and reads ONE value at the end)
    jh_context &c = jh_ctx();
    const int accum = ((!normsq && defer) || private_slot >= 0) ? 1 : 0;
    double *dst = private_slot >= 0 ? c.red_dev + private_slot : (accum ? c.red_dev + JH_NORMSQ_SLOT : c.red_dev);
    if (private_slot >= 0) normsq = nullptr;
    if (nparts > 8192) {        // two levels: <= 1024 chunk sums (red_dev + 16 ...), then one workgroup
        const int64_t nchunk = 1024, chunk = (nparts + nchunk - 1) / nchunk;
        hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)nchunk), dim3(256), 0, c.stream, c.part_dev, nparts, chunk, c.red_dev + 16, 0);
        JH_CHECK_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c.stream, c.red_dev + 16, nchunk, nchunk, dst, accum);
    } else {
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c.stream, c.part_dev, nparts, nparts, dst, accum);
    }
    JH_CHECK_HIP(hipGetLastError());
    if (normsq) {
        JH_CHECK_HIP(hipMemcpyAsync(c.red_host, c.red_dev, sizeof(double), hipMemcpyDeviceToHost, c.stream));
        JH_CHECK_HIP(hipMemcpyAsync(c.red_host + 3, c.red_dev + JH_CHAIN_ERR_SLOT, sizeof(double), hipMemcpyDeviceToHost, c.stream));
        JH_CHECK_HIP(hipStreamSynchronize(c.stream));
        *normsq = c.red_host[0];
        JH_TRY(jh_chain_err_check());
    }
    return JH_OK;
}

template <typename S, int E, int NS>
int launch_fwd_update(const jh_blockop *op, void *d, const void *m, int64_t n_scalars, double alpha, double beta, double *normsq, bool wide = false)
{
    jh_context &c = jh_ctx();
    const S *a_base = op->diag_strided ? (const S *)op->blocks[0].coeff : nullptr;
    const int64_t a_stride = op->diag_stride_elems * E;
    const int64_t nvec = n_scalars / NS;
    // three streams per row (a, d in, d out).  Late round 4: one pack per lane, two rows per workgroup, COLUMN bands of 32 tiles (k_tall_diag_fwd's
    // walk: 128 KiB of a row group, then the same tiles of the next, ...) -- against round 1's 256 x 4 packs x 4 rows sequential: beta = 0 (the pass of
    // `(a * A) * m`) 6.07 / 5.65 / 5.92 / 5.88 -> 6.22 / 6.14 / 6.13 / 6.17 TB/s at 128 x 256^3 / 256 x 256^3 / 1024 x 128^3 / 1024 x 256^3,
    // beta != 0 5.73-5.82 -> 6.02-6.27 (profiles/exp_r04_update_fwd.txt)
    int wg = 256, U = 1, G = 2, walk = 32;
    // knob overrides: fwd_unroll 4 / 1 (the two instantiated tilings), fwd_group rows per workgroup, fwd_order 0 / 1 the sequential / row-concurrent
    // walk of rounds 1-3, fwd_ctiles tiles per band
    if (c.fwd_unroll == 4) { U = 4; G = 4; }
    if (c.fwd_group) G = (int)c.fwd_group;
    if (c.fwd_order == 0 || c.fwd_order == 1) walk = (int)c.fwd_order;
    if (c.fwd_ctiles >= 2) walk = (int)c.fwd_ctiles;
    else if (c.fwd_ctiles == 0 && walk >= 2) walk = 0;
    // Which walk: like the plain forward the row-concurrent walk (256 x 4 packs, two rows) wins in some processes at full size (1024 x 256^3,
    // beta = 0: 6.42 against 6.17) and loses in others.  This kernel updates d in place, so it cannot be re-run for timing: the first two real calls
    // on a large operator use the bands and the row-concurrent walk and are timed with events (only when the caller asked for the norm, i.e. the
    // call synchronises anyway); later calls use the faster one (upd_walk: 0 bands, 1 row-concurrent).
    const double stream_bytes = 3.0 * (double)op->nrow * (double)n_scalars * sizeof(S);
    const bool knobs_free = !c.fwd_wg && !c.fwd_unroll && !c.fwd_group && c.fwd_order < 0 && c.fwd_ctiles < 0;
    const bool mixed = !op->all_diag;                       // rows of several elementwise kinds: the bands, always
    const bool tunable = !mixed && c.autotune && knobs_free && normsq && stream_bytes >= 8.0 * (double)(1ull << 30) && op->nrow >= 64;
    int trial = -1;
    if (tunable) {
        int which = op->upd_walk;
        if (which < 0) { trial = op->upd_trials; which = trial; }        // trial 0 -> bands, trial 1 -> row-concurrent
        if (which == 1) { U = 4; G = 2; walk = 1; }
    }
    if (G > op->nrow) G = (int)op->nrow;
    const int64_t gx = (nvec + (int64_t)wg * U - 1) / ((int64_t)wg * U);
    int64_t gy = (op->nrow + G - 1) / G;
    while (gx * gy * wg >= ((int64_t)1 << 32) && G < op->nrow) { G *= 2; gy = (op->nrow + G - 1) / G; }   // HIP: grid x block < 2^32 threads
    JH_REQUIRE(gx * gy * wg < ((int64_t)1 << 32), "fused forward update: grid of %lld workgroups is too large", (long long)(gx * gy));
    JH_TRY(jh_ensure_partials(gx * gy));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (trial >= 0) {
        JH_CHECK_HIP(hipEventCreate(&e0));
        JH_CHECK_HIP(hipEventCreate(&e1));
        JH_CHECK_HIP(hipEventRecord(e0, c.stream));
    }
#define JH_LAUNCH_W(BLK, UU, MX, WD)                                                                                   \
    hipLaunchKernelGGL((k_tall_diag_fwd_update<S, E, NS, UU, BLK, MX, WD>), dim3((unsigned)(gx * gy)), dim3(BLK), 0, c.stream, \
                       op->dev_blocks, op->nrow, G, a_base, a_stride, (const S *)m, (S *)d, n_scalars, (unsigned)gx,   \
                       (unsigned)gy, walk, (S)alpha, (S)beta, c.part_dev, alpha)
    // (a wide scalar: Float32 elements only, beta == 0 -- checked by the caller; one instantiation per tiling)
    if constexpr (sizeof(S) == 4) {
        if (wide) {
            if (mixed && U == 4) JH_LAUNCH_W(256, 4, true, true);
            else if (mixed) JH_LAUNCH_W(256, 1, true, true);
            else if (U == 4) JH_LAUNCH_W(256, 4, false, true);
            else JH_LAUNCH_W(256, 1, false, true);
        }
    }
    if (!(sizeof(S) == 4 && wide)) {
        if (mixed && U == 4) JH_LAUNCH_W(256, 4, true, false);
        else if (mixed) JH_LAUNCH_W(256, 1, true, false);
        else if (U == 4) JH_LAUNCH_W(256, 4, false, false);
        else JH_LAUNCH_W(256, 1, false, false);
    }
#undef JH_LAUNCH_W
    JH_CHECK_HIP(hipGetLastError());
    if (trial >= 0) JH_CHECK_HIP(hipEventRecord(e1, c.stream));
    const int st = finish_normsq(gx * gy, normsq);          // synchronises (normsq != NULL on a trial)
    if (trial >= 0) {
        float ms = 0.f;
        if (st == JH_OK && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) {
            op->upd_ms[trial] = ms;
            op->upd_trials = trial + 1;
            if (op->upd_trials == 2) op->upd_walk = (op->upd_ms[1] < op->upd_ms[0]) ? 1 : 0;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    return st;
}

template <typename S, int E, int NS>
int launch_adj_update(const jh_blockop *op, void *out, const void *in, int64_t n_scalars, double alpha, double beta, double gamma,
                      double *normsq, bool wide = false)
{
    jh_context &c = jh_ctx();
    const S *a_base = op->diag_strided ? (const S *)op->blocks[0].coeff : nullptr;
    const int64_t a_stride = op->diag_stride_elems * E;
    const int64_t nvec = n_scalars / NS;
    const int direct = op->nrow == 1 ? 1 : 0;
    {   // many rows of small blocks: the row sum through the split walk of the plain adjoint, then out = (alpha*gamma)*t + beta*out
        // with ||out||^2 in a small epilogue (tolerance parity, like every split sum)
        void *tmp = nullptr;
        if (!wide) JH_TRY((split_adjoint_tmp<S, NS>(op, n_scalars, &tmp)));   // (a wide in_scale is applied per d_i before the sum: the ordered walk)
        if (tmp) {
            JH_TRY((launch_tall_adj<S, E, NS, 0>(op, tmp, in, n_scalars)));
            int64_t g = (n_scalars + 255) / 256;
            if (g > 2048) g = 2048;
            JH_TRY(jh_ensure_partials(g));
            hipLaunchKernelGGL((k_axpby_norm<S>), dim3((unsigned)g), dim3(256), 0, c.stream, (S *)out, (const S *)tmp, n_scalars,
                               (S)(alpha * gamma), (S)beta, c.part_dev);
            JH_CHECK_HIP(hipGetLastError());
            return finish_normsq(g, normsq);
        }
    }
    c.last_adj_parts = 1;
    int wg = 256, U = 1;
    if (nvec >= 4 * 256 * 256) U = 4;
    else if (nvec >= 2 * 256 * 256) U = 2;
    if (nvec >= ((int64_t)1 << 22)) wg = 512;
    const int64_t gx = (nvec + (int64_t)wg * U - 1) / ((int64_t)wg * U);
    JH_TRY(jh_ensure_partials(gx));
#define JH_LAUNCH_W(BLK, UU, DD, WD)                                                                                   \
    hipLaunchKernelGGL((k_tall_diag_adj_update<S, E, NS, UU, DD, BLK, WD>), dim3((unsigned)gx), dim3(BLK), 0, c.stream, \
                       op->dev_blocks, op->nrow, a_base, a_stride, (S *)out, (const S *)in, n_scalars, direct, (S)alpha, \
                       (S)beta, (S)gamma, c.part_dev, gamma)
    if constexpr (sizeof(S) == 4) {
        if (wide) {
            if (wg == 512) JH_LAUNCH_W(512, 4, 4, true);
            else if (U == 4) JH_LAUNCH_W(256, 4, 2, true);
            else if (U == 2) JH_LAUNCH_W(256, 2, 4, true);
            else JH_LAUNCH_W(256, 1, 4, true);
        }
    }
    if (!(sizeof(S) == 4 && wide)) {
        if (wg == 512) JH_LAUNCH_W(512, 4, 4, false);
        else if (U == 4) JH_LAUNCH_W(256, 4, 2, false);
        else if (U == 2) JH_LAUNCH_W(256, 2, 4, false);
        else JH_LAUNCH_W(256, 1, 4, false);
    }
#undef JH_LAUNCH_W
    JH_CHECK_HIP(hipGetLastError());
    return finish_normsq(gx, normsq);
}

// The one-pass step's launch shape (workgroup x packs per thread x rows in flight) for an operator of nvec 16-byte packs per row.
// profiles/bench_lsqr_step_r01.txt: 1024 x 256^3 wants thin threads with many rows in flight (512 x 1 x 4: 6.13 TB/s),
// 128 x 256^3 fat ones (512 x 4 x 2: 5.46), 64 x 128^3 256 x 4 x 1 (5.9)
struct StepShape { int wg, U, D; };
static StepShape pick_step_shape(const jh_blockop *op, int64_t nvec, bool complex_f32)
{
    jh_context &c = jh_ctx();
    int wg = 256, U = 1, D = 4;
    if (nvec >= 4 * 256 * 256) { U = 4; D = 1; }
    else if (nvec >= 2 * 256 * 256) { U = 2; D = 2; }
    if (nvec >= ((int64_t)1 << 22)) { wg = 512; U = 4; D = 2; }
    if (nvec >= ((int64_t)1 << 22) && op->nrow >= 512) { U = 1; D = 4; }
    if (c.adj_wg) wg = (int)c.adj_wg;                       // the adjoint's knobs select among the instantiated shapes
    if (c.adj_unroll) U = (int)c.adj_unroll;
    if (c.adj_depth) D = (int)c.adj_depth;
    const bool mixed = !op->all_diag;                       // rows of several elementwise kinds (tall_mixed_ok): four instantiated shapes
    if (mixed) {
        auto inst = [](int w_, int u_, int d_) { return (w_ == 512 && u_ == 1 && d_ == 4) || (w_ == 256 && ((u_ == 2 && d_ == 2) || (u_ == 4 && d_ == 1) || (u_ == 1 && d_ == 4))); };
        if (!inst(wg, U, D)) {                              // the all-diagonal rule without the knobs, 512 threads always as 512 x 1 x 4
            wg = 256; U = 1; D = 4;
            if (nvec >= 4 * 256 * 256) { U = 4; D = 1; }
            else if (nvec >= 2 * 256 * 256) { U = 2; D = 2; }
            if (nvec >= ((int64_t)1 << 22)) { wg = 512; U = 1; D = 4; }
        }
    }
    if (complex_f32 && wg == 1024 && U == 4 && D == 1) { U = 2; D = 2; }   // ComplexF32 at 128 VGPRs per lane: 4 x 1 spilled 20 bytes to scratch
    return {wg, U, D};
}

template <typename S, int E, int NS>
int launch_bidiag(const jh_blockop *op, void *u, const void *v, void *w, int64_t n_scalars, double alpha, double beta, double *normsq,
                  int64_t s_begin = 0, int64_t s_end = -1, bool defer = false)
{
    if (s_end < 0) s_end = n_scalars;
    if (s_end <= s_begin) { if (normsq) *normsq = 0.0; return JH_OK; }
    jh_context &c = jh_ctx();
    const S *a_base = op->diag_strided ? (const S *)op->blocks[0].coeff : nullptr;
    const int64_t a_stride = op->diag_stride_elems * E;
    const int64_t nvec = n_scalars / NS;
    const int direct = op->nrow == 1 ? 1 : 0;
    const StepShape shape = pick_step_shape(op, nvec, E == 2 && sizeof(S) == 4);
    int wg = shape.wg, U = shape.U, D = shape.D;
    const bool mixed = !op->all_diag;                       // rows of several elementwise kinds (tall_mixed_ok)
    const int64_t gx = ((s_end - s_begin) / NS + (int64_t)wg * U - 1) / ((int64_t)wg * U);
    // many rows of small blocks: split-row walk (pick_adj_parts): u's rows are updated as before, w's sum is folded from slabs
    int64_t parts = direct ? 1 : pick_adj_parts(gx, op->nrow);
    int64_t rows_per_part = 0;
    const int64_t part_stride = s_end - s_begin;
    void *slabs = nullptr;
    if (parts > 1) {
        rows_per_part = (op->nrow + parts - 1) / parts;
        parts = (op->nrow + rows_per_part - 1) / rows_per_part;
        JH_TRY(jh_ensure_scratch((size_t)parts * (size_t)part_stride * sizeof(S), &slabs));
    }
    c.last_adj_parts = parts;
    JH_TRY(jh_ensure_partials(gx * parts));
    // the knob adj_rows_per_launch splits this walk too (w's ordered sum continues; ||u||^2 adds up), but unlike the plain
    // adjoint it does not pay here: 37.0 ms in two launches of 512 rows vs 34.2 ms in one at 1024 x 256^3 (each launch ends
    // with the read-back of its share of ||u||^2), so one launch is the default
    int64_t rows_per_launch = op->nrow;
    if (c.adj_rows_per_launch > 0) rows_per_launch = c.adj_rows_per_launch < op->nrow ? c.adj_rows_per_launch : op->nrow;
    if (parts > 1) rows_per_launch = op->nrow;
    // HOW the step walks is chosen per operator by measurement (lazy_next: the first seven eligible calls -- whole-vector or
    // ranged alike, the pipelined multi-GPU step only ever makes ranged ones -- each run one mode between two events; no extra
    // launches, no host synchronisation), because which one is fastest depends on the row count AND on where the slabs landed:
    //   mode 0  plain walk: a workgroup lives for all rows of its tile (k_tall_diag_bidiag)
    //   mode 1  the same with XCD-contiguous tiles (+3 % at 128-256 rows of 64 MiB at power-of-two strides, else neutral or worse)
    //   mode 2  chained row chunks (k_tall_diag_bidiag_chain): one batch of 8 rows per workgroup, the ordered sum handed from
    //           chunk to chunk -- same bits; 6.1-6.2 TB/s at 64-512 rows of 64 MiB where the plain walk gives 5.2-5.5 in most
    //           processes and the same in some, -2 ... +4 % at 1024 rows (profiles/ab_r02_step_chain.txt); rows of >= 16 MiB only
    //           (rows of any elementwise kind: the MIXED instantiation, profiles/bench_mixed_rows_r02.txt)
    // All three compute the same bits.  A mode other than 0 stays only if it wins by 1 %.  Knob step_chain: -1 measure,
    // 0 never chain, 1 chain whenever the shape allows (tests); jh_blockop_tune_get/set "step_mode" exports / imports the choice.
    constexpr int CD = 8;                                                 // rows per chunk = rows in flight
    const int64_t span = s_end - s_begin, nchunks = (op->nrow + CD - 1) / CD;
    const bool knobs_free = !c.adj_wg && !c.adj_unroll && !c.adj_depth;
    int cb = 0;                                                           // chained: workgroup size (0: the shape does not allow it)
    if (!direct && parts == 1 && rows_per_launch == op->nrow && nchunks >= 2)
        for (int b : {1024, 512, 256}) {
            if (c.step_chain == 1 && c.adj_wg && c.adj_wg != b) continue;  // (knobs step_chain = 1 + adj_wg: that workgroup size, for sweeps)
            if (span % ((int64_t)b * NS) == 0 && (c.step_chain == 1 || (b == 1024 && span / ((int64_t)b * NS) >= 1024 && op->nrow >= 16))) { cb = b; break; }
        }
    const int64_t ntiles = cb ? span / ((int64_t)cb * NS) : 0;
    if (cb && !(ntiles * nchunks * cb < ((int64_t)1 << 32) && ntiles < ((int64_t)1 << 24))) cb = 0;
    const bool chain_ok = cb != 0 && c.step_chain != 0 && (c.step_chain == 1 || knobs_free);
    const bool remap_ok = parts == 1 && gx % 8 == 0 && gx >= 64 && rows_per_launch == op->nrow;
    int mode = 0, slot = -1;
    if (c.step_coef_dev) {                                                // the graph-captured loop: one plain launch, nothing measured
        if (parts > 1 || rows_per_launch != op->nrow) return jh_fail(JH_ERR_UNSUPPORTED, "one-pass step with device-resident coefficients: the split walk is not supported");
    } else
    if (c.step_chain == 1 && chain_ok) mode = 2;
    else if (op->step_mode >= 0) mode = op->step_mode;
    else if ((remap_ok || chain_ok) && c.autotune && !stream_is_capturing(c.stream) && (op->step_span == 0 || op->step_span == span) &&
             3.0 * (double)op->nrow * (double)span * sizeof(S) >= 1.0 * (double)(1ull << 30)) {
        op->step_span = span;                                             // the trials belong to ONE call shape (whole-vector or one range size)
        mode = lazy_next(op->step_tune, 3, 2, 1, 0.01f, &op->step_mode, &slot);
    }
    if (mode == 2 && !chain_ok) mode = 0;                                 // a trial of a mode this call cannot take runs (and times) the plain walk
    // an expired hand-off poll (never observed; the kernel goes on with an invalid partial sum and raises the sticky error word) must
    // fail the call that CONSUMES w: that is whichever call reads ||u||^2 back -- this one, jh_normsq_read, jh_comm_allreduce_normsq.
    // A call that asks for no norm at all has no such reader, so it never takes the chained walk.
    if (mode == 2 && !normsq && !defer) mode = 0;
    if (mode == 1 && !remap_ok) mode = 0;
    const int remap = mode == 1 ? 1 : 0;
    const bool timing = slot >= 0 && lazy_begin(op->step_tune, slot, c.stream);
    auto trial_done = [&](int st) {
        if (slot >= 0) lazy_end(op->step_tune, slot, c.stream, timing && st == JH_OK);
        return st;
    };
    if (mode == 2) {
        if (c.chain_sync_cap < 2 + ntiles) {
            if (c.chain_sync) { JH_CHECK_HIP(hipStreamSynchronize(c.stream)); JH_CHECK_HIP(hipFree(c.chain_sync)); c.chain_sync = nullptr; c.chain_sync_cap = 0; }
            int64_t cap = 4096;
            while (cap < 2 + ntiles) cap *= 2;
            JH_CHECK_HIP(jh_device_malloc(c.device, (void **)&c.chain_sync, sizeof(unsigned) * (size_t)cap));
            c.chain_sync_cap = cap;
            c.buf_gen++;
        }
        void *wpart = nullptr;
        JH_TRY(jh_ensure_scratch(2 * (size_t)span * sizeof(S), &wpart));
        JH_TRY(jh_ensure_partials(ntiles * nchunks));
        JH_CHECK_HIP(hipMemsetAsync(c.chain_sync, 0, sizeof(unsigned) * (size_t)(2 + ntiles), c.stream));   // ticket counter + flags
        unsigned *err = reinterpret_cast<unsigned *>(c.red_dev + JH_CHAIN_ERR_SLOT);
        // column bands of the chained walk (knob step_band: -1 the default below, 0 none = tiles fastest over the whole row, k tiles per band)
        int64_t cband = c.step_band >= 0 ? c.step_band : 0;
        if (cband >= ntiles) cband = 0;
#define JH_CHAIN(BLK, MIX)                                                                                                \
    hipLaunchKernelGGL((k_tall_diag_bidiag_chain<S, E, NS, 1, CD, BLK, MIX>), dim3((unsigned)(ntiles * nchunks)), dim3(BLK), 0, c.stream, \
                       op->dev_blocks, op->nrow, a_base, a_stride, (S *)u, (const S *)v, (S *)w, n_scalars, (S)alpha, (S)beta,   \
                       c.part_dev, s_begin, s_end, (unsigned)ntiles, (unsigned)nchunks, c.chain_sync, (S *)wpart, err, (unsigned)cband)
        if (mixed) {
            if (cb == 1024) JH_CHAIN(1024, true);
            else if (cb == 512) JH_CHAIN(512, true);
            else JH_CHAIN(256, true);
        } else {
            if (cb == 1024) JH_CHAIN(1024, false);
            else if (cb == 512) JH_CHAIN(512, false);
            else JH_CHAIN(256, false);
        }
#undef JH_CHAIN
        JH_CHECK_HIP(hipGetLastError());
        c.last_step_chain = nchunks;
        c.last_adj_parts = 1;
        double part = 0.0;
        const int st_ = finish_normsq(ntiles * nchunks, normsq ? &part : nullptr, defer);
        if (st_ == JH_OK && normsq) *normsq = part;
        return trial_done(st_);
    }
    c.last_step_chain = 0;
#define JH_LAUNCH(BLK, UU, DD) JH_LAUNCH_M(BLK, UU, DD, false)
#define JH_LAUNCH_T(BLK, UU, DD) JH_LAUNCH_N(BLK, UU, DD, false, false)
#define JH_LAUNCH_M(BLK, UU, DD, MIX) JH_LAUNCH_N(BLK, UU, DD, MIX, true)
#define JH_LAUNCH_N(BLK, UU, DD, MIX, NTV)                                                                              \
    if constexpr (!(E == 2 && sizeof(S) == 4 && BLK == 1024 && UU == 4 && DD == 1))                                      \
    if (wg == BLK && U == UU && D == DD && mixed == MIX) {                                                              \
        double total = 0.0;                                                                                              \
        const bool several = rows_per_launch < op->nrow && normsq != nullptr;   /* one read-back for all the launches */     \
        if (several) JH_CHECK_HIP(hipMemsetAsync(c.red_dev + 9, 0, sizeof(double), c.stream));                           \
        for (int64_t r0 = 0; r0 < op->nrow; r0 += rows_per_launch) {                                                     \
            const int64_t r1 = r0 + rows_per_launch < op->nrow ? r0 + rows_per_launch : op->nrow;                          \
            hipLaunchKernelGGL((k_tall_diag_bidiag<S, E, NS, UU, DD, BLK, MIX, NTV>), dim3((unsigned)gx, (unsigned)parts), dim3(BLK), 0, \
                               c.stream,                                                                                 \
                               op->dev_blocks, op->nrow, a_base, a_stride, (S *)u, (const S *)v, (S *)w, n_scalars,      \
                               direct, (S)alpha, (S)beta, c.part_dev, s_begin, s_end, r0, r1, r0 > 0 ? 1 : 0,              \
                               rows_per_part, (S *)slabs, part_stride, remap, c.step_coef_dev, c.step_done_dev);        \
            JH_CHECK_HIP(hipGetLastError());                                                                             \
            if (parts > 1) JH_TRY((launch_fold_parts<S, NS>(slabs, part_stride, parts, w, s_begin, s_end)));              \
            double part = 0.0;                                                                                           \
            c.last_step_parts = gx * parts;                                                                              \
            if (c.step_coef_dev && c.step_skip_fold) return trial_done(JH_OK);   /* the caller folds part_dev itself */    \
            const int st_ = finish_normsq(gx * parts, normsq ? &part : nullptr, defer, several ? 9 : -1);                \
            if (st_ != JH_OK) return trial_done(st_);                                                                    \
            total += part;                                                                                               \
        }                                                                                                                \
        if (several) {                                                                                                   \
            JH_CHECK_HIP(hipMemcpyAsync(c.red_host + 7, c.red_dev + 9, sizeof(double), hipMemcpyDeviceToHost, c.stream)); \
            JH_CHECK_HIP(hipStreamSynchronize(c.stream));                                                                \
            total = c.red_host[7];                                                                                       \
        }                                                                                                                \
        if (normsq) *normsq = total;                                                                                     \
        return trial_done(JH_OK);                                                                                        \
    }
    // operators whose pass fits the Infinity Cache (jh_stream_nt: knob nt) run the three shapes small blocks select with TEMPORAL loads / stores
    if (!mixed && wg == 256 && !jh_stream_nt(2.0 * (double)op->nrow * (double)n_scalars * sizeof(S))) {
        JH_LAUNCH_T(256, 1, 4) JH_LAUNCH_T(256, 2, 2) JH_LAUNCH_T(256, 4, 1)
    }
    JH_LAUNCH(256, 1, 4) JH_LAUNCH(256, 2, 2) JH_LAUNCH(256, 4, 1) JH_LAUNCH(256, 4, 2) JH_LAUNCH(256, 1, 8)
    JH_LAUNCH(512, 1, 4) JH_LAUNCH(512, 2, 2) JH_LAUNCH(512, 4, 1) JH_LAUNCH(512, 4, 2) JH_LAUNCH(512, 1, 8)
    JH_LAUNCH(1024, 1, 4) JH_LAUNCH(1024, 2, 2) JH_LAUNCH(1024, 4, 1)      // 1024 x 4 x 2 would need > 128 VGPRs per lane
    JH_LAUNCH_M(512, 1, 4, true) JH_LAUNCH_M(256, 2, 2, true) JH_LAUNCH_M(256, 4, 1, true) JH_LAUNCH_M(256, 1, 4, true)
#undef JH_LAUNCH
#undef JH_LAUNCH_T
#undef JH_LAUNCH_M
#undef JH_LAUNCH_N
    return jh_fail(JH_ERR_INVALID, "fused bidiagonalisation step: shape %d x %d x %d is not instantiated", wg, U, D);
}

// ---- per-block loop (operators containing DENSE blocks): the reference's loops (src/Jets.jl:1010-1057)
// with device temporaries -- one child launch (+ one accumulate launch) per non-zero block.
int child_apply(int dtype, const jh_block_desc &b, void *out, const void *in, bool transposed, bool fmode = false)
{
    const bool adj = (b.adjoint != 0) != transposed;          // (op')' = op
    const int64_t n_out = adj ? b.nc : b.nr;
    switch (b.kind) {
    case JH_OP_SQUARE:
        if (fmode && !b.adjoint) return jh_launch_hadamard_raw(out, in, in, dtype, n_out, 0);   // d .= m.^2
        return jh_launch_square_jvp_raw(out, b.coeff, in, dtype, n_out, adj ? 1 : 0);
    case JH_OP_ZERO: return jh_launch_fill_range(out, dtype, n_out, 0.0, 0.0);                  // d .= 0 (942), f! path only
    case JH_OP_DENSE: return jh_launch_gemv(b.coeff, b.nr, b.nc, dtype, out, in, adj ? 1 : 0);
    case JH_OP_DIAG: return jh_launch_hadamard_raw(out, b.coeff, in, dtype, n_out, adj ? 1 : 0);
    case JH_OP_SCALE: {
        const double cre = b.scale_re, cim = adj ? -b.scale_im : b.scale_im;
        const void *xs[1] = {in};
        const int32_t fl = b.scale_flags;
        return jh_launch_lincomb_raw(out, dtype, n_out, 1, &cre, &cim, xs, &fl);
    }
    case JH_OP_IDENTITY:
        if (n_out > 0) JH_CHECK_HIP(hipMemcpyAsync(out, in, (size_t)n_out * jh_dtype_size(dtype), hipMemcpyDeviceToDevice, jh_ctx().stream));
        return JH_OK;
    default: return jh_fail(JH_ERR_INVALID, "child_apply: unexpected block kind %d", b.kind);
    }
}

int accumulate(int dtype, void *acc, const void *term, int64_t n)     // acc .+= term
{
    const double one[2] = {1.0, 1.0}, zero[2] = {0.0, 0.0};
    const void *xs[2] = {acc, term};
    return jh_launch_lincomb_raw(acc, dtype, n, 2, one, zero, xs);
}

// M x K operator of uniform un-adjointed dense children (M, K >= 2): block COLUMN j is a tall operator of dense children, so
// the batched kernels run once per column instead of one child launch per block.  Forward: d_i = ((found + A_i1 m_1) + A_i2 m_2)
// + ... -- column by column through a range-sized temporary, the reference's order (1020-1024); adjoint: m_j = sum_i A_ij' d_i.
int dense_grid_fwd(const jh_blockop *op, void *d, const void *m)
{
    const size_t es = jh_dtype_size(op->dtype);
    const int64_t nr = op->blocks[0].nr, nc = op->blocks[0].nc, nrange = op->row_off[(size_t)op->nrow];
    void *tmp = nullptr;
    JH_TRY(jh_ensure_scratch((size_t)nrange * es + 16, &tmp));
    for (int64_t j = 0; j < op->ncol; j++) {
        JH_TRY(jh_launch_gemv_batched(op->dev_blocks + j * op->nrow, op->nrow, nr, nc, op->dtype, tmp, (const char *)m + (size_t)(j * nc) * es, 0,
                                      op->dense_aligned, false));
        JH_TRY(accumulate(op->dtype, d, tmp, nrange));                                    // _d .+= dtmp   (1024 / 1001)
    }
    return JH_OK;
}

int dense_grid_adj(const jh_blockop *op, void *m, const void *d)
{
    const size_t es = jh_dtype_size(op->dtype);
    const int64_t nr = op->blocks[0].nr, nc = op->blocks[0].nc;
    for (int64_t j = 0; j < op->ncol; j++)
        JH_TRY(jh_launch_gemv_batched(op->dev_blocks + j * op->nrow, op->nrow, nr, nc, op->dtype, (char *)m + (size_t)(j * nc) * es, d, 1,
                                      op->dense_aligned, false));
    return JH_OK;
}

int loop_fwd(const jh_blockop *op, void *d, const void *m, bool fmode = false)   // JetBlock_df! / JetBlock_f!
{
    const size_t es = jh_dtype_size(op->dtype);
    for (int64_t i = 0; i < op->nrow; i++) {                          // (1015)
        char *_d = (char *)d + (size_t)op->row_off[(size_t)i] * es;
        for (int64_t j = 0; j < op->ncol; j++) {                      // (1020)
            const jh_block_desc &b = op->blocks[(size_t)(i + j * op->nrow)];
            if (b.kind == JH_OP_ZERO && !fmode) continue;             // (1022); not in JetBlock_f!
            const char *_m = (const char *)m + (size_t)op->col_off[(size_t)j] * es;
            if (op->ncol > 1) {
                void *dtmp = nullptr;                                 // (1013, 1018)
                JH_TRY(jh_ensure_scratch((size_t)op->row_len[(size_t)i] * es + 16, &dtmp));
                JH_TRY(child_apply(op->dtype, b, dtmp, _m, false, fmode));   // mul!(dtmp, op, _m)
                JH_TRY(accumulate(op->dtype, _d, dtmp, op->row_len[(size_t)i]));   // _d .+= dtmp   (1024 / 1001)
            } else {
                JH_TRY(child_apply(op->dtype, b, _d, _m, false, fmode));     // (1026 / 1003)
            }
        }
    }
    return JH_OK;
}

int loop_adj(const jh_blockop *op, void *m, const void *d)           // JetBlock_df'!
{
    const size_t es = jh_dtype_size(op->dtype);
    for (int64_t j = 0; j < op->ncol; j++) {                          // (1039)
        char *_m = (char *)m + (size_t)op->col_off[(size_t)j] * es;
        if (op->nrow > 1) JH_TRY(jh_launch_fill_range(_m, op->dtype, op->col_len[(size_t)j], 0.0, 0.0));   // _m .= 0  (1042)
        for (int64_t i = 0; i < op->nrow; i++) {                      // (1045)
            const jh_block_desc &b = op->blocks[(size_t)(i + j * op->nrow)];
            if (b.kind == JH_OP_ZERO) continue;                       // (1047)
            const char *_d = (const char *)d + (size_t)op->row_off[(size_t)i] * es;
            if (op->nrow > 1) {
                void *mtmp = nullptr;                                 // (1037, 1043)
                JH_TRY(jh_ensure_scratch((size_t)op->col_len[(size_t)j] * es + 16, &mtmp));
                JH_TRY(child_apply(op->dtype, b, mtmp, _d, true));    // mul!(mtmp, op', _d)
                JH_TRY(accumulate(op->dtype, _m, mtmp, op->col_len[(size_t)j]));   // _m .+= mtmp   (1049)
            } else {
                JH_TRY(child_apply(op->dtype, b, _m, _d, true));      // (1051)
            }
        }
    }
    return JH_OK;
}

// ---- operators that mix BIG dense children with other kinds (round 3; the per-block loop's launch-bound corner) ------------------
// Forward: ONE batched GEMV launch leaves A_ij m_j of every un-adjointed dense child in slab j (jh_dense.hip: k_gemv_rows_mixed, the
// sequential column loop: the bits of the per-child kernel) -- and one more, of the wave-reduction kernel, B' m_j of the ADJOINTED
// ones (block = B'), when there are any -- then ONE launch of the general forward kernel walks every block row in the reference's
// order (1020-1024), `_d .+=` into d as found, taking a dense block's term from its slab: the products and the additions of the
// reference's loop in its order, so bit-identical to the per-block loop wherever that loop's child kernel keeps one column chunk.
// Adjoint: the same with the two kernels' roles swapped (an un-adjointed child needs B' d_i: fp64 wave reduction, rounded like mtmp; an
// adjointed one B d_i: sequential), slab i, and the general adjoint kernel summing every block column in row order (1042-1049).
// Two launches per mul!, three when adjointed and un-adjointed dense children meet.  Exception: when the dense children are few
// AND big (the batched launch would leave the chip empty; the per-child kernel splits a big child's columns / rows over the grid
// instead) they run child by child into the same slabs -- those operators are not launch-bound.
template <typename S, int E>
// fmode (round 4): JetBlock_f! (988-1008) of such an operator -- the dense children's products are the same launches, the combine is
// the general kernel in its f! mode (a zero block's `d .= 0` is added, not skipped; a SQUARE child squares): two launches where the
// per-block loop made two per block
int dense_mixed_apply(const jh_blockop *op, void *out, const void *in, bool transposed, bool fmode = false)
{
    jh_context &c = jh_ctx();
    const size_t es = jh_dtype_size(op->dtype);
    const int64_t nrange = op->row_off[(size_t)op->nrow], ndomain = op->col_off[(size_t)op->ncol];
    const int64_t per16 = (int64_t)(16 / es) > 0 ? (int64_t)(16 / es) : 1;
    const int64_t line_len = transposed ? ndomain : nrange;                     // a slab is laid out like the OUTPUT vector
    const int64_t stride = (line_len + per16 - 1) / per16 * per16;              // elements; slabs stay 16-byte aligned
    const int64_t nslabs = transposed ? op->nrow : op->ncol;
    void *slabs = nullptr;
    JH_TRY(jh_ensure_scratch((size_t)nslabs * (size_t)stride * es + 16, &slabs));
    // which kernel a dense child needs in this direction: block = B (un-adjointed) or B' (adjointed), the operator's adjoint flips it;
    // B x is the sequential rows kernel, B' x the wave-reduction cols kernel
    int64_t launches = 0, ndense = 0, rows_max_out = 0, cols_max_out = 0, wgs = 0;
    double max_bytes = 0.0;
    for (int64_t j = 0; j < op->ncol; j++)
        for (int64_t i = 0; i < op->nrow; i++) {
            const jh_block_desc &b = op->blocks[(size_t)(i + j * op->nrow)];
            if (b.kind != JH_OP_DENSE) continue;
            ndense++;
            const int64_t out_len = transposed ? op->col_len[(size_t)j] : op->row_len[(size_t)i];
            const bool rows_pass = (b.adjoint != 0) == transposed;
            if (rows_pass) { if (out_len > rows_max_out) rows_max_out = out_len; wgs += (out_len * (int64_t)es / 16 + 255) / 256; }
            else { if (out_len > cols_max_out) cols_max_out = out_len; wgs += (out_len + 3) / 4; }
            const double by = (double)b.nr * (double)b.nc * (double)es;
            if (by > max_bytes) max_bytes = by;
        }
    if (ndense && (rows_max_out > 0 || cols_max_out > 0)) {
        if (max_bytes >= (double)(8 << 20) && wgs < 2048) {                     // few BIG children: child by child (column / row split inside); measured
                                                                                // crossover 4-16 MiB per child (profiles/bench_dense_mixed_r03.txt)
            for (int64_t j = 0; j < op->ncol; j++)
                for (int64_t i = 0; i < op->nrow; i++) {
                    const jh_block_desc &b = op->blocks[(size_t)(i + j * op->nrow)];
                    if (b.kind != JH_OP_DENSE) continue;
                    const bool adj = (b.adjoint != 0) != transposed;             // (op')' = op
                    char *o = (char *)slabs + (transposed ? ((size_t)i * (size_t)stride + (size_t)op->col_off[(size_t)j]) : ((size_t)j * (size_t)stride + (size_t)op->row_off[(size_t)i])) * es;
                    const char *x = (const char *)in + (size_t)(transposed ? op->row_off[(size_t)i] : op->col_off[(size_t)j]) * es;
                    JH_TRY(jh_launch_gemv(b.coeff, b.nr, b.nc, op->dtype, o, x, adj ? 1 : 0));
                    launches++;
                }
        } else {
            JH_TRY(jh_launch_gemv_mixed_all(op->dev_blocks, op->nrow, op->ncol, rows_max_out, cols_max_out, op->dtype, slabs, stride, in, transposed ? 1 : 0,
                                            op->dense_mixed_aligned, op->dev_row_off, op->dev_col_off));
            launches += (rows_max_out > 0) + (cols_max_out > 0);
        }
    }
    // the combine: one launch of the general kernel (scalar form: the vectors are small beside the matrices), XCD-aware decode as usual
    const int64_t nlines = transposed ? op->ncol : op->nrow;
    int64_t maxn = 0;
    for (int64_t k = 0; k < nlines; k++) {
        const int64_t len = transposed ? op->col_len[(size_t)k] : op->row_len[(size_t)k];
        maxn = len > maxn ? len : maxn;
    }
    c.last_adj_parts = 1;
    if (maxn > 0) {
        unsigned ntiles, grid;
        int64_t want = (maxn + 255) / 256;
        if (want > 4096) want = 4096;
        general_grid(want, nlines, ntiles, grid, general_use_xcd((transposed ? nrange : ndomain) * (int64_t)es));
        if (!transposed)
            hipLaunchKernelGGL((k_block_fwd_general<S, E>), dim3(grid, 1), dim3(256), 0, c.stream, op->dev_blocks, op->nrow, op->ncol, op->dev_row_off,
                               op->dev_col_off, (const S *)in, (S *)out, fmode ? 1 : 0, ntiles, (int64_t)0, (S *)nullptr, (int64_t)0, (const S *)slabs, stride);
        else
            hipLaunchKernelGGL((k_block_adj_general<S, E>), dim3(grid, 1), dim3(256), 0, c.stream, op->dev_blocks, op->nrow, op->ncol, op->dev_row_off,
                               op->dev_col_off, (S *)out, (const S *)in, ntiles, (int64_t)0, (S *)nullptr, (int64_t)0, (const S *)slabs, stride);
        JH_CHECK_HIP(hipGetLastError());
        launches++;
    }
    c.last_launches = launches;
    return JH_OK;
}

int dense_mixed(const jh_blockop *op, void *out, const void *in, bool transposed, bool fmode = false)
{
    switch (op->dtype) {
    case JH_F32: return dense_mixed_apply<float, 1>(op, out, in, transposed, fmode);
    case JH_F64: return dense_mixed_apply<double, 1>(op, out, in, transposed, fmode);
    case JH_C32: return dense_mixed_apply<float, 2>(op, out, in, transposed, fmode);
    case JH_C64: return dense_mixed_apply<double, 2>(op, out, in, transposed, fmode);
    }
    return jh_fail(JH_ERR_INVALID, "dense_mixed: unknown dtype %d", op->dtype);
}

int loop_small(const jh_blockop *op, void *out, const void *in, int transposed, int fmode)
{
    const int64_t nlines = transposed ? op->ncol : op->nrow;
    const std::vector<int64_t> &lens = transposed ? op->col_len : op->row_len;
    int64_t maxn = 0;
    for (int64_t v : lens) maxn = v > maxn ? v : maxn;
    if (maxn == 0) return JH_OK;
    int64_t gx = (maxn + 255) / 256;
    if (gx > 65535) gx = 65535;
    hipStream_t st = jh_ctx().stream;
#define JH_SMALL(S, E)                                                                                                     \
    hipLaunchKernelGGL((k_block_loop_small<S, E>), dim3((unsigned)gx, (unsigned)nlines), dim3(256), 0, st, op->dev_blocks, op->dev_dims, \
                       op->nrow, op->ncol, op->dev_row_off, op->dev_col_off, (S *)out, (const S *)in, transposed, fmode)
    switch (op->dtype) {
    case JH_F32: JH_SMALL(float, 1); break;
    case JH_F64: JH_SMALL(double, 1); break;
    case JH_C32: JH_SMALL(float, 2); break;
    case JH_C64: JH_SMALL(double, 2); break;
    default: return jh_fail(JH_ERR_INVALID, "loop_small: unknown dtype %d", op->dtype);
    }
#undef JH_SMALL
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

void drop_loop_graphs(const jh_blockop *op)
{
    for (auto &g : op->loop_graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    op->loop_graphs.clear();
}

// The per-block loop is launch-bound for small blocks (one child launch + one accumulate launch per non-zero block, each a
// few microseconds of work).  The first call with a given (output, input) pair runs eagerly (it sizes the scratch and
// partial buffers), the second is captured into a hipGraph, later calls replay it with ONE launch.  A graph holds raw
// pointers, so it is keyed on the vectors' addresses, dropped when the context's scratch buffers move (buf_gen) or the
// operator is re-pointed, and never built while the caller is itself capturing the stream.  Knob: jh_tune_set("graphs", 0).
template <typename F>
int run_loop_graphed(const jh_blockop *op, int mode, const void *out, const void *in, F &&body)
{
    jh_context &c = jh_ctx();
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (!c.graphs || !op->launch_bound || hipStreamIsCapturing(c.stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return body();
    jh_blockop::LoopGraph *g = nullptr;
    for (auto &e : op->loop_graphs)
        if (e.out == out && e.in == in && e.mode == mode) { g = &e; break; }
    if (g && g->exec && g->gen == c.buf_gen) {
        JH_CHECK_HIP(hipGraphLaunch(g->exec, c.stream));
        c.graph_replays++;
        return JH_OK;
    }
    if (g && g->exec) {                                    // stale: the scratch buffers moved since the capture
        (void)hipGraphExecDestroy(g->exec);
        g->exec = nullptr;
        g->seen = 0;
    }
    if (!g) {
        if (op->loop_graphs.size() >= 8) {                 // solvers cycle through a handful of vectors; keep the table small
            if (op->loop_graphs.front().exec) (void)hipGraphExecDestroy(op->loop_graphs.front().exec);
            op->loop_graphs.erase(op->loop_graphs.begin());
        }
        op->loop_graphs.push_back(jh_blockop::LoopGraph{out, in, mode, 0, 0, nullptr});
        g = &op->loop_graphs.back();
    }
    if (g->seen < 1) {
        g->seen++;
        return body();
    }
    const uint64_t gen0 = c.buf_gen;
    if (hipStreamBeginCapture(c.stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
        (void)hipGetLastError();
        return body();
    }
    const int st = body();
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(c.stream, &graph);
    hipGraphExec_t exec = nullptr;
    if (st == JH_OK && e == hipSuccess && graph && c.buf_gen == gen0 && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
        (void)hipGraphDestroy(graph);
        g->exec = exec;
        g->gen = gen0;
        JH_CHECK_HIP(hipGraphLaunch(exec, c.stream));
        c.graph_replays++;
        return JH_OK;
    }
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    g->seen = -1000000;                                    // not capturable (a buffer had to grow, or an error): stay eager
    return body();
}

int check_vectors(const jh_blockop *op, const jh_bvec *rng, const jh_bvec *dom, const char *who)
{
    JH_REQUIRE(op && rng && dom, "%s: null argument", who);
    JH_REQUIRE(rng->dtype == op->dtype && dom->dtype == op->dtype, "%s: dtype mismatch (op %d, range %d, domain %d)", who,
               op->dtype, rng->dtype, dom->dtype);
    JH_REQUIRE(rng->length == op->row_off[(size_t)op->nrow], "%s: range vector has %lld elements, operator range has %lld", who,
               (long long)rng->length, (long long)op->row_off[(size_t)op->nrow]);
    JH_REQUIRE(dom->length == op->col_off[(size_t)op->ncol], "%s: domain vector has %lld elements, operator domain has %lld", who,
               (long long)dom->length, (long long)op->col_off[(size_t)op->ncol]);
    return JH_OK;
}

}  // namespace

bool jh_blockop_tall_fast(const jh_blockop *op, const void *rng_ptr, const void *dom_ptr)
{
    return tall_fast_ok(op, rng_ptr, dom_ptr) || (tall_mixed_ok(op, rng_ptr, dom_ptr) && !(op->nonlinear && !op->pointed));
}

// into how many row ranges the one-pass step over the whole domain cuts this operator (1: one plain launch -- what the graph-replayed
// solver loops of jh_lsqr.hip need, because only the plain launch reads its coefficients from the device)
int64_t jh_bidiag_step_parts(const jh_blockop *op)
{
    if (op->nrow == 1) return 1;
    const int64_t ssize = (int64_t)jh_dtype_size(op->dtype) / (jh_dtype_complex(op->dtype) ? 2 : 1);
    const int64_t n_scalars = op->col_len[0] * (jh_dtype_complex(op->dtype) ? 2 : 1), NS = 16 / ssize;
    const StepShape sh = pick_step_shape(op, n_scalars / NS, op->dtype == JH_C32);
    const int64_t gx = (n_scalars / NS + (int64_t)sh.wg * sh.U - 1) / ((int64_t)sh.wg * sh.U);
    return pick_adj_parts(gx, op->nrow);
}

// all-DIAG tall operators only (the caller checks: jh_lsqr.hip, cg_graph_impl); one pack per lane, 8 rows in flight
int jh_launch_cg_normal(const jh_blockop *op, jh_bvec *p, const jh_bvec *s, jh_bvec *y, const jh_cg_dev *st, double *partials, int64_t *nparts)
{
    jh_context &c = jh_ctx();
    JH_REQUIRE(op->all_diag && tall_fast_ok(op, nullptr, p->data), "cg normal pass: needs a tall all-DIAG operator with equal, 16-byte aligned blocks");
    const int64_t n = op->row_len[0];
    const int64_t packs = (n * (int64_t)jh_dtype_size(op->dtype)) / 16;
    // launch-bound domains: thin workgroups with eight rows in flight; from 2 MiB blocks on the fused normal operator's shapes
    const int shape = packs >= ((int64_t)1 << 22) ? 2 : (packs >= ((int64_t)1 << 17) ? 1 : 0);
    const int64_t per_wg = shape == 2 ? 4096 : (shape == 1 ? 1024 : 256);
    const int64_t grid = (packs + per_wg - 1) / per_wg;
    JH_REQUIRE(grid >= 1 && grid < ((int64_t)1 << 22), "cg normal pass: domain of %lld elements is out of range", (long long)n);
    *nparts = grid;
#define JH_CGN_S(S, E, NS, DEPTH, BLK, U, NTV)                                                                                                  \
    hipLaunchKernelGGL((k_cg_normal<S, E, NS, DEPTH, BLK, U, NTV>), dim3((unsigned)grid), dim3(BLK), 0, c.stream, op->dev_blocks, op->nrow,      \
                       op->diag_strided ? (const S *)op->blocks[0].coeff : (const S *)nullptr, op->diag_stride_elems * E, (S *)p->data,          \
                       (const S *)s->data, (S *)y->data, n * E, st, partials)
    // the coefficients of an operator that an iteration re-reads and that fit the Infinity Cache are loaded TEMPORAL (jh_stream_nt)
    const bool nt = jh_stream_nt(2.0 * (double)op->nrow * (double)n * (double)jh_dtype_size(op->dtype));
#define JH_CGN(S, E, NS)                                                                                                                        \
    do {                                                                                                                                        \
        if (shape == 2) JH_CGN_S(S, E, NS, ((E == 2 && sizeof(S) == 4) ? 2 : 4), 1024, 4, true);                                                \
        else if (shape == 1) { if (nt) JH_CGN_S(S, E, NS, 2, 512, 2, true); else JH_CGN_S(S, E, NS, 2, 512, 2, false); }                        \
        else { if (nt) JH_CGN_S(S, E, NS, 8, 256, 1, true); else JH_CGN_S(S, E, NS, 8, 256, 1, false); }                                        \
    } while (0)
    switch (op->dtype) {
    case JH_F32: JH_CGN(float, 1, 4); break;
    case JH_F64: JH_CGN(double, 1, 2); break;
    case JH_C32: JH_CGN(float, 2, 4); break;
    case JH_C64: JH_CGN(double, 2, 2); break;
    default: return jh_fail(JH_ERR_INVALID, "cg normal pass: unknown dtype %d", op->dtype);
    }
#undef JH_CGN_S
#undef JH_CGN
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

extern "C" {

int jh_blockop_create(int64_t nrow, int64_t ncol, const jh_block_desc *blocks, const int64_t *row_len, const int64_t *col_len,
                      int dtype, jh_blockop **out)
{
    JH_TRY(jh_require_ready());
    JH_REQUIRE(out && blocks && row_len && col_len, "jh_blockop_create: null argument");
    JH_REQUIRE(nrow >= 1 && ncol >= 1, "jh_blockop_create: need at least one block row and column (got %lld x %lld)",
               (long long)nrow, (long long)ncol);
    JH_REQUIRE(jh_dtype_size(dtype) != 0, "jh_blockop_create: unknown dtype %d", dtype);
    jh_blockop *op = new jh_blockop();
    op->ctx = jh_ctx().id;                                   // the coefficient pointers in descs must belong to the current context's device
    op->dtype = dtype;
    op->nrow = nrow;
    op->ncol = ncol;
    op->blocks.assign(blocks, blocks + nrow * ncol);
    op->row_len.assign(row_len, row_len + nrow);
    op->col_len.assign(col_len, col_len + ncol);
    op->row_off.assign((size_t)nrow + 1, 0);
    op->col_off.assign((size_t)ncol + 1, 0);
    for (int64_t i = 0; i < nrow; i++) op->row_off[(size_t)i + 1] = op->row_off[(size_t)i] + row_len[i];
    for (int64_t j = 0; j < ncol; j++) op->col_off[(size_t)j + 1] = op->col_off[(size_t)j] + col_len[j];
    op->tall = (ncol == 1);
    op->uniform_rows = true;
    op->all_diag = true;
    op->elementwise = true;
    int status = JH_OK;
    for (int64_t j = 0; j < ncol && status == JH_OK; j++)
        for (int64_t i = 0; i < nrow && status == JH_OK; i++) {
            const jh_block_desc &b = op->blocks[(size_t)(i + j * nrow)];
            if (row_len[i] != row_len[0]) op->uniform_rows = false;
            const int64_t rl = b.adjoint ? b.nc : b.nr, dl = b.adjoint ? b.nr : b.nc;
            if (row_len[i] < 0 || col_len[j] < 0)
                status = jh_fail(JH_ERR_INVALID, "jh_blockop_create: negative block length");
            else if (rl != row_len[i] || dl != col_len[j])
                status = jh_fail(JH_ERR_INVALID, "jh_blockop_create: block (%lld,%lld) maps %lld -> %lld but its row/column are %lld / %lld",
                                 (long long)i, (long long)j, (long long)dl, (long long)rl, (long long)col_len[j], (long long)row_len[i]);
            else if (b.kind == JH_OP_DENSE) {
                op->elementwise = false;
                op->all_diag = false;
                if (!b.coeff && b.nr * b.nc > 0)
                    status = jh_fail(JH_ERR_INVALID, "jh_blockop_create: DENSE block (%lld,%lld) has no matrix", (long long)i, (long long)j);
            } else if (b.kind == JH_OP_DIAG || b.kind == JH_OP_IDENTITY || b.kind == JH_OP_SCALE || b.kind == JH_OP_SQUARE) {
                if (b.kind == JH_OP_SQUARE) {
                    if (b.adjoint)   // adjoint() takes a JopLn (src/Jets.jl:382-383): a JopNl child has none
                        status = jh_fail(JH_ERR_INVALID, "jh_blockop_create: nonlinear block (%lld,%lld) cannot carry the adjoint flag",
                                         (long long)i, (long long)j);
                    op->nonlinear = true;
                    op->blocks[(size_t)(i + j * nrow)].coeff = nullptr;     // the point arrives through jh_blockop_point
                }
                if (b.nr != b.nc)
                    status = jh_fail(JH_ERR_INVALID, "jh_blockop_create: elementwise block (%lld,%lld) must be square (%lld x %lld)",
                                     (long long)i, (long long)j, (long long)b.nr, (long long)b.nc);
                if (b.kind == JH_OP_DIAG && !b.coeff && b.nr > 0)
                    status = jh_fail(JH_ERR_INVALID, "jh_blockop_create: DIAG block (%lld,%lld) has no coefficients", (long long)i, (long long)j);
                if (b.kind != JH_OP_DIAG) op->all_diag = false;
                if (b.kind == JH_OP_DIAG && b.adjoint && jh_dtype_complex(dtype)) op->all_diag = false;
                if (b.kind == JH_OP_SCALE && (b.scale_im != 0.0 || (b.scale_flags & JH_SCALAR_COMPLEX)) && !jh_dtype_complex(dtype))
                    status = jh_fail(JH_ERR_INVALID, "jh_blockop_create: complex scale on a real operator");
                if (b.kind == JH_OP_SCALE && (b.scale_flags & JH_SCALAR_WIDE) && (dtype == JH_F32 || dtype == JH_C32)) {
                    // a Float64-based scalar against 32-bit elements computes in Float64 (JH_SCALAR_WIDE): the per-block loop's scalar stage
                    // (the typed lincomb) does that; the fused elementwise kernels keep their registers for the element type
                    op->wide_scale = true;
                    op->elementwise = false;
                }
                if (b.kind == JH_OP_SCALE && (b.scale_flags & ~(JH_SCALAR_COMPLEX | JH_SCALAR_WIDE)))
                    status = jh_fail(JH_ERR_INVALID, "jh_blockop_create: unknown scale_flags %d on block (%lld,%lld)", b.scale_flags, (long long)i, (long long)j);
            } else if (b.kind == JH_OP_ZERO) {
                op->all_diag = false;
            } else {
                status = jh_fail(JH_ERR_INVALID, "jh_blockop_create: unknown block kind %d at (%lld,%lld)", b.kind, (long long)i, (long long)j);
            }
        }
    if (status != JH_OK) { delete op; return status; }

    // tall operator of >= 2 un-adjointed dense children of one shape: the batched GEMV kernels (jh_dense.hip) instead of the child loop
    if (op->tall && nrow >= 2 && !op->elementwise) {
        op->dense_batch = true;
        op->dense_aligned = true;
        for (int64_t i = 0; i < nrow && op->dense_batch; i++) {
            const jh_block_desc &b = op->blocks[(size_t)i];
            if (b.kind != JH_OP_DENSE || b.adjoint || b.nr != op->blocks[0].nr || b.nc != op->blocks[0].nc || b.nr == 0 || b.nc == 0) op->dense_batch = false;
            if (((uintptr_t)b.coeff) & 15u) op->dense_aligned = false;
        }
    }

    if (op->tall && nrow >= 2 && !op->elementwise && !op->dense_batch) {       // ... or ragged: same column count, different row counts
        op->dense_batch_ragged = true;
        bool al = true;
        const size_t es = jh_dtype_size(dtype);
        int64_t maxnr = 0;
        for (int64_t i = 0; i < nrow && op->dense_batch_ragged; i++) {
            const jh_block_desc &b = op->blocks[(size_t)i];
            if (b.kind != JH_OP_DENSE || b.adjoint || b.nc != op->blocks[0].nc || b.nr == 0 || b.nc == 0) op->dense_batch_ragged = false;
            if ((((uintptr_t)b.coeff) & 15u) || ((size_t)b.nr * es) % 16 || ((size_t)op->row_off[(size_t)i] * es) % 16) al = false;
            if (b.nr > maxnr) maxnr = b.nr;
        }
        // the ragged kernels write every child's rows directly: the columns must not need splitting (children small or many)
        const double max_bytes = (double)maxnr * (double)op->blocks[0].nc * (double)es;
        if (op->dense_batch_ragged && max_bytes >= (double)(1 << 20) && ((maxnr * (int64_t)es / 16 + 255) / 256) * nrow < 2048) op->dense_batch_ragged = false;
        if (op->dense_batch_ragged) { op->dense_aligned = al; op->dense_max_nr = maxnr; }
    }
    if (nrow >= 2 && ncol >= 2 && !op->elementwise) {                          // ... or a grid of them: one tall batch per block column
        op->dense_batch_grid = true;
        op->dense_aligned = true;
        for (size_t k = 0; k < op->blocks.size() && op->dense_batch_grid; k++) {
            const jh_block_desc &b = op->blocks[k];
            if (b.kind != JH_OP_DENSE || b.adjoint || b.nr != op->blocks[0].nr || b.nc != op->blocks[0].nc || b.nr == 0 || b.nc == 0) op->dense_batch_grid = false;
            if (((uintptr_t)b.coeff) & 15u) op->dense_aligned = false;
        }
    }
    if (nrow == 1 && ncol >= 2 && ncol <= 32768 && !op->elementwise) {         // ... or wide: one block row of such children
        op->dense_batch_wide = true;
        op->dense_aligned = true;
        for (int64_t j = 0; j < ncol && op->dense_batch_wide; j++) {
            const jh_block_desc &b = op->blocks[(size_t)j];
            if (b.kind != JH_OP_DENSE || b.adjoint || b.nr != op->blocks[0].nr || b.nc != op->blocks[0].nc || b.nr == 0 || b.nc == 0) op->dense_batch_wide = false;
            if (((uintptr_t)b.coeff) & 15u) op->dense_aligned = false;
        }
    }

    // anything else with DENSE blocks -- adjointed children, dense next to elementwise kinds (the reference's 3 x 4 test operator),
    // children of differing shapes.  Two routes besides the reference's per-block loop:
    //   small_loop   the whole block loop in ONE launch, a thread forming a dense child's dot product itself (bit-exact both ways; any
    //                adjoint flags): for SMALL operators -- every matrix <= 256 KiB (beyond that the per-child kernels win,
    //                profiles/bench_graphs_r02.txt) and at most 512 sequential products per output element (a 64 x 2 grid of 256^2
    //                children ran its adjoint in 1.0 ms there: 16 384 dependent loads per thread; profiles/bench_dense_mixed_r03.txt)
    //   dense_mixed  one (two, with adjointed AND un-adjointed children) batched launch for all dense children + one combine launch
    //                (dense_mixed_apply): everything else
    if (!op->elementwise && !op->wide_scale && !op->dense_batch && !op->dense_batch_ragged && !op->dense_batch_grid && !op->dense_batch_wide && nrow <= 65535 && ncol <= 65535) {
        const size_t es = jh_dtype_size(dtype);
        bool small = true, eligible = true, aligned = true;
        std::vector<int64_t> fwd_work((size_t)nrow, 0), adj_work((size_t)ncol, 0);
        for (int64_t j = 0; j < ncol; j++)
            for (int64_t i = 0; i < nrow; i++) {
                const jh_block_desc &b = op->blocks[(size_t)(i + j * nrow)];
                if (b.kind != JH_OP_DENSE) continue;
                if ((double)b.nr * (double)b.nc * (double)es > (double)(256 << 10)) small = false;
                if (b.adjoint ? (b.nc != op->row_len[(size_t)i] || b.nr != op->col_len[(size_t)j])                   // block = B': B is col_len x row_len
                              : (b.nr != op->row_len[(size_t)i] || b.nc != op->col_len[(size_t)j])) eligible = false;
                if ((((uintptr_t)b.coeff) & 15u) || ((size_t)b.nr * es) % 16) aligned = false;
                fwd_work[(size_t)i] += b.adjoint ? b.nr : b.nc;
                adj_work[(size_t)j] += b.adjoint ? b.nc : b.nr;
            }
        for (int64_t j = 0; j < ncol; j++)
            if (((size_t)op->col_off[(size_t)j] * es) % 16) aligned = false;                         // (inputs and slab offsets of both directions)
        for (int64_t i = 0; i < nrow; i++)
            if (((size_t)op->row_off[(size_t)i] * es) % 16) aligned = false;
        int64_t line_work = 0;
        for (int64_t v : fwd_work) line_work = v > line_work ? v : line_work;
        for (int64_t v : adj_work) line_work = v > line_work ? v : line_work;
        op->small_loop = small && (line_work <= 512 || !eligible);           // (the one-launch loop is bit-exact in the adjoint too: kept for small operators)
        op->dense_mixed = eligible && !op->small_loop;
        op->dense_mixed_aligned = aligned;
    }

    // strided-diagonal detection: coeff[i] = coeff[0] + i*stride  (e.g. one slab holding all diagonals)
    if (op->tall && op->all_diag && nrow >= 1) {
        op->diag_strided = true;
        const size_t es = jh_dtype_size(dtype);
        if (nrow >= 2) {
            const intptr_t st = (const char *)op->blocks[1].coeff - (const char *)op->blocks[0].coeff;
            if (st <= 0 || (size_t)st % es != 0) op->diag_strided = false;
            for (int64_t i = 2; i < nrow && op->diag_strided; i++)
                if ((const char *)op->blocks[(size_t)i].coeff - (const char *)op->blocks[(size_t)i - 1].coeff != st) op->diag_strided = false;
            if (op->diag_strided) op->diag_stride_elems = (int64_t)((size_t)st / es);
        } else {
            op->diag_stride_elems = 0;
        }
    }

    std::vector<jh_dev_block> host((size_t)(nrow * ncol));
    for (size_t k = 0; k < host.size(); k++) {
        host[k] = jh_dev_block_of(op->blocks[k]);
    }
    hipStream_t st = jh_ctx().stream;
    hipError_t e = jh_device_malloc(jh_ctx().device, (void **)&op->dev_blocks, host.size() * sizeof(jh_dev_block));
    if (e == hipSuccess) e = jh_device_malloc(jh_ctx().device, (void **)&op->dev_row_off, sizeof(int64_t) * ((size_t)nrow + 1));
    if (e == hipSuccess) e = jh_device_malloc(jh_ctx().device, (void **)&op->dev_col_off, sizeof(int64_t) * ((size_t)ncol + 1));
    if (e == hipSuccess) e = hipMemcpyAsync(op->dev_blocks, host.data(), host.size() * sizeof(jh_dev_block), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(op->dev_row_off, op->row_off.data(), sizeof(int64_t) * ((size_t)nrow + 1), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(op->dev_col_off, op->col_off.data(), sizeof(int64_t) * ((size_t)ncol + 1), hipMemcpyHostToDevice, st);
    // which block rows the linear forward writes at all (a row of zero blocks only stays as found, 1022): the split walk's fold needs it
    std::vector<unsigned char> touched((size_t)nrow, 0);
    for (int64_t i = 0; i < nrow; i++)
        for (int64_t j = 0; j < ncol; j++)
            if (op->blocks[(size_t)(i + j * nrow)].kind != JH_OP_ZERO) { touched[(size_t)i] = 1; break; }
    if (e == hipSuccess) e = jh_device_malloc(jh_ctx().device, (void **)&op->dev_row_touched, (size_t)nrow);
    if (e == hipSuccess) e = hipMemcpyAsync(op->dev_row_touched, touched.data(), (size_t)nrow, hipMemcpyHostToDevice, st);
    std::vector<int64_t> dims;
    if (op->small_loop) {
        dims.resize(2 * host.size());
        for (size_t k = 0; k < host.size(); k++) { dims[2 * k] = op->blocks[k].nr; dims[2 * k + 1] = op->blocks[k].nc; }
        if (e == hipSuccess) e = jh_device_malloc(jh_ctx().device, (void **)&op->dev_dims, sizeof(int64_t) * dims.size());
        if (e == hipSuccess) e = hipMemcpyAsync(op->dev_dims, dims.data(), sizeof(int64_t) * dims.size(), hipMemcpyHostToDevice, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);   // host staging vectors die at return
    if (e != hipSuccess) {
        jh_handle_born(op->ctx);                                  // (jh_blockop_destroy counts it out again)
        jh_blockop_destroy(op);
        return jh_fail(JH_ERR_HIP, "jh_blockop_create: %s", hipGetErrorString(e));
    }
    *out = op;
    jh_handle_born(op->ctx);
    // the tall twin of a wide elementwise operator (jh_internal.h): block i = the adjoint of block (0, i).  For a real element type
    // the adjoint of a diagonal / scalar / identity block is the block itself, so the twin stays all-DIAG where the wide one is
    if (nrow == 1 && ncol >= 2 && op->elementwise && !op->nonlinear) {
        std::vector<jh_block_desc> tb(op->blocks);
        bool uniform = true;
        for (auto &b : tb) {
            if (b.kind == JH_OP_SQUARE) uniform = false;
            if (b.nr != tb[0].nr || b.nc != tb[0].nc || b.nr != b.nc) uniform = false;
            if (jh_dtype_complex(dtype)) b.adjoint = b.adjoint ? 0 : 1;
        }
        if (uniform && tb[0].nr > 0 && ((size_t)tb[0].nr * jh_dtype_size(dtype)) % 16 == 0) {
            jh_blockop *tw = nullptr;
            if (jh_blockop_create(ncol, 1, tb.data(), col_len, row_len, dtype, &tw) == JH_OK) op->twin = tw;   // (a failure only costs the fast path)
        }
    }
    return JH_OK;
}

int jh_blockop_destroy(jh_blockop *op)
{
    if (!op) return JH_OK;
    jh_quiesce_scope quiet(op->ctx);                                     // (not jh_enter: a finaliser must not change the thread's current context)
    drop_loop_graphs(op);
    lazy_release(op->fwd_tune);
    lazy_release(op->step_tune);
    if (op->dev_blocks) (void)hipFree(op->dev_blocks);
    if (op->dev_row_off) (void)hipFree(op->dev_row_off);
    if (op->dev_col_off) (void)hipFree(op->dev_col_off);
    if (op->dev_row_touched) (void)hipFree(op->dev_row_touched);
    if (op->dev_dims) (void)hipFree(op->dev_dims);
    if (op->twin) (void)jh_blockop_destroy(op->twin);
    jh_handle_died(op->ctx);
    delete op;
    return JH_OK;
}

int jh_blockop_point(jh_blockop *op, const jh_bvec *mo)
{
    JH_TRY(jh_enter(op, mo));
    JH_REQUIRE(op && mo, "jh_blockop_point: null argument");
    JH_REQUIRE(mo->dtype == op->dtype, "jh_blockop_point: dtype mismatch (op %d, point %d)", op->dtype, mo->dtype);
    JH_REQUIRE(mo->length == op->col_off[(size_t)op->ncol], "jh_blockop_point: point has %lld elements, operator domain has %lld",
               (long long)mo->length, (long long)op->col_off[(size_t)op->ncol]);
    op->pointed = true;
    if (!op->nonlinear) return JH_OK;
    drop_loop_graphs(op);                                               // captured loops hold the old point                                   // linear children ignore the point (upstate! default, 176)
    const size_t es = jh_dtype_size(op->dtype);
    std::vector<jh_dev_block> host(op->blocks.size());
    for (int64_t j = 0; j < op->ncol; j++)                              // (1062)
        for (int64_t i = 0; i < op->nrow; i++) {
            jh_block_desc &b = op->blocks[(size_t)(i + j * op->nrow)];
            if (b.kind == JH_OP_SQUARE) b.coeff = (const char *)mo->data + (size_t)op->col_off[(size_t)j] * es;   // getblock(mo, icol) (1063)
        }
    for (size_t k = 0; k < host.size(); k++) {
        host[k] = jh_dev_block_of(op->blocks[k]);
    }
    hipStream_t st = jh_ctx().stream;
    JH_CHECK_HIP(hipMemcpyAsync(op->dev_blocks, host.data(), host.size() * sizeof(jh_dev_block), hipMemcpyHostToDevice, st));
    JH_CHECK_HIP(hipStreamSynchronize(st));                             // host staging vector dies at return
    return JH_OK;
}

int jh_blockop_f(const jh_blockop *op, jh_bvec *d, const jh_bvec *m)
{
    JH_TRY(jh_enter(op, d, m));
    JH_TRY(check_vectors(op, d, m, "jh_blockop_f"));
    if (op->dense_batch) return jh_launch_gemv_batched(op->dev_blocks, op->nrow, op->blocks[0].nr, op->blocks[0].nc, op->dtype, d->data, m->data, 0, op->dense_aligned, false);
    if (op->dense_batch_wide) return jh_launch_gemv_batched(op->dev_blocks, op->ncol, op->blocks[0].nr, op->blocks[0].nc, op->dtype, d->data, m->data, 0, op->dense_aligned, true);
    if (op->dense_batch_grid) return dense_grid_fwd(op, d->data, m->data);
    if (op->dense_batch_ragged) return jh_launch_gemv_batched(op->dev_blocks, op->nrow, op->dense_max_nr, op->blocks[0].nc, op->dtype, d->data, m->data, 0, op->dense_aligned, false, op->dev_row_off);
    if (op->small_loop && jh_ctx().small_loop) return loop_small(op, d->data, m->data, 0, 1);
    if (op->dense_mixed && jh_ctx().dense_mixed) return dense_mixed(op, d->data, m->data, false, true);
    if (!op->elementwise) return run_loop_graphed(op, 2, d->data, m->data, [&] { return loop_fwd(op, d->data, m->data, true); });
    switch (op->dtype) {
    case JH_F32: return general_fwd<float, 1>(op, d->data, m->data, 1);
    case JH_F64: return general_fwd<double, 1>(op, d->data, m->data, 1);
    case JH_C32: return general_fwd<float, 2>(op, d->data, m->data, 1);
    case JH_C64: return general_fwd<double, 2>(op, d->data, m->data, 1);
    }
    return jh_fail(JH_ERR_INVALID, "jh_blockop_f: unknown dtype %d", op->dtype);
}

int jh_blockop_mul(const jh_blockop *op, jh_bvec *d, const jh_bvec *m)
{
    JH_TRY(jh_enter(op, d, m));
    JH_TRY(check_vectors(op, d, m, "jh_blockop_mul"));
    if (op->nonlinear && !op->pointed)
        return jh_fail(JH_ERR_STATE, "jh_blockop_mul: operator has nonlinear blocks and no linearisation point (jh_blockop_point)");
    if (tall_fast_ok(op, d->data, m->data)) {
        const int64_t n = op->row_len[0];
        switch (op->dtype) {
        case JH_F32: return launch_tall_fwd<float, 1, 4>(op, d->data, m->data, n);
        case JH_F64: return launch_tall_fwd<double, 1, 2>(op, d->data, m->data, n);
        case JH_C32: return launch_tall_fwd<float, 2, 4>(op, d->data, m->data, 2 * n);
        case JH_C64: return launch_tall_fwd<double, 2, 2>(op, d->data, m->data, 2 * n);
        }
    }
    if (tall_mixed_ok(op, d->data, m->data)) {                 // rows of several elementwise kinds: the tall tiling with a per-row kind
        const int64_t n = op->row_len[0];
        switch (op->dtype) {
        case JH_F32: return launch_tall_fwd_mixed<float, 1, 4>(op, d->data, m->data, n);
        case JH_F64: return launch_tall_fwd_mixed<double, 1, 2>(op, d->data, m->data, n);
        case JH_C32: return launch_tall_fwd_mixed<float, 2, 4>(op, d->data, m->data, 2 * n);
        case JH_C64: return launch_tall_fwd_mixed<double, 2, 2>(op, d->data, m->data, 2 * n);
        }
    }
    if (op->dense_batch) return jh_launch_gemv_batched(op->dev_blocks, op->nrow, op->blocks[0].nr, op->blocks[0].nc, op->dtype, d->data, m->data, 0, op->dense_aligned, false);
    if (op->dense_batch_wide) return jh_launch_gemv_batched(op->dev_blocks, op->ncol, op->blocks[0].nr, op->blocks[0].nc, op->dtype, d->data, m->data, 0, op->dense_aligned, true);
    if (op->dense_batch_grid) return dense_grid_fwd(op, d->data, m->data);
    if (op->dense_batch_ragged) return jh_launch_gemv_batched(op->dev_blocks, op->nrow, op->dense_max_nr, op->blocks[0].nc, op->dtype, d->data, m->data, 0, op->dense_aligned, false, op->dev_row_off);
    if (op->small_loop && jh_ctx().small_loop) return loop_small(op, d->data, m->data, 0, 0);
    if (op->dense_mixed && jh_ctx().dense_mixed) return dense_mixed(op, d->data, m->data, false);
    if (!op->elementwise) return run_loop_graphed(op, 0, d->data, m->data, [&] { return loop_fwd(op, d->data, m->data); });
    // a wide operator's forward d = d_found + sum_j A_1j m_j (1024: no zeroing) is its tall twin's ordered adjoint sum started from
    // what d holds -- the same additions in the same order.  Large blocks only: the ordered walk needs >= one workgroup per CU
    // (many small blocks take the general kernel's split walk instead)
    if (op->twin && jh_ctx().wide_twin && jh_ctx().adj_split <= 0 &&
        (jh_ctx().wide_twin == 2 || (size_t)op->row_len[0] * jh_dtype_size(op->dtype) >= ((size_t)16 << 20)) &&
        (tall_fast_ok(op->twin, m->data, d->data) || tall_mixed_ok(op->twin, m->data, d->data))) {
        jh_context &c = jh_ctx();
        c.adj_from_found = 1;
        const int st = jh_blockop_mul_adj(op->twin, d, m);
        c.adj_from_found = 0;
        return st;
    }
    switch (op->dtype) {
    case JH_F32: return general_fwd<float, 1>(op, d->data, m->data);
    case JH_F64: return general_fwd<double, 1>(op, d->data, m->data);
    case JH_C32: return general_fwd<float, 2>(op, d->data, m->data);
    case JH_C64: return general_fwd<double, 2>(op, d->data, m->data);
    }
    return jh_fail(JH_ERR_INVALID, "jh_blockop_mul: unknown dtype %d", op->dtype);
}

int jh_blockop_mul_adj(const jh_blockop *op, jh_bvec *m, const jh_bvec *d)
{
    JH_TRY(jh_enter(op, m, d));
    JH_TRY(check_vectors(op, d, m, "jh_blockop_mul_adj"));
    if (op->nonlinear && !op->pointed)
        return jh_fail(JH_ERR_STATE, "jh_blockop_mul_adj: operator has nonlinear blocks and no linearisation point (jh_blockop_point)");
    if (tall_fast_ok(op, d->data, m->data)) {
        const int64_t n = op->row_len[0];
        switch (op->dtype) {
        case JH_F32: return launch_tall_adj<float, 1, 4, 0>(op, m->data, d->data, n);
        case JH_F64: return launch_tall_adj<double, 1, 2, 0>(op, m->data, d->data, n);
        case JH_C32: return launch_tall_adj<float, 2, 4, 0>(op, m->data, d->data, 2 * n);
        case JH_C64: return launch_tall_adj<double, 2, 2, 0>(op, m->data, d->data, 2 * n);
        }
    }
    if (tall_mixed_ok(op, d->data, m->data)) {
        const int64_t n = op->row_len[0];
        switch (op->dtype) {
        case JH_F32: return launch_tall_adj_mixed<float, 1, 4, 0>(op, m->data, d->data, n);
        case JH_F64: return launch_tall_adj_mixed<double, 1, 2, 0>(op, m->data, d->data, n);
        case JH_C32: return launch_tall_adj_mixed<float, 2, 4, 0>(op, m->data, d->data, 2 * n);
        case JH_C64: return launch_tall_adj_mixed<double, 2, 2, 0>(op, m->data, d->data, 2 * n);
        }
    }
    if (op->dense_batch) return jh_launch_gemv_batched(op->dev_blocks, op->nrow, op->blocks[0].nr, op->blocks[0].nc, op->dtype, m->data, d->data, 1, op->dense_aligned, false);
    if (op->dense_batch_wide) return jh_launch_gemv_batched(op->dev_blocks, op->ncol, op->blocks[0].nr, op->blocks[0].nc, op->dtype, m->data, d->data, 1, op->dense_aligned, true);
    if (op->dense_batch_grid) return dense_grid_adj(op, m->data, d->data);
    if (op->dense_batch_ragged) return jh_launch_gemv_batched(op->dev_blocks, op->nrow, op->dense_max_nr, op->blocks[0].nc, op->dtype, m->data, d->data, 1, op->dense_aligned, false, op->dev_row_off);
    if (op->small_loop && jh_ctx().small_loop) return loop_small(op, m->data, d->data, 1, 0);
    if (op->dense_mixed && jh_ctx().dense_mixed) return dense_mixed(op, m->data, d->data, true);
    if (!op->elementwise) return run_loop_graphed(op, 1, m->data, d->data, [&] { return loop_adj(op, m->data, d->data); });
    // a wide operator's adjoint is its tall twin's forward (same bits: one rounded product per element, zero blocks untouched)
    if (op->twin && jh_ctx().wide_twin && (tall_fast_ok(op->twin, m->data, d->data) || tall_mixed_ok(op->twin, m->data, d->data)))
        return jh_blockop_mul(op->twin, m, d);
    switch (op->dtype) {
    case JH_F32: return general_adj<float, 1>(op, m->data, d->data);
    case JH_F64: return general_adj<double, 1>(op, m->data, d->data);
    case JH_C32: return general_adj<float, 2>(op, m->data, d->data);
    case JH_C64: return general_adj<double, 2>(op, m->data, d->data);
    }
    return jh_fail(JH_ERR_INVALID, "jh_blockop_mul_adj: unknown dtype %d", op->dtype);
}

int jh_blockop_mul_adj_range(const jh_blockop *op, jh_bvec *m, const jh_bvec *d, int64_t first_elem, int64_t count)
{
    JH_TRY(jh_enter(op, m, d));
    JH_TRY(check_vectors(op, d, m, "jh_blockop_mul_adj_range"));
    JH_REQUIRE(first_elem >= 0 && count >= 0 && first_elem + count <= m->length,
               "jh_blockop_mul_adj_range: elements [%lld, %lld) outside the domain vector (%lld elements)", (long long)first_elem,
               (long long)(first_elem + count), (long long)m->length);
    const bool mixed = tall_mixed_ok(op, d->data, m->data);
    if (mixed && op->nonlinear && !op->pointed)
        return jh_fail(JH_ERR_STATE, "jh_blockop_mul_adj_range: operator has nonlinear blocks and no linearisation point (jh_blockop_point)");
    if (!mixed && !tall_fast_ok(op, d->data, m->data))
        return jh_fail(JH_ERR_UNSUPPORTED, "jh_blockop_mul_adj_range: needs a tall operator of elementwise rows with equal, 16-byte aligned blocks");
    const int64_t es = (int64_t)jh_dtype_size(op->dtype);
    JH_REQUIRE((first_elem * es) % 16 == 0 && (count * es) % 16 == 0, "jh_blockop_mul_adj_range: chunk boundaries must be 16-byte aligned");
    const int64_t n = op->row_len[0];
    if (mixed) switch (op->dtype) {
    case JH_F32: return launch_tall_adj_mixed<float, 1, 4, 0>(op, m->data, d->data, n, first_elem, first_elem + count);
    case JH_F64: return launch_tall_adj_mixed<double, 1, 2, 0>(op, m->data, d->data, n, first_elem, first_elem + count);
    case JH_C32: return launch_tall_adj_mixed<float, 2, 4, 0>(op, m->data, d->data, 2 * n, 2 * first_elem, 2 * (first_elem + count));
    case JH_C64: return launch_tall_adj_mixed<double, 2, 2, 0>(op, m->data, d->data, 2 * n, 2 * first_elem, 2 * (first_elem + count));
    }
    switch (op->dtype) {
    case JH_F32: return launch_tall_adj<float, 1, 4, 0>(op, m->data, d->data, n, first_elem, first_elem + count);
    case JH_F64: return launch_tall_adj<double, 1, 2, 0>(op, m->data, d->data, n, first_elem, first_elem + count);
    case JH_C32: return launch_tall_adj<float, 2, 4, 0>(op, m->data, d->data, 2 * n, 2 * first_elem, 2 * (first_elem + count));
    case JH_C64: return launch_tall_adj<double, 2, 2, 0>(op, m->data, d->data, 2 * n, 2 * first_elem, 2 * (first_elem + count));
    }
    return jh_fail(JH_ERR_INVALID, "jh_blockop_mul_adj_range: unknown dtype %d", op->dtype);
}

int jh_blockop_normal_mul(const jh_blockop *op, jh_bvec *y, const jh_bvec *m)
{
    JH_TRY(jh_enter(op, y, m));
    JH_REQUIRE(op && y && m, "jh_blockop_normal_mul: null argument");
    JH_REQUIRE(y->dtype == op->dtype && m->dtype == op->dtype, "jh_blockop_normal_mul: dtype mismatch");
    JH_REQUIRE(y->length == op->col_off[(size_t)op->ncol] && m->length == y->length,
               "jh_blockop_normal_mul: domain vectors have %lld / %lld elements, operator domain has %lld", (long long)y->length,
               (long long)m->length, (long long)op->col_off[(size_t)op->ncol]);
    JH_REQUIRE(y->data != m->data, "jh_blockop_normal_mul: y must not alias m");
    const int64_t n = op->row_len[0];
    if (tall_mixed_ok(op, y->data, m->data)) {                  // rows of several elementwise kinds (a zero row adds nothing: 1022 + 1047)
        if (op->nonlinear && !op->pointed)
            return jh_fail(JH_ERR_STATE, "jh_blockop_normal_mul: operator has nonlinear blocks and no linearisation point (jh_blockop_point)");
        switch (op->dtype) {
        case JH_F32: return launch_tall_adj_mixed<float, 1, 4, 1>(op, y->data, m->data, n);
        case JH_F64: return launch_tall_adj_mixed<double, 1, 2, 1>(op, y->data, m->data, n);
        case JH_C32: return launch_tall_adj_mixed<float, 2, 4, 1>(op, y->data, m->data, 2 * n);
        case JH_C64: return launch_tall_adj_mixed<double, 2, 2, 1>(op, y->data, m->data, 2 * n);
        }
    }
    if (!tall_fast_ok(op, y->data, m->data) || op->nrow < 2)
        return jh_fail(JH_ERR_UNSUPPORTED,
                       "jh_blockop_normal_mul: fused A'A needs a tall (>= 2 rows) operator of elementwise rows with equal, 16-byte aligned blocks; "
                       "chain jh_blockop_mul and jh_blockop_mul_adj instead");
    switch (op->dtype) {
    case JH_F32: return launch_tall_adj<float, 1, 4, 1>(op, y->data, m->data, n);
    case JH_F64: return launch_tall_adj<double, 1, 2, 1>(op, y->data, m->data, n);
    case JH_C32: return launch_tall_adj<float, 2, 4, 1>(op, y->data, m->data, 2 * n);
    case JH_C64: return launch_tall_adj<double, 2, 2, 1>(op, y->data, m->data, 2 * n);
    }
    return jh_fail(JH_ERR_INVALID, "jh_blockop_normal_mul: unknown dtype %d", op->dtype);
}

// The fused A'A over the elements [first_elem, first_elem + count) of the domain: for a host that pipelines the exchange of y range by
// range against the kernels (CG on the normal equations over a row partition).  Elementwise rows: y's range depends on m's range only.
int jh_blockop_normal_mul_range(const jh_blockop *op, jh_bvec *y, const jh_bvec *m, int64_t first_elem, int64_t count)
{
    JH_TRY(jh_enter(op, y, m));
    JH_REQUIRE(op && y && m, "jh_blockop_normal_mul_range: null argument");
    JH_REQUIRE(y->dtype == op->dtype && m->dtype == op->dtype, "jh_blockop_normal_mul_range: dtype mismatch");
    JH_REQUIRE(y->length == op->col_off[(size_t)op->ncol] && m->length == y->length,
               "jh_blockop_normal_mul_range: domain vectors have %lld / %lld elements, operator domain has %lld", (long long)y->length,
               (long long)m->length, (long long)op->col_off[(size_t)op->ncol]);
    JH_REQUIRE(y->data != m->data, "jh_blockop_normal_mul_range: y must not alias m");
    JH_REQUIRE(first_elem >= 0 && count >= 0 && first_elem + count <= y->length,
               "jh_blockop_normal_mul_range: elements [%lld, %lld) outside the domain vector (%lld elements)", (long long)first_elem,
               (long long)(first_elem + count), (long long)y->length);
    const int64_t es = (int64_t)jh_dtype_size(op->dtype);
    JH_REQUIRE((first_elem * es) % 16 == 0 && (count * es) % 16 == 0, "jh_blockop_normal_mul_range: chunk boundaries must be 16-byte aligned");
    const int64_t n = op->row_len[0], lo = first_elem, hi = first_elem + count;
    if (tall_mixed_ok(op, y->data, m->data)) {
        if (op->nonlinear && !op->pointed)
            return jh_fail(JH_ERR_STATE, "jh_blockop_normal_mul_range: operator has nonlinear blocks and no linearisation point (jh_blockop_point)");
        switch (op->dtype) {
        case JH_F32: return launch_tall_adj_mixed<float, 1, 4, 1>(op, y->data, m->data, n, lo, hi);
        case JH_F64: return launch_tall_adj_mixed<double, 1, 2, 1>(op, y->data, m->data, n, lo, hi);
        case JH_C32: return launch_tall_adj_mixed<float, 2, 4, 1>(op, y->data, m->data, 2 * n, 2 * lo, 2 * hi);
        case JH_C64: return launch_tall_adj_mixed<double, 2, 2, 1>(op, y->data, m->data, 2 * n, 2 * lo, 2 * hi);
        }
    }
    if (!tall_fast_ok(op, y->data, m->data) || op->nrow < 2)
        return jh_fail(JH_ERR_UNSUPPORTED,
                       "jh_blockop_normal_mul_range: fused A'A needs a tall (>= 2 rows) operator of elementwise rows with equal, 16-byte aligned blocks");
    switch (op->dtype) {
    case JH_F32: return launch_tall_adj<float, 1, 4, 1>(op, y->data, m->data, n, lo, hi);
    case JH_F64: return launch_tall_adj<double, 1, 2, 1>(op, y->data, m->data, n, lo, hi);
    case JH_C32: return launch_tall_adj<float, 2, 4, 1>(op, y->data, m->data, 2 * n, 2 * lo, 2 * hi);
    case JH_C64: return launch_tall_adj<double, 2, 2, 1>(op, y->data, m->data, 2 * n, 2 * lo, 2 * hi);
    }
    return jh_fail(JH_ERR_INVALID, "jh_blockop_normal_mul_range: unknown dtype %d", op->dtype);
}

}  // extern "C" (templated launch helpers of the fused sum follow)

// *wide (optional): set when a term's scale carries JH_SCALAR_WIDE and the elements are 32-bit -- the launch then takes the WIDE
// instantiation and every scale that is NOT wide goes in as double(Float32(a)) (see k_tall_sum_fwd)
// forward: coef = sign * scale; adjoint: coef = scale, the sign is applied to the term's ordered row sum
static int sum_prepare(int nterms, const jh_blockop *const *ops, const double *scale, const double *sign, const jh_bvec *rng,
                       const jh_bvec *dom, SumArgs &a, bool *strided, bool adjoint, const char *who, const int32_t *flags = nullptr, bool *wide = nullptr)
{
    JH_REQUIRE(ops && scale && sign && rng && dom, "%s: null argument", who);
    JH_REQUIRE(nterms >= 1 && nterms <= JH_SUM_MAX, "%s: %d terms in one group (1..%d)", who, nterms, JH_SUM_MAX);
    a.k = nterms;
    bool all_strided = true;
    for (int t = 0; t < nterms; t++) {
        const jh_blockop *op = ops[t];
        JH_REQUIRE(op, "%s: null operator %d", who, t);
        JH_TRY(check_vectors(op, rng, dom, who));
        if (!tall_fast_ok(op, rng->data, dom->data))
            return jh_fail(JH_ERR_UNSUPPORTED, "%s: term %d is not a tall all-DIAG operator with equal, 16-byte aligned blocks", who, t);
        JH_REQUIRE(op->nrow == ops[0]->nrow && op->row_len[0] == ops[0]->row_len[0] && op->dtype == ops[0]->dtype,
                   "%s: term %d has a different shape or element type", who, t);
        JH_REQUIRE(sign[t] == 1.0 || sign[t] == -1.0, "%s: sign %d must be +1 or -1", who, t);
        if (!(op->diag_strided && (op->nrow == 1 || op->diag_stride_elems == ops[0]->diag_stride_elems) && ops[0]->diag_strided)) all_strided = false;
        if (flags) {
            JH_REQUIRE((flags[t] & ~(JH_SCALAR_COMPLEX | JH_SCALAR_WIDE)) == 0, "%s: unknown scale flags %d on term %d", who, flags[t], t);
            if (flags[t] & JH_SCALAR_COMPLEX) return jh_fail(JH_ERR_UNSUPPORTED, "%s: a Complex scale (term %d) takes the unfused chain", who, t);
        }
    }
    const bool narrow = ops[0]->dtype == JH_F32 || ops[0]->dtype == JH_C32;
    bool any_wide = false;
    if (flags && narrow)
        for (int t = 0; t < nterms; t++) any_wide = any_wide || (flags[t] & JH_SCALAR_WIDE);
    a.stride = all_strided ? ops[0]->diag_stride_elems * (jh_dtype_complex(ops[0]->dtype) ? 2 : 1) : 0;
    for (int t = 0; t < JH_SUM_MAX; t++) {
        const int tt = t < nterms ? t : 0;                                   // a slot beyond k repeats term 0's addresses; its arithmetic is dropped
        a.a0[t] = all_strided ? ops[tt]->blocks[0].coeff : (const void *)ops[tt]->dev_blocks;
        double sc = t < nterms ? scale[t] : 0.0;
        if (t < nterms && any_wide && !(flags[t] & JH_SCALAR_WIDE)) sc = (double)(float)sc;   // T(a), exactly representable: same bits either way
        a.sign[t] = t < nterms ? sign[t] : 1.0;
        a.coef[t] = adjoint ? sc : a.sign[t] * sc;
        a.coef32[t] = (float)a.coef[t];
        a.sign32[t] = (float)a.sign[t];
    }
    *strided = all_strided;
    if (wide) *wide = any_wide;
    return JH_OK;
}

template <typename S, int E, int NS>
static int sum_fwd_launch(const SumArgs &a, bool strided, const jh_blockop *op0, void *d, const void *m, int64_t n_scalars, int accumulate, bool wide = false)
{
    jh_context &c = jh_ctx();
    constexpr int BLK = 256;
    // rows per workgroup, from same-box sweeps (profiles/bench_jetsum_r03.txt): workgroups that move ONE batch and exit stream best --
    // eight coefficient streams: one row of two packs per lane (5.9 TB/s; two rows of one pack 5.3-5.5, four rows 5.0-5.2);
    // up to four streams: two rows
    // round 4: NINE to SIXTEEN streams in one launch (one row of ONE pack per lane: 64 registers of coefficients in flight) -- an 11-term
    // sum as 8 + 3 read and wrote the output twice (4.97 TB/s = 62 % of the roofline over its algorithmic bytes, bench_jetsum_r03.txt)
    const int U = a.k > 8 ? 1 : 2;
    int G = c.fwd_group > 0 ? (int)c.fwd_group : (a.k > 4 ? 1 : 2);                       // rows per workgroup (knob fwd_group: sweeps)
    if (G > op0->nrow) G = (int)op0->nrow;
    const int64_t nvec = n_scalars / NS;
    const int64_t gx = (nvec + (int64_t)BLK * U - 1) / ((int64_t)BLK * U);
    int64_t gy = (op0->nrow + G - 1) / G;
    while (gx * gy * BLK >= ((int64_t)1 << 32) && G < op0->nrow) { G *= 2; gy = (op0->nrow + G - 1) / G; }
    JH_REQUIRE(gx * gy * BLK < ((int64_t)1 << 32), "fused sum forward: grid too large");
#define JH_SUM_FWD(UU, KM, ST, WD)                                                                                                       \
    hipLaunchKernelGGL((k_tall_sum_fwd<S, E, NS, UU, BLK, KM, ST, WD>), dim3((unsigned)(gx * gy)), dim3(BLK), 0, c.stream, a, op0->nrow, G, \
                       (const S *)m, (S *)d, n_scalars, (unsigned)gx, accumulate)
#define JH_SUM_FWD_FEW(UU, KM, ST, WD)                                                                                                   \
    hipLaunchKernelGGL((k_tall_sum_fwd_few<S, E, NS, UU, BLK, KM, 1, ST, WD>), dim3((unsigned)(gx * gy)), dim3(BLK), 0, c.stream, a, op0->nrow, G, \
                       (const S *)m, (S *)d, n_scalars, (unsigned)gx, accumulate)
#define JH_SUM_FWD_K(ST, WD)                                                                                                             \
    do {                                                                                                                                 \
        if (a.k > 12) JH_SUM_FWD(1, 16, ST, WD);                                                                                         \
        else if (a.k > 8) JH_SUM_FWD(1, 12, ST, WD);                                                                                     \
        else if (a.k > 4) JH_SUM_FWD_FEW(2, 8, ST, WD);                                                                                  \
        else JH_SUM_FWD_FEW(2, 4, ST, WD);                                                                                               \
    } while (0)
    bool done = false;
    if constexpr (sizeof(S) == 4) {
        if (wide) {                                                         // a Float64 scale on 32-bit elements: promoted products (k_tall_sum_fwd)
            if (strided) JH_SUM_FWD_K(true, true); else JH_SUM_FWD_K(false, true);
            done = true;
        }
    }
    if (!done) {
        if (strided) JH_SUM_FWD_K(true, false); else JH_SUM_FWD_K(false, false);
    }
#undef JH_SUM_FWD_K
#undef JH_SUM_FWD_FEW
#undef JH_SUM_FWD
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

// many rows of small blocks: every term's row sum through the split walk of the plain adjoint, combined term by term
// (m = sum_k sign_k * scale_k * (A_k' d); tolerance parity)
template <typename S, int E, int NS>
static int sum_adj_split(int nterms, const jh_blockop *const *ops, const double *scale, const double *sign, void *m, const void *d,
                         int64_t n_scalars, void *tmp)
{
    const int dtype = ops[0]->dtype;
    const int64_t n_elems = n_scalars / E;
    for (int t = 0; t < nterms; t++) {
        JH_TRY((launch_tall_adj<S, E, NS, 0>(ops[t], tmp, d, n_scalars)));
        const double cre[2] = {t == 0 ? sign[t] * scale[t] : 1.0, sign[t] * scale[t]}, cim[2] = {0.0, 0.0};
        const void *xs[2] = {t == 0 ? tmp : m, tmp};
        JH_TRY(jh_launch_lincomb_raw(m, dtype, n_elems, t == 0 ? 1 : 2, cre, cim, xs));
    }
    return JH_OK;
}

template <typename S, int E, int NS>
static int sum_adj_launch(const SumArgs &a, bool strided, const jh_blockop *op0, void *m, const void *d, int64_t n_scalars, int accumulate, bool wide = false)
{
    jh_context &c = jh_ctx();
    c.last_adj_parts = 1;
    constexpr int BLK = 256, DEPTH = 2;
    const int U = a.k > 4 ? 1 : 2;
    const int64_t nvec = n_scalars / NS;
    const int64_t gx = (nvec + (int64_t)BLK * U - 1) / ((int64_t)BLK * U);
#define JH_SUM_ADJ_K(UU, DD, KM, ST, WD)                                                                                              \
    hipLaunchKernelGGL((k_tall_sum_adj<S, E, NS, UU, DD, BLK, KM, ST, WD>), dim3((unsigned)gx), dim3(BLK), 0, c.stream, a, op0->nrow, (S *)m, \
                       (const S *)d, n_scalars, accumulate)
#define JH_SUM_ADJ_FEW(UU, DD, KM, ST, WD)                                                                                            \
    hipLaunchKernelGGL((k_tall_sum_adj_few<S, E, NS, UU, DD, BLK, KM, ST, WD>), dim3((unsigned)gx), dim3(BLK), 0, c.stream, a, op0->nrow, (S *)m, \
                       (const S *)d, n_scalars, accumulate)
#define JH_SUM_ADJ_S(ST, WD)                                                                                                          \
    do {                                                                                                                              \
        if (a.k > 12) JH_SUM_ADJ_K(1, 1, 16, ST, WD);      /* sixteen accumulators, one row in flight (knob sum_adj_group = 16) */     \
        else if (a.k > 8) JH_SUM_ADJ_K(1, 1, 12, ST, WD);                                                                             \
        else if (a.k > 4) JH_SUM_ADJ_FEW(1, DEPTH, 8, ST, WD);                                                                        \
        else JH_SUM_ADJ_FEW(2, DEPTH, 4, ST, WD);                                                                                     \
    } while (0)
    bool done = false;
    if constexpr (sizeof(S) == 4) {
        if (wide) {
            if (strided) JH_SUM_ADJ_S(true, true); else JH_SUM_ADJ_S(false, true);
            done = true;
        }
    }
    if (!done) {
        if (strided) JH_SUM_ADJ_S(true, false); else JH_SUM_ADJ_S(false, false);
    }
#undef JH_SUM_ADJ_S
#undef JH_SUM_ADJ_FEW
#undef JH_SUM_ADJ_K
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

extern "C" {

// Any number of terms: groups of JH_SUM_MAX, every group after the first continuing the left-to-right sum from what the output
// holds -- the unfused chain's sequence ((0 +- t1) +- t2) +- ... whatever the grouping.
int jh_blocksum_mul(int nterms, const jh_blockop *const *ops, const double *scale, const double *sign, jh_bvec *d, const jh_bvec *m)
{
    return jh_blocksum_mul_typed(nterms, ops, scale, nullptr, sign, d, m);
}

// scale_flags (nterms x JH_SCALAR_*, or NULL: every scale is taken in the element type): the Julia TYPE of each term's scalar.  A
// Float64 scale against 32-bit elements (JH_SCALAR_WIDE) keeps the sum fused -- the WIDE instantiations of the sum kernels compute that
// term's scalar stage as the promoted product rounded once, the bits of the unfused chain (jh_blockop_mul, jh_lincomb_typed, signed add);
// a Complex scale (JH_SCALAR_COMPLEX) is JH_ERR_UNSUPPORTED: the unfused chain.
int jh_blocksum_mul_typed(int nterms, const jh_blockop *const *ops, const double *scale, const int32_t *scale_flags, const double *sign,
                          jh_bvec *d, const jh_bvec *m)
{
    JH_TRY(jh_enter(d, m));
    JH_REQUIRE(ops && scale && sign && d && m, "jh_blocksum_mul: null argument");
    JH_REQUIRE(nterms >= 1 && nterms <= 4096, "jh_blocksum_mul: %d terms (1..4096 supported)", nterms);
    for (int t = 0; t < nterms; t++) {                                   // validate EVERYTHING before the first launch touches d
        JH_REQUIRE(ops[t], "jh_blocksum_mul: null operator %d", t);
        JH_REQUIRE(ops[t]->ctx == d->ctx, "jh_blocksum_mul: operator %d lives in context %d, the vectors in %d", t, ops[t]->ctx, d->ctx);
        SumArgs probe;
        bool st1 = false;
        JH_TRY(sum_prepare(1, ops + t, scale + t, sign + t, d, m, probe, &st1, false, "jh_blocksum_mul", scale_flags ? scale_flags + t : nullptr));
        JH_REQUIRE(ops[t]->nrow == ops[0]->nrow && ops[t]->row_len[0] == ops[0]->row_len[0] && ops[t]->dtype == ops[0]->dtype,
                   "jh_blocksum_mul: term %d has a different shape or element type", t);
    }
    const int64_t n = ops[0]->row_len[0];
    const int group = jh_ctx().sum_group == 4 ? 4 : (jh_ctx().sum_group == 8 ? 8 : JH_SUM_MAX);   // knob sum_group: 4 / 8 = round 2's / round 3's terms per launch (A/B), 16 (default)
    for (int t0 = 0; t0 < nterms; t0 += group) {
        const int k = nterms - t0 < group ? nterms - t0 : group;
        SumArgs a;
        bool wide = false, strided = false;
        JH_TRY(sum_prepare(k, ops + t0, scale + t0, sign + t0, d, m, a, &strided, false, "jh_blocksum_mul", scale_flags ? scale_flags + t0 : nullptr, &wide));
        const int acc = t0 > 0 ? 1 : 0;
        int st = JH_OK;
        switch (ops[0]->dtype) {
        case JH_F32: st = sum_fwd_launch<float, 1, 4>(a, strided, ops[0], d->data, m->data, n, acc, wide); break;
        case JH_F64: st = sum_fwd_launch<double, 1, 2>(a, strided, ops[0], d->data, m->data, n, acc); break;
        case JH_C32: st = sum_fwd_launch<float, 2, 4>(a, strided, ops[0], d->data, m->data, 2 * n, acc, wide); break;
        case JH_C64: st = sum_fwd_launch<double, 2, 2>(a, strided, ops[0], d->data, m->data, 2 * n, acc); break;
        default: return jh_fail(JH_ERR_INVALID, "jh_blocksum_mul: unknown dtype");
        }
        JH_TRY(st);
    }
    return JH_OK;
}

int jh_blocksum_mul_adj(int nterms, const jh_blockop *const *ops, const double *scale, const double *sign, jh_bvec *m, const jh_bvec *d)
{
    return jh_blocksum_mul_adj_typed(nterms, ops, scale, nullptr, sign, m, d);
}

int jh_blocksum_mul_adj_typed(int nterms, const jh_blockop *const *ops, const double *scale, const int32_t *scale_flags, const double *sign,
                              jh_bvec *m, const jh_bvec *d)
{
    JH_TRY(jh_enter(m, d));
    JH_REQUIRE(ops && scale && sign && d && m, "jh_blocksum_mul_adj: null argument");
    JH_REQUIRE(nterms >= 1 && nterms <= 4096, "jh_blocksum_mul_adj: %d terms (1..4096 supported)", nterms);
    for (int t = 0; t < nterms; t++) {
        JH_REQUIRE(ops[t], "jh_blocksum_mul_adj: null operator %d", t);
        JH_REQUIRE(ops[t]->ctx == d->ctx, "jh_blocksum_mul_adj: operator %d lives in context %d, the vectors in %d", t, ops[t]->ctx, d->ctx);
        SumArgs probe;
        bool st1 = false;
        JH_TRY(sum_prepare(1, ops + t, scale + t, sign + t, d, m, probe, &st1, true, "jh_blocksum_mul_adj", scale_flags ? scale_flags + t : nullptr));
        JH_REQUIRE(ops[t]->nrow == ops[0]->nrow && ops[t]->row_len[0] == ops[0]->row_len[0] && ops[t]->dtype == ops[0]->dtype,
                   "jh_blocksum_mul_adj: term %d has a different shape or element type", t);
    }
    const int64_t n = ops[0]->row_len[0];
    const int group = jh_ctx().sum_group == 4 ? 4 : (jh_ctx().sum_adj_group == 16 ? JH_SUM_MAX : JH_SUM_ADJ_MAX);   // (each term keeps its own accumulator in the adjoint: eight per launch; knob sum_adj_group = 16: sixteen)
    void *tmp = nullptr;
    bool any_wide = false;
    if (scale_flags && (ops[0]->dtype == JH_F32 || ops[0]->dtype == JH_C32))
        for (int t = 0; t < nterms; t++) any_wide = any_wide || (scale_flags[t] & JH_SCALAR_WIDE);
    switch (ops[0]->dtype) {
#define JH_SUM_ADJ(S, E, NS, NSCAL)                                                                         \
    if (!any_wide) JH_TRY((split_adjoint_tmp<S, NS>(ops[0], NSCAL, &tmp)));   /* (a wide scale is applied per d_i before the sum: the ordered walk) */ \
    if (tmp) return sum_adj_split<S, E, NS>(nterms, ops, scale, sign, m->data, d->data, NSCAL, tmp);      \
    for (int t0 = 0; t0 < nterms; t0 += group) {                                                            \
        const int k = nterms - t0 < group ? nterms - t0 : group;                                            \
        SumArgs a;                                                                                          \
        bool wide = false, strided = false;                                                                 \
        JH_TRY(sum_prepare(k, ops + t0, scale + t0, sign + t0, d, m, a, &strided, true, "jh_blocksum_mul_adj", scale_flags ? scale_flags + t0 : nullptr, &wide)); \
        JH_TRY((sum_adj_launch<S, E, NS>(a, strided, ops[0], m->data, d->data, NSCAL, t0 > 0 ? 1 : 0, wide)));  \
    }                                                                                                       \
    return JH_OK;
    case JH_F32: JH_SUM_ADJ(float, 1, 4, n)
    case JH_F64: JH_SUM_ADJ(double, 1, 2, n)
    case JH_C32: JH_SUM_ADJ(float, 2, 4, 2 * n)
    case JH_C64: JH_SUM_ADJ(double, 2, 2, 2 * n)
#undef JH_SUM_ADJ
    }
    return jh_fail(JH_ERR_INVALID, "jh_blocksum_mul_adj: unknown dtype");
}

int jh_blockop_bidiag_step(const jh_blockop *op, jh_bvec *u, const jh_bvec *v, jh_bvec *w, double alpha, double beta, double *normsq)
{
    JH_TRY(jh_enter(op, u, v, w));
    JH_TRY(check_vectors(op, u, v, "jh_blockop_bidiag_step"));
    JH_REQUIRE(w && w->dtype == op->dtype && w->length == v->length, "jh_blockop_bidiag_step: w must be a domain vector of the operator");
    JH_REQUIRE(w->data != v->data, "jh_blockop_bidiag_step: w must not alias v");
    if (!jh_blockop_tall_fast(op, u->data, v->data) || (((uintptr_t)w->data) & 15u))
        return jh_fail(JH_ERR_UNSUPPORTED, "jh_blockop_bidiag_step: needs a tall operator of elementwise rows with equal, 16-byte aligned blocks");
    const int64_t n = op->row_len[0];
    switch (op->dtype) {
    case JH_F32: return launch_bidiag<float, 1, 4>(op, u->data, v->data, w->data, n, alpha, beta, normsq);
    case JH_F64: return launch_bidiag<double, 1, 2>(op, u->data, v->data, w->data, n, alpha, beta, normsq);
    case JH_C32: return launch_bidiag<float, 2, 4>(op, u->data, v->data, w->data, 2 * n, alpha, beta, normsq);
    case JH_C64: return launch_bidiag<double, 2, 2>(op, u->data, v->data, w->data, 2 * n, alpha, beta, normsq);
    }
    return jh_fail(JH_ERR_INVALID, "jh_blockop_bidiag_step: unknown dtype %d", op->dtype);
}

int jh_blockop_bidiag_step_range(const jh_blockop *op, jh_bvec *u, const jh_bvec *v, jh_bvec *w, double alpha, double beta,
                                 int64_t first_elem, int64_t count, double *normsq)
{
    JH_TRY(jh_enter(op, u, v, w));
    JH_TRY(check_vectors(op, u, v, "jh_blockop_bidiag_step_range"));
    JH_REQUIRE(w && w->dtype == op->dtype && w->length == v->length, "jh_blockop_bidiag_step_range: w must be a domain vector of the operator");
    JH_REQUIRE(w->data != v->data, "jh_blockop_bidiag_step_range: w must not alias v");
    JH_REQUIRE(first_elem >= 0 && count >= 0 && first_elem + count <= v->length,
               "jh_blockop_bidiag_step_range: elements [%lld, %lld) outside the domain vector (%lld elements)", (long long)first_elem,
               (long long)(first_elem + count), (long long)v->length);
    if (!jh_blockop_tall_fast(op, u->data, v->data) || (((uintptr_t)w->data) & 15u))
        return jh_fail(JH_ERR_UNSUPPORTED, "jh_blockop_bidiag_step_range: needs a tall operator of elementwise rows with equal, 16-byte aligned blocks");
    const int64_t es = (int64_t)jh_dtype_size(op->dtype);
    JH_REQUIRE((first_elem * es) % 16 == 0 && (count * es) % 16 == 0, "jh_blockop_bidiag_step_range: chunk boundaries must be 16-byte aligned");
    const int64_t n = op->row_len[0], lo = first_elem, hi = first_elem + count;
    switch (op->dtype) {
    case JH_F32: return launch_bidiag<float, 1, 4>(op, u->data, v->data, w->data, n, alpha, beta, normsq, lo, hi, true);
    case JH_F64: return launch_bidiag<double, 1, 2>(op, u->data, v->data, w->data, n, alpha, beta, normsq, lo, hi, true);
    case JH_C32: return launch_bidiag<float, 2, 4>(op, u->data, v->data, w->data, 2 * n, alpha, beta, normsq, 2 * lo, 2 * hi, true);
    case JH_C64: return launch_bidiag<double, 2, 2>(op, u->data, v->data, w->data, 2 * n, alpha, beta, normsq, 2 * lo, 2 * hi, true);
    }
    return jh_fail(JH_ERR_INVALID, "jh_blockop_bidiag_step_range: unknown dtype %d", op->dtype);
}

int jh_blockop_tune_get(const jh_blockop *op, const char *name, int64_t *value)
{
    JH_REQUIRE(op && name && value, "jh_blockop_tune_get: null argument");
    if (!strcmp(name, "fwd_walk")) *value = op->fwd_walk;                       // -1: not chosen yet
    else if (!strcmp(name, "fwd_walk_inherited")) *value = op->walk_inherited ? 1 : 0;   // the choice came from an earlier operator of the same shape
    else if (!strcmp(name, "fwd_trials")) *value = op->fwd_tune.launched;
    else if (!strcmp(name, "fwd_switches")) *value = op->fwd_tune.switches;       // times the periodic re-check rotated another walk in
    else if (!strcmp(name, "fwd_playoff")) *value = op->fwd_tune.playoff[0] >= 0 ? op->fwd_tune.playoff[0] * 16 + op->fwd_tune.playoff[1] : -1;
    else if (!strcmp(name, "step_trials")) *value = op->step_tune.launched;
    else if (!strcmp(name, "upd_walk")) *value = op->upd_walk;
    else if (!strcmp(name, "step_mode")) *value = op->step_mode;
    else return jh_fail(JH_ERR_INVALID, "jh_blockop_tune_get: unknown per-operator knob '%s'", name);
    return JH_OK;
}

int jh_blockop_tune_set(jh_blockop *op, const char *name, int64_t value)
{
    JH_REQUIRE(op && name, "jh_blockop_tune_set: null argument");
    if (!strcmp(name, "fwd_walk")) {
        JH_REQUIRE(value >= -1 && value < K_FWD_CANDIDATES, "jh_blockop_tune_set: fwd_walk must be -1 (measure again) or 0..%d", K_FWD_CANDIDATES - 1);
        lazy_reset(op->fwd_tune);
        op->fwd_walk = (int)value;
        op->walk_measure_again = value < 0;                                 // -1: THIS operator measures, whatever operators of its shape found before
        op->walk_inherited = false;
    } else if (!strcmp(name, "upd_walk")) {
        JH_REQUIRE(value >= -1 && value <= 1, "jh_blockop_tune_set: upd_walk must be -1, 0 or 1");
        op->upd_walk = (int)value;
        op->upd_trials = value < 0 ? 0 : 2;
    } else if (!strcmp(name, "step_mode")) {
        JH_REQUIRE(value >= -1 && value <= 2, "jh_blockop_tune_set: step_mode must be -1 (measure), 0 (plain walk), 1 (XCD-contiguous tiles) or 2 (chained row chunks)");
        lazy_reset(op->step_tune);
        op->step_span = 0;
        op->step_mode = (int)value;
    } else return jh_fail(JH_ERR_INVALID, "jh_blockop_tune_set: unknown per-operator knob '%s'", name);
    return JH_OK;
}

int jh_normsq_reset(void)
{
    JH_TRY(jh_require_ready());
    jh_context &c = jh_ctx();
    JH_CHECK_HIP(hipMemsetAsync(c.red_dev + JH_NORMSQ_SLOT, 0, sizeof(double), c.stream));
    return JH_OK;
}

int jh_normsq_read(double *out)
{
    JH_TRY(jh_require_ready());
    JH_REQUIRE(out, "jh_normsq_read: null output");
    jh_context &c = jh_ctx();
    JH_CHECK_HIP(hipMemcpyAsync(c.red_host + 6, c.red_dev + JH_NORMSQ_SLOT, sizeof(double), hipMemcpyDeviceToHost, c.stream));
    JH_CHECK_HIP(hipMemcpyAsync(c.red_host + 3, c.red_dev + JH_CHAIN_ERR_SLOT, sizeof(double), hipMemcpyDeviceToHost, c.stream));
    JH_CHECK_HIP(hipStreamSynchronize(c.stream));
    *out = c.red_host[6];
    return jh_chain_err_check();
}

int jh_blockop_mul_axpby(const jh_blockop *op, jh_bvec *d, const jh_bvec *m, double alpha, double beta, double *normsq)
{
    JH_TRY(jh_enter(op, d, m));
    JH_TRY(check_vectors(op, d, m, "jh_blockop_mul_axpby"));
    if (!jh_blockop_tall_fast(op, d->data, m->data))
        return jh_fail(JH_ERR_UNSUPPORTED, "jh_blockop_mul_axpby: needs a tall operator of elementwise rows with equal, 16-byte aligned blocks; "
                                           "use jh_blockop_mul into a temporary, jh_lincomb and jh_norm instead");
    const int64_t n = op->row_len[0];
    switch (op->dtype) {
    case JH_F32: return launch_fwd_update<float, 1, 4>(op, d->data, m->data, n, alpha, beta, normsq);
    case JH_F64: return launch_fwd_update<double, 1, 2>(op, d->data, m->data, n, alpha, beta, normsq);
    case JH_C32: return launch_fwd_update<float, 2, 4>(op, d->data, m->data, 2 * n, alpha, beta, normsq);
    case JH_C64: return launch_fwd_update<double, 2, 2>(op, d->data, m->data, 2 * n, alpha, beta, normsq);
    }
    return jh_fail(JH_ERR_INVALID, "jh_blockop_mul_axpby: unknown dtype %d", op->dtype);
}

int jh_blockop_mul_adj_axpby(const jh_blockop *op, jh_bvec *m, const jh_bvec *d, double alpha, double beta, double in_scale,
                             double *normsq)
{
    JH_TRY(jh_enter(op, m, d));
    JH_TRY(check_vectors(op, d, m, "jh_blockop_mul_adj_axpby"));
    if (!tall_fast_ok(op, d->data, m->data))
        return jh_fail(JH_ERR_UNSUPPORTED, "jh_blockop_mul_adj_axpby: needs a tall all-DIAG operator with equal, 16-byte aligned blocks; "
                                           "use jh_blockop_mul_adj into a temporary, jh_lincomb and jh_norm instead");
    const int64_t n = op->row_len[0];
    switch (op->dtype) {
    case JH_F32: return launch_adj_update<float, 1, 4>(op, m->data, d->data, n, alpha, beta, in_scale, normsq);
    case JH_F64: return launch_adj_update<double, 1, 2>(op, m->data, d->data, n, alpha, beta, in_scale, normsq);
    case JH_C32: return launch_adj_update<float, 2, 4>(op, m->data, d->data, 2 * n, alpha, beta, in_scale, normsq);
    case JH_C64: return launch_adj_update<double, 2, 2>(op, m->data, d->data, 2 * n, alpha, beta, in_scale, normsq);
    }
    return jh_fail(JH_ERR_INVALID, "jh_blockop_mul_adj_axpby: unknown dtype %d", op->dtype);
}

// (a * A) m and (a * A)' d = A'(conj(a) d) of the scalar-times-operator chain (src/Jets.jl:1159-1164) in one pass each, for a REAL scalar of
// any Julia type: jh_blockop_mul_axpby(alpha = a, beta = 0) / jh_blockop_mul_adj_axpby(in_scale = a) when a is taken in the element type,
// the WIDE instantiations of the same kernels (Float64 product, one rounding) when a is Float64-based and the elements are 32-bit
int jh_blockop_mul_scaled(const jh_blockop *op, jh_bvec *d, const jh_bvec *m, double a, int a_flags)
{
    JH_REQUIRE((a_flags & ~(JH_SCALAR_COMPLEX | JH_SCALAR_WIDE)) == 0, "jh_blockop_mul_scaled: unknown flags %d", a_flags);
    if (a_flags & JH_SCALAR_COMPLEX) return jh_fail(JH_ERR_UNSUPPORTED, "jh_blockop_mul_scaled: a Complex scalar takes the unfused chain (jh_blockop_mul, jh_lincomb_typed)");
    JH_TRY(jh_enter(op, d, m));
    JH_TRY(check_vectors(op, d, m, "jh_blockop_mul_scaled"));
    if (!jh_blockop_tall_fast(op, d->data, m->data))
        return jh_fail(JH_ERR_UNSUPPORTED, "jh_blockop_mul_scaled: needs a tall operator of elementwise rows with equal, 16-byte aligned blocks");
    const int64_t n = op->row_len[0];
    const bool wide = (a_flags & JH_SCALAR_WIDE) != 0;
    switch (op->dtype) {
    case JH_F32: return launch_fwd_update<float, 1, 4>(op, d->data, m->data, n, a, 0.0, nullptr, wide);
    case JH_F64: return launch_fwd_update<double, 1, 2>(op, d->data, m->data, n, a, 0.0, nullptr);
    case JH_C32: return launch_fwd_update<float, 2, 4>(op, d->data, m->data, 2 * n, a, 0.0, nullptr, wide);
    case JH_C64: return launch_fwd_update<double, 2, 2>(op, d->data, m->data, 2 * n, a, 0.0, nullptr);
    }
    return jh_fail(JH_ERR_INVALID, "jh_blockop_mul_scaled: unknown dtype %d", op->dtype);
}

int jh_blockop_mul_adj_scaled(const jh_blockop *op, jh_bvec *m, const jh_bvec *d, double a, int a_flags)
{
    JH_REQUIRE((a_flags & ~(JH_SCALAR_COMPLEX | JH_SCALAR_WIDE)) == 0, "jh_blockop_mul_adj_scaled: unknown flags %d", a_flags);
    if (a_flags & JH_SCALAR_COMPLEX) return jh_fail(JH_ERR_UNSUPPORTED, "jh_blockop_mul_adj_scaled: a Complex scalar takes the unfused chain (jh_lincomb_typed, jh_blockop_mul_adj)");
    JH_TRY(jh_enter(op, m, d));
    JH_TRY(check_vectors(op, d, m, "jh_blockop_mul_adj_scaled"));
    if (!tall_fast_ok(op, d->data, m->data))
        return jh_fail(JH_ERR_UNSUPPORTED, "jh_blockop_mul_adj_scaled: needs a tall all-DIAG operator with equal, 16-byte aligned blocks");
    const int64_t n = op->row_len[0];
    const bool wide = (a_flags & JH_SCALAR_WIDE) != 0;
    switch (op->dtype) {
    case JH_F32: return launch_adj_update<float, 1, 4>(op, m->data, d->data, n, 1.0, 0.0, a, nullptr, wide);
    case JH_F64: return launch_adj_update<double, 1, 2>(op, m->data, d->data, n, 1.0, 0.0, a, nullptr);
    case JH_C32: return launch_adj_update<float, 2, 4>(op, m->data, d->data, 2 * n, 1.0, 0.0, a, nullptr, wide);
    case JH_C64: return launch_adj_update<double, 2, 2>(op, m->data, d->data, 2 * n, 1.0, 0.0, a, nullptr);
    }
    return jh_fail(JH_ERR_INVALID, "jh_blockop_mul_adj_scaled: unknown dtype %d", op->dtype);
}

}  // extern "C"
