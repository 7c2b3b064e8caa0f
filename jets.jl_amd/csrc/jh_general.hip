// jh_general.hip -- the general path of JetBlock_df! / JetBlock_df'! / JetBlock_f! (src/Jets.jl:988-1057): any nrow x ncol mix of ZERO /
// IDENTITY / SCALE / DIAG / SQUARE blocks with ragged block lengths (k_block_*_general[_vec]: one thread per element walks a block row or
// column in the reference's loop order with the same rounding sequence), grids of equal blocks register-tiled (k_grid_diag, k_grid_tile,
// k_general_tile), the one-launch loop for small dense children, the split walk's fold, and the per-block loops / batched routes of
// operators with dense children.  One of the translation units jh_blockop.hip was split into in round 5 (jh_blockop_common.h).
#include "jh_blockop_common.h"

namespace {

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
                                    const S *__restrict__ dense_prod = nullptr, const int *__restrict__ steps = nullptr, int step_stride = 0)
{
    // dense_prod != null (dense_mixed_fwd): block (i, j) of kind DENSE contributes the product A_ij m_j a batched GEMV launch has
    // left, rounded like the reference's dtmp (1024), at the place its table entry names (jh_dev_block_prod_off: a compact scratch vector,
    // one piece per dense child -- late round 5; rounds 3-4 kept ncol slabs of the whole range vector's length)
    // steps != null (late round 5; never with the split walk or in f! mode): the row's non-zero blocks from its step list (jh_blockop_create) -- a
    // block-diagonal operator of 1024 x 1024 blocks walks one table entry per row instead of 1024
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
        const int *sidx = steps ? steps + i * step_stride + 1 : nullptr;
        if (sidx) j_hi = sidx[-1];
        for (int64_t jj = j_lo; jj < j_hi; jj++) {                         // (1020)
            const int64_t j = sidx ? (int64_t)sidx[jj] : jj;
            const jh_dev_block b = blocks[i + j * nrow];
            if (b.kind == JH_OP_ZERO && !fmode) continue;                  // (1022); JetBlock_f! has no such test
            elem<S, E> p;
            p.re = 0; p.im = 0;                                            // a zero block's `d .= 0` (942): no load -- its column may be shorter than this row
            if (b.kind == JH_OP_DENSE) {
                p = eload<S, E>(dense_prod, jh_dev_block_prod_off(b, false) + e);   // mul!(dtmp, op, _m), computed by the batched launch
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
                                    const S *__restrict__ dense_prod = nullptr, const int *__restrict__ steps = nullptr, int step_stride = 0)
{
    // dense_prod != null (dense_mixed_adj): block (i, j) of kind DENSE contributes A_ij' d_i, left by the batched launch, rounded
    // like the reference's mtmp (1049), at the place its table entry names; steps: the column's non-zero blocks (see the forward)
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
        const int *sidx = steps ? steps + j * step_stride + 1 : nullptr;
        if (sidx) i_hi = sidx[-1];
        for (int64_t ii = i_lo; ii < i_hi; ii++) {                         // (1045)
            const int64_t i = sidx ? (int64_t)sidx[ii] : ii;
            const jh_dev_block b = blocks[i + j * nrow];
            if (b.kind == JH_OP_ZERO) continue;                            // (1047)
            elem<S, E> p;
            if (b.kind == JH_OP_DENSE) {
                p = eload<S, E>(dense_prod, jh_dev_block_prod_off(b, true) + e);   // mul!(mtmp, op', _d), computed by the batched launch
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
                                        int64_t q_per_part, S *__restrict__ slabs, int64_t slab_stride,
                                        const int *__restrict__ steps = nullptr, int step_stride = 0)
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
    // steps != null (late round 5; never with the split walk or in f! mode): walk the row's step list -- its non-zero blocks, ascending -- instead of
    // every table entry (a ragged block-diagonal grid of 64 x 64 blocks ran at 1.4 TB/s on its 63 zero blocks per row)
    const int *sidx = steps ? steps + i * step_stride + 1 : nullptr;
    if (sidx) j_hi = sidx[-1];
#define JH_COL(jj) (sidx ? (int64_t)sidx[jj] : (int64_t)(jj))
    const int64_t ns = (row_off[i + 1] - row_off[i]) * E;                 // scalars in this block row
    // (round 5, session 3) lines need not be whole, 16-byte aligned packs: under-aligned accesses, a line's last pack loaded from ns - NS and stored from its
    // own first scalar on (jh_blockop_common.h: ldu / pack_start / st_pack; general_vec_ok)
    for (int64_t s = (tile * 256 + threadIdx.x) * NS; s < ns; s += (int64_t)ntiles * 256 * NS) {
        const int64_t sc = pack_start<NS>(s, ns);
        V acc = (V)(S)0;
        bool touched = split;
        if (ncol > 1 && !split) acc = ldu<false, S, NS>(d + row_off[i] * E + sc);
        // the block table and the column offsets are read ONE GROUP AHEAD (scalar loads): a block's vector loads need them, and
        // waiting for them block by block serialises two latencies per block (the mixed one-pass step lost 12 % to that)
        jh_dev_block nb[GENERAL_Q];
        int64_t noff[GENERAL_Q];
#pragma unroll
        for (int q = 0; q < GENERAL_Q; q++)
            if (j_lo + q < j_hi) { const int64_t jc = JH_COL(j_lo + q); nb[q] = blocks[i + jc * nrow]; noff[q] = col_off[jc]; }
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
                if (jn < j_hi) { const int64_t jc = JH_COL(jn); nb[q] = blocks[i + jc * nrow]; noff[q] = col_off[jc]; }
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
                        x[q] = ldu<false, S, NS>(m + off[q] * E + sc);
                        if (block_reads_coeff(b[q], fmode != 0)) c[q] = ldu<true, S, NS>((const S *)b[q].coeff + sc);   // streamed once
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
            if (split) st_pack<false, S, NS>(d + row_off[i] * E, s, sc, acc);                    // a slab: the fold reads it back
            else st_pack<true, S, NS>(d + row_off[i] * E, s, sc, acc);                           // the result row: written once
        }
    }
}

#undef JH_COL

template <typename S, int E, int NS>
__global__ void k_block_adj_general_vec(const jh_dev_block *__restrict__ blocks, int64_t nrow, int64_t ncol,
                                        const int64_t *__restrict__ row_off, const int64_t *__restrict__ col_off,
                                        S *__restrict__ m, const S *__restrict__ d, unsigned ntiles,
                                        int64_t q_per_part, S *__restrict__ slabs, int64_t slab_stride, int nt_out,
                                        const int *__restrict__ steps = nullptr, int step_stride = 0)
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
    const int *sidx = steps ? steps + j * step_stride + 1 : nullptr;       // the column's non-zero blocks (see the forward)
    if (sidx) i_hi = sidx[-1];
#define JH_ROW(ii) (sidx ? (int64_t)sidx[ii] : (int64_t)(ii))
    const int64_t ns = (col_off[j + 1] - col_off[j]) * E;
    for (int64_t s = (tile * 256 + threadIdx.x) * NS; s < ns; s += (int64_t)ntiles * 256 * NS) {
        const int64_t sc = pack_start<NS>(s, ns);                             // (lines off the pack grid: see the forward)
        V acc = (V)(S)0;
        bool touched = (nrow > 1);
        jh_dev_block nb[GENERAL_Q];                                        // block table and row offsets one group ahead (see the forward)
        int64_t noff[GENERAL_Q];
#pragma unroll
        for (int q = 0; q < GENERAL_Q; q++)
            if (i_lo + q < i_hi) { const int64_t ir = JH_ROW(i_lo + q); nb[q] = blocks[ir + j * nrow]; noff[q] = row_off[ir]; }
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
                if (in < i_hi) { const int64_t ir = JH_ROW(in); nb[q] = blocks[ir + j * nrow]; noff[q] = row_off[ir]; }
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
                        x[q] = ldu<false, S, NS>(d + off[q] * E + sc);
                        if (block_reads_coeff(b[q], false)) c[q] = ldu<true, S, NS>((const S *)b[q].coeff + sc);   // streamed once
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
            if (nt_out) st_pack<true, S, NS>(m + col_off[j] * E, s, sc, acc);           // a large result written once (a wide operator's adjoint)
            else st_pack<false, S, NS>(m + col_off[j] * E, s, sc, acc);                 // a slab of the split walk / a small domain vector: read again soon
        }
    }
}

#undef JH_ROW

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
        int64_t s[U], sn[U];                                                 // sn: where the pack nominally starts; s: where it is loaded from (blocks need not be
        bool ok[U];                                                         // whole, 16-byte aligned packs: ldu / pack_start / st_pack; round 5, last session)
#pragma unroll
        for (int u = 0; u < U; u++) {                                       // U packs per lane, 256 lanes apart (clamped: branch-free loads)
            ok[u] = s0 + (int64_t)u * 256 * NS < n_scalars;
            sn[u] = ok[u] ? s0 + (int64_t)u * 256 * NS : s0;
            s[u] = pack_start<NS>(sn[u], n_scalars);
        }
        V acc[R][U];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int u = 0; u < U; u++)
                acc[r][u] = TRANSPOSED ? (V)(S)0 : ldu<true, S, NS>(out + line[r] * n_scalars + s[u]);   // `_m .= 0` (1042) / d as found (1024)
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
                    x[q][u] = ldu<false, S, NS>(in + (q0 + q) * n_scalars + s[u]);          // shared by every line group: through the caches
#pragma unroll
                    for (int r = 0; r < R; r++) c[q][r][u] = ldu<true, S, NS>(a[q][r] + s[u]);   // streamed once
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
                const V x = ldu<false, S, NS>(in + q * n_scalars + s[u]);
                V c[R];
#pragma unroll
                for (int r = 0; r < R; r++) c[r] = ldu<true, S, NS>(aq[r] + s[u]);
#pragma unroll
                for (int r = 0; r < R; r++) acc[r][u] = acc[r][u] + vmul<S, E, NS, V>(c[r], x, TRANSPOSED);
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int u = 0; u < U; u++)
                if (ok[u] && grp * R + r < nlines) st_pack<true, S, NS>(out + line[r] * n_scalars, sn[u], s[u], acc[r][u]);
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
//
// LIST (round 5, late): the steps of a line group come from a list built at create (jh_blockop.hip: build_step_lists) -- the summed
// block indices at which ANY line of the group has a non-zero block, ascending -- instead of 0 ... nsum - 1.  A step that is left out
// had only zero blocks, which contribute nothing and are skipped by the reference too (1022 / 1047): same terms, same order, same bits;
// what goes away is the step's input-pack load (HBM / L2 bytes nobody needs), its dummy loads and its branches.  Block-diagonal,
// banded and arrow-shaped operators walk a handful of steps per line group instead of a whole block row / column.
// Layout: one record of `step_stride` ints per line group -- [count, index 0, index 1, ..., two padding entries] -- so that the record's address does
// not depend on a loaded value (count and the first indices arrive with one scalar load); the indices are fetched TWO steps ahead and the table
// entries one step ahead, so no step waits for an index -> table entry -> coefficient chain of dependent round trips.
template <typename S, int E, int NS, int QQ, int U, bool TRANSPOSED, int R = 2, bool LIST = false>
__global__ __launch_bounds__(256) void k_general_tile(const jh_dev_block *__restrict__ blocks, int64_t nrow, int64_t ncol, int64_t n_scalars,
                                                      const S *__restrict__ in, S *__restrict__ out, unsigned ntiles, unsigned ngroups,
                                                      const int *__restrict__ steps, int step_stride)
{
    typedef typename vec_of<S, NS>::type V;
    int64_t grp, tile;
    general_line_tile(ngroups, ntiles, grp, tile);
    ntiles &= 0x0fffffffu;
    if (tile >= ntiles) return;
    const int64_t nlines = TRANSPOSED ? ncol : nrow;
    const int *sidx = LIST ? steps + grp * step_stride + 1 : nullptr;
    const int64_t nsum = LIST ? (int64_t)sidx[-1] : (TRANSPOSED ? nrow : ncol);        // LIST: the group's own step count
    const int64_t qstep = TRANSPOSED ? 1 : nrow, lstep = TRANSPOSED ? nrow : 1;   // block (line l, q) = blocks[l * lstep + q * qstep]
    int64_t line[R];
#pragma unroll
    for (int r = 0; r < R; r++) line[r] = grp * R + r < nlines ? grp * R + r : nlines - 1;
    for (int64_t s0 = (tile * 256 * U + threadIdx.x) * NS; s0 < n_scalars; s0 += (int64_t)ntiles * 256 * U * NS) {
        int64_t s[U];
        bool ok[U];
        int64_t sn[U];                                                      // (round 5, session 3) sn: where the pack nominally starts; s: where it is loaded from --
#pragma unroll                                                              // blocks need not be whole, 16-byte aligned packs (ldu / pack_start / st_pack)
        for (int u = 0; u < U; u++) {                                       // U packs per lane, 256 lanes apart (clamped: branch-free loads)
            ok[u] = s0 + (int64_t)u * 256 * NS < n_scalars;
            sn[u] = ok[u] ? s0 + (int64_t)u * 256 * NS : s0;
            s[u] = pack_start<NS>(sn[u], n_scalars);
        }
        V acc[R][U];
        bool touched[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
#pragma unroll
            for (int u = 0; u < U; u++)
                acc[r][u] = TRANSPOSED ? (V)(S)0 : ldu<true, S, NS>(out + line[r] * n_scalars + s[u]);   // `_m .= 0` (1042) / d as found (1024)
            touched[r] = TRANSPOSED;
        }
        jh_dev_block nb[QQ][R];                                               // block table entries one group of steps ahead
        int nq[QQ], nq2[QQ];                                                  // LIST: the summed block indices one and two groups of steps ahead (the record's padding
#pragma unroll                                                                //       entries are index 0: valid memory, never combined)
        for (int q = 0; q < QQ; q++) {
            nq[q] = LIST ? sidx[q] : 0;
            nq2[q] = LIST ? sidx[QQ + q] : 0;
        }
#pragma unroll
        for (int q = 0; q < QQ; q++)
#pragma unroll
            for (int r = 0; r < R; r++) nb[q][r] = blocks[line[r] * lstep + (LIST ? (int64_t)nq[q] : (q < nsum ? q : 0)) * qstep];
        for (int64_t q0 = 0; q0 < nsum; q0 += QQ) {
            jh_dev_block b[QQ][R];
            int qcur[QQ];
#pragma unroll
            for (int q = 0; q < QQ; q++) {
                qcur[q] = nq[q];
                if (LIST) { nq[q] = nq2[q]; nq2[q] = sidx[q0 + 2 * QQ + q]; }
            }
#pragma unroll
            for (int q = 0; q < QQ; q++) {
                const int64_t qn = q0 + QQ + q;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    b[q][r] = nb[q][r];
                    nb[q][r] = blocks[line[r] * lstep + (LIST ? (int64_t)nq[q] : (qn < nsum ? qn : 0)) * qstep];
                }
            }
            V x[QQ][U], c[QQ][R][U];
#pragma unroll
            for (int q = 0; q < QQ; q++) {
                const S *xb = in + (LIST ? (int64_t)qcur[q] : (q0 + q < nsum ? q0 + q : 0)) * n_scalars;   // (a step beyond the end re-reads block 0: unused)
#pragma unroll
                for (int u = 0; u < U; u++) x[q][u] = ldu<false, S, NS>(xb + s[u]);                     // shared by every line group: through the caches
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const bool has = block_reads_coeff(b[q][r], false);
                    const S *cb = has ? (const S *)b[q][r].coeff : xb;                                    // no coefficient array: the input pack again (L1)
#pragma unroll
                    for (int u = 0; u < U; u++)
                        c[q][r][u] = has ? ldu<true, S, NS>(cb + s[u]) : ldu<false, S, NS>(cb + s[u]);
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
                if (touched[r] && ok[u] && grp * R + r < nlines) st_pack<true, S, NS>(out + line[r] * n_scalars, sn[u], s[u], acc[r][u]);
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
    return op->coeff_aligned16;
}

// the register-tiled kernel of such grids also takes blocks off the 16-byte pack grid (under-aligned packs, a partial last pack per block)
bool grid_tile_ok(const jh_blockop *op, const void *rng_ptr, const void *dom_ptr)
{
    if (grid_diag_ok(op, rng_ptr, dom_ptr)) return true;
    if (!(op->all_diag && op->nrow >= 2 && op->ncol >= 2) || jh_ctx().tall_unaligned == 0) return false;
    const size_t es = jh_dtype_size(op->dtype), sa = jh_dtype_complex(op->dtype) ? es / 2 : es;
    const int64_t n = op->row_len[0];
    if (n * (int64_t)es < 16 || !op->uniform_rows) return false;
    for (int64_t v : op->col_len) if (v != n) return false;
    if ((((uintptr_t)rng_ptr) | ((uintptr_t)dom_ptr)) & (sa - 1)) return false;
    return op->coeff_scalar_aligned;
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

// every block boundary / coefficient pointer / vector base on a 16-byte boundary?
// Round 5, session 3: or -- blocks of odd lengths in one slab -- every non-empty line at least one pack long and everything aligned like its scalar:
// the 16-byte-per-lane kernels then work on under-aligned packs (knob tall_unaligned = 0: the 4-byte-per-lane kernels as before)
bool general_vec_ok(const jh_blockop *op, const void *rng_ptr, const void *dom_ptr)
{
    const bool aligned = !((((uintptr_t)rng_ptr) | ((uintptr_t)dom_ptr)) & 15u) && op->lens_aligned16 && op->coeff_aligned16;   // (known since jh_blockop_create / jh_blockop_point)
    if (aligned) return true;
    const size_t es = jh_dtype_size(op->dtype), sa = jh_dtype_complex(op->dtype) ? es / 2 : es;
    if ((((uintptr_t)rng_ptr) | ((uintptr_t)dom_ptr)) & (sa - 1)) return false;
    return jh_ctx().tall_unaligned != 0 && op->lens_hold_a_pack && op->coeff_scalar_aligned;
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
    general_grid(((n_scalars + NS - 1) / NS + 256 * U - 1) / (256 * U), ngroups, ntiles, grid, general_use_xcd(in_bytes));
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
    // (round 5: 4 lines x 2 steps, 8 lines, and two packs per lane on the four-line shape were measured on the PMC evidence that the kernel moves
    // exactly its unique bytes -- all within -7 ... +2 % of this shape, 8 lines far below: profiles/exp_r05_general_tile_shapes.txt)
    const int64_t ngroups = four ? (nlines + 3) / 4 : (nlines + 1) / 2;
    constexpr int U = 1;                                                 // (two packs per lane did not pay here: measured in rounds 4-5, its instantiations dropped in round 6)
    unsigned ntiles, grid;
    general_grid(((n_scalars + NS - 1) / NS + 256 * U - 1) / (256 * U), ngroups, ntiles, grid, general_use_xcd(in_bytes));
    // late round 5: a SPARSE grid walks step lists built at create (k_general_tile LIST) -- per group of four lines the summed block indices at which one
    // of the lines has a non-zero block (the input pack still shared by four lines), or per line its own non-zero blocks (no dummy load at all, four
    // steps' loads in flight).  Same bits as the plain walk.  32 x 32 block-diagonal of 128^3: forward 2.9 -> 5.6 TB/s, adjoint 2.3 -> 5.5; 64 x 64 of
    // 64^3 1.7 -> 6.6 / 1.4 -> 6.1; 16 x 16 block-bidiagonal of 256^3 5.2 -> 6.1 / 4.8 -> 6.0 (profiles/bench_grid_sparse_r05.txt).
    // Knob general_list: 0 never, 2 / 3 always the four-line / per-line lists (tests), 1 automatic: lists only when the four-line lists leave out at
    // least an eighth of the steps (a dense mix keeps the plain walk); WHICH walk then is measured per operator and direction over its first seven
    // calls (lazy_next, as the tall forward's grid walk: no extra launches, no host synchronisation) when the operator moves >= 64 MiB per call --
    // four-line against per-line lists differ by -12 ... +37 % with the pattern, 8 x 8 grids of 256^3 are as fast or faster on the plain walk;
    // smaller operators take the four-line lists when they save >= 8 steps per group on average, else the plain walk.
    const int dir = TRANSPOSED ? 1 : 0;
    const int64_t nsum_all = TRANSPOSED ? op->nrow : op->ncol, full_steps = ngroups * nsum_all;
    const bool have4 = four && U == 1 && op->dev_steps[dir][0], have1 = U == 1 && op->dev_steps[dir][1];
    int list = 0, slot = -1;                                              // list: 0 the plain walk, 1 the four-line lists, 2 the per-line lists
    if (c.general_list == 2) list = have4 ? 1 : 0;
    else if (c.general_list == 3) list = have1 ? 2 : 0;
    else if (c.general_list == 1 && have4 && op->list_steps[dir][0] * 8 <= full_steps * 7) {
        list = full_steps - op->list_steps[dir][0] >= 8 * ngroups ? 1 : 0;
        if (have1 && c.autotune && (double)op->list_steps[dir][1] * (double)n_scalars * sizeof(S) >= (double)((int64_t)64 << 20) && !stream_is_capturing(c.stream)) {
            static const int walk_list[3] = {1, 2, 0};
            int cand = op->gen_walk[dir];
            if (cand < 0) cand = lazy_next(op->gen_tune[dir], 3, 2, 1, 0.01f, &op->gen_walk[dir], &slot);
            list = walk_list[cand >= 0 && cand < 3 ? cand : 0];
        }
    }
    const bool timing = slot >= 0 && lazy_begin(op->gen_tune[dir], slot, c.stream);
    c.last_general_list = list;
    if (list == 2) {
        // every line on its own: a step = a non-zero block (no dummy loads at all), four steps' loads in flight
        general_grid(((n_scalars + NS - 1) / NS + 255) / 256, nlines, ntiles, grid, general_use_xcd(in_bytes));
        hipLaunchKernelGGL((k_general_tile<S, E, NS, 4, 1, TRANSPOSED, 1, true>), dim3(grid), dim3(256), 0, c.stream, op->dev_blocks, op->nrow, op->ncol, n_scalars, in, out,
                           ntiles, (unsigned)nlines, op->dev_steps[dir][1], (int)op->step_stride[dir][1]);
    } else if (list == 1)
        hipLaunchKernelGGL((k_general_tile<S, E, NS, 1, 1, TRANSPOSED, 4, true>), dim3(grid), dim3(256), 0, c.stream, op->dev_blocks, op->nrow, op->ncol, n_scalars, in, out,
                           ntiles, (unsigned)ngroups, op->dev_steps[dir][0], (int)op->step_stride[dir][0]);
    else if (four)
        hipLaunchKernelGGL((k_general_tile<S, E, NS, 1, 1, TRANSPOSED, 4>), dim3(grid), dim3(256), 0, c.stream, op->dev_blocks, op->nrow, op->ncol, n_scalars, in, out,
                           ntiles, (unsigned)ngroups, (const int *)nullptr, 0);
    else
        hipLaunchKernelGGL((k_general_tile<S, E, NS, 2, 1, TRANSPOSED>), dim3(grid), dim3(256), 0, c.stream, op->dev_blocks, op->nrow, op->ncol, n_scalars, in, out,
                           ntiles, (unsigned)ngroups, (const int *)nullptr, 0);
    const hipError_t le = hipGetLastError();
    if (slot >= 0) lazy_end(op->gen_tune[dir], slot, c.stream, timing && le == hipSuccess);
    JH_CHECK_HIP(le);
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
    int64_t want = vec ? ((maxn * E + NS - 1) / NS + 255) / 256 : (maxn + 255) / 256;     // vec: one pack per thread (see jh_vecops.hip: grid_full)
    if (!vec && want > 4096) want = 4096;
    const bool gdiag = vec && !fmode && c.grid_diag && grid_diag_ok(op, d, m);  // a grid of plain diagonals: the branch-free kernel,
    const int gu = gdiag ? (c.grid_diag >= 4 ? 4 : (c.grid_diag >= 2 ? 2 : 1)) : 1;   // gu packs per lane
    general_grid(want, op->nrow, ntiles, grid, general_use_xcd(op->col_off[(size_t)op->ncol] * (int64_t)(sizeof(S) * E)));
    // split walk over the block columns (general_parts)
    const int64_t out_scalars = op->row_off[(size_t)op->nrow] * E;
    int64_t parts = (op->nrow <= 65535) ? general_parts((int64_t)grid, op->ncol, out_scalars * (int64_t)sizeof(S)) : 1;
    // late round 5: a SPARSE grid walks each row's step list (its non-zero blocks) instead of every table entry -- then there is nothing to split either
    // (unless the lists themselves are long and the launch small: then the split walk's parallelism over the block columns is what the shape needs)
    const bool use_list = !fmode && c.general_list != 0 && op->dev_steps[0][1] && op->list_steps[0][1] * 8 <= op->nrow * op->ncol * 7 &&
                          !(parts > 1 && op->list_steps[0][1] >= 32 * op->nrow);
    const int *lsteps = use_list ? op->dev_steps[0][1] : nullptr;
    const int lstride = use_list ? (int)op->step_stride[0][1] : 0;
    if (use_list) parts = 1;
    int64_t per = 0;
    void *slabs = nullptr;
    if (parts > 1) {
        per = (op->ncol + parts - 1) / parts;
        parts = (op->ncol + per - 1) / per;
        JH_TRY(jh_ensure_scratch((size_t)parts * (size_t)out_scalars * sizeof(S), &slabs));
    }
    c.last_adj_parts = parts;
    if (vec && !fmode && parts == 1 && c.grid_diag && c.grid_tile && grid_tile_ok(op, d, m))
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
                           per, (S *)slabs, out_scalars, lsteps, lstride);
    else
        hipLaunchKernelGGL((k_block_fwd_general<S, E>), dim3(grid, (unsigned)parts), dim3(256), 0, c.stream,
                           op->dev_blocks, op->nrow, op->ncol, op->dev_row_off, op->dev_col_off, (const S *)m, (S *)d, fmode, ntiles,
                           per, (S *)slabs, out_scalars, (const S *)nullptr, lsteps, lstride);
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
    int64_t want = vec ? ((maxn * E + NS - 1) / NS + 255) / 256 : (maxn + 255) / 256;
    if (!vec && want > 4096) want = 4096;
    general_grid(want, op->ncol, ntiles, grid, general_use_xcd(op->row_off[(size_t)op->nrow] * (int64_t)(sizeof(S) * E)));
    // split walk over the block rows (general_parts); nrow >= 4 there, so every column is zeroed first (1042): all lines touched
    const int64_t out_scalars = op->col_off[(size_t)op->ncol] * E;
    int64_t parts = (op->ncol <= 65535) ? general_parts((int64_t)grid, op->nrow, out_scalars * (int64_t)sizeof(S)) : 1;
    const bool use_list = c.general_list != 0 && op->dev_steps[1][1] && op->list_steps[1][1] * 8 <= op->nrow * op->ncol * 7 &&
                          !(parts > 1 && op->list_steps[1][1] >= 32 * op->ncol);                                           // (see general_fwd)
    const int *lsteps = use_list ? op->dev_steps[1][1] : nullptr;
    const int lstride = use_list ? (int)op->step_stride[1][1] : 0;
    if (use_list) parts = 1;
    int64_t per = 0;
    void *slabs = nullptr;
    if (parts > 1) {
        per = (op->nrow + parts - 1) / parts;
        parts = (op->nrow + per - 1) / per;
        JH_TRY(jh_ensure_scratch((size_t)parts * (size_t)out_scalars * sizeof(S), &slabs));
    }
    c.last_adj_parts = parts;
    if (vec && parts == 1 && c.grid_diag && c.grid_tile && grid_tile_ok(op, d, m))
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
                           per, (S *)slabs, out_scalars, (parts == 1 && c.nt && out_scalars * (int64_t)sizeof(S) >= ((int64_t)64 << 20)) ? 1 : 0, lsteps, lstride);
    else
        hipLaunchKernelGGL((k_block_adj_general<S, E>), dim3(grid, (unsigned)parts), dim3(256), 0, c.stream,
                           op->dev_blocks, op->nrow, op->ncol, op->dev_row_off, op->dev_col_off, (S *)m, (const S *)d, ntiles,
                           per, (S *)slabs, out_scalars, (const S *)nullptr, lsteps, lstride);
    JH_CHECK_HIP(hipGetLastError());
    if (parts > 1) return launch_fold_general<S>(slabs, out_scalars, parts, m, op->dev_col_off, E, op->ncol, maxn * E, nullptr, 0);
    return JH_OK;
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
}  // namespace
namespace jhb {
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
}  // namespace jhb
namespace {

}  // namespace
namespace jhb {
int dense_grid_adj(const jh_blockop *op, void *m, const void *d)
{
    const size_t es = jh_dtype_size(op->dtype);
    const int64_t nr = op->blocks[0].nr, nc = op->blocks[0].nc;
    for (int64_t j = 0; j < op->ncol; j++)
        JH_TRY(jh_launch_gemv_batched(op->dev_blocks + j * op->nrow, op->nrow, nr, nc, op->dtype, (char *)m + (size_t)(j * nc) * es, d, 1,
                                      op->dense_aligned, false));
    return JH_OK;
}
}  // namespace jhb
namespace {

}  // namespace
namespace jhb {
int loop_fwd(const jh_blockop *op, void *d, const void *m, bool fmode)   // JetBlock_df! / JetBlock_f!
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
}  // namespace jhb
namespace {

}  // namespace
namespace jhb {
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
}  // namespace jhb
namespace {

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
// The combine launch of an operator whose non-zero blocks are ALL dense children (round 6): output element e of line l is the ordered sum of the products the
// children's launches left in the scratch vector -- `_d .+= mul!(dtmp, op, _m)` (1024) from d as found, `_m .+= mul!(mtmp, op', _d)` (1049) from the zero of
// 1042, or a single overwrite (1026 / 1051) -- read from a per-line list of product places built at create.  The general kernel does the same sum walking the
// block table entry by entry, each step waiting for the one before: 17-32 us for the 8 K ... 16 K output elements of a 32 x 32 / 64 x 64 grid of dense
// children, 40 % of the whole forward (profiles/rocprof_r06_dense_odd_summary.md).  Here eight products are requested at once; the adds stay in order: same bits.
// mode: 0 from what the output holds, 1 from +0, 2 the line's single product overwrites.
template <typename S, int E>
__global__ __launch_bounds__(256) void k_combine_dense(const int *__restrict__ line_ptr, const int64_t *__restrict__ prod_at, const int64_t *__restrict__ line_off,
                                                       const S *__restrict__ prod, S *__restrict__ out, unsigned ntiles, int mode, int zero_empty)
{
    const unsigned l = blockIdx.x / ntiles, tile = blockIdx.x - l * ntiles;
    const int64_t o0 = line_off[l], n = line_off[l + 1] - o0;
    const int p0 = line_ptr[l], p1 = line_ptr[l + 1];
    for (int64_t e = (int64_t)tile * 256 + threadIdx.x; e < n; e += (int64_t)ntiles * 256) {
        if (p0 == p1) {                                                    // a line without blocks: the forward leaves it as found (1022), the adjoint has zeroed it (1042)
            if (zero_empty) { elem<S, E> z; z.re = 0; z.im = 0; estore<S, E>(out, o0 + e, z); }
            continue;
        }
        elem<S, E> acc;
        if (mode == 0) acc = eload<S, E>(out, o0 + e);
        else { acc.re = 0; acc.im = 0; }
        int p = p0;
        for (; p + 8 <= p1; p += 8) {
            elem<S, E> v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = eload<S, E>(prod, prod_at[p + k] + e);
#pragma unroll
            for (int k = 0; k < 8; k++) acc = (mode == 2) ? v[k] : eadd<S, E>(acc, v[k]);
        }
        for (; p < p1; p++) {
            const elem<S, E> v = eload<S, E>(prod, prod_at[p] + e);
            acc = (mode == 2) ? v : eadd<S, E>(acc, v);
        }
        estore<S, E>(out, o0 + e, acc);
    }
}

template <typename S, int E>
// fmode (round 4): JetBlock_f! (988-1008) of such an operator -- the dense children's products are the same launches, the combine is
// the general kernel in its f! mode (a zero block's `d .= 0` is added, not skipped; a SQUARE child squares): two launches where the
// per-block loop made two per block
int dense_mixed_apply(const jh_blockop *op, void *out, const void *in, bool transposed, bool fmode = false)
{
    jh_context &c = jh_ctx();
    const size_t es = jh_dtype_size(op->dtype);
    const int64_t nrange = op->row_off[(size_t)op->nrow], ndomain = op->col_off[(size_t)op->ncol];
    const int dir = transposed ? 1 : 0;
    // late round 5, direct mode: when every output line holds ONE block, a dense child (a block-diagonal operator), the children's launch writes the output
    // vector itself -- the product rounded like dtmp / mtmp, then (forward of an operator with several block columns) added to d as found (1024): what the
    // combine launch would have computed, in one launch, without scratch.  Knob dense_direct (1; 0: always through the scratch vector and the combine)
    if (c.dense_list && c.dense_direct && op->dense_direct[dir] && !fmode) {
        int64_t launches = 0;
        const int add_found = (!transposed && op->ncol > 1) ? 1 : 0;
        for (int pass = 0; pass < 2; pass++)
            if (op->n_items[dir][pass] > 0) {
                JH_TRY(jh_launch_gemv_list(op->dev_items[dir][pass], op->n_items[dir][pass], op->items_max_out[dir][pass], op->items_max_in[dir][pass], pass, op->dtype,
                                           nullptr, in, op->dense_mixed_aligned ? 2 : (op->lens_hold_a_pack ? 1 : 0), out, add_found));
                launches++;
            }
        c.last_adj_parts = 1;
        c.last_launches = launches;
        return JH_OK;
    }
    void *slabs = nullptr;                                                      // the products of the dense children: one compact piece each (op->prod_off)
    JH_TRY(jh_ensure_scratch((size_t)op->prod_total[dir] * es + 16, &slabs));
    // which kernel a dense child needs in this direction: block = B (un-adjointed) or B' (adjointed), the operator's adjoint flips it;
    // B x is the sequential rows kernel, B' x the wave-reduction cols kernel
    int64_t launches = 0, ndense = 0, rows_max_out = 0, cols_max_out = 0, wgs = 0;
    double max_bytes = 0.0;
    if (c.dense_list) {
        // late round 5: the children from their LISTS (built at create: nothing here walks the M x K block table on the host -- a block-diagonal operator of
        // 1024 x 1024 blocks spent 1 ms per call in the scan below --, no workgroup for a block pair without a dense child, few children spread over the chip)
        for (int pass = 0; pass < 2; pass++)
            if (op->n_items[dir][pass] > 0) {
                JH_TRY(jh_launch_gemv_list(op->dev_items[dir][pass], op->n_items[dir][pass], op->items_max_out[dir][pass], op->items_max_in[dir][pass], pass, op->dtype,
                                           slabs, in, op->dense_mixed_aligned ? 2 : (op->lens_hold_a_pack ? 1 : 0)));
                launches++;
            }
    } else
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
                    char *o = (char *)slabs + (size_t)op->prod_off[dir][(size_t)(i + j * op->nrow)] * es;
                    const char *x = (const char *)in + (size_t)(transposed ? op->row_off[(size_t)i] : op->col_off[(size_t)j]) * es;
                    JH_TRY(jh_launch_gemv(b.coeff, b.nr, b.nc, op->dtype, o, x, adj ? 1 : 0));
                    launches++;
                }
        } else {
            JH_TRY(jh_launch_gemv_mixed_all(op->dev_blocks, op->nrow, op->ncol, rows_max_out, cols_max_out, op->dtype, slabs, 0, in, transposed ? 1 : 0,
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
        // the lines' non-zero blocks from their step lists when those leave out an eighth of the table entries (never in f! mode: it adds a zero block's +0)
        const int64_t nsum_all = transposed ? op->nrow : op->ncol;
        const bool use_list = !fmode && c.general_list != 0 && op->dev_steps[dir][1] && op->list_steps[dir][1] * 8 <= nlines * nsum_all * 7;
        const int *lsteps = use_list ? op->dev_steps[dir][1] : nullptr;
        const int lstride = use_list ? (int)op->step_stride[dir][1] : 0;
        if (!fmode && c.dense_combine && op->dev_comb_ptr[dir]) {             // every non-zero block a dense child: the products' ordered sum from its lists
            const int64_t nsum = transposed ? op->nrow : op->ncol;
            // forward: several block columns accumulate into d as found (1024), one column overwrites (1026); adjoint: several rows sum from the zero of 1042,
            // one row writes directly (1051) and leaves a column without a block untouched (1047)
            const int mode = nsum > 1 ? (transposed ? 1 : 0) : 2;
            int64_t ct = (maxn + 255) / 256;
            if (ct > 64) ct = 64;
            hipLaunchKernelGGL((k_combine_dense<S, E>), dim3((unsigned)(nlines * ct)), dim3(256), 0, c.stream, op->dev_comb_ptr[dir], op->dev_comb_off[dir],
                               transposed ? op->dev_col_off : op->dev_row_off, (const S *)slabs, (S *)out, (unsigned)ct, mode, (transposed && nsum > 1) ? 1 : 0);
        } else if (!transposed)
            hipLaunchKernelGGL((k_block_fwd_general<S, E>), dim3(grid, 1), dim3(256), 0, c.stream, op->dev_blocks, op->nrow, op->ncol, op->dev_row_off,
                               op->dev_col_off, (const S *)in, (S *)out, fmode ? 1 : 0, ntiles, (int64_t)0, (S *)nullptr, (int64_t)0, (const S *)slabs, lsteps, lstride);
        else
            hipLaunchKernelGGL((k_block_adj_general<S, E>), dim3(grid, 1), dim3(256), 0, c.stream, op->dev_blocks, op->nrow, op->ncol, op->dev_row_off,
                               op->dev_col_off, (S *)out, (const S *)in, ntiles, (int64_t)0, (S *)nullptr, (int64_t)0, (const S *)slabs, lsteps, lstride);
        JH_CHECK_HIP(hipGetLastError());
        launches++;
    }
    c.last_launches = launches;
    return JH_OK;
}

}  // namespace
namespace jhb {
int dense_mixed(const jh_blockop *op, void *out, const void *in, bool transposed, bool fmode)
{
    switch (op->dtype) {
    case JH_F32: return dense_mixed_apply<float, 1>(op, out, in, transposed, fmode);
    case JH_F64: return dense_mixed_apply<double, 1>(op, out, in, transposed, fmode);
    case JH_C32: return dense_mixed_apply<float, 2>(op, out, in, transposed, fmode);
    case JH_C64: return dense_mixed_apply<double, 2>(op, out, in, transposed, fmode);
    }
    return jh_fail(JH_ERR_INVALID, "dense_mixed: unknown dtype %d", op->dtype);
}
}  // namespace jhb
namespace {

}  // namespace
namespace jhb {
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
}  // namespace jhb
namespace {


}  // namespace

namespace jhb {

int general_fwd(const jh_blockop *op, void *d, const void *m, int fmode)
{
    switch (op->dtype) {
    case JH_F32: return ::general_fwd<float, 1>(op, d, m, fmode);
    case JH_F64: return ::general_fwd<double, 1>(op, d, m, fmode);
    case JH_C32: return ::general_fwd<float, 2>(op, d, m, fmode);
    case JH_C64: return ::general_fwd<double, 2>(op, d, m, fmode);
    }
    return jh_fail(JH_ERR_INVALID, "general forward: unknown dtype %d", op->dtype);
}

int general_adj(const jh_blockop *op, void *m, const void *d)
{
    switch (op->dtype) {
    case JH_F32: return ::general_adj<float, 1>(op, m, d);
    case JH_F64: return ::general_adj<double, 1>(op, m, d);
    case JH_C32: return ::general_adj<float, 2>(op, m, d);
    case JH_C64: return ::general_adj<double, 2>(op, m, d);
    }
    return jh_fail(JH_ERR_INVALID, "general adjoint: unknown dtype %d", op->dtype);
}

}  // namespace jhb
