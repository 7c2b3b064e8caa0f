// jh_tall_step.hip -- the solver-side kernels of the tall path: the fused updates d = alpha (A m) + beta d / m = alpha (A' d) + beta m with
// ||result||^2 (the halves of an LSQR / CGLS iteration, and the passes of `(a * A) * m` / `(a * A)' * d`, src/Jets.jl:1159-1164), the ONE-PASS
// Golub-Kahan step (k_tall_diag_bidiag) and its chained-row-chunk form, their launch rules and entry points.
// One of the translation units jh_blockop.hip was split into in round 5 (jh_blockop_common.h).
#include "jh_blockop_common.h"

namespace {

// rows that are not whole, 16-byte aligned packs (jh_tall.hip: tall_unaligned_ok): such operators run the MIXED instantiations, whose accesses are under-aligned
static inline bool packs_unaligned(const jh_blockop *op, size_t row_bytes, const void *a, const void *b, const void *c = nullptr)
{
    return row_bytes % 16 != 0 || !op->coeff_aligned16 || ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15u) != 0;
}

// forward: d_i = alpha * (a_i .* m) + beta * d_i ; sequential row sweep (tile index fastest)
// WIDE (S = float, beta == 0): the scalar is Julia's Float64 (JH_SCALAR_WIDE) -- d_i = Float32(wscal * Float64(a_i .* m)), the promoted
// product of `d .= a * tmp` (src/Jets.jl:1159) rounded once on the store
template <typename S, int E, int NS, int U, int BLK, bool MIXED = false, bool WIDE = false, bool NT = true>
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
        // MIXED (round 5, session 3): these instantiations also serve rows that are not whole, 16-byte aligned packs (jh_tall.hip: tall_unaligned_ok) --
        // under-aligned accesses, the row's last pack loaded from n - NS, stored and counted from its own first scalar on (st_pack, vnorm2_from)
        if constexpr (MIXED) {
            sk[k] = ok[k] ? pack_start<NS>(s0 + (int64_t)k * BLK * NS, n_scalars) : 0;
            mv[k] = ldu<false, S, NS>(m + sk[k]);
        } else {
            sk[k] = ok[k] ? s0 + (int64_t)k * BLK * NS : 0;
            mv[k] = ld<false>(reinterpret_cast<const V *>(m + sk[k]));
        }
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
            if constexpr (MIXED) {                                      // (NT = false: rows off the 16-byte grid, see launch_fwd_update)
                av[k] = rc ? ldu<NT, S, NS>(a + sk[k]) : (V)(S)0;
                dv[k] = use_old ? ldu<NT, S, NS>(di + sk[k]) : (V)(S)0;
            } else {
                av[k] = rc ? ld<true>(reinterpret_cast<const V *>(a + sk[k])) : (V)(S)0;
                dv[k] = use_old ? ld<true>(reinterpret_cast<const V *>(di + sk[k])) : (V)(S)0;   // beta == 0: d is write-only
            }
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
                if constexpr (MIXED) {
                    const int64_t sn = s0 + (int64_t)k * BLK * NS;
                    st_pack<true, S, NS>(di, sn, sk[k], r);               // (streaming whatever NT says: jh_tall.hip, k_tall_diag_fwd)
                    nrm += vnorm2_from<S, NS, V>(r, (int)(sn - sk[k]));
                } else {
                    st<true>(reinterpret_cast<V *>(di + sk[k]), r);
                    nrm += vnorm2<S, NS, V>(r);
                }
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
        sk[k] = ok[k] ? pack_start<NS>(s0 + (int64_t)k * BLK * NS, n_scalars) : 0;   // (rows need not be whole, 16-byte aligned packs: ldu / st_pack / vnorm2_from,
        acc[k] = (V)(S)0;                                                            //  jh_blockop_common.h; round 5, last session)
    }
    int64_t i = 0;
    for (; !direct && i + DEPTH <= nrow; i += DEPTH) {
        V av[DEPTH][U], dv[DEPTH][U];
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            const S *a = a_base ? a_base + (i + j) * a_stride : (const S *)blocks[i + j].coeff;
#pragma unroll
            for (int k = 0; k < U; k++) {
                av[j][k] = ldu<true, S, NS>(a + sk[k]);
                dv[j][k] = ldu<true, S, NS>(in + (i + j) * n_scalars + sk[k]);
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
            V p = vmul<S, E, NS, V>(ldu<true, S, NS>(a + sk[k]),
                                    scaled(ldu<true, S, NS>(in + i * n_scalars + sk[k])), true);
            acc[k] = direct ? p : acc[k] + p;
        }
    }
    double nrm = 0.0;
#pragma unroll
    for (int k = 0; k < U; k++) {
        V s1 = (V)alpha * acc[k];
        V r = s1;
        if (beta != (S)0) { V s2 = (V)beta * ldu<false, S, NS>(out + sk[k]); r = s1 + s2; }
        if (ok[k]) {
            const int64_t sn = s0 + (int64_t)k * BLK * NS;
            st_pack<false, S, NS>(out, sn, sk[k], r);
            nrm += vnorm2_from<S, NS, V>(r, (int)(sn - sk[k]));
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
// (Round 5, tried and dropped: the all-diagonal walk SOFTWARE-PIPELINED -- the loads of batch b + 1 issued before batch b is combined and
// stored, two register buffers: 34.27 against 34.50 ms at 1024 x 256^3, 8.80 against 8.79 at 256 x 256^3, profiles/exp_r05_step_pipe.txt.)
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
    int e0[U];                                                              // MIXED: the first scalar of pack k that this lane OWNS (0 but for a row's partial last pack)
#pragma unroll
    for (int k = 0; k < U; k++) {
        ok[k] = (s0 + (int64_t)k * BLK * NS) < s_end;
        // MIXED (round 5, session 3): these instantiations also serve rows that are not whole, 16-byte aligned packs (jh_tall.hip: tall_unaligned_ok) --
        // under-aligned accesses, the last pack loaded from s_end - NS, stored and counted from its own first scalar on (st_pack, vnorm2_from)
        if constexpr (MIXED) {
            sk[k] = ok[k] ? pack_start<NS>(s0 + (int64_t)k * BLK * NS, s_end) : s_begin;
            e0[k] = ok[k] ? (int)(s0 + (int64_t)k * BLK * NS - sk[k]) : 0;
            acc[k] = (accumulate && ok[k]) ? ldu<false, S, NS>(w + sk[k]) : (V)(S)0;
            vv[k] = ldu<false, S, NS>(v + sk[k]);
        } else {
            sk[k] = ok[k] ? s0 + (int64_t)k * BLK * NS : s_begin;
            e0[k] = 0;
            acc[k] = (accumulate && ok[k]) ? ld<false>(reinterpret_cast<const V *>(w + sk[k])) : (V)(S)0;
            vv[k] = ld<false>(reinterpret_cast<const V *>(v + sk[k]));
        }
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
                        av[j][k] = ldu<NT, S, NS>(ap + sk[k]);
                        if constexpr (OLD) uv[j][k] = ldu<NT, S, NS>(u + (i + j) * n_scalars + sk[k]);
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
                                st_pack<true, S, NS>(u + (i + j) * n_scalars, sk[k] + e0[k], sk[k], r);
                                nrm += vnorm2_from<S, NS, V>(r, e0[k]);
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
                            st_pack<true, S, NS>(u + (i + j) * n_scalars, sk[k] + e0[k], sk[k], r);
                            nrm += vnorm2_from<S, NS, V>(r, e0[k]);
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
                const V c = ldu<NT, S, NS>(ap + sk[k]);
                const V t = on ? apply_block_loaded<S, E, NS, V>(blk, vv[k], c, false, false) : (V)(S)0;
                V r = (V)alpha * t;
                if (use_old) { V s2 = (V)beta * ldu<NT, S, NS>(u + i * n_scalars + sk[k]); r = r + s2; }
                if (ok[k]) {
                    st_pack<true, S, NS>(u + i * n_scalars, sk[k] + e0[k], sk[k], r);
                    nrm += vnorm2_from<S, NS, V>(r, e0[k]);
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
        if (ok[k]) {
            if constexpr (MIXED) st_pack<false, S, NS>(w, sk[k] + e0[k], sk[k], acc[k]);
            else st<false>(reinterpret_cast<V *>(w + sk[k]), acc[k]);
        }
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
        // otherwise (256 x 256^3 over separate arrays: chained step 5.7 TB/s against 6.1-6.3 over one slab).
        // The whole section is BRANCH-FREE and its addresses are opaque to the compiler (round 5, from the ISA): a `use_old ? load : 0` or
        // `reads a coefficient ? load : 0` per row compiles to a branch per load and, at the first merge, an s_waitcnt vmcnt(0) for that
        // load to RETURN before the other 63 are issued -- one memory round trip per workgroup with nothing else resident on the CU.
        // beta == 0 (u may hold anything and is not used) aims the u loads at v's pack, and so does a row without a coefficient array with
        // its coefficient load: hits in L2 (every chunk reads v) whose values nobody reads.  (Aimed at the row's u pack instead, the second
        // streaming load of a line still in flight fetched it twice: 50 % scalar rows 5.8 -> 5.2 TB/s.)
        const S *ub = use_old ? u + row0 * n_scalars : v, *vd = v;
        int64_t ustep = use_old ? n_scalars : 0;
        asm volatile("" : "+s"(ub), "+s"(ustep), "+s"(vd));
#pragma unroll
        for (int j = 0; j < DEPTH; j++)
#pragma unroll
            for (int k = 0; k < U; k++) uv[j][k] = ld<true>(reinterpret_cast<const V *>(ub + j * ustep + sk[k]));
        if (!MIXED && a_base) {
            const S *ap = a_base + row0 * a_stride;
#pragma unroll
            for (int j = 0; j < DEPTH; j++)
#pragma unroll
                for (int k = 0; k < U; k++) av[j][k] = ld<true>(reinterpret_cast<const V *>(ap + j * a_stride + sk[k]));
        } else {
            constexpr int G = DEPTH < 8 ? DEPTH : 8;                       // table entries are fetched G rows at a time: one scalar round trip per group
#pragma unroll
            for (int g = 0; g < DEPTH; g += G) {
                const S *ap[G];
#pragma unroll
                for (int j = 0; j < G; j++) {
                    if constexpr (MIXED) {
                        blk[g + j] = blocks[row0 + g + j];
                        ap[j] = block_reads_coeff(blk[g + j], false) ? (const S *)blk[g + j].coeff : vd;
                    } else
                        ap[j] = (const S *)blocks[row0 + g + j].coeff;
                }
#pragma unroll
                for (int j = 0; j < G; j++)
#pragma unroll
                    for (int k = 0; k < U; k++) av[g + j][k] = ld<true>(reinterpret_cast<const V *>(ap[j] + sk[k]));
            }
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
    const int64_t nvec = (n_scalars + NS - 1) / NS;
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
    const bool mixed = !op->all_diag || packs_unaligned(op, (size_t)n_scalars * sizeof(S), d, m);   // rows of several elementwise kinds (or off the pack grid): the bands, always
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
#define JH_LAUNCH_WN(BLK, UU, MX, WD, NTV)                                                                             \
    hipLaunchKernelGGL((k_tall_diag_fwd_update<S, E, NS, UU, BLK, MX, WD, NTV>), dim3((unsigned)(gx * gy)), dim3(BLK), 0, c.stream, \
                       op->dev_blocks, op->nrow, G, a_base, a_stride, (const S *)m, (S *)d, n_scalars, (unsigned)gx,   \
                       (unsigned)gy, walk, (S)alpha, (S)beta, c.part_dev, alpha)
#define JH_LAUNCH_W(BLK, UU, MX, WD) JH_LAUNCH_WN(BLK, UU, MX, WD, true)
    // rows off the 16-byte grid: temporal accesses (jh_tall.hip: launch_tall_fwd_mixed), one pack per lane
    const bool off_grid = packs_unaligned(op, (size_t)n_scalars * sizeof(S), d, m);
    if (off_grid && !(sizeof(S) == 4 && wide) && (c.ua_nt == 0 || c.ua_nt < 0)) {
        if (U == 4) JH_LAUNCH_WN(256, 4, true, false, false); else JH_LAUNCH_WN(256, 1, true, false, false);
    } else
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
        if (off_grid && (c.ua_nt == 0 || c.ua_nt < 0)) { /* launched above */ }
        else if (mixed && U == 4) JH_LAUNCH_W(256, 4, true, false);
        else if (mixed) JH_LAUNCH_W(256, 1, true, false);
        else if (U == 4) JH_LAUNCH_W(256, 4, false, false);
        else JH_LAUNCH_W(256, 1, false, false);
    }
#undef JH_LAUNCH_W
#undef JH_LAUNCH_WN
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
    const int64_t nvec = (n_scalars + NS - 1) / NS;
    const int direct = op->nrow == 1 ? 1 : 0;
    {   // many rows of small blocks: the row sum through the split walk of the plain adjoint, then out = (alpha*gamma)*t + beta*out
        // with ||out||^2 in a small epilogue (tolerance parity, like every split sum)
        void *tmp = nullptr;
        if (!wide) JH_TRY(jhb::split_adjoint_tmp(op, &tmp));   // (a wide in_scale is applied per d_i before the sum: the ordered walk)
        if (tmp) {
            JH_TRY(jhb::tall_adj(op, tmp, in, 0, !tall_fast_ok(op, in, tmp)));   // (rows off the pack grid: the MIXED instantiations)
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
static StepShape pick_step_shape(const jh_blockop *op, int64_t nvec, bool complex_f32, bool mixed)
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
    if (mixed) {                                            // rows of several elementwise kinds (tall_mixed_ok) or off the pack grid: four instantiated shapes
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
    const bool unaligned = packs_unaligned(op, (size_t)n_scalars * sizeof(S), u, v, w);   // (whole-vector calls only: jh_blockop_bidiag_step)
    const bool mixed = !op->all_diag || unaligned;          // rows of several elementwise kinds (tall_mixed_ok), or off the pack grid
    const StepShape shape = pick_step_shape(op, nvec, E == 2 && sizeof(S) == 4, mixed);
    int wg = shape.wg, U = shape.U, D = shape.D;
    const int64_t gx = ((s_end - s_begin + NS - 1) / NS + (int64_t)wg * U - 1) / ((int64_t)wg * U);
    // many rows of small blocks: split-row walk (pick_adj_parts): u's rows are updated as before, w's sum is folded from slabs
    int64_t parts = (direct || s_end - s_begin < NS) ? 1 : pick_adj_parts(gx, op->nrow);   // (a range shorter than one pack -- the tail of an off-grid vector -- loads from before
    int64_t rows_per_part = 0;                                                            //  its begin: no slabs indexed from there)
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
    // rows per chunk = rows in flight.  Round 5: THIRTY-TWO rows of one 256-lane tile per workgroup for all-diagonal operators (thirty-two rows
    // of a and u in flight are 256 registers per lane: one wave per SIMD, 64 loads in flight per lane) -- a quarter of the hand-offs and of the
    // re-reads of v of the 8-row chunks of rounds 2-4, and each workgroup still "born, moves one batch, dies": 1024 x 256^3 5.98 -> 6.23 TB/s
    // (34.5 -> 33.1 ms), 256 x 256^3 5.99 -> 6.23, 128 x 256^3 6.11 -> 6.36, and rows of 8 MiB, where the 8-row chunks never paid, 5.67 -> 6.02
    // (profiles/exp_r05_step_chunks.txt; 16-row chunks of 512 lanes: 6.19-6.29 on 64 MiB rows, 5.2 on 8 MiB rows).  Rows of several kinds keep
    // 8 rows x 1024 lanes (their row descriptors live in SGPRs).  Knob step_chunk: 0 this rule, 8 / 16 / 32 that many rows.
    const int64_t span = s_end - s_begin;
    const bool knobs_free = !c.adj_wg && !c.adj_unroll && !c.adj_depth;
    int CD = 8, cb = 0;                                                   // rows per chunk; chained: workgroup size (0: the shape does not allow it)
    // (rows off the 16-byte grid keep the plain walk: a chained walk with a ragged last tile -- lanes that own part of a pack or nothing, handing on only what they own --
    // was built, tested and measured at -3 ... +6 % against it, profiles/exp_r05_chain_tail.txt; not kept)
    const bool chain_base = !direct && parts == 1 && rows_per_launch == op->nrow && !unaligned;
    if (chain_base && !mixed && (c.step_chunk == 0 || c.step_chunk == 32) && span % ((int64_t)256 * NS) == 0 && op->nrow > 32 &&
        (c.step_chain == 1 || (span / ((int64_t)256 * NS) >= 2048 && op->nrow >= 64)) && !(c.step_chain == 1 && c.adj_wg && c.adj_wg != 256)) {
        CD = 32;
        cb = 256;
    } else if (!mixed && c.step_chunk == 16) {
        CD = 16;
    }
    const int64_t nchunks = (op->nrow + CD - 1) / CD;
    if (!cb && chain_base && nchunks >= 2)
        for (int b : {1024, 512, 256}) {
            if (CD == 16 && b == 1024) continue;
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
#define JH_CHAIN(BLK, MIX) JH_CHAIN_D(BLK, MIX, 8)
#define JH_CHAIN_D(BLK, MIX, CDD)                                                                                         \
    hipLaunchKernelGGL((k_tall_diag_bidiag_chain<S, E, NS, 1, CDD, BLK, MIX>), dim3((unsigned)(ntiles * nchunks)), dim3(BLK), 0, c.stream, \
                       op->dev_blocks, op->nrow, a_base, a_stride, (S *)u, (const S *)v, (S *)w, n_scalars, (S)alpha, (S)beta,   \
                       c.part_dev, s_begin, s_end, (unsigned)ntiles, (unsigned)nchunks, c.chain_sync, (S *)wpart, err, (unsigned)cband)
        if (mixed) {
            if (cb == 1024) JH_CHAIN(1024, true);
            else if (cb == 512) JH_CHAIN(512, true);
            else JH_CHAIN(256, true);
        } else if (CD == 32) {
            JH_CHAIN_D(256, false, 32);
        } else if (CD == 16) {
            if (cb == 512) JH_CHAIN_D(512, false, 16);
            else JH_CHAIN_D(256, false, 16);
        } else {
            if (cb == 1024) JH_CHAIN(1024, false);
            else if (cb == 512) JH_CHAIN(512, false);
            else JH_CHAIN(256, false);
        }
#undef JH_CHAIN
#undef JH_CHAIN_D
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
            if (parts > 1) JH_TRY(jhb::fold_parts(op->dtype, slabs, part_stride, parts, w, s_begin, s_end));              \
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
    if (unaligned && c.ua_nt <= 0) {                                       // rows off the 16-byte grid: temporal accesses (jh_tall.hip: launch_tall_fwd_mixed)
        JH_LAUNCH_N(512, 1, 4, true, false) JH_LAUNCH_N(256, 2, 2, true, false) JH_LAUNCH_N(256, 4, 1, true, false) JH_LAUNCH_N(256, 1, 4, true, false)
    }
    JH_LAUNCH_M(512, 1, 4, true) JH_LAUNCH_M(256, 2, 2, true) JH_LAUNCH_M(256, 4, 1, true) JH_LAUNCH_M(256, 1, 4, true)
#undef JH_LAUNCH
#undef JH_LAUNCH_T
#undef JH_LAUNCH_M
#undef JH_LAUNCH_N
    return jh_fail(JH_ERR_INVALID, "fused bidiagonalisation step: shape %d x %d x %d is not instantiated", wg, U, D);
}


}  // namespace

// into how many row ranges the one-pass step over the whole domain cuts this operator (1: one plain launch -- what the graph-replayed
// solver loops of jh_lsqr.hip need, because only the plain launch reads its coefficients from the device)
int64_t jh_bidiag_step_parts(const jh_blockop *op)
{
    if (op->nrow == 1) return 1;
    const int64_t ssize = (int64_t)jh_dtype_size(op->dtype) / (jh_dtype_complex(op->dtype) ? 2 : 1);
    const int64_t n_scalars = op->col_len[0] * (jh_dtype_complex(op->dtype) ? 2 : 1), NS = 16 / ssize;
    const bool mixed = !op->all_diag || !op->coeff_aligned16 || (n_scalars * ssize) % 16 != 0;
    const StepShape sh = pick_step_shape(op, n_scalars / NS, op->dtype == JH_C32, mixed);
    const int64_t gx = (n_scalars / NS + (int64_t)sh.wg * sh.U - 1) / ((int64_t)sh.wg * sh.U);
    return pick_adj_parts(gx, op->nrow);
}

extern "C" {

int jh_blockop_bidiag_step(const jh_blockop *op, jh_bvec *u, const jh_bvec *v, jh_bvec *w, double alpha, double beta, double *normsq)
{
    JH_TRY(jh_enter(op, u, v, w));
    JH_TRY(check_vectors(op, u, v, "jh_blockop_bidiag_step"));
    JH_REQUIRE(w && w->dtype == op->dtype && w->length == v->length, "jh_blockop_bidiag_step: w must be a domain vector of the operator");
    JH_REQUIRE(w->data != v->data, "jh_blockop_bidiag_step: w must not alias v");
    // (round 5, session 3: rows off the 16-byte pack grid -- odd block lengths in one slab -- run the plain walk's MIXED instantiations on under-aligned packs)
    if (!jh_blockop_tall_step_ok(op, u->data, v->data) || (((uintptr_t)w->data) & (jh_dtype_size(op->dtype) / (jh_dtype_complex(op->dtype) ? 2 : 1) - 1)))
        return jh_fail(JH_ERR_UNSUPPORTED, "jh_blockop_bidiag_step: needs a tall operator of >= 2 equal elementwise rows");
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
    // (rows off the 16-byte pack grid: the plain walk's MIXED instantiations, like the whole-vector call -- the LAST range may then end inside a pack)
    if (!jh_blockop_tall_step_ok(op, u->data, v->data) || (((uintptr_t)w->data) & (jh_dtype_size(op->dtype) / (jh_dtype_complex(op->dtype) ? 2 : 1) - 1)))
        return jh_fail(JH_ERR_UNSUPPORTED, "jh_blockop_bidiag_step_range: needs a tall operator of >= 2 equal elementwise rows");
    const int64_t es = (int64_t)jh_dtype_size(op->dtype);
    JH_REQUIRE((first_elem * es) % 16 == 0 && ((count * es) % 16 == 0 || first_elem + count == v->length),
               "jh_blockop_bidiag_step_range: chunk boundaries must be 16-byte aligned (the last chunk may end with the vector)");
    const int64_t n = op->row_len[0], lo = first_elem, hi = first_elem + count;
    switch (op->dtype) {
    case JH_F32: return launch_bidiag<float, 1, 4>(op, u->data, v->data, w->data, n, alpha, beta, normsq, lo, hi, true);
    case JH_F64: return launch_bidiag<double, 1, 2>(op, u->data, v->data, w->data, n, alpha, beta, normsq, lo, hi, true);
    case JH_C32: return launch_bidiag<float, 2, 4>(op, u->data, v->data, w->data, 2 * n, alpha, beta, normsq, 2 * lo, 2 * hi, true);
    case JH_C64: return launch_bidiag<double, 2, 2>(op, u->data, v->data, w->data, 2 * n, alpha, beta, normsq, 2 * lo, 2 * hi, true);
    }
    return jh_fail(JH_ERR_INVALID, "jh_blockop_bidiag_step_range: unknown dtype %d", op->dtype);
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
    if (!jh_blockop_tall_step_ok(op, d->data, m->data))
        return jh_fail(JH_ERR_UNSUPPORTED, "jh_blockop_mul_axpby: needs a tall operator of equal elementwise rows; "
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
    if (!tall_fast_ok(op, d->data, m->data) && !(op->all_diag && tall_unaligned_ok(op, d->data, m->data)))   // (rows off the 16-byte pack grid: under-aligned packs)
        return jh_fail(JH_ERR_UNSUPPORTED, "jh_blockop_mul_adj_axpby: needs a tall all-DIAG operator with equal blocks; "
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
    if (op && (op->dtype == JH_F64 || op->dtype == JH_C64)) a_flags &= ~JH_SCALAR_WIDE;   // nothing is wider than 64-bit elements (the header: "ignored there")
    if (a_flags & JH_SCALAR_COMPLEX) return jh_fail(JH_ERR_UNSUPPORTED, "jh_blockop_mul_scaled: a Complex scalar takes the unfused chain (jh_blockop_mul, jh_lincomb_typed)");
    JH_TRY(jh_enter(op, d, m));
    JH_TRY(check_vectors(op, d, m, "jh_blockop_mul_scaled"));
    if (!jh_blockop_tall_step_ok(op, d->data, m->data))
        return jh_fail(JH_ERR_UNSUPPORTED, "jh_blockop_mul_scaled: needs a tall operator of equal elementwise rows");
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
    if (op && (op->dtype == JH_F64 || op->dtype == JH_C64)) a_flags &= ~JH_SCALAR_WIDE;   // nothing is wider than 64-bit elements (the header: "ignored there")
    if (a_flags & JH_SCALAR_COMPLEX) return jh_fail(JH_ERR_UNSUPPORTED, "jh_blockop_mul_adj_scaled: a Complex scalar takes the unfused chain (jh_lincomb_typed, jh_blockop_mul_adj)");
    JH_TRY(jh_enter(op, m, d));
    JH_TRY(check_vectors(op, d, m, "jh_blockop_mul_adj_scaled"));
    if (!tall_fast_ok(op, d->data, m->data) && !(op->all_diag && tall_unaligned_ok(op, d->data, m->data))) {
        // rows of several kinds (round 5, last session): the MIXED adjoint with its input scaled on the way in -- one pass instead of the chain's two
        // (a narrow real scalar; a wide one keeps the chain)
        if (!(a_flags & JH_SCALAR_WIDE) && !(op->nonlinear && !op->pointed) &&
            (tall_mixed_ok(op, d->data, m->data) || tall_unaligned_ok(op, d->data, m->data))) {
            jh_context &c = jh_ctx();
            c.adj_in_scale = a;
            const int st = jhb::tall_adj(op, m->data, d->data, 0, true);
            c.adj_in_scale = 1.0;
            return st;
        }
        return jh_fail(JH_ERR_UNSUPPORTED, "jh_blockop_mul_adj_scaled: needs a tall operator of equal elementwise rows (a Float64 scalar on 32-bit elements: all diagonals)");
    }
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
