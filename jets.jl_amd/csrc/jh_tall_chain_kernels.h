// jh_tall_chain.hip -- JetComposite chains of ANY depth through a tall block operator in ONE pass (round 6).
//
// The reference applies a composite stage by stage, right to left, every stage into a freshly allocated zeros(range(op_i))
// (src/Jets.jl:524-540); a sum applies every term into one temporary and accumulates it (630-655); scalar * operator is one more
// stage (1159-1164).  Around a tall operator A (N block rows x 1 column, rows of any elementwise kind) every OTHER stage of such a
// chain is elementwise -- a scalar, a diagonal on the domain, a diagonal ("weights") on the range -- so the whole chain is
//     FORWARD   d_i = R(a_i .* P(m))                                   W o A o M          reads a, w; writes d
//     ADJOINT   m   = Q( sum_i conj(a_i) .* R(d_i) )                   M' o A' o W'        reads a, w, d
//     NORMAL    y   = Q( sum_i conj(a_i) .* R(a_i .* P(m)) )           M' o A' o W o A o M   reads a, w   (the weighted normal equations)
// with P / Q stage lists on the domain side (before A / after A') and R on the range side.  The unfused chain moves a range-sized
// temporary in and out per stage (A' o W o A: 8 N n s bytes against 2 N n s here).  Every stage keeps its own rounding -- a stage's
// product is formed and rounded in the element type before the next stage reads it, the row sum adds the rounded products in row
// order from +0 (1042, 1049), a zero block of A leaves the zeros of its stage's temporary (1022) -- so the result has the bits of
// the stage-by-stage chain (tests/test_gpu_chains.py against the chain on the device, the oracle and the softfloat known answers).
// `accumulate` is JetSum's `broadcast!(sgn, d, d, tmp)` (634/643/652) fused into the last stage: a term of a sum that is itself a
// chain never materialises its range- or domain-sized result.
//
// Layout in HBM as everywhere in this library: the range vector is ONE slab (block row i at element i * n), the domain vector a
// plain array; range-side diagonals are addressed through a per-row pointer table (8 bytes per row: a weight vector in one slab
// and a block-diagonal operator's separate children look the same to the kernel), domain-side ones by their base pointer.
// Every access is an under-aligned pack (jh_blockop_common.h: ldu / st_pack), so block lengths off the 16-byte grid take the
// same kernels.  HBM-bound: bytes per launch are the streamed operands once, (1 + NW) N n s (+ N n s for the ADJOINT's input or
// the FORWARD's output) + the domain-sized vectors.
#pragma once
#include "jh_blockop_common.h"

namespace {

constexpr int JH_CHAIN_MAX_STAGES = 4;   // per side
constexpr int JH_CHAIN_MAX_STREAMS = 2;  // DIAG stages per side that read a coefficient array of their own

// stage kinds as the kernels see them (0: no stage -- the lists are padded with it)
enum { CK_NONE = 0, CK_SCALE = 1, CK_SCALE_WIDE = 2, CK_DIAG = 3, CK_DIAG_CONJ = 4 };
enum : uint32_t { CK_ROWSUM = 1u << 8 };        // (stage word: kind | stream << 4 | CK_ROWSUM)

// One side's stage list, packed for the scalar unit: a stage is ONE 32-bit word (kind | stream << 4) and its scalar one float (32-bit elements) or
// double -- the range-side list lives in SGPRs for the whole row loop, beside the rows' table entries and the streams' base addresses.
struct ChainProg {
    uint32_t st[JH_CHAIN_MAX_STAGES];
    float a32[JH_CHAIN_MAX_STAGES];      // SCALE on 32-bit elements: T(a)
    uint32_t pad_[JH_CHAIN_MAX_STAGES];  // (keeps a32[3] and a[0] apart: adjacent, the vectoriser fused their loads into one <4 x float> that it then staged
                                         //  through a 20-byte stack copy of the argument -- a scratch frame nobody reads, but a scratch frame)
    double a[JH_CHAIN_MAX_STAGES];       // SCALE on 64-bit elements; WIDE: Julia's Float64 scalar against 32-bit elements
};

// THE ROW TABLE.  One record of (1 + NW) 64-bit words per block row, built when the chain is created: word 0 describes A's block of the row, words
// 1 .. NW the row's blocks of the range-side coefficient streams.  A word is a device pointer (48 bits) with its flags above it -- one s_load_dwordx2 per
// row and stream, where the operator's own table entry is 8 dwords: with four rows in flight and the next four being fetched the whole entries did not fit
// the SGPR file, and without fetching ahead the adjoint-shaped walk waited for a scalar round trip per batch (4096 x 64^3, one workgroup per CU: 3.4 TB/s).
//   word 0:   bits 48-50 the block's kind (jh_opkind), bit 51 its adjoint flag, bit 52 "a SCALE block's scalar is Real"; a SCALE row's scalar itself is read
//             from the operator's table where it is used
//   word 1+:  bit 48 the row's own conj flag (a child that is the adjoint of a diagonal), bit 49 a zero block; a null pointer is an identity row
constexpr uint64_t CR_PTR = (((uint64_t)1) << 48) - 1;
constexpr uint64_t CW_SPECIAL = ((uint64_t)1) << 49;   // a weight word's zero-block flag
__device__ inline int cr_kind(uint64_t e) { return (int)((e >> 48) & 7u); }
__device__ inline bool cr_adj(uint64_t e) { return ((e >> 51) & 1u) != 0; }
__device__ inline bool cr_real(uint64_t e) { return ((e >> 52) & 1u) != 0; }
__device__ inline bool cr_reads(uint64_t e) { const int k = cr_kind(e); return k == JH_OP_DIAG || k == JH_OP_SQUARE; }
__device__ inline bool cw_conj(uint64_t e) { return ((e >> 48) & 1u) != 0; }
__device__ inline bool cw_zero(uint64_t e) { return ((e >> 49) & 1u) != 0; }
template <typename S> __device__ inline const S *cr_ptr(uint64_t e) { return reinterpret_cast<const S *>(e & CR_PTR); }

struct ChainArgs {
    ChainProg pre, mid, post;
    const void *pre_c[JH_CHAIN_MAX_STREAMS];       // domain-sized coefficient arrays of P
    const void *post_c[JH_CHAIN_MAX_STREAMS];      // ... of Q
    const uint64_t *rows;                          // the row table: nrow records of (1 + NW) words
};

// x .= a * x for a REAL scalar: part by part (Julia's a::Real * z, src/Jets.jl:1159); WIDE: the promoted product rounded once
template <typename S, int NS, typename V> __device__ inline V stage_scale(const ChainProg &p, int s, bool wide, V x)
{
    if constexpr (sizeof(S) == 4) {
        if (wide) {
            V o;
#pragma unroll
            for (int e = 0; e < NS; e++) o[e] = (S)(p.a[s] * (double)x[e]);
            return o;
        }
        return (V)p.a32[s] * x;
    } else {
        return (V)p.a[s] * x;
    }
}

// a stage list on the DOMAIN side: the coefficient packs are loaded here (once per thread or per workgroup tile, outside the row loop).  Unrolled with
// constant stage indices: indexed by a loop variable, the by-value argument struct was copied to scratch in some Float64 shapes (36 bytes per lane).
template <typename S, int E, int NS, typename V>
__device__ inline V dom_stage(const ChainProg &p, int s, uint32_t kind, const void *c0, const void *c1, V x, int64_t sk)
{
    if (kind <= CK_SCALE_WIDE) return stage_scale<S, NS, V>(p, s, kind == CK_SCALE_WIDE, x);
    const V c = ldu<false, S, NS>((const S *)(((p.st[s] >> 4) & 15u) ? c1 : c0) + sk);   // (a select of two pointers, not a computed index into the argument struct)
    return vmul<S, E, NS, V>(c, x, kind == CK_DIAG_CONJ);
}
// (the two coefficient pointers by value: handing the kernel argument's array on by address kept a copy of the struct on the stack)
template <typename S, int E, int NS, typename V>
__device__ inline V dom_prog(const ChainProg &p, const void *c0, const void *c1, V x, int64_t sk)
{
#pragma unroll
    for (int s = 0; s < JH_CHAIN_MAX_STAGES; s++) {
        const uint32_t kind = p.st[s] & 15u;
        if (kind != CK_NONE) x = dom_stage<S, E, NS, V>(p, s, kind, c0, c1, x, sk);
    }
    return x;
}

// one RANGE-side stage on a row's pack, weight packs already loaded (wv[w]; e[1 + w]: the row's table words)
template <typename S, int E, int NS, int NW, typename V>
__device__ inline V mid_stage(const ChainProg &p, int s, uint32_t kind, V t, const V *wv, const uint64_t *e)
{
    if (kind <= CK_SCALE_WIDE) return stage_scale<S, NS, V>(p, s, kind == CK_SCALE_WIDE, t);
    if constexpr (NW > 0) {
        const bool second = NW > 1 && ((p.st[s] >> 4) & 15u) != 0;
        const uint64_t w = second ? e[NW] : e[1];
        const V c = second ? wv[NW - 1] : wv[0];
        // the children of a block-diagonal BLOCK OPERATOR (several block columns) are accumulated: `_d .+= mul!(dtmp, op, _m)` into zeros (1024), `_m .= 0` then
        // `_m .+= ...` (1042 / 1049) -- a product of -0 leaves +0.  Found by tools/fuzz_chains.py: a zero row of A under a negative scalar, then such a stage.
        const V zero_plus = (V)(S)0;
        const bool rowsum = (p.st[s] & CK_ROWSUM) != 0;
        if (__builtin_expect((w & (CW_SPECIAL | CR_PTR)) > CR_PTR || (w & CR_PTR) == 0, 0)) {      // an identity row (null pointer) or a zero block: rare
            if (cw_zero(w)) return zero_plus;                                   // a zero block on W's diagonal: the stage's zeros() stay (1022)
            if ((w & CR_PTR) == 0) return rowsum ? zero_plus + t : t;           // an identity row -- d .= m, bit for bit
        }
        V r;
        if constexpr (E == 1) r = c * t;                                        // (real elements: conj is the identity)
        else r = vmul<S, E, NS, V>(c, t, (kind == CK_DIAG_CONJ) != cw_conj(w));
        return rowsum ? zero_plus + r : r;
    }
    return t;
}

// the RANGE-side stage list of one block row.  UNROLLED on purpose: rolled (`#pragma unroll 1`: a stage = a scalar load and a branch) the family is 1 MB
// smaller and no instantiation spills an SGPR -- and it is SLOWER: Float32 A' o W o A 5.14 -> 5.35 ms (-4 %), ComplexF32 5.45 -> 8.37 ms (-35 %: the
// packs no longer stay in registers across the loop), same box, alternating (profiles/ab_r06_chain_rolled.txt).  Nested, so that a list of k stages costs
// k + 1 scalar compares, not four: with one wave per SIMD (rows of a few hundred KiB: one workgroup per CU) nothing hides the arithmetic phase of a batch,
// and a taken scalar branch is ~20 cycles -- the first version spent 20+ of them per row and pack (4096 x 64^3 A' o W o A: 3.4 TB/s).
template <typename S, int E, int NS, int NW, typename V>
__device__ inline V mid_prog(const ChainProg &p, V t, const V *wv, const uint64_t *e)
{
    const uint32_t k0 = p.st[0] & 15u;
    if (k0 == CK_NONE) return t;
    t = mid_stage<S, E, NS, NW, V>(p, 0, k0, t, wv, e);
    const uint32_t k1 = p.st[1] & 15u;
    if (k1 == CK_NONE) return t;
    t = mid_stage<S, E, NS, NW, V>(p, 1, k1, t, wv, e);
    const uint32_t k2 = p.st[2] & 15u;
    if (k2 == CK_NONE) return t;
    t = mid_stage<S, E, NS, NW, V>(p, 2, k2, t, wv, e);
    const uint32_t k3 = p.st[3] & 15u;
    if (k3 == CK_NONE) return t;
    return mid_stage<S, E, NS, NW, V>(p, 3, k3, t, wv, e);
}

template <typename S, int NS, typename V> __device__ inline V chain_accumulate(int accumulate, V found, V r)
{
    // JetSum's broadcast!(sgn, d, d, tmp) (src/Jets.jl:634): 1 / -1 continue from what the output holds, 2 / -2 the first term after `d .= 0`
    // (0 + t, 0 - t: not t, -t -- the sign of a zero)
    if (accumulate == 0) return r;
    const V base = (accumulate == 1 || accumulate == -1) ? found : (V)(S)0;
    return accumulate > 0 ? base + r : base - r;
}

// child mul! of row i on a pack (jh_blockop_common.h: apply_block_loaded, on the row's table word)
template <typename S, int E, int NS, typename V>
__device__ inline V chain_apply_row(uint64_t e, const jh_dev_block *blocks, int64_t i, V x, V c, bool transposed)
{
    const bool cj = cr_adj(e) != transposed;
    if (__builtin_expect(cr_kind(e) == JH_OP_DIAG, 1)) {                          // (the common row first: one compare)
        if constexpr (E == 1) return c * x;
        else return vmul<S, E, NS, V>(c, x, cj);
    }
    switch (cr_kind(e)) {
    case JH_OP_IDENTITY: return x;
    case JH_OP_SQUARE: return vmul<S, E, NS, V>(c + c, x, cj);
    case JH_OP_SCALE: {
        const double sre = blocks[i].sre;
        if (E == 1 || cr_real(e)) return (V)(S)sre * x;
        const double sim = blocks[i].sim;
        V a;
#pragma unroll
        for (int q = 0; q < NS; q += 2) { a[q] = (S)sre; a[q + 1] = (S)sim; }
        return vmul<S, E, NS, V>(a, x, cj);
    }
    case JH_OP_DIAG: return vmul<S, E, NS, V>(c, x, cj);
    default: return (V)(S)0;
    }
}

// ------------------------------------------------------------------ FORWARD:  d_i = R(a_i .* P(m)) -------------------------------------
// The tiling of the MIXED tall forward (jh_tall.hip: one pack per lane, `rows_per_wg` rows per workgroup, column bands of `ctiles` tiles): a
// workgroup forms P(m) for its tile once and streams its rows through it.
template <typename S, int E, int NS, bool NT, int BLK, int NW>
__global__ __launch_bounds__(BLK) void k_chain_fwd(const jh_dev_block *__restrict__ blocks, int64_t nrow, int rows_per_wg, const ChainArgs ca,
                                                   const S *__restrict__ m, S *__restrict__ d, int64_t n_scalars, unsigned ntiles,
                                                   unsigned ngroups, unsigned ctiles, int accumulate)
{
    typedef typename vec_of<S, NS>::type V;
    constexpr int NWA = NW > 0 ? NW : 1, RW = 1 + NW;
    unsigned tile, grp;
    if (ctiles) {
        const unsigned per_c = ctiles * ngroups;
        const unsigned cb = blockIdx.x / per_c;
        const unsigned r = blockIdx.x - cb * per_c;
        const unsigned cw = (cb * ctiles + ctiles <= ntiles) ? ctiles : ntiles - cb * ctiles;
        grp = r / cw;
        tile = cb * ctiles + r % cw;
    } else {
        tile = blockIdx.x % ntiles;
        grp = blockIdx.x / ntiles;
    }
    const int64_t s0 = ((int64_t)tile * BLK + threadIdx.x) * NS;
    const int64_t i0 = (int64_t)grp * rows_per_wg;
    const int64_t i1 = (i0 + rows_per_wg < nrow) ? i0 + rows_per_wg : nrow;
    const bool ok = s0 < n_scalars;
    const int64_t sk = pack_start<NS>(ok ? s0 : 0, n_scalars);
    const V pm = dom_prog<S, E, NS, V>(ca.pre, ca.pre_c[0], ca.pre_c[1], ldu<false, S, NS>(m + sk), sk);
    const bool rmw = accumulate == 1 || accumulate == -1;
    uint64_t nxt[RW];
#pragma unroll
    for (int w = 0; w < RW; w++) nxt[w] = i0 < i1 ? ca.rows[i0 * RW + w] : 0;
    for (int64_t i = i0; i < i1; i++) {
        uint64_t e[RW];
#pragma unroll
        for (int w = 0; w < RW; w++) e[w] = nxt[w];
        if (i + 1 < i1) {
#pragma unroll
            for (int w = 0; w < RW; w++) nxt[w] = ca.rows[(i + 1) * RW + w];
        }
        S *di = d + i * n_scalars;
        const V c = cr_reads(e[0]) ? ldu<NT, S, NS>(cr_ptr<S>(e[0]) + sk) : (V)(S)0;
        V wv[NWA];
#pragma unroll
        for (int w = 0; w < NWA; w++) wv[w] = (NW > 0 && (e[NW > 0 ? 1 + w : 0] & CR_PTR)) ? ldu<NT, S, NS>(cr_ptr<S>(e[NW > 0 ? 1 + w : 0]) + sk) : (V)(S)0;
        const V found = rmw ? ldu<NT, S, NS>(di + sk) : (V)(S)0;
        // a zero block of A: the stage's zeros() stay (1022), the later stages see them
        V t = cr_kind(e[0]) == JH_OP_ZERO ? (V)(S)0 : chain_apply_row<S, E, NS, V>(e[0], blocks, i, pm, c, false);
        t = mid_prog<S, E, NS, NW, V>(ca.mid, t, wv, e);
        if (ok) st_pack<true, S, NS>(di, s0, sk, chain_accumulate<S, NS, V>(accumulate, found, t));
    }
}

// ------------------------------------------------------------------ ADJOINT / NORMAL -----------------------------------------------------
// MODE 0:  out = Q( sum_i conj(a_i) .* R(d_i) )          MODE 1:  out = Q( sum_i conj(a_i) .* R(a_i .* P(in)) )
// The ordered walk of k_tall_diag_adj (jh_tall.hip): a thread owns U packs of the domain and walks all rows in order, DEPTH rows' loads in flight, the
// next DEPTH rows' table records already requested.
template <typename S, int E, int NS, int U, int DEPTH, bool NT, int MODE, int BLK, int NW>
__global__ __launch_bounds__(BLK) void k_chain_adj(const jh_dev_block *__restrict__ blocks, int64_t nrow, const ChainArgs ca, S *__restrict__ out,
                                                   const S *__restrict__ in, int64_t n_scalars, int accumulate, int64_t rows_per_part, S *__restrict__ part_out)
{
    typedef typename vec_of<S, NS>::type V;
    constexpr int NWA = NW > 0 ? NW : 1, RW = 1 + NW;
    const int64_t s0 = ((int64_t)blockIdx.x * U * BLK + threadIdx.x) * NS;
    bool ok[U];
    int64_t sk[U];
    V acc[U], mv[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        ok[k] = (s0 + (int64_t)k * BLK * NS) < n_scalars;
        sk[k] = pack_start<NS>(ok[k] ? s0 + (int64_t)k * BLK * NS : 0, n_scalars);
        acc[k] = (V)(S)0;                                                               // m .= 0 (1042)
        if (MODE == 1) mv[k] = ldu<false, S, NS>(in + sk[k]);
    }
    // (behind ONE wave-uniform test of the list's first stage, and after the loads: spelled inside the loop above, the stage list left 28 instantiations
    // with a 68-byte scratch frame that no instruction touches -- tools/kernel_resources.py)
    if (MODE == 1 && (ca.pre.st[0] & 15u) != CK_NONE) {
#pragma unroll
        for (int k = 0; k < U; k++) mv[k] = dom_prog<S, E, NS, V>(ca.pre, ca.pre_c[0], ca.pre_c[1], mv[k], sk[k]);
    }
    // one batch of D rows whose table records are in `e`: all loads, then the arithmetic, rows in order
    auto batch = [&](int64_t i, const auto &rec, auto depth_tag) {
        constexpr int D = decltype(depth_tag)::value;
        const auto &e = rec.w;
        V av[D][U], dv[D][U], wv[D][U][NWA];
        {
            // a batch of PLAIN diagonals (kind DIAG, not adjointed: almost every batch of any operator) takes the all-diagonal kernel's tight loop -- the per-row
            // kind switch of chain_apply_row was what held the bare NORMAL chain at 5.0 TB/s on rows of a few MiB where the strided all-diagonal kernel runs 6.8
            // (one workgroup per CU: nothing hides the issue slots): 256 x 2 MiB 4.98 -> 6.7-6.8, 1024 x 1 MiB 4.8 -> 7.0, 4096 x 1 MiB 5.3 -> 7.3
            // (profiles/exp_r06_chain_vs_mixed.txt).  With range-side weights: plain weight rows too (a pointer, no flag).  Same arithmetic, same order: the same bits.
            bool plain = true;
#pragma unroll
            for (int j = 0; j < D; j++) {
                plain = plain && ((e[j][0] >> 48) & 0xFu) == (uint64_t)JH_OP_DIAG;
#pragma unroll
                for (int w = 0; w < NW; w++) plain = plain && (e[j][1 + w] & CR_PTR) != 0 && (e[j][1 + w] >> 48) == 0;
            }
            if (plain) {
#pragma unroll
                for (int j = 0; j < D; j++)
#pragma unroll
                    for (int k = 0; k < U; k++) {
                        av[j][k] = ldu<NT, S, NS>(cr_ptr<S>(e[j][0]) + sk[k]);
                        if (MODE == 0) dv[j][k] = ldu<NT, S, NS>(in + (i + j) * n_scalars + sk[k]);
#pragma unroll
                        for (int w = 0; w < NW; w++) wv[j][k][w] = ldu<NT, S, NS>(cr_ptr<S>(e[j][1 + w]) + sk[k]);
                    }
#pragma unroll
                for (int j = 0; j < D; j++)
#pragma unroll
                    for (int k = 0; k < U; k++) {
                        V t = (MODE == 0) ? dv[j][k] : vmul<S, E, NS, V>(av[j][k], mv[k], false);
                        if (NW > 0 || (ca.mid.st[0] & 15u) != CK_NONE) t = mid_prog<S, E, NS, NW, V>(ca.mid, t, wv[j][k], e[j]);
                        acc[k] = acc[k] + vmul<S, E, NS, V>(av[j][k], t, true);
                    }
                return;
            }
        }
#pragma unroll
        for (int j = 0; j < D; j++) {
            const bool on = cr_kind(e[j][0]) != JH_OP_ZERO, rc = cr_reads(e[j][0]);
#pragma unroll
            for (int k = 0; k < U; k++) {
                av[j][k] = rc ? ldu<NT, S, NS>(cr_ptr<S>(e[j][0]) + sk[k]) : (V)(S)0;
                dv[j][k] = (MODE == 0 && on) ? ldu<NT, S, NS>(in + (i + j) * n_scalars + sk[k]) : (V)(S)0;
#pragma unroll
                for (int w = 0; w < NWA; w++) {
                    const uint64_t we = e[j][NW > 0 ? 1 + w : 0];
                    wv[j][k][w] = (NW > 0 && on && (we & CR_PTR)) ? ldu<NT, S, NS>(cr_ptr<S>(we) + sk[k]) : (V)(S)0;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < D; j++)
            if (cr_kind(e[j][0]) != JH_OP_ZERO) {                                       // a zero block is skipped (1047)
#pragma unroll
                for (int k = 0; k < U; k++) {
                    V t = (MODE == 0) ? dv[j][k] : chain_apply_row<S, E, NS, V>(e[j][0], blocks, i + j, mv[k], av[j][k], false);
                    t = mid_prog<S, E, NS, NW, V>(ca.mid, t, wv[j][k], e[j]);
                    acc[k] = acc[k] + chain_apply_row<S, E, NS, V>(e[j][0], blocks, i + j, t, av[j][k], true);   // _m .+= mul!(mtmp, op', _d) (1049)
                }
            }
    };
    // split-row walk (many rows of small blocks, jh_tall.hip: pick_adj_parts): workgroup row blockIdx.y sums rows [y, y + 1) * rows_per_part in order into slab y
    // of part_out (n_scalars apart); the fold and the list after A' follow in launches of their own
    int64_t i = 0;
    if (part_out) {
        i = (int64_t)blockIdx.y * rows_per_part;
        nrow = nrow < i + rows_per_part ? nrow : i + rows_per_part;
    }
    struct RecD { uint64_t w[DEPTH][RW]; };                                            // (records travel as values: handed to the lambda by pointer, some
    struct Rec1 { uint64_t w[1][RW]; };                                                //  Float64 shapes kept them in 36 bytes of scratch per lane)
    RecD nxt;
    if (i + DEPTH <= nrow) {
#pragma unroll
        for (int j = 0; j < DEPTH; j++)
#pragma unroll
            for (int w = 0; w < RW; w++) nxt.w[j][w] = ca.rows[(i + j) * RW + w];
    }
    for (; i + DEPTH <= nrow; i += DEPTH) {
        RecD e;
        const int64_t ahead = (i + 2 * DEPTH <= nrow) ? i + DEPTH : i;                  // (the last full batch re-reads its own records)
#pragma unroll
        for (int j = 0; j < DEPTH; j++)
#pragma unroll
            for (int w = 0; w < RW; w++) {
                e.w[j][w] = nxt.w[j][w];
                nxt.w[j][w] = ca.rows[(ahead + j) * RW + w];
            }
        batch(i, e, std::integral_constant<int, DEPTH>{});
    }
    for (; i < nrow; i++) {
        Rec1 e;
#pragma unroll
        for (int w = 0; w < RW; w++) e.w[0][w] = ca.rows[i * RW + w];
        batch(i, e, std::integral_constant<int, 1>{});
    }
    if (part_out) {
        S *slab = part_out + (int64_t)blockIdx.y * n_scalars;
#pragma unroll
        for (int k = 0; k < U; k++)
            if (ok[k]) st_pack<false, S, NS>(slab, s0 + (int64_t)k * BLK * NS, sk[k], acc[k]);
        return;
    }
    const bool rmw = accumulate == 1 || accumulate == -1;
#pragma unroll
    for (int k = 0; k < U; k++) {
        const V found = rmw ? ldu<false, S, NS>(out + sk[k]) : (V)(S)0;
        const V r = dom_prog<S, E, NS, V>(ca.post, ca.post_c[0], ca.post_c[1], acc[k], sk[k]);
        if (ok[k]) st_pack<false, S, NS>(out, s0 + (int64_t)k * BLK * NS, sk[k], chain_accumulate<S, NS, V>(accumulate, found, r));
    }
}

// The split walk's last step when the chain goes on after A' (or accumulates into what `out` holds): out = accumulate(out, Q(folded)).
template <typename S, int E, int NS>
__global__ __launch_bounds__(256) void k_chain_finish(const ChainArgs ca, S *__restrict__ out, const S *__restrict__ folded, int64_t n_scalars, int accumulate)
{
    typedef typename vec_of<S, NS>::type V;
    const int64_t s = ((int64_t)blockIdx.x * 256 + threadIdx.x) * NS;
    if (s >= n_scalars) return;
    const int64_t sc = pack_start<NS>(s, n_scalars);
    const bool rmw = accumulate == 1 || accumulate == -1;
    const V found = rmw ? ldu<false, S, NS>(out + sc) : (V)(S)0;
    const V r = dom_prog<S, E, NS, V>(ca.post, ca.post_c[0], ca.post_c[1], ldu<false, S, NS>(folded + sc), sc);
    st_pack<false, S, NS>(out, s, sc, chain_accumulate<S, NS, V>(accumulate, found, r));
}

}  // namespace

// ------------------------------------------------------------------ the handle -----------------------------------------------------------
struct jh_chain {
    int ctx = -1;
    const jh_blockop *op = nullptr;          // borrowed: must outlive the chain
    int type = 0;
    ChainArgs args{};
    int nw = 0;                              // range-side coefficient streams
    uint64_t *dev_tab = nullptr;             // the row table: nrow records of (1 + nw) words
    std::vector<uint64_t> host_tab;          // its host copy (word 0 of every record is rebuilt when the operator is pointed again: jh_blockop_point moves
    int64_t op_gen = -1;                     //  the SQUARE rows' arrays) and the operator's table generation it was built for
    bool coeff16 = true;                     // every coefficient array of the stages on the 16-byte grid
    double stream_bytes = 0;                 // N n s (1 + nw): what one pass streams besides the vectors
};

namespace {

template <typename S, int E, int NS>
int launch_chain_fwd(const jh_chain *ch, void *d, const void *m, int64_t n_scalars, int accumulate)
{
    jh_context &c = jh_ctx();
    const jh_blockop *op = ch->op;
    constexpr int BLK = 256;
    const int64_t row_bytes = n_scalars * (int64_t)sizeof(S);
    int64_t G = c.fwd_group > 0 ? c.fwd_group : (row_bytes <= 2560 ? 8 : (row_bytes <= 5120 ? 4 : 2));
    if (G > op->nrow) G = op->nrow;
    const int64_t gx = (n_scalars + (int64_t)BLK * NS - 1) / ((int64_t)BLK * NS);
    int64_t gy = (op->nrow + G - 1) / G;
    while (gx * gy * BLK >= ((int64_t)1 << 32) && G < op->nrow) { G *= 2; gy = (op->nrow + G - 1) / G; }
    JH_REQUIRE(gx * gy * BLK < (int64_t)1 << 32, "chain forward: grid of %lld workgroups is too large", (long long)(gx * gy));
    int64_t ctiles = c.fwd_ctiles >= 0 ? c.fwd_ctiles : 32;
    if (ctiles > gx) ctiles = gx;
    const bool off_grid = row_bytes % 16 != 0 || !op->coeff_aligned16 || !ch->coeff16 || ((((uintptr_t)d) | ((uintptr_t)m)) & 15u) != 0;
    const bool nt = jh_stream_nt(ch->stream_bytes + (double)op->nrow * (double)row_bytes) && !(c.ua_nt == 0 || (c.ua_nt < 0 && off_grid));
#define JH_CHAIN_FWD(NTV, NWV)                                                                                                              \
    hipLaunchKernelGGL((k_chain_fwd<S, E, NS, NTV, BLK, NWV>), dim3((unsigned)(gx * gy)), dim3(BLK), 0, c.stream, op->dev_blocks, op->nrow, (int)G, \
                       ch->args, (const S *)m, (S *)d, n_scalars, (unsigned)gx, (unsigned)gy, (unsigned)ctiles, accumulate)
    switch (ch->nw) {
    case 0: if (nt) JH_CHAIN_FWD(true, 0); else JH_CHAIN_FWD(false, 0); break;
    case 1: if (nt) JH_CHAIN_FWD(true, 1); else JH_CHAIN_FWD(false, 1); break;
    default: if (nt) JH_CHAIN_FWD(true, 2); else JH_CHAIN_FWD(false, 2); break;
    }
#undef JH_CHAIN_FWD
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

template <typename S, int E, int NS, int MODE>
int launch_chain_adj(const jh_chain *ch, void *out, const void *in, int64_t n_scalars, int accumulate)
{
    jh_context &c = jh_ctx();
    const jh_blockop *op = ch->op;
    const int64_t packs = (n_scalars + NS - 1) / NS;
    const int64_t row_bytes = n_scalars * (int64_t)sizeof(S);
    const bool off_grid = row_bytes % 16 != 0 || !op->coeff_aligned16 || !ch->coeff16 || ((((uintptr_t)out) | ((uintptr_t)in)) & 15u) != 0;
    const double streamed = ch->stream_bytes + (MODE == 0 ? (double)op->nrow * (double)row_bytes : 0.0);
    const bool nt = jh_stream_nt(streamed) && !(c.ua_nt == 0 || (c.ua_nt < 0 && off_grid && row_bytes >= ((int64_t)32 << 20)));
    // shapes (lanes x packs per lane x rows in flight): thin workgroups (256 x 1 x 4) for rows of a few KiB, 512 x 2 x 2 in between, fat ones (512 x 4 x 2) once
    // a row holds >= 256 K packs (4 MiB of Float32: the all-diagonal adjoint's rule, jh_tall.hip: pick_adj_shape); the streams in flight per row are
    // 1 + NW (+ 1 for the ADJOINT's input)
    // (same box, A' o W o A: 1024 x 128^3 thin 6.10 / 512 x 2 x 2 5.65 / fat 6.37 TB/s; 256 x 256^3 within 1 % of each other; profiles/ab_r06_chain_shapes.txt)
    static const int64_t per_wg_of[3] = {256, 1024, 2048};
    int shape = packs < 2048 ? 0 : (packs >= ((int64_t)1 << 18) ? 2 : 1);
    while (shape > 0 && (packs + per_wg_of[shape] - 1) / per_wg_of[shape] < c.cu_count) shape--;   // a workgroup per CU at least, if the rows are long enough for it
    if (c.adj_wg == 256) shape = 0; else if (c.adj_wg == 512 && c.adj_unroll == 4) shape = 2; else if (c.adj_wg == 512) shape = 1;
    const int64_t per_wg = per_wg_of[shape];
    const int64_t gx = (packs + per_wg - 1) / per_wg;
    // many rows of small blocks: the split-row walk (jh_tall.hip: pick_adj_parts; adj_split = 0 keeps the ordered, bit-exact walk) -- parts of the row sum into
    // slabs of the scratch buffer, the fold of k_fold_parts, then the stages after A' and the accumulation on the folded vector (k_chain_finish; folded
    // straight into `out` when there is neither).  Tolerance parity, like every split walk.
    int64_t parts = n_scalars < NS ? 1 : jhb::pick_adj_parts(gx, op->nrow), rows_per_part = 0;
    // (one workgroup per CU, up to two: the chain's three or four streams per row leave the ordered walk latency-bound there -- 4096 x 64^3, 256 workgroups:
    // A' o W o A 4.37 TB/s in one part, 7.04 in two, 6.3 in four or more; profiles/bench_chains_r06_split.txt)
    if (parts == 1 && c.adj_split < 0 && n_scalars >= NS && op->nrow >= 256 && gx < 2 * (int64_t)c.cu_count) parts = 2;
    const bool finish = accumulate != 0 || (ch->args.post.st[0] & 15u) != CK_NONE;
    S *slabs = nullptr, *folded = (S *)out;
    if (parts > 1) {
        rows_per_part = (op->nrow + parts - 1) / parts;
        parts = (op->nrow + rows_per_part - 1) / rows_per_part;
        void *sp = nullptr;
        JH_TRY(jhb::split_slabs(out, (size_t)(parts + (finish ? 1 : 0)) * (size_t)n_scalars * sizeof(S), &sp));
        slabs = (S *)sp;
        if (finish) folded = slabs + parts * n_scalars;
    }
    c.last_adj_parts = parts;
#define JH_CHAIN_ADJ(BLKV, UV, DV, NTV, NWV)                                                                                                 \
    hipLaunchKernelGGL((k_chain_adj<S, E, NS, UV, DV, NTV, MODE, BLKV, NWV>), dim3((unsigned)gx, (unsigned)parts), dim3(BLKV), 0, c.stream, op->dev_blocks, \
                       op->nrow, ch->args, (S *)out, (const S *)in, n_scalars, accumulate, rows_per_part, slabs)
#define JH_CHAIN_ADJ_NW(BLKV, UV, DV, NTV)                                                                                                    \
    switch (ch->nw) {                                                                                                                      \
    case 0: JH_CHAIN_ADJ(BLKV, UV, DV, NTV, 0); break;                                                                                     \
    case 1: JH_CHAIN_ADJ(BLKV, UV, DV, NTV, 1); break;                                                                                     \
    default: JH_CHAIN_ADJ(BLKV, UV, DV, NTV, 2); break;                                                                                    \
    }
#define JH_CHAIN_ADJ_SHAPE(NTV)                                                                                                              \
    /* (rows of a few hundred KiB at most, one workgroup per CU or fewer: eight rows in flight instead of four bought nothing -- 4096 x 64^3 3.79 -> 3.73   \
       TB/s -- and their sixteen row records pushed the SGPR spills past what fits the lanes of the spill registers) */                             \
    if (shape == 0) { JH_CHAIN_ADJ_NW(256, 1, 4, NTV) }                                                                                    \
    else if (shape == 1) { JH_CHAIN_ADJ_NW(512, 2, 2, NTV) }                                                                               \
    else { JH_CHAIN_ADJ_NW(512, 4, 2, NTV) }
    if (nt) { JH_CHAIN_ADJ_SHAPE(true) } else { JH_CHAIN_ADJ_SHAPE(false) }
#undef JH_CHAIN_ADJ_SHAPE
#undef JH_CHAIN_ADJ_NW
#undef JH_CHAIN_ADJ
    JH_CHECK_HIP(hipGetLastError());
    if (parts > 1) {
        JH_TRY(jhb::fold_parts(sizeof(S) == 4 ? JH_F32 : JH_F64, slabs, n_scalars, parts, folded, 0, n_scalars));
        if (finish) {
            hipLaunchKernelGGL((k_chain_finish<S, E, NS>), dim3((unsigned)((packs + 255) / 256)), dim3(256), 0, c.stream, ch->args, (S *)out, (const S *)folded,
                               n_scalars, accumulate);
            JH_CHECK_HIP(hipGetLastError());
        }
    }
    return JH_OK;
}

}  // namespace
