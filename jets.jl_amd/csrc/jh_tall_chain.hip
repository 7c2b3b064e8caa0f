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
#include "jh_blockop_common.h"

namespace {

constexpr int JH_CHAIN_MAX_STAGES = 4;   // per side
constexpr int JH_CHAIN_MAX_STREAMS = 2;  // DIAG stages per side that read a coefficient array of their own

// stage kinds as the kernels see them (0: no stage -- the lists are padded with it)
enum { CK_NONE = 0, CK_SCALE = 1, CK_SCALE_WIDE = 2, CK_DIAG = 3, CK_DIAG_CONJ = 4 };

// One side's stage list, packed for the scalar unit: a stage is ONE 32-bit word (kind | stream << 4) and its scalar one float (32-bit elements) or
// double -- the range-side list lives in SGPRs for the whole row loop, beside the rows' table entries and the streams' base addresses.
struct ChainProg {
    uint32_t st[JH_CHAIN_MAX_STAGES];
    float a32[JH_CHAIN_MAX_STAGES];      // SCALE on 32-bit elements: T(a)
    double a[JH_CHAIN_MAX_STAGES];       // SCALE on 64-bit elements; WIDE: Julia's Float64 scalar against 32-bit elements
};

struct ChainArgs {
    ChainProg pre, mid, post;
    const void *pre_c[JH_CHAIN_MAX_STREAMS];       // domain-sized coefficient arrays of P
    const void *post_c[JH_CHAIN_MAX_STREAMS];      // ... of Q
    const uint64_t *wtab[JH_CHAIN_MAX_STREAMS];    // R: per block row, the row's coefficient pointer | bit 0: the row's own conj flag | bit 1: a zero block
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

// a stage list on the DOMAIN side: the coefficient packs are loaded here (once per thread or per workgroup tile, outside the row loop)
template <typename S, int E, int NS, typename V>
__device__ inline V dom_prog(const ChainProg &p, const void *const *coef, V x, int64_t sk)
{
#pragma unroll
    for (int s = 0; s < JH_CHAIN_MAX_STAGES; s++) {
        const uint32_t kind = p.st[s] & 15u;
        if (kind == CK_SCALE || kind == CK_SCALE_WIDE) x = stage_scale<S, NS, V>(p, s, kind == CK_SCALE_WIDE, x);
        else if (kind != CK_NONE) {
            const V c = ldu<false, S, NS>((const S *)coef[p.st[s] >> 4] + sk);
            x = vmul<S, E, NS, V>(c, x, kind == CK_DIAG_CONJ);
        }
    }
    return x;
}

// the RANGE-side stage list of one block row, weight packs already loaded (wv[w]; we[w]: the rows' table entries)
template <typename S, int E, int NS, int NW, typename V>
__device__ inline V mid_prog(const ChainProg &p, V t, const V *wv, const uint64_t *we)
{
#pragma unroll
    for (int s = 0; s < JH_CHAIN_MAX_STAGES; s++) {
        const uint32_t kind = p.st[s] & 15u;
        if (kind == CK_SCALE || kind == CK_SCALE_WIDE) t = stage_scale<S, NS, V>(p, s, kind == CK_SCALE_WIDE, t);
        else if (kind != CK_NONE) {
            if constexpr (NW > 0) {
                const bool second = NW > 1 && (p.st[s] >> 4) != 0;
                const uint64_t e = second ? we[NW - 1] : we[0];
                const V c = second ? wv[NW - 1] : wv[0];
                if (e & 2u) t = (V)(S)0;                                        // a zero block on W's diagonal: the stage's zeros() stay (1022)
                else if (e & ~(uint64_t)3) t = vmul<S, E, NS, V>(c, t, (kind == CK_DIAG_CONJ) != ((e & 1u) != 0));
                // (a null pointer: an identity row -- d .= m, bit for bit)
            }
        }
    }
    return t;
}

template <typename S, int NS, typename V> __device__ inline V chain_accumulate(int accumulate, V found, V r)
{
    // JetSum's broadcast!(sgn, d, d, tmp) (src/Jets.jl:634): 1 / -1 continue from what the output holds, 2 / -2 the first term after `d .= 0`
    // (0 + t, 0 - t: not t, -t -- the sign of a zero)
    if (accumulate == 0) return r;
    const V base = (accumulate == 1 || accumulate == -1) ? found : (V)(S)0;
    return accumulate > 0 ? base + r : base - r;
}

// A row of A as the kernels hold it: 4 dwords of its 32-byte table entry (the coefficient pointer, the kind word, the adjoint flag); a SCALE
// row's scalar is fetched where it is used.  (jh_dev_block is 8 dwords: with four rows in flight and the weights' entries beside them the
// whole entries did not fit the SGPR file.)
struct ChainRow {
    const void *coeff;
    uint32_t kw;                         // jh_dev_block's bit-field word: kind (low 16 bits, signed), real_scale (high 16)
    int32_t adjoint;
};
__device__ inline ChainRow chain_row(const jh_dev_block *blocks, int64_t i)
{
    const uint64_t *p = reinterpret_cast<const uint64_t *>(blocks + i);
    ChainRow r;
    r.coeff = reinterpret_cast<const void *>(p[0]);
    const uint64_t w = p[3];
    r.kw = (uint32_t)w;
    r.adjoint = (int32_t)(w >> 32);
    return r;
}
__device__ inline int chain_row_kind(const ChainRow &r) { return (int)(int16_t)(r.kw & 0xffffu); }
__device__ inline bool chain_row_reads(const ChainRow &r) { const int k = chain_row_kind(r); return k == JH_OP_DIAG || k == JH_OP_SQUARE; }

// child mul! of row i on a pack (jh_blockop_common.h: apply_block_loaded, on the 4-dword row)
template <typename S, int E, int NS, typename V>
__device__ inline V chain_apply_row(const ChainRow &r, const jh_dev_block *blocks, int64_t i, V x, V c, bool transposed)
{
    const bool cj = (r.adjoint != 0) != transposed;
    switch (chain_row_kind(r)) {
    case JH_OP_IDENTITY: return x;
    case JH_OP_SQUARE: return vmul<S, E, NS, V>(c + c, x, cj);
    case JH_OP_SCALE: {
        const double sre = blocks[i].sre;
        if (E == 1 || (r.kw >> 16) != 0) return (V)(S)sre * x;
        const double sim = blocks[i].sim;
        V a;
#pragma unroll
        for (int e = 0; e < NS; e += 2) { a[e] = (S)sre; a[e + 1] = (S)sim; }
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
    constexpr int NWA = NW > 0 ? NW : 1;
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
    const V pm = dom_prog<S, E, NS, V>(ca.pre, ca.pre_c, ldu<false, S, NS>(m + sk), sk);
    const bool rmw = accumulate == 1 || accumulate == -1;
    ChainRow nxt{};
    uint64_t wnxt[NWA] = {};
    if (i0 < i1) {
        nxt = chain_row(blocks, i0);
#pragma unroll
        for (int w = 0; w < NW; w++) wnxt[w] = ca.wtab[w][i0];
    }
    for (int64_t i = i0; i < i1; i++) {
        const ChainRow row = nxt;
        uint64_t we[NWA];
#pragma unroll
        for (int w = 0; w < NWA; w++) we[w] = wnxt[w];
        if (i + 1 < i1) {
            nxt = chain_row(blocks, i + 1);
#pragma unroll
            for (int w = 0; w < NW; w++) wnxt[w] = ca.wtab[w][i + 1];
        }
        S *di = d + i * n_scalars;
        const V c = chain_row_reads(row) ? ldu<NT, S, NS>((const S *)row.coeff + sk) : (V)(S)0;
        V wv[NWA];
#pragma unroll
        for (int w = 0; w < NWA; w++) wv[w] = (NW > 0 && (we[w] & ~(uint64_t)3)) ? ldu<NT, S, NS>((const S *)(we[w] & ~(uint64_t)3) + sk) : (V)(S)0;
        const V found = rmw ? ldu<NT, S, NS>(di + sk) : (V)(S)0;
        // a zero block of A: the stage's zeros() stay (1022), the later stages see them
        V t = chain_row_kind(row) == JH_OP_ZERO ? (V)(S)0 : chain_apply_row<S, E, NS, V>(row, blocks, i, pm, c, false);
        t = mid_prog<S, E, NS, NW, V>(ca.mid, t, wv, we);
        if (ok) st_pack<true, S, NS>(di, s0, sk, chain_accumulate<S, NS, V>(accumulate, found, t));
    }
}

// ------------------------------------------------------------------ ADJOINT / NORMAL -----------------------------------------------------
// MODE 0:  out = Q( sum_i conj(a_i) .* R(d_i) )          MODE 1:  out = Q( sum_i conj(a_i) .* R(a_i .* P(in)) )
// The ordered walk of k_tall_diag_adj (jh_tall.hip): a thread owns U packs of the domain and walks all rows in order, DEPTH rows' loads in flight.
template <typename S, int E, int NS, int U, int DEPTH, bool NT, int MODE, int BLK, int NW>
__global__ __launch_bounds__(BLK) void k_chain_adj(const jh_dev_block *__restrict__ blocks, int64_t nrow, const ChainArgs ca, S *__restrict__ out,
                                                   const S *__restrict__ in, int64_t n_scalars, int accumulate)
{
    typedef typename vec_of<S, NS>::type V;
    constexpr int NWA = NW > 0 ? NW : 1;
    const int64_t s0 = ((int64_t)blockIdx.x * U * BLK + threadIdx.x) * NS;
    bool ok[U];
    int64_t sk[U];
    V acc[U], mv[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        ok[k] = (s0 + (int64_t)k * BLK * NS) < n_scalars;
        sk[k] = pack_start<NS>(ok[k] ? s0 + (int64_t)k * BLK * NS : 0, n_scalars);
        acc[k] = (V)(S)0;                                                               // m .= 0 (1042)
        if (MODE == 1) mv[k] = dom_prog<S, E, NS, V>(ca.pre, ca.pre_c, ldu<false, S, NS>(in + sk[k]), sk[k]);
    }
    auto batch = [&](int64_t i, auto depth_tag) {
        constexpr int D = decltype(depth_tag)::value;
        ChainRow row[D];
        uint64_t we[D][NWA];
#pragma unroll
        for (int j = 0; j < D; j++) {
            row[j] = chain_row(blocks, i + j);
#pragma unroll
            for (int w = 0; w < NWA; w++) we[j][w] = NW > 0 ? ca.wtab[w][i + j] : 0;
        }
        V av[D][U], dv[D][U], wv[D][U][NWA];
#pragma unroll
        for (int j = 0; j < D; j++) {
            const bool on = chain_row_kind(row[j]) != JH_OP_ZERO, rc = chain_row_reads(row[j]);
#pragma unroll
            for (int k = 0; k < U; k++) {
                av[j][k] = rc ? ldu<NT, S, NS>((const S *)row[j].coeff + sk[k]) : (V)(S)0;
                dv[j][k] = (MODE == 0 && on) ? ldu<NT, S, NS>(in + (i + j) * n_scalars + sk[k]) : (V)(S)0;
#pragma unroll
                for (int w = 0; w < NWA; w++)
                    wv[j][k][w] = (NW > 0 && on && (we[j][w] & ~(uint64_t)3)) ? ldu<NT, S, NS>((const S *)(we[j][w] & ~(uint64_t)3) + sk[k]) : (V)(S)0;
            }
        }
#pragma unroll
        for (int j = 0; j < D; j++)
            if (chain_row_kind(row[j]) != JH_OP_ZERO) {                                 // a zero block is skipped (1047)
#pragma unroll
                for (int k = 0; k < U; k++) {
                    V t = (MODE == 0) ? dv[j][k] : chain_apply_row<S, E, NS, V>(row[j], blocks, i + j, mv[k], av[j][k], false);
                    t = mid_prog<S, E, NS, NW, V>(ca.mid, t, wv[j][k], we[j]);
                    acc[k] = acc[k] + chain_apply_row<S, E, NS, V>(row[j], blocks, i + j, t, av[j][k], true);   // _m .+= mul!(mtmp, op', _d) (1049)
                }
            }
    };
    int64_t i = 0;
    for (; i + DEPTH <= nrow; i += DEPTH) batch(i, std::integral_constant<int, DEPTH>{});
    for (; i < nrow; i++) batch(i, std::integral_constant<int, 1>{});
    const bool rmw = accumulate == 1 || accumulate == -1;
#pragma unroll
    for (int k = 0; k < U; k++) {
        const V found = rmw ? ldu<false, S, NS>(out + sk[k]) : (V)(S)0;
        const V r = dom_prog<S, E, NS, V>(ca.post, ca.post_c, acc[k], sk[k]);
        if (ok[k]) st_pack<false, S, NS>(out, s0 + (int64_t)k * BLK * NS, sk[k], chain_accumulate<S, NS, V>(accumulate, found, r));
    }
}

}  // namespace

// ------------------------------------------------------------------ the handle -----------------------------------------------------------
struct jh_chain {
    int ctx = -1;
    const jh_blockop *op = nullptr;          // borrowed: must outlive the chain
    int type = 0;
    ChainArgs args{};
    int nw = 0;                              // range-side coefficient streams
    uint64_t *dev_tab = nullptr;             // nw * nrow table entries
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
    // shapes (lanes x packs per lane x rows in flight): thin workgroups for rows of a few KiB, 512 x 2 x 2 in between, fat ones once a row holds >= 4 M packs
    // (the all-diagonal adjoint's rule, jh_tall.hip: pick_adj_shape); the streams in flight per row are 1 + NW (+ 1 for the ADJOINT's input)
    int shape = packs < 2048 ? 0 : (packs >= ((int64_t)1 << 22) ? 2 : 1);
    if (c.adj_wg == 256) shape = 0; else if (c.adj_wg == 512 && c.adj_unroll == 4) shape = 2; else if (c.adj_wg == 512) shape = 1;
    const int64_t per_wg = shape == 0 ? 256 : (shape == 1 ? 1024 : 2048);
    const int64_t gx = (packs + per_wg - 1) / per_wg;
    // many rows of small blocks want the split-row walk (jh_tall.hip: pick_adj_parts): not built for chains -- the caller takes the stage-by-stage chain
    if (jhb::pick_adj_parts(gx, op->nrow) > 1)
        return jh_fail(JH_ERR_UNSUPPORTED, "fused chain: %lld rows of %lld-byte blocks want the split-row walk; apply the chain stage by stage",
                       (long long)op->nrow, (long long)row_bytes);
    c.last_adj_parts = 1;
#define JH_CHAIN_ADJ(BLKV, UV, DV, NTV, NWV)                                                                                                 \
    hipLaunchKernelGGL((k_chain_adj<S, E, NS, UV, DV, NTV, MODE, BLKV, NWV>), dim3((unsigned)gx), dim3(BLKV), 0, c.stream, op->dev_blocks, op->nrow, \
                       ch->args, (S *)out, (const S *)in, n_scalars, accumulate)
#define JH_CHAIN_ADJ_NW(BLKV, UV, DV, NTV)                                                                                                    \
    switch (ch->nw) {                                                                                                                      \
    case 0: JH_CHAIN_ADJ(BLKV, UV, DV, NTV, 0); break;                                                                                     \
    case 1: JH_CHAIN_ADJ(BLKV, UV, DV, NTV, 1); break;                                                                                     \
    default: JH_CHAIN_ADJ(BLKV, UV, DV, NTV, 2); break;                                                                                    \
    }
#define JH_CHAIN_ADJ_SHAPE(NTV)                                                                                                              \
    if (shape == 0) { JH_CHAIN_ADJ_NW(256, 1, 4, NTV) }                                                                                    \
    else if (shape == 1) { JH_CHAIN_ADJ_NW(512, 2, 2, NTV) }                                                                               \
    else { JH_CHAIN_ADJ_NW(512, 4, 2, NTV) }
    if (nt) { JH_CHAIN_ADJ_SHAPE(true) } else { JH_CHAIN_ADJ_SHAPE(false) }
#undef JH_CHAIN_ADJ_SHAPE
#undef JH_CHAIN_ADJ_NW
#undef JH_CHAIN_ADJ
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

// one side's stages -> its program; DIAG stages get a stream each unless they name an array the side already streams
int build_prog(const char *side, int n, const jh_chain_stage *st, int dtype, int64_t nrow_ptrs, ChainProg &p, std::vector<const jh_chain_stage *> &streams)
{
    JH_REQUIRE(n >= 0 && n <= JH_CHAIN_MAX_STAGES, "jh_chain_create: %d %s stages (at most %d)", n, side, JH_CHAIN_MAX_STAGES);
    JH_REQUIRE(n == 0 || st, "jh_chain_create: null %s stage list", side);
    const bool wide_ok = dtype == JH_F32 || dtype == JH_C32;
    for (int s = 0; s < JH_CHAIN_MAX_STAGES; s++) { p.st[s] = CK_NONE; p.a32[s] = 0.f; p.a[s] = 0.0; }
    for (int s = 0; s < n; s++) {
        const jh_chain_stage &g = st[s];
        if (g.kind == JH_STAGE_SCALE) {
            JH_REQUIRE((g.flags & ~(JH_SCALAR_COMPLEX | JH_SCALAR_WIDE)) == 0, "jh_chain_create: unknown flags %d on a SCALE stage", g.flags);
            if (g.flags & JH_SCALAR_COMPLEX) return jh_fail(JH_ERR_UNSUPPORTED, "jh_chain_create: a Complex scalar takes the stage-by-stage chain");
            const bool wide = (g.flags & JH_SCALAR_WIDE) && wide_ok;
            p.st[s] = wide ? CK_SCALE_WIDE : CK_SCALE;
            p.a[s] = g.a;
            p.a32[s] = (float)g.a;
        } else if (g.kind == JH_STAGE_DIAG) {
            JH_REQUIRE((g.flags & ~JH_STAGE_CONJ) == 0, "jh_chain_create: unknown flags %d on a DIAG stage", g.flags);
            JH_REQUIRE(g.coeff, "jh_chain_create: a DIAG stage without coefficient pointers");
            int found = -1;
            for (size_t q = 0; q < streams.size() && found < 0; q++) {
                bool same = (streams[q]->row_flags == nullptr) == (g.row_flags == nullptr);
                for (int64_t i = 0; i < nrow_ptrs && same; i++)
                    same = streams[q]->coeff[i] == g.coeff[i] && (!g.row_flags || streams[q]->row_flags[i] == g.row_flags[i]);
                if (same) found = (int)q;
            }
            if (found < 0) {
                if ((int)streams.size() >= JH_CHAIN_MAX_STREAMS)
                    return jh_fail(JH_ERR_UNSUPPORTED, "jh_chain_create: more than %d coefficient arrays on the %s side", JH_CHAIN_MAX_STREAMS, side);
                found = (int)streams.size();
                streams.push_back(&g);
            }
            p.st[s] = (uint32_t)((g.flags & JH_STAGE_CONJ) ? CK_DIAG_CONJ : CK_DIAG) | ((uint32_t)found << 4);
        } else {
            return jh_fail(JH_ERR_INVALID, "jh_chain_create: unknown stage kind %d", g.kind);
        }
    }
    return JH_OK;
}

}  // namespace

extern "C" {

int jh_chain_create(const jh_blockop *op, int type, int npre, const jh_chain_stage *pre, int nmid, const jh_chain_stage *mid, int npost,
                    const jh_chain_stage *post, jh_chain **out)
{
    JH_TRY(jh_enter(op));
    JH_REQUIRE(op && out, "jh_chain_create: null argument");
    JH_REQUIRE(type == JH_CHAIN_FORWARD || type == JH_CHAIN_ADJOINT || type == JH_CHAIN_NORMAL, "jh_chain_create: unknown chain type %d", type);
    JH_REQUIRE(!(type == JH_CHAIN_FORWARD && npost) && !(type == JH_CHAIN_ADJOINT && npre),
               "jh_chain_create: a FORWARD chain has no stages after A', an ADJOINT chain none before A");
    if (!(op->tall && op->uniform_rows && op->elementwise) || op->nrow < 2 || op->row_len[0] * (int64_t)jh_dtype_size(op->dtype) < 16)
        return jh_fail(JH_ERR_UNSUPPORTED, "jh_chain_create: needs a tall operator of >= 2 equal elementwise rows of at least 16 bytes");
    const size_t es = jh_dtype_size(op->dtype);
    const size_t sa = jh_dtype_complex(op->dtype) ? es / 2 : es;
    jh_chain *ch = new jh_chain();
    ch->ctx = op->ctx;
    ch->op = op;
    ch->type = type;
    std::vector<const jh_chain_stage *> s_pre, s_mid, s_post;
    int st = build_prog("domain-side (before A)", npre, pre, op->dtype, 1, ch->args.pre, s_pre);
    if (st == JH_OK) st = build_prog("range-side", nmid, mid, op->dtype, op->nrow, ch->args.mid, s_mid);
    if (st == JH_OK) st = build_prog("domain-side (after A')", npost, post, op->dtype, 1, ch->args.post, s_post);
    if (st != JH_OK) { delete ch; return st; }
    bool scalar_aligned = true;
    auto note = [&](const void *p) {
        if (((uintptr_t)p) & 15u) ch->coeff16 = false;
        if (((uintptr_t)p) & (sa - 1)) scalar_aligned = false;
    };
    for (size_t q = 0; q < s_pre.size(); q++) { ch->args.pre_c[q] = s_pre[q]->coeff[0]; note(s_pre[q]->coeff[0]); }
    for (size_t q = 0; q < s_post.size(); q++) { ch->args.post_c[q] = s_post[q]->coeff[0]; note(s_post[q]->coeff[0]); }
    ch->nw = (int)s_mid.size();
    if (ch->nw) {
        std::vector<uint64_t> host((size_t)ch->nw * (size_t)op->nrow);
        for (int w = 0; w < ch->nw; w++)
            for (int64_t i = 0; i < op->nrow; i++) {
                const void *p = s_mid[(size_t)w]->coeff[i];
                const unsigned fl = s_mid[(size_t)w]->row_flags ? (s_mid[(size_t)w]->row_flags[i] & 3u) : 0u;
                note(p);
                host[(size_t)w * (size_t)op->nrow + (size_t)i] = (uint64_t)(uintptr_t)p | fl;
            }
        if (!scalar_aligned) { delete ch; return jh_fail(JH_ERR_UNSUPPORTED, "jh_chain_create: a coefficient array is not aligned like its scalar"); }
        hipError_t e = jh_device_malloc(jh_ctx().device, (void **)&ch->dev_tab, host.size() * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMemcpyAsync(ch->dev_tab, host.data(), host.size() * sizeof(uint64_t), hipMemcpyHostToDevice, jh_ctx().stream);
        if (e == hipSuccess) e = hipStreamSynchronize(jh_ctx().stream);
        if (e != hipSuccess) {
            if (ch->dev_tab) (void)hipFree(ch->dev_tab);
            delete ch;
            return jh_fail(e == hipErrorOutOfMemory ? JH_ERR_NOMEM : JH_ERR_HIP, "jh_chain_create: %s", hipGetErrorString(e));
        }
        for (int w = 0; w < ch->nw; w++) ch->args.wtab[w] = ch->dev_tab + (size_t)w * (size_t)op->nrow;
    }
    if (!scalar_aligned) { if (ch->dev_tab) (void)hipFree(ch->dev_tab); delete ch; return jh_fail(JH_ERR_UNSUPPORTED, "jh_chain_create: a coefficient array is not aligned like its scalar"); }
    ch->stream_bytes = (double)op->nrow * (double)op->row_len[0] * (double)es * (double)(1 + ch->nw);
    jh_handle_born(ch->ctx);
    *out = ch;
    return JH_OK;
}

int jh_chain_destroy(jh_chain *ch)
{
    if (!ch) return JH_OK;
    jh_quiesce_scope quiet(ch->ctx);
    if (ch->dev_tab) (void)hipFree(ch->dev_tab);
    jh_handle_died(ch->ctx);
    delete ch;
    return JH_OK;
}

int jh_chain_apply(const jh_chain *ch, jh_bvec *out, const jh_bvec *x, int accumulate)
{
    JH_REQUIRE(ch && out && x, "jh_chain_apply: null argument");
    const jh_blockop *op = ch->op;
    JH_TRY(jh_enter(op, out, x));
    JH_REQUIRE(accumulate >= -2 && accumulate <= 2, "jh_chain_apply: accumulate must be 0, +-1 or +-2 (got %d)", accumulate);
    JH_REQUIRE(out->dtype == op->dtype && x->dtype == op->dtype, "jh_chain_apply: dtype mismatch");
    const int64_t nrange = op->row_off[(size_t)op->nrow], ndom = op->col_off[(size_t)op->ncol];
    const int64_t want_out = ch->type == JH_CHAIN_FORWARD ? nrange : ndom, want_in = ch->type == JH_CHAIN_ADJOINT ? nrange : ndom;
    JH_REQUIRE(out->length == want_out && x->length == want_in, "jh_chain_apply: vectors have %lld / %lld elements, the chain maps %lld -> %lld",
               (long long)out->length, (long long)x->length, (long long)want_in, (long long)want_out);
    JH_REQUIRE(out->data != x->data, "jh_chain_apply: the output must not alias the input");
    if (op->nonlinear && !op->pointed)
        return jh_fail(JH_ERR_STATE, "jh_chain_apply: operator has nonlinear blocks and no linearisation point (jh_blockop_point)");
    const void *rng = ch->type == JH_CHAIN_FORWARD ? out->data : (ch->type == JH_CHAIN_ADJOINT ? x->data : nullptr);
    const void *dom = ch->type == JH_CHAIN_FORWARD ? x->data : out->data;
    if (!jhb::tall_unaligned_ok(op, rng, dom) || (ch->type == JH_CHAIN_NORMAL && !jhb::tall_unaligned_ok(op, nullptr, x->data)))
        return jh_fail(JH_ERR_UNSUPPORTED, "jh_chain_apply: a vector or coefficient array is not aligned like its scalar");
    const int64_t n = op->row_len[0];
#define JH_CHAIN_CALL(S, E, NS)                                                                                                             \
    (ch->type == JH_CHAIN_FORWARD ? launch_chain_fwd<S, E, NS>(ch, out->data, x->data, n * E, accumulate)                                   \
     : ch->type == JH_CHAIN_ADJOINT ? launch_chain_adj<S, E, NS, 0>(ch, out->data, x->data, n * E, accumulate)                              \
                                    : launch_chain_adj<S, E, NS, 1>(ch, out->data, x->data, n * E, accumulate))
    switch (op->dtype) {
    case JH_F32: return JH_CHAIN_CALL(float, 1, 4);
    case JH_F64: return JH_CHAIN_CALL(double, 1, 2);
    case JH_C32: return JH_CHAIN_CALL(float, 2, 4);
    case JH_C64: return JH_CHAIN_CALL(double, 2, 2);
    }
#undef JH_CHAIN_CALL
    return jh_fail(JH_ERR_INVALID, "jh_chain_apply: unknown dtype %d", op->dtype);
}

}  // extern "C"
