// jh_grid_normal.hip -- the fused normal operator y = A'(A m) of an N x K GRID of equal diagonal blocks, K = 2 .. 4 (round 6).
//
// The reference applies (A', A) stage by stage (src/Jets.jl:530-534): JetBlock_df! (1010-1032) writes the N range blocks
//     t_i = ((0 + a_i1 .* m_1) + a_i2 .* m_2) + ...            (the composite's zeros(range(A)), 531, then d_i .+= A_ik m_k in column order, 1024)
// and JetBlock_df'! (1034-1057) sums them back per block column,
//     y_k = ((0 + conj(a_1k) .* t_1) + conj(a_2k) .* t_2) + ...   (m_k .= 0, 1042, then rows in order, 1049)
// -- 2 N K n s bytes of coefficients and 2 N n s of the range-sized temporary.  A multi-parameter operator (N shots x K model
// parameters) has a SMALL K: a lane that owns one pack position of the blocks keeps m_1 .. m_K and y_1 .. y_K in registers,
// walks the rows in order and reads every coefficient ONCE: N K n s + 2 K n s bytes, no temporary.  Every product and every sum
// is rounded where the two-stage chain rounds it (-ffp-contract=off), so the result has the chain's bits.
// Many rows of small blocks take the split-row walk of the tall kernels (jh_tall.hip: pick_adj_parts; tolerance parity, adj_split = 0
// keeps the ordered walk).  Blocks need not be whole 16-byte packs (under-aligned packs, jh_blockop_common.h).
//
// Later in round 6: grids whose blocks are of SEVERAL elementwise kinds -- the regularised multi-parameter operator [[A11 A12]; [lam I, 0]; [0, lam I]]: zero
// blocks (skipped: 1022 / 1047), identities, scalars, adjointed diagonals -- through a packed table of one 64-bit word per block (pointer | kind << 48 |
// adjoint << 51 | real scalar << 52, built on first use) and the lesson of the tall kernels: a batch of rows whose blocks are all PLAIN diagonals takes the tight
// loop, any other batch the per-block switch.
#include "jh_blockop_common.h"
#include <vector>

namespace {

constexpr uint64_t GW_PTR = (((uint64_t)1) << 48) - 1;
__device__ inline int gw_kind(uint64_t w) { return (int)((w >> 48) & 7u); }
__device__ inline bool gw_adj(uint64_t w) { return ((w >> 51) & 1u) != 0; }
__device__ inline bool gw_real(uint64_t w) { return ((w >> 52) & 1u) != 0; }

// child mul! of block `idx` on a pack (jh_blockop_common.h: apply_block_loaded, on the block's table word; scalars from the block table)
template <typename S, int E, int NS, typename V>
__device__ inline V grid_apply(uint64_t w, const jh_dev_block *blocks, int64_t idx, V x, V c, bool transposed)
{
    const bool cj = gw_adj(w) != transposed;
    switch (gw_kind(w)) {
    case JH_OP_DIAG: return vmul<S, E, NS, V>(c, x, cj);
    case JH_OP_IDENTITY: return x;
    case JH_OP_SCALE: {
        const double sre = blocks[idx].sre;
        if (E == 1 || gw_real(w)) return (V)(S)sre * x;
        const double sim = blocks[idx].sim;
        V a;
#pragma unroll
        for (int q = 0; q < NS; q += 2) { a[q] = (S)sre; a[q + 1] = (S)sim; }
        return vmul<S, E, NS, V>(a, x, cj);
    }
    default: return (V)(S)0;
    }
}

// the same walk over the PACKED table (MIXED grids): words[(i) * K + k]
template <typename S, int E, int NS, int K, int DEPTH, bool NT>
__global__ __launch_bounds__(256) void k_grid_normal_mixed(const jh_dev_block *__restrict__ blocks, const uint64_t *__restrict__ words, int64_t nrow, int64_t n_scalars,
                                                           const S *__restrict__ m, S *__restrict__ y, int64_t rows_per_part, S *__restrict__ part_out)
{
    typedef typename vec_of<S, NS>::type V;
    const int64_t s0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * NS;
    const bool ok = s0 < n_scalars;
    const int64_t sk = pack_start<NS>(ok ? s0 : 0, n_scalars);
    V x[K], acc[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        x[k] = ldu<false, S, NS>(m + (int64_t)k * n_scalars + sk);
        acc[k] = (V)(S)0;                                                                 // m_k .= 0 (1042)
    }
    int64_t i = 0, iend = nrow;
    if (part_out) {
        i = (int64_t)blockIdx.y * rows_per_part;
        iend = iend < i + rows_per_part ? iend : i + rows_per_part;
    }
    uint64_t nxt[DEPTH][K];
#pragma unroll
    for (int j = 0; j < DEPTH; j++)
#pragma unroll
        for (int k = 0; k < K; k++) nxt[j][k] = words[(i + j < iend ? i + j : i) * K + k];
    for (; i + DEPTH <= iend; i += DEPTH) {
        uint64_t w[DEPTH][K];
        bool plain = true;
#pragma unroll
        for (int j = 0; j < DEPTH; j++)
#pragma unroll
            for (int k = 0; k < K; k++) {
                w[j][k] = nxt[j][k];
                const int64_t r = i + DEPTH + j;
                nxt[j][k] = words[(r < iend ? r : i) * K + k];
                plain = plain && ((w[j][k] >> 48) & 0xFu) == (uint64_t)JH_OP_DIAG;       // kind DIAG, not adjointed
            }
        V c[DEPTH][K];
        if (plain) {
#pragma unroll
            for (int j = 0; j < DEPTH; j++)
#pragma unroll
                for (int k = 0; k < K; k++) c[j][k] = ldu<NT, S, NS>(reinterpret_cast<const S *>(w[j][k] & GW_PTR) + sk);
#pragma unroll
            for (int j = 0; j < DEPTH; j++) {
                V t = (V)(S)0;
#pragma unroll
                for (int k = 0; k < K; k++) t = t + vmul<S, E, NS, V>(c[j][k], x[k], false);
#pragma unroll
                for (int k = 0; k < K; k++) acc[k] = acc[k] + vmul<S, E, NS, V>(c[j][k], t, true);
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < DEPTH; j++)
#pragma unroll
            for (int k = 0; k < K; k++)
                c[j][k] = gw_kind(w[j][k]) == JH_OP_DIAG ? ldu<NT, S, NS>(reinterpret_cast<const S *>(w[j][k] & GW_PTR) + sk) : (V)(S)0;
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            V t = (V)(S)0;                                                                // zeros(range(A)) (531); a zero block is skipped (1022 / 1047)
#pragma unroll
            for (int k = 0; k < K; k++)
                if (gw_kind(w[j][k]) != JH_OP_ZERO) t = t + grid_apply<S, E, NS, V>(w[j][k], blocks, (i + j) + (int64_t)k * nrow, x[k], c[j][k], false);
#pragma unroll
            for (int k = 0; k < K; k++)
                if (gw_kind(w[j][k]) != JH_OP_ZERO) acc[k] = acc[k] + grid_apply<S, E, NS, V>(w[j][k], blocks, (i + j) + (int64_t)k * nrow, t, c[j][k], true);
        }
    }
    for (; i < iend; i++) {
        uint64_t w[K];
        V c[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            w[k] = words[i * K + k];
            c[k] = gw_kind(w[k]) == JH_OP_DIAG ? ldu<NT, S, NS>(reinterpret_cast<const S *>(w[k] & GW_PTR) + sk) : (V)(S)0;
        }
        V t = (V)(S)0;
#pragma unroll
        for (int k = 0; k < K; k++)
            if (gw_kind(w[k]) != JH_OP_ZERO) t = t + grid_apply<S, E, NS, V>(w[k], blocks, i + (int64_t)k * nrow, x[k], c[k], false);
#pragma unroll
        for (int k = 0; k < K; k++)
            if (gw_kind(w[k]) != JH_OP_ZERO) acc[k] = acc[k] + grid_apply<S, E, NS, V>(w[k], blocks, i + (int64_t)k * nrow, t, c[k], true);
    }
    if (!ok) return;
    S *o = part_out ? part_out + (int64_t)blockIdx.y * (K * n_scalars) : y;
#pragma unroll
    for (int k = 0; k < K; k++) st_pack<false, S, NS>(o + (int64_t)k * n_scalars, s0, sk, acc[k]);
}

template <typename S, int E, int NS, int K, int DEPTH, bool NT>
__global__ __launch_bounds__(256) void k_grid_normal(const jh_dev_block *__restrict__ blocks, int64_t nrow, int64_t n_scalars, const S *__restrict__ m,
                                                     S *__restrict__ y, int64_t rows_per_part, S *__restrict__ part_out)
{
    typedef typename vec_of<S, NS>::type V;
    const int64_t s0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * NS;
    const bool ok = s0 < n_scalars;
    const int64_t sk = pack_start<NS>(ok ? s0 : 0, n_scalars);
    V x[K], acc[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        x[k] = ldu<false, S, NS>(m + (int64_t)k * n_scalars + sk);
        acc[k] = (V)(S)0;                                                                 // m_k .= 0 (1042)
    }
    int64_t i = 0, iend = nrow;
    if (part_out) {
        i = (int64_t)blockIdx.y * rows_per_part;
        iend = iend < i + rows_per_part ? iend : i + rows_per_part;
    }
    // block (i, k) = blocks[i + k * nrow] (column-major table); the next batch's pointers are requested while this batch's packs are in flight
    const S *nxt[DEPTH][K];
#pragma unroll
    for (int j = 0; j < DEPTH; j++)
#pragma unroll
        for (int k = 0; k < K; k++) nxt[j][k] = (const S *)blocks[(i + j < iend ? i + j : i) + (int64_t)k * nrow].coeff;
    for (; i + DEPTH <= iend; i += DEPTH) {
        const S *a[DEPTH][K];
#pragma unroll
        for (int j = 0; j < DEPTH; j++)
#pragma unroll
            for (int k = 0; k < K; k++) {
                a[j][k] = nxt[j][k];
                const int64_t r = i + DEPTH + j;
                nxt[j][k] = (const S *)blocks[(r < iend ? r : i) + (int64_t)k * nrow].coeff;
            }
        V c[DEPTH][K];
#pragma unroll
        for (int j = 0; j < DEPTH; j++)
#pragma unroll
            for (int k = 0; k < K; k++) c[j][k] = ldu<NT, S, NS>(a[j][k] + sk);
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            V t = (V)(S)0;                                                                // zeros(range(A)) (531)
#pragma unroll
            for (int k = 0; k < K; k++) t = t + vmul<S, E, NS, V>(c[j][k], x[k], false);  // d_i .+= A_ik m_k (1024)
#pragma unroll
            for (int k = 0; k < K; k++) acc[k] = acc[k] + vmul<S, E, NS, V>(c[j][k], t, true);   // m_k .+= A_ik' d_i (1049)
        }
    }
    for (; i < iend; i++) {
        V c[K];
#pragma unroll
        for (int k = 0; k < K; k++) c[k] = ldu<NT, S, NS>((const S *)blocks[i + (int64_t)k * nrow].coeff + sk);
        V t = (V)(S)0;
#pragma unroll
        for (int k = 0; k < K; k++) t = t + vmul<S, E, NS, V>(c[k], x[k], false);
#pragma unroll
        for (int k = 0; k < K; k++) acc[k] = acc[k] + vmul<S, E, NS, V>(c[k], t, true);
    }
    if (!ok) return;
    S *o = part_out ? part_out + (int64_t)blockIdx.y * (K * n_scalars) : y;
#pragma unroll
    for (int k = 0; k < K; k++) st_pack<false, S, NS>(o + (int64_t)k * n_scalars, s0, sk, acc[k]);
}

template <typename S, int E, int NS, int K, int DEPTH>
int launch_grid_normal(const jh_blockop *op, void *y, const void *m)
{
    jh_context &c = jh_ctx();
    const int64_t n_scalars = op->row_len[0] * E, packs = (n_scalars + NS - 1) / NS;
    const int64_t gx = (packs + 255) / 256;
    int64_t parts = jhb::pick_adj_parts(gx, op->nrow), rows_per_part = 0;
    // (one workgroup per CU, up to two: the ordered walk is latency-bound there, as for the chains -- jh_tall_chain.hip)
    if (parts == 1 && c.adj_split < 0 && op->nrow >= 256 && gx < 2 * (int64_t)c.cu_count) parts = 2;
    void *slabs = nullptr;
    if (parts > 1) {
        rows_per_part = (op->nrow + parts - 1) / parts;
        parts = (op->nrow + rows_per_part - 1) / rows_per_part;
        JH_TRY(jhb::split_slabs(y, (size_t)parts * (size_t)K * (size_t)n_scalars * sizeof(S), &slabs));
    }
    c.last_adj_parts = parts;
    c.last_adj_launches = 1;
    const bool nt = jh_stream_nt((double)op->nrow * (double)K * (double)n_scalars * sizeof(S));
    if (!op->all_diag) {                                       // (the packed table for grids of plain diagonals too: within the noise of this kernel, profiles/bench_grid_normal_r06.txt)
        if (nt)
            hipLaunchKernelGGL((k_grid_normal_mixed<S, E, NS, K, DEPTH, true>), dim3((unsigned)gx, (unsigned)parts), dim3(256), 0, c.stream, op->dev_blocks,
                               (const uint64_t *)op->grid_words, op->nrow, n_scalars, (const S *)m, (S *)y, rows_per_part, (S *)slabs);
        else
            hipLaunchKernelGGL((k_grid_normal_mixed<S, E, NS, K, DEPTH, false>), dim3((unsigned)gx, (unsigned)parts), dim3(256), 0, c.stream, op->dev_blocks,
                               (const uint64_t *)op->grid_words, op->nrow, n_scalars, (const S *)m, (S *)y, rows_per_part, (S *)slabs);
    } else if (nt)
        hipLaunchKernelGGL((k_grid_normal<S, E, NS, K, DEPTH, true>), dim3((unsigned)gx, (unsigned)parts), dim3(256), 0, c.stream, op->dev_blocks, op->nrow,
                           n_scalars, (const S *)m, (S *)y, rows_per_part, (S *)slabs);
    else
        hipLaunchKernelGGL((k_grid_normal<S, E, NS, K, DEPTH, false>), dim3((unsigned)gx, (unsigned)parts), dim3(256), 0, c.stream, op->dev_blocks, op->nrow,
                           n_scalars, (const S *)m, (S *)y, rows_per_part, (S *)slabs);
    JH_CHECK_HIP(hipGetLastError());
    if (parts > 1) return jhb::fold_parts(sizeof(S) == 4 ? JH_F32 : JH_F64, slabs, (int64_t)K * n_scalars, parts, y, 0, (int64_t)K * n_scalars);
    return JH_OK;
}

template <typename S, int E, int NS>
int grid_normal_k(const jh_blockop *op, void *y, const void *m)
{
    // rows in flight: K x DEPTH = 8 (6 for K = 3) coefficient packs per lane; 12 / 9 / 12 measured 2-3 % slower for K = 2 / 4 and within the noise for
    // K = 3 (profiles/bench_grid_normal_r06.txt)
    switch (op->ncol) {
    case 2: return launch_grid_normal<S, E, NS, 2, 4>(op, y, m);
    case 3: return launch_grid_normal<S, E, NS, 3, 2>(op, y, m);
    default: return launch_grid_normal<S, E, NS, 4, 2>(op, y, m);
    }
}

}  // namespace

namespace jhb {

// an N x K grid (N >= 2, K = 2 .. 4) of equal blocks -- plain diagonals, or (knob grid_normal = 1: later in round 6) zero / identity / scalar / diagonal blocks, no
// nonlinear child --, vectors and coefficients aligned like their scalar
bool grid_normal_ok(const jh_blockop *op, const void *y, const void *m)
{
    const int64_t knob = jh_ctx().grid_normal;
    if (!(op->nrow >= 2 && op->ncol >= 2 && op->ncol <= 4 && op->uniform_rows) || knob == 0) return false;
    if (!op->all_diag) {
        if (knob == 2 || !op->elementwise || op->nonlinear || op->wide_scale) return false;
        for (const jh_block_desc &b : op->blocks)
            if (b.kind != JH_OP_ZERO && b.kind != JH_OP_IDENTITY && b.kind != JH_OP_SCALE && b.kind != JH_OP_DIAG) return false;
    }
    const size_t es = jh_dtype_size(op->dtype), sa = jh_dtype_complex(op->dtype) ? es / 2 : es;
    const int64_t n = op->row_len[0];
    if (n * (int64_t)es < 16) return false;
    for (int64_t v : op->col_len)
        if (v != n) return false;
    if ((((uintptr_t)y) | ((uintptr_t)m)) & (sa - 1)) return false;
    return op->coeff_scalar_aligned;
}

int grid_normal(const jh_blockop *op, void *y, const void *m)
{
    if (!op->all_diag && !op->grid_words) {                   // the packed table of a mixed grid, row-major (N x K words), built on first use
        if (stream_is_capturing(jh_ctx().stream))             // (an allocation and a synchronous copy: not inside a stream capture -- the caller chains the two stages)
            return jh_fail(JH_ERR_UNSUPPORTED, "grid normal: the block table of this operator is built on the first call, which must not be inside a stream capture");
        const size_t nw = (size_t)op->nrow * (size_t)op->ncol;
        std::vector<uint64_t> h(nw);
        for (int64_t i = 0; i < op->nrow; i++)
            for (int64_t k = 0; k < op->ncol; k++) {
                const jh_block_desc &b = op->blocks[(size_t)(i + k * op->nrow)];
                const jh_dev_block db = jh_dev_block_of(b);
                const uint64_t p = (uint64_t)(uintptr_t)(b.kind == JH_OP_DIAG ? b.coeff : nullptr);
                JH_REQUIRE((p >> 48) == 0, "grid normal: a coefficient address does not fit 48 bits");
                h[(size_t)(i * op->ncol + k)] = p | ((uint64_t)(b.kind & 7) << 48) | ((uint64_t)(b.adjoint ? 1 : 0) << 51) | ((uint64_t)(db.real_scale ? 1 : 0) << 52);
            }
        void *dev = nullptr;
        JH_CHECK_HIP(hipMalloc(&dev, nw * sizeof(uint64_t)));
        const hipError_t e = hipMemcpy(dev, h.data(), nw * sizeof(uint64_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(dev); JH_CHECK_HIP(e); }
        op->grid_words = dev;
    }
    switch (op->dtype) {
    case JH_F32: return grid_normal_k<float, 1, 4>(op, y, m);
    case JH_F64: return grid_normal_k<double, 1, 2>(op, y, m);
    case JH_C32: return grid_normal_k<float, 2, 4>(op, y, m);
    case JH_C64: return grid_normal_k<double, 2, 2>(op, y, m);
    default: return jh_fail(JH_ERR_INVALID, "grid_normal: unknown dtype %d", op->dtype);
    }
}

}  // namespace jhb
