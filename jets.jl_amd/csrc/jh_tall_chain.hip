// jh_tall_chain.hip -- the chain handle and the C ABI of the fused chains (jh_chain_*); the kernels and their launchers live in jh_tall_chain_kernels.h and are
// instantiated in three translation units (this one: FORWARD; jh_tall_chain_adj.hip: ADJOINT; jh_tall_chain_nrm.hip: NORMAL) -- one unit took 104 s to
// compile, the critical path of every build.
#include "jh_tall_chain_kernels.h"

namespace jhb {
int chain_launch_adjoint(const jh_chain *ch, void *out, const void *in, int accumulate);   // jh_tall_chain_adj.hip
int chain_launch_normal(const jh_chain *ch, void *out, const void *in, int accumulate);    // jh_tall_chain_nrm.hip
}  // namespace jhb

namespace {

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
            JH_REQUIRE((g.flags & ~(JH_STAGE_CONJ | JH_STAGE_ROWSUM)) == 0, "jh_chain_create: unknown flags %d on a DIAG stage", g.flags);
            JH_REQUIRE(!(g.flags & JH_STAGE_ROWSUM) || g.row_flags, "jh_chain_create: JH_STAGE_ROWSUM is for the children of a block-diagonal block operator (range side, row_flags given)");
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
            p.st[s] = (uint32_t)((g.flags & JH_STAGE_CONJ) ? CK_DIAG_CONJ : CK_DIAG) | ((uint32_t)found << 4) | ((g.flags & JH_STAGE_ROWSUM) ? CK_ROWSUM : 0u);
        } else {
            return jh_fail(JH_ERR_INVALID, "jh_chain_create: unknown stage kind %d", g.kind);
        }
    }
    return JH_OK;
}

// word 0 of every record from the operator's blocks as they are NOW, then the table to the device (at create, and again when jh_blockop_point has
// moved the SQUARE rows' arrays since)
int chain_sync_rows(jh_chain *ch)
{
    const jh_blockop *op = ch->op;
    const size_t rw = (size_t)(1 + ch->nw);
    for (int64_t i = 0; i < op->nrow; i++) {
        const jh_block_desc &b = op->blocks[(size_t)i];
        const jh_dev_block db = jh_dev_block_of(b);
        const uint64_t p = (uint64_t)(uintptr_t)b.coeff;
        JH_REQUIRE((p >> 48) == 0, "fused chain: a coefficient address does not fit 48 bits");
        ch->host_tab[(size_t)i * rw] = p | ((uint64_t)(b.kind & 7) << 48) | ((uint64_t)(b.adjoint ? 1 : 0) << 51) | ((uint64_t)(db.real_scale ? 1 : 0) << 52);
    }
    JH_CHECK_HIP(hipMemcpyAsync(ch->dev_tab, ch->host_tab.data(), ch->host_tab.size() * sizeof(uint64_t), hipMemcpyHostToDevice, jh_ctx().stream));
    JH_CHECK_HIP(hipStreamSynchronize(jh_ctx().stream));
    ch->op_gen = op->table_gen;
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
    const size_t rw = (size_t)(1 + ch->nw);
    ch->host_tab.assign(rw * (size_t)op->nrow, 0);
    for (int w = 0; w < ch->nw; w++)
        for (int64_t i = 0; i < op->nrow; i++) {
            const void *p = s_mid[(size_t)w]->coeff[i];
            const uint64_t fl = s_mid[(size_t)w]->row_flags ? (s_mid[(size_t)w]->row_flags[i] & 3u) : 0u;
            note(p);
            if (((uint64_t)(uintptr_t)p) >> 48) scalar_aligned = false;           // (a device address above 2^48 does not exist on this platform)
            ch->host_tab[(size_t)i * rw + 1 + (size_t)w] = (uint64_t)(uintptr_t)p | (fl << 48);
        }
    if (!scalar_aligned) { delete ch; return jh_fail(JH_ERR_UNSUPPORTED, "jh_chain_create: a coefficient array is not aligned like its scalar"); }
    hipError_t e = jh_device_malloc(jh_ctx().device, (void **)&ch->dev_tab, ch->host_tab.size() * sizeof(uint64_t));
    if (e != hipSuccess) {
        delete ch;
        return jh_fail(e == hipErrorOutOfMemory ? JH_ERR_NOMEM : JH_ERR_HIP, "jh_chain_create: %s", hipGetErrorString(e));
    }
    ch->args.rows = ch->dev_tab;
    {
        const int st2 = chain_sync_rows(ch);
        if (st2 != JH_OK) { (void)hipFree(ch->dev_tab); delete ch; return st2; }
    }
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
    if (ch->op_gen != op->table_gen) JH_TRY(chain_sync_rows(const_cast<jh_chain *>(ch)));   // (the operator was pointed again: its SQUARE rows' arrays moved)
    const int64_t n = op->row_len[0];
    if (ch->type == JH_CHAIN_ADJOINT) return jhb::chain_launch_adjoint(ch, out->data, x->data, accumulate);
    if (ch->type == JH_CHAIN_NORMAL) return jhb::chain_launch_normal(ch, out->data, x->data, accumulate);
#define JH_CHAIN_CALL(S, E, NS) launch_chain_fwd<S, E, NS>(ch, out->data, x->data, n * E, accumulate)
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

// ---- the chain kernels as the library's own adjoint / fused A'A of operators with rows of several kinds (round 6) -----------------------------------
// With EMPTY stage lists the ADJOINT chain is m = sum_i conj(a_i) .* d_i -- jh_blockop_mul_adj --, the NORMAL chain jh_blockop_normal_mul, over packed 8-byte row records requested a batch ahead,
// where k_tall_diag_adj<MIXED> reads a 48-byte block descriptor per row behind a kind switch.  On rows of up to ~2 MiB (one workgroup per CU or the split
// walk) that is the difference: same box, one identity row among the diagonals, TB/s library | chain kernel: 256 x 2 MiB 5.4-5.8 | 6.2-6.85,
// 4096 x 1 MiB 5.35-5.64 | 6.8-7.0, 2048 x 512 KiB 5.3-6.1 | 6.1-6.6, 262144 x 513 elements (off the grid) 4.4 | 5.5; from 4 MiB rows on they are
// level and the fat shapes of k_tall_diag_adj win (tools/exp_chain_vs_mixed.py, profiles/exp_r06_chain_vs_mixed.txt).  Later in the round both kernels learnt to run a
// batch of PLAIN diagonals through the all-diagonal kernel's tight loop (the per-row kind switch was the cost): the adjoints drew level, and the fused A'A of the chain
// kernel pulled ahead -- one identity row: 256 x 2 MiB 5.2 | 6.6-6.7, 1024 x 1 MiB 5.8 | 6.7-7.0, 1024 x 4 MiB 6.2 | 6.95 -- so rows of up to 4 MiB take it for both.  Same bits (rows in order from +0);
// where the rows are summed in parts the part count may differ from k_tall_diag_adj's (tolerance parity either way).
namespace jhb {
int bare_chain(const jh_blockop *op, void *out, const void *in, int mode, bool *took)
{
    *took = false;
    jh_context &c = jh_ctx();
    const int64_t row_bytes = op->row_len[0] * (int64_t)jh_dtype_size(op->dtype);
    if (!c.adj_bare_chain || op->nrow < 2 || row_bytes < 16 || row_bytes > ((int64_t)4 << 20)) return JH_OK;
    if (c.adj_rows_per_launch != 0 || (mode == 0 && (c.adj_from_found || c.adj_in_scale != 1.0))) return JH_OK;
    const char *sb = (const char *)c.scratch_dev;
    if (sb && (const char *)out >= sb && (const char *)out < sb + c.scratch_cap) return JH_OK;   // an output reserved behind split_adjoint_tmp's slabs: that route's part count
    jh_chain *&slot = op->bare_chain[mode ? 1 : 0];
    if (!slot || slot->op_gen != op->table_gen) {
        if (stream_is_capturing(c.stream)) return JH_OK;                                          // (building or refreshing the row table copies to the device)
        if (!slot) {
            jh_chain *ch = nullptr;
            const int st = jh_chain_create(op, mode ? JH_CHAIN_NORMAL : JH_CHAIN_ADJOINT, 0, nullptr, 0, nullptr, 0, nullptr, &ch);
            if (st == JH_ERR_UNSUPPORTED) return JH_OK;
            JH_TRY(st);
            slot = ch;
        } else {
            JH_TRY(chain_sync_rows(slot));
        }
    }
    *took = true;
    return mode ? chain_launch_normal(slot, out, in, 0) : chain_launch_adjoint(slot, out, in, 0);
}
}  // namespace jhb
