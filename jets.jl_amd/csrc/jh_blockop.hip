// jh_blockop.hip -- the block operator behind the C ABI: jh_blockop_create / _destroy / _point and the dispatch of JetBlock_df! /
// JetBlock_df'! / JetBlock_f! (src/Jets.jl:988-1057) and of the fused A'oA (530-534 over (A', A)) to the kernel families:
//   jh_tall.hip       tall fast path (ncol == 1, equal elementwise rows; blocks off the 16-byte grid on the MIXED instantiations): the BASELINE.json workload
//   jh_tall_step.hip  fused solver updates and the one-pass LSQR step          jh_tall_sum.hip   fused JetSum
//   jh_general.hip    any nrow x ncol mix of kinds, grids, per-block loops     jh_dense.hip      dense children
// (round 5: one 4 200-line translation unit until then; jh_blockop_common.h)
#include "jh_blockop_common.h"

namespace {

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

// what the per-call route tests need to know about the coefficient arrays of ALL blocks, once per create / point! (a 512 x 512 block-diagonal operator
// spent 0.2 ms per call walking its descriptors): on the 16-byte grid? aligned like their scalar at least (a device array of the element type always is;
// a caller's raw pointer -- or a point vector wrapped at an odd address -- need not be)?
static void scan_coeff_alignment(jh_blockop *op)
{
    const size_t sa = jh_dtype_complex(op->dtype) ? jh_dtype_size(op->dtype) / 2 : jh_dtype_size(op->dtype);
    op->coeff_aligned16 = true;
    op->coeff_scalar_aligned = true;
    for (const auto &b : op->blocks)
        if (b.kind == JH_OP_DIAG || b.kind == JH_OP_SQUARE) {
            if (((uintptr_t)b.coeff) & 15u) op->coeff_aligned16 = false;
            if (((uintptr_t)b.coeff) & (sa - 1)) op->coeff_scalar_aligned = false;
        }
}

}  // namespace

namespace jhb {
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


}  // namespace jhb

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
    if (nrow >= 2 && ncol >= 2 && !op->elementwise && jh_ctx().dense_grid) {   // ... or a grid of them: one tall batch per block column
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
        // late round 5: an eligible operator is ALSO prepared for the batched route -- MANY small children (a block-diagonal operator of 256 matrices of
        // 256 x 256) move far more bytes than one launch of the loop can stream (a thread forms a whole dot product: 0.2-0.5 TB/s); use_small_loop decides per call
        op->dense_mixed = eligible;
        for (const auto &b : op->blocks)
            if (b.kind == JH_OP_DENSE) op->dense_bytes += (double)b.nr * (double)b.nc * (double)es;
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

    // where the products of a dense_mixed operator's dense children go: one compact scratch vector per direction, a 16-byte aligned piece per child
    if (op->dense_mixed) {
        const int64_t per16 = (int64_t)(16 / jh_dtype_size(dtype)) > 0 ? (int64_t)(16 / jh_dtype_size(dtype)) : 1;
        for (int dir = 0; dir < 2; dir++) {
            op->prod_off[dir].assign((size_t)(nrow * ncol), -1);
            int64_t at = 0;
            for (int64_t j = 0; j < ncol; j++)
                for (int64_t i = 0; i < nrow; i++) {
                    if (op->blocks[(size_t)(i + j * nrow)].kind != JH_OP_DENSE) continue;
                    const int64_t out_len = dir ? op->col_len[(size_t)j] : op->row_len[(size_t)i];
                    op->prod_off[dir][(size_t)(i + j * nrow)] = at;
                    at += (out_len + per16 - 1) / per16 * per16;
                }
            op->prod_total[dir] = at;
        }
    }
    std::vector<jh_dev_block> host((size_t)(nrow * ncol));
    for (size_t k = 0; k < host.size(); k++) {
        host[k] = jh_dev_block_of(op->blocks[k]);
        if (op->dense_mixed && op->blocks[k].kind == JH_OP_DENSE) jh_dev_block_set_prod_off(host[k], op->prod_off[0][k], op->prod_off[1][k]);
    }
    // what the per-call route tests need to know about ALL blocks, once (a 512 x 512 block-diagonal operator spent 0.2 ms per call walking its descriptors)
    scan_coeff_alignment(op);
    op->lens_hold_a_pack = true;                                           // every non-empty row / column at least 16 bytes long
    for (int64_t v : op->row_len) if (v > 0 && (size_t)v * jh_dtype_size(dtype) < 16) op->lens_hold_a_pack = false;
    for (int64_t v : op->col_len) if (v > 0 && (size_t)v * jh_dtype_size(dtype) < 16) op->lens_hold_a_pack = false;
    op->lens_aligned16 = true;
    for (int64_t v : op->row_len) if (((size_t)v * jh_dtype_size(dtype)) % 16) op->lens_aligned16 = false;
    for (int64_t v : op->col_len) if (((size_t)v * jh_dtype_size(dtype)) % 16) op->lens_aligned16 = false;
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
    // step lists for sparse grids (jh_general.hip: k_general_tile LIST): grids of EQUAL elementwise blocks; four-line groups (directions with >= 4 lines)
    // and single lines
    // (the per-line lists also serve the one-line general kernels -- the combine launch of dense_mixed_apply --, for grids of any blocks)
    std::vector<int> steps[2][2];
    if (nrow >= 2 && ncol >= 2 && nrow < ((int64_t)1 << 30) && ncol < ((int64_t)1 << 30)) {
        bool equal = op->elementwise && op->uniform_rows;                  // the four-line lists are k_general_tile's: grids of EQUAL elementwise blocks
        for (int64_t v : op->col_len) if (v != op->row_len[0]) equal = false;
        for (int dir = 0; dir < 2; dir++)
            for (int set = equal ? 0 : 1; set < 2; set++) {
                const int64_t nlines = dir ? ncol : nrow, nsum = dir ? nrow : ncol, R = set ? 1 : 4;
                if (nlines < R) continue;
                const int64_t ngroups = (nlines + R - 1) / R;
                std::vector<std::vector<int>> per((size_t)ngroups);
                size_t longest = 0;
                int64_t total = 0;
                for (int64_t g = 0; g < ngroups; g++) {
                    for (int64_t q = 0; q < nsum; q++) {
                        bool any = false;
                        for (int64_t l = R * g; l < R * g + R && l < nlines && !any; l++)
                            any = op->blocks[(size_t)(dir ? q + l * nrow : l + q * nrow)].kind != JH_OP_ZERO;
                        if (any) per[(size_t)g].push_back((int)q);
                    }
                    longest = per[(size_t)g].size() > longest ? per[(size_t)g].size() : longest;
                    total += (int64_t)per[(size_t)g].size();
                }
                const size_t stride = (longest + 1 + JH_STEP_PAD + 3) / 4 * 4;   // count + indices + look-ahead entries, records on 16-byte boundaries
                if ((double)stride * (double)ngroups > 1.0e9) continue;        // (a table of this size is not a grid of big blocks)
                std::vector<int> &L = steps[dir][set];
                L.assign(stride * (size_t)ngroups, 0);
                for (int64_t g = 0; g < ngroups; g++) {
                    int *rec = L.data() + (size_t)g * stride;
                    rec[0] = (int)per[(size_t)g].size();
                    for (size_t k = 0; k < per[(size_t)g].size(); k++) rec[1 + k] = per[(size_t)g][k];
                }
                op->list_steps[dir][set] = total;
                op->step_stride[dir][set] = (int64_t)stride;
                if (e == hipSuccess) e = jh_device_malloc(jh_ctx().device, (void **)&op->dev_steps[dir][set], L.size() * sizeof(int));
                if (e == hipSuccess) e = hipMemcpyAsync(op->dev_steps[dir][set], L.data(), L.size() * sizeof(int), hipMemcpyHostToDevice, st);
            }
    }
    // the non-zero rows of a tall operator whose rows are not all diagonals (jh_tall.hip: launch_tall_fwd_mixed)
    std::vector<int> rows_nz;
    if (op->tall && op->elementwise && !op->all_diag && nrow >= 2 && nrow < ((int64_t)1 << 31)) {
        for (int64_t i = 0; i < nrow; i++)
            if (op->blocks[(size_t)i].kind != JH_OP_ZERO) rows_nz.push_back((int)i);
        op->n_rows_nz = (int64_t)rows_nz.size();
        if (!rows_nz.empty() && op->n_rows_nz < nrow) {
            if (e == hipSuccess) e = jh_device_malloc(jh_ctx().device, (void **)&op->dev_rows_nz, rows_nz.size() * sizeof(int));
            if (e == hipSuccess) e = hipMemcpyAsync(op->dev_rows_nz, rows_nz.data(), rows_nz.size() * sizeof(int), hipMemcpyHostToDevice, st);
        }
    }
    // the dense children of a dense_mixed operator as lists per direction and pass (jh_dense.hip: k_gemv_rows_list / k_gemv_cols_list)
    std::vector<jh_dense_item> items[2][2];
    std::vector<int> comb_ptr[2];
    std::vector<int64_t> comb_off[2];
    if (op->dense_mixed) {
        for (int dir = 0; dir < 2; dir++) {
            for (int64_t j = 0; j < ncol; j++)
                for (int64_t i = 0; i < nrow; i++) {
                    const jh_block_desc &b = op->blocks[(size_t)(i + j * nrow)];
                    if (b.kind != JH_OP_DENSE) continue;
                    const int64_t out_len = dir ? op->col_len[(size_t)j] : op->row_len[(size_t)i], in_len = dir ? op->row_len[(size_t)i] : op->col_len[(size_t)j];
                    if (out_len == 0) continue;                               // (an empty product; a child with an empty INPUT stays: its slab piece must hold zeros)
                    const int pass = ((b.adjoint != 0) == (dir != 0)) ? 0 : 1;
                    jh_dense_item it;
                    it.A = b.coeff;
                    it.nr = pass == 0 ? out_len : in_len;
                    it.nc = pass == 0 ? in_len : out_len;
                    it.x_off = dir ? op->row_off[(size_t)i] : op->col_off[(size_t)j];
                    it.out_off = op->prod_off[dir][(size_t)(i + j * nrow)];
                    it.line_off = dir ? op->col_off[(size_t)j] : op->row_off[(size_t)i];
                    items[dir][pass].push_back(it);
                    if (out_len > op->items_max_out[dir][pass]) op->items_max_out[dir][pass] = out_len;
                    if (in_len > op->items_max_in[dir][pass]) op->items_max_in[dir][pass] = in_len;
                }
            // direct mode: every line of this direction with exactly one non-zero block, a dense child (forward: lines without blocks allowed -- they stay as
            // found, 1022; adjoint of a grid: none -- a column without blocks must be zeroed, 1042, which is the combine launch's work)
            {
                const int64_t nlines = dir ? ncol : nrow, nsum = dir ? nrow : ncol;
                bool direct = true;
                for (int64_t l = 0; l < nlines && direct; l++) {
                    int nz = 0, dense = 0;
                    for (int64_t q = 0; q < nsum; q++) {
                        const jh_block_desc &b = op->blocks[(size_t)(dir ? q + l * nrow : l + q * nrow)];
                        if (b.kind != JH_OP_ZERO) nz++;
                        if (b.kind == JH_OP_DENSE) dense++;
                    }
                    const int64_t len = dir ? op->col_len[(size_t)l] : op->row_len[(size_t)l];
                    if (nz > 1 || nz != dense) direct = false;
                    if (nz == 0 && dir == 1 && nrow > 1 && len > 0) direct = false;
                }
                op->dense_direct[dir] = direct;
            }
            // the combine's lists (round 6): only when every non-zero block is a dense child
            {
                const int64_t nlines = dir ? ncol : nrow, nsum = dir ? nrow : ncol;
                bool only_dense = nlines < ((int64_t)1 << 30);
                for (const auto &b : op->blocks)
                    if (b.kind != JH_OP_DENSE && b.kind != JH_OP_ZERO) only_dense = false;
                if (only_dense) {
                    comb_ptr[dir].assign((size_t)nlines + 1, 0);
                    for (int64_t l = 0; l < nlines; l++) {
                        for (int64_t q = 0; q < nsum; q++) {
                            const size_t k = (size_t)(dir ? q + l * nrow : l + q * nrow);
                            if (op->blocks[k].kind == JH_OP_DENSE) comb_off[dir].push_back(op->prod_off[dir][k]);
                        }
                        comb_ptr[dir][(size_t)l + 1] = (int)comb_off[dir].size();
                    }
                    if (!comb_off[dir].empty()) {
                        if (e == hipSuccess) e = jh_device_malloc(jh_ctx().device, (void **)&op->dev_comb_ptr[dir], comb_ptr[dir].size() * sizeof(int));
                        if (e == hipSuccess) e = jh_device_malloc(jh_ctx().device, (void **)&op->dev_comb_off[dir], comb_off[dir].size() * sizeof(int64_t));
                        if (e == hipSuccess) e = hipMemcpyAsync(op->dev_comb_ptr[dir], comb_ptr[dir].data(), comb_ptr[dir].size() * sizeof(int), hipMemcpyHostToDevice, st);
                        if (e == hipSuccess) e = hipMemcpyAsync(op->dev_comb_off[dir], comb_off[dir].data(), comb_off[dir].size() * sizeof(int64_t), hipMemcpyHostToDevice, st);
                    }
                }
            }
            for (int pass = 0; pass < 2; pass++) {
                op->n_items[dir][pass] = (int64_t)items[dir][pass].size();
                if (items[dir][pass].empty()) continue;
                const size_t bytes = items[dir][pass].size() * sizeof(jh_dense_item);
                if (e == hipSuccess) e = jh_device_malloc(jh_ctx().device, (void **)&op->dev_items[dir][pass], bytes);
                if (e == hipSuccess) e = hipMemcpyAsync(op->dev_items[dir][pass], items[dir][pass].data(), bytes, hipMemcpyHostToDevice, st);
            }
        }
    }
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
    lazy_release(op->gen_tune[0]);
    lazy_release(op->gen_tune[1]);
    if (op->dev_blocks) (void)hipFree(op->dev_blocks);
    if (op->dev_row_off) (void)hipFree(op->dev_row_off);
    if (op->dev_col_off) (void)hipFree(op->dev_col_off);
    if (op->dev_row_touched) (void)hipFree(op->dev_row_touched);
    if (op->dev_dims) (void)hipFree(op->dev_dims);
    if (op->dev_rows_nz) (void)hipFree(op->dev_rows_nz);
    for (int dir = 0; dir < 2; dir++)
        for (int set = 0; set < 2; set++)
            if (op->dev_steps[dir][set]) (void)hipFree(op->dev_steps[dir][set]);
    for (int dir = 0; dir < 2; dir++)
        for (int pass = 0; pass < 2; pass++)
            if (op->dev_items[dir][pass]) (void)hipFree(op->dev_items[dir][pass]);
    for (int dir = 0; dir < 2; dir++) {
        if (op->dev_comb_ptr[dir]) (void)hipFree(op->dev_comb_ptr[dir]);
        if (op->dev_comb_off[dir]) (void)hipFree(op->dev_comb_off[dir]);
    }
    if (op->grid_words) (void)hipFree(op->grid_words);
    for (int k = 0; k < 2; k++)
        if (op->bare_chain[k]) (void)jh_chain_destroy(op->bare_chain[k]);
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
        if (op->dense_mixed && op->blocks[k].kind == JH_OP_DENSE) jh_dev_block_set_prod_off(host[k], op->prod_off[0][k], op->prod_off[1][k]);
    }
    scan_coeff_alignment(op);                                            // (the SQUARE blocks' arrays have moved)
    op->table_gen++;
    hipStream_t st = jh_ctx().stream;
    JH_CHECK_HIP(hipMemcpyAsync(op->dev_blocks, host.data(), host.size() * sizeof(jh_dev_block), hipMemcpyHostToDevice, st));
    JH_CHECK_HIP(hipStreamSynchronize(st));                             // host staging vector dies at return
    return JH_OK;
}

// the one-launch loop for operators of small dense children -- unless the batched route is available too and the children together are
// big enough to be a streaming problem (knob small_loop_max_kib: dense bytes from which the batched route takes over)
static inline bool use_small_loop(const jh_blockop *op)
{
    const jh_context &c = jh_ctx();
    if (!(op->small_loop && c.small_loop)) return false;
    return !(op->dense_mixed && c.dense_mixed && c.dense_list && op->dense_bytes >= (double)c.small_loop_max_kib * 1024.0);
}
// the batched route (dense_mixed_apply); an operator that is ALSO a one-launch-loop operator takes it only in place of that loop (knob small_loop = 0 still
// means the reference's per-block loop for those)
static inline bool use_dense_mixed(const jh_blockop *op)
{
    const jh_context &c = jh_ctx();
    return op->dense_mixed && c.dense_mixed && (!op->small_loop || (c.small_loop && !use_small_loop(op)));
}

int jh_blockop_f(const jh_blockop *op, jh_bvec *d, const jh_bvec *m)
{
    JH_TRY(jh_enter(op, d, m));
    JH_TRY(check_vectors(op, d, m, "jh_blockop_f"));
    // round 5, last session: a TALL nonlinear operator of elementwise children (SQUARE, diagonals, identity, scalar, zero rows) evaluates F(m) on the tall
    // tiling -- the model tile in registers, one pass over the rows (256 x 256^3 SQUARE children: see profiles/bench_unaligned_r05.txt) -- instead of
    // the general one-line kernels; every row is written (1003), the same products
    if (jh_ctx().tall_f != 0 && (tall_mixed_ok(op, d->data, m->data) || tall_unaligned_ok(op, d->data, m->data)))
        return jhb::tall_fwd_mixed(op, d->data, m->data, 1);
    if (op->dense_batch) return jh_launch_gemv_batched(op->dev_blocks, op->nrow, op->blocks[0].nr, op->blocks[0].nc, op->dtype, d->data, m->data, 0, op->dense_aligned, false);
    if (op->dense_batch_wide) return jh_launch_gemv_batched(op->dev_blocks, op->ncol, op->blocks[0].nr, op->blocks[0].nc, op->dtype, d->data, m->data, 0, op->dense_aligned, true);
    if (op->dense_batch_grid) return dense_grid_fwd(op, d->data, m->data);
    if (op->dense_batch_ragged) return jh_launch_gemv_batched(op->dev_blocks, op->nrow, op->dense_max_nr, op->blocks[0].nc, op->dtype, d->data, m->data, 0, op->dense_aligned, false, op->dev_row_off);
    if (use_small_loop(op)) return loop_small(op, d->data, m->data, 0, 1);
    if (use_dense_mixed(op)) return dense_mixed(op, d->data, m->data, false, true);
    if (!op->elementwise) return run_loop_graphed(op, 2, d->data, m->data, [&] { return loop_fwd(op, d->data, m->data, true); });
    return jhb::general_fwd(op, d->data, m->data, 1);
    return jh_fail(JH_ERR_INVALID, "jh_blockop_f: unknown dtype %d", op->dtype);
}

int jh_blockop_mul(const jh_blockop *op, jh_bvec *d, const jh_bvec *m)
{
    JH_TRY(jh_enter(op, d, m));
    JH_TRY(check_vectors(op, d, m, "jh_blockop_mul"));
    if (op->nonlinear && !op->pointed)
        return jh_fail(JH_ERR_STATE, "jh_blockop_mul: operator has nonlinear blocks and no linearisation point (jh_blockop_point)");
    if (tall_fast_ok(op, d->data, m->data)) {
        return jhb::tall_fwd(op, d->data, m->data);
    }
    if (tall_mixed_ok(op, d->data, m->data) || tall_unaligned_ok(op, d->data, m->data)) {   // rows of several elementwise kinds: the tall tiling with a per-row kind
        return jhb::tall_fwd_mixed(op, d->data, m->data);                                   // (and rows that are not whole, aligned packs: the same instantiations)
    }
    if (op->dense_batch) return jh_launch_gemv_batched(op->dev_blocks, op->nrow, op->blocks[0].nr, op->blocks[0].nc, op->dtype, d->data, m->data, 0, op->dense_aligned, false);
    if (op->dense_batch_wide) return jh_launch_gemv_batched(op->dev_blocks, op->ncol, op->blocks[0].nr, op->blocks[0].nc, op->dtype, d->data, m->data, 0, op->dense_aligned, true);
    if (op->dense_batch_grid) return dense_grid_fwd(op, d->data, m->data);
    if (op->dense_batch_ragged) return jh_launch_gemv_batched(op->dev_blocks, op->nrow, op->dense_max_nr, op->blocks[0].nc, op->dtype, d->data, m->data, 0, op->dense_aligned, false, op->dev_row_off);
    if (use_small_loop(op)) return loop_small(op, d->data, m->data, 0, 0);
    if (use_dense_mixed(op)) return dense_mixed(op, d->data, m->data, false);
    if (!op->elementwise) return run_loop_graphed(op, 0, d->data, m->data, [&] { return loop_fwd(op, d->data, m->data); });
    // a wide operator's forward d = d_found + sum_j A_1j m_j (1024: no zeroing) is its tall twin's ordered adjoint sum started from
    // what d holds -- the same additions in the same order.  Large blocks only: the ordered walk needs >= one workgroup per CU
    // (many small blocks take the general kernel's split walk instead)
    if (op->twin && jh_ctx().wide_twin && jh_ctx().adj_split <= 0 &&
        (jh_ctx().wide_twin == 2 || (size_t)op->row_len[0] * jh_dtype_size(op->dtype) >= ((size_t)16 << 20)) &&
        (tall_fast_ok(op->twin, m->data, d->data) || tall_mixed_ok(op->twin, m->data, d->data) || tall_unaligned_ok(op->twin, m->data, d->data))) {
        jh_context &c = jh_ctx();
        c.adj_from_found = 1;
        const int st = jh_blockop_mul_adj(op->twin, d, m);
        c.adj_from_found = 0;
        return st;
    }
    return jhb::general_fwd(op, d->data, m->data);
    return jh_fail(JH_ERR_INVALID, "jh_blockop_mul: unknown dtype %d", op->dtype);
}

int jh_blockop_mul_adj(const jh_blockop *op, jh_bvec *m, const jh_bvec *d)
{
    JH_TRY(jh_enter(op, m, d));
    JH_TRY(check_vectors(op, d, m, "jh_blockop_mul_adj"));
    if (op->nonlinear && !op->pointed)
        return jh_fail(JH_ERR_STATE, "jh_blockop_mul_adj: operator has nonlinear blocks and no linearisation point (jh_blockop_point)");
    if (tall_fast_ok(op, d->data, m->data)) {
        return jhb::tall_adj(op, m->data, d->data, 0, false);
    }
    if (tall_mixed_ok(op, d->data, m->data) || tall_unaligned_ok(op, d->data, m->data)) {
        return jhb::tall_adj(op, m->data, d->data, 0, true);
    }
    if (op->dense_batch) return jh_launch_gemv_batched(op->dev_blocks, op->nrow, op->blocks[0].nr, op->blocks[0].nc, op->dtype, m->data, d->data, 1, op->dense_aligned, false);
    if (op->dense_batch_wide) return jh_launch_gemv_batched(op->dev_blocks, op->ncol, op->blocks[0].nr, op->blocks[0].nc, op->dtype, m->data, d->data, 1, op->dense_aligned, true);
    if (op->dense_batch_grid) return dense_grid_adj(op, m->data, d->data);
    if (op->dense_batch_ragged) return jh_launch_gemv_batched(op->dev_blocks, op->nrow, op->dense_max_nr, op->blocks[0].nc, op->dtype, m->data, d->data, 1, op->dense_aligned, false, op->dev_row_off);
    if (use_small_loop(op)) return loop_small(op, m->data, d->data, 1, 0);
    if (use_dense_mixed(op)) return dense_mixed(op, m->data, d->data, true);
    if (!op->elementwise) return run_loop_graphed(op, 1, m->data, d->data, [&] { return loop_adj(op, m->data, d->data); });
    // a wide operator's adjoint is its tall twin's forward (same bits: one rounded product per element, zero blocks untouched)
    if (op->twin && jh_ctx().wide_twin &&
        (tall_fast_ok(op->twin, m->data, d->data) || tall_mixed_ok(op->twin, m->data, d->data) || tall_unaligned_ok(op->twin, m->data, d->data)))
        return jh_blockop_mul(op->twin, m, d);
    return jhb::general_adj(op, m->data, d->data);
    return jh_fail(JH_ERR_INVALID, "jh_blockop_mul_adj: unknown dtype %d", op->dtype);
}

int jh_blockop_mul_adj_range(const jh_blockop *op, jh_bvec *m, const jh_bvec *d, int64_t first_elem, int64_t count)
{
    JH_TRY(jh_enter(op, m, d));
    JH_TRY(check_vectors(op, d, m, "jh_blockop_mul_adj_range"));
    JH_REQUIRE(first_elem >= 0 && count >= 0 && first_elem + count <= m->length,
               "jh_blockop_mul_adj_range: elements [%lld, %lld) outside the domain vector (%lld elements)", (long long)first_elem,
               (long long)(first_elem + count), (long long)m->length);
    // (rows off the 16-byte pack grid: the MIXED instantiations, like the whole-vector call -- the LAST range may then end inside a pack)
    const bool fast = tall_fast_ok(op, d->data, m->data);
    const bool mixed = !fast && (tall_mixed_ok(op, d->data, m->data) || tall_unaligned_ok(op, d->data, m->data));
    if (mixed && op->nonlinear && !op->pointed)
        return jh_fail(JH_ERR_STATE, "jh_blockop_mul_adj_range: operator has nonlinear blocks and no linearisation point (jh_blockop_point)");
    if (!mixed && !fast)
        return jh_fail(JH_ERR_UNSUPPORTED, "jh_blockop_mul_adj_range: needs a tall operator of >= 2 equal elementwise rows");
    const int64_t es = (int64_t)jh_dtype_size(op->dtype);
    JH_REQUIRE((first_elem * es) % 16 == 0 && ((count * es) % 16 == 0 || first_elem + count == m->length),
               "jh_blockop_mul_adj_range: chunk boundaries must be 16-byte aligned (the last chunk may end with the vector)");
    if (mixed) return jhb::tall_adj(op, m->data, d->data, 0, true, first_elem, first_elem + count);
    return jhb::tall_adj(op, m->data, d->data, 0, false, first_elem, first_elem + count);
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
    if (tall_mixed_ok(op, y->data, m->data) || (!tall_fast_ok(op, y->data, m->data) && tall_unaligned_ok(op, y->data, m->data))) {   // rows of several elementwise kinds (a zero row adds nothing: 1022 + 1047)
        if (op->nonlinear && !op->pointed)
            return jh_fail(JH_ERR_STATE, "jh_blockop_normal_mul: operator has nonlinear blocks and no linearisation point (jh_blockop_point)");
        return jhb::tall_adj(op, y->data, m->data, 1, true);
    }
    if (jhb::grid_normal_ok(op, y->data, m->data)) return jhb::grid_normal(op, y->data, m->data);   // N x (2 .. 4) grids of equal diagonals (round 6)
    if (!tall_fast_ok(op, y->data, m->data) || op->nrow < 2)
        return jh_fail(JH_ERR_UNSUPPORTED,
                       "jh_blockop_normal_mul: fused A'A needs a tall (>= 2 rows) operator of equal elementwise rows or an N x (2 .. 4) grid of equal "
                       "diagonals; chain jh_blockop_mul and jh_blockop_mul_adj instead");
    return jhb::tall_adj(op, y->data, m->data, 1, false);
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
    JH_REQUIRE((first_elem * es) % 16 == 0 && ((count * es) % 16 == 0 || first_elem + count == y->length),
               "jh_blockop_normal_mul_range: chunk boundaries must be 16-byte aligned (the last chunk may end with the vector)");
    const int64_t lo = first_elem, hi = first_elem + count;
    if (tall_mixed_ok(op, y->data, m->data) || (!tall_fast_ok(op, y->data, m->data) && tall_unaligned_ok(op, y->data, m->data))) {   // (rows off the pack grid too)
        if (op->nonlinear && !op->pointed)
            return jh_fail(JH_ERR_STATE, "jh_blockop_normal_mul_range: operator has nonlinear blocks and no linearisation point (jh_blockop_point)");
        return jhb::tall_adj(op, y->data, m->data, 1, true, lo, hi);
    }
    if (!tall_fast_ok(op, y->data, m->data) || op->nrow < 2)
        return jh_fail(JH_ERR_UNSUPPORTED,
                       "jh_blockop_normal_mul_range: fused A'A needs a tall (>= 2 rows) operator of equal elementwise rows");
    return jhb::tall_adj(op, y->data, m->data, 1, false, lo, hi);
    return jh_fail(JH_ERR_INVALID, "jh_blockop_normal_mul_range: unknown dtype %d", op->dtype);
}


}  // extern "C"
