// jh_internal.h -- shared internals of libjetship.so (gfx950 only; no CUDA/compat paths).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <vector>
#include "../../include/jetship.h"

// ---- error plumbing: status code + thread-local message, no exceptions across the ABI --------
int jh_fail(int status, const char *fmt, ...);
#define JH_CHECK_HIP(expr)                                                                      \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return jh_fail(_e == hipErrorOutOfMemory ? JH_ERR_NOMEM : JH_ERR_HIP, "%s: %s (%s:%d)", \
                           #expr, hipGetErrorString(_e), __FILE__, __LINE__);                   \
    } while (0)
#define JH_REQUIRE(cond, ...)                                  \
    do {                                                       \
        if (!(cond)) return jh_fail(JH_ERR_INVALID, __VA_ARGS__); \
    } while (0)
#define JH_TRY(expr)                  \
    do {                              \
        int _s = (expr);              \
        if (_s != JH_OK) return _s;   \
    } while (0)

// A CONTEXT is one device + one HIP stream + the library's workspaces and knobs for it.  jh_init(device) creates the PRIMARY
// context of a device; jh_context_create adds further ones (also on a device that already has one: an independent stream).
// A host thread has a CURRENT context (jh_context_use; the first one created until then); every handle remembers the context it
// was created in and every entry point that takes handles switches to theirs (jh_enter) -- so one process can drive several
// GPUs through this ABI, the way SURVEY section 8e sketches it.
constexpr int JH_MAX_CTX = 64;          // table slots (a power of two)
// A context id is slot + JH_MAX_CTX * generation: a slot is reused after jh_context_destroy / jh_shutdown, its generation is not, so
// a handle that outlives its context never matches the unrelated context that later lands in the same slot.
inline int jh_ctx_slot(int id) { return id & (JH_MAX_CTX - 1); }
// slab cache (jh_core.hip): big device allocations of destroyed vectors, kept per device for the next vector of that size
hipError_t jh_slab_alloc(int device, size_t bytes, void **out, int role = 0);   // role: jh_context::alloc_role
hipError_t jh_device_malloc(int device, void **out, size_t bytes);     // hipMalloc that takes memory back from the cache when the driver says no
void jh_slab_free(int device, void *p, size_t bytes, hipStream_t probe_stream = nullptr);   // probe_stream: a stream of `device` for the write probe of slabs >= 4 GiB
void jh_slab_trim(int device);
size_t jh_slab_cached_bytes(int device);
int64_t jh_slab_probed(int device);    // cached slabs of the device with a write-probe record
int64_t jh_slab_last_choice();         // 100 * candidates + rank (by fill time, 0 = fastest) of the slab the last role-guided allocation took; -1: none

struct jh_context {
    bool ready = false;
    int id = -1;                       // slot + JH_MAX_CTX * generation
    bool primary = false;              // created by jh_init(device): jh_init(device) again returns it
    int device = -1;
    std::atomic<int64_t> live_handles{0};   // vectors (views included), operators and events created here and not yet destroyed
    int cu_count = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;      // the stream everything is enqueued on
    // reduction workspace (per-workgroup partials + a pinned host landing zone)
    double *red_dev = nullptr;         // JH_RED_SLOTS * 4 doubles
    double *red_host = nullptr;        // pinned, 8 doubles
    double *part_dev = nullptr;        // growable per-workgroup partials of the fused solver updates
    int64_t part_cap = 0;
    // the device tables of the latest batched broadcast (jh_bcast.hip: apply_many_batched), kept while the SAME batch is issued again
    struct bcast_group { const struct jh_bcast *bc; int64_t len, gx; int gcount, item_fast, shared_mask; size_t tbl_at, sc_at; bool wide_scal; };
    struct bcast_batch {
        std::vector<const void *> key;       // progs, dsts, xs -- the argument arrays, element by element
        std::vector<double> scal;
        int count = -1;
        int64_t gen = -1, knob_item = 0, knob_band = 0;
        void *dev = nullptr;
        size_t dev_cap = 0, tbl_bytes = 0;
        std::vector<bcast_group> groups;
    } bcast_last;
    void *scratch_dev = nullptr;       // growable scratch: dtmp / mtmp of the per-block loop (src/Jets.jl:1013, 1037)
    size_t scratch_cap = 0;
    unsigned *chain_sync = nullptr;    // chained one-pass step (k_tall_diag_bidiag_chain): [0] ticket counter, [1] unused, [2..] per-tile hand-off flags
    int64_t chain_sync_cap = 0;
    int64_t step_chain = -1;           // knob: one-pass step as chained row chunks: -1 automatic, 0 never, 1 always (when the shape allows)
    // tuning knobs (jh_tune_set)
    // 0 = pick from the problem size (jh_tall.hip: pick_fwd_shape / pick_adj_shape)
    int64_t fwd_group = 0;             // block rows streamed per workgroup (tall forward)
    int64_t fwd_unroll = 0;            // 16-byte vectors per thread per block (tall forward)
    int64_t fwd_wg = 0;                // threads per workgroup (tall forward)
    int64_t adj_unroll = 0;            // 16-byte vectors per thread (tall adjoint / fused normal)
    int64_t adj_depth = 0;             // block rows in flight per thread (tall adjoint / fused normal)
    int64_t adj_wg = 0;                // threads per workgroup (tall adjoint / fused normal)
    int64_t fwd_order = -1;            // -1: automatic; 0: sequential row sweep; 1: all row groups concurrent; k>1: bands of k row groups
    int64_t adj_rows_per_launch = 0;   // tall adjoint / fused normal: block rows per launch (0 = all rows in one launch); same bits either way
    int64_t adj_split = -1;            // split-row walk of the adjoint-shaped kernels: -1 automatic, 0 never (ordered, bit-exact), k > 1 parts
    int64_t last_adj_parts = 1;        // row parts of the most recent tall adjoint / fused normal / one-pass step (read-only knob)
    int64_t bcast_item_fast = -1;      // batched broadcasts with a shared operand: items as the fastest block index (-1 automatic, 0 never, 1 always)
    int64_t nt = 1;                    // nontemporal loads/stores on the streamed operands: 0 never, 1 unless the working set stays in the Infinity Cache (jh_stream_nt), 2 always
    int64_t nt_resident_mib = 256;     // knob: working sets up to this many MiB count as cache-resident for nt = 1 (0: none -- everything streams).  256 from
                                       // profiles/exp_r05_nt_small.txt: with coefficients + range vector <= 256 MiB temporal loads win (pair +7 %, one-pass step +7 ... +19 %,
                                       // LSQR iteration -9 %), from 512 MiB on nontemporal ones do (+10 ... +15 %)
    int64_t autotune = 1;              // time both grid walks of the tall forward once per large operator
    int64_t general_xcd = 1;           // general M x K kernels: 1 = XCD-aware (line, tile) decode from 32 MiB of input on, else line by line; 0 never; 2 always
    int64_t graphs = 1;                // replay launch-bound per-block loops as hipGraphs (jh_blockop.hip: run_loop_graphed)
    int64_t force_dist = 0;            // tests: jh_lsqr_solve_partitioned runs its exchange even with ONE rank (validates the pipelined path on a one-GPU box)
    int64_t small_loop = 1;            // operators mixing small DENSE children with other kinds: the whole block loop in one launch (0: the per-block loop)
    int64_t graph_replays = 0;         // read-only counter: hipGraphLaunch calls made by run_loop_graphed
    uint64_t buf_gen = 0;              // bumped whenever part_dev / scratch_dev is reallocated: captured graphs holding the old pointers are stale
    int64_t red_wgs = 16384;           // workgroups of a reduction launch (4 packs per lane in flight); profiles/sweep_r01_reduce.txt
    int64_t last_fwd_rows_per_wg = 0;  // block rows per workgroup of the most recent tall forward launch (read-only knob)
    int64_t last_adj_launches = 1;     // kernel launches of the most recent tall adjoint / fused normal call (read-only knob)
    int64_t last_fwd_walk = 0;         // grid walk used by the most recent tall forward launch (read-only knob)
    int64_t last_step_chain = 0;       // row chunks of the most recent one-pass step (0: the plain walk) (read-only knob)
    int64_t step_chunk = 0;            // knob: rows per chunk of the chained one-pass step: 0 automatic (32 rows x 256 lanes for all-diagonal operators, else 8), 8, 16, 32
    int64_t step_band = -1;            // knob: the chained one-pass step in column bands of this many tiles (-1: the default, 0: none -- tiles fastest over the whole row)
    int64_t grid_diag = 1;             // knob: M x K grids of plain diagonals on the branch-free kernel (0: the general kernels)
    int64_t grid_tile = 1;             // knob: ... register-tiled (k_grid_tile: R lines x one tile per workgroup): 1 automatic R, 2 / 4 / 8 that R, 0: k_grid_diag
    int64_t general_tile = 1;          // knob: grids of EQUAL elementwise blocks of any kinds register-tiled (k_general_tile: two lines x one tile per workgroup); 0: k_block_*_general_vec
    int64_t general_list = 1;          // knob: sparse grids walk lists of their non-zero steps (k_general_tile LIST): 1 automatic (jh_general.hip: launch_general_tile), 0 never, 2 always the four-line lists, 3 always the per-line lists
    int64_t last_general_list = 0;     // read-only: the most recent register-tiled general launch walked 0 no list, 1 its four-line lists, 2 its per-line lists
    int64_t sum_group = 16;            // knob: terms of a fused JetSum per FORWARD launch (16; 8 / 4 = round 3's / round 2's grouping, for A/B); the adjoint takes 8 (4)
    int64_t bcast_band = 0;            // knob: batched broadcasts with a shared operand in column bands of this many tiles (0: 32; 1: items fastest, no bands)
    int64_t general_band = 8;          // knob: tiles of every line the general M x K kernels walk before the next group of tiles starts: 8, 16, 32, 64
    int64_t fwd_ctiles = -1;           // knob: tall forward in column bands of this many tiles (-1: the shape's own, 0: none)
    int64_t sum_adj_group = 16;        // knob: terms of a fused JetSum ADJOINT per launch (each keeps its own accumulator): 16 (one row in flight; +1-2 % at 11-16 terms), or 8
    int64_t small_loop_max_kib = 512;  // knob: operators of small dense children whose matrices together reach this many KiB take the batched route (lists) instead of the one-launch loop
    int64_t dense_list = 1;            // knob: the dense children of dense_mixed operators from their LIST (k_gemv_*_list, late round 5); 0: the grid over every block pair (k_gemv_*_mixed)
    int64_t dense_list_split = 1;      // knob: ... 1 the rows pass picks its lane layout (column groups per workgroup: deterministic, tolerance parity), 0 columns in order (the sequential loop's bits)
    int64_t dense_grid = 0;            // knob, read by jh_blockop_create: M x K grids of uniform dense children on the list route (0; late round 5: 8 x 8 of 1024^2 1.8 -> 5.5 TB/s forward, 32 x 32 of 256^2 0.43 -> 4.6) or as one tall batch per block column (1: rounds 2-4)
    int64_t dense_direct = 1;          // knob: block-diagonal operators of dense children in ONE launch (the list kernels write the output vector); 0: scratch + combine
    int64_t dense_list_shared = 1;     // knob: the rows pass over children with columns off the 16-byte grid numbers its workgroups XCD by XCD and loads temporal (k_gemv_rows_list<SHARED>); 0: round 5's streaming loads
    int64_t dense_combine = 1;         // knob: operators whose non-zero blocks are all dense children combine the products from per-line lists (k_combine_dense, round 6); 0: the general kernel's table walk
    int64_t dense_list_rl_min = 0;     // knob: lists of dense children whose columns start off the 16-byte grid keep at least 2^this row lanes per workgroup in the rows pass (0: 6 = 1 KiB runs per column; 4 = round 5's rule)
    int64_t dense_list_cpw = 0;        // knob: columns per lane group of the list kernel of y = B' x: 0 by column length, 1 / 2 / 4
    int64_t last_dense_rl = 0;         // read-only: row lanes per workgroup of the latest rows pass of the list kernels (256: columns in order)
    int64_t dense_mixed = 1;           // knob: operators mixing big dense children with other kinds: one batched launch + one combine launch (0: the per-block loop)
    int64_t last_launches = 0;         // read-only: kernel launches of the most recent dense_mixed forward / adjoint
    int64_t dense_fused = 1;           // knob: tall / wide adjoint of many small uniform dense children on the fused kernel (k_gemv_cols_fused, round 4); 0: the three-launch path
    int64_t dense_fwd_wgs = 0;         // knob: workgroups the batched dense forward splits its columns for (0 = 2048)
    int64_t dense_gw = 0;              // knob: children per WAVE of that kernel (0 = from the operator's shape)
    int64_t last_dense_fused = 0;      // read-only: 1 when the most recent batched dense adjoint ran on the fused kernel
    double adj_in_scale = 1.0;         // internal: the MIXED tall adjoint multiplies every d_i by it first (jh_blockop_mul_adj_scaled on rows of several kinds); 1 outside that call
    int64_t tall_f = 1;                // knob: F(m) of a tall nonlinear operator of elementwise children on the tall tiling (jh_blockop_f): 1 yes, 0 the general kernels
    int64_t ua_nt = -1;                // knob: accesses of the tall kernels on rows off the 16-byte grid: -1 temporal there, nontemporal on aligned rows; 0 / 1 temporal / nontemporal always
    int64_t tall_unaligned = 1;        // knob: tall operators whose rows are not whole, 16-byte aligned packs (odd block lengths in one slab) on the under-aligned tall kernels (jh_tall.hip: tall_unaligned_ok); 0: the general kernels as before
    int64_t red_blocks_wave = 1;       // knob: per-block reductions of many short blocks with a wave per block (k_reduce_blocks_wave); 0: a workgroup per block
    int64_t adj_bare_chain = 1;        // knob: jh_blockop_mul_adj / _normal_mul of tall operators with rows of several kinds / off the 16-byte grid on the chain kernels (rows up to 4 MiB); 0: k_tall_diag_adj<MIXED>
    int64_t adj_thin_mixed = 1;        // knob: the MIXED tall adjoint on thin workgroups when the fat shape would not fill the chip (rows of 1-8 MiB); 0: round 5's rule
    int64_t grid_normal = 1;           // knob: (A', A) on an N x (2 .. 4) grid of equal elementwise blocks in one pass (jh_grid_normal.hip); 2: grids of plain diagonals only; 0: JH_ERR_UNSUPPORTED as in rounds 1-5 (the caller chains the two stages)
    int64_t fwd_anchor = -1;           // knob: the tall forward of rows that are not whole packs on lanes anchored to each row's own 16-byte grid (k_tall_fwd_anchored): -1 from 64 KiB rows on, 0 never, 1 always
    int64_t wide_twin = 1;             // knob: wide elementwise operators on their tall twin: 0 never (general kernels), 1 adjoint always + forward from 16 MiB blocks, 2 both always (tests)
    const double *step_coef_dev = nullptr;   // internal, set around the calls of the graph-captured LSQR loop: the one-pass step reads (alpha, beta) from
    const int *step_done_dev = nullptr;      // here instead of its arguments and returns at once when *step_done_dev != 0 (jh_lsqr.hip: lsqr_graph_impl)
    int step_skip_fold = 0;                  // internal: ... and leaves the fold of its per-workgroup partial sums (part_dev[0 .. last_step_parts)) to the caller
    int64_t last_step_parts = 0;             // per-workgroup partial sums the most recent one-pass step wrote
    int64_t lsqr_graph = 1;            // knob: small operators' LSQR loop with device-resident recurrences, replayed as a hipGraph (0: the host loop)
    int64_t last_lsqr_graph = 0;       // read-only: graph replays of the most recent jh_lsqr_solve (0: the host loop ran)
    int64_t walk_memory = 1;           // knob: a new tall operator starts with the forward walk the last operator of its shape chose (0: every operator measures)
    int64_t alloc_role = 0;            // knob: what the NEXT vectors of this context are for -- 0 unknown, 1 an operator's output (the cached slab that is fastest to write), 2 data written once and read from then on (the slowest to write: those read fastest); slabs >= 4 GiB only
    int64_t cg_dev = 1;                // knob: small operators' CGLS / CG-on-the-normal-equations loops on the fused kernels of cg_dev_impl (graph-replayed unless lsqr_graph = 0); 0: cgls_impl / cgnr_impl
    int64_t last_cg_graph = 0;         // read-only: graph replays of the most recent jh_cgls_solve / jh_cgnr_solve (0: a host-driven loop ran)
    int64_t cgls_trace = 0;            // knob (tests): jh_cgls_solve_team stamps every member's pass 1 of its first iteration with events ...
    int64_t last_cgls_overlaps = -1;   // read-only: ... and counts the consecutive members whose pass 1 began before the previous member's had finished (-1: not traced)
    int red_defer = 0;                 // internal, set around ONE reduction: enqueue it and its read-back, do not wait (jh_dot_begin / jh_dot_end)
    int adj_from_found = 0;            // internal, set around ONE call: the tall adjoint continues from what its output holds (the
                                       // forward of a wide operator through its tall twin: `_d .+=` into d as found, src/Jets.jl:1024); never split
};
jh_context &jh_ctx();                  // the calling thread's current context (a never-ready dummy before jh_init)
// Should a kernel that is launched again and again over the same operands (a solver's step, the fused A'A of CG) stream them NONTEMPORAL?
// Nontemporal loads do not stay in the 256 MiB Infinity Cache: right for operands far larger than it (every byte is used once per pass),
// wrong for an operator that fits -- its coefficients would come from HBM every iteration although the cache could hold them.
// Knob nt: 0 never, 2 always, 1 (default): nontemporal unless the working set of one pass is at most nt_resident_mib.
namespace jhb { bool grid_normal_ok(const jh_blockop *op, const void *y, const void *m); }   // jh_grid_normal.hip
inline bool jh_stream_nt(double working_set_bytes)
{
    const jh_context &c = jh_ctx();
    if (c.nt == 0) return false;
    if (c.nt >= 2) return true;
    return !(working_set_bytes <= (double)c.nt_resident_mib * 1048576.0);
}
jh_context *jh_ctx_by_id(int id);      // nullptr when there is no such context (also: the slot now holds a later generation)
// Destructors (jh_bvec_destroy, jh_blockop_destroy) run at moments a garbage collector chooses: they wait for the handle's
// context and free on its device WITHOUT making it the calling thread's current context (the device is restored when the scope
// ends, the thread's current context is never touched).
struct jh_quiesce_scope {
    int prev = -1;
    bool switched = false;
    explicit jh_quiesce_scope(int ctx);
    ~jh_quiesce_scope();
    jh_quiesce_scope(const jh_quiesce_scope &) = delete;
    jh_quiesce_scope &operator=(const jh_quiesce_scope &) = delete;
};
inline void jh_handle_born(int ctx) { if (jh_context *c = jh_ctx_by_id(ctx)) c->live_handles++; }
inline void jh_handle_died(int ctx) { if (jh_context *c = jh_ctx_by_id(ctx)) c->live_handles--; }
int jh_require_ready();                // the current context exists; re-selects its device if another library switched
int jh_enter_ids(const int *ids, int n);
// entry points: make the handles' context current.  Null handles are skipped (the entry point reports them itself); handles of
// different contexts in one call are refused.
template <class... H> inline int jh_enter(const H *...h)
{
    const int ids[] = {(h ? h->ctx : -1)...};
    return jh_enter_ids(ids, (int)sizeof...(H));
}
inline int jh_enter() { return jh_require_ready(); }

constexpr int JH_RED_SLOTS = 4096;     // max workgroups in a reduction launch
constexpr int JH_CHAIN_ERR_SLOT = 10;  // red_dev[10] (as an unsigned): sticky flag "a hand-off poll of the chained step ran into its bound" -- never expected;
                                       // copied to red_host[3] and checked wherever ||u||^2 is read back
constexpr int JH_NORMSQ_SLOT = 8;      // red_dev[8]: the deferred ||u||^2 accumulator (jh_normsq_reset / jh_normsq_read / jh_comm_allreduce_normsq)

static inline size_t jh_dtype_size(int dtype)
{
    switch (dtype) {
    case JH_F32: return 4;
    case JH_F64: return 8;
    case JH_C32: return 8;
    case JH_C64: return 16;
    default: return 0;
    }
}
static inline bool jh_dtype_complex(int dtype) { return dtype == JH_C32 || dtype == JH_C64; }

struct jh_bvec {
    int ctx = -1;                       // the context it was created in
    int dtype = JH_F32;
    int64_t nblocks = 0;
    int64_t length = 0;                 // total elements
    std::vector<int64_t> off;           // nblocks+1 cumulative element offsets (0-based)
    void *data = nullptr;               // device pointer to element 0
    bool owns = false;                  // hipFree on destroy
    bool uniform = false;               // all blocks the same length
    inline int64_t len(int64_t i) const { return off[i + 1] - off[i]; }
    inline char *ptr(int64_t elem) const { return (char *)data + (size_t)elem * jh_dtype_size(dtype); }
};

struct jh_event {
    int ctx = -1;
    hipEvent_t ev = nullptr;
};

// one entry per block, device-resident, column-major nrow x ncol
struct jh_dev_block {
    const void *coeff;
    double sre, sim;     // SCALE: the scalar
    int32_t kind : 16;        // (bit-fields of ONE 32-bit word, not two int16_t: gfx950 has no sub-dword scalar load, so an int16_t member of a
    int32_t real_scale : 16;  //  wave-uniform table entry is fetched with a VECTOR global_load_ushort + readfirstlane -- a memory round trip in front of
    int32_t adjoint;          //  every step's coefficient loads in the general kernels)      real_scale: SCALE, 1 = a REAL scalar (see jh_dev_block_of)
};
static_assert(sizeof(jh_dev_block) == 32, "one table entry = one s_load_dwordx8");
// A DENSE entry has no scalar: its two scalar slots hold where the child's product goes in the scratch vector of dense_mixed_apply (elements; forward in
// `sre`, adjoint in `sim`, as the bit patterns of two int64) -- one compact piece per dense child and direction (late round 5)
__host__ __device__ inline int64_t jh_dev_block_prod_off(const jh_dev_block &b, bool transposed)
{
    const double v = transposed ? b.sim : b.sre;
    int64_t o;
    __builtin_memcpy(&o, &v, sizeof(o));
    return o;
}
inline void jh_dev_block_set_prod_off(jh_dev_block &b, int64_t fwd, int64_t adj)
{
    __builtin_memcpy(&b.sre, &fwd, sizeof(fwd));
    __builtin_memcpy(&b.sim, &adj, sizeof(adj));
}
constexpr int JH_STEP_PAD = 12;        // padding entries of a step-list record: three groups of up to four steps in flight (k_general_tile LIST)

// the device form of a block description.  A SCALE block's scalar is Real unless it is flagged JH_SCALAR_COMPLEX or has a non-zero
// imaginary part: Julia's `a::Real * z` multiplies part by part, a Complex `a` takes the full product even when imag(a) == 0 (with THAT
// zero's sign).  The kernels tell the two apart by the flag `real_scale` (round 5; round 4 used sim = NaN as the sentinel, which made a
// Complex scalar with a NaN imaginary part look Real and drop the NaN Julia's full product gives -- `imag != 0.0` is true for NaN, so
// such a scalar is Complex here and keeps its NaN).
// (Blocks whose scalar is WIDE never reach the fused kernels: jh_blockop_create routes such operators through the per-block loop,
// whose scalar stage is the typed lincomb.)
inline jh_dev_block jh_dev_block_of(const jh_block_desc &b)
{
    jh_dev_block d{};
    d.coeff = b.coeff;
    d.sre = b.scale_re;
    const bool cplx = (b.scale_flags & JH_SCALAR_COMPLEX) || b.scale_im != 0.0;
    d.sim = cplx ? b.scale_im : 0.0;
    d.real_scale = cplx ? 0 : 1;
    d.kind = (int16_t)b.kind;
    d.adjoint = b.adjoint;
    return d;
}

// one dense child of a sparse / mixed operator, as the list kernels of jh_dense.hip see it: B is nr x nc, column-major; the pass computes y = B x
// (rows pass: x has nc elements, y nr) or y = B' x (cols pass: x has nr, y nc); x_off: where x starts in the operator's input vector, out_off: where
// y goes in the slabs of dense_mixed_apply (both in elements)
struct jh_dense_item {
    const void *A;
    int64_t nr, nc, x_off, out_off;
    int64_t line_off;                  // where the child's output LINE starts in the operator's output vector (elements): the direct mode writes there
};

struct jh_blockop {
    int ctx = -1;
    int dtype = JH_F32;
    int64_t nrow = 0, ncol = 0;
    std::vector<jh_block_desc> blocks;   // host copy
    std::vector<int64_t> row_len, col_len;
    std::vector<int64_t> row_off, col_off;   // cumulative (nrow+1 / ncol+1)
    jh_dev_block *dev_blocks = nullptr;      // nrow*ncol
    int64_t *dev_row_off = nullptr;          // nrow+1
    int64_t *dev_col_off = nullptr;          // ncol+1
    unsigned char *dev_row_touched = nullptr; // nrow: 1 when the block row has a non-zero block (the linear forward writes it)
    // step lists of the register-tiled general kernel (k_general_tile LIST): [dir][set], dir 0 forward (lines = block rows), 1 adjoint (lines = block
    // columns); set 0: groups of FOUR lines (a step = a summed block index at which one of the group's lines has a non-zero block), set 1: every line
    // on its own (a step = a non-zero block).  One record of step_stride ints per group: the count, the ascending indices, JH_STEP_PAD padding entries
    // (the kernel fetches indices two groups of steps ahead).
    int *dev_steps[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    int64_t step_stride[2][2] = {{0, 0}, {0, 0}};
    int64_t list_steps[2][2] = {{0, 0}, {0, 0}};   // steps of all groups together (compare with ngroups * nsum: the plain walk's steps)
    // classification for the fast paths
    bool tall = false;                       // ncol == 1
    bool uniform_rows = false;               // all row_len equal
    bool all_diag = false;                   // every block is an un-adjointed... DIAG (adjoint flag irrelevant up to conj)
    bool elementwise = false;                // no DENSE block (and no wide scalar)
    bool wide_scale = false;                 // a SCALE block with JH_SCALAR_WIDE on 32-bit elements: the per-block loop (its scalar stage computes in Float64)
    bool dense_batch = false;                // tall, >= 2 rows, every block an un-adjointed DENSE matrix of one shape: batched kernels (jh_dense.hip)
    bool dense_batch_wide = false;           // the same for ONE block row of >= 2 such children
    bool dense_batch_ragged = false;         // tall, every block an un-adjointed DENSE matrix with the same column count, row counts differ (one column chunk suffices)
    int64_t dense_max_nr = 0;
    bool dense_batch_grid = false;           // the same for an M x K grid (M, K >= 2): one tall batch per block column
    bool dense_aligned = false;              // ... and every matrix pointer on a 16-byte boundary
    bool launch_bound = true;                // the per-block loop of an operator with DENSE blocks is replayed as a hipGraph (it pays for small children)
    bool dense_mixed = false;                // DENSE blocks (none adjointed) next to other kinds or of differing shapes, too big for the one-launch loop: one
                                             // batched launch for all dense children + one launch of the general kernels (dense_mixed_apply)
    bool dense_mixed_aligned = false;        // ... every dense matrix, row length and row offset on 16 bytes
    // the dense children of a dense_mixed operator as lists, [direction: 0 forward, 1 adjoint][pass: 0 y = B x, 1 y = B' x] (jh_dense.hip: k_gemv_*_list)
    jh_dense_item *dev_items[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    // round 6: operators whose non-zero blocks are ALL dense (full grids, banded grids of dense children): the combine launch reads, per output line, the list of its
    // blocks' product places in summing order -- [dir] CSR: comb_ptr (nlines + 1 ints) and comb_off (where each product starts in the scratch vector, elements)
    int *dev_comb_ptr[2] = {nullptr, nullptr};
    int64_t *dev_comb_off[2] = {nullptr, nullptr};
    int64_t n_items[2][2] = {{0, 0}, {0, 0}}, items_max_out[2][2] = {{0, 0}, {0, 0}}, items_max_in[2][2] = {{0, 0}, {0, 0}};
    bool dense_direct[2] = {false, false};   // every output line of this direction holds exactly ONE non-zero block, a dense child (block-diagonal operators and
                                             // their permutations; forward: or none; adjoint: every column has one): the list kernels write the output vector
                                             // themselves -- one launch, no scratch, no combine (the same product, rounded, then the same addition to d as found)
    std::vector<int64_t> prod_off[2];        // per block (column-major like `blocks`; DENSE blocks of a dense_mixed operator only, else -1): where its product goes
    int64_t prod_total[2] = {0, 0};          // elements of the product scratch vector per direction
    double dense_bytes = 0.0;                // bytes of all DENSE children together
    bool small_loop = false;                 // DENSE blocks (adjointed or not) mixed with elementwise kinds, every matrix small: the whole loop in ONE launch (k_block_loop_small)
    int64_t *dev_dims = nullptr;             // nrow*ncol x {nr, nc} of the described operators (small_loop only)
    bool nonlinear = false;                  // has a SQUARE block (JopNl child)
    bool pointed = false;                    // jh_blockop_point has been called (SQUARE blocks have their mo)
    int64_t table_gen = 0;                   // bumped whenever the device block table is rewritten (jh_blockop_point): what is derived from it (a fused chain's row table) is rebuilt
    int *dev_rows_nz = nullptr;              // a tall operator of elementwise rows of several kinds: its non-ZERO rows, ascending (the forward of an operator with many
    int64_t n_rows_nz = 0;                   // zero rows -- muted shots -- launches workgroups for those only)
    bool coeff_aligned16 = true;             // every coefficient array of a DIAG / SQUARE block starts on a 16-byte boundary (jh_blockop_create; again at jh_blockop_point,
                                             // which moves the SQUARE blocks' arrays): what the per-call route tests used to find by walking all M x K descriptors
    bool lens_aligned16 = true;              // every row and column length a multiple of 16 bytes
    bool lens_hold_a_pack = true;            // every non-empty row and column length at least 16 bytes (the general kernels' under-aligned packs, jh_general.hip: general_vec_ok)
    bool coeff_scalar_aligned = true;        // every such array starts on a multiple of its scalar's size (what the under-aligned tall route needs, jh_tall.hip: tall_unaligned_ok)
    bool diag_strided = false;               // coeff[i] = coeff[0] + i*stride bytes
    // hipGraph replay of the per-block loop (operators with DENSE blocks: 2 launches per block), keyed on the vectors' addresses
    struct LoopGraph { const void *out; const void *in; int mode; int seen; uint64_t gen; hipGraphExec_t exec; };
    mutable std::vector<LoopGraph> loop_graphs;
    mutable int fwd_walk = -1;               // autotuned tall-forward shape: -1 untried, else an index into k_fwd_candidates (jh_tall.hip)
    mutable bool walk_inherited = false;     // ... taken over from an earlier operator of the same shape (jh_tall.hip: walk_recall)
    bool walk_measure_again = false;         // jh_blockop_tune_set(op, "fwd_walk", -1): this operator measures for itself
    // lazy autotune (jh_tall.hip: lazy_*): the first real calls each run ONE candidate between two events -- no extra
    // launches, no host synchronisation -- and finished timings are harvested with hipEventQuery on later calls
    struct LazyTune {
        static constexpr int SLOTS = 24;     // up to 10 candidates x 2 passes + a play-off of 4
        static constexpr int MAXC = 12;      // candidates whose best times are kept
        hipEvent_t ev[SLOTS][2] = {};
        float ms[SLOTS] = {};
        unsigned char state[SLOTS] = {};     // 0 not launched, 1 in flight, 2 measured, 3 failed
        int launched = 0;
        // round 3: when the runner-up is within 3 % of the winner, the two are timed twice more, alternating (the play-off slots
        // follow the regular ones), before the choice is made; the candidates' best times are kept for the periodic re-check
        int playoff[2] = {-1, -1};
        float best_ms[MAXC] = {};
        // periodic re-check of the chosen candidate (every 64th call is timed; three slow samples in a row rotate the runner-up in)
        int64_t calls = 0;
        hipEvent_t rc_ev[2] = {};
        bool rc_in_flight = false;
        int rc_slow = 0;                     // consecutive samples more than 3 % slower than the best other candidate's record
        int switches = 0;
    };
    mutable LazyTune fwd_tune;               // tall forward: K_FWD_CANDIDATES shapes x 2 passes -> fwd_walk
    mutable LazyTune step_tune;              // one-pass step: plain / tile map / chained x 2 passes (+ a warm-up) -> step_mode
    mutable LazyTune gen_tune[2];            // sparse grids on the register-tiled general kernel, per direction: four-line lists / per-line lists / plain walk -> gen_walk
    mutable void *grid_words = nullptr;      // the packed block table of a mixed N x (2 .. 4) grid (jh_grid_normal.hip), built on first use
    mutable struct jh_chain *bare_chain[2] = {nullptr, nullptr};   // the library's own ADJOINT / NORMAL chains with empty stage lists (jh_tall_chain.hip: bare_chain), built on first use
    mutable int gen_walk[2] = {-1, -1};      // -1 untried, 0 the four-line step lists, 1 the per-line lists, 2 the plain walk (jh_general.hip: launch_general_tile)
    mutable int upd_walk = -1;               // same for the fused forward update (timed on its first two real calls)
    mutable int upd_trials = 0;
    mutable float upd_ms[2] = {0.f, 0.f};
    mutable int step_mode = -1;              // one-pass step: -1 untried (measured lazily over its first seven eligible calls), 0 plain walk, 1 XCD-contiguous tiles, 2 chained row chunks
    mutable int64_t step_span = 0;           // the call shape (scalars per call) the trials are being run on

    int64_t diag_stride_elems = 0;
    // a WIDE operator (1 x K) of elementwise children: its adjoint m_j = A_1j' d (1051: direct write, zero blocks skipped) IS the
    // forward of the tall K x 1 operator of the blocks A_1j' -- `twin` is that operator (same coefficient arrays), so the wide
    // adjoint runs on the tall forward kernels with their measured grid walk instead of the general kernel
    jh_blockop *twin = nullptr;
};

bool jh_blockop_tall_fast(const jh_blockop *op, const void *rng_ptr, const void *dom_ptr);   // tall, all DIAG, equal 16-byte aligned blocks
bool jh_blockop_tall_step_ok(const jh_blockop *op, const void *rng_ptr, const void *dom_ptr);   // that, or rows off the 16-byte pack grid (whole-vector fused passes only; jh_tall.hip)
// Device-resident state of CG on the normal equations / CGLS (jh_lsqr.hip, round 4): the recurrences' scalars live here, one-thread
// epilogues of the fold kernels update them with the same fp64 operations, in the same order, as the host-driven loops (the two
// update functions are shared `__host__ __device__` code), and every vector kernel reads its coefficients from it -- so an
// iteration has fixed launch parameters, is captured once as a hipGraph and replayed; a finished solve turns the kernels into no-ops.
struct jh_cg_dev {
    double gamma, gamma0, rr, bnorm, damp2, atol, btol;
    double alpha, bk, delta;
    double coef_step[2];               // CGLS: (-alpha, 1) for the one-pass step r <- r - alpha A p
    int itn, istop, done, skip_p, maxiter, force, cgls, pad;
};
// lane t of a 256-lane workgroup adds p[t], p[t + 256], p[t + 512], ... in THAT order (the folds of the CG kernels all use this, so
// that every one of them forms the same bits); four loads in flight, a load past the end re-reads the last element and adds 0
__device__ inline double jh_strided_sum256(const double *__restrict__ p, int64_t n)
{
    double v = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const int64_t i1 = i + 256 < n ? i + 256 : n - 1, i2 = i + 512 < n ? i + 512 : n - 1, i3 = i + 768 < n ? i + 768 : n - 1;
        const double a0 = p[i], b1 = p[i1], b2 = p[i2], b3 = p[i3];
        v += a0;
        v += i + 256 < n ? b1 : 0.0;
        v += i + 512 < n ? b2 : 0.0;
        v += i + 768 < n ? b3 : 0.0;
    }
    return v;
}
// the two scalar updates of an iteration (Hestenes & Stiefel's recurrences), shared by the host-driven loop and the device kernels
__host__ __device__ inline void cg_s1(jh_cg_dev *st, double pap)
{
    st->itn++;
    st->delta = pap;
    if (!(pap > 0) || !((pap - pap) == 0.0)) {                            // breakdown: p in the null space, or the recurrences have left the range
        st->istop = 6;
        st->itn--;
        st->done = 1;
        return;
    }
    st->alpha = st->gamma / pap;
    st->coef_step[0] = -st->alpha;
    st->coef_step[1] = 1.0;
}

__host__ __device__ inline void cg_s2(jh_cg_dev *st, double ssum, double rsum, double *history)
{
    if (st->cgls) {
        st->rr = rsum;                                                    // ||r||^2 from the step itself
    } else {
        st->rr -= st->alpha * st->gamma;                                  // ||r_k||^2 = ||r_{k-1}||^2 - alpha_k gamma_{k-1} (exact for CG)
        if (st->rr < 0) st->rr = 0;
    }
    st->bk = ssum / st->gamma;
    st->gamma = ssum;
    st->skip_p = 0;
    const double rnorm = sqrt(st->rr), arnorm = sqrt(st->gamma);
    const int itn = st->itn;
    if (history) { history[2 * (itn - 1)] = rnorm; history[2 * (itn - 1) + 1] = arnorm; }
    int istop = st->istop;
    if (itn >= st->maxiter) istop = 7;
    if (arnorm <= st->atol * sqrt(st->gamma0)) istop = 2;
    if (rnorm <= st->btol * st->bnorm) istop = 1;
    st->istop = istop;
    if (istop && !(st->force && itn < st->maxiter && st->gamma > 0)) st->done = 1;
}

// jh_tall.hip: one launch = [p <- s + bk p unless st->skip_p] ; y = A'A p (+ damp2 p) with the bits of jh_blockop_normal_mul (+ the
// lincomb) ; one fp64 partial of <p, y> per workgroup of 256 packs.  *nparts = the number of partials written.
int64_t jh_bidiag_step_parts(const jh_blockop *op);   // row ranges of the one-pass step over the whole domain (1: one plain launch)
int jh_launch_cg_normal(const jh_blockop *op, jh_bvec *p, const jh_bvec *s, jh_bvec *y, const jh_cg_dev *st, double *partials, int64_t *nparts);
void jh_bcast_clear_cache();            // jh_bcast.hip: unload every JIT-compiled broadcast program (jh_shutdown)
int jh_chain_err_check();               // jh_core.hip: fails loudly if the chained step's sticky error word (copied to red_host[3]) is set
int jh_ensure_partials(int64_t n);     // grows ctx.part_dev to >= n doubles (may synchronise + reallocate)
extern "C" int jh_dot_begin(const jh_bvec *x, const jh_bvec *y);      // jh_vecops.hip: jh_dot in two halves (enqueue / wait + read), internal
extern "C" int jh_dot_end(const jh_bvec *x, double *re, double *im);
// vecops entry used by blockop for generic pieces
int jh_launch_fill_range(void *ptr, int dtype, int64_t count, double re, double im);
int jh_launch_copy_bytes(void *dst, const void *src, size_t bytes);
int jh_launch_hadamard_raw(void *dst, const void *x, const void *y, int dtype, int64_t count, int conj_x);
// dst = (2 .* mo) .* x  (conj: conj.(2 .* mo) .* x): the Jacobian of d .= m.^2 about mo
int jh_launch_square_jvp_raw(void *dst, const void *mo, const void *x, int dtype, int64_t count, int conj_mo);
extern std::atomic<int64_t> jh_bvec_generation;   // bumped whenever a vector handle is DESTROYED (enough: a cached table can only go stale through a handle it names dying; a handle never changes otherwise)
int jh_launch_lincomb_raw(void *dst, int dtype, int64_t count, int k, const double *cre, const double *cim, const void *const *x, const int32_t *flags = nullptr);
// dense child operator (jh_dense.hip): y = A x (rows) or y = A^H x / A^T x (cols); A column-major nr x nc
int jh_launch_gemv(const void *A, int64_t nr, int64_t nc, int dtype, void *y, const void *x, int adjoint);
int jh_launch_gemv_batched(const jh_dev_block *dev_blocks, int64_t nchild, int64_t nr, int64_t nc, int dtype, void *y, const void *x,
                           int adjoint, bool aligned, bool wide, const int64_t *dev_row_off = nullptr);   // jh_dense.hip: every child of a tall (or wide) operator of uniform dense blocks at once
int jh_launch_gemv_mixed_all(const jh_dev_block *blocks, int64_t nrow, int64_t ncol, int64_t rows_max_out, int64_t cols_max_out, int dtype, void *slabs,
                             int64_t slab_stride, const void *x, int transposed, bool aligned, const int64_t *dev_row_off,
                             const int64_t *dev_col_off);           // jh_dense.hip: every dense child of a mixed operator in one or two launches
int jh_launch_gemv_list(const jh_dense_item *items, int64_t nitems, int64_t max_out, int64_t max_in, int pass, int dtype, void *slabs, const void *x,
                        int aligned, void *direct_out = nullptr, int add_found = 0);                                  // jh_dense.hip: the dense children of one direction and pass of a dense_mixed operator, from their list
int jh_ensure_scratch(size_t bytes, void **out);
extern "C" int jh_comm_exists(int *yes);  // jh_comm.hip (internal): the current context's communicator: 0 none, 1 of jh_comm_init_rank, 2 member of a single-process team
extern "C" int jh_comm_destroy(void);   // jh_comm.hip; jh_shutdown tears the communicator down first   // growable device scratch (block-loop temporaries)
