/*
 * jetship.h -- C ABI of libjetship.so: MI355X (gfx950) native block-operator mul! path for the
 * Jets.jl operator API.
 *
 * This is the drop-in boundary.  Jets.jl (/root/reference/src/Jets.jl) is pure Julia and has no
 * FFI of its own; its only extension point is the Jet keyword constructor (src/Jets.jl:170-188)
 * whose closures are called from mul! (src/Jets.jl:390-392).  A Julia maintainer binds the entry
 * points below with `ccall` inside methods of JetBlock_df!/JetBlock_df'! (src/Jets.jl:1010-1057),
 * BlockArray (src/Jets.jl:809-924) and JetBSpace (src/Jets.jl:736-807); INTEGRATION.md shows the
 * stubs.  Each entry point names the reference lines it replaces.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes only.  Every function returns a jh_status (0 = ok);
 *    jh_last_error() returns a thread-local message for the last failure (the Julia wrapper turns
 *    it into `error(msg)`, src/Jets.jl:131,179,1116).
 *  - Block indices and element offsets are 0-based here; the Julia wrapper converts (1-based
 *    inclusive ranges of src/Jets.jl:742-748  <->  offset = start-1, len = stop-start+1).
 *  - A CONTEXT is one GPU + one HIP stream + the library's workspaces and knobs for it.  jh_init(device)
 *    creates the device's primary context; the usual deployment is one process per GPU with exactly
 *    that one.  One process may also drive SEVERAL GPUs (a single Julia session):
 *    jh_init on each device (or jh_context_create for extra contexts), and every handle remembers the
 *    context it was created in -- an entry point that takes handles switches to THEIR context
 *    (hipSetDevice included) and refuses handles of different contexts in one call.  Calls without a
 *    handle (jh_bvec_create, jh_tune_set, jh_synchronize, ...) act on the calling thread's CURRENT
 *    context: the one chosen by jh_context_use / jh_set_device / jh_init, else the one the thread's
 *    last handle call switched to, else the first context created.
 *  - All work of a context is enqueued on its one HIP stream (jh_get_stream/jh_set_stream);
 *    functions return after enqueue, except those that return a scalar or copy to host memory,
 *    which synchronise the stream first.
 *  - Handles are opaque, created/destroyed explicitly; the library never frees caller memory.
 *    Destroying a vector invalidates borrowed block pointers and views of it.
 *  - A block vector ("bvec", the device BlockArray) is ONE contiguous slab; block i lives at
 *    element offset sum(len[0..i-1]) -- the layout of JetBSpace.indices (src/Jets.jl:742-748), so
 *    reshape(flat, R) (src/Jets.jl:1112) and convert(Array, x) (src/Jets.jl:862-868) are free.
 *    A plain N-d array (the domain of a one-column block operator, src/Jets.jl:927) is a bvec
 *    with one block.
 */
#ifndef JETSHIP_H
#define JETSHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libjetship.so is built with -fvisibility=hidden: exactly the entry points declared in this header are exported (round 6; tests/test_abi_symbols.py
 * compares the library's dynamic symbol table with this file). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define JETSHIP_ABI_VERSION 4   /* 2 (round 4): jh_block_desc.scale_flags, jh_lincomb_typed; 3 (round 5): jh_blocksum_mul[_adj]_typed; 4 (round 6): jh_chain_*, jh_norm_blocks / jh_dot_blocks */

typedef enum {
    JH_OK = 0,
    JH_ERR_INVALID = 1,      /* bad argument: index out of range, length/dtype mismatch, null handle */
    JH_ERR_HIP = 2,          /* a HIP runtime call failed (message carries hipGetErrorString)       */
    JH_ERR_NOMEM = 3,        /* device or host allocation failed                                    */
    JH_ERR_UNSUPPORTED = 4,  /* valid request the device path does not implement                    */
    JH_ERR_STATE = 5,        /* library not initialised / already shut down                         */
    JH_ERR_COMM = 6          /* RCCL failure                                                        */
} jh_status;

typedef enum { JH_F32 = 0, JH_F64 = 1, JH_C32 = 2, JH_C64 = 3 } jh_dtype;

/* device-native block kinds: the closed set of child operators the fused kernels understand.
 * The Julia side recognises them by typeof(df!) exactly like iszero/isblockop do
 * (src/Jets.jl:949, 1085, 1097). */
typedef enum {
    JH_OP_ZERO = 0,      /* JopZeroBlock, src/Jets.jl:941-942 (skipped by the block loops: 1022, 1047) */
    JH_OP_IDENTITY = 1,  /* d .= m                                                                     */
    JH_OP_SCALE = 2,     /* d .= a*m ; adjoint m .= conj(a)*d, src/Jets.jl:1159-1160                   */
    JH_OP_DIAG = 3,      /* d .= diagonal .* m ; adjoint conj.(diagonal) .* d, test/runtests.jl:3-4    */
    JH_OP_DENSE = 4,     /* d .= A*m ; adjoint A'*d (column-major A), test/runtests.jl:27-28           */
    JH_OP_SQUARE = 5     /* NONLINEAR child (JopNl): f! d .= m.^2 ; Jacobian at mo dd .= 2 .* mo .* dm,
                          * test/runtests.jl:19-24 (JopBar); adjoint of the Jacobian conj.(2 .* mo) .* dd
                          * (== df! for real eltypes, the fixture's default src/Jets.jl:184-186).
                          * coeff is the block of mo the child is linearised about: set by jh_blockop_point. */
} jh_opkind;

typedef struct {
    int32_t kind;        /* jh_opkind                                                        */
    int32_t adjoint;     /* 1: this block is the JopAdjoint of the described operator        */
    const void *coeff;   /* DEVICE pointer. DIAG: nr elements. DENSE: column-major nr x nc.  *
                          * SQUARE: ignored at create (see jh_blockop_point)                 */
    double scale_re;     /* SCALE: a                                                         */
    double scale_im;
    int64_t nr, nc;      /* range / domain length of the described (un-adjointed) operator   */
    int32_t scale_flags; /* SCALE: JH_SCALAR_* -- the TYPE of a, which Julia dispatches on    */
    int32_t reserved;    /* 0                                                                */
} jh_block_desc;

/* The TYPE of a scalar that arrives as (re, im) doubles (src/Jets.jl:1159 `d .= a * m`, 889-911 broadcast): Julia's arithmetic follows it.
 *   JH_SCALAR_COMPLEX  a is a Complex: the full complex product (re*re' - im*im', re*im' + im*re') even when its imaginary part is zero.
 *                      Without the flag an imaginary part of exactly zero stands for a Real: `a::Real * z` multiplies part by part (no
 *                      0 * Inf = NaN, signed zeros kept); a non-zero imaginary part implies the flag.
 *   JH_SCALAR_WIDE     a is Float64-based (Float64 / ComplexF64) and the elements are 32-bit: Julia promotes, computes in Float64 and rounds
 *                      ONCE when the result is stored into the Float32 / ComplexF32 destination.  Without it the scalar is converted to the
 *                      element type first (`T(a) * x`: Float32, integers, irrationals, or any scalar against 64-bit elements -- where the
 *                      flag is ignored).  In jh_lincomb_typed a sum with a wide term is a Float64 sum from there on, as in Julia.
 * A binding sets them from the scalar's type; 0 is what a binding that does not know passes. */
enum { JH_SCALAR_COMPLEX = 1, JH_SCALAR_WIDE = 2 };

typedef struct jh_bvec jh_bvec;
typedef struct jh_blockop jh_blockop;
typedef struct jh_event jh_event;

/* ---------------------------------------------------------------- context ------------------- */
int jh_abi_version(void);
const char *jh_last_error(void);
int jh_device_count(int *count);
int jh_init(int device);                 /* the device's primary context (created on first call) becomes current; idempotent */
int jh_shutdown(void);                   /* destroys every context of the process                                            */
/* several contexts in one process (see Conventions).  Context ids are small integers (< 64). */
int jh_context_create(int device, int *ctx);      /* an ADDITIONAL context (own stream, workspaces, knobs) -- also on a device
                                                   * that already has one; it becomes current */
int jh_context_use(int ctx);                      /* make it the calling thread's current context (hipSetDevice included) */
int jh_context_current(int *ctx, int *device);
int jh_context_destroy(int ctx);                  /* refused (JH_ERR_STATE) while vectors / operators / events created in it are alive */
int jh_set_device(int device);                    /* = jh_context_use(the primary context of `device`) */
int jh_device_info(char *name, int name_cap, int64_t *total_mem, int64_t *free_mem, int *cu_count);
/* Slab cache.  The reference's style allocates range-sized temporaries all the time (`A*m` returns a fresh vector, `zeros(range(A))`
 * per composite stage, src/Jets.jl:399, 526-533), and hipMalloc of a 64 GiB slab costs 2-6 seconds on this machine whenever the
 * runtime goes to the driver for it (profiles/exp_r03_alloc_cost.txt).  So the device memory of a destroyed vector of 16 MiB or more
 * is kept (per device; the cache holds at most all but 32 GiB of the device, and cached plus live memory always leave the device's
 * last 8 GiB to other allocators of the process -- RCCL, torch; knob "slab_free_floor_mib") and handed to the next jh_bvec_create of
 * exactly that size -- after a device-wide synchronisation, as hipFree would have made; when a limit is passed or the driver refuses
 * an allocation of the library, slabs go back to it, chosen to cover the shortfall with about the fewest bytes (re-used device memory
 * is cleared by the driver at about 20 GB/s: that is where the seconds go).  jh_device_info counts cached memory as free.
 * jh_trim() returns it to the driver (before another library of the process needs the memory); jh_tune_set("slab_cache", 0) turns
 * the cache off (and empties it, on every device); jh_tune_get("slab_cached_mib") reads what it holds.
 * WHICH cached slab (round 4): the time to write a 64 GiB slab is a property of the slab on this chip (fill 7.2 or 6.4 TB/s; the
 * tall forward into it 20.6-21.5 or 23.8-24.8 ms; the fast-write slabs read 3 % slower).  A slab of 4 GiB or more is probed once
 * when it enters the cache (two fills of the dead memory, the second timed), and jh_tune_set("alloc_role", r) before an allocation says
 * what the vector is for: 1 = an operator's output (the cached slab of its size with the fastest fill), 2 = data written once and
 * read from then on (the slowest), 0 (default) = no preference, the most recently freed.  The host bindings set it around `A*m`
 * (src/Jets.jl:399), stage temporaries (526-533) and rand / randn (105-108).  jh_tune_get("last_alloc_choice") = 100 x probed
 * candidates + rank by fill time of the one taken (-1: no choice was made), "slab_probed" = cached slabs with a record. */
int jh_trim(void);
int jh_get_stream(void **hip_stream);    /* hipStream_t the library enqueues on                   */
int jh_set_stream(void *hip_stream);     /* NULL restores the library's own stream                */
int jh_synchronize(void);
/* HIP events on the library stream: roofline timing in bench.py */
int jh_event_create(jh_event **ev);
int jh_event_record(jh_event *ev);
int jh_event_elapsed_ms(jh_event *start, jh_event *stop, float *ms);   /* synchronises on stop */
int jh_event_destroy(jh_event *ev);

/* ---------------------------------------------------------------- block vectors ------------- */
/* zeros(R::JetBSpace) / Array(R) storage, src/Jets.jl:922-924; JetBSpace ctor 739-750. Zero-filled. */
int jh_bvec_create(int64_t nblocks, const int64_t *block_len, int dtype, jh_bvec **out);
/* Array(R) = Array{T,N}(undef, size(R)), src/Jets.jl:105: the same vector WITHOUT the zero fill (12 ms per 64 GiB) -- its contents are
 * whatever the memory held.  For outputs a call overwrites entirely: `A*m` of a tall operator without zero blocks (src/Jets.jl:399 passes
 * zeros(range(A)) because a general df! may accumulate, 1024; a one-column operator without zero blocks overwrites every row, 1026). */
int jh_bvec_create_uninit(int64_t nblocks, const int64_t *block_len, int dtype, jh_bvec **out);
/* reshape(x::AbstractArray, R::JetBSpace), src/Jets.jl:1112: block view over caller-owned device memory */
int jh_bvec_wrap(void *device_ptr, int64_t nblocks, const int64_t *block_len, int dtype, jh_bvec **out);
/* view of blocks [first, first+count) of parent sharing memory (getblock(x,i) by reference, 914) */
int jh_bvec_view(jh_bvec *parent, int64_t first_block, int64_t count, jh_bvec **out);
int jh_bvec_destroy(jh_bvec *v);
int jh_bvec_info(const jh_bvec *v, int64_t *nblocks, int64_t *length, int *dtype, void **device_ptr);
int jh_bvec_context(const jh_bvec *v, int *ctx, int *device);   /* the context (and its device) the vector lives in */
/* indices(x,i) / getblock(x,i) as a borrowed device pointer, src/Jets.jl:858, 914 */
int jh_bvec_block(const jh_bvec *v, int64_t iblock, int64_t *offset, int64_t *len, void **device_ptr);

/* getblock!(x, i, dst), src/Jets.jl:915 */
int jh_getblock_copy(const jh_bvec *v, int64_t iblock, void *dst, int dst_on_device);
/* setblock!(x, i, src::AbstractArray), src/Jets.jl:916 */
int jh_setblock_copy(jh_bvec *v, int64_t iblock, const void *src, int src_on_device);
/* setblock!(x, i, scalar), src/Jets.jl:916 with a Number (test/runtests.jl:518-519) */
int jh_setblock_fill(jh_bvec *v, int64_t iblock, double re, double im);
/* fill!(x, a), src/Jets.jl:880-885 */
int jh_fill(jh_bvec *v, double re, double im);
/* y .= x for two block vectors of equal length (src/Jets.jl:905-911 with bc = identity) */
int jh_copy(jh_bvec *dst, const jh_bvec *src);
/* convert(Array, x) (src/Jets.jl:862-868) and its inverse, on an element range of the slab */
int jh_download(const jh_bvec *v, int64_t offset, int64_t count, void *host_dst);
int jh_upload(jh_bvec *v, int64_t offset, int64_t count, const void *host_src);
/* Page-locked host memory for the copies above.  Measured on the MI355X box (profiles/bench_pcie_r01.txt): 55-57 GB/s
 * each way for warm host arrays, pageable or page-locked alike, but 8-9 GB/s into a freshly allocated pageable array whose
 * pages are first touched by the copy; a page-locked buffer never pays that.  jh_host_alloc/free give the host language a
 * pinned buffer to wrap as an array (Julia: unsafe_wrap); jh_host_register/unregister pin an EXISTING host array in place.
 * Neither changes any result; both are optional. */
int jh_host_alloc(size_t bytes, void **out);
int jh_host_free(void *ptr);
int jh_host_register(void *ptr, size_t bytes);
int jh_host_unregister(void *ptr);
/* rand(R) (src/Jets.jl:922-924) from the counter-based generator of SURVEY.md 8d:
 * element k = mix64(key + (k+1)*0x9E3779B97F4A7C15), key = mix64(seed*0x9E37...+stream);
 * index_base shifts k so a row-partitioned shard reproduces its slice of the global vector. */
int jh_fill_uniform(jh_bvec *v, uint64_t seed, uint64_t stream, int64_t index_base);

/* randn(R) (src/Jets.jl:105-108, 922-924): Box-Muller over the same counter generator (two draws per scalar lane;
 * complex lanes scaled by 1/sqrt(2)).  Reproducible on the CPU to transcendental-function accuracy, not bitwise. */
int jh_fill_normal(jh_bvec *v, uint64_t seed, uint64_t stream, int64_t index_base);
/* abs.(x) into a real vector of the same length (test/runtests.jl:545-547) */
int jh_abs(jh_bvec *dst_real, const jh_bvec *x);

/* BlockArray broadcast, src/Jets.jl:889-911.  dst = c0*x0 .+ c1*x1 .+ ... evaluated left to right in
 * eltype T (each product and each sum rounded, no FMA); coef is k (re,im) pairs; dst may alias any x. */
int jh_lincomb(jh_bvec *dst, int k, const double *coef_re_im, const jh_bvec *const *x);
/* the same with the coefficients' TYPES (k x JH_SCALAR_*; NULL: jh_lincomb).  A wide coefficient makes its product, and every sum after
 * it, Float64 arithmetic rounded once on the store -- Julia's promotion in `x .= a .* u .+ b .* v` with a::Float64, u::Vector{Float32}. */
int jh_lincomb_typed(jh_bvec *dst, int k, const double *coef_re_im, const int32_t *coef_flags, const jh_bvec *const *x);
/* dst = x .* y -- the masks of dot_product_test, src/Jets.jl:1215-1219.  conj_x is a flag word: bit 0 conj.(x) .* y;
 * bit 1 (2 .* x) .* y, the Jacobian of d .= m.^2 about x (test/runtests.jl:20); 3 = conj.(2 .* x) .* y. */
int jh_hadamard(jh_bvec *dst, const jh_bvec *x, const jh_bvec *y, int conj_x);
/* BlockArray broadcast for ANY elementwise expression, src/Jets.jl:889-911, fused into one pass over the slabs.
 * `expr` is a C expression over x0..x{nvec-1} (the elements of the vector operands, type T), s0..s{nscal-1} (scalars of
 * type T) and literals -- what the host language prints from its broadcast tree (Julia: a Broadcasted{BlockArrayStyle}),
 * e.g. "s0*x0 + x1/x2" or "exp(-abs2(x0)) * conj(x1)".  Available: + - * /, the HIP math library (exp, log, sqrt, sin, cos,
 * tanh, pow, fmin, fmax, ...), and conj/real/imag/abs/abs2/sign; complex T has + - * / conj abs abs2 exp.  It is compiled once
 * per (expr, dtype, nvec, nscal) with hiprtc for gfx950 -- the device-side twin of Julia compiling the broadcast kernel -- with
 * -ffp-contract=off, so every operation is rounded as written (no FMA): for + - * / the result has the bits of the reference's
 * CPU broadcast.  Programs are cached for the life of the context.  jh_bcast_check only compiles (no device needed).
 * dst may alias any operand.  Operands are whole vectors of dst's length (a BlockArray is one slab: src/Jets.jl:899-904
 * pairs blocks of equal index, which is the same thing). */
typedef struct jh_bcast jh_bcast;
int jh_bcast_check(const char *expr, int dtype, int nvec, int nscal);
int jh_bcast_compile(const char *expr, int dtype, int nvec, int nscal, jh_bcast **out);
/* Mixed element types (src/Jets.jl:899-904 pairs the blocks of every BlockArray operand whatever its eltype): bit k of real_mask
 * says vector operand k of a COMPLEX program is REAL of the matching precision (Float32 in a ComplexF32 program, ...) -- a real
 * mask or weight on a complex vector; bit nvec + k says SCALAR k is a real number (its imaginary slot is ignored).  Real (x) complex
 * arithmetic is Julia's: a*(x + iy) = (a*x) + i(a*y), a + z adds to the real part -- so a real scalar never meets the 0 * Inf = NaN
 * of the four-multiplication formula.  real_mask == 0 is jh_bcast_compile.
 * Outside a compiled program the scalar's type travels as JH_SCALAR_* flags (jh_lincomb_typed, jh_block_desc.scale_flags).
 * jh_bcast_compile_typed adds wide_mask: bit k says scalar k is Float64-based (JH_SCALAR_WIDE) in a 32-bit program -- it enters the
 * expression as a double, every operation it meets is a double operation (Julia's promotion is the language's own), and the element
 * type comes back with the ONE rounding of the store: `x .= a .* u .+ v` with a::Float64, u, v::Vector{Float32}.  Ignored for 64-bit
 * programs; wide_mask == 0 is jh_bcast_compile_mixed. */
int jh_bcast_compile_mixed(const char *expr, int dtype, int nvec, int real_mask, int nscal, jh_bcast **out);
int jh_bcast_compile_typed(const char *expr, int dtype, int nvec, int real_mask, int nscal, int wide_mask, jh_bcast **out);
int jh_bcast_check_typed(const char *expr, int dtype, int nvec, int real_mask, int nscal, int wide_mask);   /* compiles only, like jh_bcast_check */
int jh_bcast_apply(const jh_bcast *bc, jh_bvec *dst, const jh_bvec *const *x, const double *scal_re_im);
/* `count` broadcasts in one call (a tall nonlinear operator evaluates one per child: F(m) and point! are `count` launches
 * enqueued back to back instead of `count` trips through the host language).  Operand k's vectors and scalars follow
 * each other in the flattened lists `x` (sum of nvec entries) and `scal_re_im` (2 * sum of nscal doubles).  When every item holds
 * at least one 16-byte pack and no operand overlaps another item's destination (the items then have no order among them), items that
 * share a program and a length run as ONE launch (same bits); otherwise the items are launched in order. */
int jh_bcast_apply_many(int count, const jh_bcast *const *progs, jh_bvec *const *dsts, const jh_bvec *const *x, const double *scal_re_im);
int jh_bcast_destroy(jh_bcast *bc);
/* dot(x,y), src/Jets.jl:850-856 (conjugates x). fp64 accumulation, deterministic order. */
int jh_dot(const jh_bvec *x, const jh_bvec *y, double *re, double *im);
/* norm(x,p), src/Jets.jl:834-848: p = 2, 1, 0, +Inf, -Inf or any other real.  fp64 accumulation over the whole vector, result in
 * double.  Deviation from the reference, kept on purpose: the reference forms norm(block, p)^p in the ELEMENT precision (843-846), so a
 * BlockArray whose block norms pass sqrt(floatmax) (1.8e19 in Float32) gives Inf there and block norms below sqrt(floatmin) give 0; here
 * the powers are summed in fp64 and, when even that sum leaves the double range, once more on x / 2^k -- the answer is the true norm.
 * (A plain array, one block, behaves like the stdlib's norm, which rescales too.)  A vector of zeros costs one pass. */
int jh_norm(const jh_bvec *x, double p, double *out);
/* extrema(x), src/Jets.jl:870-878 (real dtypes) */
int jh_extrema(const jh_bvec *x, double *mn, double *mx);
/* The block-wise norms and inner products themselves, ALL blocks in one pass (round 6): out[i] = norm(x_i, p) / dot(x_i, y_i) for i = 0 .. nblocks - 1 --
 * what src/Jets.jl:836-846 / 850-856 compute block by block before they combine them, and what per-shot residual norms `[norm(getblock(r, i)) for i in
 * 1:nblocks(r)]` ask for (through jh_norm on a view: one launch and one host round trip per block -- 34 ms for 1024 blocks of 64 MiB where one pass takes
 * 10).  Same accumulation (fp64 lanes, fixed order: deterministic) and the same p rules as jh_norm; conj on jh_dot_blocks' first argument, im may be NULL.
 * Synchronise; out / re / im are HOST arrays of nblocks doubles. */
int jh_norm_blocks(const jh_bvec *x, double p, double *out);
int jh_dot_blocks(const jh_bvec *x, const jh_bvec *y, double *re, double *im);

/* child mul! of a dense operator (test/runtests.jl:27-33 JopBaz): y = A x (adjoint = 0) or y = A' x (adjoint = 1)
 * for a column-major nr x nc matrix in device memory.  Forward: columns accumulated in order, product rounded
 * then added (bit-identical to the sequential loop); adjoint: fp64 wave reduction (tolerance parity). */
int jh_gemv(const void *A_device, int64_t nr, int64_t nc, int dtype, jh_bvec *y, const jh_bvec *x, int adjoint);

/* ---------------------------------------------------------------- block operators ----------- */
/* JetBlock(ops) for device-native blocks, src/Jets.jl:926-930. blocks is column-major nrow x ncol
 * (a Julia Matrix{Jop}); row_len[i] = length(range(ops[i,1])), col_len[j] = length(domain(ops[1,j])). */
int jh_blockop_create(int64_t nrow, int64_t ncol, const jh_block_desc *blocks, const int64_t *row_len,
                      const int64_t *col_len, int dtype, jh_blockop **out);
int jh_blockop_destroy(jh_blockop *op);
/* mul!(d, A, m) -> JetBlock_df!, src/Jets.jl:1010-1032, one fused launch.  Reference quirks kept:
 * zero blocks are skipped (1022); with ncol > 1 the result is accumulated into d without zeroing (1024).
 * With >= 256 block columns of blocks too small to fill the chip, the sum over the columns is cut into parts and folded
 * (deterministic, tolerance parity; jh_tune_set("adj_split", 0) keeps the ordered sum) -- see jh_blockop_mul_adj.
 * Tall, wide and M x K operators made of un-adjointed DENSE blocks of one shape (tall ones may differ in row counts) run
 * batched GEMV kernels: every child in one launch. */
int jh_blockop_mul(const jh_blockop *op, jh_bvec *d, const jh_bvec *m);
/* BLOCK LENGTHS NEED NOT BE MULTIPLES OF 16 BYTES (round 5).  A block vector is one contiguous slab (src/Jets.jl:742-748), so with blocks of 101^3 Float32
 * elements most blocks start off a 16-byte boundary and end inside a 16-byte pack.  The fused kernels address their packs under-aligned and treat a
 * row's last, partial pack separately: jh_blockop_mul / _mul_adj / _normal_mul / _f (tall, wide and M x K operators of elementwise blocks),
 * jh_blockop_mul_axpby / _mul_scaled, jh_blockop_bidiag_step and the one-shard solver loops built on them (jh_lsqr_solve, jh_cgls_solve, jh_cgnr_solve),
 * the fused sums (jh_blocksum_*), the block-vector primitives and jh_bcast_apply[_many] -- same bits as on aligned blocks' kernels (the same terms in the
 * same order) -- and the *_range calls with the partitioned / team solvers built on them: the ranges are cut in the DOMAIN at 16-byte bounds as before, the
 * last one may end with the vector (inside a pack).  jh_blockop_mul_adj_axpby / _mul_adj_scaled and the graph-replayed small-operator loops likewise.
 * jh_tune_set("tall_unaligned", 0) restores the 4-byte-per-lane kernels of rounds 1-4 for every such operator. */
/* mul!(m, A', d) -> JetBlock_df'!, src/Jets.jl:1034-1057: m zeroed when nrow > 1 (1042), rows summed
 * in order i = 0..nrow-1 with the product rounded before the add (1049) => bit-exact on one GPU.
 * Exception, automatic from 256 rows on when a block is too small to fill the chip with one thread per 16 bytes of the
 * domain (fewer workgroups than CUs): the rows are cut into contiguous parts, each summed in order, and the parts are
 * folded in a fixed order with fp64 accumulation -- deterministic, at least as accurate as the ordered sum, but not
 * its bits (100x faster on 1 GiB of 4096-element rows).  jh_tune_set("adj_split", 0) keeps the ordered walk always.
 * The same applies to jh_blockop_normal_mul, jh_blockop_mul_adj_range and the w of jh_blockop_bidiag_step. */
int jh_blockop_mul_adj(const jh_blockop *op, jh_bvec *m, const jh_bvec *d);
/* mul!(d, F, m) for a NONLINEAR block operator -> JetBlock_f!, src/Jets.jl:988-1008, one fused launch.  SQUARE children
 * square their input; linear children run df! (src/Jets.jl:391-392).  Unlike the linear loop NO block is skipped: with
 * ncol > 1 every child's output (a zero block's zeros included) is added into d as found (1001); with ncol == 1 every
 * child overwrites its row (1003). */
int jh_blockop_f(const jh_blockop *op, jh_bvec *d, const jh_bvec *m);
/* point!(jet(F), mo) for a block jet, src/Jets.jl:1059-1066: child (i,j) is linearised about block j of mo (the whole of
 * mo for a one-column operator).  mo is BORROWED, like jet.m0 in the reference (src/Jets.jl:297-301): it must stay alive
 * and unchanged while jh_blockop_mul / jh_blockop_mul_adj are used as the Jacobian.  Those two fail with JH_ERR_STATE on an
 * operator that has SQUARE blocks and no point (the reference hits a DimensionMismatch on its empty m0). */
int jh_blockop_point(jh_blockop *op, const jh_bvec *mo);
/* The same adjoint restricted to the elements [first_elem, first_elem+count) of the domain vector (16-byte aligned
 * bounds -- the last range may end with the vector instead --; tall operators of elementwise rows): lets a multi-GPU host pipeline the exchange chunk by chunk -- all-reduce chunk k
 * while the kernel computes chunk k+1.  Results are identical to jh_blockop_mul_adj on those elements. */
int jh_blockop_mul_adj_range(const jh_blockop *op, jh_bvec *m, const jh_bvec *d, int64_t first_elem, int64_t count);
/* (A' o A) m -> JetComposite_df! over (A', A), src/Jets.jl:530-534, fused: A's coefficients are read
 * once and the range-side intermediate is never materialised.  Same rounding sequence as the
 * unfused pair, so the result is bit-identical to jh_blockop_mul followed by jh_blockop_mul_adj.
 * Operators: tall (N x 1, N >= 2) of equal elementwise rows, and -- round 6 -- N x K GRIDS of equal elementwise blocks (diagonals, zero / identity / scalar blocks) with K = 2 .. 4
 * (a multi-parameter operator: N shots x K model parameters; N K n s + 2 K n s bytes where the pair moves 2 N K n s + 2 N n s; knob
 * "grid_normal").  JH_ERR_UNSUPPORTED otherwise: chain the two calls.  Many rows of small blocks: the split-row walk, as jh_blockop_mul_adj. */
int jh_blockop_normal_mul(const jh_blockop *op, jh_bvec *y, const jh_bvec *m);
/* The same fused A'A restricted to the elements [first_elem, first_elem+count) of the domain (16-byte aligned bounds): for a host
 * that pipelines the exchange of y range by range against the kernels -- CG on the normal equations over a row partition
 * (jh_cgnr_solve_partitioned / _team do exactly that).  Identical to jh_blockop_normal_mul on those elements. */
int jh_blockop_normal_mul_range(const jh_blockop *op, jh_bvec *y, const jh_bvec *m, int64_t first_elem, int64_t count);
/* JetSum of tall operators, src/Jets.jl:628-655, fused: d = sum_k sign_k*(scale_k*(A_k m)) and its adjoint
 * m = sum_k sign_k*(A_k'(scale_k d)) for tall all-DIAG operators of identical shape, ANY number of them (four per launch, later
 * launches continuing the left-to-right sum: same sequence; (K + 2*ceil(K/4) - 1) range-sized streams where the unfused chain
 * moves 5K + 1) (the terms A_k or s_k*A_k of
 * `1.0*A1 - 2.0*A2 + ...`, docs/src/index.md), one pass over the range vector, same rounding sequence as the unfused
 * chain (one temporary + one accumulate pass per term).  sign_k is +1 or -1, scale_k = 1 for a bare operator. */
int jh_blocksum_mul(int nterms, const jh_blockop *const *ops, const double *scale, const double *sign, jh_bvec *d, const jh_bvec *m);
int jh_blocksum_mul_adj(int nterms, const jh_blockop *const *ops, const double *scale, const double *sign, jh_bvec *m, const jh_bvec *d);
/* The same with the scalars' TYPES (nterms x JH_SCALAR_*; NULL: the two calls above, every scale_k taken in the element type).  The
 * reference's own example `A = 1.0*A1 - 2.0*A2 + 3.0*A3` (src/Jets.jl:686, 703) on Float32 operators has Float64 scalars: Julia's scalar
 * stage `d .= a * tmp` (1159) / `tmp .= conj(a) * d` (1160) is then the promoted product rounded once.  JH_SCALAR_WIDE on a term makes
 * the launch take the WIDE instantiations of the sum kernels (that stage as Float32(a * Float64(x)) per element; terms without the flag
 * keep T(a) -- for them both formulas round the same exact product): still ONE pass, the bits of the unfused chain.  The flag is ignored
 * for 64-bit elements.  JH_SCALAR_COMPLEX: JH_ERR_UNSUPPORTED (the unfused chain, whose scalar stage is jh_lincomb_typed). */
int jh_blocksum_mul_typed(int nterms, const jh_blockop *const *ops, const double *scale, const int32_t *scale_flags, const double *sign,
                          jh_bvec *d, const jh_bvec *m);
int jh_blocksum_mul_adj_typed(int nterms, const jh_blockop *const *ops, const double *scale, const int32_t *scale_flags, const double *sign,
                              jh_bvec *m, const jh_bvec *d);

/* ---------------------------------------------------------------- fused chains (round 6) --- */
/* JetComposite chains of ANY depth through a tall operator, in ONE pass (src/Jets.jl:522-550: the reference applies a composite stage by
 * stage, right to left, each stage into a fresh zeros(range(op_i)), 524-540).  Around a tall operator A (>= 2 equal rows of any elementwise
 * kinds) every other device-native stage of such a chain is elementwise -- a scalar times (1159-1164), a diagonal on the domain (a model
 * mask, a preconditioner), a diagonal on the range (data weights: `W o A`, `A' o W o A`) -- so the chain is one of
 *     JH_CHAIN_FORWARD   out_i = R(a_i .* P(x))                                W o A o M              x: domain, out: range
 *     JH_CHAIN_ADJOINT   out   = Q(sum_i conj(a_i) .* R(x_i))                  M' o A' o W'            x: range,  out: domain
 *     JH_CHAIN_NORMAL    out   = Q(sum_i conj(a_i) .* R(a_i .* P(x)))          M' o A' o W o A o M     x, out: domain (weighted normal equations)
 * with stage lists P (`pre`: on the domain, applied before A, in the order given), R (`mid`: on the range, after A / before A') and
 * Q (`post`: on the domain, after A').  Every stage rounds like its own mul! would (product formed and rounded in the element type
 * before the next stage reads it; rows summed in order from +0, 1042/1049; a zero block of A leaves its stage's zeros, 1022): the result
 * is bit-identical to the stage-by-stage chain, which moves a range-sized temporary in and out per stage (A' o W o A: 8 N n s bytes
 * against 2 N n s here).  Stages:
 *     JH_STAGE_SCALE  x .= a * x for a REAL scalar; flags JH_SCALAR_WIDE as in jh_blockop_mul_scaled (JH_SCALAR_COMPLEX: JH_ERR_UNSUPPORTED)
 *     JH_STAGE_DIAG   x .= c .* x, with JH_STAGE_CONJ conj.(c) .* x.  coeff: DEVICE pointers -- ONE on the domain side; on the range side
 *                     one PER BLOCK ROW of A (a weight vector in one slab: base + i*n; a block-diagonal operator's children: wherever they
 *                     are; NULL: the row is an identity block).  row_flags (range side, optional): per row bit 0 = this row is conjugated
 *                     (a child that is the adjoint of a diagonal), bit 1 = a zero block on the diagonal (the row becomes zeros, 1022).
 *                     JH_STAGE_ROWSUM (with row_flags): the stage is a block-diagonal BLOCK OPERATOR -- several block columns, so the reference
 *                     ACCUMULATES each row, `_d .+= mul!(dtmp, op, _m)` into zeros (1024; adjoint 1042 / 1049): the row is 0 + c .* x (a product of -0
 *                     becomes +0), where a plain diagonal operator stores c .* x itself.
 * At most 4 stages per list and 2 distinct coefficient arrays per list (a list may name the same array twice -- W' o W reads w once).
 * The handle borrows `op` and the coefficient arrays (both must outlive it) and copies everything else.
 * jh_chain_apply(accumulate): 0 out = chain(x); +1 / -1 out = out +- chain(x) -- JetSum's `broadcast!(sgn, d, d, tmp)` (634/643/652) fused
 * into the chain's last stage, so a term of a sum that is itself a chain (A'oA + ..., W1oA1 - W2oA2) never materialises; +2 / -2 the FIRST
 * term of a sum: out = 0 +- chain(x), the reference's `d .= 0` (631/640/649) without the fill pass (0 - t, not -t: the sign of a zero).
 * Many rows of small blocks (hundreds of rows whose ordered walk would leave the chip idle): the ADJOINT / NORMAL chains sum their rows in parts
 * like jh_blockop_mul_adj (deterministic, tolerance parity; the stages after A' and the accumulation run on the folded sum;
 * jh_tune_set("adj_split", 0) keeps the ordered, bit-exact walk; counter "last_adj_parts").
 * JH_ERR_UNSUPPORTED (take the stage-by-stage chain): operators that are not tall / elementwise / equal rows, one-row operators, arrays not
 * aligned like their scalar. */
typedef struct jh_chain jh_chain;
typedef enum { JH_CHAIN_FORWARD = 0, JH_CHAIN_ADJOINT = 1, JH_CHAIN_NORMAL = 2 } jh_chain_type;
typedef enum { JH_STAGE_SCALE = 1, JH_STAGE_DIAG = 2 } jh_stage_kind;
enum { JH_STAGE_CONJ = 4, JH_STAGE_ROWSUM = 8 };   /* (beside JH_SCALAR_* in jh_chain_stage.flags) */
typedef struct {
    int32_t kind;                        /* jh_stage_kind                                                             */
    int32_t flags;                       /* SCALE: JH_SCALAR_*; DIAG: JH_STAGE_CONJ, JH_STAGE_ROWSUM                  */
    double a;                            /* SCALE: the real scalar                                                    */
    const void *const *coeff;            /* DIAG: host array of device pointers (1 on the domain side, nrow on the range side) */
    const uint8_t *row_flags;            /* DIAG on the range side: optional host array of nrow flag bytes (see above) */
} jh_chain_stage;
int jh_chain_create(const jh_blockop *op, int type, int npre, const jh_chain_stage *pre, int nmid, const jh_chain_stage *mid, int npost,
                    const jh_chain_stage *post, jh_chain **out);
int jh_chain_apply(const jh_chain *chain, jh_bvec *out, const jh_bvec *x, int accumulate);
int jh_chain_destroy(jh_chain *chain);

/* Fused solver updates (the two halves of an LSQR / CGLS iteration; callers: IterativeSolvers-style loops over
 * vec(A), src/Jets.jl:1138-1154).  d = alpha*(A m) + beta*d  /  m = alpha*(A' d) + beta*m  with real alpha, beta,
 * and ||result||^2 (fp64) returned through normsq when it is not NULL (then the call synchronises).
 * Same rounding sequence as mul! into a temporary followed by `y .= alpha*tmp .+ beta*y`; no temporary
 * range vector, no separate axpby or norm pass.  beta == 0 never reads the output (BLAS convention).  The adjoint
 * form also takes in_scale: every d_i is multiplied by it (rounded) before A_i' is applied -- with alpha = 1,
 * beta = 0 this is (a*A)' d = A'(conj(a) d) of the scalar-times-operator chain (src/Jets.jl:1159-1164) in one
 * launch; pass 1.0 otherwise (exact).  Tall all-DIAG operators only (else JH_ERR_UNSUPPORTED). */
int jh_blockop_mul_axpby(const jh_blockop *op, jh_bvec *d, const jh_bvec *m, double alpha, double beta, double *normsq);
int jh_blockop_mul_adj_axpby(const jh_blockop *op, jh_bvec *m, const jh_bvec *d, double alpha, double beta, double in_scale,
                             double *normsq);
/* The scalar-times-operator chain (src/Jets.jl:1159-1164) in one pass each way, for a REAL scalar of any Julia type:
 *   jh_blockop_mul_scaled      d = a * (A m)          (`d .= a * tmp`, 1159)
 *   jh_blockop_mul_adj_scaled  m = A' (conj(a) d)     (`tmp .= conj(a) * d`, 1160; conj(a) == a)
 * a_flags = JH_SCALAR_WIDE for a Float64 scalar against 32-bit elements: the scalar stage is then the promoted product rounded once
 * (Float32(a * Float64(x))), the bits of the unfused chain with jh_lincomb_typed; 0: a is taken in the element type, i.e.
 * jh_blockop_mul_axpby(alpha = a, beta = 0) / jh_blockop_mul_adj_axpby(in_scale = a).  A Complex scalar (JH_SCALAR_COMPLEX) is
 * JH_ERR_UNSUPPORTED: the unfused chain.  Same operators as the two calls above; round 5: jh_blockop_mul_adj_scaled with a scalar of the element
 * type also takes tall operators with rows of SEVERAL kinds (the adjoint scales d_i on the way in: one pass, the chain's bits). */
int jh_blockop_mul_scaled(const jh_blockop *op, jh_bvec *d, const jh_bvec *m, double a, int a_flags);
int jh_blockop_mul_adj_scaled(const jh_blockop *op, jh_bvec *m, const jh_bvec *d, double a, int a_flags);
/* One Golub-Kahan / LSQR step over vec(A) (src/Jets.jl:1138-1154) in ONE pass over the operator and the range vector:
 *   u <- alpha*(A v) + beta*u ;   w <- A' u (the new u) ;   *normsq = ||u||^2
 * i.e. jh_blockop_mul_axpby(op, u, v, alpha, beta, normsq) followed by jh_blockop_mul_adj(op, w, u), with u and w
 * bit-identical to that sequence, but every coefficient and every element of u is read once and u written once:
 * (3*N*n + 2*n)*s bytes instead of (5*N*n + 3*n)*s.  The solver then forms v <- w/||u|| - ||u||*v on domain-sized vectors
 * (A' is linear, so the normalisation of u can follow the pass).  Tall all-DIAG operators; w must not alias v. */
int jh_blockop_bidiag_step(const jh_blockop *op, jh_bvec *u, const jh_bvec *v, jh_bvec *w, double alpha, double beta, double *normsq);
/* The same step restricted to the elements [first_elem, first_elem+count) of the domain (16-byte aligned bounds): updates
 * those columns of every row of u, writes that range of w, and returns that range's share of ||u||^2 (the shares add up).
 * Lets a row-partitioned multi-GPU solver all-reduce chunk k of w while chunk k+1 is being computed.  Identical values
 * to jh_blockop_bidiag_step on those elements. */
int jh_blockop_bidiag_step_range(const jh_blockop *op, jh_bvec *u, const jh_bvec *v, jh_bvec *w, double alpha, double beta,
                                 int64_t first_elem, int64_t count, double *normsq);
/* Deferred ||u||^2 for a step enqueued range by range: with normsq == NULL jh_blockop_bidiag_step_range does not synchronise;
 * it ADDS its range's share to a device-side accumulator (stream-ordered, so the additions happen in enqueue order and the
 * sum is deterministic).  jh_normsq_reset zeroes the accumulator (enqueued on the library stream), jh_normsq_read reads it
 * back (synchronises): reset, k ranged steps each followed by the host's all-reduce of that range of w, ONE read-back. */
int jh_normsq_reset(void);
int jh_normsq_read(double *out);

/* LSQR (Paige & Saunders 1982) on min ||A x - b||_2 (+ damp^2 ||x||^2) for a tall all-DIAG operator: the solver loop the
 * reference's users run as `lsqr(vec(A), vec(d))` (IterativeSolvers.jl; src/Jets.jl:1143-1152, docs/src/index.md:235-246),
 * here behind the ABI so that every host language gets the one-pass iteration (jh_blockop_bidiag_step) without
 * re-implementing the recurrences.  `u` holds b on entry and is OVERWRITTEN (it becomes the Lanczos vector; at the headline
 * size it is 64 GiB); `x` holds x0 on entry when use_x0 != 0 and the solution on return.  Stopping rules and istop codes are
 * the paper's (1: ||r|| small, 2: ||A'r|| small, 3: cond(A) > conlim, 4-6: the same at machine precision, 7: maxiter);
 * force_maxiter != 0 keeps iterating (throughput measurements).  history (optional, 2*maxiter doubles) receives
 * (r1norm, arnorm) per iteration.  jh_lsqr_solve is always LOCAL to this process, whether or not a communicator exists.
 * jh_lsqr_solve_partitioned is the row-partitioned solve: `op`/`u` are this rank's block rows, x is replicated, and the
 * exchange (one all-reduce of the domain vector and one of a scalar per iteration) runs over the communicator of
 * jh_comm_init_rank; EVERY rank must call it, in lock-step (it is a collective).  With one rank it equals jh_lsqr_solve. */
typedef struct {
    int32_t istop, itn;
    double r1norm, r2norm, anorm, acond, arnorm, xnorm;
} jh_lsqr_result;
int jh_lsqr_solve(const jh_blockop *op, jh_bvec *u, jh_bvec *x, int use_x0, double damp, double atol, double btol, double conlim,
                  int maxiter, int force_maxiter, jh_lsqr_result *res, double *history);
int jh_lsqr_solve_partitioned(const jh_blockop *op, jh_bvec *u, jh_bvec *x, int use_x0, double damp, double atol, double btol,
                              double conlim, int maxiter, int force_maxiter, jh_lsqr_result *res, double *history);
/* The same solve for a single-process TEAM (jh_comm_init_all, below): member k's context holds ops[k] (its block rows), us[k] (its
 * rows of b; overwritten) and xs[k] (its replica of x; all replicas come out bit-identical).  One call, one host thread: the
 * members' kernels are enqueued one after the other, the ranged all-reduces of a range go out as one group, scalars are added on
 * the host. */
int jh_lsqr_solve_team(int n, const jh_blockop *const *ops, jh_bvec *const *us, jh_bvec *const *xs, int use_x0, double damp, double atol,
                       double btol, double conlim, int maxiter, int force_maxiter, jh_lsqr_result *res, double *history);
/* CGLS (conjugate gradients on the normal equations: Hestenes & Stiefel 1952; Bjorck 1996, algorithm 7.4.1) on the same problem,
 * min ||A x - b||^2 + damp^2 ||x||^2, for a tall (>= 2 rows) all-DIAG operator -- SURVEY section 8 f-1 names it next to LSQR; like
 * LSQR it has no counterpart inside Jets.jl (src/Jets.jl:1143-1152 points at IterativeSolvers.jl).  An iteration is TWO passes and
 * no range-sized temporary: ||A p||^2 = <p, A'A p> through the fused normal operator (N n s bytes), then r <- r - alpha A p, ||r||^2
 * and A'r in ONE pass of the Golub-Kahan step kernel (3 N n s) -- 4 N n s against the 7 N n s of the textbook loop over
 * jh_blockop_mul_axpby / jh_blockop_mul_adj_axpby with its q = A p vector.  `u` holds b on entry and the residual r = b - A x on
 * return; `x` holds x0 when use_x0 != 0.  istop: 1 ||r|| <= btol ||b||; 2 ||A'r - damp^2 x|| <= atol times its starting value;
 * 6 breakdown (<p, (A'A + damp^2) p> not positive); 7 maxiter.  history (optional, 2*maxiter doubles): (||r||, ||A'r - damp^2 x||) per
 * iteration.  The result record is LSQR's (r1norm = ||r||, r2norm = sqrt(||r||^2 + damp^2 ||x||^2), arnorm, xnorm; anorm = acond = 0).
 * _partitioned / _team: the same exchange modes as the LSQR entry points (pass 1 exchanges ONE scalar; pass 2 is the pipelined step). */
int jh_cgls_solve(const jh_blockop *op, jh_bvec *u, jh_bvec *x, int use_x0, double damp, double atol, double btol, int maxiter,
                  int force_maxiter, jh_lsqr_result *res, double *history);
int jh_cgls_solve_partitioned(const jh_blockop *op, jh_bvec *u, jh_bvec *x, int use_x0, double damp, double atol, double btol, int maxiter,
                              int force_maxiter, jh_lsqr_result *res, double *history);
int jh_cgls_solve_team(int n, const jh_blockop *const *ops, jh_bvec *const *us, jh_bvec *const *xs, int use_x0, double damp, double atol,
                       double btol, int maxiter, int force_maxiter, jh_lsqr_result *res, double *history);
/* Conjugate gradients on the normal equations (A'A + damp^2 I) x = A'b THROUGH THE FUSED NORMAL OPERATOR (jh_blockop_normal_mul = the
 * reference's JetComposite (A', A), src/Jets.jl:530-534, in one kernel) -- the solver BASELINE.json's "A' o A normal-equations matvec"
 * config is the matvec of.  After one adjoint pass for A'b an iteration reads the coefficients once (N n s bytes: a third of the LSQR
 * iteration, a quarter of the CGLS one) and works on domain-sized vectors; `b` is read once and NOT written; ||r|| follows the exact CG
 * recurrence ||r_k||^2 = ||r_{k-1}||^2 - alpha_k gamma_{k-1}.  Same iterates as CGLS / LSQR in exact arithmetic; in floating point its
 * residual A'r is updated by recurrence in the domain, so the attainable accuracy goes with cond(A)^2 (CGLS / LSQR: cond(A)) -- meant
 * for well-conditioned operators and for throughput.  istop / history / result record as jh_cgls_solve (r2norm = the recurrence's
 * sqrt(||r||^2 + damp^2 ||x||^2)).  _partitioned / _team: the domain vector A_k'A_k p is the only thing
 * exchanged, in 4 element ranges (jh_blockop_normal_mul_range), each all-reduced under the next range's kernel.
 * Round 6: jh_cgnr_solve also takes an N x (2 .. 4) grid of equal diagonals (the fused A'A of jh_blockop_normal_mul for such grids). */
int jh_cgnr_solve(const jh_blockop *op, jh_bvec *b, jh_bvec *x, int use_x0, double damp, double atol, double btol, int maxiter,
                  int force_maxiter, jh_lsqr_result *res, double *history);
int jh_cgnr_solve_partitioned(const jh_blockop *op, jh_bvec *b, jh_bvec *x, int use_x0, double damp, double atol, double btol, int maxiter,
                              int force_maxiter, jh_lsqr_result *res, double *history);
int jh_cgnr_solve_team(int n, const jh_blockop *const *ops, jh_bvec *const *bs, jh_bvec *const *xs, int use_x0, double damp, double atol,
                       double btol, int maxiter, int force_maxiter, jh_lsqr_result *res, double *history);
/* ---------------------------------------------------------------- RCCL over xGMI ----------- */
/* Row partition of a tall operator across the GPUs of a node (one process per GPU): the forward needs no exchange
 * (src/Jets.jl:1015-1031), the adjoint is a sum over rows (1045-1053) -> one in-place all-reduce of the domain vector
 * after the local jh_blockop_mul_adj; range-side dot/norm -> scalar all-reduce of the local partials.  The host
 * language distributes the 128-byte id of rank 0.  Collectives are enqueued on the library stream.
 * A communicator belongs to a CONTEXT (the current one at jh_comm_init_rank; collectives on a vector use its context's). */
int jh_comm_available(void);                                    /* JH_OK when an RCCL can be loaded; no side effects (no bootstrap root, no communicator) */
int jh_comm_unique_id(void *out128);
int jh_comm_init_rank(const void *id128, int nranks, int rank);
int jh_comm_destroy(void);                                      /* the current context's communicator (a team: all members') */
int jh_comm_info(int *nranks, int *rank);                       /* of the current context; (1, 0) without a communicator */
/* ONE process driving several GPUs (ncclCommInitAll, one stream per device, grouped calls): the n contexts
 * become a TEAM, member k = contexts[k] -- on pairwise distinct devices (RCCL), or all on ONE device (several streams of one
 * GPU: the sum is then a device kernel over the members' buffers, members added in rank order; RCCL refuses two ranks on a
 * device).  The host issues jh_comm_allreduce_sum / _sum_range once PER MEMBER between jh_comm_group_begin and
 * jh_comm_group_end (ncclGroupStart / ncclGroupEnd; outside a group a member's call is refused -- a single thread would block
 * in it).  Scalars need no collective in a team: the host reads every member's partial (jh_normsq_read, jh_dot, jh_norm) and
 * combines them (jh_lsqr_solve_team does all of this behind one call); jh_comm_allreduce_scalars / _normsq and
 * jh_lsqr_solve_partitioned are for one-process-per-GPU ranks. */
int jh_comm_init_all(int n, const int *contexts);
int jh_comm_group_begin(void);                                  /* on a member context of the team */
int jh_comm_group_end(void);
int jh_comm_allreduce_sum(jh_bvec *v);
int jh_comm_allreduce_scalars(double *values, int n, int op);   /* op: 0 sum, 1 max, 2 min; synchronises */
/* Pipelined exchange.  jh_comm_allreduce_sum_range sums the elements [first_elem, first_elem+count) of v over all ranks on the
 * communicator's OWN stream, ordered after everything enqueued on the library stream so far and concurrent with what is enqueued
 * afterwards: call it after jh_blockop_mul_adj_range / jh_blockop_bidiag_step_range of a range and the all-reduce of that range
 * runs while the kernel of the next range computes.  jh_comm_join makes the library stream wait for every ranged all-reduce
 * enqueued so far (no host synchronisation).  jh_comm_allreduce_normsq returns the sum over all ranks of the deferred ||u||^2
 * accumulator (jh_normsq_reset + ranged steps with normsq == NULL) behind the ranged all-reduces: the one host synchronisation
 * of a distributed one-pass step.  Every rank enqueues the same ranges in the same order. */
int jh_comm_allreduce_sum_range(jh_bvec *v, int64_t first_elem, int64_t count);
int jh_comm_join(void);
int jh_comm_allreduce_normsq(double *out);
/* A team's operator applications behind ONE call each (round 4): member k's context holds ops[k] (its block rows), its shard of the
 * range vector and its replica of the domain vector.  jh_team_mul: d_k = A_k m_k on every member (src/Jets.jl:1015-1031: block rows
 * are independent -- no exchange).  jh_team_mul_adj: every member's m_k = the sum over ALL members' rows of A_i' d_i (1045-1053):
 * the members' ordered local sums in `nranges` element ranges (jh_blockop_mul_adj_range), the grouped all-reduce of a finished range
 * running under the next range's kernels, the library streams waiting for the exchange by event.  jh_team_normal_mul: the fused
 * A'A (530-534 over (A', A)) exchanged the same way (jh_blockop_normal_mul_range).  All three return after enqueue, without a host
 * synchronisation.  They do what a host language would otherwise spell as about 9 ABI calls per member and pair. */
int jh_team_mul(int n, const jh_blockop *const *ops, jh_bvec *const *ds, const jh_bvec *const *ms);
int jh_team_mul_adj(int n, const jh_blockop *const *ops, jh_bvec *const *ms, const jh_bvec *const *ds, int nranges);
int jh_team_normal_mul(int n, const jh_blockop *const *ops, jh_bvec *const *ys, const jh_bvec *const *ms, int nranges);

/* kernel-shape tuning knobs (bench/tests only): 0 = automatic (fwd_order: -1); name in {"fwd_group","fwd_unroll","fwd_wg","adj_unroll","adj_depth","adj_wg" (round 6: a
 * triple that names a shape the size rule never selects runs the nearest compiled one; same bits),"fwd_order","nt" (nontemporal loads / stores on the streamed operands: 0 never, 1 unless one pass's working set is at most "nt_resident_mib" MiB -- an operator that stays in the 256 MiB Infinity Cache between a solver's iterations is loaded temporal --, 2 always),"nt_resident_mib","autotune" (0: tall forwards keep the size-based default shape; 1: per-operator lazy measurement, see jh_blockop_tune_get),
 * "graphs" (1: operators that run the per-block loop -- those with DENSE blocks -- replay it as a hipGraph from the
 * third call with the same vectors on; 0: always eager), "general_xcd" (the general M x K kernels' grid order: 1 automatic -- XCD-aware when the
 * input vector is >= 32 MiB, line by line below --, 0 never XCD-aware, 2 always), "red_wgs", "bcast_item_fast" (batched broadcasts with a shared operand: -1 automatic, 0 plain kernel, 1 items fastest), "step_chain" (the one-pass step as chained row chunks: -1 measured per operator, 0 never, 1 whenever the shape allows), "step_chunk" (rows per chunk of the chained step: 0 automatic -- 32 rows of a 256-lane tile for all-diagonal operators, 8 rows of a 1024-lane tile otherwise --, 8, 16, 32; same bits), "step_band" (the chained step in column bands of this many tiles: -1 the default, 0 none -- tiles fastest over the whole row; same bits), "adj_split" (split-row walk of the tall adjoint / fused normal /
 * one-pass step: -1 automatic, 0 never -- always the ordered, bit-exact walk --, k > 1 that many row parts)};
 * "lsqr_graph" (jh_lsqr_solve below 1 GiB per pass -- 2 GiB when the blocks are below 16 MiB --: the loop with device-resident recurrences replayed as a hipGraph, 1 yes, 2 at any size, 0 the
 * host loop; same iterates), "grid_diag" (M x K grids of plain diagonals on the branch-free kernel k_grid_diag: 1 yes -- 2 / 4: that many packs per lane, measured no
 * better --, 0 the general kernels), "grid_tile" (those grids register-tiled, k_grid_tile -- a workgroup owns R lines x one element
 * tile, the shared input pack loaded once per R products: 1 automatic R, 2 / 4 / 8 that R, 0: k_grid_diag; same bits),
 * "sum_group" (terms of a fused JetSum forward per launch: 16, 8, or 4 = round 2's grouping; same bits), "bcast_band" (batched broadcasts with a shared operand -- F(m) of a tall nonlinear operator -- in column bands of that many tiles: 0 = 32, 1 = items
 * fastest without bands; same bits), "general_band" (tiles of every line the general M x K kernels walk before the next group of tiles starts: 8, or 16 / 32 / 64 -- measured neutral;
 * same bits), "fwd_ctiles" (tall forward in column bands of that many tiles of one row -- then the same tiles of the next rows, then the next band: -1 the shape's own, 0 none; same bits), "sum_adj_group" (terms of its adjoint per launch,
 * each with its own accumulator: 8, or 16; same bits), "general_tile" (grids of equal elementwise
 * blocks of ANY kinds register-tiled, k_general_tile: 1 automatic -- four lines per workgroup when there are four, else two --, 2 / 4 that many
 * lines, 0 the one-line-per-workgroup general kernels; same bits), "general_list" (late round 5: SPARSE grids of equal elementwise blocks walk lists of
 * their non-zero steps built at create instead of whole block rows / columns -- the zero blocks the reference skips, src/Jets.jl:1022 / 1047, then cost
 * nothing: 1 automatic -- only when the lists leave out an eighth of the steps; four-line lists, per-line lists or the plain walk by measurement, see
 * "gen_walk_fwd" below --, 0 never, 2 / 3 always the four-line / per-line lists; same bits; counter "last_general_list": 0 plain, 1 four-line, 2 per-line),
 * "wide_twin" (1 x K elementwise operators on the tall kernels through their tall twin: 1 automatic -- the adjoint always, the forward
 * from 16 MiB blocks --, 0 never: the general kernels, 2 both always), "tall_unaligned" (round 5: operators whose blocks are not whole, 16-byte
 * aligned packs -- odd block lengths in one slab -- on the 16-byte-per-lane kernels with under-aligned packs: 1 yes, 0 the 4-byte-per-lane kernels
 * of rounds 1-4; same bits), "ua_nt" (loads of the tall kernels on such rows: -1 temporal -- a 128-byte line two neighbouring waves share is then read
 * from HBM once -- in the forward, the forward update and the step, in the adjoint from 32 MiB rows on, nontemporal on aligned rows; 0 / 1 temporal /
 * nontemporal always; same bits), "fwd_anchor" (round 6: the tall forward of such rows on lanes anchored to each row's own 16-byte grid -- aligned stores, and
 * aligned loads of diagonals laid out like the range vector: -1 rows of 64 KiB or more, 0 never, 1 always; same bits), "tall_f" (F(m) of a tall nonlinear operator of elementwise children -- jh_blockop_f -- on the tall tiling: 1 yes, 0 the
 * general kernels; same bits), "dense_list_shared" (round 6: the rows pass of y = B x for DENSE children whose columns are off the 16-byte grid numbers its
 * chunks XCD by XCD and loads temporally, so the 128-byte line two neighbouring rows share is fetched from HBM once: 1 yes, 0 round 5's pass; same bits),
 * "dense_list_rl_min" (log2 of the fewest row lanes per workgroup of that pass, 0: automatic), "red_blocks_wave" (round 6: jh_norm_blocks / jh_dot_blocks of many blocks of at most 16 KiB with a wave per block in one launch: 1 yes, 0 a workgroup per block + the fold; within the reductions' tolerance of each other), "adj_bare_chain" (round 6: jh_blockop_mul_adj and jh_blockop_normal_mul of a tall operator with rows of several kinds, or rows off the 16-byte grid, of up to 4 MiB on the chain kernels with empty stage lists -- packed row records --: 1 yes, 0 the MIXED tall kernel; same bits unless the split walk's part count changes), "adj_thin_mixed" (round 6: the adjoint of a tall operator with rows of several kinds on thin workgroups when fat ones would leave CUs idle -- rows of 1-8 MiB --: 1 yes, 0 round 5's shapes; same bits unless the split walk's part count changes), "grid_normal" (round 6: jh_blockop_normal_mul on N x (2 .. 4) grids of equal elementwise blocks -- diagonals, zero / identity / scalar blocks -- in one pass: 1 yes, 2 grids of plain diagonals only, 0 JH_ERR_UNSUPPORTED as before; same bits), "dense_combine" (round 6: operators whose non-zero blocks are
 * all DENSE children sum the products of a block line from CSR lists in one launch: 1 yes, 0 the general step lists; same bits);
 * round 4: "cg_dev" (jh_cgls_solve / jh_cgnr_solve with the recurrences on the device, graph-replayed unless lsqr_graph = 0: 1 automatic -- CGLS
 * like lsqr_graph, CG through the fused A'A up to 2 GiB of coefficients --, 2 at any size, 0 never: the host loops; within solver tolerance
 * of each other), "dense_fused" (adjoint of many small DENSE children / forward of a 1 x K operator of them in ONE fused launch + fold: 1
 * automatic, 0 the batched kernels of round 3; tolerance parity either way), "dense_gw" (children per group of that launch, 0 automatic),
 * "dense_fwd_wgs" (workgroups the column-split dense forward aims for, 0 = 512), "walk_memory" (1: an operator of a (device, eltype, rows,
 * block size) already measured in this process starts from that forward walk and confirms it; 0: every operator measures all candidates),
 * "cgls_trace" (1: jh_cgls_solve_team records when each member's first pass began and ended; counter "last_cgls_overlaps"), "small_loop"
 * (operators whose DENSE children are all small in one launch: 1 yes, 0 the per-block loop), "force_dist" (1: run the exchange of the partitioned
 * solvers with a one-rank communicator too; validation), "adj_rows_per_launch" (tall adjoint / fused normal: block rows per launch, 0 all
 * rows in one; same bits), "dense_mixed" (operators mixing big DENSE children with other kinds as one batched launch + one launch of the
 * general kernels: 1 yes, 0 the per-block loop; counter "last_launches"); late round 5: "dense_list" (1: those batched launches walk LISTS of the dense
 * children built at create -- no workgroup for a block pair without a dense child, few big children spread over the chip, products in one compact scratch
 * vector, the combine launch over each line's step list: block-diagonal / block-banded operators of dense children at any block count; 0: round 3's grid
 * over every block pair), "dense_list_split" (1: the list kernel of y = B x picks its lane layout -- column groups of one workgroup meet in LDS:
 * deterministic, tolerance parity like any BLAS gemv --, 0: columns in order, the sequential loop's bits; counter "last_dense_rl" = row lanes per workgroup of
 * the latest such launch, 256: columns in order), "dense_grid" (read by jh_blockop_create: M x K grids of uniform dense children on the list route -- 0, the default -- or as one tall batch per
 * block column -- 1, rounds 2-4), "dense_direct" (1: an operator whose every output line holds ONE block, a dense child -- block-diagonal --, in ONE launch: the list kernels write the
 * output vector, same products and additions; 0: scratch vector + combine launch), "dense_list_cpw" (columns per lane group of the list kernel of y = B' x: 0 by column length, 1 / 2 / 4; same contract),
 * "small_loop_max_kib" (operators of SMALL dense children whose matrices together reach this many KiB take
 * the list route instead of the one-launch loop: 512);
 * jh_tune_get also reads the counters "last_fwd_walk" (grid walk of the latest tall forward: 0 sequential, 1 all rows, 2 column bands),
 * "last_fwd_rows_per_wg", "last_adj_launches", "last_adj_parts", "last_step_chain" (row chunks of the latest one-pass step, 0: the plain walk), "graph_replays", "last_lsqr_graph" / "last_cg_graph" (graph replays of the latest
 * jh_lsqr_solve / jh_cgls_solve or jh_cgnr_solve; 0: the host loop ran) and "last_dense_fused" (1: the latest dense adjoint / wide forward took the
 * fused launch). */
int jh_tune_set(const char *name, int64_t value);
int jh_tune_get(const char *name, int64_t *value);
/* Per-operator choices made by measurement.  "fwd_walk": the grid walk of the tall forward of an operator far larger than the
 * caches (which one is fastest depends on where the slabs landed physically).  It is chosen LAZILY: while it is -1 each
 * jh_blockop_mul runs the next candidate between two events -- no extra launches, no host synchronisation, jh_blockop_mul
 * returns after enqueue -- and after 16 calls (20 for operators of fewer than 1024 rows, which also try two column-band walks: candidates 8, 9) the fastest is kept ("fwd_trials" counts the timed calls so far).  A host that
 * wants the steady state at once (or the same choice in every process) reads it from one operator and sets it on another;
 * setting -1 measures again.  "upd_walk" is the same for jh_blockop_mul_axpby (0 / 1, chosen over its first two calls), "step_mode"
 * for jh_blockop_bidiag_step: 0 plain walk, 1 the same with XCD-contiguous tiles, 2 chained row chunks (one batch of 8 rows per
 * workgroup, the ordered sum handed from chunk to chunk: same bits), chosen over its first seven eligible calls ("step_trials" counts
 * them), "gen_walk_fwd" / "gen_walk_adj" for jh_blockop_mul / _mul_adj of a sparse M x K grid that moves >= 64 MiB per call: 0 the four-line
 * step lists, 1 the per-line lists, 2 the plain walk, chosen over the first seven calls of each direction ("gen_trials" counts them; same bits).
 * Read-only: "fwd_walk_inherited" (1: the forward walk came from an earlier operator of the same device, eltype, row count and
 * block size -- knob "walk_memory"), "fwd_switches" (times the periodic re-check rotated another walk in), "fwd_playoff" (the two walks of
 * the final play-off as 16 a + b, -1: none yet). */
int jh_blockop_tune_get(const jh_blockop *op, const char *name, int64_t *value);
int jh_blockop_tune_set(jh_blockop *op, const char *name, int64_t value);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* JETSHIP_H */
