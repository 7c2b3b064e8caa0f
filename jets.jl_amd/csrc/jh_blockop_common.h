// jh_blockop_common.h -- what the translation units of the block-operator family share (round 5: jh_blockop.hip, 4 200 lines and
// about 1 000 kernel instantiations in ONE translation unit, was the 80-second critical path of every build; it is now split by kernel
// family -- jh_tall.hip (forward / adjoint / fused A'A), jh_tall_step.hip (solver updates, the one-pass step), jh_tall_sum.hip (JetSum),
// jh_general.hip (M x K, grids, per-block loops) and jh_blockop.hip (create / destroy / dispatch) -- which `make -j` builds side by side).
//  * device helpers (templates / inline): every unit gets its own copy;
//  * host functions that more than one unit needs: declared here in namespace jhb, defined in the unit named beside them.
#pragma once
#include "jh_internal.h"
#include <type_traits>
#include <tuple>

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

// UNDER-ALIGNED packs (round 5, session 3).  A block vector is ONE contiguous slab (src/Jets.jl:742-748: block i starts where block i - 1 ends), so with
// blocks of 101^3 Float32 elements -- odd grid sizes are the rule in the reference's applications -- only every fourth block starts on a 16-byte boundary
// and a row is not a whole number of packs.  gfx950 executes global_load / global_store_dwordx4 at any dword-aligned address; what must not happen is the
// COMPILER believing in 16 bytes, so these helpers access the pack through a type aligned like its scalar (same instruction, no alignment assumed).
// The row's last, partial pack (n % NS != 0) is LOADED from n - NS -- inside the row, overlapping its neighbour -- and STORED scalar by scalar, only the
// scalars that neighbour does not own (st_pack): every scalar is written once, by one lane, from loads of its own position -- safe in place too.
template <bool NT, typename S, int NS> __device__ inline typename vec_of<S, NS>::type ldu(const S *p)
{
    typedef typename vec_of<S, NS>::type V;
    typedef V __attribute__((aligned(alignof(S)))) UV;
    typedef const UV __attribute__((address_space(1))) *gp;
    if (NT) return __builtin_nontemporal_load((gp)p);
    return *(gp)p;
}
template <bool NT, typename S, int NS> __device__ inline void stu(S *p, typename vec_of<S, NS>::type v)
{
    typedef typename vec_of<S, NS>::type V;
    typedef V __attribute__((aligned(alignof(S)))) UV;
    typedef UV __attribute__((address_space(1))) *gp;
    if (NT) __builtin_nontemporal_store(v, (gp)p);
    else *(gp)p = v;
}
// where the pack that nominally starts at scalar s of a row of n >= NS scalars is loaded from
template <int NS> __device__ inline int64_t pack_start(int64_t s, int64_t n) { return s + NS <= n ? s : n - NS; }
// store the pack loaded from `sc` = pack_start(s, n): all of it, or (the partial pack) the scalars from s on
template <bool NT, typename S, int NS> __device__ inline void st_pack(S *row, int64_t s, int64_t sc, typename vec_of<S, NS>::type v)
{
    if (sc == s) stu<NT, S, NS>(row + s, v);
    else {
#pragma unroll
        for (int e = 0; e < NS; e++)
            if (sc + e >= s) st<NT>(row + sc + e, (S)v[e]);
    }
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

template <typename S, int NS, typename V> __device__ inline double vnorm2(V r)
{
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < NS; e++) acc += (double)r[e] * (double)r[e];
    return acc;
}
// the same over the scalars e >= e0 of the pack: a row's partial last pack (loaded from n - NS, pack_start) counts the scalars it OWNS, e0 = s - pack_start(s, n)
template <typename S, int NS, typename V> __device__ inline double vnorm2_from(V r, int e0)
{
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < NS; e++) acc += e >= e0 ? (double)r[e] * (double)r[e] : 0.0;
    return acc;
}


}  // namespace

struct TallShape { int wg, unroll, aux, order, ctiles = 0; };   // aux = rows per workgroup (forward) / rows in flight (adjoint); ctiles: forward column bands (tiles per band, 0: none)

namespace jhb {
// ---- jh_tall.hip
bool tall_fast_ok(const jh_blockop *op, const void *rng_ptr, const void *dom_ptr);     // tall, all DIAG, equal 16-byte aligned blocks
bool tall_mixed_ok(const jh_blockop *op, const void *rng_ptr, const void *dom_ptr);    // tall, >= 2 equal rows of any elementwise kind
bool tall_unaligned_ok(const jh_blockop *op, const void *rng_ptr, const void *dom_ptr);   // the same without the 16-byte conditions (whole-vector forward / adjoint / A'A)
TallShape pick_adj_shape(int64_t nvec, int64_t nrow, int mode);
int64_t pick_adj_parts(int64_t gx, int64_t nrow);
// ---- jh_tall_chain.hip
int bare_chain(const jh_blockop *op, void *out, const void *in, int mode, bool *took);   // the ADJOINT (mode 0) / NORMAL (1) chain kernel with empty stage lists as jh_blockop_mul_adj / _normal_mul of mixed / off-grid rows up to 4 MiB
// ---- jh_grid_normal.hip
bool grid_normal_ok(const jh_blockop *op, const void *y, const void *m);   // an N x (2 .. 4) grid of equal plain diagonals
int grid_normal(const jh_blockop *op, void *y, const void *m);             // y = A'(A m) in one pass, the two-stage chain's bits
int split_slabs(void *out, size_t bytes, void **slabs);   // the split walk's slabs at the start of the scratch buffer; an error if they would lie over `out`
void lazy_release(jh_blockop::LazyTune &t);
void lazy_reset(jh_blockop::LazyTune &t);
int lazy_next(jh_blockop::LazyTune &t, int ncand, int npass, int warm, float margin, int *choice, int *slot, bool playoff = false, const int *order = nullptr);
bool lazy_begin(jh_blockop::LazyTune &t, int slot, hipStream_t st);
void lazy_end(jh_blockop::LazyTune &t, int slot, hipStream_t st, bool ok);
bool stream_is_capturing(hipStream_t st);
// by element type; vectors as raw device pointers.  tall_adj: mode 0 the adjoint, 1 the fused A'A; mixed: rows of several kinds (the caller has
// checked tall_fast_ok / tall_mixed_ok); [first_elem, end_elem) of the domain, end_elem < 0: the whole vector
int tall_fwd(const jh_blockop *op, void *d, const void *m);
int tall_fwd_mixed(const jh_blockop *op, void *d, const void *m, int fmode = 0);   // fmode: JetBlock_f! of a tall nonlinear operator (every row written, SQUARE children square)
int tall_adj(const jh_blockop *op, void *out, const void *in, int mode, bool mixed, int64_t first_elem = 0, int64_t end_elem = -1);
int fold_parts(int dtype, const void *parts, int64_t part_stride, int64_t nparts, void *out, int64_t s_begin, int64_t s_end);   // (scalars: a complex vector is 2n reals)
int split_adjoint_tmp(const jh_blockop *op, void **tmp);
// ---- jh_general.hip
int general_fwd(const jh_blockop *op, void *d, const void *m, int fmode = 0);
int general_adj(const jh_blockop *op, void *m, const void *d);
int loop_small(const jh_blockop *op, void *out, const void *in, int transposed, int fmode);
int dense_mixed(const jh_blockop *op, void *out, const void *in, bool transposed, bool fmode = false);
int dense_grid_fwd(const jh_blockop *op, void *d, const void *m);
int dense_grid_adj(const jh_blockop *op, void *m, const void *d);
int loop_fwd(const jh_blockop *op, void *d, const void *m, bool fmode = false);
int loop_adj(const jh_blockop *op, void *m, const void *d);
// ---- jh_blockop.hip
int check_vectors(const jh_blockop *op, const jh_bvec *rng, const jh_bvec *dom, const char *who);
}  // namespace jhb
using namespace jhb;
