// jh_tall_sum.hip -- fused JetSum of tall diagonal operators (src/Jets.jl:628-655): k_tall_sum_fwd / k_tall_sum_adj (nine to sixteen
// coefficient streams per launch, straight-line load sections), the few-term kernels, WIDE instantiations for Float64 scalars on 32-bit
// elements, and the entry points jh_blocksum_mul[_adj][_typed].
// One of the translation units jh_blockop.hip was split into in round 5 (jh_blockop_common.h).
#include "jh_blockop_common.h"

namespace {

// ------------------------------------------------------------------ fused JetSum of tall operators ---------
// d_i = sum_k sign_k * (scale_k * (a_{k,i} .* m))      (JetSum_df!, src/Jets.jl:639-646, of terms A_k or s_k*A_k)
// m   = sum_k sign_k * (sum_i conj(a_{k,i}) .* (scale_k * d_i))                 (JetSum_df'!, 648-655)
// for up to JH_SUM_MAX tall all-DIAG operators of identical shape, in ONE pass over the range vector: every
// coefficient slab is read once, d is written (forward) or read (adjoint) once -- the unfused chain moves 5 range-sized
// streams per term.  Rounding sequence == the unfused chain: product, scale (exact when 1), signed add, terms in order;
// in the adjoint each term's rows are summed in order into its own accumulator before the terms are combined.
// Up to four terms run on the KM = 4 instantiations (two packs per lane); five to eight on KM = 8 (one pack per lane, to stay within
// the registers of four waves per SIMD) -- round 3: eight terms used to be two launches, the second re-reading and re-writing d.
constexpr int JH_SUM_MAX = 16;           // coefficient streams per launch (forward since round 4, adjoint: sixteen accumulators, knob sum_adj_group)
constexpr int JH_SUM_ADJ_MAX = 8;
// Round 5: the load section of both kernels is STRAIGHT-LINE code.  The round-4 kernels decided per term and row whether the term exists
// (t < k) and how its row is addressed (a slab's stride or the block table) -- two scalar branches and an s_waitcnt lgkmcnt(0) in front
// of every one of up to sixteen loads -- and held sixteen bases, strides, scales and signs in SGPRs: the sixteen-term forward was out of
// SGPRs (106 of 106) and spilled them into VGPR lanes (138 v_readlane + 108 v_writelane in its ISA, 148 VGPRs = three waves per SIMD).
// Now: STRIDED (every term's diagonals in one slab, all with the same row stride: one 64-bit row offset per row, a base per term) or the
// block TABLES (every term's row pointer is read, all scalar loads issued together) is a template parameter; terms beyond k are filled
// with term 0's addresses on the host -- their loads are L1 hits, their arithmetic is computed and dropped by a wave-uniform select --
// and sign * scale arrives as ONE factor per term.
// Round 5, session 3: rows need not be whole, 16-byte aligned packs (odd block lengths in one slab, jh_tall.hip: tall_unaligned_ok) -- every access of
// these kernels is an under-aligned pack (ldu), a row's last pack is loaded from n - NS and stored from its own first scalar on (st_pack).
struct SumArgs {
    const void *a0[JH_SUM_MAX];          // STRIDED: row 0 of term t's coefficients; else term t's device block table (jh_dev_block *)
    int64_t stride;                      // STRIDED: scalars from one row to the next (the same for every term)
    double coef[JH_SUM_MAX];             // forward: sign_t * scale_t (-(s*x) == (-s)*x exactly); adjoint: scale_t
    double sign[JH_SUM_MAX];             // adjoint: the sign of term t's ordered row sum in the final combination
    float coef32[JH_SUM_MAX], sign32[JH_SUM_MAX];   // the same in Float32, for 32-bit elements: read straight into SGPRs (a double converted in the
                                                    // kernel lands in a VGPR, and the compiler then keeps sixteen splatted packs live: 64 registers)
    int k;
};
template <typename S> __device__ inline S sum_coef(const SumArgs &a, int t) { if constexpr (sizeof(S) == 4) return a.coef32[t]; else return a.coef[t]; }
template <typename S> __device__ inline S sum_sign(const SumArgs &a, int t) { if constexpr (sizeof(S) == 4) return a.sign32[t]; else return a.sign[t]; }

// WIDE (S = float; round 5): some scale_k is Julia's Float64 (JH_SCALAR_WIDE: `1.0*A1 - 2.0*A2 + 3.0*A3` on Float32 operators, the reference's
// own docstring example, src/Jets.jl:686) -- the scalar stage `_d .= a * tmp` (1159) is then the promoted product rounded once,
// Float32(a * Float64(tmp)); the signed add stays a Float32 add (`broadcast!(sgn, d, d, _d)`, 644).  EVERY term of such a launch is
// computed that way with scale_k as a double: for a scalar that is T(a) in Julia the host passes double(T(a)), and
// Float32(double(a32) * Float64(x)) == a32 * x in Float32 arithmetic (the Float64 product of two Float32 values is exact, so both
// round the exact product once) -- one instantiation serves sums that mix wide and narrow scalars, with the chain's bits.
template <typename S, int E, int NS, int U, int BLK, int KM, bool STRIDED, bool WIDE = false, int D = 1>
__global__ __launch_bounds__(BLK) void k_tall_sum_fwd(SumArgs args, int64_t nrow, int rows_per_wg, const S *__restrict__ m,
                                                      S *__restrict__ d, int64_t n_scalars, unsigned ntiles, int accumulate)
{
    // accumulate != 0: continue the left-to-right sum from what d holds (terms 17..32, ... of a long JetSum: same sequence)
    typedef typename vec_of<S, NS>::type V;
    const unsigned tile = blockIdx.x % ntiles, grp = blockIdx.x / ntiles;
    const int64_t s0 = ((int64_t)tile * U * BLK + threadIdx.x) * NS;
    const int64_t i0 = (int64_t)grp * rows_per_wg;
    const int64_t i1 = (i0 + rows_per_wg < nrow) ? i0 + rows_per_wg : nrow;
    bool ok[U];
    int64_t sk[U];
    V mv[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        ok[k] = (s0 + (int64_t)k * BLK * NS) < n_scalars;
        sk[k] = ok[k] ? pack_start<NS>(s0 + (int64_t)k * BLK * NS, n_scalars) : 0;
        mv[k] = ldu<false, S, NS>(m + sk[k]);
    }
    // The term count as a VECTOR register the compiler cannot see through (round 6): `t < args.k` on the scalar unit made it hoist sixteen wave-uniform
    // 64-bit select masks out of the row loop -- 32 SGPRs held for the whole kernel beside sixteen base addresses and sixteen coefficients: the 16-term
    // forward spilled 60-146 SGPRs into VGPR lanes (tools/kernel_resources.py).  A per-lane compare is one VALU instruction per select and holds nothing.
    int kv = args.k;
    // D rows per iteration, their D x KM coefficient packs (x U) in flight (D = 1: one row).  The row loop is kept rolled: left to itself the
    // compiler unrolls the eight-stream shape to 241 VGPRs (one wave per SIMD: 1.6 TB/s).  A row beyond the group's last re-reads that
    // last row (branch-free loads) and is not stored.
#pragma unroll 1
    for (int64_t i = i0; i < i1; i += D) {
        asm volatile("" : "+v"(kv));                                             // (inside the loop: a loop-invariant compare is hoisted, masks and all)
        V av[D][KM][U], dv[D][U];
#pragma unroll
        for (int j = 0; j < D; j++) {
            const int64_t ij = i + j < i1 ? i + j : i1 - 1;
            if constexpr (STRIDED) {
                const int64_t roff = ij * args.stride;
#pragma unroll
                for (int t = 0; t < KM; t++) {
                    const S *a = (const S *)args.a0[t] + roff;
#pragma unroll
                    for (int k = 0; k < U; k++) av[j][t][k] = ldu<true, S, NS>(a + sk[k]);
                }
            } else {
                const S *ap[KM];
#pragma unroll
                for (int t = 0; t < KM; t++) ap[t] = (const S *)((const jh_dev_block *)args.a0[t])[ij].coeff;     // KM scalar loads, one wait
#pragma unroll
                for (int t = 0; t < KM; t++)
#pragma unroll
                    for (int k = 0; k < U; k++) av[j][t][k] = ldu<true, S, NS>(ap[t] + sk[k]);
            }
#pragma unroll
            for (int k = 0; k < U; k++)
                dv[j][k] = accumulate ? ldu<true, S, NS>(d + ij * n_scalars + sk[k]) : (V)(S)0;   // d .= 0  (639-640)
        }
#pragma unroll
        for (int j = 0; j < D; j++)
#pragma unroll
            for (int k = 0; k < U; k++) {
                V acc = dv[j][k];
#pragma unroll
                for (int t = 0; t < KM; t++) {
                    const V prod = vmul<S, E, NS, V>(av[j][t][k], mv[k], false);         // mul!(_d, A_t, m)
                    V term;
                    if constexpr (WIDE) {
#pragma unroll
                        for (int e = 0; e < NS; e++) term[e] = (S)(args.coef[t] * (double)prod[e]);
                    } else {
                        const S cf = sum_coef<S>(args, t);                           // (s_t * .) then the sign: -(s*x) == (-s)*x exactly.  Scalar by scalar: a packed
#pragma unroll                                                                               //  multiply wants the factor as a (c, c) PAIR of SGPRs -- sixteen pairs held across the row loop
                        for (int e = 0; e < NS; e++) term[e] = cf * prod[e];             //  were what still spilled after the masks had gone (round 6)
                    }
                    const V sum = acc + term;                                        // broadcast!(sgn, d, d, _d)
                    acc = (t < kv) ? sum : acc;                                      // (a term beyond k: dropped)
                }
                if (ok[k] && i + j < i1) st_pack<true, S, NS>(d + (i + j) * n_scalars, s0 + (int64_t)k * BLK * NS, sk[k], acc);
            }
    }
}

// WIDE: as in the forward -- the adjoint's scalar stage `tmp .= conj(a) * d` (1160) is Float32(a * Float64(d_i)) per element
template <typename S, int E, int NS, int U, int DEPTH, int BLK, int KM, bool STRIDED, bool WIDE = false>
__global__ __launch_bounds__(BLK) void k_tall_sum_adj(SumArgs args, int64_t nrow, S *__restrict__ out, const S *__restrict__ in,
                                                      int64_t n_scalars, int accumulate)
{
    typedef typename vec_of<S, NS>::type V;
    const int64_t s0 = ((int64_t)blockIdx.x * U * BLK + threadIdx.x) * NS;
    bool ok[U];
    int64_t sk[U];
    V acc[KM][U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        ok[k] = (s0 + (int64_t)k * BLK * NS) < n_scalars;
        sk[k] = ok[k] ? pack_start<NS>(s0 + (int64_t)k * BLK * NS, n_scalars) : 0;
#pragma unroll
        for (int t = 0; t < KM; t++) acc[t][k] = (V)(S)0;
    }
    // nrow == 1: mul!(_m, op', _d) writes directly (1051) -- the product itself, not 0 + product, which differs for a product of -0.
    // Starting the accumulators at -0 instead gives exactly that ((-0) + p == p for every p, signed zeros included) without a select per add.
    if (nrow == 1) {
#pragma unroll
        for (int k = 0; k < U; k++)
#pragma unroll
            for (int t = 0; t < KM; t++) acc[t][k] = (V)(S)(-0.0);
    }
    auto batch = [&](int64_t i, auto depth_tag) {                             // rows [i, i + D): all loads, then the arithmetic, rows in order
        constexpr int D = decltype(depth_tag)::value;
        V dv[D][U], av[D][KM][U];
#pragma unroll
        for (int j = 0; j < D; j++) {
#pragma unroll
            for (int k = 0; k < U; k++) dv[j][k] = ldu<true, S, NS>(in + (i + j) * n_scalars + sk[k]);
            if constexpr (STRIDED) {
                const int64_t roff = (i + j) * args.stride;
#pragma unroll
                for (int t = 0; t < KM; t++) {
                    const S *a = (const S *)args.a0[t] + roff;
#pragma unroll
                    for (int k = 0; k < U; k++) av[j][t][k] = ldu<true, S, NS>(a + sk[k]);
                }
            } else {
                const S *ap[KM];
#pragma unroll
                for (int t = 0; t < KM; t++) ap[t] = (const S *)((const jh_dev_block *)args.a0[t])[i + j].coeff;   // KM scalar loads, one wait
#pragma unroll
                for (int t = 0; t < KM; t++)
#pragma unroll
                    for (int k = 0; k < U; k++) av[j][t][k] = ldu<true, S, NS>(ap[t] + sk[k]);
            }
        }
#pragma unroll
        for (int j = 0; j < D; j++)
#pragma unroll
            for (int t = 0; t < KM; t++)
#pragma unroll
                for (int k = 0; k < U; k++) {
                    V sd;
                    if constexpr (WIDE) {
#pragma unroll
                        for (int e = 0; e < NS; e++) sd[e] = (S)(args.coef[t] * (double)dv[j][k][e]);
                    } else {
                        sd = sum_coef<S>(args, t) * dv[j][k];
                    }
                    acc[t][k] = acc[t][k] + vmul<S, E, NS, V>(av[j][t][k], sd, true);                      // conj(a_i) .* (s_t * d_i), rows in order
                }
    };
    int64_t i = 0;
    for (; i + DEPTH <= nrow; i += DEPTH) batch(i, std::integral_constant<int, DEPTH>{});
    for (; i < nrow; i++) batch(i, std::integral_constant<int, 1>{});
    int kv = args.k;                                                     // (a vector register: see k_tall_sum_fwd)
    asm volatile("" : "+v"(kv));
#pragma unroll
    for (int k = 0; k < U; k++) {
        V r = accumulate ? ldu<false, S, NS>(out + sk[k]) : (V)(S)0;   // m .= 0  (648-649), or the sum so far
#pragma unroll
        for (int t = 0; t < KM; t++) {
            const V sum = r + sum_sign<S>(args, t) * acc[t][k];                  // broadcast!(sgn, m, m, _m)
            r = (t < kv) ? sum : r;
        }
        if (ok[k]) st_pack<false, S, NS>(out, s0 + (int64_t)k * BLK * NS, sk[k], r);
    }
}

// ---- sums of up to EIGHT terms keep round 4's kernels: their loads sit behind `t < k` (a sum of three terms on the four-stream shape loads three
// streams, not four), which is what such short sums want -- same box, 3 terms: forward 5.9-6.1 against 5.5-5.6 TB/s for the padded straight-line
// form below, adjoint 5.9-6.6 against 5.1; 8 terms within noise (profiles/ab_r05_jetsum.txt).  With few streams there is no SGPR pressure
// either.  Addressing and coefficients as in the straight-line kernels (STRIDED is a template parameter, sign * scale one factor).
template <typename S, int E, int NS, int U, int BLK, int KM, int D, bool STRIDED, bool WIDE = false>
__global__ __launch_bounds__(BLK) void k_tall_sum_fwd_few(SumArgs args, int64_t nrow, int rows_per_wg, const S *__restrict__ m,
                                                      S *__restrict__ d, int64_t n_scalars, unsigned ntiles, int accumulate)
{
    // accumulate != 0: continue the left-to-right sum from what d holds (terms 5..8, 9..12, ... of a long JetSum: same sequence)
    typedef typename vec_of<S, NS>::type V;
    const unsigned tile = blockIdx.x % ntiles, grp = blockIdx.x / ntiles;
    const int64_t s0 = ((int64_t)tile * U * BLK + threadIdx.x) * NS;
    const int64_t i0 = (int64_t)grp * rows_per_wg;
    const int64_t i1 = (i0 + rows_per_wg < nrow) ? i0 + rows_per_wg : nrow;
    bool ok[U];
    int64_t sk[U];
    V mv[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        ok[k] = (s0 + (int64_t)k * BLK * NS) < n_scalars;
        sk[k] = ok[k] ? pack_start<NS>(s0 + (int64_t)k * BLK * NS, n_scalars) : 0;
        mv[k] = ldu<false, S, NS>(m + sk[k]);
    }
    // D rows' loads in flight.  The row loop is kept rolled: left to itself the compiler unrolls the eight-stream shape to 241 VGPRs
    // (one wave per SIMD: 1.6 TB/s)
#pragma unroll 1
    for (int64_t i = i0; i < i1; i += D) {
        V av[D][KM][U], dv[D][U];
#pragma unroll
        for (int j = 0; j < D; j++) {
            const int64_t ij = i + j < i1 ? i + j : i1 - 1;                    // clamped: branch-free loads of valid memory
#pragma unroll
            for (int t = 0; t < KM; t++)
                if (t < args.k) {
                    const S *a = STRIDED ? (const S *)args.a0[t] + ij * args.stride : (const S *)((const jh_dev_block *)args.a0[t])[ij].coeff;
#pragma unroll
                    for (int k = 0; k < U; k++) av[j][t][k] = ldu<true, S, NS>(a + sk[k]);
                }
#pragma unroll
            for (int k = 0; k < U; k++)
                dv[j][k] = accumulate ? ldu<true, S, NS>(d + ij * n_scalars + sk[k]) : (V)(S)0;   // d .= 0  (639-640)
        }
#pragma unroll
        for (int j = 0; j < D; j++)
#pragma unroll
            for (int k = 0; k < U; k++) {
                V acc = dv[j][k];
#pragma unroll
                for (int t = 0; t < KM; t++)
                    if (t < args.k) {
                        V prod = vmul<S, E, NS, V>(av[j][t][k], mv[k], false);       // mul!(_d, A_t, m)
                        V term;
                        if constexpr (WIDE) {
                            const double sd = args.coef[t];                           // sign * scale: -(Float32(s*x)) == Float32((-s)*x) exactly
#pragma unroll
                            for (int e = 0; e < NS; e++) term[e] = (S)(sd * (double)prod[e]);
                        } else {
                            term = sum_coef<S>(args, t) * prod;                       // (s_t * .) then the sign: -(s*x) == (-s)*x exactly
                        }
                        acc = acc + term;                                            // broadcast!(sgn, d, d, _d)
                    }
                if (ok[k] && i + j < i1) st_pack<true, S, NS>(d + (i + j) * n_scalars, s0 + (int64_t)k * BLK * NS, sk[k], acc);
            }
    }
}

// WIDE: as in the forward -- the adjoint's scalar stage `tmp .= conj(a) * d` (1160) is Float32(a * Float64(d_i)) per element
template <typename S, int E, int NS, int U, int DEPTH, int BLK, int KM, bool STRIDED, bool WIDE = false>
__global__ __launch_bounds__(BLK) void k_tall_sum_adj_few(SumArgs args, int64_t nrow, S *__restrict__ out, const S *__restrict__ in,
                                                      int64_t n_scalars, int accumulate)
{
    typedef typename vec_of<S, NS>::type V;
    const int64_t s0 = ((int64_t)blockIdx.x * U * BLK + threadIdx.x) * NS;
    bool ok[U];
    int64_t sk[U];
    V acc[KM][U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        ok[k] = (s0 + (int64_t)k * BLK * NS) < n_scalars;
        sk[k] = ok[k] ? pack_start<NS>(s0 + (int64_t)k * BLK * NS, n_scalars) : 0;
#pragma unroll
        for (int t = 0; t < KM; t++) acc[t][k] = (V)(S)0;
    }
    const bool direct = (nrow == 1);                                            // mul!(_m, op', _d) writes directly (1051)
    for (int64_t i = 0; i < nrow; i += DEPTH) {
        V dv[DEPTH][U], av[DEPTH][KM][U];
#pragma unroll
        for (int j = 0; j < DEPTH; j++)
            if (i + j < nrow) {
#pragma unroll
                for (int k = 0; k < U; k++) dv[j][k] = ldu<true, S, NS>(in + (i + j) * n_scalars + sk[k]);
#pragma unroll
                for (int t = 0; t < KM; t++)
                    if (t < args.k) {
                        const S *a = STRIDED ? (const S *)args.a0[t] + (i + j) * args.stride : (const S *)((const jh_dev_block *)args.a0[t])[i + j].coeff;
#pragma unroll
                        for (int k = 0; k < U; k++) av[j][t][k] = ldu<true, S, NS>(a + sk[k]);
                    }
            }
#pragma unroll
        for (int j = 0; j < DEPTH; j++)
            if (i + j < nrow) {
#pragma unroll
                for (int t = 0; t < KM; t++)
                    if (t < args.k) {
#pragma unroll
                        for (int k = 0; k < U; k++) {
                            V sd;
                            if constexpr (WIDE) {
#pragma unroll
                                for (int e = 0; e < NS; e++) sd[e] = (S)(args.coef[t] * (double)dv[j][k][e]);
                            } else {
                                sd = sum_coef<S>(args, t) * dv[j][k];
                            }
                            V p = vmul<S, E, NS, V>(av[j][t][k], sd, true);                                // conj(a_i) .* (s_t * d_i)
                            acc[t][k] = direct ? p : acc[t][k] + p;
                        }
                    }
            }
    }
#pragma unroll
    for (int k = 0; k < U; k++) {
        V r = accumulate ? ldu<false, S, NS>(out + sk[k]) : (V)(S)0;   // m .= 0  (648-649), or the sum so far
#pragma unroll
        for (int t = 0; t < KM; t++)
            if (t < args.k) r = r + sum_sign<S>(args, t) * acc[t][k];            // broadcast!(sgn, m, m, _m)
        if (ok[k]) st_pack<false, S, NS>(out, s0 + (int64_t)k * BLK * NS, sk[k], r);
    }
}



}  // namespace


// *wide (optional): set when a term's scale carries JH_SCALAR_WIDE and the elements are 32-bit -- the launch then takes the WIDE
// instantiation and every scale that is NOT wide goes in as double(Float32(a)) (see k_tall_sum_fwd)
// forward: coef = sign * scale; adjoint: coef = scale, the sign is applied to the term's ordered row sum
static int sum_prepare(int nterms, const jh_blockop *const *ops, const double *scale, const double *sign, const jh_bvec *rng,
                       const jh_bvec *dom, SumArgs &a, bool *strided, bool adjoint, const char *who, const int32_t *flags = nullptr, bool *wide = nullptr)
{
    JH_REQUIRE(ops && scale && sign && rng && dom, "%s: null argument", who);
    JH_REQUIRE(nterms >= 1 && nterms <= JH_SUM_MAX, "%s: %d terms in one group (1..%d)", who, nterms, JH_SUM_MAX);
    a.k = nterms;
    bool all_strided = true;
    for (int t = 0; t < nterms; t++) {
        const jh_blockop *op = ops[t];
        JH_REQUIRE(op, "%s: null operator %d", who, t);
        JH_TRY(check_vectors(op, rng, dom, who));
        if (!tall_fast_ok(op, rng->data, dom->data) && !(op->all_diag && tall_unaligned_ok(op, rng->data, dom->data)))   // (rows off the 16-byte pack grid: under-aligned packs)
            return jh_fail(JH_ERR_UNSUPPORTED, "%s: term %d is not a tall all-DIAG operator with equal blocks", who, t);
        JH_REQUIRE(op->nrow == ops[0]->nrow && op->row_len[0] == ops[0]->row_len[0] && op->dtype == ops[0]->dtype,
                   "%s: term %d has a different shape or element type", who, t);
        JH_REQUIRE(sign[t] == 1.0 || sign[t] == -1.0, "%s: sign %d must be +1 or -1", who, t);
        if (!(op->diag_strided && (op->nrow == 1 || op->diag_stride_elems == ops[0]->diag_stride_elems) && ops[0]->diag_strided)) all_strided = false;
        if (flags) {
            JH_REQUIRE((flags[t] & ~(JH_SCALAR_COMPLEX | JH_SCALAR_WIDE)) == 0, "%s: unknown scale flags %d on term %d", who, flags[t], t);
            if (flags[t] & JH_SCALAR_COMPLEX) return jh_fail(JH_ERR_UNSUPPORTED, "%s: a Complex scale (term %d) takes the unfused chain", who, t);
        }
    }
    const bool narrow = ops[0]->dtype == JH_F32 || ops[0]->dtype == JH_C32;
    bool any_wide = false;
    if (flags && narrow)
        for (int t = 0; t < nterms; t++) any_wide = any_wide || (flags[t] & JH_SCALAR_WIDE);
    a.stride = all_strided ? ops[0]->diag_stride_elems * (jh_dtype_complex(ops[0]->dtype) ? 2 : 1) : 0;
    for (int t = 0; t < JH_SUM_MAX; t++) {
        const int tt = t < nterms ? t : 0;                                   // a slot beyond k repeats term 0's addresses; its arithmetic is dropped
        a.a0[t] = all_strided ? ops[tt]->blocks[0].coeff : (const void *)ops[tt]->dev_blocks;
        double sc = t < nterms ? scale[t] : 0.0;
        if (t < nterms && any_wide && !(flags[t] & JH_SCALAR_WIDE)) sc = (double)(float)sc;   // T(a), exactly representable: same bits either way
        a.sign[t] = t < nterms ? sign[t] : 1.0;
        a.coef[t] = adjoint ? sc : a.sign[t] * sc;
        a.coef32[t] = (float)a.coef[t];
        a.sign32[t] = (float)a.sign[t];
    }
    *strided = all_strided;
    if (wide) *wide = any_wide;
    return JH_OK;
}

template <typename S, int E, int NS>
static int sum_fwd_launch(const SumArgs &a, bool strided, const jh_blockop *op0, void *d, const void *m, int64_t n_scalars, int accumulate, bool wide = false)
{
    jh_context &c = jh_ctx();
    constexpr int BLK = 256;
    // rows per workgroup, from same-box sweeps (profiles/bench_jetsum_r03.txt): workgroups that move ONE batch and exit stream best --
    // eight coefficient streams: one row of two packs per lane (5.9 TB/s; two rows of one pack 5.3-5.5, four rows 5.0-5.2);
    // up to four streams: two rows
    // round 4: NINE to SIXTEEN streams in one launch (one row of ONE pack per lane: 64 registers of coefficients in flight) -- an 11-term
    // sum as 8 + 3 read and wrote the output twice (4.97 TB/s = 62 % of the roofline over its algorithmic bytes, bench_jetsum_r03.txt)
    const int U = a.k > 8 ? 1 : 2;
    int G = c.fwd_group > 0 ? (int)c.fwd_group : (a.k > 4 ? 1 : 2);                       // rows per workgroup (knob fwd_group: sweeps)
    if (G > op0->nrow) G = (int)op0->nrow;
    const int64_t nvec = (n_scalars + NS - 1) / NS;
    const int64_t gx = (nvec + (int64_t)BLK * U - 1) / ((int64_t)BLK * U);
    int64_t gy = (op0->nrow + G - 1) / G;
    while (gx * gy * BLK >= ((int64_t)1 << 32) && G < op0->nrow) { G *= 2; gy = (op0->nrow + G - 1) / G; }
    JH_REQUIRE(gx * gy * BLK < ((int64_t)1 << 32), "fused sum forward: grid too large");
#define JH_SUM_FWD_D(UU, KM, ST, WD, DD)                                                                                                 \
    hipLaunchKernelGGL((k_tall_sum_fwd<S, E, NS, UU, BLK, KM, ST, WD, DD>), dim3((unsigned)(gx * gy)), dim3(BLK), 0, c.stream, a, op0->nrow, G, \
                       (const S *)m, (S *)d, n_scalars, (unsigned)gx, accumulate)
// (D = 2 rows in flight per workgroup were measured and lose 3-6 % in the forward: profiles/exp_r05_jetsum_rows.txt)
#define JH_SUM_FWD(UU, KM, ST, WD) JH_SUM_FWD_D(UU, KM, ST, WD, 1)
#define JH_SUM_FWD_FEW(UU, KM, ST, WD)                                                                                                   \
    hipLaunchKernelGGL((k_tall_sum_fwd_few<S, E, NS, UU, BLK, KM, 1, ST, WD>), dim3((unsigned)(gx * gy)), dim3(BLK), 0, c.stream, a, op0->nrow, G, \
                       (const S *)m, (S *)d, n_scalars, (unsigned)gx, accumulate)
#define JH_SUM_FWD_K(ST, WD)                                                                                                             \
    do {                                                                                                                                 \
        if (a.k > 12) JH_SUM_FWD(1, 16, ST, WD);                                                                                         \
        else if (a.k > 8) JH_SUM_FWD(1, 12, ST, WD);                                                                                     \
        else if (a.k > 4) JH_SUM_FWD_FEW(2, 8, ST, WD);                                                                                  \
        else JH_SUM_FWD_FEW(2, 4, ST, WD);                                                                                               \
    } while (0)
    bool done = false;
    if constexpr (sizeof(S) == 4) {
        if (wide) {                                                         // a Float64 scale on 32-bit elements: promoted products (k_tall_sum_fwd)
            if (strided) JH_SUM_FWD_K(true, true); else JH_SUM_FWD_K(false, true);
            done = true;
        }
    }
    if (!done) {
        if (strided) JH_SUM_FWD_K(true, false); else JH_SUM_FWD_K(false, false);
    }
#undef JH_SUM_FWD_K
#undef JH_SUM_FWD_FEW
#undef JH_SUM_FWD
#undef JH_SUM_FWD_D
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

// many rows of small blocks: every term's row sum through the split walk of the plain adjoint, combined term by term
// (m = sum_k sign_k * scale_k * (A_k' d); tolerance parity)
template <typename S, int E, int NS>
static int sum_adj_split(int nterms, const jh_blockop *const *ops, const double *scale, const double *sign, void *m, const void *d,
                         int64_t n_scalars, void *tmp)
{
    const int dtype = ops[0]->dtype;
    const int64_t n_elems = n_scalars / E;
    for (int t = 0; t < nterms; t++) {
        JH_TRY(jhb::tall_adj(ops[t], tmp, d, 0, !tall_fast_ok(ops[t], d, tmp)));   // (rows off the pack grid: the MIXED instantiations)
        const double cre[2] = {t == 0 ? sign[t] * scale[t] : 1.0, sign[t] * scale[t]}, cim[2] = {0.0, 0.0};
        const void *xs[2] = {t == 0 ? tmp : m, tmp};
        JH_TRY(jh_launch_lincomb_raw(m, dtype, n_elems, t == 0 ? 1 : 2, cre, cim, xs));
    }
    return JH_OK;
}

template <typename S, int E, int NS>
static int sum_adj_launch(const SumArgs &a, bool strided, const jh_blockop *op0, void *m, const void *d, int64_t n_scalars, int accumulate, bool wide = false)
{
    jh_context &c = jh_ctx();
    c.last_adj_parts = 1;
    constexpr int BLK = 256, DEPTH = 2;
    const int U = a.k > 4 ? 1 : 2;
    const int64_t nvec = (n_scalars + NS - 1) / NS;
    const int64_t gx = (nvec + (int64_t)BLK * U - 1) / ((int64_t)BLK * U);
#define JH_SUM_ADJ_K(UU, DD, KM, ST, WD)                                                                                              \
    hipLaunchKernelGGL((k_tall_sum_adj<S, E, NS, UU, DD, BLK, KM, ST, WD>), dim3((unsigned)gx), dim3(BLK), 0, c.stream, a, op0->nrow, (S *)m, \
                       (const S *)d, n_scalars, accumulate)
#define JH_SUM_ADJ_FEW(UU, DD, KM, ST, WD)                                                                                            \
    hipLaunchKernelGGL((k_tall_sum_adj_few<S, E, NS, UU, DD, BLK, KM, ST, WD>), dim3((unsigned)gx), dim3(BLK), 0, c.stream, a, op0->nrow, (S *)m, \
                       (const S *)d, n_scalars, accumulate)
#define JH_SUM_ADJ_S(ST, WD)                                                                                                          \
    do {                                                                                                                              \
        if (a.k > 12) JH_SUM_ADJ_K(1, 1, 16, ST, WD);      /* sixteen accumulators, one row in flight (two: +1 ... +2 %, noise level: exp_r05_jetsum_rows.txt) */ \
        else if (a.k > 8) JH_SUM_ADJ_K(1, 1, 12, ST, WD);                                                                             \
        else if (a.k > 4) JH_SUM_ADJ_FEW(1, DEPTH, 8, ST, WD);                                                                        \
        else JH_SUM_ADJ_FEW(2, DEPTH, 4, ST, WD);                                                                                     \
    } while (0)
    bool done = false;
    if constexpr (sizeof(S) == 4) {
        if (wide) {
            if (strided) JH_SUM_ADJ_S(true, true); else JH_SUM_ADJ_S(false, true);
            done = true;
        }
    }
    if (!done) {
        if (strided) JH_SUM_ADJ_S(true, false); else JH_SUM_ADJ_S(false, false);
    }
#undef JH_SUM_ADJ_S
#undef JH_SUM_ADJ_FEW
#undef JH_SUM_ADJ_K
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

extern "C" {

// Any number of terms: groups of JH_SUM_MAX, every group after the first continuing the left-to-right sum from what the output
// holds -- the unfused chain's sequence ((0 +- t1) +- t2) +- ... whatever the grouping.
int jh_blocksum_mul(int nterms, const jh_blockop *const *ops, const double *scale, const double *sign, jh_bvec *d, const jh_bvec *m)
{
    return jh_blocksum_mul_typed(nterms, ops, scale, nullptr, sign, d, m);
}

// scale_flags (nterms x JH_SCALAR_*, or NULL: every scale is taken in the element type): the Julia TYPE of each term's scalar.  A
// Float64 scale against 32-bit elements (JH_SCALAR_WIDE) keeps the sum fused -- the WIDE instantiations of the sum kernels compute that
// term's scalar stage as the promoted product rounded once, the bits of the unfused chain (jh_blockop_mul, jh_lincomb_typed, signed add);
// a Complex scale (JH_SCALAR_COMPLEX) is JH_ERR_UNSUPPORTED: the unfused chain.
int jh_blocksum_mul_typed(int nterms, const jh_blockop *const *ops, const double *scale, const int32_t *scale_flags, const double *sign,
                          jh_bvec *d, const jh_bvec *m)
{
    JH_TRY(jh_enter(d, m));
    JH_REQUIRE(ops && scale && sign && d && m, "jh_blocksum_mul: null argument");
    JH_REQUIRE(nterms >= 1 && nterms <= 4096, "jh_blocksum_mul: %d terms (1..4096 supported)", nterms);
    for (int t = 0; t < nterms; t++) {                                   // validate EVERYTHING before the first launch touches d
        JH_REQUIRE(ops[t], "jh_blocksum_mul: null operator %d", t);
        JH_REQUIRE(ops[t]->ctx == d->ctx, "jh_blocksum_mul: operator %d lives in context %d, the vectors in %d", t, ops[t]->ctx, d->ctx);
        SumArgs probe;
        bool st1 = false;
        JH_TRY(sum_prepare(1, ops + t, scale + t, sign + t, d, m, probe, &st1, false, "jh_blocksum_mul", scale_flags ? scale_flags + t : nullptr));
        JH_REQUIRE(ops[t]->nrow == ops[0]->nrow && ops[t]->row_len[0] == ops[0]->row_len[0] && ops[t]->dtype == ops[0]->dtype,
                   "jh_blocksum_mul: term %d has a different shape or element type", t);
    }
    const int64_t n = ops[0]->row_len[0];
    const int group = jh_ctx().sum_group == 4 ? 4 : (jh_ctx().sum_group == 8 ? 8 : JH_SUM_MAX);   // knob sum_group: 4 / 8 = round 2's / round 3's terms per launch (A/B), 16 (default)
    for (int t0 = 0; t0 < nterms; t0 += group) {
        const int k = nterms - t0 < group ? nterms - t0 : group;
        SumArgs a;
        bool wide = false, strided = false;
        JH_TRY(sum_prepare(k, ops + t0, scale + t0, sign + t0, d, m, a, &strided, false, "jh_blocksum_mul", scale_flags ? scale_flags + t0 : nullptr, &wide));
        const int acc = t0 > 0 ? 1 : 0;
        int st = JH_OK;
        switch (ops[0]->dtype) {
        case JH_F32: st = sum_fwd_launch<float, 1, 4>(a, strided, ops[0], d->data, m->data, n, acc, wide); break;
        case JH_F64: st = sum_fwd_launch<double, 1, 2>(a, strided, ops[0], d->data, m->data, n, acc); break;
        case JH_C32: st = sum_fwd_launch<float, 2, 4>(a, strided, ops[0], d->data, m->data, 2 * n, acc, wide); break;
        case JH_C64: st = sum_fwd_launch<double, 2, 2>(a, strided, ops[0], d->data, m->data, 2 * n, acc); break;
        default: return jh_fail(JH_ERR_INVALID, "jh_blocksum_mul: unknown dtype");
        }
        JH_TRY(st);
    }
    return JH_OK;
}

int jh_blocksum_mul_adj(int nterms, const jh_blockop *const *ops, const double *scale, const double *sign, jh_bvec *m, const jh_bvec *d)
{
    return jh_blocksum_mul_adj_typed(nterms, ops, scale, nullptr, sign, m, d);
}

int jh_blocksum_mul_adj_typed(int nterms, const jh_blockop *const *ops, const double *scale, const int32_t *scale_flags, const double *sign,
                              jh_bvec *m, const jh_bvec *d)
{
    JH_TRY(jh_enter(m, d));
    JH_REQUIRE(ops && scale && sign && d && m, "jh_blocksum_mul_adj: null argument");
    JH_REQUIRE(nterms >= 1 && nterms <= 4096, "jh_blocksum_mul_adj: %d terms (1..4096 supported)", nterms);
    for (int t = 0; t < nterms; t++) {
        JH_REQUIRE(ops[t], "jh_blocksum_mul_adj: null operator %d", t);
        JH_REQUIRE(ops[t]->ctx == d->ctx, "jh_blocksum_mul_adj: operator %d lives in context %d, the vectors in %d", t, ops[t]->ctx, d->ctx);
        SumArgs probe;
        bool st1 = false;
        JH_TRY(sum_prepare(1, ops + t, scale + t, sign + t, d, m, probe, &st1, true, "jh_blocksum_mul_adj", scale_flags ? scale_flags + t : nullptr));
        JH_REQUIRE(ops[t]->nrow == ops[0]->nrow && ops[t]->row_len[0] == ops[0]->row_len[0] && ops[t]->dtype == ops[0]->dtype,
                   "jh_blocksum_mul_adj: term %d has a different shape or element type", t);
    }
    const int64_t n = ops[0]->row_len[0];
    const int group = jh_ctx().sum_group == 4 ? 4 : (jh_ctx().sum_adj_group == 16 ? JH_SUM_MAX : JH_SUM_ADJ_MAX);   // (each term keeps its own accumulator in the adjoint: eight per launch; knob sum_adj_group = 16: sixteen)
    void *tmp = nullptr;
    bool any_wide = false;
    if (scale_flags && (ops[0]->dtype == JH_F32 || ops[0]->dtype == JH_C32))
        for (int t = 0; t < nterms; t++) any_wide = any_wide || (scale_flags[t] & JH_SCALAR_WIDE);
    switch (ops[0]->dtype) {
#define JH_SUM_ADJ(S, E, NS, NSCAL)                                                                         \
    if (!any_wide) JH_TRY(jhb::split_adjoint_tmp(ops[0], &tmp));   /* (a wide scale is applied per d_i before the sum: the ordered walk) */ \
    if (tmp) return sum_adj_split<S, E, NS>(nterms, ops, scale, sign, m->data, d->data, NSCAL, tmp);      \
    for (int t0 = 0; t0 < nterms; t0 += group) {                                                            \
        const int k = nterms - t0 < group ? nterms - t0 : group;                                            \
        SumArgs a;                                                                                          \
        bool wide = false, strided = false;                                                                 \
        JH_TRY(sum_prepare(k, ops + t0, scale + t0, sign + t0, d, m, a, &strided, true, "jh_blocksum_mul_adj", scale_flags ? scale_flags + t0 : nullptr, &wide)); \
        JH_TRY((sum_adj_launch<S, E, NS>(a, strided, ops[0], m->data, d->data, NSCAL, t0 > 0 ? 1 : 0, wide)));  \
    }                                                                                                       \
    return JH_OK;
    case JH_F32: JH_SUM_ADJ(float, 1, 4, n)
    case JH_F64: JH_SUM_ADJ(double, 1, 2, n)
    case JH_C32: JH_SUM_ADJ(float, 2, 4, 2 * n)
    case JH_C64: JH_SUM_ADJ(double, 2, 2, 2 * n)
#undef JH_SUM_ADJ
    }
    return jh_fail(JH_ERR_INVALID, "jh_blocksum_mul_adj: unknown dtype");
}


}  // extern "C"
