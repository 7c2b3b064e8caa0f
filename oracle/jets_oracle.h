/*
 * jets_oracle.h -- CPU ORACLE for the block-operator mul! path of ChevronETC/Jets.jl.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (jets.jl_amd/, libjetship.so)
 * never links, loads or calls anything in this directory.
 *
 * It is a plain-C restatement of the reference's algorithm for the hot path, written from
 * the behaviour of /root/reference/src/Jets.jl (pure Julia, v1.4.1).  Every function cites
 * the reference lines it follows.  Loops are sequential, one thread, same operation order
 * as the reference, multiply rounded before add (built with -ffp-contract=off).
 *
 * PARITY PINNING.  The reference stores no golden vectors and cannot be executed in this
 * image (no Julia toolchain).  The oracle is pinned against the reference's own test
 * identities (test/runtests.jl:512-551, 553-600, 602-620, 622-695, 704-758, 789-795,
 * 901-918) re-encoded in tests/test_oracle_*.py with an independent numpy closed form
 * (the role Julia's plain matrices play in those tests), and against the literal-valued
 * checks at test/runtests.jl:518-526.  Since round 2 the hot loops (JetBlock_df!, JetBlock_df'!,
 * the (A', A) composite, JetSum) are ALSO pinned by known answers that do not come from this
 * file: tests/golden/make_known_answers.py derives them from the reference's source lines with
 * exact rational arithmetic and an IEEE round-to-nearest-even of its own (exact-integer and
 * order-revealing cases), and tests/test_known_answers.py requires this oracle to reproduce
 * every one bit for bit.  Intra-block summation order of dot/norm is Julia-stdlib/BLAS defined
 * (not part of the reference source) => tolerance parity only for reductions; everything else is
 * bit-exact by construction.  There is no run of the reference itself behind any of this.
 */
#ifndef JETS_ORACLE_H
#define JETS_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types (same numbering as include/jetship.h) */
enum { JO_F32 = 0, JO_F64 = 1, JO_C32 = 2, JO_C64 = 3 };

/* block kinds (same numbering as include/jetship.h) */
enum { JO_OP_ZERO = 0, JO_OP_IDENTITY = 1, JO_OP_SCALE = 2, JO_OP_DIAG = 3, JO_OP_DENSE = 4, JO_OP_SQUARE = 5 };

typedef struct {
    int32_t kind;       /* JO_OP_*                                                    */
    int32_t adjoint;    /* 1: the block is the JopAdjoint of the described operator   */
    const void *coeff;  /* DIAG: diagonal (len = block len); DENSE: column-major nr x nc; SQUARE: the point mo */
    double sre, sim;    /* SCALE: the scalar a                                        */
    int64_t nr, nc;     /* range / domain length of the described (un-adjointed) op   */
    int32_t sflags;     /* SCALE: JO_SCALAR_* -- what Julia knows from the scalar's TYPE   */
    int32_t reserved;
} jo_block;

/* the TYPE of a scalar (src/Jets.jl:1159 `a * m`, 889-911 broadcast): Julia dispatches on it, a (re, im) pair alone cannot say */
enum { JO_SCALAR_COMPLEX = 1,   /* a Complex scalar: the full complex product even when its imaginary part is zero */
       JO_SCALAR_WIDE = 2 };    /* a Float64-based scalar against 32-bit elements: promoted arithmetic, one rounding on the store */

/* src/Jets.jl:739-750  JetBSpace constructor: cumulative 1-based inclusive ranges */
void jo_bspace_indices(int64_t nblocks, const int64_t *lens, int64_t *start1, int64_t *stop1);
/* src/Jets.jl:820-823  linear index (1-based) -> (block (1-based), local (1-based)) */
int jo_barr_locate(int64_t nblocks, const int64_t *start1, const int64_t *stop1, int64_t i1,
                   int64_t *iblock1, int64_t *ilocal1);

/* counter-based U[0,1) generator shared with the device: hash(j) = mix64(key + (j+1)*GOLDEN) with
 * key = mix64(seed*GOLDEN + stream).  f64 lane k: the top 53 bits of hash(k).  f32 lane k (round 4): 24 bits of hash(k >> 1),
 * the top 24 for an even lane and bits 39..16 for an odd one -- one hash per two values.  complex: re = lane 2k, im = lane 2k+1. */
uint64_t jo_rng_key(uint64_t seed, uint64_t stream);
void jo_rng_u01(int dtype, uint64_t seed, uint64_t stream, int64_t index0, int64_t count, void *out);

/* BlockArray reductions / utilities.  arrays[i] points at block i (len lens[i]). */
/* src/Jets.jl:834-848 */
double jo_barr_norm(int dtype, int64_t nb, const void *const *arrays, const int64_t *lens, double p);
/* src/Jets.jl:850-856 ; result in (re, im) */
void jo_barr_dot(int dtype, int64_t nb, const void *const *x, const void *const *y,
                 const int64_t *lens, double *re, double *im);
/* src/Jets.jl:870-878 (real dtypes) */
void jo_barr_extrema(int dtype, int64_t nb, const void *const *arrays, const int64_t *lens,
                     double *mn, double *mx);
/* src/Jets.jl:880-885 */
void jo_barr_fill(int dtype, int64_t nb, void *const *arrays, const int64_t *lens, double re, double im);
/* src/Jets.jl:862-868 : gather blocks into one flat vector */
void jo_barr_convert(int dtype, int64_t nb, const void *const *arrays, const int64_t *lens, void *flat);
/* src/Jets.jl:905-911 with bc = c1*x1 .+ c2*x2 .+ ... (left-to-right, each op rounded in T) */
void jo_barr_lincomb(int dtype, int64_t nb, void *const *dst, const int64_t *lens, int k,
                     const double *coef_re_im, const int32_t *coef_flags /* k x JO_SCALAR_*, or NULL */, const void *const *const *srcs);

/* child mul! for the device-native block kinds.
 * DIAG  : test/runtests.jl:3-4   d .= diagonal .* m ; m .= conj.(diagonal) .* d
 * SCALE : src/Jets.jl:1159-1160  d .= a*m ; m .= conj(a)*d
 * ZERO  : src/Jets.jl:942        d .= 0
 * DENSE : test/runtests.jl:27-28 d .= A*m ; m .= A'*d      (column-major A, sequential k loop)
 * IDENT : d .= m
 * SQUARE: test/runtests.jl:19-24 (JopBar, NONLINEAR) f!: d .= m.^2 ; Jacobian at mo (coeff): dd .= 2 .* mo .* dm ;
 *         its adjoint conj.(2 .* mo) .* dd (the fixture is Float64 and lets df'! default to df!, src/Jets.jl:184-186) */
void jo_child_mul(int dtype, const jo_block *b, void *d, const void *m);      /* src/Jets.jl:391 */
void jo_child_mul_adj(int dtype, const jo_block *b, void *m, const void *d);  /* src/Jets.jl:392 */

/* src/Jets.jl:1010-1032  JetBlock_df!  (ops column-major nrow x ncol, like a Julia Matrix).
 * d_arrays: nrow range blocks.  m_arrays: ncol domain blocks (ncol==1 => the plain domain array). */
void jo_block_df(int dtype, int64_t nrow, int64_t ncol, const jo_block *ops, void *const *d_arrays,
                 const void *const *m_arrays);
/* src/Jets.jl:988-1008  JetBlock_f!  (nonlinear forward: SQUARE blocks square their input, linear kinds run df!;
 * zero blocks are NOT skipped here) */
void jo_block_f(int dtype, int64_t nrow, int64_t ncol, const jo_block *ops, void *const *d_arrays,
                const void *const *m_arrays);
/* src/Jets.jl:1034-1057  JetBlock_df'! */
void jo_block_df_adj(int dtype, int64_t nrow, int64_t ncol, const jo_block *ops, void *const *m_arrays,
                     const void *const *d_arrays);
/* src/Jets.jl:530-534 applied to (A', A): y = A'(A m) through freshly zeroed temporaries */
void jo_normal_df(int dtype, int64_t nrow, int64_t ncol, const jo_block *ops, void *const *y_arrays,
                  const void *const *m_arrays);

/* All-cores variant of the tall diagonal Float32 pair, for the SEPARATELY LABELLED cpu baseline only:
 * the reference is single-threaded, this is not its structure.  Forward: rows x element chunks in parallel.  Adjoint:
 * element chunks in parallel, rows in order inside a chunk, product rounded then added -- same bits as the sequential
 * loop.  Returns the number of threads used. */
/* fills rows[i][k] = u01(seed, stream 0, element (row0+i)*n + k) with the partition of the two loops below (first touch) */
int jo_fill_u01_omp_f32(int64_t nrow, int64_t n, uint64_t seed, int64_t row0, float *const *rows);
int jo_tall_diag_fwd_omp_f32(int64_t nrow, int64_t n, const float *const *a, const float *m, float *const *d);
int jo_tall_diag_adj_omp_f32(int64_t nrow, int64_t n, const float *const *a, float *m, const float *const *d);

#ifdef __cplusplus
}
#endif
#endif
