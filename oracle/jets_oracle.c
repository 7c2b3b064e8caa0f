/* jets_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see jets_oracle.h for the rules).
 * Plain-C restatement of the block-operator mul! path of /root/reference/src/Jets.jl.
 * Build: see oracle/Makefile (gcc -O2 -ftree-vectorize -ffp-contract=off: no FMA contraction, so a product is
 * rounded before it is added, exactly like the reference's two-pass `.+= mul!(tmp, ...)`). */
#include "jets_oracle.h"
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define T float
#define R float
#define SFX f32
#define CPLX 0
#include "jets_oracle_body.inc"
#undef T
#undef R
#undef SFX
#undef CPLX

#define T double
#define R double
#define SFX f64
#define CPLX 0
#include "jets_oracle_body.inc"
#undef T
#undef R
#undef SFX
#undef CPLX

#define T float _Complex
#define R float
#define SFX c32
#define CPLX 1
#include "jets_oracle_body.inc"
#undef T
#undef R
#undef SFX
#undef CPLX

#define T double _Complex
#define R double
#define SFX c64
#define CPLX 1
#include "jets_oracle_body.inc"
#undef T
#undef R
#undef SFX
#undef CPLX

#define DISPATCH(dtype, call_f32, call_f64, call_c32, call_c64) \
    switch (dtype) {                                             \
    case JO_F32: call_f32; break;                                \
    case JO_F64: call_f64; break;                                \
    case JO_C32: call_c32; break;                                \
    case JO_C64: call_c64; break;                                \
    default: abort();                                            \
    }

/* src/Jets.jl:739-750 */
void jo_bspace_indices(int64_t nblocks, const int64_t *lens, int64_t *start1, int64_t *stop1)
{
    int64_t stop = 0;                                   /* :743 */
    for (int64_t i = 0; i < nblocks; i++) {             /* :744 */
        int64_t start = stop + 1;                       /* :745 */
        stop = start + lens[i] - 1;                     /* :746 */
        start1[i] = start; stop1[i] = stop;             /* :747 */
    }
}

/* src/Jets.jl:820-823: j = findfirst(rng -> i in rng, indices); local = i - indices[j][1] + 1 */
int jo_barr_locate(int64_t nblocks, const int64_t *start1, const int64_t *stop1, int64_t i1,
                   int64_t *iblock1, int64_t *ilocal1)
{
    for (int64_t j = 0; j < nblocks; j++) {
        if (i1 >= start1[j] && i1 <= stop1[j]) { *iblock1 = j + 1; *ilocal1 = i1 - start1[j] + 1; return 0; }
    }
    return -1;   /* findfirst returned nothing: the reference throws a TypeError on `::Int` */
}

#define JO_GOLDEN 0x9E3779B97F4A7C15ULL
static inline uint64_t jo_mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
uint64_t jo_rng_key(uint64_t seed, uint64_t stream) { return jo_mix64(seed * JO_GOLDEN + stream); }
/* Float32 lane k: 24 bits of hash(k >> 1) -- the top 24 for an even lane, bits 39..16 for an odd one (one hash per two values: the
 * device generator is bound by its 64-bit multiplies).  Float64 lane k: the top 53 bits of hash(k). */
static inline float jo_u01_f32(uint64_t key, int64_t lane)
{
    const uint64_t h = jo_mix64(key + (uint64_t)((lane >> 1) + 1) * JO_GOLDEN);
    return (float)((lane & 1) ? ((h >> 16) & 0xFFFFFFull) : (h >> 40)) * 0x1.0p-24f;
}

/* scalar lanes: a complex vector of n elements is 2n scalar lanes (re, im interleaved) */
void jo_rng_u01(int dtype, uint64_t seed, uint64_t stream, int64_t index0, int64_t count, void *out)
{
    uint64_t key = jo_rng_key(seed, stream);
    int is64 = (dtype == JO_F64 || dtype == JO_C64);
    int lanes = (dtype == JO_C32 || dtype == JO_C64) ? 2 : 1;
    int64_t k0 = index0 * lanes, n = count * lanes;
    for (int64_t k = 0; k < n; k++) {
        if (is64) {
            uint64_t h = jo_mix64(key + (uint64_t)(k0 + k + 1) * JO_GOLDEN);
            ((double *)out)[k] = (double)(h >> 11) * 0x1.0p-53;
        } else {
            ((float *)out)[k] = jo_u01_f32(key, k0 + k);
        }
    }
}

double jo_barr_norm(int dtype, int64_t nb, const void *const *arrays, const int64_t *lens, double p)
{
    double r = 0;
    DISPATCH(dtype, r = barr_norm_f32(nb, arrays, lens, p), r = barr_norm_f64(nb, arrays, lens, p),
             r = barr_norm_c32(nb, arrays, lens, p), r = barr_norm_c64(nb, arrays, lens, p));
    return r;
}

void jo_barr_dot(int dtype, int64_t nb, const void *const *x, const void *const *y, const int64_t *lens,
                 double *re, double *im)
{
    DISPATCH(dtype, barr_dot_f32(nb, x, y, lens, re, im), barr_dot_f64(nb, x, y, lens, re, im),
             barr_dot_c32(nb, x, y, lens, re, im), barr_dot_c64(nb, x, y, lens, re, im));
}

void jo_barr_extrema(int dtype, int64_t nb, const void *const *arrays, const int64_t *lens, double *mn, double *mx)
{
    switch (dtype) {
    case JO_F32: barr_extrema_f32(nb, arrays, lens, mn, mx); break;
    case JO_F64: barr_extrema_f64(nb, arrays, lens, mn, mx); break;
    default: abort();   /* extrema of complex values is undefined in the reference too (isless) */
    }
}

void jo_barr_fill(int dtype, int64_t nb, void *const *arrays, const int64_t *lens, double re, double im)
{
    DISPATCH(dtype, barr_fill_f32(nb, arrays, lens, re, im), barr_fill_f64(nb, arrays, lens, re, im),
             barr_fill_c32(nb, arrays, lens, re, im), barr_fill_c64(nb, arrays, lens, re, im));
}

void jo_barr_convert(int dtype, int64_t nb, const void *const *arrays, const int64_t *lens, void *flat)
{
    DISPATCH(dtype, barr_convert_f32(nb, arrays, lens, flat), barr_convert_f64(nb, arrays, lens, flat),
             barr_convert_c32(nb, arrays, lens, flat), barr_convert_c64(nb, arrays, lens, flat));
}

void jo_barr_lincomb(int dtype, int64_t nb, void *const *dst, const int64_t *lens, int k,
                     const double *coef, const int32_t *flags, const void *const *const *srcs)
{
    DISPATCH(dtype, barr_lincomb_f32(nb, dst, lens, k, coef, flags, srcs), barr_lincomb_f64(nb, dst, lens, k, coef, flags, srcs),
             barr_lincomb_c32(nb, dst, lens, k, coef, flags, srcs), barr_lincomb_c64(nb, dst, lens, k, coef, flags, srcs));
}

void jo_child_mul(int dtype, const jo_block *b, void *d, const void *m)
{
    DISPATCH(dtype, child_mul_f32(b, d, m), child_mul_f64(b, d, m), child_mul_c32(b, d, m), child_mul_c64(b, d, m));
}

void jo_child_mul_adj(int dtype, const jo_block *b, void *m, const void *d)
{
    DISPATCH(dtype, child_mul_adj_f32(b, m, d), child_mul_adj_f64(b, m, d), child_mul_adj_c32(b, m, d),
             child_mul_adj_c64(b, m, d));
}

void jo_block_df(int dtype, int64_t nrow, int64_t ncol, const jo_block *ops, void *const *d_arrays,
                 const void *const *m_arrays)
{
    DISPATCH(dtype, block_df_f32(nrow, ncol, ops, d_arrays, m_arrays), block_df_f64(nrow, ncol, ops, d_arrays, m_arrays),
             block_df_c32(nrow, ncol, ops, d_arrays, m_arrays), block_df_c64(nrow, ncol, ops, d_arrays, m_arrays));
}

void jo_block_df_adj(int dtype, int64_t nrow, int64_t ncol, const jo_block *ops, void *const *m_arrays,
                     const void *const *d_arrays)
{
    DISPATCH(dtype, block_df_adj_f32(nrow, ncol, ops, m_arrays, d_arrays),
             block_df_adj_f64(nrow, ncol, ops, m_arrays, d_arrays),
             block_df_adj_c32(nrow, ncol, ops, m_arrays, d_arrays),
             block_df_adj_c64(nrow, ncol, ops, m_arrays, d_arrays));
}

void jo_block_f(int dtype, int64_t nrow, int64_t ncol, const jo_block *ops, void *const *d_arrays,
                const void *const *m_arrays)
{
    DISPATCH(dtype, block_f_f32(nrow, ncol, ops, d_arrays, m_arrays), block_f_f64(nrow, ncol, ops, d_arrays, m_arrays),
             block_f_c32(nrow, ncol, ops, d_arrays, m_arrays), block_f_c64(nrow, ncol, ops, d_arrays, m_arrays));
}

void jo_normal_df(int dtype, int64_t nrow, int64_t ncol, const jo_block *ops, void *const *y_arrays,
                  const void *const *m_arrays)
{
    DISPATCH(dtype, normal_df_f32(nrow, ncol, ops, y_arrays, m_arrays), normal_df_f64(nrow, ncol, ops, y_arrays, m_arrays),
             normal_df_c32(nrow, ncol, ops, y_arrays, m_arrays), normal_df_c64(nrow, ncol, ops, y_arrays, m_arrays));
}

#ifdef _OPENMP
#include <omp.h>
#endif

/* Both all-cores loops give a thread the SAME element chunks of every row (static schedule over chunks, rows inside), and
 * jo_fill_u01_omp_f32 first-touches the arrays with that very partition, so on a multi-socket host every thread streams
 * memory of its own NUMA node. */
#define JO_OMP_CHUNK 16384

int jo_fill_u01_omp_f32(int64_t nrow, int64_t n, uint64_t seed, int64_t row0, float *const *rows)
{
    int nt = 1;
    const uint64_t key = jo_rng_key(seed, 0);
#pragma omp parallel
    {
#ifdef _OPENMP
#pragma omp single
        nt = omp_get_num_threads();
#endif
#pragma omp for schedule(static)
        for (int64_t c = 0; c < (n + JO_OMP_CHUNK - 1) / JO_OMP_CHUNK; c++) {
            const int64_t lo = c * JO_OMP_CHUNK, hi = lo + JO_OMP_CHUNK < n ? lo + JO_OMP_CHUNK : n;
            for (int64_t i = 0; i < nrow; i++)
                for (int64_t k = lo; k < hi; k++) {
                    rows[i][k] = jo_u01_f32(key, (row0 + i) * n + k);
                }
        }
    }
    return nt;
}

int jo_tall_diag_fwd_omp_f32(int64_t nrow, int64_t n, const float *const *a, const float *m, float *const *d)
{
    int nt = 1;
#pragma omp parallel
    {
#ifdef _OPENMP
#pragma omp single
        nt = omp_get_num_threads();
#endif
#pragma omp for schedule(static)
        for (int64_t c = 0; c < (n + JO_OMP_CHUNK - 1) / JO_OMP_CHUNK; c++) {
            const int64_t lo = c * JO_OMP_CHUNK, hi = lo + JO_OMP_CHUNK < n ? lo + JO_OMP_CHUNK : n;
            for (int64_t i = 0; i < nrow; i++)
                for (int64_t k = lo; k < hi; k++) d[i][k] = a[i][k] * m[k];              /* d_i .= diagonal_i .* m */
        }
    }
    return nt;
}

int jo_tall_diag_adj_omp_f32(int64_t nrow, int64_t n, const float *const *a, float *m, const float *const *d)
{
    int nt = 1;
#pragma omp parallel
    {
#ifdef _OPENMP
#pragma omp single
        nt = omp_get_num_threads();
#endif
#pragma omp for schedule(static)
        for (int64_t c = 0; c < (n + JO_OMP_CHUNK - 1) / JO_OMP_CHUNK; c++) {
            const int64_t lo = c * JO_OMP_CHUNK, hi = lo + JO_OMP_CHUNK < n ? lo + JO_OMP_CHUNK : n;
            for (int64_t k = lo; k < hi; k++) m[k] = 0.0f;                            /* _m .= 0 */
            for (int64_t i = 0; i < nrow; i++)                                         /* rows in order */
                for (int64_t k = lo; k < hi; k++) { float p = a[i][k] * d[i][k]; m[k] = m[k] + p; }
        }
    }
    return nt;
}
