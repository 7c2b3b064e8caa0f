/* host_pair.c -- a C host of the drop-in boundary: no Python, no torch, nothing but include/jetship.h.
 *
 * TEST CODE.  It drives libjetship.so exactly as the reference-side binding would (one call per mul!), on the
 * BASELINE.json path at a small size, and checks every result bit for bit against the CPU oracle
 * (oracle/libjets_oracle.so -- test infrastructure, linked only here).  Built and run by tests/test_c_host.py:
 *
 *   gcc -O2 -std=gnu11 -Iinclude -Ioracle tests/c_host/host_pair.c -o host_pair \
 *       -Ljets.jl_amd -ljetship -Loracle -ljets_oracle -Wl,-rpath,...
 *
 * Exit code 0 and a last line "C HOST OK" on success.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jetship.h"
#include "jets_oracle.h"

#define CK(call)                                                                            \
    do {                                                                                    \
        int st_ = (call);                                                                   \
        if (st_ != JH_OK) {                                                                 \
            fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #call, st_, jh_last_error()); \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

#define REQUIRE(cond, what)                                                  \
    do {                                                                     \
        if (!(cond)) { fprintf(stderr, "FAILED: %s (%s:%d)\n", what, __FILE__, __LINE__); return 1; } \
    } while (0)

enum { NROW = 12, EDGE = 24 };

int main(void)
{
    const int64_t n = (int64_t)EDGE * EDGE * EDGE;      /* one block: EDGE^3 Float32, contiguous, column-major */
    int ndev = 0;
    REQUIRE(jh_abi_version() == JETSHIP_ABI_VERSION, "the library and the header it was compiled against speak the same ABI version");
    CK(jh_device_count(&ndev));
    REQUIRE(ndev >= 1, "no HIP device");
    CK(jh_init(0));
    char name[128];
    int64_t hbm_total = 0, hbm_free = 0;
    int cus = 0;
    CK(jh_device_info(name, (int)sizeof name, &hbm_total, &hbm_free, &cus));
    printf("device: %s, %d CUs, %.0f GiB HBM, ABI v%d\n", name, cus, (double)hbm_total / (1 << 30), jh_abi_version());

    /* ---- block vectors: coefficients (one slab = all diagonals), range vector d, domain vectors m, mt, y */
    int64_t lens[NROW];
    for (int i = 0; i < NROW; i++) lens[i] = n;
    jh_bvec *coeff = NULL, *d = NULL, *m = NULL, *mt = NULL, *y = NULL;
    CK(jh_bvec_create(NROW, lens, JH_F32, &coeff));
    CK(jh_bvec_create(NROW, lens, JH_F32, &d));
    CK(jh_bvec_create(1, lens, JH_F32, &m));
    CK(jh_bvec_create(1, lens, JH_F32, &mt));
    CK(jh_bvec_create(1, lens, JH_F32, &y));
    CK(jh_fill_uniform(coeff, 1, 0, 0));               /* seeds as in SURVEY.md 8d: a = 1, m = 2, d = 3 */
    CK(jh_fill_uniform(m, 2, 0, 0));
    CK(jh_fill(d, 7.0, 0.0));                          /* dirty output */
    CK(jh_fill(mt, 7.0, 0.0));

    /* ---- the operator: NROW x 1 diagonal blocks, coefficient pointers borrowed from the slab */
    jh_block_desc desc[NROW];
    memset(desc, 0, sizeof desc);
    for (int i = 0; i < NROW; i++) {
        int64_t off = 0, len = 0;
        void *ptr = NULL;
        CK(jh_bvec_block(coeff, i, &off, &len, &ptr));
        REQUIRE(off == (int64_t)i * n && len == n, "JetBSpace ranges (src/Jets.jl:742-748)");
        desc[i].kind = JH_OP_DIAG;
        desc[i].coeff = ptr;
        desc[i].nr = desc[i].nc = n;
    }
    int64_t col_len[1] = {n};
    jh_blockop *A = NULL;
    CK(jh_blockop_create(NROW, 1, desc, lens, col_len, JH_F32, &A));

    /* ---- the hot path: one call per mul! */
    CK(jh_blockop_mul(A, d, m));                       /* d = A m      (JetBlock_df!)  */
    CK(jh_blockop_mul_adj(A, mt, d));                  /* mt = A' d    (JetBlock_df'!) */
    CK(jh_blockop_normal_mul(A, y, m));                /* y = A'(A m)  fused           */
    double nrm = 0, dre = 0, dim = 0, mn = 0, mx = 0;
    CK(jh_norm(d, 2.0, &nrm));
    CK(jh_dot(d, d, &dre, &dim));
    CK(jh_extrema(d, &mn, &mx));

    /* ---- bring the results to the host */
    float *hd = malloc((size_t)NROW * n * sizeof(float)), *hmt = malloc((size_t)n * sizeof(float)), *hy = malloc((size_t)n * sizeof(float));
    REQUIRE(hd && hmt && hy, "host allocation");
    CK(jh_download(d, 0, (int64_t)NROW * n, hd));
    CK(jh_download(mt, 0, n, hmt));
    CK(jh_download(y, 0, n, hy));
    {   /* the fused A'A range by range (what a host pipelining the exchange of y enqueues) into storage that was never zeroed */
        jh_bvec *y2 = NULL;
        const int64_t len1[1] = {n};
        CK(jh_bvec_create_uninit(1, len1, JH_F32, &y2));
        const int64_t cut = n >= 8 ? (n / 2) / 4 * 4 : n;
        CK(jh_blockop_normal_mul_range(A, y2, m, 0, cut));
        CK(jh_blockop_normal_mul_range(A, y2, m, cut, n - cut));
        float *hy2 = malloc((size_t)n * sizeof(float));
        REQUIRE(hy2, "host allocation");
        CK(jh_download(y2, 0, n, hy2));
        REQUIRE(memcmp(hy2, hy, (size_t)n * sizeof(float)) == 0, "ranged fused A'A == the whole one, bit for bit");
        REQUIRE(jh_blockop_normal_mul_range(A, y2, m, n - 4, 8) == JH_ERR_INVALID, "a range past the end is refused");
        free(hy2);
        CK(jh_bvec_destroy(y2));
        CK(jh_trim());                                   /* whatever the slab cache kept goes back to the driver */
        int64_t cached = -1;
        CK(jh_tune_get("slab_cached_mib", &cached));
        REQUIRE(cached == 0, "jh_trim empties the slab cache");
    }

    /* ---- the oracle on the same seeded inputs */
    float *oa = malloc((size_t)NROW * n * sizeof(float)), *om = malloc((size_t)n * sizeof(float));
    float *od = malloc((size_t)NROW * n * sizeof(float)), *omt = malloc((size_t)n * sizeof(float));
    REQUIRE(oa && om && od && omt, "host allocation");
    jo_rng_u01(JO_F32, 1, 0, 0, (int64_t)NROW * n, oa);
    jo_rng_u01(JO_F32, 2, 0, 0, n, om);
    jo_block ops[NROW];
    void *d_arrays[NROW];
    const void *m_arrays[1] = {om};
    memset(ops, 0, sizeof ops);
    for (int i = 0; i < NROW; i++) {
        ops[i].kind = JO_OP_DIAG;
        ops[i].coeff = oa + (size_t)i * n;
        ops[i].nr = ops[i].nc = n;
        d_arrays[i] = od + (size_t)i * n;
    }
    for (int64_t k = 0; k < NROW * n; k++) od[k] = 7.0f;
    for (int64_t k = 0; k < n; k++) omt[k] = 7.0f;
    jo_block_df(JO_F32, NROW, 1, ops, d_arrays, m_arrays);
    void *mt_arrays[1] = {omt};
    jo_block_df_adj(JO_F32, NROW, 1, ops, mt_arrays, (const void *const *)d_arrays);

    REQUIRE(memcmp(hd, od, (size_t)NROW * n * sizeof(float)) == 0, "forward bit-exact vs oracle");
    REQUIRE(memcmp(hmt, omt, (size_t)n * sizeof(float)) == 0, "adjoint bit-exact vs oracle");
    REQUIRE(memcmp(hy, omt, (size_t)n * sizeof(float)) == 0, "fused A'A bit-exact vs the unfused pair");
    int64_t olens[NROW];
    for (int i = 0; i < NROW; i++) olens[i] = n;
    const double onrm = jo_barr_norm(JO_F32, NROW, (const void *const *)d_arrays, olens, 2.0);
    double ore = 0, oim = 0, omn = 0, omx = 0;
    jo_barr_dot(JO_F32, NROW, (const void *const *)d_arrays, (const void *const *)d_arrays, olens, &ore, &oim);
    jo_barr_extrema(JO_F32, NROW, (const void *const *)d_arrays, olens, &omn, &omx);
    REQUIRE(fabs(nrm - onrm) <= 1e-5 * onrm, "norm within 1e-5");
    REQUIRE(fabs(dre - ore) <= 1e-5 * fabs(ore), "dot within 1e-5");
    REQUIRE(mn == omn && mx == omx, "extrema exact");

    /* ---- dot-product test (src/Jets.jl:1211-1226): <m, A'd> == <A m, d> */
    double lhs = 0, rhs = 0, tmp = 0;
    CK(jh_dot(m, mt, &lhs, &tmp));
    rhs = dre;
    REQUIRE(fabs(lhs - rhs) <= 1e-5 * fabs(lhs + rhs), "dot-product test");

    /* ---- errors come back as status codes + a message, never as a crash */
    REQUIRE(jh_blockop_mul(A, m, m) == JH_ERR_INVALID, "length mismatch is reported");
    REQUIRE(strlen(jh_last_error()) > 0, "error message is set");
    REQUIRE(jh_setblock_fill(d, NROW, 0.0, 0.0) == JH_ERR_INVALID, "block index out of range is reported");

    /* ---- getblock! / setblock! round trip through a host block (src/Jets.jl:915-916) */
    CK(jh_setblock_fill(d, 3, 2.5, 0.0));
    CK(jh_getblock_copy(d, 3, hy, 0));
    for (int64_t k = 0; k < n; k++) REQUIRE(hy[k] == 2.5f, "setblock!(d, 4, 2.5); getblock!(d, 4, out)");
    CK(jh_setblock_copy(d, 5, hmt, 0));
    CK(jh_getblock_copy(d, 5, hy, 0));
    REQUIRE(memcmp(hy, hmt, (size_t)n * sizeof(float)) == 0, "setblock!(d, 6, array) round trip");

    /* ---- the solver loop behind the ABI: b = A x_true, solve with LSQR (one pass over A and u per iteration) */
    {
        jh_bvec *xt = NULL, *bb = NULL, *xs = NULL;
        CK(jh_bvec_create(1, lens, JH_F32, &xt));
        CK(jh_bvec_create(NROW, lens, JH_F32, &bb));
        CK(jh_bvec_create(1, lens, JH_F32, &xs));
        CK(jh_fill_uniform(xt, 4, 0, 0));
        CK(jh_blockop_mul(A, bb, xt));
        jh_lsqr_result lr;
        double hist[2 * 40];
        CK(jh_lsqr_solve(A, bb, xs, 0, 0.0, 1e-7, 1e-7, 0.0, 40, 0, &lr, hist));
        REQUIRE(lr.itn >= 1 && lr.itn <= 40 && lr.istop >= 1, "LSQR stops by a rule");
        REQUIRE(hist[2 * (lr.itn - 1)] < 1e-4 * hist[0] + 1e-6, "residual norm falls");
        float *hx = malloc((size_t)n * sizeof(float)), *hxt = malloc((size_t)n * sizeof(float));
        REQUIRE(hx && hxt, "host allocation");
        CK(jh_download(xs, 0, n, hx));
        CK(jh_download(xt, 0, n, hxt));
        double num = 0, den = 0;
        for (int64_t k = 0; k < n; k++) { num += (double)(hx[k] - hxt[k]) * (hx[k] - hxt[k]); den += (double)hxt[k] * hxt[k]; }
        REQUIRE(sqrt(num / den) < 1e-3, "LSQR recovers x_true");
        printf("jh_lsqr_solve: %d iterations, istop %d, ||x - x_true|| / ||x_true|| = %.2e\n", lr.itn, lr.istop, sqrt(num / den));
        /* the same solve without a history buffer, and through the host loop (knob lsqr_graph = 0): the same iteration count and x */
        jh_lsqr_result lr2, lr3;
        CK(jh_blockop_mul(A, bb, xt));
        CK(jh_lsqr_solve(A, bb, xs, 0, 0.0, 1e-7, 1e-7, 0.0, 40, 0, &lr2, NULL));
        float *hx2 = malloc((size_t)n * sizeof(float));
        REQUIRE(hx2, "host allocation");
        CK(jh_download(xs, 0, n, hx2));
        REQUIRE(lr2.itn == lr.itn && lr2.istop == lr.istop && memcmp(hx2, hx, (size_t)n * sizeof(float)) == 0, "LSQR without a history buffer: same solve");
        int64_t replays = -1;
        CK(jh_tune_get("last_lsqr_graph", &replays));
        REQUIRE(replays > 0, "a small operator's LSQR loop is replayed as a hipGraph");
        CK(jh_tune_set("lsqr_graph", 0));
        CK(jh_blockop_mul(A, bb, xt));
        CK(jh_lsqr_solve(A, bb, xs, 0, 0.0, 1e-7, 1e-7, 0.0, 40, 0, &lr3, NULL));
        CK(jh_tune_set("lsqr_graph", 1));
        CK(jh_download(xs, 0, n, hx2));
        REQUIRE(lr3.itn == lr.itn && memcmp(hx2, hx, (size_t)n * sizeof(float)) == 0, "graph-replayed loop == host loop, bit for bit");
        /* round 3: CGLS on the same system (two passes per iteration, no range-sized temporary); bb comes back holding r = b - A x */
        jh_lsqr_result cr;
        double chist[2 * 60];
        CK(jh_blockop_mul(A, bb, xt));
        CK(jh_cgls_solve(A, bb, xs, 0, 0.0, 1e-7, 1e-6, 60, 0, &cr, chist));
        REQUIRE(cr.itn >= 1 && cr.itn <= 60 && (cr.istop == 1 || cr.istop == 2), "CGLS stops by a rule");
        CK(jh_download(xs, 0, n, hx2));
        num = 0;
        for (int64_t k = 0; k < n; k++) num += (double)(hx2[k] - hxt[k]) * (hx2[k] - hxt[k]);
        REQUIRE(sqrt(num / den) < 1e-3, "CGLS recovers x_true");
        double rnorm = 0.0;
        CK(jh_norm(bb, 2.0, &rnorm));
        REQUIRE(fabs(rnorm - cr.r1norm) <= 1e-5 * (rnorm + cr.r1norm) + 1e-12, "on return u holds the residual the record reports");
        printf("jh_cgls_solve: %d iterations, istop %d, ||x - x_true|| / ||x_true|| = %.2e\n", cr.itn, cr.istop, sqrt(num / den));
        REQUIRE(jh_comm_available() == JH_OK, "an RCCL can be loaded (probe without side effects)");
        free(hx2);
        free(hx); free(hxt);
        CK(jh_bvec_destroy(xt));
        CK(jh_bvec_destroy(bb));
        CK(jh_bvec_destroy(xs));
    }

    /* ---- round-2 entry points from plain C: the one-pass step on two element ranges with the deferred ||u||^2, the
     *      partitioned solve (one rank, no communicator: equals the local solve), per-operator tune export / import */
    {
        jh_bvec *u = NULL, *v = NULL, *w = NULL, *w2 = NULL, *u2 = NULL;
        CK(jh_bvec_create(NROW, lens, JH_F32, &u));
        CK(jh_bvec_create(NROW, lens, JH_F32, &u2));
        CK(jh_bvec_create(1, lens, JH_F32, &v));
        CK(jh_bvec_create(1, lens, JH_F32, &w));
        CK(jh_bvec_create(1, lens, JH_F32, &w2));
        CK(jh_fill_uniform(u, 7, 0, 0));
        CK(jh_fill_uniform(u2, 7, 0, 0));
        CK(jh_fill_uniform(v, 8, 0, 0));
        double whole = 0, ranged = 0;
        CK(jh_blockop_bidiag_step(A, u, v, w, 0.75, -0.5, &whole));
        const int64_t half = (n / 2) / 4 * 4;
        CK(jh_normsq_reset());
        CK(jh_blockop_bidiag_step_range(A, u2, v, w2, 0.75, -0.5, 0, half, NULL));
        CK(jh_blockop_bidiag_step_range(A, u2, v, w2, 0.75, -0.5, half, n - half, NULL));
        CK(jh_normsq_read(&ranged));
        float *ha = malloc((size_t)n * sizeof(float)), *hb = malloc((size_t)n * sizeof(float));
        REQUIRE(ha && hb, "host allocation");
        CK(jh_download(w, 0, n, ha));
        CK(jh_download(w2, 0, n, hb));
        REQUIRE(memcmp(ha, hb, (size_t)n * sizeof(float)) == 0, "ranged one-pass step == whole step (w)");
        REQUIRE(fabs(whole - ranged) <= 1e-12 * whole, "deferred ||u||^2 of the ranges adds up");
        free(ha); free(hb);
        int64_t walk = 99;
        CK(jh_blockop_tune_get(A, "fwd_walk", &walk));
        REQUIRE(walk == -1, "a small operator's forward walk is not measured");
        CK(jh_blockop_tune_set(A, "step_mode", 0));
        CK(jh_blockop_tune_get(A, "step_mode", &walk));
        REQUIRE(walk == 0, "per-operator tune round trip");
        REQUIRE(jh_blockop_tune_set(A, "no_such_knob", 1) == JH_ERR_INVALID, "unknown per-operator knob is reported");
        jh_lsqr_result lr;
        CK(jh_fill(v, 0.0, 0.0));
        CK(jh_lsqr_solve_partitioned(A, u, v, 0, 0.0, 0.0, 0.0, 0.0, 5, 1, &lr, NULL));
        REQUIRE(lr.itn == 5, "partitioned solve with one rank runs without a communicator");
        CK(jh_bvec_destroy(u)); CK(jh_bvec_destroy(u2)); CK(jh_bvec_destroy(v)); CK(jh_bvec_destroy(w)); CK(jh_bvec_destroy(w2));
    }

    /* ---- round 5 from plain C: `1.0*A1 - 2.0*A2 + 3.14*A3` (src/Jets.jl:686) with Float64 scalars on Float32 operators as ONE fused call each way
     *      (jh_blocksum_mul_typed / _adj_typed with JH_SCALAR_WIDE), against the reference's loop spelled out: product in Float32, scalar stage
     *      promoted and rounded once (1159-1160), signed add in Float32 (644, 653) */
    {
        jh_bvec *c2 = NULL, *c3 = NULL, *ds = NULL, *ms = NULL;
        jh_blockop *A2 = NULL, *A3 = NULL;
        CK(jh_bvec_create(NROW, lens, JH_F32, &c2));
        CK(jh_bvec_create(NROW, lens, JH_F32, &c3));
        CK(jh_bvec_create(NROW, lens, JH_F32, &ds));
        CK(jh_bvec_create(1, lens, JH_F32, &ms));
        CK(jh_fill_uniform(c2, 11, 0, 0));
        CK(jh_fill_uniform(c3, 12, 0, 0));
        jh_block_desc d2[NROW], d3[NROW];
        memset(d2, 0, sizeof d2);
        memset(d3, 0, sizeof d3);
        for (int i = 0; i < NROW; i++) {
            int64_t off = 0, len = 0;
            void *p2 = NULL, *p3 = NULL;
            CK(jh_bvec_block(c2, i, &off, &len, &p2));
            CK(jh_bvec_block(c3, i, &off, &len, &p3));
            d2[i].kind = d3[i].kind = JH_OP_DIAG;
            d2[i].coeff = p2;
            d3[i].coeff = p3;
            d2[i].nr = d2[i].nc = d3[i].nr = d3[i].nc = n;
        }
        CK(jh_blockop_create(NROW, 1, d2, lens, lens, JH_F32, &A2));
        CK(jh_blockop_create(NROW, 1, d3, lens, lens, JH_F32, &A3));
        const jh_blockop *terms[3] = {A, A2, A3};
        const double scale[3] = {1.0, 2.0, 3.14}, sign[3] = {1.0, -1.0, 1.0};
        const int32_t flags[3] = {JH_SCALAR_WIDE, JH_SCALAR_WIDE, JH_SCALAR_WIDE};
        CK(jh_fill(ds, 7.0, 0.0));                                                   /* dirty: the sum starts from d .= 0 (640) */
        CK(jh_blocksum_mul_typed(3, terms, scale, flags, sign, ds, m));
        CK(jh_fill(ms, 7.0, 0.0));
        CK(jh_blocksum_mul_adj_typed(3, terms, scale, flags, sign, ms, ds));
        float *hs = malloc((size_t)NROW * n * sizeof(float)), *hms = malloc((size_t)n * sizeof(float));
        float *o2 = malloc((size_t)NROW * n * sizeof(float)), *o3 = malloc((size_t)NROW * n * sizeof(float));
        float *ws = malloc((size_t)NROW * n * sizeof(float)), *wm = malloc((size_t)n * sizeof(float));
        REQUIRE(hs && hms && o2 && o3 && ws && wm, "host allocation");
        CK(jh_download(ds, 0, (int64_t)NROW * n, hs));
        CK(jh_download(ms, 0, n, hms));
        jo_rng_u01(JO_F32, 11, 0, 0, (int64_t)NROW * n, o2);
        jo_rng_u01(JO_F32, 12, 0, 0, (int64_t)NROW * n, o3);
        const float *oc[3] = {oa, o2, o3};
        for (int64_t k = 0; k < (int64_t)NROW * n; k++) {                            /* forward, element by element */
            float acc = 0.0f;
            for (int t = 0; t < 3; t++) {
                const volatile float prod = oc[t][k] * om[k % n];                    /* mul!(_d, A_t, m): a Float32 product */
                const float term = (float)(scale[t] * (double)prod);                 /* _d .= a * tmp with a::Float64: promoted, rounded once */
                acc = sign[t] > 0 ? acc + term : acc - term;                         /* broadcast!(sgn, d, d, _d) */
            }
            ws[k] = acc;
        }
        REQUIRE(memcmp(hs, ws, (size_t)NROW * n * sizeof(float)) == 0, "fused JetSum with Float64 scalars: forward bit-exact vs the reference's loop");
        for (int64_t e = 0; e < n; e++) {                                            /* adjoint: per term the ordered row sum of conj(a) .* (s * d_i) */
            float acc = 0.0f;
            for (int t = 0; t < 3; t++) {
                float rowsum = 0.0f;
                for (int i = 0; i < NROW; i++) {
                    const float sd = (float)(scale[t] * (double)ws[(size_t)i * n + e]);
                    const volatile float p = oc[t][(size_t)i * n + e] * sd;
                    rowsum = rowsum + p;
                }
                acc = sign[t] > 0 ? acc + rowsum : acc - rowsum;
            }
            wm[e] = acc;
        }
        REQUIRE(memcmp(hms, wm, (size_t)n * sizeof(float)) == 0, "fused JetSum with Float64 scalars: adjoint bit-exact vs the reference's loop");
        /* (a * A) m in one pass with the scalar's type */
        CK(jh_blockop_mul_scaled(A2, ds, m, 3.14, JH_SCALAR_WIDE));
        CK(jh_download(ds, 0, (int64_t)NROW * n, hs));
        for (int64_t k = 0; k < (int64_t)NROW * n; k++) {
            const volatile float prod = o2[k] * om[k % n];
            ws[k] = (float)(3.14 * (double)prod);
        }
        REQUIRE(memcmp(hs, ws, (size_t)NROW * n * sizeof(float)) == 0, "(3.14 * A) m in one pass: the promoted product rounded once");
        REQUIRE(jh_blocksum_mul_typed(3, terms, scale, (const int32_t[3]){JH_SCALAR_COMPLEX, 0, 0}, sign, ds, m) == JH_ERR_UNSUPPORTED,
                "a Complex scale is refused with JH_ERR_UNSUPPORTED (the binding runs the unfused chain)");
        printf("typed fused sums and scaled passes from C: ok\n");
        free(hs); free(hms); free(o2); free(o3); free(ws); free(wm);
        CK(jh_blockop_destroy(A2)); CK(jh_blockop_destroy(A3));
        CK(jh_bvec_destroy(c2)); CK(jh_bvec_destroy(c3)); CK(jh_bvec_destroy(ds)); CK(jh_bvec_destroy(ms));
    }

    /* ---- ONE process, several contexts (include/jetship.h Conventions): a team of two contexts -- one per device
     *      when two devices are visible (RCCL), else two streams of this device (the sum is a device kernel) -- each holding
     *      half of A's rows; the grouped, ranged all-reduce makes both replicas of A'd the full sum */
    {
        int ctx[2] = {-1, -1}, dev0 = -1;
        CK(jh_context_current(&ctx[0], &dev0));
        if (ndev >= 2) { CK(jh_init(1)); CK(jh_context_current(&ctx[1], NULL)); }
        else CK(jh_context_create(0, &ctx[1]));
        REQUIRE(ctx[1] != ctx[0], "a second context");
        CK(jh_comm_init_all(2, ctx));
        const int half_rows = NROW / 2;
        jh_bvec *cf[2], *dd[2], *mm[2], *mo[2];
        jh_blockop *Ah[2];
        for (int k = 0; k < 2; k++) {
            CK(jh_context_use(ctx[k]));                         /* factories allocate in the CURRENT context */
            CK(jh_bvec_create(half_rows, lens, JH_F32, &cf[k]));
            CK(jh_bvec_create(half_rows, lens, JH_F32, &dd[k]));
            CK(jh_bvec_create(1, lens, JH_F32, &mm[k]));
            CK(jh_bvec_create(1, lens, JH_F32, &mo[k]));
            CK(jh_fill_uniform(cf[k], 1, 0, (int64_t)k * half_rows * n));      /* rows k*6 .. k*6+5 of the same generator stream */
            CK(jh_fill_uniform(mm[k], 2, 0, 0));
            int kc = -1;
            CK(jh_bvec_context(cf[k], &kc, NULL));
            REQUIRE(kc == ctx[k], "a vector remembers its context");
            jh_block_desc dk[NROW];
            memset(dk, 0, sizeof dk);
            for (int i = 0; i < half_rows; i++) {
                void *ptr = NULL;
                CK(jh_bvec_block(cf[k], i, NULL, NULL, &ptr));
                dk[i].kind = JH_OP_DIAG; dk[i].coeff = ptr; dk[i].nr = dk[i].nc = n;
            }
            CK(jh_blockop_create(half_rows, 1, dk, lens, col_len, JH_F32, &Ah[k]));
        }
        CK(jh_context_use(ctx[0]));                             /* whatever is current: calls run in their handles' context */
        for (int k = 0; k < 2; k++) CK(jh_blockop_mul(Ah[k], dd[k], mm[k]));
        REQUIRE(jh_blockop_mul(Ah[0], dd[1], mm[0]) == JH_ERR_INVALID, "handles of different contexts in one call are refused");
        REQUIRE(jh_comm_allreduce_sum(mo[0]) == JH_ERR_STATE, "a member's collective outside a group is refused");
        const int64_t cut = (n / 2) / 16384 * 16384 > 0 ? (n / 2) / 16384 * 16384 : (n / 2) / 4 * 4;
        const int64_t lo[2] = {0, cut}, cnt[2] = {cut, n - cut};
        for (int r = 0; r < 2; r++) {
            for (int k = 0; k < 2; k++) CK(jh_blockop_mul_adj_range(Ah[k], mo[k], dd[k], lo[r], cnt[r]));
            CK(jh_comm_group_begin());
            for (int k = 0; k < 2; k++) CK(jh_comm_allreduce_sum_range(mo[k], lo[r], cnt[r]));
            CK(jh_comm_group_end());
        }
        float *h0 = malloc((size_t)n * sizeof(float)), *h1 = malloc((size_t)n * sizeof(float));
        REQUIRE(h0 && h1, "host allocation");
        for (int k = 0; k < 2; k++) { CK(jh_context_use(ctx[k])); CK(jh_comm_join()); }
        CK(jh_download(mo[0], 0, n, h0));
        CK(jh_download(mo[1], 0, n, h1));
        REQUIRE(memcmp(h0, h1, (size_t)n * sizeof(float)) == 0, "the members' replicas of A'd are identical");
        double num = 0, den = 0;                               /* vs the one-context ordered sum: another summation order */
        for (int64_t k = 0; k < n; k++) { num += (double)(h0[k] - hmt[k]) * (h0[k] - hmt[k]); den += (double)hmt[k] * hmt[k]; }
        REQUIRE(sqrt(num / den) < 1e-6, "team adjoint within 1e-6 of the ordered sum");
        printf("team of two contexts (%s): replicas identical, rel. difference to the ordered sum %.1e\n", ndev >= 2 ? "two devices, RCCL" : "one device", sqrt(num / den));
        free(h0); free(h1);
        CK(jh_context_use(ctx[0]));
        CK(jh_comm_destroy());
        for (int k = 0; k < 2; k++) {
            CK(jh_blockop_destroy(Ah[k]));
            CK(jh_bvec_destroy(cf[k])); CK(jh_bvec_destroy(dd[k])); CK(jh_bvec_destroy(mm[k])); CK(jh_bvec_destroy(mo[k]));
        }
        if (ndev < 2) CK(jh_context_destroy(ctx[1]));
        CK(jh_context_use(ctx[0]));
    }

    CK(jh_blockop_destroy(A));
    CK(jh_bvec_destroy(coeff));
    CK(jh_bvec_destroy(d));
    CK(jh_bvec_destroy(m));
    CK(jh_bvec_destroy(mt));
    CK(jh_shutdown());
    /* a handle that outlives jh_shutdown: its calls fail with a status, its destruction still releases the memory */
    REQUIRE(jh_fill(y, 1.0, 0.0) == JH_ERR_STATE, "a vector of a destroyed context is refused, not dereferenced");
    REQUIRE(strlen(jh_last_error()) > 0, "error message is set");
    CK(jh_bvec_destroy(y));
    REQUIRE(jh_synchronize() == JH_ERR_STATE, "handle-less calls after jh_shutdown report the missing context");
    free(hd); free(hmt); free(hy); free(oa); free(om); free(od); free(omt);
    printf("forward, adjoint, fused A'A: bit-exact vs oracle; norm/dot within 1e-5; dot-product test %.6e ~ %.6e\n", lhs, rhs);
    printf("C HOST OK\n");
    return 0;
}
