// jh_lsqr.hip -- LSQR (Paige & Saunders, ACM TOMS 8(1), 1982) over a tall all-DIAG block operator, entirely behind the C
// ABI: the caller of the block mul! path that BASELINE.json configs[4] names.  The reference reaches it as
// `lsqr(vec(A), vec(d))` of IterativeSolvers.jl (src/Jets.jl:1143-1152, docs/src/index.md:235-246), an un-vendored
// package, so this is the published algorithm written from scratch -- the same recurrences as jets.jl_amd/lsqr.py, which
// stays the driver for arbitrary operators; results are checked against the fp64 CPU LSQR of oracle/lsqr_ref.py.
//
// One iteration = ONE pass over the operator and the range vector (jh_blockop_bidiag_step: u <- A v - (alpha/beta) u,
// ||u||^2 and A'u together), then domain-sized updates.  u is never normalised in memory: it holds beta*u and the scale is
// carried into the next step.  With jh_comm_init_rank done (one process per GPU, this rank holding its block rows) the local
// A'u is all-reduced with RCCL and ||u||^2 with a scalar all-reduce: row-partitioned LSQR without a line of host code.
#include "jh_internal.h"

#include <cmath>

namespace {

// ---- the domain-side half of an iteration in two kernels (instead of five broadcasts and two norms) ----------------
// All coefficients are real, so a complex vector is 2n reals here.  Rounding sequence of every element == the broadcasts
// they replace (jh_lincomb: every product rounded, then the sum); the norms are summed in another (fixed) order than
// jh_norm's, so alpha agrees with the unfused loop to fp64 round-off, not to the bit.
template <int BLK> __device__ inline void wg_sum_to(double v, double *slot)
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

// v <- c0*atu + c1*v ; partial ||v||^2
template <typename S>
__global__ __launch_bounds__(256) void k_lsqr_vhat(S *v, const S *__restrict__ atu, int64_t n, S c0, S c1, double *__restrict__ partials)
{
    double nrm = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const S a = c0 * atu[i], b = c1 * v[i];
        const S r = a + b;
        v[i] = r;
        nrm += (double)r * (double)r;
    }
    wg_sum_to<256>(nrm, partials + blockIdx.x);
}

// v <- cv*v (normalise) ; x <- x + t1*w ; w <- v + t2*w ; partial ||w_new||^2
template <typename S>
__global__ __launch_bounds__(256) void k_lsqr_xw(S *v, S *x, S *w, int64_t n, S cv, S t1, S t2, double *__restrict__ partials)
{
    double nrm = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const S vn = cv * v[i];
        const S wo = w[i];
        const S xa = (S)1 * x[i], xb = t1 * wo;
        const S wa = (S)1 * vn, wb = t2 * wo;
        const S wn = wa + wb;
        v[i] = vn;
        x[i] = xa + xb;
        w[i] = wn;
        nrm += (double)wn * (double)wn;
    }
    wg_sum_to<256>(nrm, partials + blockIdx.x);
}

// one workgroup folds the per-workgroup partials in index order (deterministic)
__global__ __launch_bounds__(256) void k_lsqr_fold(const double *__restrict__ partials, int nparts, double *__restrict__ out)
{
    double v = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) v += partials[i];
    wg_sum_to<256>(v, out);
}

struct Tmp {                                     // domain-sized work vectors, released on every exit path
    jh_bvec *v = nullptr, *w = nullptr, *atu = nullptr;
    double *parts = nullptr;                     // 2 x grid per-workgroup partials + 2 result slots
    ~Tmp()
    {
        if (parts) (void)hipFree(parts);
        if (v) (void)jh_bvec_destroy(v);
        if (w) (void)jh_bvec_destroy(w);
        if (atu) (void)jh_bvec_destroy(atu);
    }
};

int lincomb1(jh_bvec *dst, double c0, const jh_bvec *x0)
{
    const double coef[2] = {c0, 0.0};
    const jh_bvec *xs[1] = {x0};
    return jh_lincomb(dst, 1, coef, xs);
}

// launch helpers: a complex vector is 2n reals for these real-coefficient updates
int launch_vhat(int dtype, jh_bvec *v, const jh_bvec *atu, double c0, double c1, double *parts, int grid)
{
    const bool f64 = (dtype == JH_F64 || dtype == JH_C64);
    const int64_t ns = v->length * (jh_dtype_complex(dtype) ? 2 : 1);
    hipStream_t st = jh_ctx().stream;
    if (f64) hipLaunchKernelGGL((k_lsqr_vhat<double>), dim3(grid), dim3(256), 0, st, (double *)v->data, (const double *)atu->data, ns, c0, c1, parts);
    else hipLaunchKernelGGL((k_lsqr_vhat<float>), dim3(grid), dim3(256), 0, st, (float *)v->data, (const float *)atu->data, ns, (float)c0, (float)c1, parts);
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

int launch_xw(int dtype, jh_bvec *v, jh_bvec *x, jh_bvec *w, double cv, double t1, double t2, double *parts, int grid)
{
    const bool f64 = (dtype == JH_F64 || dtype == JH_C64);
    const int64_t ns = v->length * (jh_dtype_complex(dtype) ? 2 : 1);
    hipStream_t st = jh_ctx().stream;
    if (f64) hipLaunchKernelGGL((k_lsqr_xw<double>), dim3(grid), dim3(256), 0, st, (double *)v->data, (double *)x->data, (double *)w->data, ns, cv, t1, t2, parts);
    else hipLaunchKernelGGL((k_lsqr_xw<float>), dim3(grid), dim3(256), 0, st, (float *)v->data, (float *)x->data, (float *)w->data, ns, (float)cv, (float)t1, (float)t2, parts);
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

}  // namespace

// ---- small operators: the recurrences live on the DEVICE and one iteration is a hipGraph -------------------------------------
// The loop below (lsqr_impl) synchronises the host twice per iteration to turn two sums into the next coefficients.  For an
// operator whose pass takes a few hundred microseconds that costs as much as the kernels (64 x 64^3: 84 us per iteration, 35 of
// them on the device).  Here the scalars of Paige & Saunders' recurrences are a struct in device memory, two one-thread kernels
// update it from the sums exactly as the host code does (same operations, same order, fp64), the step / v / x-w kernels read
// their coefficients from it, and a finished solve turns every kernel into a no-op -- so an iteration has fixed launch parameters,
// is captured ONCE as a hipGraph and replayed; the host looks at the `done` flag every few iterations.  Same kernels, same
// arithmetic: the iterates are those of lsqr_impl bit for bit.
struct LsqrDev {
    double alpha, beta, rhobar, phibar, anorm, ddnorm, res2, xxnorm, z, cs2, sn2, bnorm, wnorm, damp, atol, btol, ctol;
    double r1norm, r2norm, acond, arnorm, xnorm;
    double coef_step[2];                         // (1, -alpha / beta): the next one-pass step
    double coef_vhat[2];                         // (1 / beta, -beta)
    double coef_xw[3];                           // (cv, t1, t2)
    int itn, istop, done, pending, skipv, maxiter, force, halt;   // halt = done || pending: the NEXT step is not to run (five-node iteration below)
};

// after the step: beta from ||u||^2, ||w|| of the previous iteration, the coefficients of the v update
__device__ inline void lsqr_s1(LsqrDev *st, double normsq, const double *__restrict__ slot_w)
{
    st->itn++;
    st->beta = sqrt(normsq);
    if (st->itn >= 2) st->wnorm = sqrt(*slot_w);
    st->skipv = 1;
    if (st->beta > 0) {
        st->anorm = sqrt(st->anorm * st->anorm + st->alpha * st->alpha + st->beta * st->beta + st->damp * st->damp);
        st->coef_vhat[0] = 1.0 / st->beta;
        st->coef_vhat[1] = -st->beta;
        st->skipv = 0;
    }
}

// after the v update: alpha, the rotations, the coefficients of the x / w update, the stopping rules (lsqr_impl, line by line)
__device__ inline void lsqr_s2(LsqrDev *st, double sum_v, double *__restrict__ history)
{
    double alpha = st->alpha, beta = st->beta, rhobar = st->rhobar, phibar = st->phibar;
    double cv = 1.0;
    if (!st->skipv) {
        alpha = sqrt(sum_v);
        if (alpha > 0) cv = 1.0 / alpha;
    }
    const double damp = st->damp;
    const double rhobar1 = sqrt(rhobar * rhobar + damp * damp);
    const double rho = sqrt(rhobar1 * rhobar1 + beta * beta);
    if (!(rhobar1 > 0 && isfinite(rho))) {       // far past convergence under force_maxiter: the recurrences have underflowed
        if (!st->istop) st->istop = 6;
        st->itn--;
        st->alpha = alpha;
        st->done = 1;
        st->halt = 1;
        return;
    }
    const double cs1 = rhobar / rhobar1, sn1 = damp / rhobar1;
    const double psi = sn1 * phibar;
    phibar = cs1 * phibar;
    const double cs = rhobar1 / rho, sn = beta / rho;
    const double theta = sn * alpha;
    rhobar = -cs * alpha;
    const double phi = cs * phibar;
    phibar = sn * phibar;
    const double tau = sn * phi;
    const double t1 = phi / rho, t2 = -theta / rho;
    st->ddnorm += (st->wnorm / rho) * (st->wnorm / rho);
    st->coef_xw[0] = cv;
    st->coef_xw[1] = t1;
    st->coef_xw[2] = t2;
    const double delta = st->sn2 * rho, gambar = -st->cs2 * rho, rhs = phi - delta * st->z, zbar = rhs / gambar;
    const double xnorm = sqrt(st->xxnorm + zbar * zbar);
    const double gamma = sqrt(gambar * gambar + theta * theta);
    st->cs2 = gambar / gamma;
    st->sn2 = theta / gamma;
    st->z = rhs / gamma;
    st->xxnorm += st->z * st->z;
    const double anorm = st->anorm;
    const double acond = anorm * sqrt(st->ddnorm);
    const double res1 = phibar * phibar;
    st->res2 += psi * psi;
    const double rnorm = sqrt(res1 + st->res2);
    const double arnorm = alpha * fabs(tau);
    const double r1sq = rnorm * rnorm - damp * damp * st->xxnorm;
    const double r1norm = sqrt(fabs(r1sq)) * (r1sq >= 0 ? 1.0 : -1.0);
    const int itn = st->itn;
    if (history) { history[2 * (itn - 1)] = r1norm; history[2 * (itn - 1) + 1] = arnorm; }
    const double eps = 2.220446049250313e-16, bnorm = st->bnorm;
    const double test1 = bnorm > 0 ? rnorm / bnorm : 0.0;
    const double test2 = rnorm > 0 ? arnorm / (anorm * rnorm + eps) : 0.0;
    const double test3 = 1.0 / (acond + eps);
    const double t1_ = bnorm > 0 ? test1 / (1 + anorm * xnorm / bnorm) : 0.0;
    const double rtol = bnorm > 0 ? st->btol + st->atol * anorm * xnorm / bnorm : 0.0;
    int istop = st->istop;
    if (itn >= st->maxiter) istop = 7;
    if (1 + test3 <= 1) istop = 6;
    if (1 + test2 <= 1) istop = 5;
    if (1 + t1_ <= 1) istop = 4;
    if (test3 <= st->ctol) istop = 3;
    if (test2 <= st->atol) istop = 2;
    if (test1 <= rtol) istop = 1;
    st->istop = istop;
    st->alpha = alpha;
    st->rhobar = rhobar;
    st->phibar = phibar;
    st->r1norm = r1norm;
    st->r2norm = rnorm;
    st->acond = acond;
    st->arnorm = arnorm;
    st->xnorm = xnorm;
    st->coef_step[0] = 1.0;
    st->coef_step[1] = -alpha / beta;
    if (istop && !(st->force && itn < st->maxiter && alpha > 0 && beta > 0)) {
        st->pending = 1;                         // x and w of THIS iteration are still updated
        st->halt = 1;
    }
}

__global__ void k_lsqr_s1(LsqrDev *st, const double *__restrict__ normsq, const double *__restrict__ slot_w)
{
    if (!st->done) lsqr_s1(st, *normsq, slot_w);
}
__global__ void k_lsqr_s2(LsqrDev *st, const double *__restrict__ slot_v, double *__restrict__ history)
{
    if (!st->done) lsqr_s2(st, *slot_v, history);
}
__global__ void k_lsqr_s3(LsqrDev *st)
{
    if (st->pending) st->done = 1;
}

// the same three steps FUSED into the single-workgroup folds that precede them (k_sum_partials / k_lsqr_fold: the same strided sums
// and the same tree, so the same bits): an iteration is then five graph nodes instead of ten, and the chain of dependent small
// kernels is what a small operator's iteration costs
template <int BLK> __device__ inline double wg_sum_value(double v)
{
    __shared__ double sm[BLK / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = sm[0];
#pragma unroll
    for (int w = 1; w < BLK / 64; w++) r += sm[w];
    return r;                                    // (valid in every lane)
}
// (round 4: FIVE nodes.  The fold of ||w||^2 that closed an iteration only fed the NEXT iteration's first scalar step, and its
// "pending -> done" can wait until then as long as the step in between does not run: the step looks at `halt`, this fold finishes
// the solve and sums the x / w kernel's partials itself -- same strided sums, same tree, same bits as the separate fold.)
__global__ __launch_bounds__(256) void k_lsqr_fold_s1(const double *__restrict__ partials, int64_t nparts, LsqrDev *st, const double *__restrict__ parts_w, int nparts_w,
                                                      double *__restrict__ slot_w)
{
    if (st->done) return;
    if (st->pending) {
        if (threadIdx.x == 0) st->done = 1;
        return;
    }
    double w = 0.0;
    for (int i = threadIdx.x; i < nparts_w; i += 256) w += parts_w[i];
    const double rw = wg_sum_value<256>(w);
    __syncthreads();                             // (wg_sum_value's scratch is used again)
    double v = 0.0;
    for (int64_t i = threadIdx.x; i < nparts; i += 256) v += partials[i];
    const double r = wg_sum_value<256>(v);
    if (threadIdx.x == 0) {
        *slot_w = rw;
        lsqr_s1(st, 0.0 + r, slot_w);            // (k_sum_partials adds its sum to a zeroed accumulator)
    }
}
__global__ __launch_bounds__(256) void k_lsqr_fold_s2(const double *__restrict__ partials, int nparts, LsqrDev *st, double *__restrict__ history)
{
    if (st->done) return;
    double v = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) v += partials[i];
    const double r = wg_sum_value<256>(v);
    if (threadIdx.x == 0) lsqr_s2(st, r, history);
}

template <typename S>
__global__ __launch_bounds__(256) void k_lsqr_vhat_dev(S *v, const S *__restrict__ atu, int64_t n, const LsqrDev *__restrict__ st, double *__restrict__ partials)
{
    if (st->done || st->skipv) return;
    const S c0 = (S)st->coef_vhat[0], c1 = (S)st->coef_vhat[1];
    double nrm = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const S a = c0 * atu[i], b = c1 * v[i];
        const S r = a + b;
        v[i] = r;
        nrm += (double)r * (double)r;
    }
    wg_sum_to<256>(nrm, partials + blockIdx.x);
}

template <typename S>
__global__ __launch_bounds__(256) void k_lsqr_xw_dev(S *v, S *x, S *w, int64_t n, const LsqrDev *__restrict__ st, double *__restrict__ partials)
{
    if (st->done) return;
    const S cv = (S)st->coef_xw[0], t1 = (S)st->coef_xw[1], t2 = (S)st->coef_xw[2];
    double nrm = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const S vn = cv * v[i];
        const S wo = w[i];
        const S xa = (S)1 * x[i], xb = t1 * wo;
        const S wa = (S)1 * vn, wb = t2 * wo;
        const S wn = wa + wb;
        v[i] = vn;
        x[i] = xa + xb;
        w[i] = wn;
        nrm += (double)wn * (double)wn;
    }
    wg_sum_to<256>(nrm, partials + blockIdx.x);
}

// WHO sums across shards of the operator:
//   none   one context holds all rows (jh_lsqr_solve);
//   ranks  op/u are THIS rank's block rows of a row-partitioned operator and the exchange runs over the communicator of
//          jh_comm_init_rank -- the CALLER says so (jh_lsqr_solve_partitioned); a communicator merely being alive never turns a
//          rank-local solve into a collective one;
//   team   this process holds M shards, one per member context of a single-process team (jh_comm_init_all): the members'
//          kernels are enqueued one after the other (they return after enqueue, so the GPUs run concurrently), the ranged
//          all-reduces of a range are issued as one group, scalars are added on the host from the members' partials, and the
//          domain-side updates run redundantly on every member with the SAME host scalars, so the replicas stay bit-identical.
enum class Exch { none, ranks, team };

static int lsqr_impl(const int M, const jh_blockop *const *ops, jh_bvec *const *us, jh_bvec *const *xs, int use_x0, double damp, double atol,
                     double btol, double conlim, int maxiter, int force_maxiter, jh_lsqr_result *res, double *history, const Exch ex)
{
    JH_REQUIRE(ops && us && xs && res && M >= 1, "jh_lsqr_solve: null argument");
    JH_REQUIRE(maxiter >= 0, "jh_lsqr_solve: maxiter must be >= 0");
    for (int k = 0; k < M; k++) JH_REQUIRE(ops[k] && us[k] && xs[k], "jh_lsqr_solve: null argument (member %d)", k);
    auto use = [&](int k) { return jh_enter(ops[k], us[k], xs[k]); };
    JH_TRY(use(0));
    int64_t nb = 0, n = 0;
    int dtype = 0;
    JH_TRY(jh_bvec_info(xs[0], &nb, &n, &dtype, nullptr));
    for (int k = 0; k < M; k++) {
        JH_REQUIRE(xs[k]->length == n && xs[k]->dtype == dtype, "jh_lsqr_solve: member %d's x differs in length or element type", k);
        // before anything is touched: the caller can still take another path.  Rows off the 16-byte pack grid are fine (jh_blockop_tall_step_ok): the
        // pipelined exchange cuts the DOMAIN at 16-byte bounds, and the last range may end with the vector
        if (!jh_blockop_tall_step_ok(ops[k], us[k]->data, xs[k]->data))
            return jh_fail(JH_ERR_UNSUPPORTED, "jh_lsqr_solve: needs a tall operator of >= 2 equal elementwise rows");
    }
    std::vector<Tmp> t((size_t)M);
    const int64_t len1[1] = {n};
    const int64_t ns_dom = n * (jh_dtype_complex(dtype) ? 2 : 1);
    int grid = (int)((ns_dom + 255) / 256 < 4096 ? (ns_dom + 255) / 256 : 4096);
    if (grid < 1) grid = 1;
    for (int k = 0; k < M; k++) {
        JH_TRY(use(k));
        JH_TRY(jh_bvec_create(1, len1, dtype, &t[k].v));
        JH_TRY(jh_bvec_create(1, len1, dtype, &t[k].w));
        JH_TRY(jh_bvec_create(1, len1, dtype, &t[k].atu));
        JH_CHECK_HIP(jh_device_malloc(jh_ctx().device, (void **)&t[k].parts, sizeof(double) * (2 * (size_t)grid + 2)));
    }
    auto parts_v = [&](int k) { return t[k].parts; };
    auto parts_w = [&](int k) { return t[k].parts + grid; };
    double *slot_v = t[0].parts + 2 * grid, *slot_w = slot_v + 1;        // member 0 folds and reports the domain-side norms
    JH_TRY(use(0));
    jh_context &c = jh_ctx();                                            // member 0's context: its stream carries the read-backs
    *res = jh_lsqr_result{};
    int64_t chunk = (n + 3) / 4;                                          // exchange ranges: 4, on 64 KiB boundaries
    chunk = (chunk + 16383) / 16384 * 16384;

    // sum of one scalar per member, then over the ranks
    auto global_sum = [&](const std::vector<double> &locals, double *out) -> int {
        double s = 0.0;
        for (double v : locals) s += v;
        *out = s;
        if (ex == Exch::ranks) JH_TRY(jh_comm_allreduce_scalars(out, 1, 0));
        return JH_OK;
    };
    // the members' all-reduces of one collective
    auto exchange = [&](auto &&one) -> int {
        if (ex == Exch::none) return JH_OK;
        if (ex == Exch::team) { JH_TRY(use(0)); JH_TRY(jh_comm_group_begin()); }
        int st = JH_OK;
        for (int k = 0; k < M && st == JH_OK; k++) st = one(k);
        if (ex == Exch::team) { (void)use(0); const int st2 = jh_comm_group_end(); if (st == JH_OK) st = st2; }
        return st;
    };
    std::vector<double> locals((size_t)M);

    double s2 = 0.0;
    for (int k = 0; k < M; k++) {
        if (!use_x0) JH_TRY(jh_fill(xs[k], 0.0, 0.0));
        double nrm = 0.0;
        JH_TRY(jh_norm(us[k], 2.0, &nrm));                               // ||b|| (this shard's rows)
        locals[k] = nrm * nrm;
    }
    JH_TRY(global_sum(locals, &s2));
    const double bnorm = std::sqrt(s2);
    double beta = bnorm;
    if (use_x0) {                                                        // u <- b - A x0
        for (int k = 0; k < M; k++) JH_TRY(jh_blockop_mul_axpby(ops[k], us[k], xs[k], -1.0, 1.0, &locals[k]));
        JH_TRY(global_sum(locals, &s2));
        beta = std::sqrt(s2);
    }
    double alpha = 0.0;
    if (beta > 0) {                                                      // v = A'u / beta
        for (int k = 0; k < M; k++) JH_TRY(jh_blockop_mul_adj(ops[k], t[k].atu, us[k]));
        JH_TRY(exchange([&](int k) { return jh_comm_allreduce_sum(t[k].atu); }));
        for (int k = 0; k < M; k++) JH_TRY(lincomb1(t[k].v, 1.0 / beta, t[k].atu));
        JH_TRY(jh_norm(t[0].v, 2.0, &alpha));                            // replicas are identical: member 0 speaks for all
    } else {
        for (int k = 0; k < M; k++) JH_TRY(jh_copy(t[k].v, xs[k]));
    }
    for (int k = 0; k < M; k++) {
        if (alpha > 0) JH_TRY(lincomb1(t[k].v, 1.0 / alpha, t[k].v));
        JH_TRY(jh_copy(t[k].w, t[k].v));
    }
    double wnorm = 0.0;                                                  // ||w_k||, needed one iteration after w_k is written
    JH_TRY(jh_norm(t[0].w, 2.0, &wnorm));
    bool wnorm_pending = false;                                          // the value is in flight to member 0's red_host[5]

    double rhobar = alpha, phibar = beta, rnorm = beta, r1norm = beta, r2norm = beta, arnorm = alpha * beta;
    double anorm = 0, acond = 0, ddnorm = 0, res2 = 0, xnorm = 0, xxnorm = 0, z = 0, cs2 = -1.0, sn2 = 0.0;
    int itn = 0, istop = 0;
    const double eps = 2.220446049250313e-16, ctol = conlim > 0 ? 1.0 / conlim : 0.0;
    if (arnorm != 0) {
        while (itn < maxiter) {
            itn++;
            // ---- bidiagonalisation in one pass:  beta*u = A v - alpha*u ;  alpha*v = A'u - beta*v
            const double beta_prev = beta;
            if (ex != Exch::none) {
                // every shard's rows in 4 element ranges: the all-reduce of a finished range of A'u runs on the exchange stream while
                // the kernel of the next range computes; ||u||^2 accumulates on the device and is summed behind the last range --
                // ONE host synchronisation per shard for the whole distributed step
                for (int k = 0; k < M; k++) { JH_TRY(use(k)); JH_TRY(jh_normsq_reset()); }
                for (int64_t lo = 0; lo < n; lo += chunk) {
                    const int64_t cnt = lo + chunk < n ? chunk : n - lo;
                    for (int k = 0; k < M; k++)
                        JH_TRY(jh_blockop_bidiag_step_range(ops[k], us[k], t[k].v, t[k].atu, 1.0, -alpha / beta_prev, lo, cnt, nullptr));
                    JH_TRY(exchange([&](int k) { return jh_comm_allreduce_sum_range(t[k].atu, lo, cnt); }));
                }
                if (ex == Exch::ranks) {
                    JH_TRY(jh_comm_allreduce_normsq(&s2));
                } else {                                                 // team: the host adds the members' accumulators
                    s2 = 0.0;
                    for (int k = M - 1; k >= 0; k--) {                   // member 0 last: its synchronisation also lands wnorm (below)
                        JH_TRY(use(k));
                        JH_TRY(jh_comm_join());
                        double part = 0.0;
                        JH_TRY(jh_normsq_read(&part));
                        s2 += part;
                    }
                }
            } else {
                JH_TRY(jh_blockop_bidiag_step(ops[0], us[0], t[0].v, t[0].atu, 1.0, -alpha / beta_prev, &s2));
            }
            beta = std::sqrt(s2);
            if (wnorm_pending) {                                         // the step's read-back synchronised member 0's stream: it has landed
                wnorm = std::sqrt(c.red_host[5]);
                wnorm_pending = false;
            }
            double cv = 1.0;                                             // normalisation of v, applied by the x/w kernel below
            if (beta > 0) {
                anorm = std::sqrt(anorm * anorm + alpha * alpha + beta * beta + damp * damp);
                for (int k = M - 1; k >= 0; k--) {                       // v <- A'(u_hat)/beta - beta v, ||v||^2 (member 0 last: it reports)
                    JH_TRY(use(k));
                    JH_TRY(launch_vhat(dtype, t[k].v, t[k].atu, 1.0 / beta, -beta, parts_v(k), grid));
                }
                hipLaunchKernelGGL(k_lsqr_fold, dim3(1), dim3(256), 0, c.stream, parts_v(0), grid, slot_v);
                JH_CHECK_HIP(hipGetLastError());
                JH_CHECK_HIP(hipMemcpyAsync(c.red_host + 4, slot_v, sizeof(double), hipMemcpyDeviceToHost, c.stream));
                JH_CHECK_HIP(hipStreamSynchronize(c.stream));
                alpha = std::sqrt(c.red_host[4]);
                if (alpha > 0) cv = 1.0 / alpha;
            }
            // ---- eliminate the damping parameter, then the plane rotation
            const double rhobar1 = std::sqrt(rhobar * rhobar + damp * damp);
            const double rho = std::sqrt(rhobar1 * rhobar1 + beta * beta);
            if (!(rhobar1 > 0 && std::isfinite(rho))) {                  // only reachable with force_maxiter, far past convergence: the recurrences
                if (!istop) istop = 6;                                   // have underflowed; x is left at its last finite update
                itn--;
                break;
            }
            const double cs1 = rhobar / rhobar1, sn1 = damp / rhobar1;
            const double psi = sn1 * phibar;
            phibar = cs1 * phibar;
            const double cs = rhobar1 / rho, sn = beta / rho;
            const double theta = sn * alpha;
            rhobar = -cs * alpha;
            const double phi = cs * phibar;
            phibar = sn * phibar;
            const double tau = sn * phi;
            // ---- update x and w
            const double t1 = phi / rho, t2 = -theta / rho;
            ddnorm += (wnorm / rho) * (wnorm / rho);
            for (int k = M - 1; k >= 0; k--) {                           // v normalised; x += t1 w; w = v + t2 w; ||w||^2
                JH_TRY(use(k));
                JH_TRY(launch_xw(dtype, t[k].v, xs[k], t[k].w, cv, t1, t2, parts_w(k), grid));
            }
            hipLaunchKernelGGL(k_lsqr_fold, dim3(1), dim3(256), 0, c.stream, parts_w(0), grid, slot_w);
            JH_CHECK_HIP(hipGetLastError());
            JH_CHECK_HIP(hipMemcpyAsync(c.red_host + 5, slot_w, sizeof(double), hipMemcpyDeviceToHost, c.stream));   // read after the next sync
            wnorm_pending = true;
            // ---- norms for the stopping rules
            const double delta = sn2 * rho, gambar = -cs2 * rho, rhs = phi - delta * z, zbar = rhs / gambar;
            xnorm = std::sqrt(xxnorm + zbar * zbar);
            const double gamma = std::sqrt(gambar * gambar + theta * theta);
            cs2 = gambar / gamma;
            sn2 = theta / gamma;
            z = rhs / gamma;
            xxnorm += z * z;
            acond = anorm * std::sqrt(ddnorm);
            const double res1 = phibar * phibar;
            res2 += psi * psi;
            rnorm = std::sqrt(res1 + res2);
            arnorm = alpha * std::fabs(tau);
            const double r1sq = rnorm * rnorm - damp * damp * xxnorm;
            r1norm = std::sqrt(std::fabs(r1sq)) * (r1sq >= 0 ? 1.0 : -1.0);
            r2norm = rnorm;
            if (history) { history[2 * (itn - 1)] = r1norm; history[2 * (itn - 1) + 1] = arnorm; }
            const double test1 = bnorm > 0 ? rnorm / bnorm : 0.0;
            const double test2 = rnorm > 0 ? arnorm / (anorm * rnorm + eps) : 0.0;
            const double test3 = 1.0 / (acond + eps);
            const double t1_ = bnorm > 0 ? test1 / (1 + anorm * xnorm / bnorm) : 0.0;
            const double rtol = bnorm > 0 ? btol + atol * anorm * xnorm / bnorm : 0.0;
            if (itn >= maxiter) istop = 7;
            if (1 + test3 <= 1) istop = 6;
            if (1 + test2 <= 1) istop = 5;
            if (1 + t1_ <= 1) istop = 4;
            if (test3 <= ctol) istop = 3;
            if (test2 <= atol) istop = 2;
            if (test1 <= rtol) istop = 1;
            if (istop && !(force_maxiter && itn < maxiter && alpha > 0 && beta > 0)) break;
        }
    }
    res->istop = istop;
    res->itn = itn;
    res->r1norm = r1norm;
    res->r2norm = r2norm;
    res->anorm = anorm;
    res->acond = acond;
    res->arnorm = arnorm;
    res->xnorm = xnorm;
    for (int k = 0; k < M; k++) {                                        // the temporaries die here
        jh_context *ck = jh_ctx_by_id(ops[k]->ctx);
        if (ck) { JH_TRY(use(k)); JH_CHECK_HIP(hipStreamSynchronize(ck->stream)); }
    }
    return JH_OK;
}

// The graph-replayed loop for ONE small operator in one context (see LsqrDev above).  *took = false: not eligible, nothing was
// touched, the caller runs lsqr_impl.
static int lsqr_graph_impl(const jh_blockop *op, jh_bvec *u, jh_bvec *x, int use_x0, double damp, double atol, double btol, double conlim,
                           int maxiter, int force_maxiter, jh_lsqr_result *res, double *history, bool *took)
{
    *took = false;
    JH_TRY(jh_enter(op, u, x));
    jh_context &c = jh_ctx();
    c.last_lsqr_graph = 0;
    if (!c.lsqr_graph || !op || !u || !x || !res || maxiter < 1) return JH_OK;
    if (!jh_blockop_tall_step_ok(op, u->data, x->data)) return JH_OK;       // (rows off the 16-byte pack grid too: the plain step's MIXED instantiations)
    // one plain launch per step: no split walk (pick_adj_parts), no row-chunked launches; and only where launches and host
    // round trips matter: a pass over the operator and u below 1 GiB
    if (!(jh_bidiag_step_parts(op) == 1 && c.adj_rows_per_launch == 0)) return JH_OK;
    const int dtype = x->dtype;
    const int64_t n = x->length;
    // (late round 4: up to 2 GiB per pass when the blocks are below 16 MiB -- the sizes whose step has one way to walk, so nothing is lost by not measuring
    // it per operator: 64 x 128^3 317 -> 305 us per iteration; knob lsqr_graph = 2: any size)
    const double pass_bytes = 3.0 * (double)op->nrow * (double)n * (double)jh_dtype_size(dtype);
    if (c.lsqr_graph < 2 && !(pass_bytes < (double)(1ull << 30) || (pass_bytes < (double)(2ull << 30) && (double)n * (double)jh_dtype_size(dtype) < (double)(16u << 20))))
        return JH_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c.stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return JH_OK;
    *took = true;

    Tmp t;
    const int64_t len1[1] = {n};
    JH_TRY(jh_bvec_create(1, len1, dtype, &t.v));
    JH_TRY(jh_bvec_create(1, len1, dtype, &t.w));
    JH_TRY(jh_bvec_create(1, len1, dtype, &t.atu));
    const int64_t ns_dom = n * (jh_dtype_complex(dtype) ? 2 : 1);
    int grid = (int)((ns_dom + 255) / 256 < 4096 ? (ns_dom + 255) / 256 : 4096);
    if (grid < 1) grid = 1;
    JH_CHECK_HIP(jh_device_malloc(jh_ctx().device, (void **)&t.parts, sizeof(double) * (2 * (size_t)grid + 2)));
    double *parts_v = t.parts, *parts_w = t.parts + grid, *slot_v = t.parts + 2 * grid, *slot_w = slot_v + 1;
    *res = jh_lsqr_result{};

    // ---- the start-up of lsqr_impl, one shard, no exchange
    if (!use_x0) JH_TRY(jh_fill(x, 0.0, 0.0));
    double nrm = 0.0;
    JH_TRY(jh_norm(u, 2.0, &nrm));
    const double bnorm = std::sqrt(nrm * nrm);
    double beta = bnorm;
    if (use_x0) {
        double local = 0.0;
        JH_TRY(jh_blockop_mul_axpby(op, u, x, -1.0, 1.0, &local));
        beta = std::sqrt(local);
    }
    double alpha = 0.0;
    if (beta > 0) {
        JH_TRY(jh_blockop_mul_adj(op, t.atu, u));
        JH_TRY(lincomb1(t.v, 1.0 / beta, t.atu));
        JH_TRY(jh_norm(t.v, 2.0, &alpha));
    } else {
        JH_TRY(jh_copy(t.v, x));
    }
    if (alpha > 0) JH_TRY(lincomb1(t.v, 1.0 / alpha, t.v));
    JH_TRY(jh_copy(t.w, t.v));
    double wnorm = 0.0;
    JH_TRY(jh_norm(t.w, 2.0, &wnorm));
    const double arnorm0 = alpha * beta;
    res->r1norm = res->r2norm = beta;
    res->arnorm = arnorm0;
    if (arnorm0 == 0) return JH_OK;                                       // (istop = itn = 0, like lsqr_impl)

    // ---- device state
    struct Dev {
        LsqrDev *st = nullptr;
        double *hist = nullptr;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        ~Dev()
        {
            if (exec) (void)hipGraphExecDestroy(exec);
            if (graph) (void)hipGraphDestroy(graph);
            if (st) (void)hipFree(st);
            if (hist) (void)hipFree(hist);
        }
    } d;
    JH_CHECK_HIP(jh_device_malloc(c.device, (void **)&d.st, sizeof(LsqrDev)));
    JH_CHECK_HIP(jh_device_malloc(c.device, (void **)&d.hist, sizeof(double) * 2 * (size_t)maxiter));
    LsqrDev h{};
    h.alpha = alpha;
    h.beta = beta;
    h.rhobar = alpha;
    h.phibar = beta;
    h.cs2 = -1.0;
    h.bnorm = bnorm;
    h.wnorm = wnorm;
    h.damp = damp;
    h.atol = atol;
    h.btol = btol;
    h.ctol = conlim > 0 ? 1.0 / conlim : 0.0;
    h.r1norm = h.r2norm = beta;
    h.arnorm = arnorm0;
    h.coef_step[0] = 1.0;
    h.coef_step[1] = -alpha / beta;
    h.maxiter = maxiter;
    h.force = force_maxiter ? 1 : 0;
    JH_CHECK_HIP(hipMemcpyAsync(d.st, &h, sizeof(h), hipMemcpyHostToDevice, c.stream));
    JH_CHECK_HIP(hipStreamSynchronize(c.stream));                        // `h` is a stack object

    const bool f64 = (dtype == JH_F64 || dtype == JH_C64);
    // one iteration, enqueued on the context's stream: eagerly the first time (workspaces get their size), captured the second
    auto iteration = [&]() -> int {
        JH_TRY(jh_normsq_reset());
        c.step_coef_dev = d.st->coef_step;
        c.step_done_dev = &d.st->done;
        const int st_ = jh_blockop_bidiag_step_range(op, u, t.v, t.atu, 1.0, 0.0, 0, n, nullptr);     // (alpha, beta) come from the device
        c.step_coef_dev = nullptr;
        c.step_done_dev = nullptr;
        JH_TRY(st_);
        hipLaunchKernelGGL(k_lsqr_s1, dim3(1), dim3(1), 0, c.stream, d.st, (const double *)(c.red_dev + JH_NORMSQ_SLOT), (const double *)slot_w);
        if (f64) hipLaunchKernelGGL((k_lsqr_vhat_dev<double>), dim3(grid), dim3(256), 0, c.stream, (double *)t.v->data, (const double *)t.atu->data, ns_dom, (const LsqrDev *)d.st, parts_v);
        else hipLaunchKernelGGL((k_lsqr_vhat_dev<float>), dim3(grid), dim3(256), 0, c.stream, (float *)t.v->data, (const float *)t.atu->data, ns_dom, (const LsqrDev *)d.st, parts_v);
        hipLaunchKernelGGL(k_lsqr_fold, dim3(1), dim3(256), 0, c.stream, (const double *)parts_v, grid, slot_v);
        hipLaunchKernelGGL(k_lsqr_s2, dim3(1), dim3(1), 0, c.stream, d.st, (const double *)slot_v, d.hist);
        if (f64) hipLaunchKernelGGL((k_lsqr_xw_dev<double>), dim3(grid), dim3(256), 0, c.stream, (double *)t.v->data, (double *)x->data, (double *)t.w->data, ns_dom, (const LsqrDev *)d.st, parts_w);
        else hipLaunchKernelGGL((k_lsqr_xw_dev<float>), dim3(grid), dim3(256), 0, c.stream, (float *)t.v->data, (float *)x->data, (float *)t.w->data, ns_dom, (const LsqrDev *)d.st, parts_w);
        hipLaunchKernelGGL(k_lsqr_fold, dim3(1), dim3(256), 0, c.stream, (const double *)parts_w, grid, slot_w);
        hipLaunchKernelGGL(k_lsqr_s3, dim3(1), dim3(1), 0, c.stream, d.st);
        JH_CHECK_HIP(hipGetLastError());
        return JH_OK;
    };
    // the same iteration with the scalar steps fused into the folds (five nodes): possible when the step's partial sums fit ONE
    // fold launch (k_sum_partials' single-level form, <= 8192 workgroups) -- known after the eager iteration
    auto iteration_fused = [&]() -> int {
        c.step_coef_dev = d.st->coef_step;
        c.step_done_dev = &d.st->halt;
        c.step_skip_fold = 1;
        const int st_ = jh_blockop_bidiag_step_range(op, u, t.v, t.atu, 1.0, 0.0, 0, n, nullptr);
        c.step_coef_dev = nullptr;
        c.step_done_dev = nullptr;
        c.step_skip_fold = 0;
        JH_TRY(st_);
        hipLaunchKernelGGL(k_lsqr_fold_s1, dim3(1), dim3(256), 0, c.stream, (const double *)c.part_dev, c.last_step_parts, d.st, (const double *)parts_w, grid, slot_w);
        if (f64) hipLaunchKernelGGL((k_lsqr_vhat_dev<double>), dim3(grid), dim3(256), 0, c.stream, (double *)t.v->data, (const double *)t.atu->data, ns_dom, (const LsqrDev *)d.st, parts_v);
        else hipLaunchKernelGGL((k_lsqr_vhat_dev<float>), dim3(grid), dim3(256), 0, c.stream, (float *)t.v->data, (const float *)t.atu->data, ns_dom, (const LsqrDev *)d.st, parts_v);
        hipLaunchKernelGGL(k_lsqr_fold_s2, dim3(1), dim3(256), 0, c.stream, (const double *)parts_v, grid, d.st, d.hist);
        if (f64) hipLaunchKernelGGL((k_lsqr_xw_dev<double>), dim3(grid), dim3(256), 0, c.stream, (double *)t.v->data, (double *)x->data, (double *)t.w->data, ns_dom, (const LsqrDev *)d.st, parts_w);
        else hipLaunchKernelGGL((k_lsqr_xw_dev<float>), dim3(grid), dim3(256), 0, c.stream, (float *)t.v->data, (float *)x->data, (float *)t.w->data, ns_dom, (const LsqrDev *)d.st, parts_w);
        JH_CHECK_HIP(hipGetLastError());
        return JH_OK;
    };
    struct Flags { int itn, istop, done, pending; } fl{};
    auto read_flags = [&]() -> int {
        JH_CHECK_HIP(hipMemcpyAsync(&fl, &d.st->itn, sizeof(fl), hipMemcpyDeviceToHost, c.stream));
        JH_CHECK_HIP(hipStreamSynchronize(c.stream));
        return JH_OK;
    };
    JH_TRY(iteration());                                                  // iteration 1, eager
    JH_TRY(read_flags());
    int64_t replays = 0;
    if (!fl.done) {
        const bool fused = c.last_step_parts > 0 && c.last_step_parts <= 8192;
        // round 4: EIGHT iterations per graph (finished solves replay as no-ops): one hipGraphLaunch costs the host 10-16 us
        const int per_graph = maxiter < 8 ? maxiter : 8;
        JH_CHECK_HIP(hipStreamBeginCapture(c.stream, hipStreamCaptureModeRelaxed));   // other host threads (other contexts) keep working meanwhile
        int st_ = JH_OK;
        for (int k = 0; k < per_graph && st_ == JH_OK; k++) st_ = fused ? iteration_fused() : iteration();
        hipError_t e = hipStreamEndCapture(c.stream, &d.graph);
        if (st_ != JH_OK) return st_;
        JH_CHECK_HIP(e);
        JH_CHECK_HIP(hipGraphInstantiate(&d.exec, d.graph, nullptr, nullptr, 0));
        while (!(fl.done || (fused && fl.pending))) {                     // two graphs between two looks at the flags (five nodes: `pending` ends the solve)
            for (int k = 0; k < 2; k++) JH_CHECK_HIP(hipGraphLaunch(d.exec, c.stream));
            replays += 2;
            JH_TRY(read_flags());
        }
    }
    c.last_lsqr_graph = replays;
    JH_CHECK_HIP(hipMemcpyAsync(&h, d.st, sizeof(h), hipMemcpyDeviceToHost, c.stream));
    JH_CHECK_HIP(hipStreamSynchronize(c.stream));
    if (history && h.itn > 0) {                                           // (on the context's stream: a legacy-stream copy would disturb another thread's capture)
        JH_CHECK_HIP(hipMemcpyAsync(history, d.hist, sizeof(double) * 2 * (size_t)h.itn, hipMemcpyDeviceToHost, c.stream));
        JH_CHECK_HIP(hipStreamSynchronize(c.stream));
    }
    res->istop = h.istop;
    res->itn = h.itn;
    res->r1norm = h.r1norm;
    res->r2norm = h.r2norm;
    res->anorm = h.anorm;
    res->acond = h.acond;
    res->arnorm = h.arnorm;
    res->xnorm = h.xnorm;
    return JH_OK;
}

extern "C" int jh_lsqr_solve(const jh_blockop *op, jh_bvec *u, jh_bvec *x, int use_x0, double damp, double atol, double btol, double conlim,
                             int maxiter, int force_maxiter, jh_lsqr_result *res, double *history)
{
    bool took = false;
    JH_TRY(lsqr_graph_impl(op, u, x, use_x0, damp, atol, btol, conlim, maxiter, force_maxiter, res, history, &took));   // small operators
    if (took) return JH_OK;
    return lsqr_impl(1, &op, &u, &x, use_x0, damp, atol, btol, conlim, maxiter, force_maxiter, res, history, Exch::none);
}

extern "C" int jh_lsqr_solve_partitioned(const jh_blockop *op, jh_bvec *u, jh_bvec *x, int use_x0, double damp, double atol, double btol,
                                         double conlim, int maxiter, int force_maxiter, jh_lsqr_result *res, double *history)
{
    JH_TRY(jh_enter(op, u, x));                                         // the communicator is the one of the handles' context
    int nranks = 1, rank = 0, has_comm = 0;
    (void)jh_comm_info(&nranks, &rank);
    (void)jh_comm_exists(&has_comm);
    if (has_comm == 2 && (nranks > 1 || jh_ctx().force_dist))
        return jh_fail(JH_ERR_UNSUPPORTED, "jh_lsqr_solve_partitioned: this context is a member of a single-process team (jh_comm_init_all): "
                       "jh_lsqr_solve_team takes all the members' shards in one call");
    // one rank: nothing to exchange (jh_comm_init_rank is then optional, so a one-GPU run of partitioned host code works);
    // the knob force_dist runs the exchange all the same (validation of the pipelined path with a one-rank communicator)
    return lsqr_impl(1, &op, &u, &x, use_x0, damp, atol, btol, conlim, maxiter, force_maxiter, res, history,
                     (nranks > 1 || (has_comm && jh_ctx().force_dist)) ? Exch::ranks : Exch::none);
}

// ONE process, a team of n member contexts (jh_comm_init_all): member k holds ops[k] (its block rows), us[k] (its rows of the
// right-hand side; overwritten) and xs[k] (its replica of the solution).  The same loop as jh_lsqr_solve_partitioned, the
// members' kernels enqueued one after the other, their ranged all-reduces grouped, scalars added on the host.
extern "C" int jh_lsqr_solve_team(int n, const jh_blockop *const *ops, jh_bvec *const *us, jh_bvec *const *xs, int use_x0, double damp, double atol,
                                  double btol, double conlim, int maxiter, int force_maxiter, jh_lsqr_result *res, double *history)
{
    JH_REQUIRE(n >= 1 && n <= JH_MAX_CTX && ops && us && xs, "jh_lsqr_solve_team: need 1..%d members", JH_MAX_CTX);
    for (int k = 0; k < n; k++) {
        JH_REQUIRE(ops[k] && us[k] && xs[k], "jh_lsqr_solve_team: null handle of member %d", k);
        JH_TRY(jh_enter(ops[k], us[k], xs[k]));
        int nranks = 1, rank = 0, has_comm = 0;
        (void)jh_comm_info(&nranks, &rank);
        (void)jh_comm_exists(&has_comm);
        JH_REQUIRE(has_comm == 2 && nranks == n && rank == k, "jh_lsqr_solve_team: the handles of member %d must live in member %d's context of a team of %d "
                   "(jh_comm_init_all); found %s, rank %d of %d", k, k, n, has_comm == 2 ? "a team" : "no team", rank, nranks);
    }
    return lsqr_impl(n, ops, us, xs, use_x0, damp, atol, btol, conlim, maxiter, force_maxiter, res, history, Exch::team);
}

// ------------------------------------------------------------------ CGLS (round 3; SURVEY section 8 f-1 "LSQR/CGLS") ------------------------
// Conjugate gradients on the normal equations (Hestenes & Stiefel 1952; Bjorck, "Numerical Methods for Least Squares Problems",
// 1996, algorithm 7.4.1 "CGLS"), min ||A x - b||^2 + damp^2 ||x||^2, over the same shards and exchange modes as lsqr_impl.  The
// reference ships no solver (src/Jets.jl:1143-1152 points at IterativeSolvers.jl, un-vendored), so this is the published recurrence,
// checked against the fp64 CPU CGLS of oracle/cgls_ref.py.
//
//   r = b - A x0 ;  s = A'r - damp^2 x ;  p = s ;  gamma = ||s||^2
//   repeat:  delta = ||A p||^2 + damp^2 ||p||^2 ;  alpha = gamma / delta ;  x += alpha p ;  r -= alpha A p
//            s = A'r - damp^2 x ;  gamma' = ||s||^2 ;  p = s + (gamma'/gamma) p
//
// The textbook loop keeps q = A p in a range-sized vector (64 GiB at the headline size) and moves 7 N n s bytes per iteration
// through the two fused halves (q = A p with ||q||^2: 2 N n s; r -= alpha q: 3 N n s; s = A'r: 2 N n s).  Here an iteration is TWO
// passes and no q:   ||A p||^2 = <p, A'A p>  from the fused normal operator (jh_blockop_normal_mul: reads the coefficients only,
// N n s bytes; the inner product is a domain-sized fp64 reduction), then ONE pass of the Golub-Kahan step kernel
// (jh_blockop_bidiag_step with (alpha, beta) = (-alpha_k, 1)):  r <- r - alpha_k A p,  ||r||^2  and  A'r  together, 3 N n s bytes.
// 4 N n s bytes per iteration against LSQR's 3 N n s.  Partitioned: the first pass needs a SCALAR exchange only (every shard's
// <p, A_k'A_k p>), the second is LSQR's pipelined step (ranged all-reduces of A'r under the kernels, one host synchronisation).
static int cgls_impl(const int M, const jh_blockop *const *ops, jh_bvec *const *us, jh_bvec *const *xs, int use_x0, double damp, double atol,
                     double btol, int maxiter, int force_maxiter, jh_lsqr_result *res, double *history, const Exch ex)
{
    JH_REQUIRE(ops && us && xs && res && M >= 1, "jh_cgls_solve: null argument");
    JH_REQUIRE(maxiter >= 0, "jh_cgls_solve: maxiter must be >= 0");
    for (int k = 0; k < M; k++) JH_REQUIRE(ops[k] && us[k] && xs[k], "jh_cgls_solve: null argument (member %d)", k);
    auto use = [&](int k) { return jh_enter(ops[k], us[k], xs[k]); };
    JH_TRY(use(0));
    int64_t nb = 0, n = 0;
    int dtype = 0;
    JH_TRY(jh_bvec_info(xs[0], &nb, &n, &dtype, nullptr));
    for (int k = 0; k < M; k++) {
        JH_REQUIRE(xs[k]->length == n && xs[k]->dtype == dtype, "jh_cgls_solve: member %d's x differs in length or element type", k);
        // before anything is touched (rows off the 16-byte pack grid too, as in lsqr_impl)
        if (!jh_blockop_tall_step_ok(ops[k], us[k]->data, xs[k]->data) || ops[k]->nrow < 2)
            return jh_fail(JH_ERR_UNSUPPORTED, "jh_cgls_solve: needs a tall (>= 2 rows) operator of equal elementwise rows");
    }
    struct Work {                                                         // domain-sized work vectors of one member
        jh_bvec *p = nullptr, *s = nullptr, *y = nullptr;
        ~Work()
        {
            if (p) (void)jh_bvec_destroy(p);
            if (s) (void)jh_bvec_destroy(s);
            if (y) (void)jh_bvec_destroy(y);
        }
    };
    std::vector<Work> t((size_t)M);
    const int64_t len1[1] = {n};
    for (int k = 0; k < M; k++) {
        JH_TRY(use(k));
        JH_TRY(jh_bvec_create(1, len1, dtype, &t[k].p));
        JH_TRY(jh_bvec_create(1, len1, dtype, &t[k].s));
        JH_TRY(jh_bvec_create(1, len1, dtype, &t[k].y));
    }
    *res = jh_lsqr_result{};
    int64_t chunk = (n + 3) / 4;                                          // exchange ranges: 4, on 64 KiB boundaries (as lsqr_impl)
    chunk = (chunk + 16383) / 16384 * 16384;
    auto global_sum = [&](const std::vector<double> &locals, double *out) -> int {
        double sum = 0.0;
        for (double v : locals) sum += v;
        *out = sum;
        if (ex == Exch::ranks) JH_TRY(jh_comm_allreduce_scalars(out, 1, 0));
        return JH_OK;
    };
    auto exchange = [&](auto &&one) -> int {
        if (ex == Exch::none) return JH_OK;
        if (ex == Exch::team) { JH_TRY(use(0)); JH_TRY(jh_comm_group_begin()); }
        int st = JH_OK;
        for (int k = 0; k < M && st == JH_OK; k++) st = one(k);
        if (ex == Exch::team) { (void)use(0); const int st2 = jh_comm_group_end(); if (st == JH_OK) st = st2; }
        return st;
    };
    auto lincomb2 = [&](jh_bvec *dst, double c0, const jh_bvec *x0, double c1, const jh_bvec *x1) {
        const double coef[4] = {c0, 0.0, c1, 0.0};
        const jh_bvec *v[2] = {x0, x1};
        return jh_lincomb(dst, 2, coef, v);
    };
    // s = A'r (already summed over the shards, in t[k].s) - damp^2 x ;  returns ||s||^2 (member 0 speaks for all: replicas are identical)
    auto finish_s = [&](double *gamma) -> int {
        for (int k = 0; k < M; k++)
            if (damp != 0.0) JH_TRY(lincomb2(t[k].s, 1.0, t[k].s, -damp * damp, xs[k]));
        double nrm = 0.0;
        JH_TRY(jh_norm(t[0].s, 2.0, &nrm));
        *gamma = nrm * nrm;
        return JH_OK;
    };
    std::vector<double> locals((size_t)M);
    jh_context *const trace_ctx = jh_ctx_by_id(ops[0]->ctx);              // knob cgls_trace / read-only last_cgls_overlaps live in member 0's context
    if (trace_ctx) trace_ctx->last_cgls_overlaps = -1;

    double s2 = 0.0;
    for (int k = 0; k < M; k++) {
        if (!use_x0) JH_TRY(jh_fill(xs[k], 0.0, 0.0));
        double nrm = 0.0;
        JH_TRY(jh_norm(us[k], 2.0, &nrm));                               // ||b|| (this shard's rows)
        locals[k] = nrm * nrm;
    }
    JH_TRY(global_sum(locals, &s2));
    const double bnorm = std::sqrt(s2);
    double rr = s2;                                                       // ||r||^2
    if (use_x0) {                                                         // r <- b - A x0
        for (int k = 0; k < M; k++) JH_TRY(jh_blockop_mul_axpby(ops[k], us[k], xs[k], -1.0, 1.0, &locals[k]));
        JH_TRY(global_sum(locals, &rr));
    }
    for (int k = 0; k < M; k++) JH_TRY(jh_blockop_mul_adj(ops[k], t[k].s, us[k]));           // s = A'r
    JH_TRY(exchange([&](int k) { return jh_comm_allreduce_sum(t[k].s); }));
    double gamma = 0.0;
    JH_TRY(finish_s(&gamma));
    for (int k = 0; k < M; k++) JH_TRY(jh_copy(t[k].p, t[k].s));
    const double gamma0 = gamma;
    double xnorm = 0.0;
    int itn = 0, istop = 0;
    if (gamma > 0) {
        while (itn < maxiter) {
            itn++;
            // ---- pass 1: delta = <p, A'A p> + damp^2 ||p||^2 (every shard's share of the quadratic form, then a scalar sum)
            // every member's pass and its inner product are ENQUEUED before the first wait: the members' GPUs run side by side (round 3
            // waited for member k's dot before member k + 1's kernel was even launched: M passes one after the other)
            const bool trace = itn == 1 && M > 1 && trace_ctx && trace_ctx->cgls_trace;   // tests: event timestamps of the first iteration's pass 1
            std::vector<hipEvent_t> tev(trace ? (size_t)(2 * M) : 0, nullptr);
            auto stamp = [&](int k, int which) -> int {
                if (!trace) return JH_OK;
                JH_TRY(use(k));
                JH_CHECK_HIP(hipEventCreate(&tev[(size_t)(2 * k + which)]));
                JH_CHECK_HIP(hipEventRecord(tev[(size_t)(2 * k + which)], jh_ctx().stream));
                return JH_OK;
            };
            for (int k = 0; k < M; k++) {
                JH_TRY(stamp(k, 0));
                JH_TRY(jh_blockop_normal_mul(ops[k], t[k].y, t[k].p));
                JH_TRY(jh_dot_begin(t[k].p, t[k].y));
                JH_TRY(stamp(k, 1));
            }
            for (int k = 0; k < M; k++) {
                double re = 0.0, im = 0.0;
                JH_TRY(jh_dot_end(t[k].p, &re, &im));
                locals[k] = re;
            }
            if (trace) {                                                  // member k + 1 began before member k had finished?  (members of ONE device: one clock)
                int64_t overlaps = 0;
                for (int k = 0; k + 1 < M; k++) {
                    float ms = 0.f;                                       // from k's end to (k + 1)'s begin: negative when they overlap
                    if (hipEventElapsedTime(&ms, tev[(size_t)(2 * k + 1)], tev[(size_t)(2 * k + 2)]) == hipSuccess && ms < 0.f) overlaps++;
                }
                (void)hipGetLastError();
                trace_ctx->last_cgls_overlaps = overlaps;
                for (hipEvent_t e : tev)
                    if (e) (void)hipEventDestroy(e);
            }
            double delta = 0.0;
            JH_TRY(global_sum(locals, &delta));
            if (damp != 0.0) {
                double pn = 0.0;
                JH_TRY(jh_norm(t[0].p, 2.0, &pn));
                delta += damp * damp * pn * pn;
            }
            if (!(delta > 0) || !std::isfinite(delta)) {                  // breakdown: p in the null space (or the recurrences have underflowed)
                istop = 6;
                itn--;
                break;
            }
            const double alpha = gamma / delta;
            for (int k = 0; k < M; k++) JH_TRY(lincomb2(xs[k], 1.0, xs[k], alpha, t[k].p));   // x += alpha p
            // ---- pass 2: r <- r - alpha A p, ||r||^2 and A'r in ONE pass over the operator and r
            if (ex != Exch::none) {
                for (int k = 0; k < M; k++) { JH_TRY(use(k)); JH_TRY(jh_normsq_reset()); }
                for (int64_t lo = 0; lo < n; lo += chunk) {
                    const int64_t cnt = lo + chunk < n ? chunk : n - lo;
                    for (int k = 0; k < M; k++) JH_TRY(jh_blockop_bidiag_step_range(ops[k], us[k], t[k].p, t[k].s, -alpha, 1.0, lo, cnt, nullptr));
                    JH_TRY(exchange([&](int k) { return jh_comm_allreduce_sum_range(t[k].s, lo, cnt); }));
                }
                if (ex == Exch::ranks) {
                    JH_TRY(jh_comm_allreduce_normsq(&rr));
                } else {
                    rr = 0.0;
                    for (int k = M - 1; k >= 0; k--) {
                        JH_TRY(use(k));
                        JH_TRY(jh_comm_join());
                        double part = 0.0;
                        JH_TRY(jh_normsq_read(&part));
                        rr += part;
                    }
                }
            } else {
                JH_TRY(jh_blockop_bidiag_step(ops[0], us[0], t[0].p, t[0].s, -alpha, 1.0, &rr));
            }
            double gamma_new = 0.0;
            JH_TRY(finish_s(&gamma_new));
            const double bk = gamma_new / gamma;
            for (int k = 0; k < M; k++) JH_TRY(lincomb2(t[k].p, 1.0, t[k].s, bk, t[k].p));   // p = s + beta p
            gamma = gamma_new;
            const double rnorm = std::sqrt(rr), arnorm = std::sqrt(gamma);
            if (history) { history[2 * (itn - 1)] = rnorm; history[2 * (itn - 1) + 1] = arnorm; }
            if (itn >= maxiter) istop = 7;
            if (arnorm <= atol * std::sqrt(gamma0)) istop = 2;            // ||A'r - damp^2 x|| small against its starting value
            if (rnorm <= btol * bnorm) istop = 1;                         // ||r|| small against ||b||
            if (istop && !(force_maxiter && itn < maxiter && gamma > 0)) break;
        }
    }
    JH_TRY(jh_norm(xs[0], 2.0, &xnorm));
    res->istop = istop;
    res->itn = itn;
    res->r1norm = std::sqrt(rr);
    res->r2norm = std::sqrt(rr + damp * damp * xnorm * xnorm);
    res->anorm = 0.0;                                                     // (LSQR's running estimates have no counterpart here)
    res->acond = 0.0;
    res->arnorm = std::sqrt(gamma);
    res->xnorm = xnorm;
    for (int k = 0; k < M; k++) {                                         // the temporaries die here
        jh_context *ck = jh_ctx_by_id(ops[k]->ctx);
        if (ck) { JH_TRY(use(k)); JH_CHECK_HIP(hipStreamSynchronize(ck->stream)); }
    }
    return JH_OK;
}

// ------------------------------------------------------------------ CGLS and CG on the normal equations for SMALL operators (round 4) -----
// The loops above (cgls_impl, cgnr_impl below) return to the host two or three times per iteration for a scalar and launch about a dozen
// small kernels: for an operator whose fused A'A takes 11 us that is 60 us per iteration (64 x 64^3, profiles/bench_cgnr_sizes_r03.txt).
// Here -- as lsqr_graph_impl does for LSQR -- the scalars of the recurrences are a struct in device memory (jh_cg_dev), the one-thread
// epilogues of two fold kernels update it, every vector kernel reads its coefficients from it and becomes a no-op once the solve has
// finished, so one iteration has fixed launch parameters, is captured ONCE as a hipGraph and replayed:
//   CG on the normal equations, 3 nodes:  k_cg_normal  [p <- s + bk p ; y = (A'A + damp^2) p ; partial <p, y>]      (jh_tall.hip)
//                                         k_cg_xs      [fold of <p, y> -> alpha, breakdown ; x += alpha p ; s -= alpha y ; partial ||s||^2]
//                                         k_cg_fold    [gamma', beta, ||r|| by recurrence, history, stopping rules]
//   CGLS, 5 nodes:  k_cg_normal ; k_cg_fold ; the one-pass step r <- r - alpha A p, A'r, partial ||r||^2 (coefficients from the device) ;
//                   k_cgls_xs [x += alpha p ; s -= damp^2 x ; partial ||s||^2] ; k_cg_fold
// The two scalar updates (cg_s1, cg_s2) are `__host__ __device__` functions: with the knob lsqr_graph = 0 the SAME kernels run eagerly
// and the host applies the same two functions between them (one read-back each) -- the host-driven form of this loop, whose iterates the
// replayed graph reproduces bit for bit (tests/test_gpu_graphs.py).  Operators of 1 GiB and more per pass, partitioned and team solves
// keep cgls_impl / cgnr_impl (their tuned kernels, split walks and pipelined exchange); both forms are checked against the fp64 CPU CGLS.
// which == 1: the sum of `partials` is <p, (A'A + damp^2) p>; which == 2: the sum of `partials` is ||s||^2 and (CGLS) the sum of
// `rparts` the step's ||r||^2 (k_sum_partials' single-level order: `0.0 + r`).  on_device: apply the scalar update here, else leave
// the sums in out[0], out[1] for the host.
__global__ __launch_bounds__(256) void k_cg_fold(const double *__restrict__ partials, int64_t nparts, const double *__restrict__ rparts, int64_t nrparts,
                                                 jh_cg_dev *st, double *__restrict__ history, double *__restrict__ out, int which, int on_device)
{
    // every load is issued before anything depends on one (the state, the flag and the partials travel together: a dependent launch of
    // a few hundred bytes is all latency); a finished solve changes nothing at the end instead of returning at the start
    jh_cg_dev loc;
    if (threadIdx.x == 0) loc = *st;
    const int done = st->done;
    const double v = jh_strided_sum256(partials, nparts), w = rparts ? jh_strided_sum256(rparts, nrparts) : 0.0;
    const double r = wg_sum_value<256>(v);
    double r2 = 0.0;
    if (rparts) {
        __syncthreads();                                                  // (wg_sum_value's staging array is reused)
        r2 = 0.0 + wg_sum_value<256>(w);
    }
    if (threadIdx.x == 0 && !done) {
        if (!on_device) { out[0] = r; out[1] = r2; }
        else {
            if (which == 1) cg_s1(&loc, r);
            else cg_s2(&loc, r, r2, history);
            *st = loc;
        }
    }
}

// fold_n != nullptr (CG on the normal equations, graph form): the fold of the <p, y> partials and the first scalar update are done HERE
// instead of in a launch of their own -- every workgroup adds the same partials in the same order (the same bits everywhere) and forms
// alpha = gamma / <p, y> itself (cg_s1 writes neither gamma nor damp2; what it does write -- itn, delta, alpha, the step coefficients and,
// on a breakdown, istop and done -- no other workgroup's result depends on: on a breakdown every workgroup returns without an update,
// whether it saw `done` or found the breakdown itself); workgroup 0 records the update.  Three graph nodes per iteration instead of
// four: a dependent launch costs 4-5 us even when it does nothing.
template <typename S>
__global__ __launch_bounds__(256) void k_cg_xs(S *__restrict__ x, S *__restrict__ s, const S *__restrict__ p, const S *__restrict__ y, int64_t n,
                                               jh_cg_dev *st, double *__restrict__ partials, const double *__restrict__ fold_n, int64_t nfold)
{
    // the first element's operands, the state and the partials to fold are all loaded before anything is decided (see k_cg_fold)
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    const int64_t j0 = i0 < n ? i0 : 0;
    S x0 = x[j0], s0 = s[j0], p0 = p[j0], y0 = y ? y[j0] : (S)0;
    const int done = st->done;
    double alpha_d = st->alpha;
    const double gamma = st->gamma, damp2 = st->damp2;
    const double v = fold_n ? jh_strided_sum256(fold_n, nfold) : 0.0;
    asm volatile("" : "+v"(x0), "+v"(s0), "+v"(p0), "+v"(y0));             // (keeps the loads above the branches below)
    if (fold_n) {
        const double pap = wg_sum_value<256>(v);
        if (done) return;
        if (blockIdx.x == 0 && threadIdx.x == 0) cg_s1(st, pap);          // itn, delta, alpha (= gamma / pap), or the breakdown (see above)
        if (!(pap > 0) || !((pap - pap) == 0.0)) return;                  // breakdown: nothing is updated (every workgroup decides alike)
        alpha_d = gamma / pap;
    } else if (done) {
        return;
    }
    const S alpha = (S)alpha_d, nalpha = (S)(-alpha_d), nd2 = (S)(-damp2);
    const bool damped = damp2 != 0.0;
    double nrm = 0.0;
    for (int64_t i = i0; i < n; i += stride) {
        if (i != i0) { x0 = x[i]; s0 = s[i]; p0 = p[i]; y0 = y ? y[i] : (S)0; }
        const S ap = alpha * p0;
        const S xn = x0 + ap;
        x[i] = xn;
        S sn = s0;
        if (y) {
            const S ay = nalpha * y0;
            sn = sn + ay;
            s[i] = sn;
        } else if (damped) {
            const S dx = nd2 * xn;
            sn = sn + dx;
            s[i] = sn;
        }
        nrm += (double)sn * (double)sn;
    }
    wg_sum_to<256>(nrm, partials + blockIdx.x);
}

static inline int64_t n_packs_of(const jh_blockop *op) { return op->col_len[0] * (int64_t)jh_dtype_size(op->dtype) / 16; }

static int cg_dev_impl(const jh_blockop *op, jh_bvec *b, jh_bvec *x, int use_x0, double damp, double atol, double btol, int maxiter, int force_maxiter,
                       jh_lsqr_result *res, double *history, const bool cgls, bool *took)
{
    *took = false;
    JH_REQUIRE(op && b && x && res, "jh_cg*_solve: null argument");
    JH_REQUIRE(maxiter >= 0, "jh_cg*_solve: maxiter must be >= 0");
    JH_TRY(jh_enter(op, b, x));
    jh_context &c = jh_ctx();
    if (c.cg_dev == 0 || maxiter < 1) return JH_OK;
    if (!(op->tall && op->all_diag) || !jh_blockop_tall_step_ok(op, b->data, x->data) || op->nrow < 2) return JH_OK;   // (rows off the 16-byte pack grid too)
    // one plain launch per pass: CGLS' second pass is the one-pass step; the fused A'A of this loop (k_cg_normal) walks all rows in every
    // workgroup, which fills the chip when the domain gives a workgroup per CU (many rows of tiny blocks keep the split walk of the host loop)
    const int64_t normal_wgs = (n_packs_of(op) + 255) / 256;
    if (c.adj_rows_per_launch != 0 || (cgls ? jh_bidiag_step_parts(op) != 1 : (c.adj_split > 0 || (op->nrow >= 256 && c.adj_split != 0 && normal_wgs < c.cu_count)))) return JH_OK;
    const int dtype = x->dtype;
    const int64_t n = x->length;
    // only where launches and host round trips matter (knob cg_dev = 2: any size): CGLS below 1 GiB per pass over the operator and r (2 GiB when the
    // blocks are below 16 MiB: the step then has one way to walk); CG through
    // the fused A'A up to 2 GiB of coefficients (64 x 128^3: 140 -> 111 us per iteration, 256 x 128^3: 368 -> 346; no difference from 4 GiB on)
    const double coeff_bytes = (double)op->nrow * (double)n * (double)jh_dtype_size(dtype);
    const bool cgls_small = 3.0 * coeff_bytes < (double)(1ull << 30) || (3.0 * coeff_bytes < (double)(2ull << 30) && coeff_bytes / (double)op->nrow < (double)(16u << 20));
    if (c.cg_dev < 2 && (cgls ? !cgls_small : coeff_bytes > (double)(2ull << 30))) return JH_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c.stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return JH_OK;
    *took = true;
    c.last_cg_graph = 0;

    struct Work {
        jh_bvec *p = nullptr, *s = nullptr, *y = nullptr;
        double *parts = nullptr, *hist = nullptr;
        jh_cg_dev *st = nullptr;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        ~Work()
        {
            if (exec) (void)hipGraphExecDestroy(exec);
            if (graph) (void)hipGraphDestroy(graph);
            if (parts) (void)hipFree(parts);
            if (hist) (void)hipFree(hist);
            if (st) (void)hipFree(st);
            if (p) (void)jh_bvec_destroy(p);
            if (s) (void)jh_bvec_destroy(s);
            if (y) (void)jh_bvec_destroy(y);
        }
    } t;
    const int64_t len1[1] = {n};
    JH_TRY(jh_bvec_create(1, len1, dtype, &t.p));
    JH_TRY(jh_bvec_create(1, len1, dtype, &t.s));
    JH_TRY(jh_bvec_create(1, len1, dtype, &t.y));
    const bool f64 = (dtype == JH_F64 || dtype == JH_C64);
    const int64_t ns_dom = n * (jh_dtype_complex(dtype) ? 2 : 1);
    int grid = (int)((ns_dom + 255) / 256 < 4096 ? (ns_dom + 255) / 256 : 4096);
    if (grid < 1) grid = 1;
    const int64_t packs = (n * (int64_t)jh_dtype_size(dtype)) / 16, grid_n = (packs + 255) / 256;
    JH_CHECK_HIP(jh_device_malloc(c.device, (void **)&t.parts, sizeof(double) * ((size_t)grid_n + (size_t)grid + 2)));
    JH_CHECK_HIP(jh_device_malloc(c.device, (void **)&t.hist, sizeof(double) * 2 * (size_t)maxiter));
    JH_CHECK_HIP(jh_device_malloc(c.device, (void **)&t.st, sizeof(jh_cg_dev)));
    double *parts_n = t.parts, *parts_s = t.parts + grid_n, *slots = parts_s + grid;
    *res = jh_lsqr_result{};
    auto lincomb2 = [&](jh_bvec *dst, double c0, const jh_bvec *x0, double c1, const jh_bvec *x1) {
        const double coef[4] = {c0, 0.0, c1, 0.0};
        const jh_bvec *v[2] = {x0, x1};
        return jh_lincomb(dst, 2, coef, v);
    };

    // ---- the start-up of cgls_impl / cgnr_impl (one shard, no exchange)
    if (!use_x0) JH_TRY(jh_fill(x, 0.0, 0.0));
    double nrm = 0.0;
    JH_TRY(jh_norm(b, 2.0, &nrm));
    const double s2 = nrm * nrm, bnorm = std::sqrt(s2);
    double rr = s2;
    if (cgls) {
        if (use_x0) JH_TRY(jh_blockop_mul_axpby(op, b, x, -1.0, 1.0, &rr));                 // r <- b - A x0 (in b's storage)
        JH_TRY(jh_blockop_mul_adj(op, t.s, b));                                            // s = A'r
        if (damp != 0.0) JH_TRY(lincomb2(t.s, 1.0, t.s, -damp * damp, x));
    } else {
        JH_TRY(jh_blockop_mul_adj(op, t.s, b));                                            // s = A'b
        if (use_x0) {                                                     // s = A'b - (A'A + damp^2) x0 ; ||r0||^2 = ||b||^2 - 2 Re<x0, A'b> + <x0, A'A x0>
            JH_TRY(jh_blockop_normal_mul(op, t.y, x));
            double xb = 0.0, xax = 0.0, im = 0.0;
            JH_TRY(jh_dot(x, t.s, &xb, &im));
            JH_TRY(jh_dot(x, t.y, &xax, &im));
            rr = s2 - 2.0 * xb + xax;
            if (damp != 0.0) {
                double x0n = 0.0;
                JH_TRY(jh_norm(x, 2.0, &x0n));
                rr += damp * damp * x0n * x0n;
            }
            JH_TRY(lincomb2(t.s, 1.0, t.s, -1.0, t.y));
            if (damp != 0.0) JH_TRY(lincomb2(t.s, 1.0, t.s, -damp * damp, x));
        }
    }
    JH_TRY(jh_norm(t.s, 2.0, &nrm));
    const double gamma0 = nrm * nrm;
    JH_TRY(jh_copy(t.p, t.s));
    jh_cg_dev h{};
    h.gamma = h.gamma0 = gamma0;
    h.rr = rr;
    h.bnorm = bnorm;
    h.damp2 = damp * damp;
    h.atol = atol;
    h.btol = btol;
    h.skip_p = 1;                                                         // p = s already
    h.maxiter = maxiter;
    h.force = force_maxiter ? 1 : 0;
    h.cgls = cgls ? 1 : 0;
    auto finish = [&]() -> int {
        double xnorm = 0.0;
        JH_TRY(jh_norm(x, 2.0, &xnorm));
        res->istop = h.istop;
        res->itn = h.itn;
        if (cgls) {
            res->r1norm = std::sqrt(h.rr);
            res->r2norm = std::sqrt(h.rr + damp * damp * xnorm * xnorm);
        } else {
            res->r2norm = std::sqrt(h.rr);
            const double r1sq = h.rr - damp * damp * xnorm * xnorm;
            res->r1norm = std::sqrt(r1sq > 0 ? r1sq : 0.0);
        }
        res->anorm = 0.0;
        res->acond = 0.0;
        res->arnorm = std::sqrt(h.gamma);
        res->xnorm = xnorm;
        return JH_OK;
    };
    if (!(gamma0 > 0)) return finish();
    auto push_state = [&]() -> int {
        JH_CHECK_HIP(hipMemcpyAsync(t.st, &h, sizeof(h), hipMemcpyHostToDevice, c.stream));
        JH_CHECK_HIP(hipStreamSynchronize(c.stream));                    // `h` is a stack object
        return JH_OK;
    };
    JH_TRY(push_state());

    int64_t nparts_n = 0, nparts_r = 0;
    // the first half of an iteration: the normal-equations pass and its fold;  the second: (CGLS: the step,) the vector updates, their fold
    bool fuse = false;                                                    // graph form of CG on the normal equations: the first fold inside k_cg_xs
    auto half1 = [&](int on_device) -> int {
        JH_TRY(jh_launch_cg_normal(op, t.p, t.s, t.y, t.st, parts_n, &nparts_n));
        if (fuse) return JH_OK;
        hipLaunchKernelGGL(k_cg_fold, dim3(1), dim3(256), 0, c.stream, (const double *)parts_n, nparts_n, (const double *)nullptr, (int64_t)0, t.st, t.hist, slots, 1,
                           on_device);
        JH_CHECK_HIP(hipGetLastError());
        return JH_OK;
    };
    auto half2 = [&](int on_device) -> int {
        const double *rparts = nullptr;
        if (cgls) {
            c.step_coef_dev = t.st->coef_step;
            c.step_done_dev = &t.st->done;
            c.step_skip_fold = 1;
            const int st_ = jh_blockop_bidiag_step_range(op, b, t.p, t.s, 0.0, 1.0, 0, n, nullptr);      // (-alpha, 1) come from the device
            c.step_coef_dev = nullptr;
            c.step_done_dev = nullptr;
            c.step_skip_fold = 0;
            JH_TRY(st_);
            nparts_r = c.last_step_parts;
            rparts = c.part_dev;
        }
        const void *yv = cgls ? nullptr : t.y->data;
        const double *fold_n = fuse ? parts_n : nullptr;
        if (f64) hipLaunchKernelGGL((k_cg_xs<double>), dim3(grid), dim3(256), 0, c.stream, (double *)x->data, (double *)t.s->data, (const double *)t.p->data, (const double *)yv, ns_dom, t.st, parts_s, fold_n, nparts_n);
        else hipLaunchKernelGGL((k_cg_xs<float>), dim3(grid), dim3(256), 0, c.stream, (float *)x->data, (float *)t.s->data, (const float *)t.p->data, (const float *)yv, ns_dom, t.st, parts_s, fold_n, nparts_n);
        hipLaunchKernelGGL(k_cg_fold, dim3(1), dim3(256), 0, c.stream, (const double *)parts_s, (int64_t)grid, rparts, nparts_r, t.st, t.hist, slots, 2, on_device);
        JH_CHECK_HIP(hipGetLastError());
        return JH_OK;
    };
    double sums[2] = {0.0, 0.0};
    auto pull_sums = [&]() -> int {
        JH_CHECK_HIP(hipMemcpyAsync(sums, slots, sizeof(sums), hipMemcpyDeviceToHost, c.stream));
        JH_CHECK_HIP(hipStreamSynchronize(c.stream));
        return JH_OK;
    };
    // one host-driven iteration: the same kernels, the scalar updates applied HERE between them
    std::vector<double> hist_host(history ? 2 * (size_t)maxiter : 0);
    auto host_iteration = [&]() -> int {
        JH_TRY(half1(0));
        JH_TRY(pull_sums());
        cg_s1(&h, sums[0]);
        JH_TRY(push_state());
        if (h.done) return JH_OK;
        JH_TRY(half2(0));
        JH_TRY(pull_sums());
        cg_s2(&h, sums[0], sums[1], history ? hist_host.data() : nullptr);
        return push_state();
    };
    if (c.lsqr_graph == 0) {                                              // knob lsqr_graph = 0: host-driven, iteration by iteration
        while (!h.done) JH_TRY(host_iteration());
        if (history && h.itn > 0) memcpy(history, hist_host.data(), sizeof(double) * 2 * (size_t)h.itn);
        return finish();
    }
    // iteration 1 eagerly, the scalar updates on the device (workspaces get their size; the step kernel's shape is settled)
    struct Flags { int itn, istop, done; } fl{};
    auto read_flags = [&]() -> int {
        JH_CHECK_HIP(hipMemcpyAsync(&fl, &t.st->itn, sizeof(fl), hipMemcpyDeviceToHost, c.stream));
        JH_CHECK_HIP(hipStreamSynchronize(c.stream));
        return JH_OK;
    };
    fuse = !cgls;                                                         // (CGLS's step needs alpha before the vector update: its first fold stays a launch)
    JH_TRY(half1(1));
    JH_TRY(half2(1));
    JH_TRY(read_flags());
    int64_t replays = 0;
    if (!fl.done) {
        if (cgls && !(nparts_r > 0 && nparts_r <= 8192)) return jh_fail(JH_ERR_STATE, "device-resident CGLS: the step left %lld partial sums (expected 1..8192)", (long long)nparts_r);
        const uint64_t gen = c.buf_gen;
        // EIGHT iterations per graph (a finished solve replays as no-ops): one hipGraphLaunch costs the host 10-16 us, about what a whole
        // iteration of a 64 x 64^3 operator takes on the device
        const int per_graph = maxiter < 8 ? maxiter : 8;
        JH_CHECK_HIP(hipStreamBeginCapture(c.stream, hipStreamCaptureModeRelaxed));
        int st_ = JH_OK;
        for (int k = 0; k < per_graph && st_ == JH_OK; k++) {
            st_ = half1(1);
            if (st_ == JH_OK) st_ = half2(1);
        }
        hipError_t e = hipStreamEndCapture(c.stream, &t.graph);
        if (st_ != JH_OK) return st_;
        JH_CHECK_HIP(e);
        JH_REQUIRE(gen == c.buf_gen, "device-resident CG: a workspace was reallocated during capture");
        JH_CHECK_HIP(hipGraphInstantiate(&t.exec, t.graph, nullptr, nullptr, 0));
        while (!fl.done) {                                                // two graphs between two looks at the flags
            for (int k = 0; k < 2; k++) JH_CHECK_HIP(hipGraphLaunch(t.exec, c.stream));
            replays += 2;
            JH_TRY(read_flags());
        }
    }
    c.last_cg_graph = replays;
    JH_CHECK_HIP(hipMemcpyAsync(&h, t.st, sizeof(h), hipMemcpyDeviceToHost, c.stream));
    JH_CHECK_HIP(hipStreamSynchronize(c.stream));
    if (history && h.itn > 0) {
        JH_CHECK_HIP(hipMemcpyAsync(history, t.hist, sizeof(double) * 2 * (size_t)h.itn, hipMemcpyDeviceToHost, c.stream));
        JH_CHECK_HIP(hipStreamSynchronize(c.stream));
    }
    return finish();
}

extern "C" int jh_cgls_solve(const jh_blockop *op, jh_bvec *u, jh_bvec *x, int use_x0, double damp, double atol, double btol, int maxiter,
                             int force_maxiter, jh_lsqr_result *res, double *history)
{
    bool took = false;
    JH_TRY(cg_dev_impl(op, u, x, use_x0, damp, atol, btol, maxiter, force_maxiter, res, history, true, &took));   // small operators: recurrences on the device
    if (took) return JH_OK;
    return cgls_impl(1, &op, &u, &x, use_x0, damp, atol, btol, maxiter, force_maxiter, res, history, Exch::none);
}

extern "C" int jh_cgls_solve_partitioned(const jh_blockop *op, jh_bvec *u, jh_bvec *x, int use_x0, double damp, double atol, double btol,
                                         int maxiter, int force_maxiter, jh_lsqr_result *res, double *history)
{
    JH_TRY(jh_enter(op, u, x));
    int nranks = 1, rank = 0, has_comm = 0;
    (void)jh_comm_info(&nranks, &rank);
    (void)jh_comm_exists(&has_comm);
    if (has_comm == 2 && (nranks > 1 || jh_ctx().force_dist))
        return jh_fail(JH_ERR_UNSUPPORTED, "jh_cgls_solve_partitioned: this context is a member of a single-process team (jh_comm_init_all): "
                       "jh_cgls_solve_team takes all the members' shards in one call");
    return cgls_impl(1, &op, &u, &x, use_x0, damp, atol, btol, maxiter, force_maxiter, res, history,
                     (nranks > 1 || (has_comm && jh_ctx().force_dist)) ? Exch::ranks : Exch::none);
}

extern "C" int jh_cgls_solve_team(int n, const jh_blockop *const *ops, jh_bvec *const *us, jh_bvec *const *xs, int use_x0, double damp, double atol,
                                  double btol, int maxiter, int force_maxiter, jh_lsqr_result *res, double *history)
{
    JH_REQUIRE(n >= 1 && n <= JH_MAX_CTX && ops && us && xs, "jh_cgls_solve_team: need 1..%d members", JH_MAX_CTX);
    for (int k = 0; k < n; k++) {
        JH_REQUIRE(ops[k] && us[k] && xs[k], "jh_cgls_solve_team: null handle of member %d", k);
        JH_TRY(jh_enter(ops[k], us[k], xs[k]));
        int nranks = 1, rank = 0, has_comm = 0;
        (void)jh_comm_info(&nranks, &rank);
        (void)jh_comm_exists(&has_comm);
        JH_REQUIRE(has_comm == 2 && nranks == n && rank == k, "jh_cgls_solve_team: the handles of member %d must live in member %d's context of a team of %d "
                   "(jh_comm_init_all); found %s, rank %d of %d", k, k, n, has_comm == 2 ? "a team" : "no team", rank, nranks);
    }
    return cgls_impl(n, ops, us, xs, use_x0, damp, atol, btol, maxiter, force_maxiter, res, history, Exch::team);
}

// ------------------------------------------------------------------ CG on the normal equations through the fused A'A (round 3) -------------
// BASELINE.json configs[2] is "JetComposite A' o A normal-equations matvec ... (chain fusion + reductions)": this is the solver that
// matvec is for.  Conjugate gradients applied to (A'A + damp^2 I) x = A'b with the fused normal operator (jh_blockop_normal_mul reads
// the coefficients ONCE and never forms a range-sized intermediate): after ONE adjoint pass for A'b an iteration moves N n s bytes --
// a third of LSQR's one-pass iteration, a quarter of CGLS's -- plus domain-sized vectors; the range vector b is read once and never
// written.  In exact arithmetic the iterates are CGLS's / LSQR's; in floating point the residual s = A'r lives in the DOMAIN and is
// updated by recurrence (s -= alpha (A'A + damp^2) p) instead of being recomputed from r, the textbook weakness of "CGNR done
// naively" (Bjorck 1996, section 7.4): the attainable accuracy degrades with cond(A)^2 where CGLS / LSQR keep cond(A).  For the
// well-conditioned operators of the benchmark (every column sums N squared coefficients) it reaches the same x; ||r|| is tracked by
// its own recurrence ||r_k||^2 = ||r_{k-1}||^2 - alpha_k gamma_{k-1} (exact for CG), so no pass over the range is needed for it.
// Partitioned: every shard's A_k'A_k p in 4 element ranges (jh_blockop_normal_mul_range), each range all-reduced on the exchange stream
// under the next range's kernel; the scalars come from the replicated vectors, so the exchange of y is the only collective.
static int cgnr_impl(const int M, const jh_blockop *const *ops, jh_bvec *const *bs, jh_bvec *const *xs, int use_x0, double damp, double atol,
                     double btol, int maxiter, int force_maxiter, jh_lsqr_result *res, double *history, const Exch ex)
{
    JH_REQUIRE(ops && bs && xs && res && M >= 1, "jh_cgnr_solve: null argument");
    JH_REQUIRE(maxiter >= 0, "jh_cgnr_solve: maxiter must be >= 0");
    for (int k = 0; k < M; k++) JH_REQUIRE(ops[k] && bs[k] && xs[k], "jh_cgnr_solve: null argument (member %d)", k);
    auto use = [&](int k) { return jh_enter(ops[k], bs[k], xs[k]); };
    JH_TRY(use(0));
    int64_t nb = 0, n = 0;
    int dtype = 0;
    JH_TRY(jh_bvec_info(xs[0], &nb, &n, &dtype, nullptr));
    for (int k = 0; k < M; k++) {
        JH_REQUIRE(xs[k]->length == n && xs[k]->dtype == dtype, "jh_cgnr_solve: member %d's x differs in length or element type", k);
        // (round 6: one shard may also be an N x (2 .. 4) grid of equal diagonals -- its fused A'A is jh_grid_normal.hip; the loop below only ever calls
        // jh_blockop_mul_adj and jh_blockop_normal_mul on whole vectors there)
        const bool grid = M == 1 && ex == Exch::none && jhb::grid_normal_ok(ops[k], xs[k]->data, xs[k]->data);
        if (!grid && (!jh_blockop_tall_step_ok(ops[k], bs[k]->data, xs[k]->data) || ops[k]->nrow < 2))
            return jh_fail(JH_ERR_UNSUPPORTED, "jh_cgnr_solve: needs a tall (>= 2 rows) operator of equal elementwise rows (or, unpartitioned, an N x (2 .. 4) grid of equal diagonals)");
    }
    struct Work {
        jh_bvec *p = nullptr, *s = nullptr, *y = nullptr;
        ~Work()
        {
            if (p) (void)jh_bvec_destroy(p);
            if (s) (void)jh_bvec_destroy(s);
            if (y) (void)jh_bvec_destroy(y);
        }
    };
    std::vector<Work> t((size_t)M);
    const int64_t len1[1] = {n};
    for (int k = 0; k < M; k++) {
        JH_TRY(use(k));
        JH_TRY(jh_bvec_create(1, len1, dtype, &t[k].p));
        JH_TRY(jh_bvec_create(1, len1, dtype, &t[k].s));
        JH_TRY(jh_bvec_create(1, len1, dtype, &t[k].y));
    }
    *res = jh_lsqr_result{};
    auto global_sum = [&](const std::vector<double> &locals, double *out) -> int {
        double sum = 0.0;
        for (double v : locals) sum += v;
        *out = sum;
        if (ex == Exch::ranks) JH_TRY(jh_comm_allreduce_scalars(out, 1, 0));
        return JH_OK;
    };
    auto allreduce = [&](auto &&vec_of) -> int {                         // every member's domain vector summed over the shards
        if (ex == Exch::none) return JH_OK;
        if (ex == Exch::team) { JH_TRY(use(0)); JH_TRY(jh_comm_group_begin()); }
        int st = JH_OK;
        for (int k = 0; k < M && st == JH_OK; k++) st = jh_comm_allreduce_sum(vec_of(k));
        if (ex == Exch::team) { (void)use(0); const int st2 = jh_comm_group_end(); if (st == JH_OK) st = st2; }
        return st;
    };
    auto lincomb2 = [&](jh_bvec *dst, double c0, const jh_bvec *x0, double c1, const jh_bvec *x1) {
        const double coef[4] = {c0, 0.0, c1, 0.0};
        const jh_bvec *v[2] = {x0, x1};
        return jh_lincomb(dst, 2, coef, v);
    };
    std::vector<double> locals((size_t)M);

    // ||b||, s = A'b
    double s2 = 0.0;
    for (int k = 0; k < M; k++) {
        if (!use_x0) JH_TRY(jh_fill(xs[k], 0.0, 0.0));
        double nrm = 0.0;
        JH_TRY(jh_norm(bs[k], 2.0, &nrm));
        locals[k] = nrm * nrm;
    }
    JH_TRY(global_sum(locals, &s2));
    const double bnorm = std::sqrt(s2);
    for (int k = 0; k < M; k++) JH_TRY(jh_blockop_mul_adj(ops[k], t[k].s, bs[k]));
    JH_TRY(allreduce([&](int k) { return t[k].s; }));
    double rr = s2;                                                       // ||b - A x||^2, by recurrence
    if (use_x0) {                                                         // s = A'b - (A'A + damp^2) x0 ;  ||r0||^2 = ||b||^2 - 2 Re<x0, A'b> + <x0, A'A x0>
        for (int k = 0; k < M; k++) JH_TRY(jh_blockop_normal_mul(ops[k], t[k].y, xs[k]));
        JH_TRY(allreduce([&](int k) { return t[k].y; }));
        double xb = 0.0, xax = 0.0, im = 0.0;
        JH_TRY(jh_dot(xs[0], t[0].s, &xb, &im));
        JH_TRY(jh_dot(xs[0], t[0].y, &xax, &im));
        rr = s2 - 2.0 * xb + xax;
        if (damp != 0.0) {                                                // the functional CG decreases is ||r||^2 + damp^2 ||x||^2
            double x0n = 0.0;
            JH_TRY(jh_norm(xs[0], 2.0, &x0n));
            rr += damp * damp * x0n * x0n;
        }
        for (int k = 0; k < M; k++) {
            JH_TRY(lincomb2(t[k].s, 1.0, t[k].s, -1.0, t[k].y));
            if (damp != 0.0) JH_TRY(lincomb2(t[k].s, 1.0, t[k].s, -damp * damp, xs[k]));
        }
    }
    double nrm = 0.0;
    JH_TRY(jh_norm(t[0].s, 2.0, &nrm));
    double gamma = nrm * nrm;
    const double gamma0 = gamma;
    for (int k = 0; k < M; k++) JH_TRY(jh_copy(t[k].p, t[k].s));
    int itn = 0, istop = 0;
    int64_t chunk = (n + 3) / 4;                                          // exchange ranges: 4, on 64 KiB boundaries (as lsqr_impl)
    chunk = (chunk + 16383) / 16384 * 16384;
    // y = sum over the shards of A_k'A_k p.  Partitioned: in 4 element ranges -- the all-reduce of a finished range runs on the exchange
    // stream while the kernel of the next range computes; the library streams then wait for the exchange (no host synchronisation)
    auto normal_all = [&]() -> int {
        if (ex == Exch::none) return jh_blockop_normal_mul(ops[0], t[0].y, t[0].p);
        for (int64_t lo = 0; lo < n; lo += chunk) {
            const int64_t cnt = lo + chunk < n ? chunk : n - lo;
            for (int k = 0; k < M; k++) JH_TRY(jh_blockop_normal_mul_range(ops[k], t[k].y, t[k].p, lo, cnt));
            if (ex == Exch::team) { JH_TRY(use(0)); JH_TRY(jh_comm_group_begin()); }
            int st = JH_OK;
            for (int k = 0; k < M && st == JH_OK; k++) st = jh_comm_allreduce_sum_range(t[k].y, lo, cnt);
            if (ex == Exch::team) { (void)use(0); const int st2 = jh_comm_group_end(); if (st == JH_OK) st = st2; }
            JH_TRY(st);
        }
        for (int k = 0; k < M; k++) { JH_TRY(use(k)); JH_TRY(jh_comm_join()); }
        return JH_OK;
    };
    if (gamma > 0) {
        while (itn < maxiter) {
            itn++;
            JH_TRY(normal_all());                                         // the ONE pass over the operator
            if (damp != 0.0)
                for (int k = 0; k < M; k++) JH_TRY(lincomb2(t[k].y, 1.0, t[k].y, damp * damp, t[k].p));
            double delta = 0.0, im = 0.0;
            JH_TRY(jh_dot(t[0].p, t[0].y, &delta, &im));                  // replicas are identical: member 0 speaks for all
            if (!(delta > 0) || !std::isfinite(delta)) {
                istop = 6;
                itn--;
                break;
            }
            const double alpha = gamma / delta;
            for (int k = 0; k < M; k++) {
                JH_TRY(lincomb2(xs[k], 1.0, xs[k], alpha, t[k].p));      // x += alpha p
                JH_TRY(lincomb2(t[k].s, 1.0, t[k].s, -alpha, t[k].y));   // s -= alpha (A'A + damp^2) p
            }
            rr -= alpha * gamma;                                          // ||r||^2 (+ damp^2 ||x||^2 when damped: the functional's value)
            if (rr < 0) rr = 0;
            JH_TRY(jh_norm(t[0].s, 2.0, &nrm));
            const double gamma_new = nrm * nrm;
            const double bk = gamma_new / gamma;
            for (int k = 0; k < M; k++) JH_TRY(lincomb2(t[k].p, 1.0, t[k].s, bk, t[k].p));
            gamma = gamma_new;
            const double rnorm = std::sqrt(rr), arnorm = std::sqrt(gamma);
            if (history) { history[2 * (itn - 1)] = rnorm; history[2 * (itn - 1) + 1] = arnorm; }
            if (itn >= maxiter) istop = 7;
            if (arnorm <= atol * std::sqrt(gamma0)) istop = 2;
            if (rnorm <= btol * bnorm) istop = 1;
            if (istop && !(force_maxiter && itn < maxiter && gamma > 0)) break;
        }
    }
    double xnorm = 0.0;
    JH_TRY(jh_norm(xs[0], 2.0, &xnorm));
    res->istop = istop;
    res->itn = itn;
    res->r2norm = std::sqrt(rr);                                          // sqrt(||r||^2 + damp^2 ||x||^2): what the recurrence tracks
    const double r1sq = rr - damp * damp * xnorm * xnorm;
    res->r1norm = std::sqrt(r1sq > 0 ? r1sq : 0.0);
    res->anorm = 0.0;
    res->acond = 0.0;
    res->arnorm = std::sqrt(gamma);
    res->xnorm = xnorm;
    for (int k = 0; k < M; k++) {
        jh_context *ck = jh_ctx_by_id(ops[k]->ctx);
        if (ck) { JH_TRY(use(k)); JH_CHECK_HIP(hipStreamSynchronize(ck->stream)); }
    }
    return JH_OK;
}

extern "C" int jh_cgnr_solve(const jh_blockop *op, jh_bvec *b, jh_bvec *x, int use_x0, double damp, double atol, double btol, int maxiter,
                             int force_maxiter, jh_lsqr_result *res, double *history)
{
    bool took = false;
    JH_TRY(cg_dev_impl(op, b, x, use_x0, damp, atol, btol, maxiter, force_maxiter, res, history, false, &took));  // small operators: recurrences on the device
    if (took) return JH_OK;
    return cgnr_impl(1, &op, &b, &x, use_x0, damp, atol, btol, maxiter, force_maxiter, res, history, Exch::none);
}

extern "C" int jh_cgnr_solve_partitioned(const jh_blockop *op, jh_bvec *b, jh_bvec *x, int use_x0, double damp, double atol, double btol,
                                         int maxiter, int force_maxiter, jh_lsqr_result *res, double *history)
{
    JH_TRY(jh_enter(op, b, x));
    int nranks = 1, rank = 0, has_comm = 0;
    (void)jh_comm_info(&nranks, &rank);
    (void)jh_comm_exists(&has_comm);
    if (has_comm == 2 && (nranks > 1 || jh_ctx().force_dist))
        return jh_fail(JH_ERR_UNSUPPORTED, "jh_cgnr_solve_partitioned: this context is a member of a single-process team (jh_comm_init_all): "
                       "jh_cgnr_solve_team takes all the members' shards in one call");
    return cgnr_impl(1, &op, &b, &x, use_x0, damp, atol, btol, maxiter, force_maxiter, res, history,
                     (nranks > 1 || (has_comm && jh_ctx().force_dist)) ? Exch::ranks : Exch::none);
}

extern "C" int jh_cgnr_solve_team(int n, const jh_blockop *const *ops, jh_bvec *const *bs, jh_bvec *const *xs, int use_x0, double damp, double atol,
                                  double btol, int maxiter, int force_maxiter, jh_lsqr_result *res, double *history)
{
    JH_REQUIRE(n >= 1 && n <= JH_MAX_CTX && ops && bs && xs, "jh_cgnr_solve_team: need 1..%d members", JH_MAX_CTX);
    for (int k = 0; k < n; k++) {
        JH_REQUIRE(ops[k] && bs[k] && xs[k], "jh_cgnr_solve_team: null handle of member %d", k);
        JH_TRY(jh_enter(ops[k], bs[k], xs[k]));
        int nranks = 1, rank = 0, has_comm = 0;
        (void)jh_comm_info(&nranks, &rank);
        (void)jh_comm_exists(&has_comm);
        JH_REQUIRE(has_comm == 2 && nranks == n && rank == k, "jh_cgnr_solve_team: the handles of member %d must live in member %d's context of a team of %d "
                   "(jh_comm_init_all); found %s, rank %d of %d", k, k, n, has_comm == 2 ? "a team" : "no team", rank, nranks);
    }
    return cgnr_impl(n, ops, bs, xs, use_x0, damp, atol, btol, maxiter, force_maxiter, res, history, Exch::team);
}
