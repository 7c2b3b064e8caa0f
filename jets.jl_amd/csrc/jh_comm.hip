// jh_comm.hip -- RCCL over xGMI from the C ABI, for hosts without torch.distributed (e.g. the Julia binding).
// One process per GPU; rank 0 produces a 128-byte unique id (jh_comm_unique_id), the host language ships it to
// the other ranks (MPI, sockets, a file), every rank calls jh_comm_init_rank.  Collectives run on the library's
// HIP stream, so they are ordered against the kernels without host synchronisation.
//
// librccl is resolved at run time (dlopen): a process that already carries an RCCL (PyTorch bundles one) reuses
// it instead of loading a second copy, and libjetship.so itself has no link-time dependency on RCCL.
#include "jh_internal.h"
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct rccl_api {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

rccl_api g_api;
ncclComm_t g_comm = nullptr;
int g_nranks = 0, g_rank = -1;
double *g_scalar_dev = nullptr;    // 64 doubles for scalar all-reduces
// the exchange stream: ranged all-reduces run here, ordered after the library stream by events, so that the all-reduce of a
// finished range of the domain vector overlaps the kernel that computes the next range (jh_comm_allreduce_sum_range / jh_comm_join)
hipStream_t g_cstream = nullptr;
hipEvent_t g_ev_main = nullptr, g_ev_comm = nullptr;
double *g_scalar_host = nullptr;   // pinned landing zone of jh_comm_allreduce_normsq

int load_rccl()
{
    if (g_api.lib) return JH_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; }     // already in the process?
    if (!h) for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) return jh_fail(JH_ERR_COMM, "jh_comm: cannot load librccl (%s)", dlerror());
    g_api.lib = h;
    g_api.GetUniqueId = (decltype(g_api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_api.CommInitRank = (decltype(g_api.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_api.CommDestroy = (decltype(g_api.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_api.AllReduce = (decltype(g_api.AllReduce))dlsym(h, "ncclAllReduce");
    g_api.GetErrorString = (decltype(g_api.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_api.GetUniqueId || !g_api.CommInitRank || !g_api.CommDestroy || !g_api.AllReduce || !g_api.GetErrorString) {
        g_api = rccl_api();
        return jh_fail(JH_ERR_COMM, "jh_comm: librccl lacks a required symbol");
    }
    return JH_OK;
}

#define JH_CHECK_NCCL(expr)                                                                              \
    do {                                                                                                 \
        ncclResult_t _r = (expr);                                                                        \
        if (_r != ncclSuccess) return jh_fail(JH_ERR_COMM, "%s: %s", #expr, g_api.GetErrorString(_r));   \
    } while (0)

}  // namespace

extern "C" {

int jh_comm_unique_id(void *out128)
{
    JH_REQUIRE(out128, "jh_comm_unique_id: null output");
    JH_TRY(load_rccl());
    ncclUniqueId id;
    JH_CHECK_NCCL(g_api.GetUniqueId(&id));
    memcpy(out128, &id, NCCL_UNIQUE_ID_BYTES);
    return JH_OK;
}

int jh_comm_init_rank(const void *id128, int nranks, int rank)
{
    JH_TRY(jh_require_ready());
    JH_REQUIRE(id128, "jh_comm_init_rank: null id");
    JH_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "jh_comm_init_rank: rank %d of %d", rank, nranks);
    if (g_comm) return jh_fail(JH_ERR_STATE, "jh_comm_init_rank: communicator already initialised (rank %d of %d)", g_rank, g_nranks);
    JH_TRY(load_rccl());
    JH_CHECK_HIP(hipSetDevice(jh_ctx().device));
    ncclUniqueId id;
    memcpy(&id, id128, NCCL_UNIQUE_ID_BYTES);
    JH_CHECK_NCCL(g_api.CommInitRank(&g_comm, nranks, id, rank));
    JH_CHECK_HIP(hipMalloc((void **)&g_scalar_dev, sizeof(double) * 64));
    JH_CHECK_HIP(hipStreamCreateWithFlags(&g_cstream, hipStreamNonBlocking));
    JH_CHECK_HIP(hipEventCreateWithFlags(&g_ev_main, hipEventDisableTiming));
    JH_CHECK_HIP(hipEventCreateWithFlags(&g_ev_comm, hipEventDisableTiming));
    JH_CHECK_HIP(hipHostMalloc((void **)&g_scalar_host, sizeof(double) * 8, hipHostMallocDefault));
    g_nranks = nranks;
    g_rank = rank;
    return JH_OK;
}

int jh_comm_destroy(void)
{
    if (!g_comm) return JH_OK;
    if (jh_ctx().ready) (void)hipStreamSynchronize(jh_ctx().stream);
    if (g_cstream) (void)hipStreamSynchronize(g_cstream);
    ncclResult_t r = g_api.CommDestroy(g_comm);
    g_comm = nullptr;
    if (g_scalar_dev) { (void)hipFree(g_scalar_dev); g_scalar_dev = nullptr; }
    if (g_cstream) { (void)hipStreamDestroy(g_cstream); g_cstream = nullptr; }
    if (g_ev_main) { (void)hipEventDestroy(g_ev_main); g_ev_main = nullptr; }
    if (g_ev_comm) { (void)hipEventDestroy(g_ev_comm); g_ev_comm = nullptr; }
    if (g_scalar_host) { (void)hipHostFree(g_scalar_host); g_scalar_host = nullptr; }
    g_nranks = 0;
    g_rank = -1;
    if (r != ncclSuccess) return jh_fail(JH_ERR_COMM, "ncclCommDestroy: %s", g_api.GetErrorString(r));
    return JH_OK;
}

int jh_comm_exists(int *yes)
{
    if (yes) *yes = g_comm ? 1 : 0;
    return JH_OK;
}

int jh_comm_info(int *nranks, int *rank)
{
    if (nranks) *nranks = g_comm ? g_nranks : 1;
    if (rank) *rank = g_comm ? g_rank : 0;
    return JH_OK;
}

// in-place sum of a replicated vector over all ranks (the adjoint accumulate of a row-partitioned tall operator)
int jh_comm_allreduce_sum(jh_bvec *v)
{
    JH_TRY(jh_require_ready());
    JH_REQUIRE(v, "jh_comm_allreduce_sum: null vector");
    if (!g_comm) return jh_fail(JH_ERR_STATE, "jh_comm_allreduce_sum: call jh_comm_init_rank first");
    if (v->length == 0) return JH_OK;
    const bool f32 = (v->dtype == JH_F32 || v->dtype == JH_C32);
    const size_t count = (size_t)v->length * (jh_dtype_complex(v->dtype) ? 2 : 1);
    JH_CHECK_NCCL(g_api.AllReduce(v->data, v->data, count, f32 ? ncclFloat32 : ncclFloat64, ncclSum, g_comm, jh_ctx().stream));
    return JH_OK;
}

// The same sum restricted to the elements [first_elem, first_elem + count) of v, ASYNCHRONOUS with respect to the library
// stream: it runs on the communicator's own stream, ordered after everything enqueued on the library stream so far (the kernel
// that has just produced that range), while kernels enqueued on the library stream afterwards (the next range) run concurrently.
// jh_comm_join makes the library stream wait for every ranged all-reduce enqueued so far.  All ranks must enqueue the same
// ranges in the same order.
int jh_comm_allreduce_sum_range(jh_bvec *v, int64_t first_elem, int64_t count)
{
    JH_TRY(jh_require_ready());
    JH_REQUIRE(v, "jh_comm_allreduce_sum_range: null vector");
    if (!g_comm) return jh_fail(JH_ERR_STATE, "jh_comm_allreduce_sum_range: call jh_comm_init_rank first");
    JH_REQUIRE(first_elem >= 0 && count >= 0 && first_elem + count <= v->length,
               "jh_comm_allreduce_sum_range: elements [%lld, %lld) outside the vector (%lld elements)", (long long)first_elem,
               (long long)(first_elem + count), (long long)v->length);
    if (count == 0) return JH_OK;
    const bool f32 = (v->dtype == JH_F32 || v->dtype == JH_C32);
    const size_t nscal = (size_t)count * (jh_dtype_complex(v->dtype) ? 2 : 1);
    void *p = v->ptr(first_elem);
    JH_CHECK_HIP(hipEventRecord(g_ev_main, jh_ctx().stream));
    JH_CHECK_HIP(hipStreamWaitEvent(g_cstream, g_ev_main, 0));
    JH_CHECK_NCCL(g_api.AllReduce(p, p, nscal, f32 ? ncclFloat32 : ncclFloat64, ncclSum, g_comm, g_cstream));
    return JH_OK;
}

int jh_comm_join(void)
{
    JH_TRY(jh_require_ready());
    if (!g_comm) return jh_fail(JH_ERR_STATE, "jh_comm_join: call jh_comm_init_rank first");
    JH_CHECK_HIP(hipEventRecord(g_ev_comm, g_cstream));
    JH_CHECK_HIP(hipStreamWaitEvent(jh_ctx().stream, g_ev_comm, 0));
    return JH_OK;
}

// Sum over all ranks of the deferred ||u||^2 accumulator (jh_normsq_reset / jh_blockop_bidiag_step_range with normsq == NULL),
// on the exchange stream behind the ranged all-reduces: when it returns, the kernels of the step, every ranged all-reduce
// enqueued before it and the scalar sum are complete -- the ONE host synchronisation of a pipelined distributed step.
int jh_comm_allreduce_normsq(double *out)
{
    JH_TRY(jh_require_ready());
    JH_REQUIRE(out, "jh_comm_allreduce_normsq: null output");
    if (!g_comm) return jh_fail(JH_ERR_STATE, "jh_comm_allreduce_normsq: call jh_comm_init_rank first");
    double *slot = jh_ctx().red_dev + JH_NORMSQ_SLOT;
    JH_CHECK_HIP(hipEventRecord(g_ev_main, jh_ctx().stream));
    JH_CHECK_HIP(hipStreamWaitEvent(g_cstream, g_ev_main, 0));
    JH_CHECK_NCCL(g_api.AllReduce(slot, g_scalar_dev, 1, ncclFloat64, ncclSum, g_comm, g_cstream));      // the local accumulator stays local
    JH_CHECK_HIP(hipMemcpyAsync(g_scalar_host, g_scalar_dev, sizeof(double), hipMemcpyDeviceToHost, g_cstream));
    JH_CHECK_HIP(hipStreamSynchronize(g_cstream));
    *out = g_scalar_host[0];
    return JH_OK;
}

// batched scalar all-reduce (range-side dot / norm^2 / extrema partials); op: 0 sum, 1 max, 2 min.  Synchronises.
int jh_comm_allreduce_scalars(double *values, int n, int op)
{
    JH_TRY(jh_require_ready());
    JH_REQUIRE(values && n >= 1 && n <= 64, "jh_comm_allreduce_scalars: need 1..64 values");
    JH_REQUIRE(op >= 0 && op <= 2, "jh_comm_allreduce_scalars: op must be 0 (sum), 1 (max) or 2 (min)");
    if (!g_comm) return jh_fail(JH_ERR_STATE, "jh_comm_allreduce_scalars: call jh_comm_init_rank first");
    hipStream_t st = jh_ctx().stream;
    JH_CHECK_HIP(hipMemcpyAsync(g_scalar_dev, values, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
    const ncclRedOp_t red = op == 0 ? ncclSum : (op == 1 ? ncclMax : ncclMin);
    JH_CHECK_NCCL(g_api.AllReduce(g_scalar_dev, g_scalar_dev, (size_t)n, ncclFloat64, red, g_comm, st));
    JH_CHECK_HIP(hipMemcpyAsync(values, g_scalar_dev, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
    JH_CHECK_HIP(hipStreamSynchronize(st));
    return JH_OK;
}

}  // extern "C"
