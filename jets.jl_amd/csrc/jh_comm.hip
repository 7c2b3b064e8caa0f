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
    g_nranks = nranks;
    g_rank = rank;
    return JH_OK;
}

int jh_comm_destroy(void)
{
    if (!g_comm) return JH_OK;
    if (jh_ctx().ready) (void)hipStreamSynchronize(jh_ctx().stream);
    ncclResult_t r = g_api.CommDestroy(g_comm);
    g_comm = nullptr;
    if (g_scalar_dev) { (void)hipFree(g_scalar_dev); g_scalar_dev = nullptr; }
    g_nranks = 0;
    g_rank = -1;
    if (r != ncclSuccess) return jh_fail(JH_ERR_COMM, "ncclCommDestroy: %s", g_api.GetErrorString(r));
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
