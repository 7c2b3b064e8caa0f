// jh_comm.hip -- RCCL over xGMI from the C ABI, for hosts without torch.distributed (e.g. the Julia binding).
//
// Two ways to form a communicator, both PER CONTEXT (jh_internal.h: a context = one device + one stream):
//   * one process per GPU: rank 0 produces a 128-byte unique id (jh_comm_unique_id), the host language ships it to the other
//     ranks (MPI, sockets, a file), every rank calls jh_comm_init_rank on its context;
//   * ONE process driving several GPUs (SURVEY section 8e's sketch: ncclCommInitAll, one stream per device, grouped calls):
//     jh_comm_init_all(n, contexts) makes the n contexts of this process a TEAM; the host then issues the same collective once
//     per member between jh_comm_group_begin / jh_comm_group_end (ncclGroupStart / ncclGroupEnd -- without the group a
//     single thread would block in the first member's call).  Scalars need no collective in a team: the host reads every
//     member's partial (jh_normsq_read, jh_dot, ...) and adds.
//     When all members of a team sit on ONE device (several contexts = several streams of one GPU; RCCL refuses that), the
//     grouped sum is a device kernel over the members' buffers -- the same semantics, deterministic (members in rank order).
// Collectives run on the context's stream (or its exchange stream), ordered against the kernels without host synchronisation.
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
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
rccl_api g_api;

constexpr int TEAM_LOCAL_MAX = 16;      // members of a one-device team (pointer table passed by value)

struct team_t {
    int n = 0;
    int ctx[JH_MAX_CTX];
    bool local = false;                 // all members on one device: the sum is a device kernel, no RCCL
    // the open group (jh_comm_group_begin .. _end)
    bool open = false;
    int nrec = 0;
    struct rec { int ctx; void *p; size_t nscal; bool f32; hipStream_t stream; } recs[JH_MAX_CTX];
};

struct comm_state {
    bool alive = false;
    ncclComm_t comm = nullptr;          // null for a member of a one-device team
    int nranks = 0, rank = -1;
    team_t *team = nullptr;             // single-process team, or null for a communicator of jh_comm_init_rank
    double *scalar_dev = nullptr;       // 64 doubles for scalar all-reduces
    // the exchange stream: ranged all-reduces run here, ordered after the library stream by events, so that the all-reduce of a
    // finished range of the domain vector overlaps the kernel that computes the next range (jh_comm_allreduce_sum_range / jh_comm_join)
    hipStream_t cstream = nullptr;
    hipEvent_t ev_main = nullptr, ev_comm = nullptr;
    double *scalar_host = nullptr;      // pinned landing zone of jh_comm_allreduce_normsq
};
comm_state g_cs[JH_MAX_CTX];
comm_state none_cs;

comm_state &cs()
{
    const int id = jh_ctx().id;
    return id >= 0 ? g_cs[jh_ctx_slot(id)] : none_cs;
}

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
    g_api.CommInitAll = (decltype(g_api.CommInitAll))dlsym(h, "ncclCommInitAll");
    g_api.CommDestroy = (decltype(g_api.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_api.AllReduce = (decltype(g_api.AllReduce))dlsym(h, "ncclAllReduce");
    g_api.GroupStart = (decltype(g_api.GroupStart))dlsym(h, "ncclGroupStart");
    g_api.GroupEnd = (decltype(g_api.GroupEnd))dlsym(h, "ncclGroupEnd");
    g_api.GetErrorString = (decltype(g_api.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_api.GetUniqueId || !g_api.CommInitRank || !g_api.CommInitAll || !g_api.CommDestroy || !g_api.AllReduce || !g_api.GroupStart ||
        !g_api.GroupEnd || !g_api.GetErrorString) {
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

// the side objects every communicator carries; the CURRENT context is the owner
void free_side_objects(comm_state &s)
{
    if (s.scalar_dev) (void)hipFree(s.scalar_dev);
    if (s.cstream) (void)hipStreamDestroy(s.cstream);
    if (s.ev_main) (void)hipEventDestroy(s.ev_main);
    if (s.ev_comm) (void)hipEventDestroy(s.ev_comm);
    if (s.scalar_host) (void)hipHostFree(s.scalar_host);
    s.scalar_dev = nullptr;
    s.cstream = nullptr;
    s.ev_main = s.ev_comm = nullptr;
    s.scalar_host = nullptr;
}

int make_side_objects(comm_state &s)
{
    hipError_t e = jh_device_malloc(jh_ctx().device, (void **)&s.scalar_dev, sizeof(double) * 64);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s.cstream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev_main, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev_comm, hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void **)&s.scalar_host, sizeof(double) * 8, hipHostMallocDefault);
    if (e != hipSuccess) {
        free_side_objects(s);                            // nothing of a half-built set stays behind
        return jh_fail(e == hipErrorOutOfMemory ? JH_ERR_NOMEM : JH_ERR_HIP, "jh_comm: side objects of the communicator: %s", hipGetErrorString(e));
    }
    return JH_OK;
}

// frees the CURRENT context's communicator state (not the team object)
int drop_state(comm_state &s)
{
    ncclResult_t r = ncclSuccess;
    if (jh_ctx().ready) (void)hipStreamSynchronize(jh_ctx().stream);
    if (s.cstream) (void)hipStreamSynchronize(s.cstream);
    if (s.comm) r = g_api.CommDestroy(s.comm);
    free_side_objects(s);
    s = comm_state();
    if (r != ncclSuccess) return jh_fail(JH_ERR_COMM, "ncclCommDestroy: %s", g_api.GetErrorString(r));
    return JH_OK;
}

// a team on ONE device: every member's buffer becomes the sum of all members' buffers, members added in rank order
struct team_ptrs { void *p[TEAM_LOCAL_MAX]; };
template <typename S>
__global__ __launch_bounds__(256) void k_team_sum(team_ptrs t, int n, size_t nscal)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nscal; i += (size_t)gridDim.x * 256) {
        S acc = ((const S *)t.p[0])[i];
        for (int k = 1; k < n; k++) acc = acc + ((const S *)t.p[k])[i];
        for (int k = 0; k < n; k++) ((S *)t.p[k])[i] = acc;
    }
}

__global__ void k_pack_normsq_err(const double *normsq, const unsigned *err_word, double *out)
{
    out[0] = *normsq;
    out[1] = *err_word ? 1.0 : 0.0;
}

// a collective of one member: inside an open group it is recorded (one-device team) or enqueued (RCCL); outside a group a team
// member cannot proceed alone
int member_allreduce(comm_state &s, void *p, size_t nscal, bool f32, hipStream_t stream, const char *who)
{
    if (!s.team) {
        JH_CHECK_NCCL(g_api.AllReduce(p, p, nscal, f32 ? ncclFloat32 : ncclFloat64, ncclSum, s.comm, stream));
        return JH_OK;
    }
    team_t &t = *s.team;
    if (!t.open)
        return jh_fail(JH_ERR_STATE, "%s: this context is member %d of a single-process team of %d: issue the call once per member between "
                       "jh_comm_group_begin and jh_comm_group_end", who, s.rank, s.nranks);
    if (!t.local) {
        JH_CHECK_NCCL(g_api.AllReduce(p, p, nscal, f32 ? ncclFloat32 : ncclFloat64, ncclSum, s.comm, stream));
        t.nrec++;
        return JH_OK;
    }
    JH_REQUIRE(t.nrec < JH_MAX_CTX, "%s: too many calls in one group", who);
    t.recs[t.nrec++] = {jh_ctx().id, p, nscal, f32, stream};
    return JH_OK;
}

// the recorded calls of a one-device team, executed: they come in rounds of n (one per member, any member order inside a round)
int run_local_group(team_t &t)
{
    if (t.nrec % t.n != 0)
        return jh_fail(JH_ERR_INVALID, "jh_comm_group_end: %d calls for a team of %d members (every member must issue every collective)", t.nrec, t.n);
    for (int r0 = 0; r0 < t.nrec; r0 += t.n) {
        team_ptrs tp;
        hipStream_t streams[TEAM_LOCAL_MAX];
        bool seen[TEAM_LOCAL_MAX] = {};
        for (int k = 0; k < t.n; k++) {
            const team_t::rec &rc = t.recs[r0 + k];
            const int rank = g_cs[jh_ctx_slot(rc.ctx)].rank;
            JH_REQUIRE(rank >= 0 && rank < t.n && !seen[rank], "jh_comm_group_end: a member issued collective %d twice", r0 / t.n);
            JH_REQUIRE(rc.nscal == t.recs[r0].nscal && rc.f32 == t.recs[r0].f32, "jh_comm_group_end: the members' calls of collective %d differ in size or type", r0 / t.n);
            seen[rank] = true;
            tp.p[rank] = rc.p;
            streams[rank] = rc.stream;
        }
        const size_t nscal = t.recs[r0].nscal;
        if (nscal == 0) continue;
        // member 0's stream does the sum after every member's stream has produced its buffer; the others wait for it
        for (int k = 1; k < t.n; k++) {
            comm_state &s = g_cs[jh_ctx_slot(t.ctx[k])];
            JH_CHECK_HIP(hipEventRecord(s.ev_main, streams[k]));
            JH_CHECK_HIP(hipStreamWaitEvent(streams[0], s.ev_main, 0));
        }
        size_t grid = (nscal + 255) / 256;
        if (grid > 16384) grid = 16384;
        if (t.recs[r0].f32) hipLaunchKernelGGL((k_team_sum<float>), dim3((unsigned)grid), dim3(256), 0, streams[0], tp, t.n, nscal);
        else hipLaunchKernelGGL((k_team_sum<double>), dim3((unsigned)grid), dim3(256), 0, streams[0], tp, t.n, nscal);
        JH_CHECK_HIP(hipGetLastError());
        comm_state &s0 = g_cs[jh_ctx_slot(t.ctx[0])];
        JH_CHECK_HIP(hipEventRecord(s0.ev_comm, streams[0]));
        for (int k = 1; k < t.n; k++) JH_CHECK_HIP(hipStreamWaitEvent(streams[k], s0.ev_comm, 0));
    }
    return JH_OK;
}

}  // namespace

extern "C" {

// Side-effect-free probe: can this process reach an RCCL at all?  (ncclGetUniqueId, by contrast, starts a bootstrap root -- a
// listening socket and a thread -- per call.)
int jh_comm_available(void) { return load_rccl(); }

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
    comm_state &s = cs();
    if (s.alive) return jh_fail(JH_ERR_STATE, "jh_comm_init_rank: communicator already initialised (rank %d of %d)", s.rank, s.nranks);
    JH_TRY(load_rccl());
    JH_CHECK_HIP(hipSetDevice(jh_ctx().device));
    ncclUniqueId id;
    memcpy(&id, id128, NCCL_UNIQUE_ID_BYTES);
    JH_CHECK_NCCL(g_api.CommInitRank(&s.comm, nranks, id, rank));
    if (const int st = make_side_objects(s); st != JH_OK) {
        (void)g_api.CommDestroy(s.comm);
        s = comm_state();
        return st;
    }
    s.nranks = nranks;
    s.rank = rank;
    s.alive = true;
    return JH_OK;
}

// ONE process, n contexts: member k of the team is contexts[k].  All on distinct devices (RCCL, ncclCommInitAll) or all on one
// device (a device-side sum; RCCL refuses two ranks on a device).
int jh_comm_init_all(int n, const int *contexts)
{
    JH_REQUIRE(contexts && n >= 1 && n <= JH_MAX_CTX, "jh_comm_init_all: need 1..%d contexts", JH_MAX_CTX);
    int devs[JH_MAX_CTX];
    bool distinct = true, same = true;
    for (int k = 0; k < n; k++) {
        jh_context *c = jh_ctx_by_id(contexts[k]);
        JH_REQUIRE(c && c->ready, "jh_comm_init_all: no context %d", contexts[k]);
        JH_REQUIRE(!g_cs[jh_ctx_slot(contexts[k])].alive, "jh_comm_init_all: context %d already has a communicator", contexts[k]);
        devs[k] = c->device;
        for (int j = 0; j < k; j++) {
            JH_REQUIRE(contexts[j] != contexts[k], "jh_comm_init_all: context %d listed twice", contexts[k]);
            if (devs[j] == devs[k]) distinct = false;
        }
        if (devs[k] != devs[0]) same = false;
    }
    const bool local = n > 1 && same;
    if (n > 1 && !distinct && !same)
        return jh_fail(JH_ERR_INVALID, "jh_comm_init_all: the contexts must sit on pairwise distinct devices (RCCL) or all on one device");
    JH_REQUIRE(!local || n <= TEAM_LOCAL_MAX, "jh_comm_init_all: at most %d contexts of one device form a team", TEAM_LOCAL_MAX);
    int before = -1;
    (void)jh_context_current(&before, nullptr);
    ncclComm_t comms[JH_MAX_CTX] = {};
    if (!local) {
        JH_TRY(load_rccl());
        JH_CHECK_NCCL(g_api.CommInitAll(comms, n, devs));
    }
    team_t *t = new team_t();
    t->n = n;
    t->local = local;
    int st = JH_OK, built = 0;
    for (int k = 0; k < n && st == JH_OK; k++) {
        t->ctx[k] = contexts[k];
        st = jh_context_use(contexts[k]);
        if (st != JH_OK) break;
        comm_state &s = g_cs[jh_ctx_slot(contexts[k])];
        st = make_side_objects(s);
        if (st != JH_OK) break;
        s.comm = comms[k];
        s.nranks = n;
        s.rank = k;
        s.team = t;
        s.alive = true;
        built = k + 1;
    }
    if (st != JH_OK) {
        // roll back: the members already set up (their communicator included), the communicators nobody owns yet, the team object;
        // the message of the failing call is kept
        char msg[512];
        snprintf(msg, sizeof(msg), "%s", jh_last_error());
        for (int k = 0; k < built; k++)
            if (jh_context_use(contexts[k]) == JH_OK) (void)drop_state(g_cs[jh_ctx_slot(contexts[k])]);
        for (int k = built; k < n; k++)
            if (comms[k]) (void)g_api.CommDestroy(comms[k]);
        delete t;
        if (before >= 0) (void)jh_context_use(before);
        return jh_fail(st, "jh_comm_init_all: %s (nothing of the team is left behind)", msg);
    }
    if (before >= 0) JH_TRY(jh_context_use(before));
    return JH_OK;
}

int jh_comm_group_begin(void)
{
    JH_TRY(jh_require_ready());
    comm_state &s = cs();
    if (!s.alive || !s.team) return jh_fail(JH_ERR_STATE, "jh_comm_group_begin: the current context is not a member of a single-process team (jh_comm_init_all)");
    JH_REQUIRE(!s.team->open, "jh_comm_group_begin: a group is already open");
    s.team->open = true;
    s.team->nrec = 0;
    if (!s.team->local) JH_CHECK_NCCL(g_api.GroupStart());
    return JH_OK;
}

int jh_comm_group_end(void)
{
    JH_TRY(jh_require_ready());
    comm_state &s = cs();
    if (!s.alive || !s.team || !s.team->open) return jh_fail(JH_ERR_STATE, "jh_comm_group_end: no group is open on the current context's team");
    team_t &t = *s.team;
    t.open = false;
    if (!t.local) {
        JH_CHECK_NCCL(g_api.GroupEnd());
        if (t.nrec % t.n != 0)
            return jh_fail(JH_ERR_INVALID, "jh_comm_group_end: %d calls for a team of %d members (every member must issue every collective)", t.nrec, t.n);
        return JH_OK;
    }
    const int before = jh_ctx().id;
    const int st = run_local_group(t);
    (void)jh_context_use(before);
    return st;
}

// the CURRENT context's communicator; a team dies as a whole
int jh_comm_destroy(void)
{
    comm_state &s = cs();
    if (!s.alive) return JH_OK;
    if (!s.team) return drop_state(s);
    team_t *t = s.team;
    const int before = jh_ctx().id;
    int st = JH_OK;
    for (int k = 0; k < t->n; k++) {
        if (!jh_ctx_by_id(t->ctx[k]) || !g_cs[jh_ctx_slot(t->ctx[k])].alive) continue;
        (void)jh_context_use(t->ctx[k]);
        const int r = drop_state(g_cs[jh_ctx_slot(t->ctx[k])]);
        if (st == JH_OK) st = r;
    }
    delete t;
    if (jh_ctx_by_id(before)) (void)jh_context_use(before);
    return st;
}

int jh_comm_exists(int *yes)
{
    if (yes) *yes = cs().alive ? (cs().team ? 2 : 1) : 0;   // 2: member of a single-process team
    return JH_OK;
}

int jh_comm_info(int *nranks, int *rank)
{
    const comm_state &s = cs();
    if (nranks) *nranks = s.alive ? s.nranks : 1;
    if (rank) *rank = s.alive ? s.rank : 0;
    return JH_OK;
}

// in-place sum of a replicated vector over all ranks (the adjoint accumulate of a row-partitioned tall operator)
int jh_comm_allreduce_sum(jh_bvec *v)
{
    JH_TRY(jh_enter(v));
    JH_REQUIRE(v, "jh_comm_allreduce_sum: null vector");
    comm_state &s = cs();
    if (!s.alive) return jh_fail(JH_ERR_STATE, "jh_comm_allreduce_sum: call jh_comm_init_rank first");
    if (v->length == 0 && !s.team) return JH_OK;
    const bool f32 = (v->dtype == JH_F32 || v->dtype == JH_C32);
    const size_t count = (size_t)v->length * (jh_dtype_complex(v->dtype) ? 2 : 1);
    return member_allreduce(s, v->data, count, f32, jh_ctx().stream, "jh_comm_allreduce_sum");
}

// The same sum restricted to the elements [first_elem, first_elem + count) of v, ASYNCHRONOUS with respect to the library
// stream: it runs on the communicator's own stream, ordered after everything enqueued on the library stream so far (the kernel
// that has just produced that range), while kernels enqueued on the library stream afterwards (the next range) run concurrently.
// jh_comm_join makes the library stream wait for every ranged all-reduce enqueued so far.  All ranks must enqueue the same
// ranges in the same order.
int jh_comm_allreduce_sum_range(jh_bvec *v, int64_t first_elem, int64_t count)
{
    JH_TRY(jh_enter(v));
    JH_REQUIRE(v, "jh_comm_allreduce_sum_range: null vector");
    comm_state &s = cs();
    if (!s.alive) return jh_fail(JH_ERR_STATE, "jh_comm_allreduce_sum_range: call jh_comm_init_rank first");
    JH_REQUIRE(first_elem >= 0 && count >= 0 && first_elem + count <= v->length,
               "jh_comm_allreduce_sum_range: elements [%lld, %lld) outside the vector (%lld elements)", (long long)first_elem,
               (long long)(first_elem + count), (long long)v->length);
    if (count == 0 && !s.team) return JH_OK;
    const bool f32 = (v->dtype == JH_F32 || v->dtype == JH_C32);
    const size_t nscal = (size_t)count * (jh_dtype_complex(v->dtype) ? 2 : 1);
    void *p = v->ptr(first_elem);
    JH_CHECK_HIP(hipEventRecord(s.ev_main, jh_ctx().stream));
    JH_CHECK_HIP(hipStreamWaitEvent(s.cstream, s.ev_main, 0));
    return member_allreduce(s, p, nscal, f32, s.cstream, "jh_comm_allreduce_sum_range");
}

int jh_comm_join(void)
{
    JH_TRY(jh_require_ready());
    comm_state &s = cs();
    if (!s.alive) return jh_fail(JH_ERR_STATE, "jh_comm_join: call jh_comm_init_rank first");
    if (s.team && s.team->open) return jh_fail(JH_ERR_STATE, "jh_comm_join: close the group first (jh_comm_group_end)");
    JH_CHECK_HIP(hipEventRecord(s.ev_comm, s.cstream));
    JH_CHECK_HIP(hipStreamWaitEvent(jh_ctx().stream, s.ev_comm, 0));
    return JH_OK;
}

// Sum over all ranks of the deferred ||u||^2 accumulator (jh_normsq_reset / jh_blockop_bidiag_step_range with normsq == NULL),
// on the exchange stream behind the ranged all-reduces: when it returns, the kernels of the step, every ranged all-reduce
// enqueued before it and the scalar sum are complete -- the ONE host synchronisation of a pipelined distributed step.
int jh_comm_allreduce_normsq(double *out)
{
    JH_TRY(jh_require_ready());
    JH_REQUIRE(out, "jh_comm_allreduce_normsq: null output");
    comm_state &s = cs();
    if (!s.alive) return jh_fail(JH_ERR_STATE, "jh_comm_allreduce_normsq: call jh_comm_init_rank first");
    if (s.team) return jh_fail(JH_ERR_STATE, "jh_comm_allreduce_normsq: in a single-process team the host adds the members' accumulators itself (jh_normsq_read on every context)");
    double *slot = jh_ctx().red_dev + JH_NORMSQ_SLOT;
    JH_CHECK_HIP(hipEventRecord(s.ev_main, jh_ctx().stream));
    JH_CHECK_HIP(hipStreamWaitEvent(s.cstream, s.ev_main, 0));
    // ONE collective of two doubles: [0] the sum over ranks of ||u||^2 (the local accumulator stays local), [1] the number of ranks
    // whose chained step raised its sticky "a hand-off poll expired" word -- so EVERY rank fails together, in the call that
    // consumes w, when one rank's step went on with an invalid partial sum
    hipLaunchKernelGGL(k_pack_normsq_err, dim3(1), dim3(1), 0, s.cstream, (const double *)slot, (const unsigned *)(jh_ctx().red_dev + JH_CHAIN_ERR_SLOT),
                       s.scalar_dev);
    JH_CHECK_HIP(hipGetLastError());
    JH_CHECK_NCCL(g_api.AllReduce(s.scalar_dev, s.scalar_dev, 2, ncclFloat64, ncclSum, s.comm, s.cstream));
    JH_CHECK_HIP(hipMemcpyAsync(s.scalar_host, s.scalar_dev, 2 * sizeof(double), hipMemcpyDeviceToHost, s.cstream));
    JH_CHECK_HIP(hipStreamSynchronize(s.cstream));
    *out = s.scalar_host[0];
    if (s.scalar_host[1] != 0.0) {
        unsigned one = 1;
        memcpy(jh_ctx().red_host + 3, &one, sizeof(one));                                               // jh_chain_err_check clears the device word and reports
        return jh_chain_err_check();
    }
    return JH_OK;
}

// batched scalar all-reduce (range-side dot / norm^2 / extrema partials); op: 0 sum, 1 max, 2 min.  Synchronises.
int jh_comm_allreduce_scalars(double *values, int n, int op)
{
    JH_TRY(jh_require_ready());
    JH_REQUIRE(values && n >= 1 && n <= 64, "jh_comm_allreduce_scalars: need 1..64 values");
    JH_REQUIRE(op >= 0 && op <= 2, "jh_comm_allreduce_scalars: op must be 0 (sum), 1 (max) or 2 (min)");
    comm_state &s = cs();
    if (!s.alive) return jh_fail(JH_ERR_STATE, "jh_comm_allreduce_scalars: call jh_comm_init_rank first");
    if (s.team) return jh_fail(JH_ERR_STATE, "jh_comm_allreduce_scalars: in a single-process team the host holds every member's scalars and combines them itself");
    hipStream_t st = jh_ctx().stream;
    JH_CHECK_HIP(hipMemcpyAsync(s.scalar_dev, values, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
    const ncclRedOp_t red = op == 0 ? ncclSum : (op == 1 ? ncclMax : ncclMin);
    JH_CHECK_NCCL(g_api.AllReduce(s.scalar_dev, s.scalar_dev, (size_t)n, ncclFloat64, red, s.comm, st));
    JH_CHECK_HIP(hipMemcpyAsync(values, s.scalar_dev, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
    JH_CHECK_HIP(hipStreamSynchronize(st));
    return JH_OK;
}

}  // extern "C"

// ---- a TEAM's forward / adjoint / normal operator behind ONE call each (round 4) -------------------------------------------------------
// The exchange step of the row partition (src/Jets.jl:1045-1053 summed over the members' rows) is issued range by range: every
// member's kernel for a range, then the members' all-reduces of that range in one group, while the next range's kernels compute.
// Driven from the host language that is 4 x (n context switches + n kernel calls + group begin + n all-reduces + group end) + n
// joins -- about 75 ABI calls per forward + adjoint pair at 8 members, each through the binding's dispatch.  Here the member loop
// runs in C: one call per operator application, whatever the host language costs per call.
static int team_check(const char *who, int n, const jh_blockop *const *ops, const void *const *a, const void *const *b)
{
    JH_REQUIRE(n >= 1 && n <= JH_MAX_CTX && ops && a && b, "%s: need 1..%d members", who, JH_MAX_CTX);
    for (int k = 0; k < n; k++) JH_REQUIRE(ops[k] && a[k] && b[k], "%s: null handle of member %d", who, k);
    return JH_OK;
}

static int team_ranged(const char *who, int n, const jh_blockop *const *ops, jh_bvec *const *outs, const jh_bvec *const *ins, int nranges, bool normal)
{
    JH_TRY(team_check(who, n, ops, (const void *const *)outs, (const void *const *)ins));
    JH_REQUIRE(nranges >= 1 && nranges <= 64, "%s: 1..64 exchange ranges", who);
    for (int k = 0; k < n; k++) {                                          // every member in its own context of ONE team, before anything is enqueued
        JH_TRY(jh_enter(ops[k], outs[k], ins[k]));
        comm_state &s = cs();
        JH_REQUIRE(s.alive && s.team && s.nranks == n && s.rank == k, "%s: the handles of member %d must live in member %d's context of a team of %d (jh_comm_init_all)",
                   who, k, k, n);
        JH_REQUIRE(outs[k]->length == outs[0]->length && outs[k]->dtype == outs[0]->dtype, "%s: member %d's domain vector differs in length or element type", who, k);
    }
    const int64_t len = outs[0]->length;
    int64_t step = (len + nranges - 1) / nranges;
    step = (step + 16383) / 16384 * 16384;                                 // range bounds on 64 KiB boundaries (as the solvers' exchange)
    for (int64_t lo = 0; lo < len; lo += step) {
        const int64_t cnt = lo + step < len ? step : len - lo;
        for (int k = 0; k < n; k++)
            JH_TRY(normal ? jh_blockop_normal_mul_range(ops[k], outs[k], ins[k], lo, cnt) : jh_blockop_mul_adj_range(ops[k], outs[k], ins[k], lo, cnt));
        JH_TRY(jh_enter(outs[0]));
        JH_TRY(jh_comm_group_begin());
        int st = JH_OK;
        for (int k = 0; k < n && st == JH_OK; k++) st = jh_comm_allreduce_sum_range(outs[k], lo, cnt);
        (void)jh_enter(outs[0]);
        const int st2 = jh_comm_group_end();
        JH_TRY(st);
        JH_TRY(st2);
    }
    for (int k = 0; k < n; k++) {
        JH_TRY(jh_enter(outs[k]));
        JH_TRY(jh_comm_join());
    }
    return JH_OK;
}

extern "C" {

// d_k = A_k m_k on every member (1015-1031: block rows are independent, no exchange); returns after enqueue
int jh_team_mul(int n, const jh_blockop *const *ops, jh_bvec *const *ds, const jh_bvec *const *ms)
{
    JH_TRY(team_check("jh_team_mul", n, ops, (const void *const *)ds, (const void *const *)ms));
    for (int k = 0; k < n; k++) JH_TRY(jh_blockop_mul(ops[k], ds[k], ms[k]));
    return JH_OK;
}

// every member's m_k = sum over ALL members' rows of A_i' d_i (1045-1053): local ordered sums range by range, the grouped all-reduce of a
// finished range under the next range's kernels, the library streams waiting for the exchange by event (no host synchronisation)
int jh_team_mul_adj(int n, const jh_blockop *const *ops, jh_bvec *const *ms, const jh_bvec *const *ds, int nranges)
{
    return team_ranged("jh_team_mul_adj", n, ops, ms, ds, nranges, false);
}

// every member's y_k = (sum over all members of A_i'A_i) m: the fused normal operator (530-534 over (A', A)) exchanged the same way
int jh_team_normal_mul(int n, const jh_blockop *const *ops, jh_bvec *const *ys, const jh_bvec *const *ms, int nranges)
{
    return team_ranged("jh_team_normal_mul", n, ops, ys, ms, nranges, true);
}

}  // extern "C"
