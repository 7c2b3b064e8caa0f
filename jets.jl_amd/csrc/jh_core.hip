// jh_core.hip -- context, device block vectors (slabs), copies, events.  gfx950 only.
#include "jh_internal.h"
#include <algorithm>
#include <initializer_list>

static thread_local char g_err[512] = "";

int jh_fail(int status, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return status;
}

// ---- the context table ----------------------------------------------------------------------------------------------------
#include <map>
#include <mutex>
static std::mutex g_ctx_mutex;                           // creation / destruction only; a handle is used by one host thread at a time
static jh_context *g_ctxs[JH_MAX_CTX] = {};
static int g_slot_gen[JH_MAX_CTX] = {};                  // bumped when a slot's context dies: ids are never reused
static int g_first_ctx = -1;                             // a thread that never chose uses the first context created
static thread_local int t_cur_ctx = -1;

jh_context *jh_ctx_by_id(int id)
{
    if (id < 0) return nullptr;
    jh_context *c = g_ctxs[jh_ctx_slot(id)];
    return (c && c->id == id) ? c : nullptr;
}

// ---- slab cache ---------------------------------------------------------------------------------------------------------------------
// hipMalloc of a range-sized slab is not cheap on this machine: 64 GiB takes 2-6 SECONDS whenever the runtime has to go to the driver
// for it, which it does unpredictably (profiles/exp_r03_alloc_cost.txt; 8 GiB: 0.2 ms).  A caller in the reference's style allocates
// such temporaries all the time (`A*m` returns a fresh vector, `zeros(range(A))` per stage, a solver's copy of b), so the slabs of
// destroyed vectors are kept -- per device, exact size, as few bytes as possible given back when the cap is reached -- and handed to the next
// jh_bvec_create of that size.  jh_trim() / knob "slab_cache" = 0 / an allocation that does not fit release them.
namespace {
struct cached_slab { void *p; size_t bytes; };
std::mutex g_slab_mutex;
// Round 4: WHICH cached slab an allocation gets.  On this chip the time to WRITE a 64 GiB slab is a property of the slab: the library's
// own fill runs at 7.2 TB/s into some and 6.4 into others (consecutive processes, one box: profiles/exp_r04_write_probe.txt), and the tall
// forward follows it (20.6-21.5 ms into a fast-write slab, 23.8-24.8 into a slow one, whichever slab holds the coefficients); the
// fast-write slabs read about 3 % slower (norm: 10.26 against 9.97 ms).  So a slab of 4 GiB or more is PROBED once when it enters the
// cache (two fills of the freed memory, the second one timed: 20 ms per 64 GiB, at destroy time) and an allocation that says what it
// is for (knob alloc_role: 1 = an operator's output, 2 = data that is written once and read from then on) takes the cached slab of
// its size with the fastest / the slowest recorded fill; without a role (0) the most recently freed one, as before.
constexpr size_t SLAB_PROBE_MIN = (size_t)4 << 30;
std::map<void *, double> g_slab_fill_ms_per_gib;           // probe records, by slab (erased when the slab goes back to the driver)
std::atomic<int64_t> g_last_alloc_choice{-1};              // 100 * candidates + rank of the chosen one by fill time (0 = fastest); -1: no choice was made

__global__ __launch_bounds__(256) void k_probe_fill(uint4 *__restrict__ p, int64_t nvec)
{
    typedef unsigned V __attribute__((ext_vector_type(4)));
    const V z = {0u, 0u, 0u, 0u};
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256)
        __builtin_nontemporal_store(z, reinterpret_cast<V *>(p) + v);
}

// (the device is current; the slab is dead memory that no stream touches any more: jh_slab_free has synchronised the device)
double slab_probe_ms_per_gib(void *p, size_t bytes, hipStream_t st)
{
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        (void)hipGetLastError();
        return 0.0;
    }
    const int64_t nvec = (int64_t)(bytes / 16);
    int64_t grid = (nvec + 255) / 256;
    if (grid > ((int64_t)1 << 23)) grid = (int64_t)1 << 23;
    double out = 0.0;
    hipLaunchKernelGGL(k_probe_fill, dim3((unsigned)grid), dim3(256), 0, st, (uint4 *)p, nvec);          // (page tables, caches)
    if (hipEventRecord(e0, st) == hipSuccess) {
        hipLaunchKernelGGL(k_probe_fill, dim3((unsigned)grid), dim3(256), 0, st, (uint4 *)p, nvec);
        float ms = 0.f;
        if (hipEventRecord(e1, st) == hipSuccess && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f)
            out = (double)ms / ((double)bytes / (double)((size_t)1 << 30));
    }
    (void)hipGetLastError();
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return out;
}
void slab_forget(void *p) { g_slab_fill_ms_per_gib.erase(p); }   // under the lock
std::map<int, std::vector<cached_slab>> g_slabs;            // device -> slabs, oldest first
std::atomic<int> g_slab_cache_on{1};
std::atomic<int64_t> g_slab_free_floor_mib{8192};           // knob slab_free_floor_mib
constexpr size_t SLAB_CACHE_MIN = (size_t)16 << 20;          // from 16 MiB on: a thousand 64 MiB blocks re-allocated after a free cost 3.6 ms EACH (profiles/exp_r03_alloc_cost.txt)

size_t slab_cap_bytes()                                       // the cache never holds the last 32 GiB of the device
{
    static const size_t cap = [] {                            // (initialised once, thread-safely)
        size_t fr = 0, tot = 0;
        return (hipMemGetInfo(&fr, &tot) == hipSuccess && tot > ((size_t)64 << 30)) ? tot - ((size_t)32 << 30) : (size_t)32 << 30;
    }();
    return cap;
}

// pick (under the lock) the slab to give back when `need` more bytes are wanted: the largest one that does not overshoot the need, else
// (every slab is bigger) the smallest -- called until the need is covered, this gives back close to the fewest bytes, and re-used
// device memory is cleared by the driver at ~20 GB/s: the bytes given back are what a later allocation pays for
bool slab_pick_victim(std::vector<cached_slab> &v, size_t need, cached_slab *victim)
{
    if (v.empty()) return false;
    size_t under = v.size(), smallest = 0;
    for (size_t k = 0; k < v.size(); k++) {
        if (v[k].bytes <= need && (under == v.size() || v[k].bytes > v[under].bytes)) under = k;
        if (v[k].bytes < v[smallest].bytes) smallest = k;
    }
    const size_t k = under < v.size() ? under : smallest;
    *victim = v[k];
    v.erase(v.begin() + (long)k);
    return true;
}

bool slab_evict_for(int device, size_t need)                   // false when the cache is empty
{
    cached_slab victim{nullptr, 0};
    {
        std::lock_guard<std::mutex> lock(g_slab_mutex);
        if (!slab_pick_victim(g_slabs[device], need, &victim)) return false;
        slab_forget(victim.p);
    }
    (void)hipFree(victim.p);
    return true;
}
}  // namespace

size_t jh_slab_cached_bytes(int device)
{
    std::lock_guard<std::mutex> lock(g_slab_mutex);
    size_t sum = 0;
    for (const cached_slab &c : g_slabs[device]) sum += c.bytes;
    return sum;
}

void jh_slab_trim(int device)                                  // device < 0: every device
{
    std::vector<cached_slab> out;
    {
        std::lock_guard<std::mutex> lock(g_slab_mutex);
        if (device >= 0) out.swap(g_slabs[device]);
        else
            for (auto &kv : g_slabs) {
                out.insert(out.end(), kv.second.begin(), kv.second.end());
                kv.second.clear();
            }
    }
    {
        std::lock_guard<std::mutex> lock(g_slab_mutex);
        for (const cached_slab &c : out) slab_forget(c.p);
    }
    for (const cached_slab &c : out) (void)hipFree(c.p);   // (hipFree takes a pointer of any device)
}

hipError_t jh_slab_alloc(int device, size_t bytes, void **out, int role)
{
    if (bytes >= SLAB_CACHE_MIN) {
        std::lock_guard<std::mutex> lock(g_slab_mutex);
        std::vector<cached_slab> &v = g_slabs[device];
        size_t pick = v.size();
        g_last_alloc_choice.store(-1);
        if (role != 0 && bytes >= SLAB_PROBE_MIN) {          // the probed candidates of this size: fastest fill for an output, slowest for read-mostly data
            std::vector<std::pair<double, size_t>> cand;
            for (size_t k = 0; k < v.size(); k++)
                if (v[k].bytes == bytes) {
                    auto it = g_slab_fill_ms_per_gib.find(v[k].p);
                    if (it != g_slab_fill_ms_per_gib.end() && it->second > 0.0) cand.push_back({it->second, k});
                }
            if (cand.size() >= 2) {
                std::sort(cand.begin(), cand.end());
                const size_t rank = role == 1 ? 0 : cand.size() - 1;
                pick = cand[rank].second;
                g_last_alloc_choice.store((int64_t)(100 * cand.size() + rank));
            }
        }
        if (pick == v.size())
            for (size_t k = v.size(); k-- > 0;)              // the most recently freed slab of that size first
                if (v[k].bytes == bytes) { pick = k; break; }
        if (pick < v.size()) {
            *out = v[pick].p;
            v.erase(v.begin() + (long)pick);
            return hipSuccess;
        }
    }
    return jh_device_malloc(device, out, bytes);
}

// hipMalloc for everything the library allocates: what the cache holds is free memory -- when the driver says no, give back the slab
// that covers the shortfall with the fewest bytes and ask again
hipError_t jh_device_malloc(int device, void **out, size_t bytes)
{
    hipError_t e = hipMalloc(out, bytes);
    while (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        size_t fr = 0, tot = 0;
        const size_t need = (hipMemGetInfo(&fr, &tot) == hipSuccess && fr < bytes) ? bytes - fr : bytes;
        if (!slab_evict_for(device, need)) break;
        e = hipMalloc(out, bytes);
    }
    return e;
}

void jh_slab_free(int device, void *p, size_t bytes, hipStream_t st)
{
    if (!p) return;
    if (g_slab_cache_on.load() && bytes >= SLAB_CACHE_MIN && bytes <= slab_cap_bytes()) {
        // hipFree waits for the whole device; a slab that goes to the cache instead must get the same guarantee before its next owner
        // zero-fills or writes it: the caller has waited for the owning context's stream only, and work of OTHER streams may still touch
        // the slab -- ranged all-reduces on a communicator's exchange stream (an early error return of a solver destroys its work
        // vectors under them), a stream the application installed with jh_set_stream and replaced since, torch / RCCL streams on a
        // wrapped vector.  Microseconds on an idle device, against the seconds the re-used slab saves.  (The device is current: both
        // callers come through jh_quiesce_scope / the vector's own context.)
        // While `st` is being captured into a graph neither is possible (the synchronisation would invalidate the capture, the probe's
        // fills would land in the graph): such a slab goes back to the driver like a small one (round-4 advisor finding).
        if (st) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
                (void)hipGetLastError();
                {
                    std::lock_guard<std::mutex> lock(g_slab_mutex);
                    slab_forget(p);
                }
                (void)hipFree(p);
                return;
            }
        }
        (void)hipDeviceSynchronize();
        if (bytes >= SLAB_PROBE_MIN && st) {                  // how fast can this slab be written?  (once per slab: see g_slab_fill_ms_per_gib)
            bool known;
            {
                std::lock_guard<std::mutex> lock(g_slab_mutex);
                known = g_slab_fill_ms_per_gib.count(p) != 0;
            }
            if (!known) {
                const double rate = slab_probe_ms_per_gib(p, bytes, st);
                std::lock_guard<std::mutex> lock(g_slab_mutex);
                g_slab_fill_ms_per_gib[p] = rate;
            }
        }
        std::vector<cached_slab> evict;
        {
            std::lock_guard<std::mutex> lock(g_slab_mutex);
            std::vector<cached_slab> &v = g_slabs[device];
            size_t held = bytes;
            for (const cached_slab &c : v) held += c.bytes;
            cached_slab victim{nullptr, 0};
            while (held > slab_cap_bytes() && slab_pick_victim(v, held - slab_cap_bytes(), &victim)) {
                held -= victim.bytes;
                slab_forget(victim.p);
                evict.push_back(victim);
            }
            v.push_back(cached_slab{p, bytes});
            // ... and it never holds the device's LAST 8 GiB: live vectors plus cached slabs must leave that much to whoever allocates
            // without going through jh_device_malloc's evict-and-retry (RCCL's buffers, torch in the same process).  Cached memory is
            // still allocated as far as the driver can tell, so `free` below does not count it.  (Not 32 GiB here: four 64 GiB vectors
            // -- coefficients, d, a second range vector, a solver's copy of b -- are 256 of the 288 GiB, and a floor of 32 sent the
            // fourth back to the driver after every solve: 2 s per lsqr(A, b) at the headline size, profiles/walkthrough_r04_headline.txt.)
            size_t fr = 0, tot = 0;
            const size_t floor_bytes = (size_t)g_slab_free_floor_mib.load() << 20;
            if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
                for (const cached_slab &c : evict) fr += c.bytes;          // (what the cap has just sent back is as good as free)
            } else {
                tot = 0;
            }
            if (tot > ((size_t)64 << 30) && fr < floor_bytes) {
                size_t need = floor_bytes - fr;
                while (need > 0 && slab_pick_victim(v, need, &victim)) {
                    slab_forget(victim.p);
                    evict.push_back(victim);
                    need = victim.bytes >= need ? 0 : need - victim.bytes;
                }
            }
        }
        for (const cached_slab &c : evict) (void)hipFree(c.p);
        return;
    }
    {
        std::lock_guard<std::mutex> lock(g_slab_mutex);
        slab_forget(p);
    }
    (void)hipFree(p);
}
int64_t jh_slab_probed(int device)
{
    std::lock_guard<std::mutex> lock(g_slab_mutex);
    int64_t n = 0;
    for (const cached_slab &c : g_slabs[device]) n += g_slab_fill_ms_per_gib.count(c.p) ? 1 : 0;
    return n;
}
int64_t jh_slab_last_choice() { return g_last_alloc_choice.load(); }

jh_quiesce_scope::jh_quiesce_scope(int ctx)
{
    jh_context *c = jh_ctx_by_id(ctx);
    if (!c || !c->ready) return;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != c->device) {
        (void)hipSetDevice(c->device);
        switched = prev >= 0;
    }
    (void)hipStreamSynchronize(c->stream);                   // stream-ordered work may still reference what is about to be freed
}

jh_quiesce_scope::~jh_quiesce_scope()
{
    if (switched) (void)hipSetDevice(prev);
}

jh_context &jh_ctx()
{
    static jh_context none;                              // never ready: jh_require_ready() reports "jh_init has not been called"
    jh_context *c = jh_ctx_by_id(t_cur_ctx >= 0 ? t_cur_ctx : g_first_ctx);
    return c ? *c : none;
}

static int ctx_use(int id)
{
    jh_context *c = jh_ctx_by_id(id);
    if (!c || !c->ready) return jh_fail(JH_ERR_STATE, "libjetship: context %d does not exist (destroyed, or its handle outlived jh_shutdown)", id);
    t_cur_ctx = id;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != c->device) JH_CHECK_HIP(hipSetDevice(c->device));
    return JH_OK;
}

int jh_enter_ids(const int *ids, int n)
{
    int first = -1;
    for (int k = 0; k < n; k++) {
        if (ids[k] < 0) continue;                        // a null handle: the entry point reports it
        if (first < 0) first = ids[k];
        else if (ids[k] != first)
            return jh_fail(JH_ERR_INVALID, "libjetship: the handles of this call live in different contexts (%d and %d; devices %d and %d)", first,
                           ids[k], jh_ctx_by_id(first) ? jh_ctx_by_id(first)->device : -1, jh_ctx_by_id(ids[k]) ? jh_ctx_by_id(ids[k])->device : -1);
    }
    if (first < 0) return jh_require_ready();
    if (first == t_cur_ctx) return jh_require_ready();   // the common case: nothing to switch
    return ctx_use(first);
}

static int ctx_create(int device, bool primary, int *id_out)
{
    int n = 0;
    JH_CHECK_HIP(hipGetDeviceCount(&n));
    if (n <= 0) return jh_fail(JH_ERR_HIP, "jh_init: no HIP device visible; libjetship has no CPU fallback");
    JH_REQUIRE(device >= 0 && device < n, "jh_init: device %d out of range (0..%d)", device, n - 1);
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    if (primary)
        for (int k = 0; k < JH_MAX_CTX; k++)
            if (g_ctxs[k] && g_ctxs[k]->primary && g_ctxs[k]->device == device) { *id_out = g_ctxs[k]->id; return JH_OK; }   // idempotent
    int slot = -1;
    for (int k = 0; k < JH_MAX_CTX; k++)
        if (!g_ctxs[k]) { slot = k; break; }
    if (slot < 0) return jh_fail(JH_ERR_STATE, "jh_context_create: all %d context slots are in use", JH_MAX_CTX);
    JH_CHECK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    JH_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return jh_fail(JH_ERR_UNSUPPORTED, "jh_init: device %d is %s; libjetship is built for gfx950 (MI355X) only", device,
                       prop.gcnArchName);
    jh_context *c = new jh_context();
    c->device = device;                                       // (before the first jh_device_malloc: its evict-and-retry looks in THIS device's slab cache)
    auto fail = [&](hipError_t e, const char *what) {
        if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
        if (c->red_dev) (void)hipFree(c->red_dev);
        if (c->red_host) (void)hipHostFree(c->red_host);
        delete c;
        return jh_fail(e == hipErrorOutOfMemory ? JH_ERR_NOMEM : JH_ERR_HIP, "jh_init: %s: %s", what, hipGetErrorString(e));
    };
    c->cu_count = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) return fail(e, "hipStreamCreateWithFlags");
    c->stream = c->own_stream;
    e = jh_device_malloc(c->device, (void **)&c->red_dev, sizeof(double) * 4 * JH_RED_SLOTS);
    if (e != hipSuccess) return fail(e, "hipMalloc");
    e = hipMemsetAsync(c->red_dev, 0, sizeof(double) * 4 * JH_RED_SLOTS, c->own_stream);   // (not the legacy stream: another thread may be capturing)
    if (e == hipSuccess) e = hipStreamSynchronize(c->own_stream);
    if (e != hipSuccess) return fail(e, "hipMemset");
    e = hipHostMalloc((void **)&c->red_host, sizeof(double) * 8, hipHostMallocDefault);
    if (e != hipSuccess) return fail(e, "hipHostMalloc");
    memset(c->red_host, 0, sizeof(double) * 8);
    const int id = slot + JH_MAX_CTX * g_slot_gen[slot];
    c->device = device;
    c->primary = primary;
    c->id = id;
    c->ready = true;
    g_ctxs[slot] = c;
    if (g_first_ctx < 0) g_first_ctx = id;
    *id_out = id;
    return JH_OK;
}

static int ctx_destroy(int id)
{
    jh_context *c = jh_ctx_by_id(id);
    if (!c) return JH_OK;
    const int before = t_cur_ctx;
    (void)ctx_use(id);
    (void)hipStreamSynchronize(c->stream);
    (void)jh_comm_destroy();                             // this context's communicator, if any
    if (c->red_dev) (void)hipFree(c->red_dev);
    if (c->part_dev) (void)hipFree(c->part_dev);
    if (c->scratch_dev) (void)hipFree(c->scratch_dev);
    if (c->chain_sync) (void)hipFree(c->chain_sync);
    if (c->bcast_last.dev) (void)hipFree(c->bcast_last.dev);
    if (c->red_host) (void)hipHostFree(c->red_host);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    {
        bool last_on_device = true;
        for (int k = 0; k < JH_MAX_CTX; k++)
            if (g_ctxs[k] && g_ctxs[k] != c && g_ctxs[k]->device == c->device) last_on_device = false;
        if (last_on_device) jh_slab_trim(c->device);
    }
    {
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        g_ctxs[jh_ctx_slot(id)] = nullptr;
        g_slot_gen[jh_ctx_slot(id)] = (g_slot_gen[jh_ctx_slot(id)] + 1) & 0xFFFFFF;   // a handle of the dead context no longer resolves
        delete c;
        if (g_first_ctx == id) {
            g_first_ctx = -1;
            for (int k = 0; k < JH_MAX_CTX; k++)
                if (g_ctxs[k]) { g_first_ctx = g_ctxs[k]->id; break; }
        }
    }
    t_cur_ctx = (before == id) ? -1 : before;
    if (t_cur_ctx >= 0) (void)ctx_use(t_cur_ctx);
    return JH_OK;
}

// the sticky "hand-off poll expired" word of the chained step, as last copied to red_host[3]
int jh_chain_err_check()
{
    jh_context &c = jh_ctx();
    unsigned bits = 0;
    memcpy(&bits, c.red_host + 3, sizeof(bits));
    if (bits) {
        (void)hipMemsetAsync(c.red_dev + JH_CHAIN_ERR_SLOT, 0, sizeof(double), c.stream);
        memset(c.red_host + 3, 0, sizeof(double));
        return jh_fail(JH_ERR_HIP, "one-pass step: a chained hand-off poll expired (results of that step are invalid); set jh_tune_set(\"step_chain\", 0)");
    }
    return JH_OK;
}

int jh_ensure_partials(int64_t n)
{
    jh_context &c = jh_ctx();
    if (n <= c.part_cap) return JH_OK;
    int64_t cap = c.part_cap ? c.part_cap : 4096;
    while (cap < n) cap *= 2;
    if (c.part_dev) {
        JH_CHECK_HIP(hipStreamSynchronize(c.stream));
        JH_CHECK_HIP(hipFree(c.part_dev));
        c.part_dev = nullptr;
        c.part_cap = 0;
    }
    JH_CHECK_HIP(jh_device_malloc(c.device, (void **)&c.part_dev, sizeof(double) * (size_t)cap));
    c.part_cap = cap;
    c.buf_gen++;
    return JH_OK;
}

int jh_ensure_scratch(size_t bytes, void **out)
{
    jh_context &c = jh_ctx();
    if (bytes > c.scratch_cap) {
        size_t cap = c.scratch_cap ? c.scratch_cap : (size_t)1 << 20;
        while (cap < bytes) cap *= 2;
        if (c.scratch_dev) {
            JH_CHECK_HIP(hipStreamSynchronize(c.stream));
            JH_CHECK_HIP(hipFree(c.scratch_dev));
            c.scratch_dev = nullptr;
            c.scratch_cap = 0;
        }
        JH_CHECK_HIP(jh_device_malloc(c.device, &c.scratch_dev, cap));
        c.scratch_cap = cap;
        c.buf_gen++;
    }
    *out = c.scratch_dev;
    return JH_OK;
}

int jh_require_ready()
{
    jh_context &c = jh_ctx();
    if (!c.ready) return jh_fail(JH_ERR_STATE, "libjetship: jh_init(device) has not been called");
    int cur = -1;                                            // another library in this thread may have switched devices
    if (hipGetDevice(&cur) == hipSuccess && cur != c.device) JH_CHECK_HIP(hipSetDevice(c.device));
    return JH_OK;
}

std::atomic<int64_t> jh_bvec_generation{0};   // see jh_internal.h

extern "C" {

int jh_abi_version(void) { return JETSHIP_ABI_VERSION; }
const char *jh_last_error(void) { return g_err; }

int jh_device_count(int *count)
{
    JH_REQUIRE(count, "jh_device_count: null output");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return jh_fail(JH_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return JH_OK;
}

int jh_init(int device)
{
    int id = -1;
    JH_TRY(ctx_create(device, true, &id));               // the device's primary context; idempotent
    return ctx_use(id);
}

int jh_context_create(int device, int *ctx)
{
    JH_REQUIRE(ctx, "jh_context_create: null output");
    int id = -1;
    JH_TRY(ctx_create(device, false, &id));
    *ctx = id;
    return ctx_use(id);
}

int jh_context_use(int ctx) { return ctx_use(ctx); }

int jh_context_current(int *ctx, int *device)
{
    JH_TRY(jh_require_ready());
    if (ctx) *ctx = jh_ctx().id;
    if (device) *device = jh_ctx().device;
    return JH_OK;
}

int jh_context_destroy(int ctx)
{
    const jh_context *c = jh_ctx_by_id(ctx);
    JH_REQUIRE(c, "jh_context_destroy: no context %d", ctx);
    if (c->live_handles > 0)
        return jh_fail(JH_ERR_STATE, "jh_context_destroy: context %d still owns %lld vectors / operators / events; destroy them first", ctx,
                       (long long)c->live_handles.load());
    return ctx_destroy(ctx);
}

int jh_set_device(int device)
{
    for (int k = 0; k < JH_MAX_CTX; k++)
        if (g_ctxs[k] && g_ctxs[k]->primary && g_ctxs[k]->device == device) return ctx_use(g_ctxs[k]->id);
    return jh_fail(JH_ERR_STATE, "jh_set_device: device %d has no context; call jh_init(%d) first", device, device);
}

int jh_shutdown(void)
{
    for (int k = 0; k < JH_MAX_CTX; k++)
        if (g_ctxs[k]) (void)ctx_destroy(g_ctxs[k]->id);
    jh_bcast_clear_cache();
    t_cur_ctx = -1;
    return JH_OK;
}

int jh_trim(void)
{
    JH_TRY(jh_require_ready());
    jh_slab_trim(jh_ctx().device);
    return JH_OK;
}

int jh_device_info(char *name, int name_cap, int64_t *total_mem, int64_t *free_mem, int *cu_count)
{
    JH_TRY(jh_require_ready());
    jh_context &c = jh_ctx();
    hipDeviceProp_t prop;
    JH_CHECK_HIP(hipGetDeviceProperties(&prop, c.device));
    if (name && name_cap > 0) snprintf(name, (size_t)name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    size_t fr = 0, tot = 0;
    JH_CHECK_HIP(hipMemGetInfo(&fr, &tot));
    if (total_mem) *total_mem = (int64_t)tot;
    if (free_mem) *free_mem = (int64_t)(fr + jh_slab_cached_bytes(c.device));   // what the slab cache holds is available to the next allocation
    if (cu_count) *cu_count = prop.multiProcessorCount;
    return JH_OK;
}

int jh_get_stream(void **hip_stream)
{
    JH_TRY(jh_require_ready());
    JH_REQUIRE(hip_stream, "jh_get_stream: null output");
    *hip_stream = (void *)jh_ctx().stream;
    return JH_OK;
}

int jh_set_stream(void *hip_stream)
{
    JH_TRY(jh_require_ready());
    jh_context &c = jh_ctx();
    c.stream = hip_stream ? (hipStream_t)hip_stream : c.own_stream;
    return JH_OK;
}

int jh_synchronize(void)
{
    JH_TRY(jh_require_ready());
    JH_CHECK_HIP(hipStreamSynchronize(jh_ctx().stream));
    return JH_OK;
}

int jh_event_create(jh_event **ev)
{
    JH_TRY(jh_require_ready());
    JH_REQUIRE(ev, "jh_event_create: null output");
    jh_event *e = new jh_event();
    e->ctx = jh_ctx().id;
    hipError_t r = hipEventCreate(&e->ev);
    if (r != hipSuccess) {
        delete e;
        return jh_fail(JH_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(r));
    }
    *ev = e;
    jh_handle_born(e->ctx);
    return JH_OK;
}

int jh_event_record(jh_event *ev)
{
    JH_TRY(jh_enter(ev));
    JH_REQUIRE(ev, "jh_event_record: null event");
    JH_CHECK_HIP(hipEventRecord(ev->ev, jh_ctx().stream));
    return JH_OK;
}

int jh_event_elapsed_ms(jh_event *start, jh_event *stop, float *ms)
{
    JH_TRY(jh_enter(start, stop));
    JH_REQUIRE(start && stop && ms, "jh_event_elapsed_ms: null argument");
    JH_CHECK_HIP(hipEventSynchronize(stop->ev));
    JH_CHECK_HIP(hipEventElapsedTime(ms, start->ev, stop->ev));
    return JH_OK;
}

int jh_event_destroy(jh_event *ev)
{
    if (!ev) return JH_OK;
    if (ev->ev) (void)hipEventDestroy(ev->ev);
    jh_handle_died(ev->ctx);
    delete ev;
    return JH_OK;
}

// ------------------------------------------------------------------ block vectors -------------
static int build_layout(jh_bvec *v, int64_t nblocks, const int64_t *block_len, int dtype, const char *who)
{
    JH_REQUIRE(nblocks >= 1, "%s: nblocks must be >= 1 (got %lld)", who, (long long)nblocks);
    JH_REQUIRE(block_len, "%s: null block_len", who);
    JH_REQUIRE(jh_dtype_size(dtype) != 0, "%s: unknown dtype %d", who, dtype);
    v->dtype = dtype;
    v->nblocks = nblocks;
    v->off.resize((size_t)nblocks + 1);
    v->off[0] = 0;
    v->uniform = true;
    for (int64_t i = 0; i < nblocks; i++) {
        JH_REQUIRE(block_len[i] >= 0, "%s: block %lld has negative length", who, (long long)i);
        // src/Jets.jl:745-746  start = stop+1 ; stop = start+length-1  (0-based: off[i+1] = off[i] + len)
        v->off[(size_t)i + 1] = v->off[(size_t)i] + block_len[i];
        if (block_len[i] != block_len[0]) v->uniform = false;
    }
    v->length = v->off[(size_t)nblocks];
    return JH_OK;
}

static int bvec_create(int64_t nblocks, const int64_t *block_len, int dtype, jh_bvec **out, bool zero_fill);

int jh_bvec_create(int64_t nblocks, const int64_t *block_len, int dtype, jh_bvec **out) { return bvec_create(nblocks, block_len, dtype, out, true); }

int jh_bvec_create_uninit(int64_t nblocks, const int64_t *block_len, int dtype, jh_bvec **out) { return bvec_create(nblocks, block_len, dtype, out, false); }

static int bvec_create(int64_t nblocks, const int64_t *block_len, int dtype, jh_bvec **out, bool zero_fill)
{
    JH_TRY(jh_require_ready());
    JH_REQUIRE(out, "jh_bvec_create: null output");
    jh_bvec *v = new jh_bvec();
    v->ctx = jh_ctx().id;
    int s = build_layout(v, nblocks, block_len, dtype, "jh_bvec_create");
    if (s != JH_OK) { delete v; return s; }
    size_t bytes = (size_t)v->length * jh_dtype_size(dtype);
    if (bytes == 0) bytes = 16;
    hipError_t e = jh_slab_alloc(jh_ctx().device, bytes, &v->data, (int)jh_ctx().alloc_role);   // a cached slab of a destroyed vector of this size, or hipMalloc
    if (e != hipSuccess) {
        delete v;
        return jh_fail(JH_ERR_NOMEM, "jh_bvec_create: hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
    }
    v->owns = true;
    if (zero_fill) {                                             // zeros(R), src/Jets.jl:922-924
        // big slabs through the library's own fill kernel (6.8 TB/s; the runtime's memset moves 5.7: 10 against 12 ms per 64 GiB)
        if (bytes >= ((size_t)64 << 20) && v->length > 0) e = jh_launch_fill_range(v->data, dtype, v->length, 0.0, 0.0) == JH_OK ? hipSuccess : hipErrorUnknown;
        else e = hipMemsetAsync(v->data, 0, bytes, jh_ctx().stream);
    }
    if (e != hipSuccess) {
        jh_slab_free(jh_ctx().device, v->data, bytes, jh_ctx().stream);
        delete v;
        return jh_fail(JH_ERR_HIP, "jh_bvec_create: hipMemsetAsync: %s", hipGetErrorString(e));
    }
    *out = v;
    jh_handle_born(v->ctx);
    return JH_OK;
}

int jh_bvec_wrap(void *device_ptr, int64_t nblocks, const int64_t *block_len, int dtype, jh_bvec **out)
{
    JH_TRY(jh_require_ready());
    JH_REQUIRE(out, "jh_bvec_wrap: null output");
    JH_REQUIRE(device_ptr, "jh_bvec_wrap: null device pointer");
    jh_bvec *v = new jh_bvec();
    v->ctx = jh_ctx().id;                                    // the caller's pointer must belong to the current context's device
    int s = build_layout(v, nblocks, block_len, dtype, "jh_bvec_wrap");
    if (s != JH_OK) { delete v; return s; }
    v->data = device_ptr;
    v->owns = false;
    *out = v;
    jh_handle_born(v->ctx);
    return JH_OK;
}

int jh_bvec_view(jh_bvec *parent, int64_t first_block, int64_t count, jh_bvec **out)
{
    JH_TRY(jh_enter(parent));
    JH_REQUIRE(parent && out, "jh_bvec_view: null argument");
    JH_REQUIRE(first_block >= 0 && count >= 1 && first_block + count <= parent->nblocks,
               "jh_bvec_view: blocks [%lld, %lld) out of range (nblocks = %lld)", (long long)first_block,
               (long long)(first_block + count), (long long)parent->nblocks);
    jh_bvec *v = new jh_bvec();
    v->ctx = parent->ctx;
    v->dtype = parent->dtype;
    v->nblocks = count;
    v->off.resize((size_t)count + 1);
    v->uniform = true;
    int64_t base = parent->off[(size_t)first_block];
    for (int64_t i = 0; i <= count; i++) v->off[(size_t)i] = parent->off[(size_t)(first_block + i)] - base;
    for (int64_t i = 0; i < count; i++)
        if (v->len(i) != v->len(0)) v->uniform = false;
    v->length = v->off[(size_t)count];
    v->data = parent->ptr(base);
    v->owns = false;
    *out = v;
    jh_handle_born(v->ctx);
    return JH_OK;
}

int jh_bvec_destroy(jh_bvec *v)
{
    if (!v) return JH_OK;
    if (v->owns && v->data) {
        jh_quiesce_scope quiet(v->ctx);                          // (not jh_enter: a finaliser must not change the thread's current context)
        jh_context *c = jh_ctx_by_id(v->ctx);
        size_t bytes = (size_t)v->length * jh_dtype_size(v->dtype);
        if (bytes == 0) bytes = 16;
        if (c) jh_slab_free(c->device, v->data, bytes, c->stream);   // big slabs wait in the cache for the next vector of their size
        else (void)hipFree(v->data);
    }
    jh_handle_died(v->ctx);
    jh_bvec_generation.fetch_add(1);                             // (cached device tables naming this handle's data are stale from here on)
    delete v;
    return JH_OK;
}

int jh_bvec_info(const jh_bvec *v, int64_t *nblocks, int64_t *length, int *dtype, void **device_ptr)
{
    JH_REQUIRE(v, "jh_bvec_info: null vector");
    if (nblocks) *nblocks = v->nblocks;
    if (length) *length = v->length;
    if (dtype) *dtype = v->dtype;
    if (device_ptr) *device_ptr = v->data;
    return JH_OK;
}

int jh_bvec_context(const jh_bvec *v, int *ctx, int *device)
{
    JH_REQUIRE(v, "jh_bvec_context: null vector");
    const jh_context *c = jh_ctx_by_id(v->ctx);
    JH_REQUIRE(c, "jh_bvec_context: the vector's context %d no longer exists", v->ctx);
    if (ctx) *ctx = v->ctx;
    if (device) *device = c->device;
    return JH_OK;
}

int jh_bvec_block(const jh_bvec *v, int64_t iblock, int64_t *offset, int64_t *len, void **device_ptr)
{
    JH_REQUIRE(v, "jh_bvec_block: null vector");
    JH_REQUIRE(iblock >= 0 && iblock < v->nblocks, "jh_bvec_block: block %lld out of range (nblocks = %lld)",
               (long long)iblock, (long long)v->nblocks);
    if (offset) *offset = v->off[(size_t)iblock];
    if (len) *len = v->len(iblock);
    if (device_ptr) *device_ptr = v->ptr(v->off[(size_t)iblock]);
    return JH_OK;
}

int jh_getblock_copy(const jh_bvec *v, int64_t iblock, void *dst, int dst_on_device)
{
    JH_TRY(jh_enter(v));
    JH_REQUIRE(v && dst, "jh_getblock_copy: null argument");
    JH_REQUIRE(iblock >= 0 && iblock < v->nblocks, "jh_getblock_copy: block %lld out of range (nblocks = %lld)",
               (long long)iblock, (long long)v->nblocks);
    size_t bytes = (size_t)v->len(iblock) * jh_dtype_size(v->dtype);
    if (bytes == 0) return JH_OK;
    hipStream_t st = jh_ctx().stream;
    JH_CHECK_HIP(hipMemcpyAsync(dst, v->ptr(v->off[(size_t)iblock]), bytes,
                                dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    if (!dst_on_device) JH_CHECK_HIP(hipStreamSynchronize(st));
    return JH_OK;
}

int jh_setblock_copy(jh_bvec *v, int64_t iblock, const void *src, int src_on_device)
{
    JH_TRY(jh_enter(v));
    JH_REQUIRE(v && src, "jh_setblock_copy: null argument");
    JH_REQUIRE(iblock >= 0 && iblock < v->nblocks, "jh_setblock_copy: block %lld out of range (nblocks = %lld)",
               (long long)iblock, (long long)v->nblocks);
    size_t bytes = (size_t)v->len(iblock) * jh_dtype_size(v->dtype);
    if (bytes == 0) return JH_OK;
    hipStream_t st = jh_ctx().stream;
    JH_CHECK_HIP(hipMemcpyAsync(v->ptr(v->off[(size_t)iblock]), src, bytes,
                                src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    if (!src_on_device) JH_CHECK_HIP(hipStreamSynchronize(st));   // caller may reuse its host buffer
    return JH_OK;
}

int jh_setblock_fill(jh_bvec *v, int64_t iblock, double re, double im)
{
    JH_TRY(jh_enter(v));
    JH_REQUIRE(v, "jh_setblock_fill: null vector");
    JH_REQUIRE(iblock >= 0 && iblock < v->nblocks, "jh_setblock_fill: block %lld out of range (nblocks = %lld)",
               (long long)iblock, (long long)v->nblocks);
    return jh_launch_fill_range(v->ptr(v->off[(size_t)iblock]), v->dtype, v->len(iblock), re, im);
}

int jh_fill(jh_bvec *v, double re, double im)
{
    JH_TRY(jh_enter(v));
    JH_REQUIRE(v, "jh_fill: null vector");
    return jh_launch_fill_range(v->data, v->dtype, v->length, re, im);
}

int jh_copy(jh_bvec *dst, const jh_bvec *src)
{
    JH_TRY(jh_enter(dst, src));
    JH_REQUIRE(dst && src, "jh_copy: null argument");
    JH_REQUIRE(dst->dtype == src->dtype, "jh_copy: dtype mismatch (%d vs %d)", dst->dtype, src->dtype);
    JH_REQUIRE(dst->length == src->length, "jh_copy: length mismatch (%lld vs %lld)", (long long)dst->length,
               (long long)src->length);
    size_t bytes = (size_t)dst->length * jh_dtype_size(dst->dtype);
    return jh_launch_copy_bytes(dst->data, src->data, bytes);
}

int jh_download(const jh_bvec *v, int64_t offset, int64_t count, void *host_dst)
{
    JH_TRY(jh_enter(v));
    JH_REQUIRE(v && host_dst, "jh_download: null argument");
    JH_REQUIRE(offset >= 0 && count >= 0 && offset + count <= v->length,
               "jh_download: range [%lld, %lld) outside vector of length %lld", (long long)offset,
               (long long)(offset + count), (long long)v->length);
    if (count == 0) return JH_OK;
    hipStream_t st = jh_ctx().stream;
    JH_CHECK_HIP(hipMemcpyAsync(host_dst, v->ptr(offset), (size_t)count * jh_dtype_size(v->dtype), hipMemcpyDeviceToHost, st));
    JH_CHECK_HIP(hipStreamSynchronize(st));
    return JH_OK;
}

int jh_upload(jh_bvec *v, int64_t offset, int64_t count, const void *host_src)
{
    JH_TRY(jh_enter(v));
    JH_REQUIRE(v && host_src, "jh_upload: null argument");
    JH_REQUIRE(offset >= 0 && count >= 0 && offset + count <= v->length,
               "jh_upload: range [%lld, %lld) outside vector of length %lld", (long long)offset,
               (long long)(offset + count), (long long)v->length);
    if (count == 0) return JH_OK;
    hipStream_t st = jh_ctx().stream;
    JH_CHECK_HIP(hipMemcpyAsync(v->ptr(offset), host_src, (size_t)count * jh_dtype_size(v->dtype), hipMemcpyHostToDevice, st));
    JH_CHECK_HIP(hipStreamSynchronize(st));
    return JH_OK;
}

int jh_host_alloc(size_t bytes, void **out)
{
    JH_TRY(jh_require_ready());
    JH_REQUIRE(out, "jh_host_alloc: null argument");
    *out = nullptr;
    if (bytes == 0) return JH_OK;
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        *out = nullptr;
        (void)hipGetLastError();
        return jh_fail(JH_ERR_NOMEM, "jh_host_alloc: hipHostMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
    }
    return JH_OK;
}

int jh_host_free(void *ptr)
{
    if (!ptr) return JH_OK;
    JH_TRY(jh_require_ready());
    JH_CHECK_HIP(hipStreamSynchronize(jh_ctx().stream));      // a copy into / out of it may still be in flight
    JH_CHECK_HIP(hipHostFree(ptr));
    return JH_OK;
}

int jh_host_register(void *ptr, size_t bytes)
{
    JH_TRY(jh_require_ready());
    JH_REQUIRE(ptr && bytes > 0, "jh_host_register: null or empty buffer");
    hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return jh_fail(JH_ERR_HIP, "jh_host_register: hipHostRegister(%zu bytes): %s", bytes, hipGetErrorString(e));
    }
    return JH_OK;
}

int jh_host_unregister(void *ptr)
{
    JH_TRY(jh_require_ready());
    JH_REQUIRE(ptr, "jh_host_unregister: null argument");
    JH_CHECK_HIP(hipStreamSynchronize(jh_ctx().stream));
    hipError_t e = hipHostUnregister(ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return jh_fail(JH_ERR_INVALID, "jh_host_unregister: %s", hipGetErrorString(e));
    }
    return JH_OK;
}

int jh_tune_set(const char *name, int64_t value)
{
    JH_REQUIRE(name, "jh_tune_set: null name");
    jh_context &c = jh_ctx();
    auto one_of = [&](std::initializer_list<int64_t> ok) { for (int64_t v : ok) if (v == value) return true; return false; };
    if (!strcmp(name, "fwd_group")) { JH_REQUIRE(value >= 0 && value <= 65536, "fwd_group out of range"); c.fwd_group = value; }
    else if (!strcmp(name, "fwd_unroll")) { JH_REQUIRE(one_of({0, 1, 2, 4, 8}), "fwd_unroll must be 0 (auto), 1, 2, 4 or 8"); c.fwd_unroll = value; }
    else if (!strcmp(name, "fwd_wg")) { JH_REQUIRE(one_of({0, 256, 512, 1024}), "fwd_wg must be 0 (auto), 256, 512 or 1024"); c.fwd_wg = value; }
    else if (!strcmp(name, "adj_unroll")) { JH_REQUIRE(one_of({0, 1, 2, 4}), "adj_unroll must be 0 (auto), 1, 2 or 4"); c.adj_unroll = value; }
    else if (!strcmp(name, "adj_depth")) { JH_REQUIRE(one_of({0, 1, 2, 4, 8}), "adj_depth must be 0 (auto), 1, 2, 4 or 8"); c.adj_depth = value; }
    else if (!strcmp(name, "adj_wg")) { JH_REQUIRE(one_of({0, 256, 512, 1024}), "adj_wg must be 0 (auto), 256, 512 or 1024"); c.adj_wg = value; }
    else if (!strcmp(name, "fwd_order")) { JH_REQUIRE(value >= -1 && value <= 65536, "fwd_order must be -1 (auto), 0 (sequential), 1 (all rows) or k > 1 (k row groups per band)"); c.fwd_order = value; }
    else if (!strcmp(name, "nt")) { JH_REQUIRE(value >= 0 && value <= 2, "nt must be 0 (temporal), 1 (nontemporal unless the working set is cache-resident) or 2 (always nontemporal)"); c.nt = value; }
    else if (!strcmp(name, "nt_resident_mib")) { JH_REQUIRE(value >= 0, "nt_resident_mib must be >= 0"); c.nt_resident_mib = value; }
    else if (!strcmp(name, "slab_cache")) { g_slab_cache_on.store(value ? 1 : 0); if (!value) jh_slab_trim(-1); }   // the switch is process-wide: so is the trim
    else if (!strcmp(name, "bcast_item_fast")) { JH_REQUIRE(value >= -1 && value <= 1, "bcast_item_fast must be -1 (auto), 0 or 1"); c.bcast_item_fast = value; }
    else if (!strcmp(name, "adj_split")) { JH_REQUIRE(value >= -1 && value <= 65535, "adj_split must be -1 (auto), 0 (never: ordered walk) or the number of row parts"); c.adj_split = value; }
    else if (!strcmp(name, "adj_rows_per_launch")) { JH_REQUIRE(value >= 0, "adj_rows_per_launch must be >= 0"); c.adj_rows_per_launch = value; }
    else if (!strcmp(name, "autotune")) { c.autotune = value ? 1 : 0; }
    else if (!strcmp(name, "graphs")) { c.graphs = value ? 1 : 0; }
    else if (!strcmp(name, "small_loop")) { c.small_loop = value ? 1 : 0; }
    else if (!strcmp(name, "force_dist")) { c.force_dist = value ? 1 : 0; }
    else if (!strcmp(name, "step_chain")) { JH_REQUIRE(value >= -1 && value <= 1, "step_chain must be -1 (auto), 0 or 1"); c.step_chain = value; }
    else if (!strcmp(name, "step_chunk")) { JH_REQUIRE(value == 0 || value == 8 || value == 16 || value == 32, "step_chunk must be 0 (automatic), 8, 16 or 32"); c.step_chunk = value; }
    else if (!strcmp(name, "step_band")) { JH_REQUIRE(value >= -1 && value < ((int64_t)1 << 24), "step_band must be -1 (default), 0 (none) or a number of tiles"); c.step_band = value; }
    else if (!strcmp(name, "general_xcd")) { JH_REQUIRE(value >= 0 && value <= 2, "general_xcd must be 0 (never), 1 (automatic) or 2 (always)"); c.general_xcd = value; }
    else if (!strcmp(name, "lsqr_graph")) { c.lsqr_graph = value < 0 ? 0 : (value > 2 ? 2 : value); }
    else if (!strcmp(name, "grid_tile")) { JH_REQUIRE(value == 0 || value == 1 || value == 2 || value == 4 || value == 8, "grid_tile must be 0 (k_grid_diag), 1 (automatic) or 2 / 4 / 8 lines per workgroup"); c.grid_tile = value; }
    else if (!strcmp(name, "dense_mixed")) { JH_REQUIRE(value == 0 || value == 1, "dense_mixed must be 0 or 1"); c.dense_mixed = value; }
    else if (!strcmp(name, "general_tile")) { JH_REQUIRE(value == 0 || value == 1 || value == 2 || value == 4, "general_tile must be 0 (one-line kernels), 1 (automatic), 2 or 4 lines per workgroup"); c.general_tile = value; }
    else if (!strcmp(name, "general_list")) { JH_REQUIRE(value >= 0 && value <= 3, "general_list must be 0 (never), 1 (automatic), 2 (always the four-line lists) or 3 (always the per-line lists)"); c.general_list = value; }
    else if (!strcmp(name, "small_loop_max_kib")) { JH_REQUIRE(value >= 0, "small_loop_max_kib must be >= 0"); c.small_loop_max_kib = value; }
    else if (!strcmp(name, "dense_list")) { JH_REQUIRE(value >= 0 && value <= 2, "dense_list must be 0 (the grid over every block pair), 1 (the children's lists) or 2 (... also for few big children)"); c.dense_list = value; }
    else if (!strcmp(name, "dense_list_shared")) { JH_REQUIRE(value == 0 || value == 1, "dense_list_shared must be 0 or 1"); c.dense_list_shared = value; }
    else if (!strcmp(name, "dense_combine")) { JH_REQUIRE(value == 0 || value == 1, "dense_combine must be 0 or 1"); c.dense_combine = value; }
    else if (!strcmp(name, "dense_list_rl_min")) { JH_REQUIRE(value == 0 || (value >= 4 && value <= 8), "dense_list_rl_min must be 0 (default) or a shift 4 .. 8"); c.dense_list_rl_min = value; }
    else if (!strcmp(name, "dense_list_cpw")) { JH_REQUIRE(value == 0 || value == 1 || value == 2 || value == 4, "dense_list_cpw must be 0 (by column length), 1, 2 or 4"); c.dense_list_cpw = value; }
    else if (!strcmp(name, "dense_grid")) { c.dense_grid = value ? 1 : 0; }
    else if (!strcmp(name, "dense_direct")) { c.dense_direct = value ? 1 : 0; }
    else if (!strcmp(name, "dense_list_split")) { JH_REQUIRE(value == 0 || value == 1, "dense_list_split must be 0 (columns in order) or 1 (automatic lane layout)"); c.dense_list_split = value; }
    else if (!strcmp(name, "dense_fused")) { JH_REQUIRE(value == 0 || value == 1, "dense_fused must be 0 or 1"); c.dense_fused = value; }
    else if (!strcmp(name, "cgls_trace")) { c.cgls_trace = value ? 1 : 0; }
    else if (!strcmp(name, "cg_dev")) { c.cg_dev = value < 0 ? 0 : (value > 2 ? 2 : value); }
    else if (!strcmp(name, "walk_memory")) { c.walk_memory = value ? 1 : 0; }
    else if (!strcmp(name, "slab_free_floor_mib")) { JH_REQUIRE(value >= 0, "slab_free_floor_mib must be >= 0"); g_slab_free_floor_mib.store(value); }
    else if (!strcmp(name, "alloc_role")) { JH_REQUIRE(value >= 0 && value <= 2, "alloc_role must be 0 (none), 1 (an operator's output) or 2 (data written once, read from then on)"); c.alloc_role = value; }
    else if (!strcmp(name, "dense_fwd_wgs")) { JH_REQUIRE(value >= 0 && value <= 65536, "dense_fwd_wgs must be 0 (automatic) .. 65536"); c.dense_fwd_wgs = value; }
    else if (!strcmp(name, "dense_gw")) { JH_REQUIRE(value >= 0 && value <= 4096, "dense_gw must be 0 (automatic) or 1 .. 4096 children per wave"); c.dense_gw = value; }
    else if (!strcmp(name, "sum_group")) { JH_REQUIRE(value == 4 || value == 8 || value == 16, "sum_group must be 4, 8 or 16 terms per forward launch"); c.sum_group = value; }
    else if (!strcmp(name, "bcast_band")) { c.bcast_band = value < 0 ? 0 : value; }
    else if (!strcmp(name, "general_band")) { JH_REQUIRE(value == 8 || value == 16 || value == 32 || value == 64, "general_band must be 8, 16, 32 or 64 tiles"); c.general_band = value; }
    else if (!strcmp(name, "fwd_ctiles")) { c.fwd_ctiles = value < -1 ? -1 : value; }
    else if (!strcmp(name, "sum_adj_group")) { c.sum_adj_group = value == 16 ? 16 : 8; }
    else if (!strcmp(name, "grid_diag")) { JH_REQUIRE(value >= 0 && value <= 4, "grid_diag must be 0 (general kernels), 1, 2 or 4 (packs per lane)"); c.grid_diag = value; }
    else if (!strcmp(name, "tall_f")) { JH_REQUIRE(value >= 0 && value <= 1, "tall_f must be 0 or 1"); c.tall_f = value; }
    else if (!strcmp(name, "red_blocks_wave")) { c.red_blocks_wave = value ? 1 : 0; }
    else if (!strcmp(name, "adj_bare_chain")) { c.adj_bare_chain = value ? 1 : 0; }
    else if (!strcmp(name, "adj_thin_mixed")) { c.adj_thin_mixed = value ? 1 : 0; }
    else if (!strcmp(name, "grid_normal")) { JH_REQUIRE(value >= 0 && value <= 2, "grid_normal must be 0, 1 or 2"); c.grid_normal = value; }
    else if (!strcmp(name, "fwd_anchor")) { JH_REQUIRE(value >= -1 && value <= 1, "fwd_anchor must be -1 (rows of >= 64 KiB that are not whole packs), 0 (never) or 1 (always)"); c.fwd_anchor = value; }
    else if (!strcmp(name, "ua_nt")) { JH_REQUIRE(value >= -1 && value <= 1, "ua_nt must be -1 (temporal accesses on rows off the 16-byte grid), 0 (temporal always) or 1 (nontemporal always)"); c.ua_nt = value; }
    else if (!strcmp(name, "tall_unaligned")) { JH_REQUIRE(value >= 0 && value <= 1, "tall_unaligned must be 0 (general kernels) or 1 (under-aligned tall kernels)"); c.tall_unaligned = value; }
    else if (!strcmp(name, "wide_twin")) { JH_REQUIRE(value >= 0 && value <= 2, "wide_twin must be 0 (never), 1 (automatic) or 2 (always)"); c.wide_twin = value; }
    else if (!strcmp(name, "red_wgs")) { JH_REQUIRE(value >= 1 && value <= 1 << 20, "red_wgs out of range"); c.red_wgs = value; }
    else return jh_fail(JH_ERR_INVALID, "jh_tune_set: unknown knob '%s'", name);
    return JH_OK;
}

int jh_tune_get(const char *name, int64_t *value)
{
    JH_REQUIRE(name && value, "jh_tune_get: null argument");
    jh_context &c = jh_ctx();
    if (!strcmp(name, "fwd_group")) *value = c.fwd_group;
    else if (!strcmp(name, "fwd_unroll")) *value = c.fwd_unroll;
    else if (!strcmp(name, "fwd_wg")) *value = c.fwd_wg;
    else if (!strcmp(name, "adj_unroll")) *value = c.adj_unroll;
    else if (!strcmp(name, "adj_depth")) *value = c.adj_depth;
    else if (!strcmp(name, "adj_wg")) *value = c.adj_wg;
    else if (!strcmp(name, "fwd_order")) *value = c.fwd_order;
    else if (!strcmp(name, "nt")) *value = c.nt;
    else if (!strcmp(name, "nt_resident_mib")) *value = c.nt_resident_mib;
    else if (!strcmp(name, "slab_cache")) *value = g_slab_cache_on.load();
    else if (!strcmp(name, "slab_cached_mib")) *value = (int64_t)(jh_slab_cached_bytes(c.device) >> 20);
    else if (!strcmp(name, "bcast_item_fast")) *value = c.bcast_item_fast;
    else if (!strcmp(name, "adj_split")) *value = c.adj_split;
    else if (!strcmp(name, "last_adj_parts")) *value = c.last_adj_parts;
    else if (!strcmp(name, "adj_rows_per_launch")) *value = c.adj_rows_per_launch;
    else if (!strcmp(name, "autotune")) *value = c.autotune;
    else if (!strcmp(name, "graphs")) *value = c.graphs;
    else if (!strcmp(name, "small_loop")) *value = c.small_loop;
    else if (!strcmp(name, "force_dist")) *value = c.force_dist;
    else if (!strcmp(name, "step_chain")) *value = c.step_chain;
    else if (!strcmp(name, "step_band")) *value = c.step_band;
    else if (!strcmp(name, "step_chunk")) *value = c.step_chunk;
    else if (!strcmp(name, "last_step_chain")) *value = c.last_step_chain;
    else if (!strcmp(name, "general_xcd")) *value = c.general_xcd;
    else if (!strcmp(name, "graph_replays")) *value = c.graph_replays;
    else if (!strcmp(name, "last_fwd_rows_per_wg")) *value = c.last_fwd_rows_per_wg;
    else if (!strcmp(name, "last_adj_launches")) *value = c.last_adj_launches;
    else if (!strcmp(name, "lsqr_graph")) *value = c.lsqr_graph;
    else if (!strcmp(name, "last_lsqr_graph")) *value = c.last_lsqr_graph;
    else if (!strcmp(name, "grid_diag")) *value = c.grid_diag;
    else if (!strcmp(name, "grid_tile")) *value = c.grid_tile;
    else if (!strcmp(name, "sum_group")) *value = c.sum_group;
    else if (!strcmp(name, "bcast_band")) *value = c.bcast_band;
    else if (!strcmp(name, "general_band")) *value = c.general_band;
    else if (!strcmp(name, "fwd_ctiles")) *value = c.fwd_ctiles;
    else if (!strcmp(name, "sum_adj_group")) *value = c.sum_adj_group;
    else if (!strcmp(name, "general_tile")) *value = c.general_tile;
    else if (!strcmp(name, "general_list")) *value = c.general_list;
    else if (!strcmp(name, "dense_list")) *value = c.dense_list;
    else if (!strcmp(name, "small_loop_max_kib")) *value = c.small_loop_max_kib;
    else if (!strcmp(name, "dense_list_split")) *value = c.dense_list_split;
    else if (!strcmp(name, "dense_direct")) *value = c.dense_direct;
    else if (!strcmp(name, "dense_grid")) *value = c.dense_grid;
    else if (!strcmp(name, "dense_list_shared")) *value = c.dense_list_shared;
    else if (!strcmp(name, "dense_combine")) *value = c.dense_combine;
    else if (!strcmp(name, "dense_list_rl_min")) *value = c.dense_list_rl_min;
    else if (!strcmp(name, "dense_list_cpw")) *value = c.dense_list_cpw;
    else if (!strcmp(name, "last_dense_rl")) *value = c.last_dense_rl;
    else if (!strcmp(name, "last_general_list")) *value = c.last_general_list;
    else if (!strcmp(name, "dense_mixed")) *value = c.dense_mixed;
    else if (!strcmp(name, "dense_fused")) *value = c.dense_fused;
    else if (!strcmp(name, "cgls_trace")) *value = c.cgls_trace;
    else if (!strcmp(name, "cg_dev")) *value = c.cg_dev;
    else if (!strcmp(name, "walk_memory")) *value = c.walk_memory;
    else if (!strcmp(name, "alloc_role")) *value = c.alloc_role;
    else if (!strcmp(name, "slab_free_floor_mib")) *value = g_slab_free_floor_mib.load();
    else if (!strcmp(name, "slab_probed")) *value = jh_slab_probed(c.device);
    else if (!strcmp(name, "last_alloc_choice")) *value = jh_slab_last_choice();
    else if (!strcmp(name, "last_cg_graph")) *value = c.last_cg_graph;
    else if (!strcmp(name, "last_cgls_overlaps")) *value = c.last_cgls_overlaps;
    else if (!strcmp(name, "dense_gw")) *value = c.dense_gw;
    else if (!strcmp(name, "dense_fwd_wgs")) *value = c.dense_fwd_wgs;
    else if (!strcmp(name, "last_dense_fused")) *value = c.last_dense_fused;
    else if (!strcmp(name, "last_launches")) *value = c.last_launches;
    else if (!strcmp(name, "tall_unaligned")) *value = c.tall_unaligned;
    else if (!strcmp(name, "red_blocks_wave")) *value = c.red_blocks_wave;
    else if (!strcmp(name, "adj_bare_chain")) *value = c.adj_bare_chain;
    else if (!strcmp(name, "adj_thin_mixed")) *value = c.adj_thin_mixed;
    else if (!strcmp(name, "grid_normal")) *value = c.grid_normal;
    else if (!strcmp(name, "fwd_anchor")) *value = c.fwd_anchor;
    else if (!strcmp(name, "ua_nt")) *value = c.ua_nt;
    else if (!strcmp(name, "tall_f")) *value = c.tall_f;
    else if (!strcmp(name, "wide_twin")) *value = c.wide_twin;
    else if (!strcmp(name, "red_wgs")) *value = c.red_wgs;
    else if (!strcmp(name, "last_fwd_walk")) *value = c.last_fwd_walk;
    else return jh_fail(JH_ERR_INVALID, "jh_tune_get: unknown knob '%s'", name);
    return JH_OK;
}

}  // extern "C"
