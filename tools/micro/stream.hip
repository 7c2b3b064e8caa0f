// Micro-benchmark: which launch shape streams a flat slab fastest on MI355X?  fill (W), copy (R+W), triad (2R+W).
// hipcc --offload-arch=gfx950 -O3 -o stream stream.hip && ./stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float V __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// mode 0: grid-stride, UN packs in flight; mode 1: one shot, thread handles UN packs strided by block; 
template <int OP, int UN, bool NT> __global__ void k_gs(V *__restrict__ d, const V *__restrict__ a, const V *__restrict__ b, int64_t nvec)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = tid; v + (UN - 1) * stride < nvec; v += UN * stride) {
        V x[UN], y[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            if (OP >= 1) x[u] = NT ? __builtin_nontemporal_load(a + v + u * stride) : a[v + u * stride];
            if (OP >= 2) y[u] = NT ? __builtin_nontemporal_load(b + v + u * stride) : b[v + u * stride];
        }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            V r = OP == 0 ? (V)(3.14f) : (OP == 1 ? x[u] : x[u] + y[u]);
            if (NT) __builtin_nontemporal_store(r, d + v + u * stride); else d[v + u * stride] = r;
        }
    }
}
// one-shot: block handles UN*blockDim consecutive packs
template <int OP, int UN, bool NT> __global__ void k_os(V *__restrict__ d, const V *__restrict__ a, const V *__restrict__ b, int64_t nvec)
{
    const int64_t base = (int64_t)blockIdx.x * blockDim.x * UN + threadIdx.x;
    V x[UN], y[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
        const int64_t v = base + (int64_t)u * blockDim.x;
        if (v < nvec) {
            if (OP >= 1) x[u] = NT ? __builtin_nontemporal_load(a + v) : a[v];
            if (OP >= 2) y[u] = NT ? __builtin_nontemporal_load(b + v) : b[v];
        }
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
        const int64_t v = base + (int64_t)u * blockDim.x;
        if (v < nvec) {
            V r = OP == 0 ? (V)(3.14f) : (OP == 1 ? x[u] : x[u] + y[u]);
            if (NT) __builtin_nontemporal_store(r, d + v); else d[v] = r;
        }
    }
}

template <typename F> float timeit(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f();
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}

int main()
{
    const int64_t bytes = (int64_t)16 << 30, nvec = bytes / 16;
    V *d, *a, *b;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    const char *names[3] = {"fill ", "copy ", "triad"};
    const double mult[3] = {1, 2, 3};
#define RUN_GS(OP, UN, NT, BLK, GRID) { float ms = timeit([&] { hipLaunchKernelGGL((k_gs<OP, UN, NT>), dim3(GRID), dim3(BLK), 0, 0, d, a, b, nvec); }); \
    printf("%s grid-stride UN=%d nt=%d blk=%4d grid=%6d : %7.3f ms %7.1f GB/s\n", names[OP], UN, (int)NT, BLK, GRID, ms, mult[OP] * bytes / ms / 1e6); }
#define RUN_OS(OP, UN, NT, BLK) { int64_t g = (nvec + (int64_t)BLK * UN - 1) / ((int64_t)BLK * UN); float ms = timeit([&] { hipLaunchKernelGGL((k_os<OP, UN, NT>), dim3((unsigned)g), dim3(BLK), 0, 0, d, a, b, nvec); }); \
    printf("%s one-shot    UN=%d nt=%d blk=%4d grid=%6lld : %7.3f ms %7.1f GB/s\n", names[OP], UN, (int)NT, BLK, (long long)g, ms, mult[OP] * bytes / ms / 1e6); }
#define ALL(OP) \
    RUN_GS(OP, 1, true, 256, 2048) RUN_GS(OP, 4, true, 256, 2048) RUN_GS(OP, 4, false, 256, 2048) RUN_GS(OP, 4, true, 256, 8192) RUN_GS(OP, 4, true, 1024, 512) RUN_GS(OP, 8, true, 1024, 1024) \
    RUN_OS(OP, 1, true, 256) RUN_OS(OP, 4, true, 256) RUN_OS(OP, 4, false, 256) RUN_OS(OP, 8, true, 256) RUN_OS(OP, 4, true, 1024) RUN_OS(OP, 8, true, 1024) RUN_OS(OP, 8, false, 1024)
    ALL(0) ALL(1) ALL(2)
    return 0;
}
