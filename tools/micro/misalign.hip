// Experiment (round 5, session 3): what does a stream that is OFF the 16-byte grid cost on MI355X -- as a load stream, as a store stream?
// d[p] = a[p] * c over 16-byte packs, one pack per lane, 256-lane workgroups (the tall forward's shape without the model vector), 8 GiB per buffer;
// the load stream starts `oa` floats off a 256-byte boundary, the store stream `od` floats off; nontemporal and temporal loads; stores nontemporal.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/misalign tools/micro/misalign.hip && tools/micro/misalign
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float V __attribute__((ext_vector_type(4)));
typedef V UV __attribute__((aligned(4)));
template <bool NT> __global__ __launch_bounds__(256) void k(const float *__restrict__ a, float *__restrict__ d, long npack, float c)
{
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= npack) return;
    typedef const UV __attribute__((address_space(1))) *gp;
    typedef UV __attribute__((address_space(1))) *gq;
    V v = NT ? __builtin_nontemporal_load((gp)(a + 4 * p)) : *(gp)(a + 4 * p);
    __builtin_nontemporal_store(v * c, (gq)(d + 4 * p));
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main()
{
    const long bytes = 8L << 30, npack = bytes / 16 - 64;
    float *a, *d;
    CK(hipMalloc(&a, bytes + 4096));
    CK(hipMalloc(&d, bytes + 4096));
    CK(hipMemset(a, 0x3c, bytes + 4096));
    CK(hipMemset(d, 0, bytes + 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rnd = 0; rnd < 2; rnd++)
        for (int nt = 1; nt >= 0; nt--)
            for (int oa = 0; oa < 2; oa++)
                for (int od = 0; od < 2; od++) {
                    float best = 1e30f;
                    for (int rep = 0; rep < 5; rep++) {
                        CK(hipEventRecord(e0));
                        if (nt) hipLaunchKernelGGL(k<true>, dim3((unsigned)((npack + 255) / 256)), dim3(256), 0, 0, a + oa, d + od, npack, 1.5f);
                        else hipLaunchKernelGGL(k<false>, dim3((unsigned)((npack + 255) / 256)), dim3(256), 0, 0, a + oa, d + od, npack, 1.5f);
                        CK(hipEventRecord(e1));
                        CK(hipEventSynchronize(e1));
                        float ms;
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        if (rep && ms < best) best = ms;
                    }
                    printf("loads %s, load stream %s the grid, store stream %s the grid: %7.3f ms  %6.0f GB/s\n", nt ? "nontemporal" : "temporal   ", oa ? "OFF" : "on ",
                           od ? "OFF" : "on ", best, 2.0 * npack * 16 / best / 1e6);
                }
    return 0;
}
