// tools/micro/alloc_cost.hip -- what hipMalloc / hipFree of a range-sized slab cost on MI355X, and when.
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/alloc_cost tools/micro/alloc_cost.hip && tools/micro/alloc_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_touch(float *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f;
}

static void cycle(const char *tag, size_t bytes, int reps, int sleep_ms, bool touch)
{
    printf("## %s: %zu GiB, %d cycles, %d ms pause after each free, %s\n", tag, bytes >> 30, reps, sleep_ms, touch ? "every page written" : "untouched");
    for (int r = 0; r < reps; r++) {
        void *p = nullptr;
        double t0 = now();
        hipError_t e = hipMalloc(&p, bytes);
        double t1 = now();
        if (e != hipSuccess) { printf("hipMalloc failed: %s\n", hipGetErrorString(e)); return; }
        double tk = 0;
        if (touch) {
            k_touch<<<16384, 256>>>((float *)p, bytes / 4);
            (void)hipDeviceSynchronize();
            tk = now() - t1;
        }
        double t2 = now();
        (void)hipFree(p);
        double t3 = now();
        printf("  cycle %d: hipMalloc %9.2f ms   write-all %8.2f ms   hipFree %8.2f ms   (%p)\n", r, t1 - t0, tk, t3 - t2, p);
        fflush(stdout);
        if (sleep_ms) std::this_thread::sleep_for(std::chrono::milliseconds(sleep_ms));
    }
}

int main(int argc, char **argv)
{
    (void)hipSetDevice(0);
    (void)hipFree(nullptr);
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    printf("# free %.1f GiB of %.1f GiB\n", free_b / 1073741824.0, total_b / 1073741824.0);
    if (argc > 1) {
        // `alloc_cost <GiB per piece>`: the FIRST allocations of a fresh process as pieces -- do 64 GiB in small requests dodge the seconds?
        const size_t piece = (size_t)atoi(argv[1]) << 30;
        const int count = (int)(((size_t)128 << 30) / piece);
        std::vector<void *> ps((size_t)count, nullptr);
        double total = 0;
        for (int k = 0; k < count; k++) {
            double t0 = now();
            hipError_t e = hipMalloc(&ps[(size_t)k], piece);
            double dt = now() - t0;
            total += dt;
            if (e != hipSuccess) { printf("hipMalloc failed: %s\n", hipGetErrorString(e)); return 1; }
            if (dt > 5.0 || k < 4) printf("  piece %3d of %zu GiB: hipMalloc %9.2f ms\n", k, piece >> 30, dt);
        }
        printf("## %d pieces of %zu GiB (128 GiB) as the first allocations of the process: %.1f ms in all\n", count, piece >> 30, total);
        for (void *p : ps) (void)hipFree(p);
        return 0;
    }
    cycle("A", (size_t)64 << 30, 5, 0, true);
    cycle("B", (size_t)64 << 30, 4, 0, false);
    cycle("C", (size_t)64 << 30, 4, 3000, true);
    cycle("D", (size_t)8 << 30, 5, 0, true);
    // two slabs held (like a and d), a third allocated and freed repeatedly (a temporary)
    void *a = nullptr, *d = nullptr;
    (void)hipMalloc(&a, (size_t)64 << 30);
    (void)hipMalloc(&d, (size_t)64 << 30);
    k_touch<<<16384, 256>>>((float *)a, ((size_t)64 << 30) / 4);
    k_touch<<<16384, 256>>>((float *)d, ((size_t)64 << 30) / 4);
    (void)hipDeviceSynchronize();
    cycle("E (128 GiB held)", (size_t)64 << 30, 5, 0, true);
    (void)hipFree(a);
    (void)hipFree(d);
    return 0;
}
