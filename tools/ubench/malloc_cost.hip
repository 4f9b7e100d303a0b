// What a hipMalloc costs by size, next to copies on another stream (tools/ubench: measurement only).
//   hipcc --offload-arch=gfx950 -O2 -o malloc_cost malloc_cost.hip && ./malloc_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipFree(nullptr);
    const size_t sizes[] = {(size_t)64 << 20, (size_t)256 << 20, (size_t)1 << 30, (size_t)1300 << 20, (size_t)4 << 30, (size_t)12 << 30};
    for (size_t s : sizes) {
        for (int rep = 0; rep < 3; ++rep) {
            void* p = nullptr;
            double t0 = now();
            if (hipMalloc(&p, s) != hipSuccess) { printf("malloc %zu failed\n", s); return 1; }
            double t1 = now();
            hipMemsetAsync(p, 0, 256, nullptr);
            hipStreamSynchronize(nullptr);
            double t2 = now();
            hipFree(p);
            double t3 = now();
            printf("size %6zu MB: hipMalloc %7.2f ms, first touch %6.2f ms, hipFree %7.2f ms\n", s >> 20, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
        }
    }
    // does a hipMalloc on one thread stall copies issued on another stream?  (copies back to back, malloc in between)
    void *h = nullptr, *d = nullptr;
    hipHostMalloc(&h, (size_t)64 << 20);
    hipMalloc(&d, (size_t)64 << 20);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int with = 0; with < 2; ++with) {
        double t0 = now();
        for (int i = 0; i < 32; ++i) hipMemcpyAsync(d, h, (size_t)64 << 20, hipMemcpyHostToDevice, st);
        void* p = nullptr;
        double m0 = now();
        if (with) hipMalloc(&p, (size_t)1300 << 20);
        double m1 = now();
        hipStreamSynchronize(st);
        double t1 = now();
        printf("32 copies of 64 MB %s a 1.3 GB hipMalloc beside them: %.2f ms (malloc call %.2f ms)\n", with ? "with" : "without", (t1 - t0) * 1e3, (m1 - m0) * 1e3);
        if (p) hipFree(p);
    }
    return 0;
}
