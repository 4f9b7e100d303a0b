// Does an L2-scope (non-sc1) global atomic run faster than a device-scope one, and is it
// coherent among the workgroups of ONE XCD?  Each XCD (read from HW_REG_XCC_ID) gets a
// private table; all its workgroups add into it with workgroup-scope atomics; the host
// checks the totals.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

template <int SCOPE>
__global__ void k_add(unsigned long long* tabs, unsigned mask, int per, unsigned salt, unsigned* xcc_hist) {
    const unsigned xcc = xcc_id();
    if (threadIdx.x == 0) atomicAdd(&xcc_hist[xcc], 1u);
    unsigned long long* tab = tabs + (size_t)xcc * (mask + 1);
    unsigned x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + salt;
    for (int i = 0; i < per; ++i) {
        x = hash32(x + i);
        __hip_atomic_fetch_add(&tab[x & mask], 1ull, __ATOMIC_RELAXED, SCOPE);
    }
}
template <typename F>
float timeit(F f, int reps = 3) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best;
}
int main() {
    const int blocks = 2048, threads = 256, per = 32;
    const double n = (double)blocks * threads * per;
    unsigned long long* d; unsigned* dh;
    for (int lg : {11, 16, 20}) {
        unsigned mask = (1u << lg) - 1;
        size_t words = (size_t)16 * (mask + 1);
        CK(hipMalloc(&d, words * 8)); CK(hipMalloc(&dh, 64));
        for (int scope = 0; scope < 2; ++scope) {
            CK(hipMemset(d, 0, words * 8)); CK(hipMemset(dh, 0, 64));
            int launches = 0;
            float t = timeit([&] { ++launches;
                if (scope == 0) hipLaunchKernelGGL((k_add<__HIP_MEMORY_SCOPE_AGENT>), dim3(blocks), dim3(threads), 0, 0, d, mask, per, 7u, dh);
                else hipLaunchKernelGGL((k_add<__HIP_MEMORY_SCOPE_WORKGROUP>), dim3(blocks), dim3(threads), 0, 0, d, mask, per, 7u, dh); });
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(words); std::vector<unsigned> hh(16);
            CK(hipMemcpy(h.data(), d, words * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hh.data(), dh, 64, hipMemcpyDeviceToHost));
            unsigned long long tot = 0; for (auto v : h) tot += v;
            printf("2^%2d slots/XCD scope=%s: %8.1f us (%6.2f G/s)  total %llu expected %.0f %s | blocks per xcc:", lg, scope ? "workgroup(L2)" : "agent", t * 1e3, n / t / 1e6,
                   tot, n * launches, tot == (unsigned long long)(n * launches) ? "OK" : "LOST UPDATES");
            for (int i = 0; i < 8; ++i) printf(" %u", hh[i] / launches);
            printf("\n");
        }
        CK(hipFree(d)); CK(hipFree(dh));
    }
    return 0;
}
