// Micro-benchmarks: global (device-scope) and LDS atomic throughput on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// each thread does `per` atomicAdds to pseudo-random slots of a table with `mask+1` entries
template <typename T, bool WITH_LOAD>
__global__ void k_gatomic(T* tab, unsigned mask, int per, unsigned salt) {
    unsigned x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + salt;
    for (int i = 0; i < per; ++i) {
        x = hash32(x + i);
        unsigned h = x & mask;
        if (WITH_LOAD) {
            T cur = __hip_atomic_load(&tab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == (T)0xdeadbeef) continue;
        }
        atomicAdd(&tab[h], (T)1);
    }
}
// Zipf-ish skew: slot = floor(mask * u^4)
template <typename T>
__global__ void k_gatomic_skew(T* tab, unsigned mask, int per, unsigned salt) {
    unsigned x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + salt;
    for (int i = 0; i < per; ++i) {
        x = hash32(x + i);
        float u = (x >> 8) * (1.0f / 16777216.0f);
        unsigned h = (unsigned)(mask * u * u * u * u);
        atomicAdd(&tab[h], (T)1);
    }
}
// LDS atomics: ds_add_u32 / u64 to random slots of an LDS table
template <typename T>
__global__ void k_lds(T* out, unsigned mask, int per, unsigned salt, int skew) {
    extern __shared__ unsigned char smem[];
    T* tab = reinterpret_cast<T*>(smem);
    for (unsigned i = threadIdx.x; i <= mask; i += blockDim.x) tab[i] = 0;
    __syncthreads();
    unsigned x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + salt;
    for (int i = 0; i < per; ++i) {
        x = hash32(x + i);
        unsigned h;
        if (skew) { float u = (x >> 8) * (1.0f / 16777216.0f); h = (unsigned)(mask * u * u * u * u); } else h = x & mask;
        atomicAdd(&tab[h], (T)1);
    }
    __syncthreads();
    if (threadIdx.x == 0 && tab[0] == (T)0xdeadbeef) out[0] = tab[1];
}
// LDS: read key then add (hit path of a hash cache)
__global__ void k_lds_probe(unsigned long long* out, unsigned mask, int per, unsigned salt) {
    extern __shared__ unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    unsigned long long* vals = keys + mask + 1;
    for (unsigned i = threadIdx.x; i <= mask; i += blockDim.x) { keys[i] = i; vals[i] = 0; }
    __syncthreads();
    unsigned x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + salt;
    for (int i = 0; i < per; ++i) {
        x = hash32(x + i);
        unsigned h = x & mask;
        unsigned long long cur = __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == h) atomicAdd(&vals[h], 1ull);
    }
    __syncthreads();
    if (threadIdx.x == 0 && vals[0] == 0xdeadbeefull) out[0] = vals[1];
}

template <typename F>
float timeit(F f, int reps = 3) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best;
}

int main() {
    const int blocks = 2048, threads = 256, per = 32;
    const double n = (double)blocks * threads * per;   // 16.8 M atomics
    unsigned long long* d64; unsigned* d32;
    CK(hipMalloc(&d64, (size_t)(1u << 26) * 8)); CK(hipMalloc(&d32, (size_t)(1u << 26) * 4));
    CK(hipMemset(d64, 0, (size_t)(1u << 26) * 8)); CK(hipMemset(d32, 0, (size_t)(1u << 26) * 4));
    printf("global atomics, %.1f M ops, 2048x256 threads\n", n / 1e6);
    for (int lg : {10, 14, 18, 20, 22, 24, 26}) {
        unsigned mask = (1u << lg) - 1;
        float a = timeit([&] { hipLaunchKernelGGL((k_gatomic<unsigned long long, false>), dim3(blocks), dim3(threads), 0, 0, d64, mask, per, 1u); });
        float b = timeit([&] { hipLaunchKernelGGL((k_gatomic<unsigned long long, true>), dim3(blocks), dim3(threads), 0, 0, d64, mask, per, 2u); });
        float c = timeit([&] { hipLaunchKernelGGL((k_gatomic<unsigned, false>), dim3(blocks), dim3(threads), 0, 0, d32, mask, per, 3u); });
        float d = timeit([&] { hipLaunchKernelGGL((k_gatomic_skew<unsigned long long>), dim3(blocks), dim3(threads), 0, 0, d64, mask, per, 4u); });
        printf("  2^%2d slots: u64 add %8.1f us (%6.2f G/s) | load+add %8.1f us (%6.2f G/s) | u32 add %8.1f us (%6.2f G/s) | u64 skewed %8.1f us (%6.2f G/s)\n",
               lg, a * 1e3, n / a / 1e6, b * 1e3, n / b / 1e6, c * 1e3, n / c / 1e6, d * 1e3, n / d / 1e6);
    }
    printf("LDS atomics per block table, 512 blocks x 256 threads x 256 ops (33.5 M ops)\n");
    const double nl = 512.0 * 256 * 256;
    for (int lg : {8, 10, 12}) {
        unsigned mask = (1u << lg) - 1;
        for (int skew : {0, 1}) {
            float a = timeit([&] { hipLaunchKernelGGL((k_lds<unsigned>), dim3(512), dim3(256), (mask + 1) * 4, 0, d32, mask, 256, 5u, skew); });
            float b = timeit([&] { hipLaunchKernelGGL((k_lds<unsigned long long>), dim3(512), dim3(256), (mask + 1) * 8, 0, d64, mask, 256, 6u, skew); });
            printf("  2^%2d slots skew=%d: ds_add_u32 %8.1f us (%6.1f G/s) | ds_add_u64 %8.1f us (%6.1f G/s)\n", lg, skew, a * 1e3, nl / a / 1e6, b * 1e3, nl / b / 1e6);
        }
        float c = timeit([&] { hipLaunchKernelGGL(k_lds_probe, dim3(512), dim3(256), (mask + 1) * 16, 0, d64, mask, 256, 7u); });
        printf("  2^%2d slots: key-read + ds_add_u64 %8.1f us (%6.1f G/s)\n", lg, c * 1e3, nl / c / 1e6);
    }
    return 0;
}
