// Cost of a grid-wide barrier (cooperative launch) at the first pass's geometry:
// 256 workgroups x 1024 threads, 43 KB of dynamic LDS.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ void __launch_bounds__(1024) k_sync(unsigned* out, int rounds) {
    extern __shared__ unsigned char smem[];
    cg::grid_group grid = cg::this_grid();
    unsigned acc = 0;
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x == 0) out[blockIdx.x] = r + acc;
        grid.sync();
        acc += out[(blockIdx.x + 1) % gridDim.x];
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc + smem[0];
}

// hand-rolled barrier: one atomic counter, spin on a generation word
__global__ void __launch_bounds__(1024) k_spin(unsigned* out, unsigned* bar, int rounds) {
    unsigned acc = 0;
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x == 0) out[blockIdx.x] = r + acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned gen = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (atomicAdd(&bar[0], 1u) == gridDim.x - 1) {
                bar[0] = 0;
                __threadfence();
                atomicAdd(&bar[1], 1u);
            } else {
                while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
            }
            __threadfence();
        }
        __syncthreads();
        acc += out[(blockIdx.x + 1) % gridDim.x];
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

int main() {
    unsigned *out, *bar;
    hipMalloc(&out, 4096 * 4);
    hipMalloc(&bar, 64);
    hipMemset(bar, 0, 64);
    hipFuncSetAttribute((const void*)k_sync, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int rounds : {1, 11, 101}) {
        void* args[] = {&out, &rounds};
        float best = 1e9;
        for (int t = 0; t < 5; ++t) {
            hipEventRecord(a);
            hipError_t e = hipLaunchCooperativeKernel((const void*)k_sync, dim3(256), dim3(1024), args, 43 * 1024, 0);
            hipEventRecord(b);
            hipEventSynchronize(b);
            if (e != hipSuccess) { printf("cooperative launch failed: %s\n", hipGetErrorString(e)); return 1; }
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("grid.sync  rounds=%3d: %.1f us\n", rounds, best * 1e3);
        best = 1e9;
        for (int t = 0; t < 5; ++t) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k_spin, dim3(256), dim3(1024), 0, 0, out, bar, rounds);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("spin       rounds=%3d: %.1f us\n", rounds, best * 1e3);
    }
    return 0;
}
