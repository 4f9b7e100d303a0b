// Launch floor of a 256 x 1024-thread grid with dynamic LDS: empty body, LDS
// clear only, and clear + one dependent global load chain, timed back to back
// with HIP events (what a kernel of the count-first pass pays before and after
// its streaming loop).   hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(1024) k_empty(int* out) {
    if (out && threadIdx.x == 2048) out[0] = 1;
}

__global__ void __launch_bounds__(1024) k_clear(int* out, int words) {
    extern __shared__ int lds[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    if (out && lds[threadIdx.x] == 7) out[0] = 1;
}

__global__ void __launch_bounds__(1024) k_chain(int* out, const int* a, const int* b, int words, int n) {
    extern __shared__ int lds[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) % n;
    const int s = a[i];          // offsets ...
    const int c = b[s % n];      // ... then the subject index
    atomicAdd(&lds[c % words], 1);
    __syncthreads();
    for (int j = threadIdx.x; j < words; j += blockDim.x) out[blockIdx.x * words + j] = lds[j];
}

int main() {
    const int words = 10575, n = 1 << 20;
    int *out, *a, *b;
    hipMalloc(&out, 256 * words * 4);
    hipMalloc(&a, n * 4);
    hipMalloc(&b, n * 4);
    hipMemset(a, 0, n * 4);
    hipMemset(b, 0, n * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 200;
    for (int variant = 0; variant < 3; ++variant) {
        for (int warm = 0; warm < 2; ++warm) {
            hipEventRecord(e0);
            for (int r = 0; r < reps; ++r) {
                if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(256), dim3(1024), words * 4, 0, out);
                if (variant == 1) hipLaunchKernelGGL(k_clear, dim3(256), dim3(1024), words * 4, 0, out, words);
                if (variant == 2) hipLaunchKernelGGL(k_chain, dim3(256), dim3(1024), words * 4, 0, out, a, b, words, n);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("variant %d: %.2f us per launch (back to back)\n", variant, ms * 1000 / reps);
    }
    return 0;
}
