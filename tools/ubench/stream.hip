// Micro-benchmarks of memory access patterns used by the classify kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -o stream stream.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// K1: grid-stride, one dword per lane per iteration
__global__ void k_scalar(const int* __restrict__ a, long n, int* out) {
    long stride = (long)gridDim.x * blockDim.x;
    int acc = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += a[i];
    if (acc == 123456789) out[0] = acc;
}
// K1v: grid-stride, int4 per lane
__global__ void k_vec4(const int4* __restrict__ a, long n4, int* out) {
    long stride = (long)gridDim.x * blockDim.x;
    int acc = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) { int4 v = a[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123456789) out[0] = acc;
}
// K2: dependent chain idx -> val (like qoff -> subj), scalar, grid-stride
__global__ void k_chain(const int* __restrict__ idx, const int* __restrict__ val, long n, int* out) {
    long stride = (long)gridDim.x * blockDim.x;
    int acc = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += val[idx[i]];
    if (acc == 123456789) out[0] = acc;
}
// K3: tile to LDS with barriers; tiles grid-strided (tile = it*grid + b) or block-contiguous
template <int T, bool CONTIG>
__global__ void k_tile(const int* __restrict__ a, long n, int* out) {
    __shared__ int buf[T];
    long ntiles = n / T;
    long per = (ntiles + gridDim.x - 1) / gridDim.x;
    int acc = 0;
    for (long it = 0; it < per; ++it) {
        long tile = CONTIG ? (long)blockIdx.x * per + it : it * gridDim.x + blockIdx.x;
        if (tile >= ntiles) break;
        buf[threadIdx.x] = a[tile * T + threadIdx.x];
        __syncthreads();
        acc += buf[(threadIdx.x * 7) % T];
        __syncthreads();
    }
    if (acc == 123456789) out[0] = acc;
}
// K4: two dependent tile phases (offsets tile, then record span) + gather from a small table
template <int T>
__global__ void k_tile2(const int* __restrict__ off, const int* __restrict__ rec, const int* __restrict__ tab, long n, int* out) {
    __shared__ int loff[T + 1];
    __shared__ int lrec[T];
    long ntiles = n / T;
    int acc = 0;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        loff[threadIdx.x] = off[tile * T + threadIdx.x];
        if (threadIdx.x == 0) loff[T] = off[tile * T + T];
        __syncthreads();
        int r0 = loff[0];
        lrec[threadIdx.x] = rec[r0 + threadIdx.x];
        __syncthreads();
        acc += tab[lrec[loff[threadIdx.x] - r0]];
        __syncthreads();
    }
    if (acc == 123456789) out[0] = acc;
}

template <typename F>
float timeit(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best;
}

int main() {
    const long n = 10'000'000;
    std::vector<int> h(n + 16), tabh(12576);
    for (long i = 0; i < n + 16; ++i) h[i] = (int)i;            // offsets 0..n  (1 record per read)
    std::vector<int> recs(n + 16);
    for (long i = 0; i < n + 16; ++i) recs[i] = (int)((i * 2654435761u) % 12576);
    int *d_off, *d_rec, *d_tab, *d_out;
    CK(hipMalloc(&d_off, (n + 16) * 4)); CK(hipMalloc(&d_rec, (n + 16) * 4)); CK(hipMalloc(&d_tab, 12576 * 4)); CK(hipMalloc(&d_out, 64));
    CK(hipMemcpy(d_off, h.data(), (n + 16) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rec, recs.data(), (n + 16) * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_tab, 0, 12576 * 4));
    for (int blocks : {512, 1024, 2048, 4096, 8192}) {
        float t1 = timeit([&] { hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(256), 0, 0, d_off, n, d_out); });
        float t2 = timeit([&] { hipLaunchKernelGGL(k_vec4, dim3(blocks), dim3(256), 0, 0, (const int4*)d_off, n / 4, d_out); });
        float t3 = timeit([&] { hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(256), 0, 0, d_off, d_rec, n, d_out); });
        printf("blocks %5d x256: scalar %7.1f us (%6.0f GB/s)  vec4 %7.1f us (%6.0f GB/s)  chain %7.1f us (%6.0f GB/s)\n", blocks,
               t1 * 1e3, 40e6 / t1 / 1e6, t2 * 1e3, 40e6 / t2 / 1e6, t3 * 1e3, 80e6 / t3 / 1e6);
    }
    for (int blocks : {256, 512, 1024, 2048}) {
        float t1 = timeit([&] { hipLaunchKernelGGL((k_tile<512, false>), dim3(blocks), dim3(512), 0, 0, d_off, n, d_out); });
        float t2 = timeit([&] { hipLaunchKernelGGL((k_tile<512, true>), dim3(blocks), dim3(512), 0, 0, d_off, n, d_out); });
        float t3 = timeit([&] { hipLaunchKernelGGL((k_tile2<512>), dim3(blocks), dim3(512), 0, 0, d_off, d_rec, d_tab, n, d_out); });
        float t4 = timeit([&] { hipLaunchKernelGGL((k_tile<1024, false>), dim3(blocks), dim3(1024), 0, 0, d_off, n, d_out); });
        printf("blocks %5d: tile512 strided %7.1f us (%6.0f GB/s)  contig %7.1f us (%6.0f GB/s)  tile2 %7.1f us (%6.0f GB/s)  tile1024 %7.1f us (%6.0f GB/s)\n",
               blocks, t1 * 1e3, 40e6 / t1 / 1e6, t2 * 1e3, 40e6 / t2 / 1e6, t3 * 1e3, 80e6 / t3 / 1e6, t4 * 1e3, 40e6 / t4 / 1e6);
    }
    return 0;
}
