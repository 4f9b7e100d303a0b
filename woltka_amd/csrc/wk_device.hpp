// wk_device.hpp — device-side building blocks shared by the kernels.
// gfx950 (CDNA4) only: 64-lane wavefronts, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/woltka_hip.h"

namespace wk {

constexpr uint64_t kEmptyKey = ~0ull;
constexpr int kWave = 64;

// device-visible error bits (OR-ed into ctx->d_err)
constexpr int kErrTableFull = 1;
constexpr int kErrKRange = 2;
constexpr int kErrFeatureRange = 4;
constexpr int kErrGroupRange = 8;
constexpr int kErrPairOverflow = 16;

__host__ __device__ __forceinline__ uint64_t make_key(uint32_t job, uint32_t k, uint32_t group,
                                                      uint32_t feature) {
    return ((uint64_t)job << 61) | ((uint64_t)k << 49) | ((uint64_t)group << 28) | (uint64_t)feature;
}

// 64-bit finaliser (splitmix64) — spreads the low feature bits over the table.
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

// ---------------------------------------------------------------------------
// Global count table: open addressing, linear probing, keys claimed by CAS.
// A slot's key goes EMPTY -> key exactly once per clear, so a stale read can
// only ever see EMPTY, which the CAS then corrects.
// ---------------------------------------------------------------------------
struct CountTable {
    unsigned long long* keys;
    unsigned long long* vals;
    uint64_t mask;  // slots - 1
    int* err;
};

__device__ __forceinline__ void table_add(const CountTable& t, uint64_t key, unsigned long long w) {
    uint64_t h = mix64(key) & t.mask;
    for (uint64_t probe = 0; probe <= t.mask; ++probe) {
        unsigned long long cur =
            __hip_atomic_load(&t.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == kEmptyKey) {
            cur = atomicCAS(&t.keys[h], (unsigned long long)kEmptyKey, (unsigned long long)key);
            if (cur == kEmptyKey) cur = key;
        }
        if (cur == key) {
            atomicAdd(&t.vals[h], w);
            return;
        }
        h = (h + 1) & t.mask;
    }
    atomicOr(t.err, kErrTableFull);
}

// ---------------------------------------------------------------------------
// Per-workgroup LDS front cache for the count table.  Hot keys (Zipf-skewed
// taxa) are absorbed in LDS and flushed once per workgroup, so a hot bin sees
// O(#workgroups) device-scope atomics instead of O(#reads).
// ---------------------------------------------------------------------------
struct LdsCache {
    unsigned long long* keys;  // [slots]
    unsigned long long* vals;  // [slots]
    uint32_t mask;
};

constexpr int kLdsProbes = 8;

__device__ __forceinline__ void lds_cache_init(const LdsCache& c) {
    for (uint32_t i = threadIdx.x; i <= c.mask; i += blockDim.x) {
        c.keys[i] = kEmptyKey;
        c.vals[i] = 0ull;
    }
    __syncthreads();
}

__device__ __forceinline__ void cached_add(const LdsCache& c, const CountTable& t, uint64_t key,
                                           unsigned long long w) {
    uint32_t h = (uint32_t)(mix64(key) >> 20) & c.mask;
#pragma unroll 1
    for (int p = 0; p < kLdsProbes; ++p) {
        unsigned long long cur =
            __hip_atomic_load(&c.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == kEmptyKey) {
            cur = atomicCAS(&c.keys[h], (unsigned long long)kEmptyKey, (unsigned long long)key);
            if (cur == kEmptyKey) cur = key;
        }
        if (cur == key) {
            atomicAdd(&c.vals[h], w);
            return;
        }
        h = (h + 1) & c.mask;
    }
    table_add(t, key, w);  // cache neighbourhood full: go to HBM directly
}

__device__ __forceinline__ void lds_cache_flush(const LdsCache& c, const CountTable& t) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= c.mask; i += blockDim.x) {
        unsigned long long k = c.keys[i];
        if (k != kEmptyKey) table_add(t, k, c.vals[i]);
    }
}

// wave-level sum of a per-lane 64-bit value (all 64 lanes must call)
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

}  // namespace wk
