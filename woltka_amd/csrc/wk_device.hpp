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

constexpr uint32_t kLogPartsMax = 1024;  // hash partitions of the miss log: 256 or 1024, chosen per launch

__host__ __device__ __forceinline__ uint64_t make_key(uint32_t job, uint32_t k, uint32_t group,
                                                      uint32_t feature) {
    return ((uint64_t)job << 61) | ((uint64_t)k << 49) | ((uint64_t)group << 28) | (uint64_t)feature;
}

// WK_WEIGHT_L / k for k in [1, 16] (exact: the quotient is an integer below
// 2^20, the reciprocal's error stays far below the 0.5 rounding margin), and
// the inverse.
__device__ __forceinline__ uint32_t weight_of(uint32_t k) {
    return (uint32_t)((float)WK_WEIGHT_L * __builtin_amdgcn_rcpf((float)k) + 0.5f);
}
constexpr uint64_t kKeyKMask = (uint64_t)WK_MAX_K << 49;

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
// O(#workgroups) device-scope atomics instead of O(#reads).  (Measured on
// MI355X: LDS atomics sustain > 1 T adds/s chip-wide even under skew, while
// device-scope atomics on one address serialise at ~15 ns each.)
//
// Layout: buckets of 4 slots, 64 B per bucket = keys[4] then vals[4], so one
// probe is two 16-byte LDS reads.  A key lives in the first free slot of its
// home bucket, else of the next bucket, else it is counted in HBM directly;
// slots never change once claimed, and every lane scans a bucket in the same
// order, so a key can never occupy two slots.
// ---------------------------------------------------------------------------
struct LdsCache {
    unsigned long long* base;  // [buckets * 8]
    uint32_t bmask;            // buckets - 1
    // dense bins for the bulk of the traffic when the id space is small: the
    // count of (job, k = 1, dense_group, feature < dense_bins) is one ds_add_u32,
    // no key compare, and the bins of all workgroups are merged without atomics
    uint32_t* dense;           // [n_jobs * dense_bins] or null
    uint32_t dense_bins;
    int32_t dense_group;       // the group the bins count (reads of other groups take the hash cache)
    // partitioned miss log: a key that finds no LDS slot is appended to the
    // stream of (this workgroup, hash partition) in HBM instead of paying a
    // device-scope atomic; partition_merge_kernel aggregates each partition in
    // LDS afterwards.  plog_cur[p] counts the appends of partition p.
    uint32_t* plog_cur;            // LDS [log_parts] or null
    unsigned long long* plog;      // HBM, this workgroup's [log_parts][plog_cap] keys
    uint32_t plog_cap;
    uint32_t plog_shift;           // 32 - log2(log_parts)
#ifdef WK_ABLATE
    uint32_t ablate;
#endif
};

__device__ __forceinline__ uint32_t hash_key(uint64_t key) {
    uint32_t h = (uint32_t)key ^ ((uint32_t)(key >> 32) * 0x9E3779B1u);
    h *= 0x85EBCA6Bu;
    return h ^ (h >> 15);
}

__device__ __forceinline__ void lds_cache_init(const LdsCache& c) {
    const uint32_t n = (c.bmask + 1) * 8;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) c.base[i] = (i & 4) ? 0ull : kEmptyKey;
    __syncthreads();
}

// try to count `key` in bucket b; true on success
__device__ __forceinline__ bool bucket_add(unsigned long long* bk, uint64_t key, unsigned long long w) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    // (the LDS address space spelled out: a volatile load through a generic
    // pointer compiles to flat_load ... sc0 sc1 with a wait after each, not to
    // two ds_read_b128 in flight together)
    typedef const volatile __attribute__((address_space(3))) u64x2* lds_keys_t;
    const lds_keys_t keys = (lds_keys_t)bk;
    const u64x2 k01 = keys[0];
    const u64x2 k23 = keys[1];
    unsigned long long snap[4] = {k01.x, k01.y, k23.x, k23.y};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned long long cur = snap[i];
        if (cur == kEmptyKey) {
            cur = atomicCAS(&bk[i], (unsigned long long)kEmptyKey, (unsigned long long)key);
            if (cur == kEmptyKey) cur = key;
        }
        if (cur == key) {
            atomicAdd(&bk[4 + i], w);
            return true;
        }
    }
    return false;
}

__device__ __forceinline__ void cached_add(const LdsCache& c, const CountTable& t, uint64_t key,
                                           unsigned long long w) {
    const uint32_t b = hash_key(key) & c.bmask;
    if (bucket_add(c.base + (size_t)b * 8, key, w)) return;
    // with a miss log behind the cache a second probe costs more than a miss
    // (sending list results straight to the log was tried: the extra live
    // value spills registers in the evaluator and costs more than it saves)
    if (!c.plog_cur && bucket_add(c.base + (size_t)((b + 1) & c.bmask) * 8, key, w)) return;
#ifdef WK_ABLATE
    if (c.ablate & 16) return;  // measurement only: drop cache misses
#endif
    if (c.plog_cur) {
        // partition by the high hash bits (the low bits pick the LDS bucket)
        const uint32_t part = (hash_key(key) * 0x9E3779B1u) >> c.plog_shift;
        const uint32_t pos = atomicAdd(&c.plog_cur[part], 1u);
        if (pos < c.plog_cap) {
            // a log entry is one contribution: a weighted key (k field 0)
            // travels with its k = L / w in the k field
            const uint64_t entry = (key & kKeyKMask) ? key : key | ((uint64_t)weight_of((uint32_t)w) << 49);
            c.plog[(size_t)part * c.plog_cap + pos] = entry;
            return;
        }
    }
    table_add(t, key, w);  // no log, or its stream is full: count in HBM directly
}

__device__ __forceinline__ void lds_cache_flush(const LdsCache& c, const CountTable& t) {
    __syncthreads();
#ifdef WK_ABLATE
    if (c.ablate & 2) return;
#endif
    const uint32_t n = (c.bmask + 1) * 4;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t slot = (i >> 2) * 8 + (i & 3);
        const unsigned long long k = c.base[slot];
        if (k != kEmptyKey) table_add(t, k, c.base[slot + 4]);
    }
}

// wave-level sum of a per-lane 64-bit value (all 64 lanes must call)
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

}  // namespace wk
