// wk_free.hpp — `--rank free` over the packed record stream.
//
// classify.assign_free (woltka/classify.py:54-78) followed by classify.counter
// (classify.py:144-171), for a job set that is one free-rank job: a read with
// one subject goes to the subject's parent (to the subject itself under
// --subok), a read with several to their lowest common ancestor (tree.find_lca,
// tree.py:513-566), nowhere if that is the root or a subject is not in the tree.
//
// The records arrive as packed words (wk_weigh.hpp) whose subject field holds
// the subject's *feature id* here — the pre-order id of its node, or
// kFreeMissing for a name that is not a node (the tokenizer on the device
// writes them that way, host-tokenised chunks are translated when they are
// appended).  Pre-order ids turn find_lca into "lowest ancestor of the smallest
// id whose subtree holds the largest" (DESIGN §2), so a read needs the minimum
// and the maximum of its records' ids and nothing else.  The position and size
// in every word make the stream self-describing: a wave looks at 64
// consecutive records, a segmented min/max over the lanes of a read takes four
// shuffle steps, and the lane of a read's last record owns the read.  Windows
// advance by 48 records, so that a read (<= 16 records) always lies inside the
// window of the wave that owns it — no offsets, no per-read gathers of
// candidate rows (round 2's evaluator: 225 M 16-byte row gathers, 1.5 ms).
#pragma once
#include "wk_classify.hpp"
#include "wk_device.hpp"
#include "wk_weigh.hpp"

namespace wk {

constexpr uint32_t kFreeMissing = kWordSubjMask;  // feature field of a subject that is not in the tree
constexpr uint32_t kFreeStride = 48;              // records a wave owns per window of 64
constexpr uint32_t kFreeThreads = 1024;

struct FreeArgs {
    const uint32_t* words;  // [n_records] feature | position << 23 | size << 27
    uint32_t n_records;
    const Node* nodes;
    uint32_t n_nodes;
    uint32_t job, group;
    uint32_t subok, unassigned;
    CountTable table;
    unsigned long long* plog;   // [gridDim.x][log_parts][plog_cap]
    uint32_t* plog_cnt;
    uint32_t plog_cap, log_parts;
    unsigned long long* stat_block;
};

__global__ void __launch_bounds__(kFreeThreads) free_stream_kernel(FreeArgs a, uint32_t lds_slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long acc[2];
    LdsCache cache{};
    cache.base = reinterpret_cast<unsigned long long*>(smem);
    cache.bmask = lds_slots / 4 - 1;
    cache.plog_cur = reinterpret_cast<uint32_t*>(smem + (size_t)lds_slots * 16);
    cache.plog = a.plog + (size_t)blockIdx.x * a.log_parts * a.plog_cap;
    cache.plog_cap = a.plog_cap;
    cache.plog_shift = (uint32_t)__clz((int)a.log_parts) + 1u;
    for (uint32_t i = threadIdx.x; i < a.log_parts; i += blockDim.x) cache.plog_cur[i] = 0u;
    if (threadIdx.x < 2) acc[threadIdx.x] = 0ull;
    lds_cache_init(cache);  // (ends with a barrier)

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t wave0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    // window w looks at records [48 w - 16, 48 w + 48) and owns [48 w, 48 w + 48)
    const uint32_t n_windows = (a.n_records + kFreeStride - 1) / kFreeStride;
    unsigned long long my_reads = 0, my_records = 0;
    for (uint32_t w = wave0; w < n_windows; w += waves) {
        const int64_t idx = (int64_t)w * kFreeStride - 16 + (int64_t)lane;
        const uint32_t word = (idx >= 0 && idx < (int64_t)a.n_records) ? a.words[idx] : 0u;
        const uint32_t size = word >> kWordSizeShift, pos = (word >> kWordSubjBits) & 15u;
        uint32_t mn = word & kWordSubjMask, mx = mn;
        // segmented inclusive min / max over the lanes of a read (its records are
        // consecutive, lane - d belongs to the same read iff pos >= d)
#pragma unroll
        for (uint32_t d = 1; d < 16u; d <<= 1) {
            const uint32_t pmn = __shfl_up(mn, d, kWave), pmx = __shfl_up(mx, d, kWave);
            if (pos >= d && lane >= d) {
                mn = pmn < mn ? pmn : mn;
                mx = pmx > mx ? pmx : mx;
            }
        }
        const bool owner = size != 0u && pos + 1u == size && lane >= 16u;
        my_records += (size != 0u && lane >= 16u) ? 1ull : 0ull;
        if (owner) {
            my_reads += 1;
            // (a missing subject carries the largest value of the field: it is the maximum)
            int32_t res = -1;
            if (size == 1u) {
                if (mn != kFreeMissing) res = a.subok ? (int32_t)mn : a.nodes[mn].parent;
            } else if (mx != kFreeMissing) {
                uint32_t anc = mn;
                while ((uint32_t)a.nodes[anc].last < mx) anc = (uint32_t)a.nodes[anc].parent;
                res = anc == 0u ? -1 : (int32_t)anc;
            }
            if (res >= 0)
                cached_add(cache, a.table, make_key(a.job, 0u, a.group, (uint32_t)res), (unsigned long long)WK_WEIGHT_L);
            else if (a.unassigned)
                cached_add(cache, a.table, make_key(a.job, 0u, a.group, (uint32_t)WK_FEATURE_UNASSIGNED),
                           (unsigned long long)WK_WEIGHT_L);
        }
    }
    my_reads = wave_sum(my_reads);
    my_records = wave_sum(my_records);
    if (lane == 0) {
        atomicAdd(&acc[0], my_reads);
        atomicAdd(&acc[1], my_records);
    }
    lds_cache_flush(cache, a.table);  // (starts with a barrier)
    if (threadIdx.x == 0) {
        a.stat_block[2 * blockIdx.x] += acc[0];
        a.stat_block[2 * blockIdx.x + 1] += acc[1];
    }
    for (uint32_t i = threadIdx.x; i < a.log_parts; i += blockDim.x) {
        const uint32_t n = cache.plog_cur[i];
        a.plog_cnt[(size_t)blockIdx.x * a.log_parts + i] = n < a.plog_cap ? n : a.plog_cap;
    }
}

// subject indices -> feature ids in place (chunks of the host tokenizer appended
// to a free-rank accumulation)
__global__ void __launch_bounds__(256) words_to_features_kernel(uint32_t* __restrict__ words, uint32_t n,
                                                                const int32_t* __restrict__ subj_feat, uint32_t n_subjects,
                                                                uint32_t n_nodes, int* __restrict__ err) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t w = words[i], s = w & kWordSubjMask;
    uint32_t f = kFreeMissing;
    if (s < n_subjects) {
        const uint32_t x = (uint32_t)subj_feat[s];
        f = x < n_nodes ? x : kFreeMissing;
    } else if (w >> kWordSizeShift) {
        atomicOr(err, kErrFeatureRange);
    }
    words[i] = (w & ~kWordSubjMask) | f;
}

}  // namespace wk
