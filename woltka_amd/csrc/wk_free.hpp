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
    // the lowest common ancestor without a walk: node_rank[v] = rank
    // of node v among the subjects ordered by pre-order id (-1: no subject), and the
    // sparse table over the LCAs of rank-adjacent subjects
    const int32_t* node_rank;
    const int32_t* sparse;   // [levels][sparse_m]
    uint32_t sparse_m;
    uint32_t job, group;
    uint32_t subok, unassigned;
    CountTable table;
    unsigned long long* plog;   // [gridDim.x][log_parts][plog_cap]
    uint32_t* plog_cnt;
    uint32_t plog_cap, log_parts;
    unsigned long long* stat_block;
};


// The results go straight to the partitioned log (one LDS counter bump and one
// 8-byte store each; partition_merge_kernel counts them): a read's result is one
// of ~10^5 nodes, an LDS hash cache of 8 k slots in front of the log would miss
// nearly always and its probes were the larger half of the first version's time.
// Without the cache the kernel needs 1-4 KB of LDS and runs several workgroups
// per CU: more dependent gather chains in flight.
template <int kWin>
__global__ void __launch_bounds__(kFreeThreads) free_stream_kernel(FreeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long acc[2];
    uint32_t* const plog_cur = reinterpret_cast<uint32_t*>(smem);
    unsigned long long* const plog = a.plog + (size_t)blockIdx.x * a.log_parts * a.plog_cap;
    const uint32_t plog_shift = (uint32_t)__clz((int)a.log_parts) + 1u;
    for (uint32_t i = threadIdx.x; i < a.log_parts; i += blockDim.x) plog_cur[i] = 0u;
    if (threadIdx.x < 2) acc[threadIdx.x] = 0ull;
    __syncthreads();
    auto count = [&](uint32_t feature) {
        const uint64_t key = make_key(a.job, 0u, a.group, feature);
        const uint32_t part = (hash_key(key) * 0x9E3779B1u) >> plog_shift;
        const uint32_t pos = atomicAdd(&plog_cur[part], 1u);
        if (pos < a.plog_cap)
            plog[(size_t)part * a.plog_cap + pos] = key | (1ull << 49);  // k = 1: one whole read (weight L under the k = 0 key)
        else
            table_add(a.table, key, (unsigned long long)WK_WEIGHT_L);
    };

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t wave0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    // window w looks at records [48 w - 16, 48 w + 48) and owns [48 w, 48 w + 48)
    const uint32_t n_windows = (a.n_records + kFreeStride - 1) / kFreeStride;
    unsigned long long my_reads = 0, my_records = 0;
    for (uint32_t w0 = wave0 * kWin; w0 < n_windows; w0 += waves * kWin) {
        uint32_t word[kWin], mn[kWin], mx[kWin];
        int32_t res[kWin];       // result node, -1 = none (yet)
        int32_t p[kWin], q[kWin];
        bool owner[kWin], multi[kWin];
#pragma unroll
        for (int u = 0; u < kWin; ++u) {
            const int64_t idx = (int64_t)(w0 + u) * kFreeStride - 16 + (int64_t)lane;
            word[u] = (w0 + u < n_windows && idx >= 0 && idx < (int64_t)a.n_records) ? a.words[idx] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kWin; ++u) {
            const uint32_t size = word[u] >> kWordSizeShift, pos = (word[u] >> kWordSubjBits) & 15u;
            mn[u] = mx[u] = word[u] & kWordSubjMask;
            // segmented inclusive min / max over the lanes of a read (its records are
            // consecutive, lane - d belongs to the same read iff pos >= d)
#pragma unroll
            for (uint32_t d = 1; d < 16u; d <<= 1) {
                const uint32_t pmn = __shfl_up(mn[u], d, kWave), pmx = __shfl_up(mx[u], d, kWave);
                if (pos >= d && lane >= d) {
                    mn[u] = pmn < mn[u] ? pmn : mn[u];
                    mx[u] = pmx > mx[u] ? pmx : mx[u];
                }
            }
            owner[u] = size != 0u && pos + 1u == size && lane >= 16u;
            // (a missing subject carries the largest value of the field: it is the maximum)
            multi[u] = owner[u] && size > 1u && mx[u] != kFreeMissing && mn[u] != mx[u];
            my_records += (size != 0u && lane >= 16u) ? 1ull : 0ull;
            my_reads += owner[u] ? 1ull : 0ull;
            res[u] = -1;
            p[u] = q[u] = -1;
        }
        // first round of gathers: the parent of a sole subject, the ranks of the extremes
#pragma unroll
        for (int u = 0; u < kWin; ++u) {
            if (!owner[u]) continue;
            const uint32_t size = word[u] >> kWordSizeShift;
            if (size == 1u) {
                if (mn[u] != kFreeMissing) res[u] = a.subok ? (int32_t)mn[u] : a.nodes[mn[u]].parent;
            } else if (multi[u]) {
                p[u] = a.node_rank[mn[u]];
                q[u] = a.node_rank[mx[u]];
            } else if (mx[u] != kFreeMissing) {
                res[u] = mn[u] == 0u ? -1 : (int32_t)mn[u];  // the same node several times (cannot happen with sets): itself, None if the root
            }
        }
        // second round: the shallowest LCA of rank-adjacent subjects over [p, q)
#pragma unroll
        for (int u = 0; u < kWin; ++u) {
            if (!multi[u]) continue;
            const uint32_t span = (uint32_t)(q[u] - p[u]);  // >= 1
            const uint32_t k = 31u - (uint32_t)__clz((int)span);
            const int32_t* row = a.sparse + (size_t)k * a.sparse_m;
            const int32_t x = row[p[u]], y = row[(uint32_t)q[u] - (1u << k)];
            const int32_t anc = x < y ? x : y;
            res[u] = anc == 0 ? -1 : anc;
        }
#pragma unroll
        for (int u = 0; u < kWin; ++u) {
            if (!owner[u]) continue;
            if (res[u] >= 0)
                count((uint32_t)res[u]);
            else if (a.unassigned)
                count((uint32_t)WK_FEATURE_UNASSIGNED);
        }
    }
    my_reads = wave_sum(my_reads);
    my_records = wave_sum(my_records);
    if (lane == 0) {
        atomicAdd(&acc[0], my_reads);
        atomicAdd(&acc[1], my_records);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.stat_block[2 * blockIdx.x] += acc[0];
        a.stat_block[2 * blockIdx.x + 1] += acc[1];
    }
    for (uint32_t i = threadIdx.x; i < a.log_parts; i += blockDim.x) {
        const uint32_t n = plog_cur[i];
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
