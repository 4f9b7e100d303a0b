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
// in every word make the stream self-describing: the last record of a read
// says where the read began — no offsets, no per-read gathers of candidate
// rows (round 2's evaluator: 225 M 16-byte row gathers, 1.5 ms).
//
// The same stream serves one rank job under an option that looks at whole reads
// (classify.assign_rank, classify.py:81-141: --uniq, --above, --major): the
// records then hold each subject's ancestor at the rank (kFreeMissing: none), a
// read whose records agree goes there, and one whose records differ goes to
// their LCA (--above; nowhere if a record has no ancestor at the rank), to the
// value that reaches the threshold (--major: one vote per subject, "none" is a
// value that can win and then assigns nothing) or nowhere (--uniq).
#pragma once
#include "wk_classify.hpp"
#include "wk_device.hpp"
#include "wk_weigh.hpp"

namespace wk {

constexpr uint32_t kFreeMissing = kWordSubjMask;  // feature field of a subject that is not in the tree
constexpr uint32_t kFreeThreads = 1024;

// rank of a subject's node among the distinct subject nodes in pre-order: a bit
// per node and a running count per 64 of them (16 B per 64 nodes: 512 KB for a
// 2 M-node tree, where a plain rank per node is 8 MB and does not stay in an
// XCD's L2)
struct RankBlock {
    unsigned long long bits;
    uint32_t before, pad;
};

struct FreeArgs {
    const uint32_t* words;  // [n_records] feature | position << 23 | size << 27
    uint32_t n_records;
    // the lowest common ancestor without a walk over the distinct
    // subject nodes d_0 < d_1 < ... in pre-order: sparse[k][i] = the smallest among
    // LCA(d_j, d_j+1), j in [i, i + 2^k); parent_d[i] = parent of d_i, self_d[i] =
    // d_i — all three as *result ids*: the possible results (subject nodes and
    // their ancestors) numbered in pre-order, the root 0
    const RankBlock* rblocks;  // [n_nodes / 64 + 1]
    const int32_t* sparse;     // [levels][sparse_m]
    const int32_t* parent_d;   // [sparse_m]
    const int32_t* self_d;     // [sparse_m]
    uint32_t sparse_m;
    uint32_t job, group;
    uint32_t subok, unassigned;
    // a rank job under --uniq / --above / --major instead of the free-rank job:
    uint32_t by_rank, above;
    double major;  // > 0.5 (a value that reaches it is the only one that can), or 0
    uint32_t count_stats;  // reads and records into stat_block (the first of several jobs over the same records)
    // reads per result id ([n_results]: 'Unassigned'), all zero between launches:
    // free_counts_kernel moves them to the count table and clears them
    uint32_t* dense;  // [n_results + 1]
    uint32_t n_results;
    unsigned long long* stat_block;
};

constexpr uint32_t kFreeQueue = 128;    // per-wave queue of reads to evaluate: < 64 left over + <= 64 new
constexpr uint32_t kFreeBlock = 256;    // records a wave loads at a time
constexpr uint32_t kFreeAdvance = 240;  // ... of which it owns the last 240
constexpr uint32_t kFreeWaveLds = kFreeQueue * 8 + kFreeBlock * 4 + kFreeBlock * 2;  // queue, staged block, list of read ends

__device__ __forceinline__ uint32_t rank_of(const RankBlock* __restrict__ blocks, uint32_t node) {
    const uint4 b = *reinterpret_cast<const uint4*>(blocks + (node >> 6));
    const unsigned long long bits = ((unsigned long long)b.y << 32) | b.x;
    const uint32_t bit = node & 63u;
    return b.z + (uint32_t)__popcll(bits & ((1ull << bit) - 1ull));
}

// A wave takes blocks of 256 records — one 16-byte load per lane, the next
// block's issued before this one is looked at — of which it owns the last 240
// (a read of <= 16 records lies inside the block that owns its last record), and
// works on a block in three steps, each with the lanes on the unit that costs
// least there (the stream is bound by vector instructions, ~1 per record now; a
// version with one record per lane and a segmented min/max across lanes spent 8
// cycles per record and SIMD):
//   1. a lane looks at its own four records and marks the last records of reads;
//      four ballots turn the marks into a list of read ends (LDS, 16-bit);
//   2. a lane takes a read end off the list and walks back over that read's
//      records in the staged block for the smallest and the largest id — no
//      work shared between lanes, so none repeated — and queues what the read
//      needs: {smallest, largest, what to do};
//   3. 64 queued reads at a time: the table gathers and the counting, every lane
//      busy, as in a kernel with one read per lane but without that kernel's
//      gathers of the reads' records.
__global__ void __launch_bounds__(kFreeThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) free_stream_kernel(FreeArgs a, uint32_t lds_slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long acc[2];
    // The counting: results are node ids and the job is one, so a slot of the
    // workgroup's LDS cache is {node, reads} in 8 bytes, and what the cache cannot
    // hold goes to a dense array of counters in HBM by a fire-and-forget atomic —
    // no log of misses to merge afterwards.  (Tried and not faster: one array per
    // XCD, picked by HW_REG_XCC_ID, with atomics of workgroup scope; counters
    // indexed by node id, 8 MB, were as fast as these ~1 MB.)
    uint32_t* const dense = a.dense;
    uint32_t* const ckeys = reinterpret_cast<uint32_t*>(smem);
    uint32_t* const ccnt = ckeys + lds_slots;
    const uint32_t cshift = (uint32_t)__clz((int)lds_slots) + 1u;
    for (uint32_t i = threadIdx.x; i < lds_slots; i += blockDim.x) {
        ckeys[i] = 0xFFFFFFFFu;
        ccnt[i] = 0u;
    }
    if (threadIdx.x < 2) acc[threadIdx.x] = 0ull;
    __syncthreads();
    auto count = [&](uint32_t node) {
        uint32_t h = (node * 0x9E3779B1u) >> cshift;
#pragma unroll
        for (int probe = 0; probe < 2; ++probe) {
            uint32_t k = ckeys[h];
            if (k == 0xFFFFFFFFu) {
                k = atomicCAS(&ckeys[h], 0xFFFFFFFFu, node);
                if (k == 0xFFFFFFFFu) k = node;
            }
            if (k == node) {
                atomicAdd(&ccnt[h], 1u);
                return;
            }
            h = (h + 1u) & (lds_slots - 1u);
        }
        atomicAdd(&dense[node], 1u);
    };
    // (plain LDS pointers: a volatile one turns the accesses into flat ones, each with a wait)
    unsigned char* const mine = smem + (size_t)lds_slots * 8 + (size_t)(threadIdx.x >> 6) * kFreeWaveLds;
    unsigned long long* const queue = reinterpret_cast<unsigned long long*>(mine);
    uint32_t* const stage = reinterpret_cast<uint32_t*>(mine + kFreeQueue * 8);
    unsigned short* const ends = reinterpret_cast<unsigned short*>(mine + kFreeQueue * 8 + kFreeBlock * 4);

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t wave0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    // entry kinds (bits 62-63): 0 = the result id itself, 1 = parent of a node, 2 = LCA of two, 3 = the node
    constexpr unsigned long long kParent = 1ull << 62, kLca = 2ull << 62, kSelf = 3ull << 62;
    auto evaluate = [&](uint32_t head, uint32_t n) {
        const bool on = lane < n;
        const unsigned long long e = on ? queue[(head + lane) & (kFreeQueue - 1)] : 0ull;
        const uint32_t kind = (uint32_t)(e >> 62), lo = (uint32_t)e & kWordSubjMask, hi = (uint32_t)(e >> 32) & kWordSubjMask;
        int32_t res = kind == 0u && on ? (int32_t)(uint32_t)e : -1;
        if (kind != 0u) {
            const uint32_t p = rank_of(a.rblocks, lo);
            if (kind == 1u) {
                res = a.parent_d[p];
            } else if (kind == 3u) {
                res = a.self_d[p];
                if (res == 0 && hi != 0u) res = -1;  // (several records, all the root: None)
            } else {
                const uint32_t q = rank_of(a.rblocks, hi);
                const uint32_t k = 31u - (uint32_t)__clz((int)(q - p));  // q > p: distinct nodes
                const int32_t* row = a.sparse + (size_t)k * a.sparse_m;
                const int32_t x = row[p], y = row[q - (1u << k)];
                const int32_t anc = x < y ? x : y;
                res = anc == 0 ? -1 : anc;
            }
            if (res < 0 && a.unassigned) res = (int32_t)a.n_results;
        }
        if (res >= 0) count((uint32_t)res);
    };
    // the queue, the list and the staged block are the wave's own: its LDS accesses
    // complete in order, the fences keep the compiler from moving them across
    auto settle = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // block b looks at records [240 b - 16, 240 b + 240) and owns the last 240
    const uint32_t n_blocks = (a.n_records + kFreeAdvance - 1) / kFreeAdvance;
    auto load_block = [&](uint32_t b) -> uint4 {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (b >= n_blocks) return v;
        const int64_t r0 = (int64_t)b * kFreeAdvance - 16 + 4 * (int64_t)lane;
        if (r0 >= 0 && r0 + 4 <= (int64_t)a.n_records) {
            v = *reinterpret_cast<const uint4*>(a.words + r0);
        } else {  // the stream's two ends
            if (r0 >= 0 && r0 < (int64_t)a.n_records) v.x = a.words[r0];
            if (r0 + 1 >= 0 && r0 + 1 < (int64_t)a.n_records) v.y = a.words[r0 + 1];
            if (r0 + 2 >= 0 && r0 + 2 < (int64_t)a.n_records) v.z = a.words[r0 + 2];
            if (r0 + 3 >= 0 && r0 + 3 < (int64_t)a.n_records) v.w = a.words[r0 + 3];
        }
        return v;
    };
    unsigned long long my_reads = 0, my_records = 0;
    uint32_t head = 0, tail = 0;  // (wave-uniform)
    uint4 cur = load_block(wave0);
    for (uint32_t b = wave0; b < n_blocks; b += waves) {
        const uint4 nxt = load_block(b + waves);
        *reinterpret_cast<uint4*>(stage + 4 * lane) = cur;
        // 1. the read ends among this lane's four records (lanes 0-3 hold the 16
        // records before the owned range)
        const uint32_t w4[4] = {cur.x, cur.y, cur.z, cur.w};
        uint32_t n_ends = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t size = w4[j] >> kWordSizeShift, pos = (w4[j] >> kWordSubjBits) & 15u;
            const bool mine_j = size != 0u && lane >= 4u;
            const bool last = mine_j && pos + 1u == size;
            my_records += mine_j ? 1ull : 0ull;
            my_reads += last ? 1ull : 0ull;
            const unsigned long long mask = __ballot(last);
            if (last) ends[n_ends + (uint32_t)__popcll(mask & below)] = (unsigned short)(4u * lane + (uint32_t)j);
            n_ends += (uint32_t)__popcll(mask);
        }
        settle();
        // 2. one read per lane
        for (uint32_t base = 0; base < n_ends; base += kWave) {
            const bool on = base + lane < n_ends;
            const uint32_t at = on ? (uint32_t)ends[base + lane] : 0u;
            const uint32_t word = on ? stage[at] : 0u;
            const uint32_t size = word >> kWordSizeShift;
            uint32_t mn = word & kWordSubjMask, mx = mn;
            // (four records a step, so that their LDS reads are in flight together;
            // steps beyond the read's first record look at that one again)
            const uint32_t first = at + 1u - (size ? size : 1u);
            for (uint32_t i = 1; __ballot(i < size) != 0ull; i += 4) {
                uint32_t v[4];
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    const uint32_t k = at - i - j;
                    v[j] = stage[(int32_t)k > (int32_t)first ? k : first];
                }
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    const uint32_t f = v[j] & kWordSubjMask;
                    mn = f < mn ? f : mn;
                    mx = f > mx ? f : mx;
                }
            }
            // (a missing subject carries the largest value of the field: it is the maximum)
            unsigned long long e = 0;
            bool put = on;
            if (a.by_rank) {
                uint32_t to = kFreeMissing;  // where the read goes without a look at the tree
                bool lca = false;
                if (mn == mx) {
                    to = mn;
                } else if (a.major > 0.0) {
                    // the only value that can reach a threshold above one half:
                    // Boyer-Moore's candidate, then its votes
                    uint32_t cand = 0, lead = 0, votes = 0;
                    for (uint32_t i = 0; __ballot(i < size) != 0ull; ++i)
                        if (i < size) {
                            const uint32_t f = stage[at - i] & kWordSubjMask;
                            if (lead == 0u) cand = f;
                            lead += (f == cand) ? 1u : (uint32_t)-1;
                        }
                    for (uint32_t i = 0; __ballot(i < size) != 0ull; ++i)
                        if (i < size) votes += ((stage[at - i] & kWordSubjMask) == cand) ? 1u : 0u;
                    if ((double)votes >= (double)size * a.major) to = cand;
                } else if (a.above) {
                    lca = mx != kFreeMissing;
                }
                if (lca)
                    e = kLca | ((unsigned long long)mx << 32) | mn;
                else if (to != kFreeMissing)
                    e = kSelf | to;
                else
                    e = (unsigned long long)a.n_results, put = on && a.unassigned != 0u;
            } else if (mx == kFreeMissing) {
                e = (unsigned long long)a.n_results, put = on && a.unassigned != 0u;
            } else if (size == 1u) {
                e = (a.subok ? kSelf : kParent) | mn;
            } else if (mn != mx) {
                e = kLca | ((unsigned long long)mx << 32) | mn;
            } else {  // the same node several times (cannot happen with sets): itself, None if the root
                e = kSelf | (1ull << 32) | mn;
            }
            const unsigned long long mask = __ballot(put);
            if (put) queue[(tail + (uint32_t)__popcll(mask & below)) & (kFreeQueue - 1)] = e;
            tail += (uint32_t)__popcll(mask);
            settle();
            // 3. (< 64 left over + <= 64 new fit the queue)
            if (tail - head >= (uint32_t)kWave) {
                evaluate(head, kWave);
                head += kWave;
                settle();
            }
        }
        cur = nxt;
    }
    settle();
    if (tail != head) evaluate(head, tail - head);
    my_reads = wave_sum(my_reads);
    my_records = wave_sum(my_records);
    if (lane == 0) {
        atomicAdd(&acc[0], my_reads);
        atomicAdd(&acc[1], my_records);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < lds_slots; i += blockDim.x)
        if (ccnt[i]) atomicAdd(&dense[ckeys[i]], ccnt[i]);
    if (threadIdx.x == 0 && a.count_stats) {
        a.stat_block[2 * blockIdx.x] += acc[0];
        a.stat_block[2 * blockIdx.x + 1] += acc[1];
    }
}

// the dense counters -> the count table (weight L per read), cleared on the way
__global__ void __launch_bounds__(256) free_counts_kernel(uint32_t* __restrict__ dense, uint32_t n_results,
                                                          const int32_t* __restrict__ result_node, uint32_t job, uint32_t group,
                                                          CountTable table) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_results) return;
    const uint32_t n = dense[i];
    if (!n) return;
    dense[i] = 0u;
    table_add(table, make_key(job, 0u, group, i == n_results ? (uint32_t)WK_FEATURE_UNASSIGNED : (uint32_t)result_node[i]), (unsigned long long)n * WK_WEIGHT_L);
}

// subject indices -> node ids (`src` may be `dst`: chunks appended to a
// single-job accumulation are rewritten in place)
__global__ void __launch_bounds__(256) words_to_features_kernel(const uint32_t* src, uint32_t* dst, uint32_t n,
                                                                const int32_t* __restrict__ subj_feat, uint32_t n_subjects,
                                                                uint32_t n_nodes, int* __restrict__ err) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t w = src[i], s = w & kWordSubjMask;
    uint32_t f = kFreeMissing;
    if (s < n_subjects) {
        const uint32_t x = (uint32_t)subj_feat[s];
        f = x < n_nodes ? x : kFreeMissing;
    } else if (w >> kWordSizeShift) {
        atomicOr(err, kErrFeatureRange);
    }
    dst[i] = (w & ~kWordSubjMask) | f;
}

}  // namespace wk
