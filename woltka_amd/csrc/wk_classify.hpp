// wk_classify.hpp — per-read assignment + count kernels.
//
// Reproduces, on packed integer ids, the reference's assigners and counters:
//   classify.assign_none  (woltka/classify.py:32-51)
//   classify.assign_free  (woltka/classify.py:54-78)   + tree.find_lca (tree.py:513-566)
//   classify.assign_rank  (woltka/classify.py:81-127)  + tree.find_rank (tree.py:467-510)
//   classify.majority     (woltka/classify.py:300-317)
//   classify.counter / counter_strat (woltka/classify.py:144-171, 216-249)
//   workflow.assign_readmap's Unassigned substitution (woltka/workflow.py:1038-1039)
#pragma once
#include "wk_device.hpp"

namespace wk {

struct Node {
    int32_t parent;  // DFS pre-order id of the parent (root: itself, id 0)
    int32_t last;    // largest pre-order id inside this node's subtree
};

struct JobDev {
    int32_t mode;
    uint32_t flags;
    const int32_t* anc;  // rank table (WK_MODE_RANK)
    double major;
};

struct ClassifyArgs {
    const int32_t* subj;   // [n_records]
    const int32_t* qoff;   // [n_reads + 1]
    const int32_t* group;  // [n_reads] or null
    int64_t n_reads;
    const Node* nodes;  // [n_nodes] or null
    int32_t n_nodes;
    int32_t n_jobs;
    int32_t subj_is_set;
    JobDev jobs[WK_MAX_JOBS];
    int32_t* out_assign;  // [n_jobs * n_reads] or null
    unsigned long long* stat_block;  // [2 * gridDim.x]: per-workgroup (reads, records) totals
    CountTable table;
    uint32_t ablate;  // measurement builds only (-DWK_ABLATE): 1 = drop counts, 2 = skip flush
};

// tree.find_rank for all nodes (tree.py:467-510): the taxon itself is tested
// first, the walk stops after the root has been tested.
__global__ void __launch_bounds__(256) rank_table_kernel(const Node* __restrict__ nodes,
                                                         const int32_t* __restrict__ rank_code,
                                                         int32_t n_nodes, int32_t code,
                                                         int32_t* __restrict__ anc) {
    int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_nodes) return;
    int32_t u = v;
    int32_t res = -1;
    for (;;) {
        if (rank_code[u] == code) {
            res = u;
            break;
        }
        int32_t p = nodes[u].parent;
        if (p == u) break;
        u = p;
    }
    anc[v] = res;
}

// Lowest common ancestor of a set of hierarchy nodes given only the smallest
// and largest pre-order id in the set: the LCA of a set equals the LCA of its
// pre-order extremes, and that is the lowest ancestor `a` of `lo` whose
// subtree interval [a, last[a]] still contains `hi`.  Equals tree.find_lca
// (tree.py:513-566) on a rooted tree.
__device__ __forceinline__ int32_t lca_of_range(const Node* __restrict__ nodes, int32_t lo,
                                                int32_t hi) {
    int32_t u = lo;
    Node nd = nodes[u];
    while (nd.last < hi) {
        u = nd.parent;
        nd = nodes[u];
    }
    return u;
}

// true iff no earlier record of the same read names the same subject
template <typename P>
__device__ __forceinline__ bool first_occurrence(P cand, int32_t j) {
    const int32_t c = cand[j];
    for (int32_t i = 0; i < j; ++i)
        if (cand[i] == c) return false;
    return true;
}

template <bool kUseLds>
__device__ __forceinline__ void count_add(const LdsCache& cache, const CountTable& table,
                                          uint64_t key) {
#ifdef WK_ABLATE
    if (cache.ablate & 1) {  // measurement only: drop the count, keep the key live
        asm volatile("" ::"v"((uint32_t)key), "v"((uint32_t)(key >> 32)));
        return;
    }
#endif
    if constexpr (kUseLds)
        cached_add(cache, table, key, 1ull);
    else
        table_add(table, key, 1ull);
}

// Everything the reference does for ONE read (query, mate) whose n >= 1
// candidate subjects are cand[0..n): all jobs (ranks) are evaluated from the
// same candidates.  `P` is a pointer into HBM or into an LDS tile.
template <bool kUseLds, typename P>
__device__ __forceinline__ void process_read(const ClassifyArgs& a, const LdsCache& cache, P cand,
                                             int32_t n, int64_t r, int32_t g, int32_t first) {
    // `first` == cand[0] (loaded ahead of time by the caller)
    // one pass over the subjects: extremes (= LCA inputs) and set size 1 test
    int32_t smin = first, smax = first;
    for (int32_t j = 1; j < n; ++j) {
        const int32_t c = cand[j];
        smin = c < smin ? c : smin;
        smax = c > smax ? c : smax;
    }
    if ((uint32_t)smax > (uint32_t)WK_MAX_FEATURE || smin < 0) atomicOr(a.table.err, kErrFeatureRange);
    const bool single = (smin == smax);
#ifdef WK_ABLATE
    if (a.ablate & 4) {  // measurement only: stop after the candidate scan
        asm volatile("" ::"v"(smin), "v"(smax));
        return;
    }
#endif

    for (int jb = 0; jb < a.n_jobs; ++jb) {
        const JobDev job = a.jobs[jb];
        int32_t res = WK_ASSIGN_NONE;  // feature id, NONE, or MULTI
        if (job.mode == WK_MODE_NONE) {
            // assign_none: sole subject, else None (uniq) or all subjects
            if (single) {
                res = first;
            } else if (!(job.flags & WK_F_UNIQ)) {
                res = WK_ASSIGN_MULTI;
                if (g >= 0) {
                    int32_t kd = n;
                    if (!a.subj_is_set) {
                        kd = 0;
                        for (int32_t j = 0; j < n; ++j) kd += first_occurrence(cand, j) ? 1 : 0;
                    }
                    if (kd > WK_MAX_K) {
                        atomicOr(a.table.err, kErrKRange);
                    } else {
                        for (int32_t j = 0; j < n; ++j)
                            if (a.subj_is_set || first_occurrence(cand, j))
                                count_add<kUseLds>(cache, a.table, make_key(jb, kd, g, (uint32_t)cand[j]));
                    }
                }
            }
        } else if (job.mode == WK_MODE_FREE) {
            // assign_free: one subject -> itself (subok) or its parent, no
            // root test; several -> LCA, None if it is the root or if any
            // subject is outside the hierarchy.
            if (single) {
                if (job.flags & WK_F_SUBOK)
                    res = first;
                else
                    res = (first < a.n_nodes) ? a.nodes[first].parent : WK_ASSIGN_NONE;
            } else if (smax < a.n_nodes) {
                const int32_t u = lca_of_range(a.nodes, smin, smax);
                res = (u == 0) ? WK_ASSIGN_NONE : u;
            }
        } else {
            // assign_rank: map every subject to its ancestor at the rank
            const int32_t* __restrict__ anc = job.anc;
            const int32_t t0 = (first < a.n_nodes) ? anc[first] : -1;
            int32_t tmin = t0, tmax = t0;
            bool all_same = true, any_none = (t0 < 0);
            for (int32_t j = 1; j < n; ++j) {
                const int32_t c = cand[j];
                const int32_t t = (c < a.n_nodes) ? anc[c] : -1;
                all_same &= (t == t0);
                any_none |= (t < 0);
                tmin = t < tmin ? t : tmin;
                tmax = t > tmax ? t : tmax;
            }
            if (all_same) {
                res = t0 < 0 ? WK_ASSIGN_NONE : t0;
            } else if (job.major > 0.0) {
                // majority rule over the distinct subjects; None is a
                // countable value (util.count_list).  Ties cannot reach a
                // threshold > 0.5, so the first maximum suffices.
                int32_t total = 0, best = -1, best_n = 0;
                for (int32_t j = 0; j < n; ++j) {
                    if (!a.subj_is_set && !first_occurrence(cand, j)) continue;
                    total += 1;
                    const int32_t cj = cand[j];
                    const int32_t tj = (cj < a.n_nodes) ? anc[cj] : -1;
                    int32_t cnt = 0;
                    for (int32_t i = 0; i < n; ++i) {
                        if (!a.subj_is_set && !first_occurrence(cand, i)) continue;
                        const int32_t ci = cand[i];
                        const int32_t ti = (ci < a.n_nodes) ? anc[ci] : -1;
                        cnt += (ti == tj) ? 1 : 0;
                    }
                    if (cnt > best_n) {
                        best_n = cnt;
                        best = tj;
                    }
                }
                res = ((double)best_n >= (double)total * job.major && best >= 0) ? best : WK_ASSIGN_NONE;
            } else if (job.flags & WK_F_ABOVE) {
                if (!any_none) {
                    const int32_t u = lca_of_range(a.nodes, tmin, tmax);
                    res = (u == 0) ? WK_ASSIGN_NONE : u;
                }
            } else if (!(job.flags & WK_F_UNIQ)) {
                // the list `taxa`: one entry per distinct subject, None
                // entries dropped before k is taken (classify.py:167-168)
                res = WK_ASSIGN_MULTI;
                if (g >= 0) {
                    int32_t kd = 0;
                    for (int32_t j = 0; j < n; ++j) {
                        const int32_t c = cand[j];
                        if (c >= a.n_nodes || anc[c] < 0) continue;
                        if (a.subj_is_set || first_occurrence(cand, j)) kd += 1;
                    }
                    if (kd > WK_MAX_K) {
                        atomicOr(a.table.err, kErrKRange);
                    } else {
                        for (int32_t j = 0; j < n; ++j) {
                            const int32_t c = cand[j];
                            if (c >= a.n_nodes) continue;
                            const int32_t t = anc[c];
                            if (t < 0) continue;
                            if (a.subj_is_set || first_occurrence(cand, j))
                                count_add<kUseLds>(cache, a.table, make_key(jb, kd, g, (uint32_t)t));
                        }
                    }
                }
            }
        }

        if (a.out_assign) a.out_assign[(int64_t)jb * a.n_reads + r] = res;
        if (g >= 0) {
            if (res >= 0)
                count_add<kUseLds>(cache, a.table, make_key(jb, 1, g, (uint32_t)res));
            else if (res == WK_ASSIGN_NONE && (job.flags & WK_F_UNASSIGNED))
                count_add<kUseLds>(cache, a.table, make_key(jb, 1, g, WK_FEATURE_UNASSIGNED));
        }
    }
}

__device__ __forceinline__ void mark_empty(const ClassifyArgs& a, int64_t r) {
    if (a.out_assign)
        for (int j = 0; j < a.n_jobs; ++j) a.out_assign[(int64_t)j * a.n_reads + r] = WK_ASSIGN_EMPTY;
}

// Statistics: one (reads, records) slot per workgroup, updated with plain
// loads/stores by one thread (launches are stream-ordered, so slot b is only
// ever touched by workgroup b of the launch in flight).  Device-scope atomics
// on a shared counter cost ~15 ns *each, serialised*: thousands of waves
// hitting two counters used to dominate the kernel.
__device__ __forceinline__ void flush_stats(const ClassifyArgs& a, unsigned long long my_reads,
                                            unsigned long long my_records) {
    __shared__ unsigned long long acc[2];
    if (threadIdx.x == 0) acc[0] = acc[1] = 0ull;
    __syncthreads();
    my_reads = wave_sum(my_reads);
    my_records = wave_sum(my_records);
    if ((threadIdx.x & (kWave - 1)) == 0) {
        atomicAdd(&acc[0], my_reads);  // LDS atomics, a handful per workgroup
        atomicAdd(&acc[1], my_records);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.stat_block[2 * blockIdx.x] += acc[0];
        a.stat_block[2 * blockIdx.x + 1] += acc[1];
    }
}

// Direct variant: one thread per read, candidates read straight from HBM.
// Kept as the simple baseline of the tiled kernel (A/B via the "tiled" option)
// and used when the LDS front cache is switched off.
template <bool kUseLds>
__global__ void __launch_bounds__(1024) classify_kernel(ClassifyArgs a, uint32_t lds_slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsCache cache{};
#ifdef WK_ABLATE
    cache.ablate = a.ablate;
#endif
    if constexpr (kUseLds) {
        cache.base = reinterpret_cast<unsigned long long*>(smem);
        cache.bmask = lds_slots / 4 - 1;
        lds_cache_init(cache);
    }
    unsigned long long my_reads = 0, my_records = 0;
    // Software pipeline over the thread's reads r, r+stride, ...: the offsets
    // of read i+2 and the first subject (and stratum) of read i+1 are in
    // flight while read i is evaluated, so the offset -> record chain is off
    // the critical path and only the table gathers of read i are exposed.
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t last = a.n_reads - 1;
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto load_offsets = [&](int64_t i, int32_t& s, int32_t& e) {
        const int64_t c = i < a.n_reads ? i : last;  // clamped: harmless re-read past the end
        s = a.qoff[c];
        e = a.qoff[c + 1];
    };
    auto load_group = [&](int64_t i) -> int32_t { return a.group ? a.group[i < a.n_reads ? i : last] : 0; };
    int32_t s0, e0, s1, e1;
    load_offsets(r, s0, e0);
    load_offsets(r + stride, s1, e1);
    int32_t f0 = (e0 > s0) ? a.subj[s0] : 0;
    int32_t g0 = load_group(r);
    for (; r < a.n_reads; r += stride) {
        int32_t s2, e2;
        load_offsets(r + 2 * stride, s2, e2);
        const int32_t f1 = (e1 > s1) ? a.subj[s1] : 0;
        const int32_t g1 = load_group(r + stride);
        const int32_t n = e0 - s0;
        if (n <= 0) {
            mark_empty(a, r);
        } else {
            my_reads += 1;
            my_records += (unsigned long long)n;
#ifdef WK_ABLATE
            if (!(a.ablate & 8))  // measurement only: offsets stream alone
#endif
            {
                if (g0 >= (1 << WK_KEY_GROUP_BITS)) atomicOr(a.table.err, kErrGroupRange);
                process_read<kUseLds>(a, cache, a.subj + s0, n, r, g0, f0);
            }
        }
        s0 = s1; e0 = e1; f0 = f1; g0 = g1;
        s1 = s2; e1 = e2;
    }
    flush_stats(a, my_reads, my_records);
    if constexpr (kUseLds) lds_cache_flush(cache, a.table);
}

// Tiled variant.  A workgroup walks over tiles of kTileReads consecutive reads.
// Per tile it loads the read offsets and then the tile's whole record range —
// one contiguous span of `subj` — into LDS with coalesced 16-byte loads, so
// HBM sees pure streaming traffic and no thread waits on a private
// offset -> record -> table chain.  Reads are then evaluated out of LDS.  A
// tile whose records do not fit the window (reads with very many hits) is
// evaluated straight from HBM instead.
constexpr int kTileThreads = 512;
constexpr int kTileReads = kTileThreads;      // one read per thread per tile
constexpr int kTileWindow = 16 * kTileReads;  // records staged per tile (32 KiB)

__global__ void __launch_bounds__(kTileThreads) classify_tiled_kernel(ClassifyArgs a, uint32_t lds_slots,
                                                                      int64_t n_records) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int32_t* lrec = reinterpret_cast<int32_t*>(smem);                      // [kTileWindow + 4]
    int32_t* loff = lrec + kTileWindow + 4;                                  // [kTileReads + 4]
    LdsCache cache{};
#ifdef WK_ABLATE
    cache.ablate = a.ablate;
#endif
    cache.base = reinterpret_cast<unsigned long long*>(loff + kTileReads + 4);
    cache.bmask = lds_slots / 4 - 1;
    lds_cache_init(cache);

    unsigned long long my_reads = 0, my_records = 0;
    const int64_t n_tiles = (a.n_reads + kTileReads - 1) / kTileReads;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t r0 = tile * kTileReads;
        const int32_t nr = (int32_t)((a.n_reads - r0) < kTileReads ? (a.n_reads - r0) : kTileReads);
        // read offsets of the tile: nr + 1 values
        for (int32_t i = threadIdx.x; i <= nr; i += kTileThreads) loff[i] = a.qoff[r0 + i];
        __syncthreads();
        const int32_t rec0 = loff[0];
        const int32_t rec1 = loff[nr];
        const int32_t base = rec0 & ~3;  // 16-byte aligned start of the span
        const bool staged = (rec1 - base) <= kTileWindow;
        if (staged) {
            // span [base, rec1) rounded up to int4; the buffer is padded
            const int32_t nvec = (rec1 - base + 3) >> 2;
            const int4* __restrict__ src = reinterpret_cast<const int4*>(a.subj + base);
            int4* dst = reinterpret_cast<int4*>(lrec);
            for (int32_t v = threadIdx.x; v < nvec; v += kTileThreads) {
                if ((int64_t)base + 4 * (int64_t)v + 4 <= n_records) {
                    dst[v] = src[v];
                } else {  // last, partial vector of the whole array
                    int32_t tmp[4] = {0, 0, 0, 0};
                    for (int q = 0; q < 4; ++q)
                        if ((int64_t)base + 4 * (int64_t)v + q < n_records) tmp[q] = a.subj[base + 4 * v + q];
                    dst[v] = make_int4(tmp[0], tmp[1], tmp[2], tmp[3]);
                }
            }
            __syncthreads();
        }
        if ((int32_t)threadIdx.x < nr) {
            const int64_t r = r0 + threadIdx.x;
            const int32_t s = loff[threadIdx.x];
            const int32_t n = loff[threadIdx.x + 1] - s;
            if (n <= 0) {
                mark_empty(a, r);
            } else {
                my_reads += 1;
                my_records += (unsigned long long)n;
                const int32_t g = a.group ? a.group[r] : 0;
                if (g >= (1 << WK_KEY_GROUP_BITS)) atomicOr(a.table.err, kErrGroupRange);
                if (staged)
                    process_read<true>(a, cache, lrec + (s - base), n, r, g, lrec[s - base]);
                else
                    process_read<true>(a, cache, a.subj + s, n, r, g, a.subj[s]);
            }
        }
        __syncthreads();  // the tile buffers are reused by the next tile
    }
    flush_stats(a, my_reads, my_records);
    lds_cache_flush(cache, a.table);
}

}  // namespace wk
