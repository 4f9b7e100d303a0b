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
    unsigned long long* stat_reads;
    unsigned long long* stat_records;
    CountTable table;
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
__device__ __forceinline__ bool first_occurrence(const int32_t* __restrict__ subj, int32_t s,
                                                 int32_t j) {
    const int32_t c = subj[j];
    for (int32_t i = s; i < j; ++i)
        if (subj[i] == c) return false;
    return true;
}

template <bool kUseLds>
__device__ __forceinline__ void count_add(const LdsCache& cache, const CountTable& table,
                                          uint64_t key) {
    if constexpr (kUseLds)
        cached_add(cache, table, key, 1ull);
    else
        table_add(table, key, 1ull);
}

// One thread per read; every job (rank) is evaluated from the same pass over
// the read's records.
template <bool kUseLds>
__global__ void __launch_bounds__(256) classify_kernel(ClassifyArgs a, uint32_t lds_slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsCache cache{nullptr, nullptr, 0};
    if constexpr (kUseLds) {
        cache.keys = reinterpret_cast<unsigned long long*>(smem);
        cache.vals = cache.keys + lds_slots;
        cache.mask = lds_slots - 1;
        lds_cache_init(cache);
    }

    unsigned long long my_reads = 0, my_records = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_reads; r += stride) {
        const int32_t s = a.qoff[r];
        const int32_t e = a.qoff[r + 1];
        const int32_t n = e - s;
        if (n <= 0) {
            if (a.out_assign)
                for (int j = 0; j < a.n_jobs; ++j) a.out_assign[(int64_t)j * a.n_reads + r] = WK_ASSIGN_EMPTY;
            continue;
        }
        my_reads += 1;
        my_records += (unsigned long long)n;
        const int32_t g = a.group ? a.group[r] : 0;
        if (g >= (1 << WK_KEY_GROUP_BITS)) atomicOr(a.table.err, kErrGroupRange);

        // one pass over the subjects: extremes + membership in the hierarchy
        const int32_t first = a.subj[s];
        int32_t smin = first, smax = first;
        for (int32_t j = s + 1; j < e; ++j) {
            const int32_t c = a.subj[j];
            smin = c < smin ? c : smin;
            smax = c > smax ? c : smax;
        }
        if ((uint32_t)smax > (uint32_t)WK_MAX_FEATURE || smin < 0) atomicOr(a.table.err, kErrFeatureRange);
        const bool single = (smin == smax);

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
                            for (int32_t j = s; j < e; ++j) kd += first_occurrence(a.subj, s, j) ? 1 : 0;
                        }
                        if (kd > WK_MAX_K) {
                            atomicOr(a.table.err, kErrKRange);
                        } else {
                            for (int32_t j = s; j < e; ++j)
                                if (a.subj_is_set || first_occurrence(a.subj, s, j))
                                    count_add<kUseLds>(cache, a.table,
                                                       make_key(jb, kd, g, (uint32_t)a.subj[j]));
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
                for (int32_t j = s + 1; j < e; ++j) {
                    const int32_t c = a.subj[j];
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
                    for (int32_t j = s; j < e; ++j) {
                        if (!a.subj_is_set && !first_occurrence(a.subj, s, j)) continue;
                        total += 1;
                        const int32_t cj = a.subj[j];
                        const int32_t tj = (cj < a.n_nodes) ? anc[cj] : -1;
                        int32_t cnt = 0;
                        for (int32_t i = s; i < e; ++i) {
                            if (!a.subj_is_set && !first_occurrence(a.subj, s, i)) continue;
                            const int32_t ci = a.subj[i];
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
                        for (int32_t j = s; j < e; ++j) {
                            const int32_t c = a.subj[j];
                            if (c >= a.n_nodes || anc[c] < 0) continue;
                            if (a.subj_is_set || first_occurrence(a.subj, s, j)) kd += 1;
                        }
                        if (kd > WK_MAX_K) {
                            atomicOr(a.table.err, kErrKRange);
                        } else {
                            for (int32_t j = s; j < e; ++j) {
                                const int32_t c = a.subj[j];
                                if (c >= a.n_nodes) continue;
                                const int32_t t = anc[c];
                                if (t < 0) continue;
                                if (a.subj_is_set || first_occurrence(a.subj, s, j))
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

    // statistics: one atomic per wave
    my_reads = wave_sum(my_reads);
    my_records = wave_sum(my_records);
    if ((threadIdx.x & (kWave - 1)) == 0) {
        if (my_reads) atomicAdd(a.stat_reads, my_reads);
        if (my_records) atomicAdd(a.stat_records, my_records);
    }
    if constexpr (kUseLds) lds_cache_flush(cache, a.table);
}

}  // namespace wk
