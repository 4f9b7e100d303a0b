// wk_classify.hpp — per-read assignment + count kernels.
//
// Reproduces, on packed integer ids, the reference's assigners and counters:
//   classify.assign_none  (woltka/classify.py:32-51)
//   classify.assign_free  (woltka/classify.py:54-78)   + tree.find_lca (tree.py:513-566)
//   classify.assign_rank  (woltka/classify.py:81-127)  + tree.find_rank (tree.py:467-510)
//   classify.majority     (woltka/classify.py:300-317)
//   classify.counter / counter_strat (woltka/classify.py:144-171, 216-249)
//   workflow.assign_readmap's Unassigned substitution (woltka/workflow.py:1038-1039)
//
// Kernels, in the order a chunk meets them (DESIGN_HISTORY.md §3.1):
//   count_subjects_kernel    pass 1 of the two-class split when the subject table
//                            fits the LDS bins: histogram of the subject indices of
//                            the reads with exactly one candidate + one "left for
//                            pass 2" bit per read
//   classify_single_kernel   pass 1 otherwise: hot subjects in bins + per-read
//                            evaluation of the others, or per-read evaluation alone
//   classify_kernel<., true, path> pass 2: merges pass 1's bins into the count table,
//                            compacts its share of the bits into a read list, walks
//                            it with the generic evaluator (process_read); one
//                            instantiation per kind of candidates (path)
//   classify_kernel<., false, path> the generic evaluator over all reads (split off /
//                            not applicable); classify_tiled_kernel: LDS-staged variant
//   partition_merge_kernel   aggregation of the partitioned miss log
//   dense_merge_kernel       column sums of dense-bin slab rows (unsplit dense mode)
//   rank_table_kernel, subject_rows_kernel   static tables (find_rank for all nodes;
//                            {feature, rank ancestors} row per subject)
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
    int32_t col;         // column of this job's rank ancestor in the subject rows
    int32_t _pad;
};

struct ClassifyArgs {
    const int32_t* subj;   // [n_records]
    const int32_t* qoff;   // [n_reads + 1]
    const int32_t* group;  // [n_reads] or null: every read belongs to group_base
    int32_t group_base;
    int64_t n_reads;
    const Node* nodes;  // [n_nodes] or null
    int32_t n_nodes;
    int32_t n_jobs;
    int32_t subj_is_set;
    // optional compact subject table: subj[] then holds dense subject indices and
    // rows[s * row_w] = {feature id, rank ancestor of job column 0, 1, ...}
    const int32_t* rows;
    int32_t row_w;
    int32_t n_subjects;
    JobDev jobs[WK_MAX_JOBS];
    int32_t* out_assign;  // [n_jobs * n_reads] or null
    unsigned long long* stat_block;  // [2 * gridDim.x]: per-workgroup (reads, records) totals
    CountTable table;
    // dense bins (see LdsCache): per-workgroup slab rows in HBM, merged by
    // dense_merge_kernel
    uint32_t dense_bins;
    uint32_t dense_total;  // bins per slab row: n_jobs * dense_bins, or the subject count (count-first pass)
    int32_t dense_by_subject;  // the bins are indexed by subject (first pass only)
    uint32_t* dense_slab;  // [gridDim.x][dense_total]
    int32_t slab16;        // slab rows hold 16-bit counts (a workgroup sees < 65536 reads): half the traffic
    // partitioned miss log (see LdsCache): [gridDim.x][log_parts][plog_cap] keys
    // and [gridDim.x][log_parts] stream lengths.  log_parts (a power of two) is
    // 256 when the distinct keys of a launch fit 256 LDS tables of the merge —
    // a workgroup's open 128-byte log lines (32 workgroups x 256 streams per
    // XCD = 1 MiB) then stay in L2 until they are full — else 1024
    unsigned long long* plog;
    uint32_t* plog_cnt;
    uint32_t plog_cap;
    uint32_t log_parts;
    // contribution log of size-normalised jobs (WK_F_SIZED): 4 x int32 per entry
    int32_t* log;
    unsigned long long* log_cursor;
    int64_t log_cap;
    uint32_t ablate;  // measurement builds only (-DWK_ABLATE): 1 = drop counts, 2 = skip flush
    // `--rank free` over subject rows without a walk up the tree: row column
    // free_col holds the subject's rank among the subjects of the tree (by
    // pre-order id), free_sparse[k * free_m + i] = the shallowest (= smallest id:
    // they are ancestors of one subject, hence comparable) of the LCAs of the
    // rank-adjacent subject pairs i .. i + 2^k - 1.  LCA(set) = LCA(smallest,
    // largest rank) = min of two entries.  Null: walk (lca_of_range).
    const int32_t* free_sparse;
    int32_t free_m;
    int32_t free_col;
    // second pass (classify_kernel after classify_single_kernel): merges the
    // first pass's slab (first_*), compacts its workgroup's share of left_mask
    // into read_list[blockIdx.x * list_seg ...] and walks that list
    const unsigned long long* left_mask;
    uint32_t n_mask_words;
    uint32_t list_seg;
    uint32_t* read_list;
    const uint32_t* first_slab;  // [first_rows][first_total] or null
    uint32_t first_rows, first_total;
    int32_t first_by_subject;  // slab columns are subject indices (else job * bins + feature)
    int32_t first_slab16;
    int32_t resume;            // continue the first pass's miss-log streams
    // chunks of feature ids (no subject rows, e.g. the gene lists of the
    // coord-match): the first pass builds a read's row from these per-rank
    // tables, column c of job.col == c
    const int32_t* col_anc[3];
    int32_t n_cols;
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

// Compact subject rows: rows[s * w] = feature id of subject s, followed by its
// ancestor at every rank column (one gather per rank table, done once per
// subject instead of once per alignment record).
struct RowCols {
    const int32_t* anc[WK_MAX_JOBS];
    int32_t n_cols;
    // column `by_subject` (or -1) is not looked up by the subject's feature in a
    // per-node table but copied from a per-subject array: the subject's rank
    // among the subjects of the tree (free_rank, see ClassifyArgs::free_sparse)
    int32_t by_subject;
    const int32_t* subject_col;
};
__global__ void __launch_bounds__(256) subject_rows_kernel(const int32_t* __restrict__ feature_of_subject,
                                                           int32_t n_subjects, int32_t n_nodes, RowCols cols,
                                                           int32_t w, int32_t* __restrict__ rows) {
    const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_subjects) return;
    const int32_t f = feature_of_subject[s];
    int32_t* row = rows + (int64_t)s * w;
    row[0] = f;
    for (int32_t c = 0; c < w - 1; ++c) {
        if (c == cols.by_subject)
            row[1 + c] = cols.subject_col[s];
        else
            row[1 + c] = (c < cols.n_cols && f >= 0 && f < n_nodes) ? cols.anc[c][f] : -1;
    }
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

// Candidate accessors.  A read's candidates are either feature ids (the
// rank ancestor is gathered from the per-rank table, one table per job), or
// dense subject indices into the compact subject rows (feature id and all rank
// ancestors of a subject sit in one row, so a record costs one short gather
// from a table that stays L2-resident: ~100 k subjects x 16 B instead of three
// 8 MB per-node tables at the config-3 shape).
template <typename P>
struct FeatureCand {
    static constexpr bool kHasCols = false;
    P cand;
    int32_t n_nodes;
    __device__ __forceinline__ int32_t id(int32_t j) const { return cand[j]; }
    __device__ __forceinline__ int32_t feat(int32_t j) const { return cand[j]; }
    __device__ __forceinline__ int32_t tax(int32_t j, const JobDev& job) const {
        const int32_t c = cand[j];
        return (c < n_nodes) ? job.anc[c] : -1;
    }
};
template <typename P>
struct RowCand {
    static constexpr bool kHasCols = false;
    P cand;
    const int32_t* __restrict__ rows;
    int32_t w;
    __device__ __forceinline__ int32_t id(int32_t j) const { return cand[j]; }
    __device__ __forceinline__ int32_t feat(int32_t j) const { return rows[(int64_t)cand[j] * w]; }
    __device__ __forceinline__ int32_t tax(int32_t j, const JobDev& job) const {
        return rows[(int64_t)cand[j] * w + 1 + job.col];
    }
};

// Per-rank-column summary of a read's candidates (inputs of assign_rank)
struct ColStats {
    int32_t t0, tmin, tmax;
    int32_t valid;  // records whose ancestor at the rank exists (= k of a list result when the read is a set)
    bool same, none;
};
__device__ __forceinline__ void col_init(ColStats& c, int32_t t) {
    c.t0 = c.tmin = c.tmax = t;
    c.valid = t >= 0 ? 1 : 0;
    c.same = true;
    c.none = t < 0;
}
__device__ __forceinline__ void col_update(ColStats& c, int32_t t) {
    c.same &= (t == c.t0);
    c.none |= (t < 0);
    c.valid += t >= 0 ? 1 : 0;
    c.tmin = t < c.tmin ? t : c.tmin;
    c.tmax = t > c.tmax ? t : c.tmax;
}

// One pass over a read's candidates: feature extremes (LCA inputs, "set has one
// element" test) and, for 4-wide subject rows, the statistics of all (<= 3)
// rank columns at once — every record is touched once for all ranks, and the
// row loads of up to four records are issued back to back.
struct ReadScan {
    int32_t smin, smax;
    ColStats col[3];
    bool bad;  // a subject index outside the subject table (checked while scanning)
};

template <typename C>
__device__ __forceinline__ void scan_candidates(const C& cand, int32_t n, int32_t first, ReadScan& sc) {
    sc.bad = false;
    sc.smin = sc.smax = first;
    for (int32_t j = 1; j < n; ++j) {
        const int32_t c = cand.feat(j);
        sc.smin = c < sc.smin ? c : sc.smin;
        sc.smax = c > sc.smax ? c : sc.smax;
    }
}

// rows of width 4: {feature, rank col 0, 1, 2}
template <typename P>
struct RowCand4 {
    static constexpr bool kHasCols = true;
    P cand;
    const int4* __restrict__ rows4;
    int4 row0;  // row of candidate 0 (loaded ahead of time)
    uint32_t n_subjects;
    __device__ __forceinline__ int32_t id(int32_t j) const { return cand[j]; }
    __device__ __forceinline__ int32_t feat(int32_t j) const { return rows4[cand[j]].x; }
    __device__ __forceinline__ int32_t tax(int32_t j, const JobDev& job) const {
        const int4 r = rows4[cand[j]];
        return job.col == 0 ? r.y : (job.col == 1 ? r.z : r.w);
    }
};

template <typename P>
__device__ __forceinline__ void scan_candidates(const RowCand4<P>& cand, int32_t n, int32_t, ReadScan& sc) {
    const int4 r0 = cand.row0;
    sc.bad = false;
    sc.smin = sc.smax = r0.x;
    col_init(sc.col[0], r0.y);
    col_init(sc.col[1], r0.z);
    col_init(sc.col[2], r0.w);
    for (int32_t j = 1; j < n; j += 4) {
        int32_t c[4];
        int4 rw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = cand.cand[(j + q < n) ? (j + q) : 0];  // pad with candidate 0 (idempotent)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // an index outside the table is reported, its row never fetched
            const bool in = (uint32_t)c[q] < cand.n_subjects;
            sc.bad |= !in;
            rw[q] = cand.rows4[in ? c[q] : 0];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sc.smin = rw[q].x < sc.smin ? rw[q].x : sc.smin;
            sc.smax = rw[q].x > sc.smax ? rw[q].x : sc.smax;
            col_update(sc.col[0], rw[q].y);
            col_update(sc.col[1], rw[q].z);
            col_update(sc.col[2], rw[q].w);
            if (j + q >= n) {  // padding repeats candidate 0: undo its `valid` contribution
                sc.col[0].valid -= rw[q].y >= 0 ? 1 : 0;
                sc.col[1].valid -= rw[q].z >= 0 ? 1 : 0;
                sc.col[2].valid -= rw[q].w >= 0 ? 1 : 0;
            }
        }
    }
}

template <typename C>
__device__ __forceinline__ ColStats rank_stats(const C& cand, const ReadScan& sc, const JobDev& job, int32_t n) {
    if constexpr (C::kHasCols) {
        // field-by-field value selects: selecting whole structs makes the
        // compiler index a stack copy (scratch memory)
        const int32_t k = job.col;
        auto pick = [k](auto x, auto y, auto z) { return k == 0 ? x : (k == 1 ? y : z); };
        ColStats cs;
        cs.t0 = pick(sc.col[0].t0, sc.col[1].t0, sc.col[2].t0);
        cs.tmin = pick(sc.col[0].tmin, sc.col[1].tmin, sc.col[2].tmin);
        cs.tmax = pick(sc.col[0].tmax, sc.col[1].tmax, sc.col[2].tmax);
        cs.valid = pick(sc.col[0].valid, sc.col[1].valid, sc.col[2].valid);
        cs.same = pick(sc.col[0].same, sc.col[1].same, sc.col[2].same);
        cs.none = pick(sc.col[0].none, sc.col[1].none, sc.col[2].none);
        return cs;
    } else {
        ColStats cs;
        col_init(cs, cand.tax(0, job));
        for (int32_t j = 1; j < n; ++j) col_update(cs, cand.tax(j, job));
        return cs;  // cs.valid counts records; equals the list's k when the read is a set
    }
}

// true iff no earlier record of the same read names the same subject
template <typename C>
__device__ __forceinline__ bool first_occurrence(const C& cand, int32_t j) {
    const int32_t c = cand.id(j);
    for (int32_t i = 0; i < j; ++i)
        if (cand.id(i) == c) return false;
    return true;
}

template <bool kUseLds>
__device__ __forceinline__ void count_add(const LdsCache& cache, const CountTable& table, int jb, uint32_t k,
                                          int32_t g, uint32_t feature) {
    // 1/k with k <= 16 is counted as L/k under k = 0 (see WK_WEIGHT_L)
    const bool weighted = k <= (uint32_t)WK_WEIGHT_MAX_K;
    const uint64_t key = make_key(jb, weighted ? 0u : k, g, feature);
    const unsigned long long w = weighted ? (unsigned long long)weight_of(k) : 1ull;
#ifdef WK_ABLATE
    if (cache.ablate & 1) {  // measurement only: drop the count, keep the key live
        asm volatile("" ::"v"((uint32_t)key), "v"((uint32_t)(key >> 32)));
        return;
    }
#endif
    if constexpr (kUseLds) {
        if (cache.dense && k == 1 && g == cache.dense_group && feature < cache.dense_bins) {
            atomicAdd(&cache.dense[(uint32_t)jb * cache.dense_bins + feature], 1u);
            return;
        }
        cached_add(cache, table, key, w);
    } else {
        table_add(table, key, w);
    }
}

// Size-normalised counting (classify.counter_size, classify.py:174-213) needs
// the (feature, subject) pair of every contribution: its value is
// sizes[subject] / divisor.  Such jobs append {feature, subject, job<<16 |
// divisor, group} to a log that the host folds exactly; one returning atomic
// per wave-level append (the active lanes share a ballot).
__device__ __forceinline__ void log_append(const ClassifyArgs& a, int32_t feature, int32_t subject, int jb,
                                           int32_t divisor, int32_t g) {
    const unsigned long long mask = __ballot(1);
    const int lane = threadIdx.x & (kWave - 1);
    const int leader = __ffsll((long long)mask) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(a.log_cursor, (unsigned long long)__popcll(mask));
    base = __shfl(base, leader, kWave);
    const unsigned long long pos = base + __popcll(mask & ((1ull << lane) - 1ull));
    if ((int64_t)pos < a.log_cap) {
        int4 e = make_int4(feature, subject, (jb << 16) | divisor, g);
        reinterpret_cast<int4*>(a.log)[pos] = e;
    }  // overflow is detected by the host from the cursor
}

// Everything the reference does for ONE read (query, mate) whose n >= 1
// candidate subjects are cand 0..n-1: all jobs (ranks) are evaluated from the
// same candidates.  `C` is a FeatureCand / RowCand over HBM or an LDS tile.
template <bool kUseLds, typename C>
__device__ __forceinline__ void process_read(const ClassifyArgs& a, const LdsCache& cache, const C& cand,
                                             int32_t n, int64_t r, int32_t g, int32_t first) {
    // `first` == cand.feat(0) (loaded ahead of time by the caller)
    // one pass over the subjects: extremes (= LCA inputs) and set size 1 test
    ReadScan sc;
    scan_candidates(cand, n, first, sc);
    if (sc.bad) {
        atomicOr(a.table.err, kErrFeatureRange);
        return;
    }
    const int32_t smin = sc.smin, smax = sc.smax;
    if ((uint32_t)smax > (uint32_t)WK_MAX_FEATURE || smin < 0) atomicOr(a.table.err, kErrFeatureRange);
    const bool single = (smin == smax);
#ifdef WK_ABLATE
    if (a.ablate & 4) {  // measurement only: stop after the candidate scan
        asm volatile("" ::"v"(smin), "v"(smax));
        return;
    }
#endif

    // Jobs whose result is a list over a *set* of subject rows are only marked
    // here and emitted together below: one more pass over the rows for all of
    // them instead of one pass per job.
    constexpr bool kFuseLists = kUseLds && C::kHasCols;
    uint32_t list_jobs = 0;

    for (int jb = 0; jb < a.n_jobs; ++jb) {
        const JobDev job = a.jobs[jb];
        int32_t res = WK_ASSIGN_NONE;  // feature id, NONE, or MULTI
        if (job.mode == WK_MODE_NONE) {
            // assign_none: sole subject, else None (uniq) or all subjects
            if (single) {
                res = first;
            } else if (!(job.flags & WK_F_UNIQ)) {
                res = WK_ASSIGN_MULTI;
                if (kFuseLists && g >= 0 && a.subj_is_set && !(job.flags & WK_F_SIZED) && n <= WK_MAX_K) {
                    list_jobs |= 1u << jb;
                } else if (g >= 0) {
                    int32_t kd = n;
                    if (!a.subj_is_set) {
                        kd = 0;
                        for (int32_t j = 0; j < n; ++j) kd += first_occurrence(cand, j) ? 1 : 0;
                    }
                    if (kd > WK_MAX_K) {
                        atomicOr(a.table.err, kErrKRange);
                    } else {
                        for (int32_t j = 0; j < n; ++j)
                            if (a.subj_is_set || first_occurrence(cand, j)) {
                                const int32_t f = cand.feat(j);
                                if (job.flags & WK_F_SIZED)
                                    log_append(a, f, f, jb, kd, g);
                                else
                                    count_add<kUseLds>(cache, a.table, jb, kd, g, (uint32_t)f);
                            }
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
                int32_t u;
                if (C::kHasCols && a.free_sparse) {
                    JobDev by_rank = job;
                    by_rank.col = a.free_col;
                    const ColStats cs = rank_stats(cand, sc, by_rank, n);
                    const uint32_t len = (uint32_t)(cs.tmax - cs.tmin);  // (> 0: the subjects differ)
                    const uint32_t lv = 31u - (uint32_t)__clz((int)len);
                    const int32_t x = a.free_sparse[(size_t)lv * a.free_m + cs.tmin];
                    const int32_t y = a.free_sparse[(size_t)lv * a.free_m + cs.tmax - (1u << lv)];
                    u = x < y ? x : y;
                } else {
                    u = lca_of_range(a.nodes, smin, smax);
                }
                res = (u == 0) ? WK_ASSIGN_NONE : u;
            }
        } else {
            // assign_rank: map every subject to its ancestor at the rank
            const ColStats cs = rank_stats(cand, sc, job, n);
            const int32_t t0 = cs.t0, tmin = cs.tmin, tmax = cs.tmax;
            const bool all_same = cs.same, any_none = cs.none;
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
                    const int32_t tj = cand.tax(j, job);
                    int32_t cnt = 0;
                    for (int32_t i = 0; i < n; ++i) {
                        if (!a.subj_is_set && !first_occurrence(cand, i)) continue;
                        cnt += (cand.tax(i, job) == tj) ? 1 : 0;
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
                if (kFuseLists && g >= 0 && a.subj_is_set && !(job.flags & WK_F_SIZED) && n <= WK_MAX_K) {
                    list_jobs |= 1u << jb;
                } else if (g >= 0) {
                    int32_t kd = cs.valid;  // exact when the read is a set
                    if (!a.subj_is_set) {
                        kd = 0;
                        for (int32_t j = 0; j < n; ++j) {
                            if (cand.tax(j, job) < 0) continue;
                            if (first_occurrence(cand, j)) kd += 1;
                        }
                    }
                    if (kd > WK_MAX_K) {
                        atomicOr(a.table.err, kErrKRange);
                    } else {
                        // candidates in groups of four: their table values are
                        // gathered back to back before any is counted
                        for (int32_t j0 = 0; j0 < n; j0 += 4) {
                            int32_t t4[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) t4[q] = (j0 + q < n) ? cand.tax(j0 + q, job) : -1;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int32_t t = t4[q];
                                if (t < 0) continue;
                                const int32_t j = j0 + q;
                                if (a.subj_is_set || first_occurrence(cand, j)) {
                                    if (job.flags & WK_F_SIZED)
                                        log_append(a, t, cand.feat(j), jb, kd, g);
                                    else
                                        count_add<kUseLds>(cache, a.table, jb, kd, g, (uint32_t)t);
                                }
                            }
                        }
                    }
                }
            }
        }

        if (a.out_assign) a.out_assign[(int64_t)jb * a.n_reads + r] = res;
        if (g >= 0) {
            int32_t f = -1;
            if (res >= 0)
                f = res;
            else if (res == WK_ASSIGN_NONE && (job.flags & WK_F_UNASSIGNED))
                f = WK_FEATURE_UNASSIGNED;
            if (f >= 0) {
                if (job.flags & WK_F_SIZED) {
                    // mean of the subjects' sizes: sum(sizes[x] for x in subs) / len(subs)
                    int32_t kd = n;
                    if (!a.subj_is_set) {
                        kd = 0;
                        for (int32_t j = 0; j < n; ++j) kd += first_occurrence(cand, j) ? 1 : 0;
                    }
                    if (kd > WK_MAX_K) {
                        atomicOr(a.table.err, kErrKRange);
                    } else {
                        for (int32_t j = 0; j < n; ++j)
                            if (a.subj_is_set || first_occurrence(cand, j)) log_append(a, f, cand.feat(j), jb, kd, g);
                    }
                } else {
                    count_add<kUseLds>(cache, a.table, jb, 1, g, (uint32_t)f);
                }
            }
        }
    }

    if constexpr (kFuseLists) {
        if (list_jobs) {
            // the rows once more, four at a time, for all marked jobs: each
            // distinct subject adds 1/k to its feature (rank none) or to its
            // ancestor at the job's rank, None entries dropped before k is taken
            for (int32_t j0 = 0; j0 < n; j0 += 4) {
                int4 rw[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) rw[q] = cand.rows4[cand.cand[(j0 + q < n) ? (j0 + q) : 0]];
                for (int jb = 0; jb < a.n_jobs; ++jb) {
                    if (!((list_jobs >> jb) & 1u)) continue;
                    const JobDev& job = a.jobs[jb];
                    const bool by_rank = job.mode == WK_MODE_RANK;
                    const int32_t col = job.col;
                    const int32_t kd =
                        !by_rank ? n : (col == 0 ? sc.col[0].valid : (col == 1 ? sc.col[1].valid : sc.col[2].valid));
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (j0 + q >= n) continue;
                        const int32_t t = !by_rank ? rw[q].x : (col == 0 ? rw[q].y : (col == 1 ? rw[q].z : rw[q].w));
                        if (t < 0) continue;
                        count_add<kUseLds>(cache, a.table, jb, (uint32_t)kd, g, (uint32_t)t);
                    }
                }
            }
        }
    }
}

// A read with a single candidate (the bulk of real inputs) needs none of the
// set logic: every assigner reduces to one table value.  `row` is the
// candidate's subject row {feature, rank col 0, 1, 2}.
// Assignment of a read with one candidate under one job (classify.py:216-331
// with a one-element subject set): the feature itself, its parent, or its
// ancestor at the job's rank.
__device__ __forceinline__ int32_t single_result(const ClassifyArgs& a, const JobDev& job, const int4 row) {
    const int32_t f = row.x;
    if (job.mode == WK_MODE_NONE) return f;
    if (job.mode == WK_MODE_FREE)
        return (job.flags & WK_F_SUBOK) ? f : ((f < a.n_nodes) ? a.nodes[f].parent : WK_ASSIGN_NONE);
    const int32_t t = job.col == 0 ? row.y : (job.col == 1 ? row.z : row.w);
    return t < 0 ? WK_ASSIGN_NONE : t;
}

template <bool kUseLds>
__device__ __forceinline__ void single_job(const ClassifyArgs& a, const LdsCache& cache, const JobDev& job, int jb,
                                           const int4 row, uint32_t r, int32_t g) {
    const int32_t res = single_result(a, job, row);
    if (a.out_assign) a.out_assign[(int64_t)jb * a.n_reads + r] = res;
    if (g < 0) return;
    int32_t out = res;
    if (res < 0) {
        if (!(job.flags & WK_F_UNASSIGNED)) return;
        out = WK_FEATURE_UNASSIGNED;
    }
    if (job.flags & WK_F_SIZED)
        log_append(a, out, row.x, jb, 1, g);
    else
        count_add<kUseLds>(cache, a.table, jb, 1, g, (uint32_t)out);
}

template <bool kUseLds>
__device__ __forceinline__ void process_single(const ClassifyArgs& a, const LdsCache& cache, const int4 row,
                                               uint32_t r, int32_t g) {
    if ((uint32_t)row.x > (uint32_t)WK_MAX_FEATURE) atomicOr(a.table.err, kErrFeatureRange);
    for (int jb = 0; jb < a.n_jobs; ++jb) single_job<kUseLds>(a, cache, a.jobs[jb], jb, row, r, g);
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

// LDS front cache of a classify workgroup: hash buckets, then either the
// miss-log cursors or the dense bins.  With `a.resume` (second pass of the
// two-class split) the cursors start from the stream lengths the first pass
// left.
__device__ __forceinline__ void cache_setup(LdsCache& cache, const ClassifyArgs& a, unsigned char* smem,
                                            uint32_t lds_slots) {
    cache.base = reinterpret_cast<unsigned long long*>(smem);
    cache.bmask = lds_slots / 4 - 1;
    if (a.plog) {
        cache.plog_cur = reinterpret_cast<uint32_t*>(smem + (size_t)lds_slots * 16);
        cache.plog = a.plog + (size_t)blockIdx.x * a.log_parts * a.plog_cap;
        cache.plog_cap = a.plog_cap;
        cache.plog_shift = (uint32_t)__clz((int)a.log_parts) + 1u;
        const uint32_t* cnt = a.plog_cnt + (size_t)blockIdx.x * a.log_parts;
        for (uint32_t i = threadIdx.x; i < a.log_parts; i += blockDim.x) cache.plog_cur[i] = a.resume ? cnt[i] : 0u;
    }
    if (a.dense_bins) {  // (behind the log cursors when both are in use: the hot-subject first pass)
        cache.dense = reinterpret_cast<uint32_t*>(smem + (size_t)lds_slots * 16 + (a.plog ? a.log_parts * 4 : 0));
        cache.dense_bins = a.dense_by_subject ? 0u : a.dense_bins;  // count_add only knows (job, feature) bins
        cache.dense_group = a.group_base;
        const uint32_t nb = a.dense_total;
        for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) cache.dense[i] = 0u;
    }
    lds_cache_init(cache);
}

__device__ __forceinline__ void cache_finish(const LdsCache& cache, const ClassifyArgs& a) {
    lds_cache_flush(cache, a.table);  // starts with a workgroup barrier
    if (cache.plog_cur) {
        uint32_t* cnt = a.plog_cnt + (size_t)blockIdx.x * a.log_parts;
        for (uint32_t i = threadIdx.x; i < a.log_parts; i += blockDim.x) {
            const uint32_t n = cache.plog_cur[i];
            cnt[i] = n < a.plog_cap ? n : a.plog_cap;
        }
    }
    if (cache.dense) {
        const uint32_t nb = a.dense_total;
        if (a.slab16) {
            // 16-bit counts, unit-major: the 64 columns of unit u from all
            // workgroups are adjacent ([u][workgroup][64]), so the merge reads
            // one contiguous 128 B x n_workgroups block per unit
            const uint32_t half = ((nb + kWave - 1u) / kWave) * (kWave / 2u);  // words, units padded to 64 columns
            for (uint32_t i = threadIdx.x; i < half; i += blockDim.x) {
                const uint32_t u = i >> 5, w = i & 31u;
                const uint32_t lo = 2u * i < nb ? cache.dense[2u * i] : 0u;
                const uint32_t hi = 2u * i + 1u < nb ? cache.dense[2u * i + 1u] : 0u;
                a.dense_slab[((size_t)u * gridDim.x + blockIdx.x) * 32u + w] = lo | (hi << 16);
            }
        } else {
            uint32_t* row = a.dense_slab + (size_t)blockIdx.x * nb;
            for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) row[i] = cache.dense[i];
        }
    }
}

// Second-pass prologue 1: the first pass's slab -> count table.  Column sums
// over the first pass's workgroups (a unit = 64 adjacent columns, its rows
// split over the waves); per-subject columns then go through every job's
// assigner once — the per-read loop of classify.assign_* collapsed to a
// per-subject one — and (job, feature) columns are keys already.  A chunk of
// the split holds < 2^30 reads, so 32-bit sums are exact.
__device__ __forceinline__ void merge_first_pass(const ClassifyArgs& a, uint32_t (*part)[kWave]) {
    if (!a.first_slab) return;
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const uint32_t units = (a.first_total + kWave - 1) / kWave;
    for (uint32_t u = blockIdx.x; u < units; u += gridDim.x) {
        const uint32_t i = u * kWave + lane;
        uint32_t sum = 0;
        if (a.first_slab16) {
            // the unit's block [first_rows][64] of 16-bit counts is contiguous:
            // 16-byte loads (8 columns of one row per lane, 8 rows per wave
            // load), then the lanes that hold the same columns are folded
            const uint4* blk = reinterpret_cast<const uint4*>(a.first_slab) + (size_t)u * a.first_rows * 8u;
            const uint32_t n16 = a.first_rows * 8u;  // 16-byte pieces in the block
            uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (uint32_t t = threadIdx.x; t < n16; t += blockDim.x) {
                const uint4 v = blk[t];
                acc[0] += v.x & 0xFFFFu; acc[1] += v.x >> 16;
                acc[2] += v.y & 0xFFFFu; acc[3] += v.y >> 16;
                acc[4] += v.z & 0xFFFFu; acc[5] += v.z >> 16;
                acc[6] += v.w & 0xFFFFu; acc[7] += v.w >> 16;
            }
            // piece t covers columns 8 * (t % 8) .. +7: lanes l, l ^ 8, l ^ 16, l ^ 32 agree
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[k] += __shfl_xor(acc[k], 8, kWave);
                acc[k] += __shfl_xor(acc[k], 16, kWave);
                acc[k] += __shfl_xor(acc[k], 32, kWave);
            }
            // lane l < 8 now holds the wave's sums of columns 8l .. 8l + 7;
            // hand column `lane` to lane `lane`
            const uint32_t src = lane >> 3;
            uint32_t mine = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t got = __shfl(acc[k], (int)src, kWave);
                if ((lane & 7u) == (uint32_t)k) mine = got;
            }
            sum = mine;
        } else if (i < a.first_total) {
#pragma unroll 4
            for (uint32_t row = wave; row < a.first_rows; row += n_waves) sum += a.first_slab[(size_t)row * a.first_total + i];
        }
        part[wave][lane] = sum;
        __syncthreads();
        if (wave == 0 && i < a.first_total) {
            for (uint32_t q = 1; q < n_waves; ++q) sum += part[q][lane];
#ifdef WK_ABLATE
            if (a.ablate & 128) sum = 0u;  // measurement only: column sums without the table adds
#endif
            if (sum != 0u) {
                if (a.first_by_subject) {
                    const int4 row = reinterpret_cast<const int4*>(a.rows)[i];
                    if ((uint32_t)row.x > (uint32_t)WK_MAX_FEATURE) atomicOr(a.table.err, kErrFeatureRange);
                    for (int jb = 0; jb < a.n_jobs; ++jb) {
                        const JobDev& job = a.jobs[jb];
                        const int32_t res = single_result(a, job, row);
                        int32_t out = res;
                        if (res < 0) {
                            if (!(job.flags & WK_F_UNASSIGNED)) continue;
                            out = WK_FEATURE_UNASSIGNED;
                        }
                        table_add(a.table, make_key((uint32_t)jb, 1u, (uint32_t)a.group_base, (uint32_t)out), sum);
                    }
                } else {
                    const uint32_t bins = a.first_total / (uint32_t)a.n_jobs;
                    table_add(a.table, make_key(i / bins, 1u, (uint32_t)a.group_base, i % bins), sum);
                }
            }
        }
        __syncthreads();
    }
}

// Second-pass prologue 2: the workgroup's share of left_mask (128-byte chunks
// of 16 words = 1024 reads, dealt round-robin so that clustered multi-hit
// reads spread over all workgroups) -> its segment of read_list.  Returns the
// number of reads listed.  No global atomics, no grid-wide step.
__device__ __forceinline__ uint32_t compact_left(const ClassifyArgs& a, uint32_t* scan_tot) {
    uint32_t* __restrict__ list = a.read_list + (size_t)blockIdx.x * a.list_seg;
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const uint32_t per_round = blockDim.x >> 4;  // chunks per round
    uint32_t running = 0;
    for (uint32_t c0 = 0; ((size_t)c0 * gridDim.x + blockIdx.x) * 16u < a.n_mask_words; c0 += per_round) {
        const uint32_t c = c0 + (threadIdx.x >> 4);
        const size_t w = (((size_t)c * gridDim.x + blockIdx.x) << 4) + (threadIdx.x & 15u);
        unsigned long long m = w < a.n_mask_words ? a.left_mask[w] : 0ull;
        const uint32_t cnt = (uint32_t)__popcll(m);
        uint32_t inc = cnt;  // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t v = __shfl_up(inc, off, kWave);
            if ((int)lane >= off) inc += v;
        }
        if (lane == kWave - 1) scan_tot[wave] = inc;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t q = 0; q < n_waves; ++q) {
            const uint32_t tq = scan_tot[q];
            before += q < wave ? tq : 0u;
            total += tq;
        }
        uint32_t pos = running + before + inc - cnt;
        while (m) {
            const uint32_t b = (uint32_t)__ffsll((long long)m) - 1u;
            list[pos++] = ((uint32_t)w << 6) + b;
            m &= m - 1ull;
        }
        running += total;
        __syncthreads();
    }
    return running;
}

// Direct variant: one thread per read, candidates read straight from HBM.
// Kept as the simple baseline of the tiled kernel (A/B via the "tiled" option)
// and used when the LDS front cache is switched off.
//
// kPath fixes the kind of candidates at compile time — 0: subject indices with
// 4-column rows, 1: subject indices with rows of another width, 2: feature ids —
// so that a launch carries one copy of the evaluator instead of three (80 KB of
// code against a 64 KB instruction cache); -1 decides at run time.
template <bool kUseLds, bool kListed = false, int kPath = -1>
__global__ void __launch_bounds__(1024) classify_kernel(ClassifyArgs a, uint32_t lds_slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool listed = kUseLds && kListed;
    int64_t n_items = a.n_reads;
    const uint32_t* __restrict__ my_list = nullptr;
    if constexpr (listed) {
        __shared__ uint32_t part[16][kWave];
        __shared__ uint32_t scan_tot[16];
#ifdef WK_ABLATE
        if (!(a.ablate & 32))
#endif
        merge_first_pass(a, part);
#ifdef WK_ABLATE
        if (a.ablate & 64) return;
#endif
        n_items = (int64_t)compact_left(a, scan_tot);  // ends with a workgroup barrier
        my_list = a.read_list + (size_t)blockIdx.x * a.list_seg;
        if (n_items == 0) {
            // nothing left over here; a pass that owns its log streams leaves
            // this workgroup's empty for the merge
            if (a.plog && !a.resume)
                for (uint32_t i = threadIdx.x; i < a.log_parts; i += blockDim.x)
                    a.plog_cnt[(size_t)blockIdx.x * a.log_parts + i] = 0u;
            return;
        }
    }
    LdsCache cache{};
#ifdef WK_ABLATE
    cache.ablate = a.ablate;
#endif
    if constexpr (kUseLds) cache_setup(cache, a, smem, lds_slots);
    unsigned long long my_reads = 0, my_records = 0;
    // Software pipeline over the thread's reads r, r+stride, ...: loads of the
    // following reads are in flight while read i is evaluated, so the
    // offset -> record (-> subject row) chain is off the critical path and only
    // the gathers of read i's further candidates are exposed.
    // (a listed pass walks its workgroup's own list: workgroup-local stride)
    const int64_t stride = listed ? (int64_t)blockDim.x : (int64_t)gridDim.x * blockDim.x;
    const int64_t last = a.n_reads - 1;
    int64_t r = listed ? (int64_t)threadIdx.x : (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto load_offsets = [&](int64_t i, int32_t& s, int32_t& e) {
        const int64_t c = i < a.n_reads ? i : last;  // clamped: harmless re-read past the end
        s = a.qoff[c];
        e = a.qoff[c + 1];
    };
    auto load_group = [&](int64_t i) -> int32_t { return a.group ? a.group[i < a.n_reads ? i : last] : a.group_base; };
    auto evaluate = [&](auto cand, int32_t n, int64_t rr, int32_t g, int32_t first) {
        my_reads += 1;
        my_records += (unsigned long long)n;
#ifdef WK_ABLATE
        if (a.ablate & 8) return;  // measurement only: offsets stream alone
#endif
        if (g >= (1 << WK_KEY_GROUP_BITS)) atomicOr(a.table.err, kErrGroupRange);
        process_read<kUseLds>(a, cache, cand, n, rr, g, first);
    };

    const bool rows4_path = kPath < 0 ? (a.rows != nullptr && a.row_w == 4) : kPath == 0;
    if (rows4_path) {
        // stages: offsets (i+3) -> first subject index (i+2) -> its row (i+1) -> evaluate (i)
        const int4* __restrict__ rows4 = reinterpret_cast<const int4*>(a.rows);
        auto load_first = [&](int32_t s, int32_t e) -> int32_t { return (e > s) ? a.subj[s] : 0; };
        auto load_row = [&](int32_t c) -> int4 {
            return ((uint32_t)c < (uint32_t)a.n_subjects) ? rows4[c] : make_int4(-1, -1, -1, -1);
        };
        // with a read list (second pass of the two-class split) one more
        // stage in front: list entry (i+4) -> offsets (i+3) -> ...
        // (list entries are 32-bit and live in registers only in the listed
        // variant; the direct variant derives the read index from i)
        auto entry = [&](int64_t i) -> uint32_t {
            if constexpr (listed) return my_list[i < n_items ? i : n_items - 1];
            else return 0u;
        };
        int64_t i = r;
        [[maybe_unused]] uint32_t q0 = entry(i), q1 = entry(i + stride), q2 = entry(i + 2 * stride),
                                  q3 = entry(i + 3 * stride);
#define WK_RID(k, q) (listed ? (int64_t)(q) : i + (k) * stride)
        int32_t s0, e0, s1, e1, s2, e2;
        load_offsets(WK_RID(0, q0), s0, e0);
        load_offsets(WK_RID(1, q1), s1, e1);
        load_offsets(WK_RID(2, q2), s2, e2);
        int32_t c0 = load_first(s0, e0), c1 = load_first(s1, e1);
        int4 row0 = load_row(c0);
        int32_t g0 = load_group(WK_RID(0, q0));
        const bool one_job = a.n_jobs == 1;
        const JobDev job0 = a.jobs[0];  // kept in registers for the common single-rank run
        for (; i < n_items; i += stride) {
            [[maybe_unused]] const uint32_t q4 = entry(i + 4 * stride);
            const int64_t r0 = WK_RID(0, q0);
            int32_t s3, e3;
            load_offsets(WK_RID(3, q3), s3, e3);
            const int32_t c2 = load_first(s2, e2);
            const int4 row1 = load_row(c1);
            const int32_t g1 = load_group(WK_RID(1, q1));
            const int32_t n = e0 - s0;
            if (n <= 0) {
                mark_empty(a, r0);
            } else if (n == 1 && (uint32_t)c0 < (uint32_t)a.n_subjects) {
                my_reads += 1;
                my_records += 1;
                if (g0 >= (1 << WK_KEY_GROUP_BITS)) atomicOr(a.table.err, kErrGroupRange);
#ifdef WK_ABLATE
                if (!(a.ablate & 8))
#endif
                {
                    if (one_job) {
                        if ((uint32_t)row0.x > (uint32_t)WK_MAX_FEATURE) atomicOr(a.table.err, kErrFeatureRange);
                        single_job<kUseLds>(a, cache, job0, 0, row0, r0, g0);
                    } else {
                        process_single<kUseLds>(a, cache, row0, r0, g0);
                    }
                }
            } else {
                // (subject indices outside the table are caught while the rows are
                // scanned; the first one here, its row was fetched ahead)
                if ((uint32_t)c0 >= (uint32_t)a.n_subjects)
                    atomicOr(a.table.err, kErrFeatureRange);
                else
                    evaluate(RowCand4<const int32_t*>{a.subj + s0, rows4, row0, (uint32_t)a.n_subjects}, n, r0, g0, row0.x);
            }
            s0 = s1; e0 = e1; s1 = s2; e1 = e2; s2 = s3; e2 = e3;
            c0 = c1; c1 = c2; row0 = row1; g0 = g1;
            if constexpr (listed) { q0 = q1; q1 = q2; q2 = q3; q3 = q4; }
        }
#undef WK_RID
    } else {
        const bool use_rows = kPath < 0 ? a.rows != nullptr : kPath == 1;
        // first candidate of a read -> its feature id (subject rows: one more gather)
        auto first_feature = [&](int32_t s, int32_t e) -> int32_t {
            if (e <= s) return 0;
            const int32_t c = a.subj[s];
            if (!use_rows) return c;
            return ((uint32_t)c < (uint32_t)a.n_subjects) ? a.rows[(int64_t)c * a.row_w] : -1;
        };
        // (second pass of the split: the workgroup's own list of reads)
        auto item = [&](int64_t i) -> int64_t {
            if constexpr (listed) return (int64_t)my_list[i < n_items ? i : n_items - 1];
            else return i;
        };
        int64_t i = r;
        int64_t ra = item(i), rb = item(i + stride);
        int32_t s0, e0, s1, e1;
        load_offsets(ra, s0, e0);
        load_offsets(rb, s1, e1);
        int32_t f0 = first_feature(s0, e0);
        int32_t g0 = load_group(ra);
        for (; i < n_items; i += stride) {
            const int64_t rc = item(i + 2 * stride);
            int32_t s2, e2;
            load_offsets(rc, s2, e2);
            const int32_t f1 = first_feature(s1, e1);
            const int32_t g1 = load_group(rb);
            const int32_t n = e0 - s0;
            if (n <= 0) {
                mark_empty(a, ra);
            } else if (use_rows) {
                bool ok = true;
                for (int32_t j = 0; j < n; ++j) ok &= ((uint32_t)a.subj[s0 + j] < (uint32_t)a.n_subjects);
                if (!ok)
                    atomicOr(a.table.err, kErrFeatureRange);
                else
                    evaluate(RowCand<const int32_t*>{a.subj + s0, a.rows, a.row_w}, n, ra, g0, f0);
            } else {
                evaluate(FeatureCand<const int32_t*>{a.subj + s0, a.n_nodes}, n, ra, g0, f0);
            }
            s0 = s1; e0 = e1; f0 = f1; g0 = g1;
            s1 = s2; e1 = e2;
            ra = rb; rb = rc;
        }
    }
    flush_stats(a, my_reads, my_records);
    if constexpr (kUseLds) cache_finish(cache, a);
}

// ---- two-class split ---------------------------------------------------------
// Real inputs are mostly reads with one candidate plus a tail of multi-hit
// reads.  Evaluating both classes in one loop leaves the lanes of the cheap
// class idle while one lane walks a candidate list, and the generic
// evaluator's register / scalar pressure slows the cheap class down as well.
// So the chunk is split by class:
//   1. classify_single_kernel: a small kernel that evaluates every read with
//      exactly one candidate (and marks empty reads), and emits one bit per
//      read — the ballot of "not handled here" — into left_mask;
//   2. classify_kernel as the second pass: every workgroup merges its share of
//      the first pass's bins into the count table, compacts its share of the
//      bits into a list of read indices and walks that list (every lane busy
//      with a multi-hit read).  Two launches per chunk, no grid-wide step.
// All index arithmetic of pass 1 is 32-bit; the host selects the split only
// when n_reads, n_records < 2^30 and n_subjects < 2^28.
//
// Count first, classify later (kBySubject): when the subject table is small
// (the usual genome-level databases: ~10-30 k subjects) the first pass does not
// even look at the subject rows — it histograms the *subject indices* of the
// single-candidate reads in LDS bins (one coalesced 4-byte load + one ds_add
// per read), and the second pass's prologue applies the assigners once per
// subject instead of once per read.  A random 16-byte row gather per read costs a full
// 128-byte L2->L1 line each, which bounds the per-read variant.
//
// kReads reads per thread and round, 64 apart inside the wave's window, so that
// every load is coalesced and kReads x more bytes are in flight per wave.
// kHot: the subject table is larger than the LDS bins; only the first
// `dense_bins` subject indices — indices follow first appearance, so with
// skewed abundances these are the bulk of the reads — are histogrammed, the
// others take the per-read path (row gather + count).
template <bool kBySubject, bool kOneJob, int kReads, bool kFeature = false, bool kHot = false>
__global__ void __launch_bounds__(1024) classify_single_kernel(ClassifyArgs a, uint32_t lds_slots,
                                                               unsigned long long* __restrict__ left_mask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsCache cache{};
#ifdef WK_ABLATE
    cache.ablate = a.ablate;
#endif
    cache_setup(cache, a, smem, lds_slots);

    const uint32_t n_reads = (uint32_t)a.n_reads, last = n_reads - 1u;
    // (feature-id chunks: the candidate is the feature itself, any id is "in the table")
    const uint32_t n_subjects = kFeature ? 0xFFFFFFFFu : (uint32_t)a.n_subjects;
    const uint32_t n_nodes = (uint32_t)a.n_nodes;
    const uint32_t tile = blockDim.x * (uint32_t)kReads;  // reads per workgroup round
    const uint32_t stride = gridDim.x * tile;
    const char* __restrict__ qoff_b = reinterpret_cast<const char*>(a.qoff);
    const char* __restrict__ subj_b = reinterpret_cast<const char*>(a.subj);
    const char* __restrict__ rows_b = reinterpret_cast<const char*>(a.rows);
    const char* __restrict__ group_b = reinterpret_cast<const char*>(a.group);
    const bool grouped = !kBySubject && a.group != nullptr;
    // read k of this thread in the round that starts at `base`: the wave's
    // window is kReads * 64 consecutive reads, lane-contiguous per k
    const uint32_t lane_off = (threadIdx.x >> 6) * (kWave * (uint32_t)kReads) + (threadIdx.x & (kWave - 1));

    struct Offs { int32_t s[kReads], e[kReads]; };
    struct Firsts { uint32_t c[kReads]; };
    struct Rows { int4 v[(kBySubject && !kHot) ? 1 : kReads]; };
    const uint32_t hot = kHot ? a.dense_total : 0xFFFFFFFFu;  // subject indices counted in the bins
    // byte offsets stay below 2^32: every access is base (SGPR pair) + 32-bit lane offset
    auto load_offsets = [&](uint32_t base, Offs& o) {
#pragma unroll
        for (int k = 0; k < kReads; ++k) {
            const uint32_t i = base + lane_off + (uint32_t)k * kWave;
            const uint32_t off = (i < last ? i : last) << 2;  // clamped: harmless re-read past the end
            o.s[k] = *reinterpret_cast<const int32_t*>(qoff_b + off);
            o.e[k] = *reinterpret_cast<const int32_t*>(qoff_b + off + 4u);
        }
    };
    auto load_firsts = [&](const Offs& o, Firsts& f) {
#pragma unroll
        for (int k = 0; k < kReads; ++k)
            f.c[k] = (o.e[k] > o.s[k]) ? *reinterpret_cast<const uint32_t*>(subj_b + ((uint32_t)o.s[k] << 2))
                                       : 0xFFFFFFFFu;
    };
    auto load_rows = [&](const Firsts& f, Rows& w) {
        if constexpr (kFeature) {
            // {feature, its ancestor at the rank of column 0, 1, 2}: one 4-byte
            // gather per rank column in use
#pragma unroll
            for (int k = 0; k < kReads; ++k) {
                const uint32_t c = f.c[k];
                const bool in = c < n_nodes;
                int4 v = make_int4((int32_t)c, -1, -1, -1);
                if (a.n_cols > 0 && in) v.y = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(a.col_anc[0]) + (c << 2));
                if (a.n_cols > 1 && in) v.z = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(a.col_anc[1]) + (c << 2));
                if (a.n_cols > 2 && in) v.w = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(a.col_anc[2]) + (c << 2));
                w.v[k] = v;
            }
        } else if constexpr (kHot) {
#pragma unroll
            for (int k = 0; k < kReads; ++k)  // only the cold subjects' rows are fetched
                w.v[k] = (f.c[k] >= hot && f.c[k] < n_subjects) ? *reinterpret_cast<const int4*>(rows_b + (f.c[k] << 4))
                                                                : make_int4(-1, -1, -1, -1);
        } else if constexpr (!kBySubject) {
#pragma unroll
            for (int k = 0; k < kReads; ++k)
                w.v[k] = (f.c[k] < n_subjects) ? *reinterpret_cast<const int4*>(rows_b + (f.c[k] << 4))
                                               : make_int4(-1, -1, -1, -1);
        }
    };

    uint32_t base = blockIdx.x * tile;
    uint32_t handled = 0, bad = 0;
    // stages: offsets (t+3) -> first subject indices (t+2) -> their rows (t+1) -> evaluate (t)
    Offs o0{}, o1{}, o2{}, o3{};
    Firsts f0{}, f1{}, f2{};
    Rows w0{}, w1{};
    load_offsets(base, o0);
    load_offsets(base + stride, o1);
    load_offsets(base + 2u * stride, o2);
    load_firsts(o0, f0);
    load_firsts(o1, f1);
    load_rows(f0, w0);
    const JobDev job0 = a.jobs[0];
    const int n_jobs = kOneJob ? 1 : a.n_jobs;
    for (; base < n_reads; base += stride) {
        load_offsets(base + 3u * stride, o3);
        load_firsts(o2, f2);
        load_rows(f1, w1);
        int32_t g[kReads];
#pragma unroll
        for (int k = 0; k < kReads; ++k) g[k] = a.group_base;
        if (grouped) {
#pragma unroll
            for (int k = 0; k < kReads; ++k) {
                const uint32_t i = base + lane_off + (uint32_t)k * kWave;
                g[k] = *reinterpret_cast<const int32_t*>(group_b + ((i < last ? i : last) << 2));
            }
        }
        bool mine[kReads];
#pragma unroll
        for (int k = 0; k < kReads; ++k) {
            const uint32_t r = base + lane_off + (uint32_t)k * kWave;
            const int32_t n = o0.e[k] - o0.s[k];
            const bool in = r < n_reads;
            mine[k] = in & (n == 1) & (f0.c[k] < n_subjects);
            if (in & (n <= 0)) mark_empty(a, r);
            handled += mine[k] ? 1u : 0u;
            // one bit per read: left for the second pass (several candidates, or
            // a subject index outside the table, which that pass reports)
            const unsigned long long left = __ballot(in & (n > 0) & !mine[k]);
            if ((threadIdx.x & (kWave - 1)) == 0 && r < n_reads) left_mask[r >> 6] = left;
            if constexpr (kBySubject) {
#ifdef WK_ABLATE
                if (a.ablate & 9) continue;
#endif
                if (mine[k] && f0.c[k] < hot) atomicAdd(&cache.dense[f0.c[k]], 1u);
                if constexpr (kHot) {
                    mine[k] = mine[k] && f0.c[k] >= hot;  // what is left for the per-read path below
                    bad |= (mine[k] && (uint32_t)w0.v[k].x > (uint32_t)WK_MAX_FEATURE) ? 1u : 0u;
                }
            } else {
                bad |= (mine[k] && (uint32_t)w0.v[k].x > (uint32_t)WK_MAX_FEATURE) ? 1u : 0u;
                bad |= (mine[k] && g[k] >= (1 << WK_KEY_GROUP_BITS)) ? 2u : 0u;
            }
        }
        if constexpr (!kBySubject || kHot) {
#ifdef WK_ABLATE
            if (!(a.ablate & 8))
#endif
            for (int jb = 0; jb < n_jobs; ++jb) {
                const JobDev& job = kOneJob ? job0 : a.jobs[jb];
#pragma unroll
                for (int k = 0; k < kReads; ++k)
                    if (mine[k])
                        single_job<true>(a, cache, job, jb, w0.v[k], base + lane_off + (uint32_t)k * kWave, g[k]);
            }
        }
        o0 = o1; o1 = o2; o2 = o3;
        f0 = f1; f1 = f2;
        w0 = w1;
    }
    if (bad & 1u) atomicOr(a.table.err, kErrFeatureRange);
    if (bad & 2u) atomicOr(a.table.err, kErrGroupRange);
    flush_stats(a, handled, handled);
    cache_finish(cache, a);
}

// Count first, classify later, as its own kernel (what classify_single_kernel<
// true, ., 4> computes when the bins cover the whole subject table): the three
// load stages are kept in a ring of kRing register sets that is addressed by
// *code position* — the loop body is unrolled kRing times — instead of being
// rotated with register copies.  A copy of a register that a load has not
// filled yet makes the compiler wait for every load in flight at the loop's
// back edge (s_waitcnt vmcnt(0)), which leaves one round of loads per wave in
// flight; here the offsets of three rounds and the subject indices of two are
// outstanding while a round is counted.
__global__ void __launch_bounds__(1024) count_subjects_kernel(ClassifyArgs a, uint32_t lds_slots,
                                                              unsigned long long* __restrict__ left_mask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsCache cache{};
#ifdef WK_ABLATE
    cache.ablate = a.ablate;
#endif
    cache_setup(cache, a, smem, lds_slots);
    constexpr int kReads = 4;  // reads per thread and round, 64 apart in the wave's window
    constexpr int kRing = 5;   // rounds in flight: offsets t+4 .. t+2, subject indices t+2 .. t+1, count t
    constexpr int kAhead = 2;  // rounds between a round's subject-index loads and its counting
    const uint32_t n_reads = (uint32_t)a.n_reads, last = n_reads - 1u;
    const uint32_t n_subjects = (uint32_t)a.n_subjects;
    const uint32_t tile = blockDim.x * (uint32_t)kReads;
    const uint32_t stride = gridDim.x * tile;
    const char* __restrict__ qoff_b = reinterpret_cast<const char*>(a.qoff);
    const char* __restrict__ subj_b = reinterpret_cast<const char*>(a.subj);
    const uint32_t lane_off = (threadIdx.x >> 6) * (kWave * (uint32_t)kReads) + (threadIdx.x & (kWave - 1));
    struct Round {
        int32_t s[kReads], e[kReads];
        uint32_t c[kReads];
    };
    Round ring[kRing];
    auto load_offsets = [&](uint32_t base, Round& x) {
#pragma unroll
        for (int k = 0; k < kReads; ++k) {
            const uint32_t i = base + lane_off + (uint32_t)k * kWave;
            const uint32_t off = (i < last ? i : last) << 2;  // clamped: harmless re-read past the end
            x.s[k] = *reinterpret_cast<const int32_t*>(qoff_b + off);
            x.e[k] = *reinterpret_cast<const int32_t*>(qoff_b + off + 4u);
        }
    };
    auto load_firsts = [&](Round& x) {
#pragma unroll
        for (int k = 0; k < kReads; ++k)
            x.c[k] = (x.e[k] > x.s[k]) ? *reinterpret_cast<const uint32_t*>(subj_b + ((uint32_t)x.s[k] << 2))
                                       : 0xFFFFFFFFu;
    };
    uint32_t handled = 0;
    auto count_round = [&](uint32_t base, const Round& x) {
#pragma unroll
        for (int k = 0; k < kReads; ++k) {
            const uint32_t r = base + lane_off + (uint32_t)k * kWave;
            const int32_t n = x.e[k] - x.s[k];
            const bool in = r < n_reads;
            const bool mine = in & (n == 1) & (x.c[k] < n_subjects);
            handled += mine ? 1u : 0u;
            // one bit per read: left for the second pass (several candidates, or
            // a subject index outside the table, which that pass reports)
            const unsigned long long left = __ballot(in & (n > 0) & !mine);
            if ((threadIdx.x & (kWave - 1)) == 0 && r < n_reads) left_mask[r >> 6] = left;
#ifdef WK_ABLATE
            if (a.ablate & 9) continue;
#endif
            if (mine) atomicAdd(&cache.dense[x.c[k]], 1u);
        }
    };
    uint32_t base = blockIdx.x * tile;
    // prologue: offsets of rounds 0 .. kRing-2, subject indices of rounds 0 .. kAhead-1
#pragma unroll
    for (int u = 0; u < kRing - 1; ++u) load_offsets(base + (uint32_t)u * stride, ring[u]);
#pragma unroll
    for (int u = 0; u < kAhead; ++u) load_firsts(ring[u]);
    bool more = base < n_reads;
    while (more) {
#pragma unroll
        for (int u = 0; u < kRing; ++u) {  // round t lives in ring[t % kRing]
            if (more) {
                load_offsets(base + (uint32_t)(kRing - 1) * stride, ring[(u + kRing - 1) % kRing]);
                load_firsts(ring[(u + kAhead) % kRing]);
                count_round(base, ring[u]);
                base += stride;
                more = base < n_reads;
            }
        }
    }
    flush_stats(a, handled, handled);
    cache_finish(cache, a);
}

// Column sums of the workgroups' dense-bin slab rows -> count table.  One key
// per non-empty bin, so no two adds ever meet on a slot.  A workgroup owns 64
// adjacent bins (one 256-byte segment per slab row) and splits the rows over
// 16 waves; partial sums meet in LDS.
__global__ void __launch_bounds__(1024) dense_merge_kernel(const uint32_t* __restrict__ slab, uint32_t n_rows,
                                                           uint32_t n_jobs, uint32_t bins, uint32_t group,
                                                           CountTable table) {
    __shared__ unsigned long long part[16][64];
    const uint32_t nb = n_jobs * bins;
    const uint32_t lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const uint32_t i = blockIdx.x * 64 + lane;
    unsigned long long sum = 0;
    if (i < nb) {
#pragma unroll 4
        for (uint32_t r = grp; r < n_rows; r += 16) sum += slab[(size_t)r * nb + i];
    }
    part[grp][lane] = sum;
    __syncthreads();
    if (grp == 0 && i < nb) {
        unsigned long long tot = 0;
#pragma unroll
        for (int g = 0; g < 16; ++g) tot += part[g][lane];
        if (tot) table_add(table, make_key(i / bins, 1, group, i % bins), tot);
    }
}

// Aggregation of the partitioned miss log: workgroup p gathers partition p's
// streams of all classify workgroups (short contiguous runs of keys), counts
// them in an LDS hash table — a partition holds 1/gridDim.x of the distinct keys —
// and adds every distinct key to the count table once.  Partitions are disjoint
// in key space, so those adds never contend.
__global__ void __launch_bounds__(1024) partition_merge_kernel(const unsigned long long* __restrict__ plog,
                                                               const uint32_t* __restrict__ plog_cnt,
                                                               uint32_t n_rows, uint32_t plog_cap,
                                                               uint32_t lds_slots, CountTable table) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t part = blockIdx.x, n_parts = gridDim.x;
    // an empty partition (a second pass with little left over) costs one look
    // at its stream lengths, not an LDS table
    uint32_t any = 0;
    for (uint32_t row = threadIdx.x; row < n_rows; row += blockDim.x) any |= plog_cnt[(size_t)row * n_parts + part];
    if (!__syncthreads_or((int)(any != 0u))) return;
    LdsCache cache{};
    cache.base = reinterpret_cast<unsigned long long*>(smem);
    cache.bmask = lds_slots / 4 - 1;
    lds_cache_init(cache);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n_waves = blockDim.x >> 6;
    // one stream per wave at a time; streams are short, so the next one's length is
    // fetched while this one is counted and its entries are loaded four to a lane
    // before the first is used (kEmptyKey is no entry: a lane past the end)
    uint32_t n = wave < n_rows ? plog_cnt[(size_t)wave * n_parts + part] : 0u;
    for (uint32_t row = wave; row < n_rows; row += n_waves) {
        const uint32_t next = row + n_waves;
        const uint32_t n_next = next < n_rows ? plog_cnt[(size_t)next * n_parts + part] : 0u;
        const unsigned long long* src = plog + ((size_t)row * n_parts + part) * plog_cap;
        for (uint32_t i = lane; i < n; i += 256) {
            unsigned long long ev[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) ev[u] = i + 64 * u < n ? src[i + 64 * u] : kEmptyKey;
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                if (ev[u] == kEmptyKey) continue;
                // an entry with k in [1, 16] is one contribution to the weighted key
                const unsigned long long e = ev[u];
                const uint32_t k = (uint32_t)(e >> 49) & (uint32_t)WK_MAX_K;
                const bool weighted = k >= 1u && k <= (uint32_t)WK_WEIGHT_MAX_K;
                cached_add(cache, table, weighted ? (e & ~kKeyKMask) : e,
                           weighted ? (unsigned long long)weight_of(k) : 1ull);
            }
        }
        n = n_next;
    }
    lds_cache_flush(cache, table);
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
                const int32_t g = a.group ? a.group[r] : a.group_base;
                if (g >= (1 << WK_KEY_GROUP_BITS)) atomicOr(a.table.err, kErrGroupRange);
                if (staged)
                    process_read<true>(a, cache, FeatureCand<const int32_t*>{lrec + (s - base), a.n_nodes}, n, r, g, lrec[s - base]);
                else
                    process_read<true>(a, cache, FeatureCand<const int32_t*>{a.subj + s, a.n_nodes}, n, r, g, a.subj[s]);
            }
        }
        __syncthreads();  // the tile buffers are reused by the next tile
    }
    flush_stats(a, my_reads, my_records);
    lds_cache_flush(cache, a.table);
}

}  // namespace wk
