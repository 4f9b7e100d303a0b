// wk_weigh.hpp — classify as a weighted histogram over subject indices.
//
// For the plain assigners — `--rank none` without --uniq (classify.assign_none,
// woltka/classify.py:32-51) and `--rank <rank>` without --uniq / --major /
// --above (classify.assign_rank, classify.py:81-127) — followed by
// classify.counter (classify.py:144-171), what a read adds to the profile does
// not depend on the rank at all as long as every candidate has an ancestor at
// every requested rank:
//
//   * candidates with different taxa: each of the n candidates adds 1/n to its
//     own taxon (classify.py:167-170; no None entry is dropped, so k = n);
//   * candidates that share one taxon: the read adds 1 to it (classify.py:
//     115-116) — which is n times 1/n on the same taxon.
//
// So a read of n <= 16 distinct candidate subjects adds L/n (L = 720720 =
// lcm(1..16), an exact integer) to the weight W[s] of each of its subjects,
// once for all ranks, and the profile at rank j is  sum over subjects s of
// W[s] on taxon_j(s).  The per-read loop over ranks, the row gathers and the
// hash-table probes of the generic evaluator collapse to one LDS add per
// alignment record; the assigners run once per *subject* afterwards
// (weigh_merge_kernel).
//
// Reads this rule does not cover — more than 16 candidates (their 1/k is not
// a multiple of 1/L), a candidate outside the subject table, or a candidate
// without an ancestor at one of the ranks (None entries change k,
// classify.py:167-168) — get a bit in left_mask and are evaluated by the
// generic second pass (classify_kernel<., true, .>) like before.
//
// Layout.  W lives in LDS as 32-bit bins; a workgroup owns one *slice* of
// kBins consecutive subject indices and one share of the reads.  With S slices
// the S workgroups of a *team* walk the same tiles of reads, each adding only
// the records that fall into its slice.  Workgroup b runs on XCD b mod 8, so a
// team is made of workgroups with equal b mod 8: the team's later readers find
// the tile in their XCD's L2 instead of HBM.  A 32-bit bin wraps after
// 2^32 / L ~ 5959 full-weight reads; the add returns the old value, the one
// add that observes the wrap bumps hi[s] in HBM (rare: at most total weight /
// 2^32 times per workgroup).
#pragma once
#include "wk_classify.hpp"
#include "wk_device.hpp"

namespace wk {

constexpr uint32_t kWeighThreads = 1024;
constexpr uint32_t kWeighMaxLds = 147456;  // 144 KiB of dynamic LDS: bins (+ validity bits)

struct WeighArgs {
    const int32_t* subj;   // [n_records] subject indices, every read a set
    const int32_t* qoff;   // [n_reads + 1]
    uint32_t n_reads;
    uint32_t n_subjects;
    uint32_t bins;         // subject indices per slice
    uint32_t n_slices;
    uint32_t teams_per_xcd;
    uint32_t n_xcd;
    // one bit per subject: no ancestor at one of the requested ranks, or a
    // feature id outside the key range; null = no subject has the bit set
    const uint32_t* invalid;
    uint32_t invalid_words;
    uint32_t* slab;   // [n_slices][n_teams][bins]
    uint32_t* hi;     // [n_subjects] wraps of the 32-bit bins (zero between launches)
    unsigned long long* left_mask;   // [ceil(n_reads / 64)]
    unsigned long long* stat_block;  // [2 * gridDim.x]
};

// 16 bytes of subject indices at a 4-byte aligned address
struct __attribute__((packed, aligned(4))) Rec4 {
    uint32_t x, y, z, w;
};

template <bool kAllValid>
__global__ void __launch_bounds__(kWeighThreads) weigh_subjects_kernel(WeighArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* bins = reinterpret_cast<uint32_t*>(smem);
    [[maybe_unused]] uint32_t* inval = bins + a.bins;

    const uint32_t xcd = blockIdx.x % a.n_xcd, m = blockIdx.x / a.n_xcd;
    if (m >= a.teams_per_xcd * a.n_slices) return;  // no whole team left on this XCD
    const uint32_t slice = m % a.n_slices;
    const uint32_t team = xcd * a.teams_per_xcd + m / a.n_slices;
    const uint32_t n_teams = a.n_xcd * a.teams_per_xcd;
    const uint32_t lo = slice * a.bins;
    const uint32_t span = min(a.bins, a.n_subjects - min(lo, a.n_subjects));

    for (uint32_t i = threadIdx.x; i < a.bins; i += kWeighThreads) bins[i] = 0u;
    if constexpr (!kAllValid)
        for (uint32_t i = threadIdx.x; i < a.invalid_words; i += kWeighThreads) inval[i] = a.invalid[i];
    __syncthreads();

    const uint32_t n_reads = a.n_reads, last = n_reads - 1u;
    const uint32_t n_tiles = (n_reads + kWeighThreads - 1u) / kWeighThreads;
    const char* __restrict__ qoff_b = reinterpret_cast<const char*>(a.qoff);
    const char* __restrict__ subj_b = reinterpret_cast<const char*>(a.subj);

    struct Stage {
        uint32_t s, e;
        Rec4 v[4];
    };
    constexpr int kRing = 3;  // offsets of tile t+2, records of t+1, adds of t
    Stage ring[kRing];
    auto load_offsets = [&](uint32_t tile, Stage& x) {
        const uint32_t r = tile * kWeighThreads + threadIdx.x;
        const uint32_t off = (r < last ? r : last) << 2;  // clamped: harmless re-read past the end
        x.s = *reinterpret_cast<const uint32_t*>(qoff_b + off);
        x.e = *reinterpret_cast<const uint32_t*>(qoff_b + off + 4u);
    };
    auto load_records = [&](Stage& x) {
        const uint32_t n = x.e - x.s;
        // (the staging buffer is padded, so the 16-byte loads may run past the last record)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (n > 4u * q) x.v[q] = *reinterpret_cast<const Rec4*>(subj_b + ((x.s + 4u * q) << 2));
    };
    uint32_t my_reads = 0, my_records = 0;
    auto add_tile = [&](uint32_t tile, const Stage& x) {
        const uint32_t r = tile * kWeighThreads + threadIdx.x;
        const uint32_t n = x.e - x.s;
        const bool in = r < n_reads;
        uint32_t c[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c[4 * q] = x.v[q].x;
            c[4 * q + 1] = x.v[q].y;
            c[4 * q + 2] = x.v[q].z;
            c[4 * q + 3] = x.v[q].w;
        }
        bool flagged = n > (uint32_t)WK_WEIGHT_MAX_K;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if ((uint32_t)j < n) {
                bool bad = c[j] >= a.n_subjects;
                if constexpr (!kAllValid)
                    if (!bad) bad = (inval[c[j] >> 5] >> (c[j] & 31u)) & 1u;
                flagged |= bad;
            }
        }
        const bool left = in & (n > 0u) & flagged;
        if (slice == 0u) {
            const unsigned long long mask = __ballot(left);
            if ((threadIdx.x & (kWave - 1)) == 0 && in) a.left_mask[r >> 6] = mask;
        }
        if (!in | flagged | (n == 0u)) return;
        my_reads += 1u;
        my_records += n;
        const uint32_t w = weight_of(n);
        // all adds first, their returned values checked afterwards: one wait
        // for the whole read instead of one LDS round trip per record
        uint32_t old[16];
        bool mine[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t idx = c[j] - lo;
            mine[j] = ((uint32_t)j < n) & (idx < span);
            old[j] = 0u;
            if (mine[j]) old[j] = atomicAdd(&bins[idx], w);
        }
        // (the empty asm pins the returned values behind all sixteen adds;
        // hipcc otherwise folds each wrap test into its add's branch and
        // waits for every LDS round trip in turn)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(old[j]));
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (mine[j] && old[j] + w < old[j]) atomicAdd(&a.hi[c[j]], 1u);  // the bin wrapped: + 2^32
    };

    uint32_t tile = team;
    load_offsets(tile, ring[0]);
    load_offsets(tile + n_teams, ring[1]);
    load_records(ring[0]);
    bool more = tile < n_tiles;
    while (more) {
#pragma unroll
        for (int u = 0; u < kRing; ++u) {  // tile t lives in ring[t % kRing]: stages addressed by code position
            if (more) {
                load_offsets(tile + 2u * n_teams, ring[(u + 2) % kRing]);
                load_records(ring[(u + 1) % kRing]);
                add_tile(tile, ring[u]);
                tile += n_teams;
                more = tile < n_tiles;
            }
        }
    }

    // statistics: the reads are counted once, by the team's first slice
    {
        __shared__ unsigned long long acc[2];
        if (threadIdx.x == 0) acc[0] = acc[1] = 0ull;
        __syncthreads();
        unsigned long long rd = wave_sum(slice == 0u ? my_reads : 0u);
        unsigned long long rc = wave_sum(slice == 0u ? my_records : 0u);
        if ((threadIdx.x & (kWave - 1)) == 0) {
            atomicAdd(&acc[0], rd);
            atomicAdd(&acc[1], rc);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            a.stat_block[2 * blockIdx.x] += acc[0];
            a.stat_block[2 * blockIdx.x + 1] += acc[1];
        }
    }
    uint32_t* row = a.slab + ((size_t)slice * n_teams + team) * a.bins;
    for (uint32_t i = threadIdx.x; i < a.bins; i += kWeighThreads) row[i] = bins[i];
}

// W[s] of every subject (column sums over the teams of its slice + the wraps)
// -> for every job the key (job, k = 0, group, taxon of s) += W[s]: the
// assigners of classify.py applied once per subject.  The adds go through an
// LDS cache (subjects of one phylum meet on one key).
struct WeighMergeArgs {
    const uint32_t* slab;
    uint32_t* hi;
    uint32_t n_subjects, bins, n_teams;
    const int32_t* rows;  // [n_subjects][row_w] = {feature, ancestor at rank column 0, 1, ...}
    int32_t row_w;
    int32_t n_jobs;
    int32_t mode[WK_MAX_JOBS];
    int32_t col[WK_MAX_JOBS];
    uint32_t group;
    CountTable table;
};

__global__ void __launch_bounds__(1024) weigh_merge_kernel(WeighMergeArgs a, uint32_t lds_slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsCache cache{};
    cache.base = reinterpret_cast<unsigned long long*>(smem);
    cache.bmask = lds_slots / 4 - 1;
    lds_cache_init(cache);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n_subjects) {
        const uint32_t slice = i / a.bins, colm = i - slice * a.bins;
        const uint32_t* p = a.slab + (size_t)slice * a.n_teams * a.bins + colm;
        unsigned long long w = 0;
#pragma unroll 8
        for (uint32_t t = 0; t < a.n_teams; ++t) w += p[(size_t)t * a.bins];
        const uint32_t wraps = a.hi[i];
        if (wraps) {
            w += (unsigned long long)wraps << 32;
            a.hi[i] = 0u;  // clean for the next launch
        }
        if (w) {
            const int32_t* row = a.rows + (size_t)i * a.row_w;
            for (int jb = 0; jb < a.n_jobs; ++jb) {
                const int32_t f = a.mode[jb] == WK_MODE_NONE ? row[0] : row[1 + a.col[jb]];
                if (f < 0 || (uint32_t)f > (uint32_t)WK_MAX_FEATURE) {
                    atomicOr(a.table.err, kErrFeatureRange);  // (cannot happen: such subjects are flagged)
                    continue;
                }
                cached_add(cache, a.table, make_key((uint32_t)jb, 0u, a.group, (uint32_t)f), w);
            }
        }
    }
    lds_cache_flush(cache, a.table);
}

// One bit per subject: the weighted histogram does not cover reads that name it
// (no ancestor at one of the rank columns in use, or a feature id outside the
// key range).  *any is set when at least one bit is.
__global__ void __launch_bounds__(256) subject_invalid_kernel(const int32_t* __restrict__ rows, int32_t row_w,
                                                              int32_t n_cols, int32_t n_subjects,
                                                              uint32_t* __restrict__ bits, uint32_t* __restrict__ any) {
    const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (s < n_subjects) {
        const int32_t* row = rows + (size_t)s * row_w;
        bad = (uint32_t)row[0] > (uint32_t)WK_MAX_FEATURE;
        for (int32_t c = 0; c < n_cols; ++c) bad |= row[1 + c] < 0;
    }
    const unsigned long long m = __ballot(bad);
    const uint32_t lane = threadIdx.x & (kWave - 1);
    if (lane == 0 && s < n_subjects) bits[s >> 5] = (uint32_t)m;
    if (lane == 32 && s < n_subjects) bits[s >> 5] = (uint32_t)(m >> 32);
    if (lane == 0 && m) atomicOr(any, 1u);
}

}  // namespace wk
