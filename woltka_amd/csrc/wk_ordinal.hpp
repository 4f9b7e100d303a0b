// wk_ordinal.hpp — coord-match ("ordinal") read <-> gene interval overlap.
//
// Reproduces ordinal.flush_chunk + match_read_gene / match_read_gene_quart
// (woltka/ordinal.py:243-335, 476-582, 650-811).  All three reference
// matchers decide the same predicate (ordinal.py:555, 580, 644-645):
//
//     hit (rs, re, rel) matches gene (gs, ge)
//         <=>  min(ge, re) - max(gs, rs) >= rel,        rel = ceil(len * th) >= 1
//
// The reference evaluates it with a sweep over a merged, sorted queue of all
// gene and read end points per genome, which needs the reads sorted per chunk.
// Here the genes of a genome are sorted once by start; every gene record
// carries the running maximum of the ends *before* it, and a grid over the
// genome's coordinates (about one cell per gene) gives, in one gather, a gene
// index at or behind the last gene that can still start early enough.  A hit
// is then matched by walking gene records backwards from there: genes that
// start too late fail the predicate like any other non-match, and the walk
// stops when nothing before the current gene reaches the hit.  Reads are
// never sorted, their records are read once, and a hit costs about three
// cache-line requests (grid cell, ~2 gene records) — the kernel is bound by
// the request rate of the L2 (one line per lane per gather), not by bytes.
#pragma once
#include "wk_device.hpp"

namespace wk {

constexpr int kMatchThreads = 256;
constexpr int kMatchItems = 4;
constexpr int kMatchTile = kMatchThreads * kMatchItems;  // hits per tile (unit of the offset scan)
constexpr int kMatchGroup = 2;                            // tiles per workgroup round (512 threads: three workgroups per CU)

struct MatchArgs {
    // per hit
    const int32_t* genome;
    const int32_t* beg;
    const int32_t* end;
    const uint32_t* len;
    int64_t n_hits;
    double th;
    // gene tables (wk_set_genes)
    const int4* gene4;       // [n_genes] {start0, end, largest end of the genes before it in the genome (INT32_MIN: none), feature}
    const int32_t* grid;     // per genome cells + 1 entries: grid[c] = first gene of the genome whose cell is >= c
    const int32_t* gfirst;   // [n_genomes] smallest start0
    const int32_t* goff;     // [n_genomes + 1] offset of the genome's cells in grid
    const unsigned char* gshift;  // [n_genomes] cell of start s = (s - first) >> shift
    int32_t n_genomes;
    int32_t ablate;
};

// rel = ceil(len * th) evaluated in fp64 exactly like numpy does in
// ordinal.py:281 (uint32 -> float64 is exact, one IEEE multiply, ceil).
__device__ __forceinline__ int64_t effective_len(uint32_t len, double th) {
    return (int64_t)ceil((double)len * th);
}

// Walk back from gene j (>= the genome's first gene): `f(feature)` for every
// match.
template <typename F>
__device__ __forceinline__ void scan_matches(const MatchArgs& a, int64_t rs, int64_t re, int64_t rel, int32_t j, F&& f) {
    const int64_t min_end = rs + rel;
    for (;; --j) {
        const int4 g = a.gene4[j];
        const int64_t gs = g.x, ge = g.y;
        const int64_t ov = (ge < re ? ge : re) - (gs > rs ? gs : rs);
        if (ov >= rel) f(g.w);
        if ((int64_t)g.z < min_end) break;  // nothing before j reaches the hit (first gene of a genome: INT32_MIN)
    }
}

// Matching genes per hit.  first2[h] = the first two matches in walk order
// ({-1, -1}: none, {x, -1}: one, {x, y}: two, {x, -2}: more than two — their
// list is written by match_write_kernel from start[h]).  kCounts: also the
// count per hit and the per-tile totals for the offset scan.
// kLdsInfo: the per-genome words live in LDS (9 bytes per genome).
template <bool kLdsInfo, bool kCounts>
__global__ void __launch_bounds__(kMatchThreads * kMatchGroup, kCounts ? 4 : 6) match_hits_kernel(MatchArgs a, int2* __restrict__ first2,
                                                                                int32_t* __restrict__ start,
                                                                                int32_t* __restrict__ cnt,
                                                                                unsigned long long* __restrict__ tile_sum) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long wsum[kMatchThreads * kMatchGroup / kWave];
    const int32_t* l_first = a.gfirst;
    const int32_t* l_goff = a.goff;
    const unsigned char* l_shift = a.gshift;
    if constexpr (kLdsInfo) {
        int32_t* f = reinterpret_cast<int32_t*>(smem);
        int32_t* o = f + a.n_genomes;
        unsigned char* sh = reinterpret_cast<unsigned char*>(o + a.n_genomes + 1);
        for (int32_t i = threadIdx.x; i < a.n_genomes; i += blockDim.x) {
            f[i] = a.gfirst[i];
            sh[i] = a.gshift[i];
        }
        for (int32_t i = threadIdx.x; i <= a.n_genomes; i += blockDim.x) o[i] = a.goff[i];
        __syncthreads();
        l_first = f;
        l_goff = o;
        l_shift = sh;
    }
    const int64_t n_tiles = (a.n_hits + kMatchTile - 1) / kMatchTile;
    const uint32_t sub = __builtin_amdgcn_readfirstlane(threadIdx.x / kMatchThreads), tid = threadIdx.x % kMatchThreads;
    for (int64_t round = blockIdx.x; round * kMatchGroup < n_tiles; round += gridDim.x) {
        const int64_t tile = round * kMatchGroup + sub;
        const int64_t base = tile * kMatchTile;
        int32_t rs[kMatchItems], re[kMatchItems], j[kMatchItems], c[kMatchItems], fx[kMatchItems], fy[kMatchItems];
        uint32_t rel[kMatchItems];  // (ceil(len * th) <= len for th <= 1; larger values saturate: such a hit matches nothing either way)
        int32_t at[kMatchItems];
#pragma unroll
        for (int it = 0; it < kMatchItems; ++it) {
            const int64_t h = base + it * kMatchThreads + tid;
            int32_t g = -1;
            uint32_t len = 0;
            rs[it] = re[it] = 0;
            if (h < a.n_hits) {
                g = a.genome[h];
                len = a.len[h];
                rs[it] = a.beg[h];
                re[it] = a.end[h];
            }
            const int64_t rel64 = effective_len(len, a.th);
            rel[it] = rel64 > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)rel64;
            at[it] = -1;
            if (g >= 0 && g < a.n_genomes && len != 0) {  // ordinal.py:231, 294-297
                // genes that can match start at or before re - rel
                const int64_t t = (int64_t)re[it] - (int64_t)rel[it];
                const int32_t first = l_first[g], o0 = l_goff[g], o1 = l_goff[g + 1];
                if (o1 - o0 > 1 && t >= (int64_t)first) {
                    int64_t cell = (t - first) >> l_shift[g];
                    const int64_t last = o1 - o0 - 2;
                    cell = cell > last ? last : cell;
                    at[it] = o0 + (int32_t)cell + 1;  // first gene behind t's cell
                }
            }
        }
#pragma unroll
        for (int it = 0; it < kMatchItems; ++it) {
            j[it] = at[it] >= 0 ? a.grid[at[it]] - 1 : -1;  // (>= the genome's first gene: its cell is <= t's)
            c[it] = 0;
            fx[it] = fy[it] = -1;
            if (kCounts && at[it] >= 0) at[it] = j[it];
        }
        // the walks of a thread's hits advance in lock step: their gathers are
        // issued back to back
        bool more = (j[0] >= 0) | (j[1] >= 0) | (j[2] >= 0) | (j[3] >= 0);
        static_assert(kMatchItems == 4, "the lock-step test above names four items");
        while (more) {
            int4 g[kMatchItems];
#pragma unroll
            for (int it = 0; it < kMatchItems; ++it) g[it] = j[it] >= 0 ? a.gene4[j[it]] : make_int4(0, 0, 0, 0);
            more = false;
#pragma unroll
            for (int it = 0; it < kMatchItems; ++it) {
                if (j[it] >= 0) {
                    const int64_t gs = g[it].x, ge = g[it].y;
                    const int64_t ov = (ge < re[it] ? ge : (int64_t)re[it]) - (gs > rs[it] ? gs : (int64_t)rs[it]);
                    if (ov >= (int64_t)rel[it]) {
                        if (c[it] == 0) fx[it] = g[it].w;
                        if (c[it] == 1) fy[it] = g[it].w;
                        c[it] += 1;
                    }
                    j[it] = ((int64_t)g[it].z >= (int64_t)rs[it] + (int64_t)rel[it]) ? j[it] - 1 : -1;
                    more |= j[it] >= 0;
                }
            }
        }
        unsigned long long mine = 0;
#pragma unroll
        for (int it = 0; it < kMatchItems; ++it) {
            const int64_t h = base + it * kMatchThreads + tid;
            if (h < a.n_hits) {
                first2[h] = make_int2(fx[it], c[it] > 2 ? -2 : fy[it]);
                if constexpr (kCounts) {
                    cnt[h] = c[it];
                    if (c[it] > 2) start[h] = at[it];
                    mine += (unsigned long long)c[it];
                }
            }
        }
        if constexpr (kCounts) {
            mine = wave_sum(mine);
            if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x / kWave] = mine;
            __syncthreads();
            if (tid == 0 && tile < n_tiles) {
                unsigned long long t = 0;
                for (int w = 0; w < kMatchThreads / kWave; ++w) t += wsum[sub * (kMatchThreads / kWave) + w];
                tile_sum[tile] = t;
            }
            __syncthreads();
        }
    }
}

// Exclusive scan of the tile totals (single workgroup; n_tiles is n_hits / 1024).
// A round covers 8 tiles per thread: serial prefix inside the thread, shuffle
// scan of the thread totals inside the wave, wave totals combined through LDS —
// two barriers per 8192 tiles.
__global__ void __launch_bounds__(1024) tile_scan_kernel(const unsigned long long* __restrict__ tile_sum,
                                                         unsigned long long* __restrict__ tile_off,
                                                         int64_t n_tiles,
                                                         unsigned long long* __restrict__ total) {
    constexpr int kPer = 8;
    __shared__ unsigned long long wave_tot[16];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    unsigned long long carry = 0;  // identical in every thread
    for (int64_t base = 0; base < n_tiles; base += (int64_t)blockDim.x * kPer) {
        const int64_t first = base + (int64_t)threadIdx.x * kPer;
        unsigned long long v[kPer], mine = 0;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            v[k] = (first + k < n_tiles) ? tile_sum[first + k] : 0ull;
            mine += v[k];
        }
        unsigned long long inc = mine;  // inclusive scan of the thread totals inside the wave
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const unsigned long long up = __shfl_up(inc, off, kWave);
            if (lane >= off) inc += up;
        }
        if (lane == kWave - 1) wave_tot[wave] = inc;
        __syncthreads();
        unsigned long long before = 0, round_total = 0;
        for (int q = 0; q < n_waves; ++q) {
            const unsigned long long t = wave_tot[q];
            before += q < wave ? t : 0ull;
            round_total += t;
        }
        unsigned long long run = carry + before + inc - mine;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            if (first + k < n_tiles) tile_off[first + k] = run;
            run += v[k];
        }
        carry += round_total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

// Pass 2: exclusive offsets per hit (tile-local scan + tile offset) and the
// matched gene feature ids.
__global__ void __launch_bounds__(kMatchThreads) match_write_kernel(MatchArgs a,
                                                                    const int32_t* __restrict__ cnt,
                                                                    const int32_t* __restrict__ start,
                                                                    const int2* __restrict__ first2,
                                                                    const unsigned long long* __restrict__ tile_off,
                                                                    int32_t* __restrict__ poff,
                                                                    int32_t* __restrict__ pairs) {
    __shared__ int32_t scan[kMatchTile];
    const int64_t base = (int64_t)blockIdx.x * kMatchTile;
    // load counts of the tile (hit order = LDS order)
#pragma unroll
    for (int it = 0; it < kMatchItems; ++it) {
        const int idx = it * kMatchThreads + threadIdx.x;
        const int64_t h = base + idx;
        scan[idx] = (h < a.n_hits) ? cnt[h] : 0;
    }
    __syncthreads();
    // each thread serially scans kMatchItems consecutive entries, then the
    // per-thread totals are scanned across the workgroup
    __shared__ int32_t tsum[kMatchThreads];
    {
        int32_t s = 0;
        const int b = threadIdx.x * kMatchItems;
#pragma unroll
        for (int k = 0; k < kMatchItems; ++k) {
            const int32_t v = scan[b + k];
            scan[b + k] = s;
            s += v;
        }
        tsum[threadIdx.x] = s;
    }
    __syncthreads();
    for (int off = 1; off < kMatchThreads; off <<= 1) {
        int32_t add = (threadIdx.x >= (unsigned)off) ? tsum[threadIdx.x - off] : 0;
        __syncthreads();
        tsum[threadIdx.x] += add;
        __syncthreads();
    }
    {
        const int b = threadIdx.x * kMatchItems;
        const int32_t pre = (threadIdx.x == 0) ? 0 : tsum[threadIdx.x - 1];
#pragma unroll
        for (int k = 0; k < kMatchItems; ++k) scan[b + k] += pre;
    }
    __syncthreads();
    const int64_t toff = (int64_t)tile_off[blockIdx.x];
#pragma unroll
    for (int it = 0; it < kMatchItems; ++it) {
        const int idx = it * kMatchThreads + threadIdx.x;
        const int64_t h = base + idx;
        if (h < a.n_hits) {
            const int64_t o = toff + scan[idx];
            poff[h] = (int32_t)o;
            const int32_t c = cnt[h];
            if (c > 2) {  // rare (nested / overlapping genes): walk again from the stored start
                int64_t w = o;
                const int64_t rs = a.beg[h], re = a.end[h];
                scan_matches(a, rs, re, effective_len(a.len[h], a.th), start[h], [&](int32_t feat) { pairs[w++] = feat; });
            } else if (c > 0) {
                const int2 f2 = first2[h];
                pairs[o] = f2.x;
                if (c == 2) pairs[o + 1] = f2.y;
            }
        }
    }
}

// out[i] = table[idx[i]] (gene table indices -> their features)
__global__ void __launch_bounds__(256) gather_i32_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ table,
                                                         int64_t n, int64_t n_table, int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t j = idx[i];
    out[i] = (j >= 0 && j < n_table) ? table[j] : -1;
}

// Per-read gene offsets: the hits of a read are contiguous, so the genes of a
// read are the concatenation of its hits' genes (union taken later by the
// classify kernel's duplicate removal; ordinal.py:331-332 builds a set).
__global__ void __launch_bounds__(256) read_offsets_kernel(const int32_t* __restrict__ hoff,
                                                           const int32_t* __restrict__ poff,
                                                           int64_t n_reads, int64_t n_hits,
                                                           const unsigned long long* __restrict__ total,
                                                           int32_t* __restrict__ qoff) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_reads) return;
    const int32_t h = hoff[r];
    qoff[r] = (h >= n_hits) ? (int32_t)(*total) : poff[h];
}

// ---- genes counted per read, straight from the matches ---------------------------
// `--coords` without a further classification (rank none: the profile's features
// are the genes, ordinal.py:331-332 + classify.assign_none + classify.counter):
// a read with n distinct genes adds 1/n to each.  One thread per read collects
// the genes of its hits from first2[] — the usual read has one or two hits
// with at most two genes each — and appends the weighted keys to the
// partitioned log (partition_merge_kernel counts them); no gene lists, offsets
// or generic classify pass.  Reads it does not cover (a hit with more than two
// genes, more than kTallySlots distinct genes, more than kTallyHits hits) get a
// bit in `left_mask`; the caller materialises the gene lists and runs the
// generic evaluator on those reads only when there is one.
constexpr int kTallySlots = 8;
constexpr int kTallyHits = 16;

struct TallyArgs {
    const int32_t* hoff;  // [n_reads + 1]
    const int2* first2;   // [n_hits]
    int64_t n_reads;
    int32_t n_jobs;               // plain rank-none jobs (all count the same keys under their job index)
    int32_t job_index[WK_MAX_JOBS];
    int32_t group;
    CountTable table;
    unsigned long long* plog;  // [gridDim.x][log_parts][plog_cap]
    uint32_t* plog_cnt;        // [gridDim.x][log_parts]
    uint32_t plog_cap;
    uint32_t log_parts;
    uint32_t* plog32;          // kRange: the log as 4-byte entries, gene | n << 28
                               // (partition = the gene's low bits: log_parts is a power of two)
    unsigned long long* stat_block;
    unsigned long long* left_mask;  // [ceil(n_reads / 64)]
    unsigned long long* n_left;  // [2]: reads left over, pairs of the tallied reads
};

constexpr uint32_t kTallyThreads = 512;
constexpr uint32_t kTallyQueue = 3072;  // reads with several hits wait here until a full workgroup's worth is queued

// kRange (one job; gene ids below 2^28, at most 16384 per partition): the genes of
// a chunk are ~10^5..10^6 keys, so an LDS hash cache in front of the log holds
// next to none of them and its probes are most of this kernel's time.  Every
// {gene, n} goes to the log instead, as one 4-byte entry, into the partition its
// low bits name (neighbouring genes — one abundant genome — spread over all
// partitions; partitions by gene *range* were 4x slower: a few streams overflow
// and their cursors serialise); range_merge_kernel then adds a partition up in a
// dense LDS array indexed by the high bits — no hashing on either side.
template <bool kRange>
__global__ void __launch_bounds__(kTallyThreads) ordinal_tally_kernel(TallyArgs a, uint32_t lds_slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long acc[3];
    __shared__ uint32_t q_n;
    // LDS: hash cache | log cursors | queue | gene sets of the queued reads
    LdsCache cache{};
    cache.base = reinterpret_cast<unsigned long long*>(smem);
    cache.bmask = lds_slots / 4 - 1;
    cache.plog_cur = reinterpret_cast<uint32_t*>(smem + (size_t)lds_slots * 16);
    cache.plog = a.plog + (size_t)blockIdx.x * a.log_parts * a.plog_cap;
    cache.plog_cap = a.plog_cap;
    cache.plog_shift = (uint32_t)__clz((int)a.log_parts) + 1u;
    uint32_t* const queue = cache.plog_cur + a.log_parts;                            // [kTallyQueue] read indices
    int32_t* const sets = reinterpret_cast<int32_t*>(queue + kTallyQueue);         // [kTallySlots][kTallyThreads]
    for (uint32_t i = threadIdx.x; i < a.log_parts; i += blockDim.x) cache.plog_cur[i] = 0u;
    if (threadIdx.x < 3) acc[threadIdx.x] = 0ull;
    if (threadIdx.x == 0) q_n = 0u;
    lds_cache_init(cache);  // (ends with a barrier; lds_slots = 0 under kRange: no cache)
    const bool one_job = a.n_jobs == 1;
    const uint32_t job0 = (uint32_t)a.job_index[0];
    uint32_t* const plog32 = a.plog32 + (size_t)blockIdx.x * a.log_parts * a.plog_cap;
    auto add_key = [&](uint64_t key, uint32_t n) { cached_add(cache, a.table, key, (unsigned long long)weight_of(n)); };
    auto add = [&](int32_t feature, uint32_t n) {
        if constexpr (kRange) {
            const uint32_t part = (uint32_t)feature & (a.log_parts - 1u);
            const uint32_t pos = atomicAdd(&cache.plog_cur[part], 1u);
            if (pos < a.plog_cap)
                plog32[(size_t)part * a.plog_cap + pos] = (uint32_t)feature | (n << 28);
            else  // the stream is full: count in HBM directly
                table_add(a.table, make_key(job0, 0u, (uint32_t)a.group, (uint32_t)feature), (unsigned long long)weight_of(n));
        } else if (one_job) {
            add_key(make_key(job0, 0u, (uint32_t)a.group, (uint32_t)feature), n);
        } else {
            for (int32_t jb = 0; jb < a.n_jobs; ++jb)
                add_key(make_key((uint32_t)a.job_index[jb], 0u, (uint32_t)a.group, (uint32_t)feature), n);
        }
    };
    unsigned long long my_reads = 0, my_records = 0, my_left = 0;
    const int64_t n_words = (a.n_reads + 63) >> 6;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    // A read with several hits: the distinct genes of its hits, in this
    // thread's column of `sets`; all threads of the workgroup work on queued
    // reads together, so the dependent loads of the later hits and the
    // set loops run with full waves.
    auto several = [&](uint32_t r) {
        const int32_t h0 = a.hoff[r], nh = a.hoff[r + 1] - h0;
        int32_t* const mine = sets + threadIdx.x;
        int n = 0, total = 0;
        bool left = nh > kTallyHits;
        // (the first three hits' matches in flight together)
        const int2 p0 = a.first2[h0], p1 = a.first2[h0 + 1], p2 = nh > 2 ? a.first2[h0 + 2] : make_int2(-1, -1);
        for (int32_t i = 0; i < nh && !left; ++i) {
            const int2 g2 = i == 0 ? p0 : i == 1 ? p1 : i == 2 ? p2 : a.first2[h0 + i];
            if (g2.y == -2) left = true;
            const int32_t cand[2] = {g2.x, g2.y};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (cand[q] < 0) continue;
                total += 1;
                bool dup = false;
                for (int z = 0; z < n; ++z) dup |= mine[z * kTallyThreads] == cand[q];
                if (dup) continue;
                if (n == kTallySlots) {
                    left = true;
                } else {
                    mine[n * kTallyThreads] = cand[q];
                    n += 1;
                }
            }
        }
        if (left) {
            atomicOr(&a.left_mask[r >> 6], 1ull << (r & 63u));
            my_left += 1;
        } else if (n > 0) {
            my_reads += 1;
            my_records += (unsigned long long)total;
            for (int z = 0; z < n; ++z) add(mine[z * kTallyThreads], (uint32_t)n);
        }
    };
    auto drain = [&]() {  // (between two barriers; every thread calls it)
        const uint32_t n = q_n;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) several(queue[i]);
        __syncthreads();
        if (threadIdx.x == 0) q_n = 0u;
    };
    // a wave takes kTallyWords consecutive words of 64 reads per round; the
    // offsets of all of them, then the matches of their first hits, are in
    // flight together.  Rounds are counted per workgroup (barriers inside).
    constexpr int kTallyWords = 4;
    const int64_t per_round = (int64_t)gridDim.x * waves * kTallyWords;
    const int64_t n_rounds = (n_words + per_round - 1) / per_round;
    for (int64_t round = 0; round < n_rounds; ++round) {
        const int64_t w0 = round * per_round + ((int64_t)blockIdx.x * waves + wave) * kTallyWords;
        int32_t h0[kTallyWords], nh[kTallyWords];
        int2 f[kTallyWords];
#pragma unroll
        for (int k = 0; k < kTallyWords; ++k) {
            const int64_t r = (w0 + k) * 64 + lane;
            int32_t e = 0;
            h0[k] = 0;
            if (r < a.n_reads) {
                h0[k] = a.hoff[r];
                e = a.hoff[r + 1];
            }
            nh[k] = e - h0[k];
        }
#pragma unroll
        for (int k = 0; k < kTallyWords; ++k) f[k] = nh[k] == 1 ? a.first2[h0[k]] : make_int2(-1, -1);
#pragma unroll
        for (int k = 0; k < kTallyWords; ++k) {
            bool left = false;
            if (nh[k] == 1) {
                if (f[k].y == -2) {
                    left = true;
                } else if (f[k].x >= 0) {
                    const bool two = f[k].y >= 0 && f[k].y != f[k].x;
                    my_reads += 1;
                    my_records += f[k].y >= 0 ? 2 : 1;
                    add(f[k].x, two ? 2u : 1u);
                    if (two) add(f[k].y, 2u);
                }
            } else if (nh[k] > 1) {
                queue[atomicAdd(&q_n, 1u)] = (uint32_t)((w0 + k) * 64 + lane);
            }
            // (bits of queued reads are added when they are drained)
            const unsigned long long bits = __ballot(left);
            if (lane == 0 && w0 + k < n_words) a.left_mask[w0 + k] = bits;
            my_left += left ? 1ull : 0ull;
        }
        // a round queues at most kTallyThreads * kTallyWords reads
        __syncthreads();
        if (q_n > kTallyQueue - kTallyThreads * kTallyWords || round + 1 == n_rounds) drain();
        __syncthreads();
    }
    my_reads = wave_sum(my_reads);
    my_records = wave_sum(my_records);
    my_left = wave_sum(my_left);
    if (lane == 0) {
        atomicAdd(&acc[0], my_reads);
        atomicAdd(&acc[1], my_records);
        atomicAdd(&acc[2], my_left);
    }
    if constexpr (kRange)
        __syncthreads();
    else
        lds_cache_flush(cache, a.table);  // (starts with a barrier)
    if (threadIdx.x == 0) {
        a.stat_block[2 * blockIdx.x] += acc[0];
        a.stat_block[2 * blockIdx.x + 1] += acc[1];
        if (acc[2]) atomicAdd(a.n_left, acc[2]);
        atomicAdd(a.n_left + 1, acc[1]);  // pairs of the tallied reads
    }
    for (uint32_t i = threadIdx.x; i < a.log_parts; i += blockDim.x) {
        const uint32_t n = cache.plog_cur[i];
        a.plog_cnt[(size_t)blockIdx.x * a.log_parts + i] = n < a.plog_cap ? n : a.plog_cap;
    }
}

// The log of ordinal_tally_kernel<true>: workgroup p adds up the entries of
// partition p — the genes g with g mod n_parts = p — from every stream in a dense
// LDS array of weights indexed by g / n_parts, and puts the genes that were hit
// into the count table.
// (kRangeMergeSplit workgroups per partition, each with every kRangeMergeSplit-th stream: the count table adds theirs up)
constexpr uint32_t kRangeMergeSplit = 8;
__global__ void __launch_bounds__(1024) range_merge_kernel(const uint32_t* __restrict__ plog32, const uint32_t* __restrict__ plog_cnt,
                                                           uint32_t n_rows, uint32_t plog_cap, uint32_t span, uint32_t job,
                                                           uint32_t group, CountTable table) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* const sum = reinterpret_cast<unsigned long long*>(smem);
    const uint32_t part = blockIdx.x, n_parts = gridDim.x, shift = 31u - (uint32_t)__clz((int)n_parts);
    const uint32_t piece = blockIdx.y, pieces = gridDim.y;
    for (uint32_t i = threadIdx.x; i < span; i += blockDim.x) sum[i] = 0ull;
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n_waves = blockDim.x >> 6;
    // one stream per wave at a time; a stream is short (~n_hits / (rows x parts)
    // entries), so the next one's length is fetched while this one is added up,
    // and its entries are loaded four to a lane before the first is used
    const uint32_t row0 = piece * n_waves + wave, stride = pieces * n_waves;
    uint32_t n = row0 < n_rows ? plog_cnt[(size_t)row0 * n_parts + part] : 0u;
    for (uint32_t row = row0; row < n_rows; row += stride) {
        const uint32_t next = row + stride;
        const uint32_t n_next = next < n_rows ? plog_cnt[(size_t)next * n_parts + part] : 0u;
        const uint32_t* src = plog32 + ((size_t)row * n_parts + part) * plog_cap;
        for (uint32_t i = lane; i < n; i += 256) {
            uint32_t e[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) e[u] = i + 64 * u < n ? src[i + 64 * u] : 0u;
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u)
                if (e[u]) atomicAdd(&sum[(e[u] & 0x0FFFFFFFu) >> shift], (unsigned long long)weight_of(e[u] >> 28));
        }
        n = n_next;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < span; i += blockDim.x)
        if (sum[i]) table_add(table, make_key(job, 0u, group, (i << shift) | part), sum[i]);
}

}  // namespace wk
