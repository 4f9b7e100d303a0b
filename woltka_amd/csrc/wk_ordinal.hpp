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
// Here the genes of a genome are sorted once by start (and carry a running
// maximum of their ends); each hit then needs one binary search plus a short
// backward scan in cache-resident tables, so reads are never sorted and their
// records are touched exactly once per pass.
#pragma once
#include "wk_device.hpp"

namespace wk {

constexpr int kMatchThreads = 256;
constexpr int kMatchItems = 4;
constexpr int kMatchTile = kMatchThreads * kMatchItems;

struct MatchArgs {
    // per hit
    const int32_t* genome;
    const int32_t* beg;
    const int32_t* end;
    const uint32_t* len;
    int64_t n_hits;
    double th;
    // gene tables
    const int32_t* genome_off;  // [n_genomes + 1]
    const int32_t* gstart;      // [n_genes] start0, ascending per genome
    const int32_t* gend;        // [n_genes]
    const int32_t* gpmax;       // [n_genes] running max of gend inside the genome
    const int32_t* gfeat;       // [n_genes] feature id
    const int4* gene4;          // [n_genes] {start0, end, running max end, feature}: one gather per scanned gene
    const int4* ginfo;          // [n_genomes] {first gene, gene count, smallest start0, float bits of (count-1)/(largest-smallest start0)}
    int32_t n_genomes;
    int32_t ablate;  // measurement builds only (-DWK_ABLATE): 1 = no scan, 2 = no search, 4 = no table lookups at all
};

// rel = ceil(len * th) evaluated in fp64 exactly like numpy does in
// ordinal.py:281 (uint32 -> float64 is exact, one IEEE multiply, ceil).
__device__ __forceinline__ int64_t effective_len(uint32_t len, double th) {
    return (int64_t)ceil((double)len * th);
}

struct HitQuery {
    int64_t rs, re, rel;
    int32_t lo, hi;  // gene range of the hit's genome (empty: hit cannot match)
    int32_t first;        // smallest gene start0 of the genome
    float scale;          // genes per base between the first and the last start
};

__device__ __forceinline__ HitQuery load_hit(const MatchArgs& a, int64_t h) {
    HitQuery q{0, 0, 1, 0, 0, 0, 0.f};
    if (h >= a.n_hits) return q;
    const int32_t g = a.genome[h];
    const uint32_t len = a.len[h];
    if (g < 0 || g >= a.n_genomes || len == 0) return q;  // ordinal.py:231, 294-297
    q.rs = a.beg[h];
    q.re = a.end[h];
    q.rel = effective_len(len, a.th);
#ifdef WK_ABLATE
    if (a.ablate & 4) return q;
#endif
    const int4 gi = a.ginfo[g];
    q.lo = gi.x;
    q.hi = gi.x + gi.y;
    q.first = gi.z;
    q.scale = __int_as_float(gi.w);
    return q;
}

// Backward scan from the upper bound `ub` (first gene with start0 > re - rel):
// a matching gene starts at or before re - rel and ends at or after rs + rel;
// the running maximum of the ends stops the scan.
template <typename F>
__device__ __forceinline__ void scan_matches(const MatchArgs& a, const HitQuery& q, int32_t ub, F&& f) {
    const int64_t min_end = q.rs + q.rel;
    for (int32_t j = ub - 1; j >= q.lo; --j) {
        const int4 g = a.gene4[j];
        if ((int64_t)g.z < min_end) break;  // nothing at or before j reaches the hit
        const int64_t gs = g.x;
        const int64_t ge = g.y;
        const int64_t ov = (ge < q.re ? ge : q.re) - (gs > q.rs ? gs : q.rs);
        if (ov >= q.rel) f(g.w);
    }
}

// Pass 1: number of matching genes per hit + per-tile totals.  The binary
// searches of a thread's kMatchItems hits advance in lock step, so their
// gathers (cache-resident gene starts) are issued back to back; the upper
// bound is kept for pass 2.
__global__ void __launch_bounds__(kMatchThreads) match_count_kernel(MatchArgs a,
                                                                    int32_t* __restrict__ cnt,
                                                                    int32_t* __restrict__ ubound,
                                                                    int2* __restrict__ first2,
                                                                    unsigned long long* __restrict__ tile_sum) {
    __shared__ unsigned long long wsum[kMatchThreads / kWave];
    const int64_t base = (int64_t)blockIdx.x * kMatchTile;
    HitQuery q[kMatchItems];
    int32_t l[kMatchItems], r[kMatchItems];
#pragma unroll
    for (int it = 0; it < kMatchItems; ++it) {
        q[it] = load_hit(a, base + it * kMatchThreads + threadIdx.x);
        l[it] = q[it].lo;
        r[it] = q[it].hi;
    }
#ifdef WK_ABLATE
    if (a.ablate & 2) {
#pragma unroll
        for (int it = 0; it < kMatchItems; ++it) q[it].hi = q[it].lo + (q[it].hi > q[it].lo ? 1 : 0);
    }
#endif
    // Upper bound = first gene with start0 > re - rel.  Genes are spread fairly
    // evenly along a genome, so an interpolated guess lands within a few genes
    // of it; a window of 8 genes around the guess is verified with two gathers
    // and only when that fails does the search widen (binary search in the
    // remaining half).  All kMatchItems searches advance in lock step.
#pragma unroll
    for (int it = 0; it < kMatchItems; ++it) {
        const int64_t t = q[it].re - q[it].rel;
        const int32_t n = q[it].hi - q[it].lo;
        if (n > 16 && t >= q[it].first) {
            // the guess only picks the window that is then verified, so float
            // precision is irrelevant for correctness
            int32_t guess = q[it].lo + (int32_t)((float)(t - q[it].first) * q[it].scale);
            guess = guess > q[it].hi - 1 ? q[it].hi - 1 : guess;
            int32_t wl = guess - 4, wr = guess + 4;
            wl = wl < q[it].lo ? q[it].lo : wl;
            wr = wr > q[it].hi - 1 ? q[it].hi - 1 : wr;
            l[it] = wl;  // provisional window [wl, wr]
            r[it] = wr;
        } else {
            l[it] = r[it] = -1;  // no window: plain binary search over the genome
        }
    }
    {
        int32_t vl[kMatchItems], vr[kMatchItems];
#pragma unroll
        for (int it = 0; it < kMatchItems; ++it) {
            vl[it] = (l[it] >= 0) ? a.gstart[l[it]] : 0;
            vr[it] = (l[it] >= 0) ? a.gstart[r[it]] : 0;
        }
#pragma unroll
        for (int it = 0; it < kMatchItems; ++it) {
            const int64_t t = q[it].re - q[it].rel;
            if (l[it] < 0) {
                l[it] = q[it].lo;
                r[it] = q[it].hi;
            } else if ((int64_t)vl[it] > t) {  // answer at or left of the window
                r[it] = l[it];
                l[it] = q[it].lo;
            } else if ((int64_t)vr[it] <= t) {  // answer right of the window
                l[it] = r[it] + 1;
                r[it] = q[it].hi;
            } else {  // gstart[wl] <= t < gstart[wr]: answer inside (wl, wr]
                l[it] = l[it] + 1;
            }
        }
    }
    bool more = true;
    while (more) {  // binary search of the (narrowed) range
        int32_t m[kMatchItems], v[kMatchItems];
#pragma unroll
        for (int it = 0; it < kMatchItems; ++it) {
            m[it] = l[it] + ((r[it] - l[it]) >> 1);
            v[it] = (l[it] < r[it]) ? a.gstart[m[it]] : 0;
        }
        more = false;
#pragma unroll
        for (int it = 0; it < kMatchItems; ++it) {
            if (l[it] < r[it]) {
                if ((int64_t)v[it] <= q[it].re - q[it].rel)
                    l[it] = m[it] + 1;
                else
                    r[it] = m[it];
            }
            more |= l[it] < r[it];
        }
    }
    unsigned long long mine = 0;
#pragma unroll
    for (int it = 0; it < kMatchItems; ++it) {
        const int64_t h = base + it * kMatchThreads + threadIdx.x;
        if (h < a.n_hits) {
            int32_t c = 0;
            int2 f2 = make_int2(-1, -1);  // the first two matches ride along: pass 2 rescans only hits with more
#ifdef WK_ABLATE
            if (!(a.ablate & 1))
#endif
                scan_matches(a, q[it], l[it], [&](int32_t feat) {
                    if (c == 0) f2.x = feat;
                    if (c == 1) f2.y = feat;
                    c += 1;
                });
            cnt[h] = c;
            if (c > 0) first2[h] = f2;      // (pass 2 reads these only for such hits)
            if (c > 2) ubound[h] = l[it];
            mine += (unsigned long long)c;
        }
    }
    mine = wave_sum(mine);
    if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x / kWave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long s = 0;
        for (int w = 0; w < kMatchThreads / kWave; ++w) s += wsum[w];
        tile_sum[blockIdx.x] = s;
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
                                                                    const int32_t* __restrict__ ubound,
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
            if (c > 2) {  // rare (nested / overlapping genes): scan again from the stored upper bound
                int64_t w = o;
                scan_matches(a, load_hit(a, h), ubound[h], [&](int32_t feat) { pairs[w++] = feat; });
            } else if (c > 0) {
                const int2 f2 = first2[h];
                pairs[o] = f2.x;
                if (c == 2) pairs[o + 1] = f2.y;
            }
        }
    }
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

}  // namespace wk
