// wk_stripe.hpp — coord-match with the hits binned by genome stripe and the genes in LDS.
//
// ordinal.flush_chunk (woltka/ordinal.py:243-335) buckets a chunk's reads by genome and sweeps each
// genome's merged queue of gene and read end points (match_read_gene, ordinal.py:476-582).  The first
// three rounds kept the hits in input order and gathered 16-byte gene and grid records from HBM per
// hit (wk_ordinal.hpp: match_hits -> first2[] -> ordinal_tally -> log -> range_merge): 2.3x the
// algorithmic bytes through the fabric and a chain of dependent gathers per hit.  Here:
//
//   stripe_count / row scan / stripe_scatter   the reads of ONE hit (93 % of config 4) are counting-
//       sorted by genome STRIPE -- consecutive genomes whose genes (<= kStripeGenes) fit the LDS --
//       as 16-byte records; the other reads (several hits: their genes are a union over hits) are
//       compacted, in order, into arrays of their own and go through wk_ordinal.hpp's kernels;
//   stripe_match   a workgroup takes a piece of one stripe's hits, loads the stripe's gene records
//       and coordinate grids into LDS once, finds every hit's genes there (the grid cell of re - rel,
//       then a walk back while an earlier gene still reaches the hit: the predicate of
//       wk_ordinal.hpp, in 32-bit arithmetic) and counts them in LDS (reads with one gene / with
//       two); the counts go to the count table as 1/n weights when the piece is done.
//       No first2[], no log, no merge pass; a hit is read once (16 bytes).
//   stripe_overflow   hits with more than two genes (nested genes) are listed and counted by a
//       workgroup each, exactly (distinct features, any number up to the count key's 4095).
//
// All of it counts what classify.assign_none + counter count for `--coords` at rank none
// (classify.py:32-51, 144-171): a read with n distinct genes adds 1/n to each.
#pragma once
#include "wk_ordinal.hpp"

namespace wk {

constexpr uint32_t kStripeGenes = 2048;      // gene records of a stripe (32 KB of LDS) and two 32-bit counters each (16 KB)
constexpr uint32_t kStripeCells = 6144;      // cells of the stripe's coordinate grids (12 KB as 16-bit gene indices)
constexpr uint32_t kStripeGenomes = 256;     // genomes of a stripe (4 KB of per-genome words)
constexpr uint32_t kStripeMax = 1024;        // stripes a chunk can be sorted into (per-tile counters in LDS)
constexpr uint32_t kStripeTileThreads = 256;
constexpr uint32_t kStripeTileItems = 16;
constexpr uint32_t kStripeTileReads = kStripeTileThreads * kStripeTileItems;  // reads per workgroup of the count / scatter passes
constexpr uint32_t kStripeMatchThreads = 1024;
constexpr uint32_t kStripePiece = 65536;     // hits per workgroup of stripe_match
constexpr uint32_t kStripeStatBlocks = 4096; // (a part of the context's kStatBlocks pairs of statistics counters)

struct StripeSortArgs {
    const int32_t* genome;
    const int32_t* beg;
    const int32_t* end;
    const uint32_t* len;
    const int32_t* hoff;  // [n_reads + 1]
    int64_t n_reads;
    const int32_t* stripe_of;  // [n_genomes] stripe of a genome, -1: none (too many genes for the LDS)
    int32_t n_genomes;
    uint32_t n_stripes;
    uint32_t n_tiles;
    // rows 0 .. n_stripes - 1: reads of one hit per stripe; row n_stripes: the other reads; row n_stripes + 1: their hits
    uint32_t* cnt;             // [n_stripes + 2][n_tiles]: counts, then (after the scan) exclusive offsets inside the row
    const unsigned long long* row_base;  // [n_stripes + 2] exclusive offsets of the rows 0 .. n_stripes - 1 among the binned hits
    int4* binned;              // [single hits] {genome, beg, end, len}
    int32_t* r_genome;         // the other reads, compacted in order: hits ...
    int32_t* r_beg;
    int32_t* r_end;
    uint32_t* r_len;
    int32_t* r_hoff;           // ... and offsets [n_rest_reads + 1]
};

// the class of a read (stripe_classes below): its stripe (one hit on a genome of a stripe), n_stripes (anything else that has hits
// worth matching), 0xFFFFFFFF (nothing to match: no hits, or one hit of length 0 / on an unknown genome)

// The lanes of a wave that name the same counter share ONE LDS atomic (a Zipf sample sends half of a
// wave's reads to the same stripe: 64 atomics on one LDS word serialise).  Returns the lane's place among
// all adds to that counter (the counter's old value + its rank among the lanes that share it).
__device__ __forceinline__ uint32_t wave_shared_add(uint32_t* counters, uint32_t which, bool valid) {
    const uint32_t lane = threadIdx.x & (kWave - 1);
    unsigned long long todo = __ballot(valid);
    uint32_t place = 0;
    for (int round = 0; todo && round < 12; ++round) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t c = (uint32_t)__shfl((int)which, leader, kWave);
        const bool mine = valid && which == c;
        const unsigned long long same = __ballot(mine);
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(&counters[c], (uint32_t)__popcll(same));
        base = (uint32_t)__shfl((int)base, leader, kWave);
        if (mine) {
            place = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            valid = false;
        }
        todo &= ~same;
    }
    if (valid) place = atomicAdd(&counters[which], 1u);  // (a wave spread over many counters: the rest one by one)
    return place;
}

// The classes of a tile's reads, kStripeTileItems per thread (read = tile base + item * threads + thread: a wave's
// loads are coalesced).  Three dependent loads lead to a class -- offsets, the hit's genome, the genome's stripe --:
// each step is done for all of the thread's items before the next, so that a thread has kStripeTileItems loads
// under way instead of one (the passes are latency-bound otherwise: a tile took as long as the whole kernel).
__device__ __forceinline__ void stripe_classes(const StripeSortArgs& a, int64_t base, uint32_t (&cls)[kStripeTileItems],
                                               int32_t (&h0)[kStripeTileItems], int32_t (&nh)[kStripeTileItems]) {
#pragma unroll
    for (uint32_t it = 0; it < kStripeTileItems; ++it) {
        const int64_t r = base + it * kStripeTileThreads + threadIdx.x;
        h0[it] = 0;
        nh[it] = 0;
        if (r < a.n_reads) {
            h0[it] = a.hoff[r];
            nh[it] = a.hoff[r + 1] - h0[it];
        }
    }
    int32_t g[kStripeTileItems];
    uint32_t ln[kStripeTileItems];
#pragma unroll
    for (uint32_t it = 0; it < kStripeTileItems; ++it) {
        g[it] = -1;
        ln[it] = 0u;
        if (nh[it] == 1) {
            g[it] = a.genome[h0[it]];
            ln[it] = a.len[h0[it]];
        }
    }
#pragma unroll
    for (uint32_t it = 0; it < kStripeTileItems; ++it) {
        uint32_t c = 0xFFFFFFFFu;
        if (nh[it] > 1) {
            c = a.n_stripes;
        } else if (nh[it] == 1 && g[it] >= 0 && g[it] < a.n_genomes && ln[it] != 0u) {  // (else: ordinal.py:231, 294-297, matches nothing)
            const int32_t st = a.stripe_of[g[it]];
            c = st < 0 ? a.n_stripes : (uint32_t)st;
        }
        cls[it] = c;
    }
}

// pass 1: per tile of reads, how many go where.  (A tile is kStripeTileReads = 4096 reads: the count matrix
// -- rows x tiles, written and read with a stride of one row -- stays small against the hits themselves.)
__global__ void __launch_bounds__(kStripeTileThreads) stripe_count_kernel(StripeSortArgs a) {
    __shared__ uint32_t cnt[kStripeMax + 2];
    for (uint32_t i = threadIdx.x; i < a.n_stripes + 2u; i += blockDim.x) cnt[i] = 0u;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kStripeTileReads;
    uint32_t cls[kStripeTileItems];
    int32_t h0[kStripeTileItems], nh[kStripeTileItems];
    stripe_classes(a, base, cls, h0, nh);
    unsigned long long rest_hits = 0;
#pragma unroll
    for (uint32_t it = 0; it < kStripeTileItems; ++it) {
        (void)wave_shared_add(cnt, cls[it], cls[it] != 0xFFFFFFFFu);
        if (cls[it] == a.n_stripes) rest_hits += (unsigned long long)nh[it];
    }
    rest_hits = wave_sum(rest_hits);
    if ((threadIdx.x & (kWave - 1)) == 0 && rest_hits) atomicAdd(&cnt[a.n_stripes + 1u], (uint32_t)rest_hits);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < a.n_stripes + 2u; i += blockDim.x) a.cnt[(size_t)i * a.n_tiles + blockIdx.x] = cnt[i];
}

// pass 2: every row scanned over the tiles (a workgroup per row), the row's total to `tot`
__global__ void __launch_bounds__(1024) stripe_rows_kernel(uint32_t* __restrict__ cnt, uint32_t n_tiles, unsigned long long* __restrict__ tot) {
    constexpr uint32_t kPer = 8;
    __shared__ unsigned long long wave_tot[16];
    uint32_t* row = cnt + (size_t)blockIdx.x * n_tiles;
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    unsigned long long carry = 0;
    for (uint32_t base = 0; base < n_tiles; base += blockDim.x * kPer) {
        const uint32_t first = base + threadIdx.x * kPer;
        uint32_t v[kPer];
        unsigned long long mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            v[k] = first + k < n_tiles ? row[first + k] : 0u;
            mine += v[k];
        }
        unsigned long long inc = mine;
#pragma unroll
        for (uint32_t d = 1; d < kWave; d <<= 1) {
            const unsigned long long up = __shfl_up(inc, d, kWave);
            if (lane >= d) inc += up;
        }
        if (lane == kWave - 1) wave_tot[wave] = inc;
        __syncthreads();
        unsigned long long before = 0, round_total = 0;
        for (uint32_t q = 0; q < n_waves; ++q) {
            const unsigned long long t = wave_tot[q];
            before += q < wave ? t : 0ull;
            round_total += t;
        }
        unsigned long long run = carry + before + inc - mine;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            if (first + k < n_tiles) row[first + k] = (uint32_t)run;  // (< 2^31 hits per chunk)
            run += v[k];
        }
        carry += round_total;
        __syncthreads();
    }
    if (threadIdx.x == 0) tot[blockIdx.x] = carry;
}

// exclusive offsets of the stripes' rows among the binned hits (n <= kStripeMax + 2 values)
__global__ void __launch_bounds__(64) stripe_bases_kernel(const unsigned long long* __restrict__ tot, unsigned long long* __restrict__ base, uint32_t n_stripes) {
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (uint32_t s = 0; s < n_stripes; ++s) {
            base[s] = run;
            run += tot[s];
        }
        base[n_stripes] = run;  // all single hits
        base[n_stripes + 1] = 0;
    }
}

// pass 3: the hits to their places.  Reads are taken in rounds of 256 consecutive ones (coalesced loads); the
// reads of several hits keep their order: a scan over the workgroup per round places them.  (Measured against a
// version that classifies all of a thread's reads first, like pass 1, and scans them with two barriers per tile:
// 240 VGPRs, 3.8-5.5 ms for 100 M reads where this one takes 2.3 -- tools/sort_probe.py.)
__device__ __forceinline__ uint32_t stripe_class(const StripeSortArgs& a, int64_t r, int32_t* h0_out, int32_t* nh_out) {
    const int32_t h0 = a.hoff[r], nh = a.hoff[r + 1] - h0;
    *h0_out = h0;
    *nh_out = nh;
    if (nh <= 0) return 0xFFFFFFFFu;
    if (nh > 1) return a.n_stripes;
    const int32_t g = a.genome[h0];
    if (g < 0 || g >= a.n_genomes || a.len[h0] == 0u) return 0xFFFFFFFFu;  // ordinal.py:231, 294-297: matches nothing
    const int32_t s = a.stripe_of[g];
    return s < 0 ? a.n_stripes : (uint32_t)s;
}


__global__ void __launch_bounds__(kStripeTileThreads) stripe_scatter_kernel(StripeSortArgs a) {
    __shared__ uint32_t cur[kStripeMax];  // next place of a stripe's hits of this tile, inside the row
    __shared__ unsigned long long wtot[kStripeTileThreads / kWave];
    __shared__ unsigned long long run_base;  // (reads | hits << 32) of the other reads placed by the rounds before
    const uint32_t tile = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < a.n_stripes; i += blockDim.x) cur[i] = a.cnt[(size_t)i * a.n_tiles + tile];
    if (threadIdx.x == 0)
        run_base = (unsigned long long)a.cnt[(size_t)a.n_stripes * a.n_tiles + tile] |
                   ((unsigned long long)a.cnt[(size_t)(a.n_stripes + 1u) * a.n_tiles + tile] << 32);
    __syncthreads();
    const int64_t base = (int64_t)tile * kStripeTileReads;
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    for (uint32_t it = 0; it < kStripeTileItems; ++it) {
        const int64_t r = base + it * kStripeTileThreads + threadIdx.x;
        int32_t h0 = 0, nh = 0;
        uint32_t cls = 0xFFFFFFFFu;
        if (r < a.n_reads) cls = stripe_class(a, r, &h0, &nh);
        // the reads of one hit: a place in their stripe's run of this tile (lanes that share a stripe share the reservation)
        const bool one = cls < a.n_stripes;
        const uint32_t at = wave_shared_add(cur, one ? cls : 0u, one);
        if (one) a.binned[a.row_base[cls] + at] = make_int4(a.genome[h0], a.beg[h0], a.end[h0], (int32_t)a.len[h0]);
        // the others, in order
        const bool rest = cls == a.n_stripes;
        const unsigned long long any = __ballot(rest);
        const unsigned long long mine = rest ? (1ull | ((unsigned long long)(uint32_t)nh << 32)) : 0ull;
        unsigned long long inc = mine;
#pragma unroll
        for (uint32_t d = 1; d < kWave; d <<= 1) {
            const unsigned long long up = __shfl_up(inc, d, kWave);
            if (lane >= d) inc += up;
        }
        if (lane == kWave - 1) wtot[wave] = inc;
        (void)any;
        __syncthreads();
        unsigned long long before = run_base, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < kStripeTileThreads / kWave; ++w) {
            before += w < wave ? wtot[w] : 0ull;
            total += wtot[w];
        }
        if (rest) {
            const unsigned long long me = before + inc - mine;
            uint32_t rh = (uint32_t)(me >> 32);
            a.r_hoff[(uint32_t)(me & 0xFFFFFFFFull)] = (int32_t)rh;
            for (int32_t k = 0; k < nh; ++k) {
                const int32_t h = h0 + k;
                a.r_genome[rh] = a.genome[h];
                a.r_beg[rh] = a.beg[h];
                a.r_end[rh] = a.end[h];
                a.r_len[rh] = a.len[h];
                ++rh;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) run_base += total;
        // (the next round's barrier orders this write before its reads)
    }
}


struct StripeUnit {  // a piece of one stripe's hits
    uint32_t stripe;
    uint32_t first, count;  // among the binned hits
    uint32_t pad;
};

struct StripeInfo {  // a stripe: consecutive genomes, their genes and the cells of their coordinate grids
    int32_t gene_lo, n_genes;
    int32_t genome_lo, n_genomes;
    int32_t cell_lo, n_cells;  // [cell_lo, cell_lo + n_cells) of the grid array (wk_set_genes), all genomes' cells + 1 each
    int32_t pad0, pad1;
};

struct StripeMatchArgs {
    const int4* binned;
    const StripeUnit* units;
    const int4* gene4;         // wk_set_genes: {start0, end, largest end before the gene in its genome, feature}
    const int32_t* gene_off;   // [n_genomes + 1] genes of a genome
    const StripeInfo* stripes;
    // the coordinate grids of wk_set_genes (wk_ordinal.hpp): per genome cells + 1 entries of grid, its smallest
    // start, the offset of its cells, the cell width as a shift
    const int32_t* grid;
    const int32_t* gfirst;
    const int32_t* goff;
    const unsigned char* gshift;
    double th;
    int32_t n_jobs;
    int32_t job_index[WK_MAX_JOBS];
    int32_t group;
    CountTable table;
    uint32_t* overflow;        // hits (indices among the binned) with more than two genes
    uint32_t overflow_cap;
    unsigned long long* stat;  // [0] reads with a gene, [1] read-gene matches, [2] overflow hits
    unsigned long long* stat_block;  // the context's (reads, records) counters, kStripeStatBlocks pairs
};

// A piece of one stripe's hits against the stripe's genes in LDS.  Per hit: the cell of re - rel in the
// genome's grid (LDS) names the last gene that can start early enough; the walk back from there tests
//     gene (gs, ge) matches hit (rs, re, rel)  <=>  min(ge, re) - max(gs, rs) >= rel
//                                             <=>  ge - gs >= rel, ge >= rs + rel, gs <= re - rel  (and re - rs >= rel)
// in 32-bit arithmetic (the three bounds are per hit), and ends where no earlier gene's end reaches rs + rel.
__global__ void __launch_bounds__(kStripeMatchThreads) stripe_match_kernel(StripeMatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int4* const genes = reinterpret_cast<int4*>(smem);
    uint32_t* const c1 = reinterpret_cast<uint32_t*>(smem + (size_t)kStripeGenes * 16);     // reads whose only gene it is
    uint32_t* const c2 = c1 + kStripeGenes;                                                   // reads with two genes
    int4* const ginfo = reinterpret_cast<int4*>(c2 + kStripeGenes);                           // {first start, cell offset, cells, shift}
    unsigned short* const lgrid = reinterpret_cast<unsigned short*>(ginfo + kStripeGenomes);  // [cells] gene index in the stripe
    __shared__ unsigned long long acc[2];
    const StripeUnit u = a.units[blockIdx.x];
    const StripeInfo si = a.stripes[u.stripe];
    for (int32_t i = threadIdx.x; i < si.n_genes; i += blockDim.x) {
        genes[i] = a.gene4[si.gene_lo + i];
        c1[i] = 0u;
        c2[i] = 0u;
    }
    for (int32_t i = threadIdx.x; i < si.n_cells; i += blockDim.x) lgrid[i] = (unsigned short)(a.grid[si.cell_lo + i] - si.gene_lo);
    for (int32_t i = threadIdx.x; i < si.n_genomes; i += blockDim.x) {
        const int32_t g = si.genome_lo + i;
        const int32_t o0 = a.goff[g], o1 = a.goff[g + 1];
        ginfo[i] = make_int4(a.gfirst[g], o0 - si.cell_lo, o1 - o0, (int32_t)a.gshift[g]);
    }
    if (threadIdx.x < 2) acc[threadIdx.x] = 0ull;
    __syncthreads();
    uint32_t my_reads = 0, my_pairs = 0;
    for (uint32_t k = threadIdx.x; k < u.count; k += blockDim.x) {
        const int4 hit = a.binned[u.first + k];
        const int32_t rs = hit.y, re = hit.z;
        const int64_t rel64 = effective_len((uint32_t)hit.w, a.th);
        const int64_t a64 = (int64_t)rs + rel64, b64 = (int64_t)re - rel64;
        // (a hit shorter than rel, or bounds no 32-bit coordinate can meet: no gene)
        if (rel64 > 0xFFFFFFFFll || (int64_t)re - (int64_t)rs < rel64 || a64 > 0x7FFFFFFFll || b64 < -0x80000000ll) continue;
        const uint32_t rel = (uint32_t)rel64;
        const int32_t end_min = (int32_t)a64, start_max = (int32_t)b64;
        const int4 gi = ginfo[hit.x - si.genome_lo];
        if (gi.z <= 1 || start_max < gi.x) continue;  // a genome without genes / every gene starts too late
        uint32_t cell = (uint32_t)(start_max - gi.x) >> gi.w;
        const uint32_t last = (uint32_t)gi.z - 2u;
        cell = cell > last ? last : cell;
        int32_t j = (int32_t)lgrid[gi.y + (int32_t)cell + 1] - 1;  // (>= the genome's first gene: its cell is <= start_max's)
        int32_t n = 0, fx = -1, fy = -1, ix = -1, iy = -1;
        for (;; --j) {
            const int4 g = genes[j];
            if ((uint32_t)(g.y - g.x) >= rel && g.y >= end_min && g.x <= start_max) {
                if (n == 0) {
                    fx = g.w;
                    ix = j;
                } else if (n == 1) {
                    fy = g.w;
                    iy = j;
                }
                n += 1;
            }
            if (g.z < end_min) break;  // nothing before j reaches the hit (first gene of a genome: INT32_MIN)
        }
        if (n == 0) continue;
        if (n > 2) {  // (nested genes: counted exactly by stripe_overflow_kernel)
            const uint32_t at = (uint32_t)atomicAdd(&a.stat[2], 1ull);
            if (at < a.overflow_cap) a.overflow[at] = u.first + k;
            continue;
        }
        my_reads += 1u;
        my_pairs += (uint32_t)n;
        if (n == 2 && fy != fx) {  // (two rows of one gene id are one gene: ordinal.py:331-332 builds a set)
            atomicAdd(&c2[ix], 1u);
            atomicAdd(&c2[iy], 1u);
        } else {
            atomicAdd(&c1[ix], 1u);
        }
    }
    const unsigned long long w_reads = wave_sum((unsigned long long)my_reads), w_pairs = wave_sum((unsigned long long)my_pairs);
    if ((threadIdx.x & (kWave - 1)) == 0) {
        atomicAdd(&acc[0], w_reads);
        atomicAdd(&acc[1], w_pairs);
    }
    __syncthreads();
    for (int32_t i = threadIdx.x; i < si.n_genes; i += blockDim.x) {
        const unsigned long long w = (unsigned long long)c1[i] * weight_of(1u) + (unsigned long long)c2[i] * weight_of(2u);
        if (!w) continue;
        const uint32_t feature = (uint32_t)genes[i].w;
        for (int32_t jb = 0; jb < a.n_jobs; ++jb) table_add(a.table, make_key((uint32_t)a.job_index[jb], 0u, (uint32_t)a.group, feature), w);
    }
    if (threadIdx.x == 0) {
        if (acc[0]) atomicAdd(&a.stat[0], acc[0]);
        if (acc[1]) atomicAdd(&a.stat[1], acc[1]);
        // (what wk_get_stats adds up: reads classified, records = matches)
        const uint32_t sb = blockIdx.x % kStripeStatBlocks;
        if (acc[0]) atomicAdd(&a.stat_block[2 * sb], acc[0]);
        if (acc[1]) atomicAdd(&a.stat_block[2 * sb + 1], acc[1]);
    }
}

constexpr size_t kStripeMatchLds = (size_t)kStripeGenes * 16 + (size_t)kStripeGenes * 8 + (size_t)kStripeGenomes * 16 + (size_t)kStripeCells * 2 + 64;

// A hit with more than two genes: a wave collects them all, keeps the distinct features and adds 1/n to each
// (n <= 16: in units of 1/L like everything else; beyond: under the key's k, which classify.counter's
// 1/len(remaining) is).  More than kOverflowMax genes: the error flag (the limit of the count key).
constexpr uint32_t kOverflowMax = 4095;
__global__ void __launch_bounds__(64) stripe_overflow_kernel(StripeMatchArgs a, uint32_t n_over) {
    __shared__ int32_t feat[kOverflowMax + 1];
    __shared__ uint32_t n_feat;
    const uint32_t o = blockIdx.x;
    if (o >= n_over) return;
    const int4 hit = a.binned[a.overflow[o]];
    const int64_t rs = hit.y, re = hit.z, rel = effective_len((uint32_t)hit.w, a.th);
    if (threadIdx.x == 0) {
        // (one lane walks: these hits are a handful per million)
        uint32_t n = 0, total = 0;
        bool too_many = false;
        const int32_t lo = a.gene_off[hit.x], hi = a.gene_off[hit.x + 1];
        const int64_t t = re - rel, min_end = rs + rel;
        int32_t l = lo, h = hi;
        while (l < h) {
            const int32_t m = (l + h) >> 1;
            if ((int64_t)a.gene4[m].x <= t)
                l = m + 1;
            else
                h = m;
        }
        for (int32_t j = l - 1; j >= lo; --j) {
            const int4 g = a.gene4[j];
            const int64_t gs = g.x, ge = g.y;
            const int64_t ov = (ge < re ? ge : re) - (gs > rs ? gs : rs);
            if (ov >= rel) {
                total += 1;
                bool dup = false;
                for (uint32_t z = 0; z < n && !dup; ++z) dup = feat[z] == g.w;
                if (!dup) {
                    if (n < kOverflowMax)
                        feat[n++] = g.w;
                    else
                        too_many = true;
                }
            }
            if ((int64_t)g.z < min_end) break;
        }
        if (too_many) {
            atomicOr(a.table.err, kErrKRange);  // (more distinct genes in a read than the count key's k can say)
            n = 0;
        }
        n_feat = n;
        if (n) {
            atomicAdd(&a.stat[0], 1ull);
            atomicAdd(&a.stat[1], (unsigned long long)total);
            atomicAdd(&a.stat_block[0], 1ull);
            atomicAdd(&a.stat_block[1], (unsigned long long)total);
        }
    }
    __syncthreads();
    const uint32_t n = n_feat;
    for (uint32_t z = threadIdx.x; z < n; z += blockDim.x)
        for (int32_t jb = 0; jb < a.n_jobs; ++jb) {
            if (n <= (uint32_t)WK_WEIGHT_MAX_K)
                table_add(a.table, make_key((uint32_t)a.job_index[jb], 0u, (uint32_t)a.group, (uint32_t)feat[z]), (unsigned long long)weight_of(n));
            else
                table_add(a.table, make_key((uint32_t)a.job_index[jb], n, (uint32_t)a.group, (uint32_t)feat[z]), 1ull);
        }
}

}  // namespace wk
