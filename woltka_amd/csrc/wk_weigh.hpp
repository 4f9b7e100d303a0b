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
// a multiple of 1/L) or a candidate
// without an ancestor at one of the ranks (None entries change k,
// classify.py:167-168) — get a bit in left_mask and are evaluated by the
// generic second pass (classify_kernel<., true, .>) like before.
//
// Layout.  W lives in LDS as 32-bit bins; a workgroup owns one *slice* of the
// subject indices (s mod S) and one share of the reads.  With S slices
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
    uint32_t n_records;    // < 2^30: byte offsets of the buffer loads stay below 2^32 - 16
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
    int* err;
    uint32_t interleave;  // subject s in slice s mod S (else s div bins)
    uint32_t reads_per_wave;  // weigh_stream_kernel: 64, 32 or 16
};

// 16 bytes of subject indices at a 4-byte aligned address
struct __attribute__((packed, aligned(4))) Rec4 {
    uint32_t x, y, z, w;
};
typedef int v4i32 __attribute__((ext_vector_type(4)));

template <bool kAllValid, int kRing = 3, int kAhead = 1>
__global__ void __launch_bounds__(kWeighThreads) weigh_subjects_kernel(WeighArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* bins = reinterpret_cast<uint32_t*>(smem);
    [[maybe_unused]] uint32_t* inval = bins + a.bins;

    const uint32_t xcd = blockIdx.x % a.n_xcd, m = blockIdx.x / a.n_xcd;
    if (m >= a.teams_per_xcd * a.n_slices) return;  // no whole team left on this XCD
    const uint32_t n_slices = a.n_slices;
    const uint32_t slice = m % n_slices;
    const uint32_t team = xcd * a.teams_per_xcd + m / n_slices;
    const uint32_t n_teams = a.n_xcd * a.teams_per_xcd;
    // subject s lives in slice s mod S, bin s div S: indices follow first
    // appearance, so the abundant subjects are dealt evenly over the slices and
    // the workgroups of a team keep the same pace.  s div S = mulhi(s, magic)
    // (exact for s < 2^32 / S).
    const uint32_t magic = n_slices > 1u ? 0xFFFFFFFFu / n_slices + 1u : 0u;
    const uint32_t lo = a.interleave ? 0u : slice * a.bins;

    for (uint32_t i = threadIdx.x; i < a.bins; i += kWeighThreads) bins[i] = 0u;
    if constexpr (!kAllValid)
        for (uint32_t i = threadIdx.x; i < a.invalid_words; i += kWeighThreads) inval[i] = a.invalid[i];
    __syncthreads();

    const uint32_t n_reads = a.n_reads, last = n_reads - 1u;
    const uint32_t n_tiles = (n_reads + kWeighThreads - 1u) / kWeighThreads;
    const char* __restrict__ qoff_b = reinterpret_cast<const char*>(a.qoff);
    struct Stage {
        uint32_t s, e;
        Rec4 v[4];
    };
    // stages addressed by code position (the loop body is unrolled kRing
    // times), as in count_subjects_kernel: offsets of tiles t+4 .. t+3 and
    // records of t+2 .. t+1 are in flight while tile t is added
    Stage ring[kRing];
    auto load_offsets = [&](uint32_t tile, Stage& x) {
        const uint32_t r = tile * kWeighThreads + threadIdx.x;
        const uint32_t off = (r < last ? r : last) << 2;  // clamped: harmless re-read past the end
        x.s = *reinterpret_cast<const uint32_t*>(qoff_b + off);
        x.e = *reinterpret_cast<const uint32_t*>(qoff_b + off + 4u);
    };
    // The records of a read: up to four 16-byte loads, as many as it has.  (The
    // loads sit under divergent branches, so hipcc cannot count the loads in
    // flight and waits for all of them at the next use; issuing all four for
    // every read through a buffer resource whose range check masks the absent
    // pieces keeps the count known, but measured slower: every piece costs
    // address-unit time whether or not it is fetched.  The staging buffer is
    // padded: a piece may run past the last record.)
    const char* __restrict__ subj_b = reinterpret_cast<const char*>(a.subj);
    auto load_records = [&](Stage& x) {
        const uint32_t n = x.e - x.s;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (n > 4u * q) x.v[q] = *reinterpret_cast<const Rec4*>(subj_b + ((x.s + 4u * q) << 2));
    };
    uint32_t my_reads = 0, my_records = 0;
    bool outside = false;  // a subject index beyond the table (reported once, at the end)
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
        // reads the histogram does not cover: more than 16 candidates, or
        // (below) a candidate without an ancestor at one of the ranks
        bool flagged = n > (uint32_t)WK_WEIGHT_MAX_K;
        if constexpr (!kAllValid) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if ((uint32_t)j < n && c[j] < a.n_subjects) flagged |= (inval[c[j] >> 5] >> (c[j] & 31u)) & 1u;
        }
        if (slice == 0u) {
            const unsigned long long mask = __ballot(in & (n > 0u) & flagged);
            if ((threadIdx.x & (kWave - 1)) == 0 && in) a.left_mask[r >> 6] = mask;
        }
        if (!in | flagged | (n == 0u)) return;
        my_reads += 1u;
        my_records += n;
        const uint32_t w = weight_of(n);
        // eight adds at a time, their returned values checked afterwards: one
        // wait per batch instead of one LDS round trip per record
        auto add8 = [&](const uint32_t h, const uint32_t c0, const uint32_t c1, const uint32_t c2, const uint32_t c3,
                        const uint32_t c4, const uint32_t c5, const uint32_t c6, const uint32_t c7) {
            const uint32_t c8[8] = {c0, c1, c2, c3, c4, c5, c6, c7};
            uint32_t old[8];
            uint32_t own_in = 0u;  // bit j: record h + j is this slice's and inside the table
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t cj = c8[j];
                uint32_t q;
                bool own;
                if (a.interleave) {
                    q = n_slices > 1u ? __umulhi(cj, magic) : cj;
                    own = (h + (uint32_t)j < n) & (cj - q * n_slices == slice);
                } else {
                    q = cj - lo;
                    own = (h + (uint32_t)j < n) & (q < a.bins);
                }
                const bool mine = own & (cj < a.n_subjects);
                outside |= own & !mine;
                own_in |= mine ? 1u << j : 0u;
                old[j] = 0u;
                if (mine) old[j] = atomicAdd(&bins[q], w);
            }
            // (the empty asm pins the returned values behind the batch's adds;
            // hipcc otherwise folds each wrap test into its add's branch and
            // waits for every LDS round trip in turn)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(old[j]));
            uint32_t wrapped = 0u;
#pragma unroll
            for (int j = 0; j < 8; ++j) wrapped |= (old[j] + w < old[j]) ? 1u << j : 0u;
            wrapped &= own_in;
            if (wrapped) {  // rare: a 32-bit bin passed 2^32
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if ((wrapped >> j) & 1u) atomicAdd(&a.hi[c8[j]], 1u);
            }
        };
        add8(0u, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
        if (n > 8u) add8(8u, c[8], c[9], c[10], c[11], c[12], c[13], c[14], c[15]);
    };

    uint32_t tile = team;
#pragma unroll
    for (int u = 0; u < kRing - 1; ++u) load_offsets(tile + (uint32_t)u * n_teams, ring[u]);
#pragma unroll
    for (int u = 0; u < kAhead; ++u) load_records(ring[u]);
    // (no test between the steps of a round: a step past the last tile loads
    // clamped addresses and adds nothing, while a conditional step would leave
    // the number of loads in flight unknown to hipcc — vmcnt(0) again)
    while (tile < n_tiles) {
#pragma unroll
        for (int u = 0; u < kRing; ++u) {  // tile t lives in ring[t % kRing]
            load_offsets(tile + (uint32_t)(kRing - 1) * n_teams, ring[(u + kRing - 1) % kRing]);
            load_records(ring[(u + kAhead) % kRing]);
            add_tile(tile, ring[u]);
            tile += n_teams;
        }
    }
    if (outside) atomicOr(a.err, kErrFeatureRange);

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

// ---- the same histogram, one lane per *record* ------------------------------
// weigh_subjects_kernel above spends one lane per read: a wave walks sixteen
// candidate slots for an average of five records, and its vector / scalar
// issue slots — not memory — bound it (~300 VALU + 200 SALU instructions per 64
// reads and slice).  Here a wave takes the records of its 64 (32, 16) reads as
// what they are in memory, one contiguous run, 64 records per row and one per
// lane, so loads are coalesced 256-byte rows and every add runs with all lanes.
// What a record needs from its read is the read's weight L/n:
//   * the read lanes set one bit per read start in a per-wave LDS bit array
//     (position = start - first start of the wave) and store the weights, in
//     read order, in a per-wave table;
//   * for the record at row r, lane l the number of starts at or before its
//     position is  (starts in earlier rows) + mbcnt(row mask): the row's 64-bit
//     start mask sits in SGPRs, v_mbcnt counts its bits below the lane — the
//     record's read is that ordinal, its weight one LDS read.
// Reads of more than 16 records weigh 0 (they go to the generic pass through
// left_mask, like all reads of a wave whose run exceeds the 512 prefetched
// positions), and a terminator start with weight 0 at the end of the run stops
// the last row.  Subjects without an ancestor at some rank need a per-read
// decision that depends on all records of the read: such tables take
// weigh_subjects_kernel<false>.
constexpr uint32_t kStreamRows = 8;                    // rows of 64 positions per chunk
constexpr uint32_t kStreamChunk = 64 * kStreamRows;    // positions covered by the bit array
constexpr uint32_t kStreamScratch = 96;                // dwords of LDS per wave: 16 (bits) + 80 (weights)
constexpr uint32_t kStreamMaxLds = 160 * 1024 - 512;   // dynamic LDS: bins + 16 x scratch

template <int kRing = 5, int kAhead = 2>
__global__ void __launch_bounds__(kWeighThreads) weigh_stream_kernel(WeighArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* bins = reinterpret_cast<uint32_t*>(smem);
    // (readfirstlane: hipcc does not know that threadIdx.x >> 6 is the same in
    // all lanes, and would keep everything derived from it — run lengths, row
    // counts, the buffer resource — in vector registers, with vector compares
    // for the row tests and waterfall loops around the buffer loads)
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t* wbits = bins + a.bins + wave * kStreamScratch;  // [16]
    uint32_t* wtab = wbits + 16;                                // [80]

    const uint32_t xcd = blockIdx.x % a.n_xcd, m = blockIdx.x / a.n_xcd;
    if (m >= a.teams_per_xcd * a.n_slices) return;  // no whole team left on this XCD
    const uint32_t slice = m % a.n_slices;
    const uint32_t team = xcd * a.teams_per_xcd + m / a.n_slices;
    const uint32_t n_teams = a.n_xcd * a.teams_per_xcd;
    const uint32_t lo = slice * a.bins;  // slice = bins consecutive subject indices
    const uint32_t span = min(a.bins, a.n_subjects - min(lo, a.n_subjects));
    const bool last_slice = slice + 1u == a.n_slices;

    for (uint32_t i = threadIdx.x; i < a.bins; i += kWeighThreads) bins[i] = 0u;
    __syncthreads();

    const uint32_t n_reads = a.n_reads, last = n_reads - 1u;
    const uint32_t rpw = a.reads_per_wave;          // reads of a wave per tile: 64, 32 or 16
    const uint32_t tile_reads = rpw * (kWeighThreads / kWave);
    const uint32_t n_tiles = (n_reads + tile_reads - 1u) / tile_reads;
    const uint32_t mask_reads = (n_reads + 63u) & ~63u;  // reads the words of left_mask cover
    const char* __restrict__ qoff_b = reinterpret_cast<const char*>(a.qoff);
    struct Stage {
        uint32_t s, e;
        uint32_t rec[kStreamRows];
    };
    Stage ring[kRing];
    auto load_offsets = [&](uint32_t tile, Stage& x) {
        const uint32_t r = tile * tile_reads + wave * rpw + lane;
        const uint32_t off = (r < last ? r : last) << 2;  // clamped: harmless re-read past the end
        x.s = *reinterpret_cast<const uint32_t*>(qoff_b + off);
        x.e = *reinterpret_cast<const uint32_t*>(qoff_b + off + 4u);
    };
    // first record and number of records of the wave's reads in a tile
    auto run_of = [&](uint32_t tile, const Stage& x, uint32_t& base, uint32_t& len) {
        const uint32_t r0 = tile * tile_reads + wave * rpw;
        // (readfirstlane keeps these in scalar registers, see `wave` above)
        const uint32_t nl = __builtin_amdgcn_readfirstlane(r0 < n_reads ? min(rpw, n_reads - r0) : 0u);  // reads of this wave inside the chunk
        base = __builtin_amdgcn_readfirstlane(x.s);
        const uint32_t end = __builtin_amdgcn_readlane(x.e, (int)(nl ? nl - 1u : 0u));
        len = __builtin_amdgcn_readfirstlane(nl ? end - base : 0u);
        return nl;
    };
    // the first kStreamChunk records of the run, one row of 64 per load: always
    // kStreamRows loads (a known number in flight, see weigh_subjects_kernel).
    // The buffer resource is rebuilt per run in scalar registers — base = first
    // record, size = the run — so a lane's offset is 4 * lane for every run, the
    // row is the instruction's scalar offset, and positions past the run fail the
    // resource's range check: zeros, no memory request, no address arithmetic.
    const uint32_t lane4 = lane << 2;
    auto load_records = [&](uint32_t tile, Stage& x) {
        uint32_t base, len;
        run_of(tile, x, base, len);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<int32_t*>(a.subj) + base, 0, (int)(min(len, kStreamChunk) << 2), 0x00020000);
#pragma unroll
        for (uint32_t row = 0; row < kStreamRows; ++row)
            x.rec[row] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)lane4, (int)(row * 256u), 0);
    };
    uint32_t my_reads = 0, my_records = 0;
    bool outside = false;  // a subject index beyond the table (reported once, at the end)
    auto add_tile = [&](uint32_t tile, const Stage& x) {
        uint32_t base, len;
        const uint32_t nl = run_of(tile, x, base, len);
        const uint32_t r0 = tile * tile_reads + wave * rpw;
        const uint32_t n = lane < nl ? x.e - x.s : 0u;
        const bool lng = n > (uint32_t)WK_WEIGHT_MAX_K;
        // a run longer than the prefetched rows (rare: the host picks the reads
        // per wave from the mean hits per read) is left to the generic pass as a
        // whole — loading its tail here would be a load of unknown issue count
        const bool over = len > kStreamChunk;
        if (slice == 0u && r0 < mask_reads) {
            // this wave's rpw bits of left_mask: reads the histogram does not cover
            const unsigned long long left = __ballot(lng | (over & (n > 0u)));
            unsigned char* dst = reinterpret_cast<unsigned char*>(a.left_mask) + (r0 >> 3);
            if (lane == 0) {
                if (rpw == 64u) *reinterpret_cast<unsigned long long*>(dst) = left;
                else if (rpw == 32u) *reinterpret_cast<uint32_t*>(dst) = (uint32_t)left;
                else *reinterpret_cast<uint16_t*>(dst) = (uint16_t)left;
            }
        }
        if (len == 0u || over) return;  // (wave-uniform)
        my_reads += ((n > 0u) & !lng) ? 1u : 0u;
        my_records += lng ? 0u : n;
        const unsigned long long nonempty = __ballot(n > 0u);
        const uint32_t ord = __builtin_amdgcn_mbcnt_hi((uint32_t)(nonempty >> 32),
                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)nonempty, 0u));
        // weights in read order, then one entry of weight 0 for the terminator
        if (n > 0u) wtab[ord] = lng ? 0u : weight_of(n);
        if (lane == 0) wtab[__builtin_popcountll(nonempty)] = 0u;
        const uint32_t start = x.s - base;
        // one chunk of up to 512 positions starting at c0, its records in c[]
        auto chunk = [&](const uint32_t c0, const uint32_t (&c)[kStreamRows]) {
            if (lane < 16u) wbits[lane] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            const uint32_t rel = start - c0;
            if (n > 0u && rel < kStreamChunk) atomicOr(&wbits[rel >> 5], 1u << (rel & 31u));
            if (lane == 0 && len - c0 < kStreamChunk) atomicOr(&wbits[(len - c0) >> 5], 1u << ((len - c0) & 31u));
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const uint32_t mybits = lane < 16u ? wbits[lane] : 0u;
            // reads that start before this chunk
            uint32_t running = c0 ? (uint32_t)__builtin_popcountll(__ballot(n > 0u && start < c0)) : 0u;
            const uint32_t n_rows = min(kStreamRows, (len - c0 + 63u) >> 6);
            uint32_t w[kStreamRows];
#pragma unroll
            for (uint32_t row = 0; row < kStreamRows; ++row) {
                w[row] = 0u;
                if (row < n_rows) {  // (wave-uniform)
                    const uint32_t mlo = __builtin_amdgcn_readlane(mybits, (int)(2u * row));
                    const uint32_t mhi = __builtin_amdgcn_readlane(mybits, (int)(2u * row + 1u));
                    // starts at or before this lane's position: bit 0 of the row
                    // + the bits 1 .. lane, i.e. mbcnt of the mask shifted down by one
                    const unsigned long long ms = (((unsigned long long)mhi << 32) | mlo) >> 1;
                    const uint32_t cnt = __builtin_amdgcn_mbcnt_hi(
                        (uint32_t)(ms >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ms, running + (mlo & 1u)));
                    w[row] = wtab[cnt - 1u];
                    running += (uint32_t)__builtin_popcount(mlo) + (uint32_t)__builtin_popcount(mhi);
                }
            }
            uint32_t old[kStreamRows];
            uint32_t own = 0u;
#pragma unroll
            for (uint32_t row = 0; row < kStreamRows; ++row) {
                old[row] = 0u;
                if (row < n_rows) {
                    const uint32_t idx = c[row] - lo;
                    const bool mine = (w[row] != 0u) & (idx < span);
                    if (last_slice) outside |= (w[row] != 0u) & (c[row] >= a.n_subjects);
                    own |= mine ? 1u << row : 0u;
                    if (mine) old[row] = atomicAdd(&bins[idx], w[row]);
                }
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (uint32_t row = 0; row < kStreamRows; ++row) asm volatile("" : "+v"(old[row]));
            uint32_t wrapped = 0u;
#pragma unroll
            for (uint32_t row = 0; row < kStreamRows; ++row) wrapped |= (old[row] + w[row] < old[row]) ? 1u << row : 0u;
            wrapped &= own;
            if (wrapped) {  // rare: a 32-bit bin passed 2^32
#pragma unroll
                for (uint32_t row = 0; row < kStreamRows; ++row)
                    if ((wrapped >> row) & 1u) atomicAdd(&a.hi[c[row]], 1u);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        };
        chunk(0u, x.rec);
    };

    uint32_t tile = team;
#pragma unroll
    for (int u = 0; u < kRing - 1; ++u) load_offsets(tile + (uint32_t)u * n_teams, ring[u]);
#pragma unroll
    for (int u = 0; u < kAhead; ++u) load_records(tile + (uint32_t)u * n_teams, ring[u]);
    while (tile < n_tiles) {
#pragma unroll
        for (int u = 0; u < kRing; ++u) {  // tile t lives in ring[t % kRing]; no test between the steps
            load_offsets(tile + (uint32_t)(kRing - 1) * n_teams, ring[(u + kRing - 1) % kRing]);
            load_records(tile + (uint32_t)kAhead * n_teams, ring[(u + kAhead) % kRing]);
            add_tile(tile, ring[u]);
            tile += n_teams;
        }
    }
    if (outside) atomicOr(a.err, kErrFeatureRange);

    {
        __shared__ unsigned long long acc[2];
        if (threadIdx.x == 0) acc[0] = acc[1] = 0ull;
        __syncthreads();
        unsigned long long rd = wave_sum(slice == 0u ? my_reads : 0u);
        unsigned long long rc = wave_sum(slice == 0u ? my_records : 0u);
        if (lane == 0) {
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
    uint32_t n_subjects, bins, n_teams, n_slices, interleave;
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
        const uint32_t colm = a.interleave ? i / a.n_slices : i % a.bins;
        const uint32_t slice = a.interleave ? i - colm * a.n_slices : i / a.bins;
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
