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
// classify.py:167-168) — get a bit in a mask and are evaluated by the
// generic second pass (classify_kernel<., true, .>) like before.
//
// Layout.  W lives in LDS as 32-bit bins; a workgroup owns one *slice* of
// consecutive subject indices and one share of the records.  With S slices
// the S workgroups of a *team* walk the same tiles of records, each adding only
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
typedef int v4i32 __attribute__((ext_vector_type(4)));

// ---- records that carry their read's size --------------------------------------
// What a record needs from its read is one number, the read's size k (its
// weight is L/k).  It is derived once per staged chunk (read_sizes_kernel:
// one byte per record, 0 for the records of reads the
// histogram does not cover; one bit per such read in a mask for the generic
// pass; done by
// wk_chunk_stage, and again by the first classify call when the subject table
// has subjects without an ancestor at a requested rank), so that the histogram itself is a plain stream over records: no read
// offsets, no per-read logic, every lane busy — a 16-byte load of four subject
// indices, a 4-byte load of their four sizes, a weight lookup and an LDS add
// per record.
// One thread per read, a workgroup per tile of kSizeTile consecutive reads:
// the tile's records are one contiguous range, so every thread writes its
// read's size into an LDS image of that range (zero for reads that are
// not covered) and the workgroup stores the image with whole dwords — the
// bytes of a dword shared with a neighbouring tile one by one.  Reads that do
// not fit the image (a tile with very long reads) are written straight to HBM
// by their thread.  One bit per read not covered goes to the mask.
constexpr uint32_t kSizeTile = 1024;                              // reads per tile = threads per workgroup
constexpr uint32_t kSizeImage = kSizeTile * WK_WEIGHT_MAX_K + 8;  // bytes of records a tile of covered reads can span (+ alignment)
template <bool kCheck>
__global__ void __launch_bounds__(kSizeTile) read_sizes_kernel(const int32_t* __restrict__ qoff, uint32_t n_reads,
                                                              const int32_t* __restrict__ subj,
                                                              const uint32_t* __restrict__ invalid, uint32_t n_subjects,
                                                              unsigned char* __restrict__ rk,
                                                              unsigned long long* __restrict__ left_mask,
                                                              unsigned long long* __restrict__ totals) {
    __shared__ __attribute__((aligned(16))) unsigned char image[kSizeImage];
    __shared__ unsigned long long acc[2];
    if (threadIdx.x == 0) acc[0] = acc[1] = 0ull;
    unsigned long long rd = 0, rc = 0;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t n_tiles = (n_reads + kSizeTile - 1) / kSizeTile;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t r0 = tile * kSizeTile, r1 = min(r0 + kSizeTile, n_reads);
        const uint32_t g0 = (uint32_t)qoff[r0], g1 = (uint32_t)qoff[r1];  // the tile's records
        const uint32_t shift = g0 & 3u;  // image[shift + i] <-> rk[g0 + i]: dwords of the image are dwords of rk
        for (uint32_t i = threadIdx.x; i < kSizeImage / 4u; i += kSizeTile) reinterpret_cast<uint32_t*>(image)[i] = 0u;
        __syncthreads();
        const uint32_t r = r0 + threadIdx.x;
        const bool in = r < r1;
        uint32_t s = g0, n = 0;
        if (in) {
            s = (uint32_t)qoff[r];
            n = (uint32_t)qoff[r + 1] - s;
        }
        // not covered: more than 16 candidates (1/k is not a multiple of 1/L), or —
        // kCheck: some subject of the table has the bit — a candidate without an
        // ancestor at one of the ranks (None entries change k, classify.py:167-168)
        bool skip = n > (uint32_t)WK_WEIGHT_MAX_K;
        if constexpr (kCheck) {
            if (!skip)
                for (uint32_t j = 0; j < n; ++j) {
                    const uint32_t c = (uint32_t)subj[s + j];
                    if (c < n_subjects) skip |= (invalid[c >> 5] >> (c & 31u)) & 1u;
                }
        }
        const unsigned long long left = __ballot(in & skip);
        if (lane == 0 && in) left_mask[r >> 6] = left;
        const unsigned char size = skip ? (unsigned char)0 : (unsigned char)n;
        const uint32_t at = shift + (s - g0);
        if (at + n <= kSizeImage) {
            if (size)
                for (uint32_t j = 0; j < n; ++j) image[at + j] = size;
        } else {  // (rare: a very long read, or a read behind one — what is past the image goes straight to HBM)
            for (uint32_t j = 0; j < n; ++j) {
                if (at + j < kSizeImage)
                    image[at + j] = size;
                else
                    rk[s + j] = size;
            }
        }
        // reads and records the histogram covers (statistics of the classify calls)
        rd += (in & !skip & (n > 0u)) ? 1ull : 0ull;
        rc += (in & !skip) ? (unsigned long long)n : 0ull;
        __syncthreads();
        // image -> rk: dwords that lie inside [g0, g1) whole, the others byte by byte
        const uint32_t span = min(g1 - g0, kSizeImage - shift);
        const uint32_t first = g0 - shift;  // (a multiple of 4)
        for (uint32_t i = threadIdx.x; 4u * i < shift + span; i += kSizeTile) {
            const uint32_t lo = first + 4u * i;
            if (lo >= g0 && lo + 4u <= g0 + span) {
                *reinterpret_cast<uint32_t*>(rk + lo) = reinterpret_cast<const uint32_t*>(image)[i];
            } else {
                for (uint32_t b = 0; b < 4u; ++b)
                    if (lo + b >= g0 && lo + b < g0 + span) rk[lo + b] = image[4u * i + b];
            }
        }
        __syncthreads();
    }
    rd = wave_sum(rd);
    rc = wave_sum(rc);
    if (lane == 0 && (rd | rc)) {
        atomicAdd(&acc[0], rd);
        atomicAdd(&acc[1], rc);
    }
    __syncthreads();
    if (threadIdx.x == 0 && (acc[0] | acc[1])) {
        atomicAdd(&totals[0], acc[0]);
        atomicAdd(&totals[1], acc[1]);
    }
}

// Packed records (what the native tokenizer emits, wk_tok_fetch_packed):
// subject index | position of the record in its read << 23 | size of its read
// << 27 (size 0: a read of more than WK_WEIGHT_MAX_K records — such reads are
// staged the other way).  One word per record is all the histogram streams.
constexpr uint32_t kWordSubjBits = 23;
constexpr uint32_t kWordSubjMask = (1u << kWordSubjBits) - 1u;
constexpr uint32_t kWordSizeShift = 27;

struct BinsArgs {
    const int32_t* subj;        // [n_records] subject indices, or packed records (kPacked)
    const unsigned char* rk;    // [n_records] read size 1..16, or 0: not covered (not kPacked)
    uint32_t n_records;
    uint32_t n_subjects;
    uint32_t bins, n_slices, teams_per_xcd, n_xcd;
    uint32_t* slab;             // [n_slices][n_teams][bins]
    uint32_t* hi;               // [n_subjects]
    int* err;
};

constexpr uint32_t kBinsTile = kWeighThreads * 4;  // records per workgroup and round
constexpr uint32_t kBinsMaxLds = 160 * 1024 - 1024;

// ---- records kept per slice of the subject table (round 4) -------------------------
// A subject table beyond the LDS is cut into slices of kSliceBins subjects.  Round
// 3's launch had a team of workgroups, one per slice, walk the same records: every
// record crossed the memory system once per slice (1.57 GB measured for 1.0 GB of
// records at three slices).  Now the records are kept apart from the start: whoever
// appends them — dtok_emit_kernel on the device, words_partition_kernel behind a
// copy from the host — writes a record to the stream of its subject's slice, and the
// histogram reads every stream exactly once, with workgroups in proportion to the
// streams' lengths (a Zipf sample has nine tenths of its records in slice 0).
constexpr int kMaxStreams = 8;
constexpr uint32_t kSliceBins = kBinsMaxLds / 4 - 96;

struct StreamSet {
    uint32_t* out[kMaxStreams];
    unsigned long long* cursor;  // [kMaxStreams] records in each stream
    uint32_t n_streams;
    uint32_t cap;                // records a stream holds
};

// Every thread of the workgroup calls this: the records (`rec`, `word`) go to the
// streams of their slices, one reservation per workgroup and slice.
template <uint32_t kThreads>
__device__ __forceinline__ void scatter_by_slice(const StreamSet& s, bool rec, uint32_t word) {
    __shared__ uint32_t cnt[kThreads / kWave][kMaxStreams];
    __shared__ unsigned long long base[kMaxStreams];
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t sl = 0;
    if (rec) {
        sl = (word & ((1u << 23) - 1u)) / kSliceBins;
        if (sl >= s.n_streams) sl = s.n_streams - 1u;  // (a subject beyond the table: the histogram reports it)
    }
    unsigned long long mine = 0;
    for (uint32_t k = 0; k < s.n_streams; ++k) {
        const unsigned long long m = __ballot(rec && sl == k);
        if (lane == 0) cnt[wave][k] = (uint32_t)__popcll(m);
        if (sl == k) mine = m;
    }
    __syncthreads();
    if (threadIdx.x < s.n_streams) {
        const uint32_t k = threadIdx.x;
        uint32_t n = 0;
        for (uint32_t w = 0; w < kThreads / kWave; ++w) {
            const uint32_t c = cnt[w][k];
            cnt[w][k] = n;
            n += c;
        }
        base[k] = n ? atomicAdd(&s.cursor[k], (unsigned long long)n) : 0ull;
    }
    __syncthreads();
    if (rec) {
        const unsigned long long at = base[sl] + cnt[wave][sl] + (unsigned long long)__popcll(mine & ((1ull << lane) - 1ull));
        if (at < s.cap) s.out[sl][at] = word;
    }
}

// The same for kItems records per thread: ONE reservation per workgroup and slice for all of them.  (A
// returning atomic on one word completes about 90 times per microsecond; with a reservation per 256
// records the cursors of the streams were the whole cost of appending records: 160 us of a 64 MB
// block's emission, 1.3 ms per 28 M host words -- profiles/r05a_e2e_lca, r04_lca_kernel_stats.)
constexpr uint32_t kScatterItems = 8;
template <uint32_t kThreads, uint32_t kItems>
__device__ __forceinline__ void scatter_by_slice_n(const StreamSet& s, const bool (&rec)[kItems], const uint32_t (&word)[kItems]) {
    __shared__ uint32_t cnt[kItems][kThreads / kWave][kMaxStreams];
    __shared__ unsigned long long base[kMaxStreams];
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t sl[kItems];
    unsigned long long mine[kItems];
#pragma unroll
    for (uint32_t r = 0; r < kItems; ++r) {
        sl[r] = 0;
        if (rec[r]) {
            sl[r] = (word[r] & ((1u << 23) - 1u)) / kSliceBins;
            if (sl[r] >= s.n_streams) sl[r] = s.n_streams - 1u;  // (a subject beyond the table: the histogram reports it)
        }
        mine[r] = 0;
        for (uint32_t k = 0; k < s.n_streams; ++k) {
            const unsigned long long m = __ballot(rec[r] && sl[r] == k);
            if (lane == 0) cnt[r][wave][k] = (uint32_t)__popcll(m);
            if (sl[r] == k) mine[r] = m;
        }
    }
    __syncthreads();
    if (threadIdx.x < s.n_streams) {
        const uint32_t k = threadIdx.x;
        uint32_t n = 0;
        for (uint32_t r = 0; r < kItems; ++r)
            for (uint32_t w = 0; w < kThreads / kWave; ++w) {
                const uint32_t c = cnt[r][w][k];
                cnt[r][w][k] = n;
                n += c;
            }
        base[k] = n ? atomicAdd(&s.cursor[k], (unsigned long long)n) : 0ull;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < kItems; ++r)
        if (rec[r]) {
            const unsigned long long at = base[sl[r]] + cnt[r][wave][sl[r]] + (unsigned long long)__popcll(mine[r] & ((1ull << lane) - 1ull));
            if (at < s.cap) s.out[sl[r]][at] = word[r];
        }
}

// records copied from the host -> the streams (kScatterItems x 256 per workgroup)
__global__ void __launch_bounds__(256) words_partition_kernel(const uint32_t* __restrict__ src, uint32_t n, StreamSet s) {
    const uint32_t first = blockIdx.x * (256u * kScatterItems) + threadIdx.x;
    bool rec[kScatterItems];
    uint32_t word[kScatterItems];
#pragma unroll
    for (uint32_t r = 0; r < kScatterItems; ++r) {
        const uint32_t i = first + r * 256u;
        rec[r] = i < n;
        word[r] = rec[r] ? src[i] : 0u;
    }
    scatter_by_slice_n<256, kScatterItems>(s, rec, word);
}

struct StreamBinsArgs {
    const uint32_t* words[kMaxStreams];
    uint32_t count[kMaxStreams];
    uint32_t wg_first[kMaxStreams + 1];  // workgroups [wg_first[k], wg_first[k + 1]) walk stream k
    uint32_t n_streams;
    uint32_t n_subjects;
    uint32_t* slab;  // [workgroups][kSliceBins]
    uint32_t* hi;    // [n_subjects]
    int* err;
};

template <int kRing = 4, bool kPacked = false>
__global__ void __launch_bounds__(kWeighThreads) weigh_bins_kernel(BinsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: 32 weights (lut[k] = L / k, lut[0] = 0) in static LDS — their
    // address is a constant of the instruction, no base register to add —,
    // the bins, 64 idle bins behind them for the lanes without a record of
    // this slice
    __shared__ uint32_t lut[32];
    uint32_t* bins = reinterpret_cast<uint32_t*>(smem);

    const uint32_t xcd = blockIdx.x % a.n_xcd, m = blockIdx.x / a.n_xcd;
    if (m >= a.teams_per_xcd * a.n_slices) return;  // no whole team left on this XCD
    const uint32_t slice = m % a.n_slices;
    const uint32_t team = xcd * a.teams_per_xcd + m / a.n_slices;
    const uint32_t n_teams = a.n_xcd * a.teams_per_xcd;
    const uint32_t lo = slice * a.bins;
    const uint32_t span = min(a.bins, a.n_subjects - min(lo, a.n_subjects));
    const bool last_slice = slice + 1u == a.n_slices;

    for (uint32_t i = threadIdx.x; i < a.bins + 64u; i += kWeighThreads) bins[i] = 0u;
    if (threadIdx.x < 32u)
        lut[threadIdx.x] = (threadIdx.x >= 1u && threadIdx.x <= (uint32_t)WK_WEIGHT_MAX_K) ? weight_of(threadIdx.x) : 0u;
    __syncthreads();

    // both streams through buffer resources: the range check returns zeros past
    // the end (size 0 = not covered), so the last round needs no special case
    const __amdgpu_buffer_rsrc_t subj_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.subj), 0, (int)(a.n_records << 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rk_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.rk), 0, (int)((a.n_records + 3u) & ~3u), 0x00020000);  // (whole dwords: the pad bytes are zero)
    const uint32_t n_tiles = (a.n_records + kBinsTile - 1u) / kBinsTile;
    struct Stage {
        v4i32 c;
        uint32_t k4;
    };
    Stage ring[kRing];
    auto load = [&](uint32_t tile, Stage& x) {
        const uint32_t i = tile * kBinsTile + threadIdx.x * 4u;  // (past the end: offsets beyond the buffers)
        const bool ok = tile < n_tiles;
        x.c = __builtin_amdgcn_raw_buffer_load_b128(subj_rsrc, (int)(ok ? i << 2 : 0xFFFFFFF0u), 0, 0);
        if constexpr (!kPacked) x.k4 = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rk_rsrc, (int)(ok ? i : 0xFFFFFFF0u), 0, 0);
    };
    const uint32_t idle_addr = a.bins + (threadIdx.x & 63u);
    bool outside = false;
    auto add = [&](const Stage& x) {
        uint32_t c[4] = {(uint32_t)x.c.x, (uint32_t)x.c.y, (uint32_t)x.c.z, (uint32_t)x.c.w};
        uint32_t w[4], at[4], old[4];
        bool in[4];
        if constexpr (kPacked) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                w[j] = lut[c[j] >> kWordSizeShift];  // (records past the end load as 0: size 0, weight 0)
                c[j] &= kWordSubjMask;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = lut[(x.k4 >> (8 * j)) & 31u];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t idx = c[j] - lo;
            const bool mine = idx < span;
            if (last_slice) outside |= (w[j] != 0u) & (c[j] >= a.n_subjects);
            // a lane without a record of this slice adds to an idle bin of its
            // own (what piles up there is never read): every lane takes part in
            // every add, so the number of LDS operations in flight is known to
            // hipcc and it waits for the oldest ones only, instead of draining
            // the LDS queue at every use
            at[j] = mine ? idx : idle_addr;
            in[j] = mine;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) old[j] = atomicAdd(&bins[at[j]], w[j]);
        bool wrapped = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) wrapped |= in[j] & (old[j] + w[j] < old[j]);
        if (wrapped) {  // rare: a 32-bit bin passed 2^32
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (in[j] & (old[j] + w[j] < old[j])) atomicAdd(&a.hi[c[j]], 1u);
        }
    };

    uint32_t tile = team;
#pragma unroll
    for (int u = 0; u < kRing - 1; ++u) load(tile + (uint32_t)u * n_teams, ring[u]);
    while (tile < n_tiles) {
#pragma unroll
        for (int u = 0; u < kRing; ++u) {  // stages addressed by code position; steps past the end add nothing
            load(tile + (uint32_t)(kRing - 1) * n_teams, ring[(u + kRing - 1) % kRing]);
            add(ring[u]);
            tile += n_teams;
        }
    }
    if (outside) atomicOr(a.err, kErrFeatureRange);
    __syncthreads();
    uint32_t* row = a.slab + ((size_t)slice * n_teams + team) * a.bins;
    for (uint32_t i = threadIdx.x; i < a.bins; i += kWeighThreads) row[i] = bins[i];
}

// The histogram over the streams: a workgroup owns the bins of its stream's slice
// and a share of that stream's tiles.  Same inner loop as weigh_bins_kernel (every
// record of the stream lies in the slice; the idle bins take the lanes past its end).
template <int kRing = 4>
__global__ void __launch_bounds__(kWeighThreads) weigh_streams_kernel(StreamBinsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t lut[32];
    uint32_t* bins = reinterpret_cast<uint32_t*>(smem);
    uint32_t k = 0;
    while (k + 1u < a.n_streams && blockIdx.x >= a.wg_first[k + 1]) ++k;
    const uint32_t team = blockIdx.x - a.wg_first[k], n_teams = a.wg_first[k + 1] - a.wg_first[k];
    const uint32_t lo = k * kSliceBins;
    const uint32_t span = min(kSliceBins, a.n_subjects - min(lo, a.n_subjects));
    const uint32_t n_records = a.count[k];

    for (uint32_t i = threadIdx.x; i < kSliceBins + 64u; i += kWeighThreads) bins[i] = 0u;
    if (threadIdx.x < 32u)
        lut[threadIdx.x] = (threadIdx.x >= 1u && threadIdx.x <= (uint32_t)WK_WEIGHT_MAX_K) ? weight_of(threadIdx.x) : 0u;
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.words[k]), 0, (int)(n_records << 2), 0x00020000);
    const uint32_t n_tiles = (n_records + kBinsTile - 1u) / kBinsTile;
    v4i32 ring[kRing];
    auto load = [&](uint32_t tile, v4i32& x) {
        const uint32_t i = tile * kBinsTile + threadIdx.x * 4u;
        x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(tile < n_tiles ? i << 2 : 0xFFFFFFF0u), 0, 0);
    };
    const uint32_t idle_addr = kSliceBins + (threadIdx.x & 63u);
    bool outside = false;
    auto add = [&](const v4i32& x) {
        uint32_t c[4] = {(uint32_t)x.x, (uint32_t)x.y, (uint32_t)x.z, (uint32_t)x.w};
        uint32_t w[4], at[4], old[4];
        bool in[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w[j] = lut[c[j] >> kWordSizeShift];  // (past the end: 0, weight 0)
            c[j] &= kWordSubjMask;
            const uint32_t idx = c[j] - lo;
            in[j] = idx < span;
            outside |= (w[j] != 0u) & !in[j];
            at[j] = in[j] ? idx : idle_addr;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) old[j] = atomicAdd(&bins[at[j]], w[j]);
        bool wrapped = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) wrapped |= in[j] & (old[j] + w[j] < old[j]);
        if (wrapped) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (in[j] & (old[j] + w[j] < old[j])) atomicAdd(&a.hi[c[j]], 1u);
        }
    };
    uint32_t tile = team;
#pragma unroll
    for (int u = 0; u < kRing - 1; ++u) load(tile + (uint32_t)u * n_teams, ring[u]);
    while (tile < n_tiles) {
#pragma unroll
        for (int u = 0; u < kRing; ++u) {
            load(tile + (uint32_t)(kRing - 1) * n_teams, ring[(u + kRing - 1) % kRing]);
            add(ring[u]);
            tile += n_teams;
        }
    }
    if (outside) atomicOr(a.err, kErrFeatureRange);
    __syncthreads();
    uint32_t* row = a.slab + (size_t)blockIdx.x * kSliceBins;
    for (uint32_t i = threadIdx.x; i < span; i += kWeighThreads) row[i] = bins[i];
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
    uint32_t streams;                    // > 0: the slab of weigh_streams_kernel (rows of kSliceBins, per stream wg_first)
    uint32_t wg_first[kMaxStreams + 1];
};

// 128 subjects per workgroup, eight threads each: a thread sums every eighth row of
// its subject's column (a column of 230 rows read by one thread was latency, not
// bandwidth: 0.084 ms for 42 MB), the eight partial sums meet in LDS.
constexpr uint32_t kMergeSubjects = 128, kMergeParts = 8;
__global__ void __launch_bounds__(1024) weigh_merge_kernel(WeighMergeArgs a, uint32_t lds_slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long part[kMergeParts][kMergeSubjects];
    LdsCache cache{};
    cache.base = reinterpret_cast<unsigned long long*>(smem);
    cache.bmask = lds_slots / 4 - 1;
    lds_cache_init(cache);
    const uint32_t sub = threadIdx.x & (kMergeSubjects - 1u), g = threadIdx.x / kMergeSubjects;
    const uint32_t i = blockIdx.x * kMergeSubjects + sub;
    unsigned long long w = 0;
    if (i < a.n_subjects) {
        if (a.streams) {
            const uint32_t k = i / kSliceBins, colm = i - k * kSliceBins;
            const uint32_t* p = a.slab + (size_t)a.wg_first[k] * kSliceBins + colm;
            const uint32_t rows = a.wg_first[k + 1] - a.wg_first[k];
#pragma unroll 4
            for (uint32_t t = g; t < rows; t += kMergeParts) w += p[(size_t)t * kSliceBins];
        } else {
            const uint32_t slice = i / a.bins, colm = i - slice * a.bins;
            const uint32_t* p = a.slab + (size_t)slice * a.n_teams * a.bins + colm;
#pragma unroll 4
            for (uint32_t t = g; t < a.n_teams; t += kMergeParts) w += p[(size_t)t * a.bins];
        }
    }
    part[g][sub] = w;
    __syncthreads();
    if (g == 0 && i < a.n_subjects) {
#pragma unroll
        for (uint32_t q = 1; q < kMergeParts; ++q) w += part[q][sub];
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
