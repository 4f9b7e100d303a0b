// wk_readmap.hpp — the read maps of `--outmap`, formatted on the device.
//
// file.write_readmap (woltka/file.py:469-500) prints one line per classified
// read: `query <tab> taxon`, or, when the read's candidates fall on several
// taxa, `query <tab> taxon:count <tab> taxon:count ...` sorted by descending
// count, then by taxon id (the sort runs on the ids; --name-as-id replaces
// them by names afterwards).  With the text tokenised on the device
// (wk_dtok.hpp) everything such a line needs is already there: the QNAME's
// place in the block, the mate (query, query/1, query/2: align.py:327-332), and
// the read's subjects.  Here the lines are built next to them and only the
// finished text crosses to the host, which compresses it.
//
// Scope: the plain assigners (classify.assign_none without --uniq,
// classify.assign_rank without --uniq / --major / --above; classify.py:32-51,
// 81-127) over subjects that all have a taxon at the rank — the job sets the
// weighted histogram takes (wk_weigh.hpp); then a read's result is the taxon
// its subjects share, or the list of their taxa.  The host supplies, per job,
// for every subject the *slot* of its taxon in a compact table of the taxa in
// use, each slot's place in the order of the id strings, and the text shown
// for it.
//
// Kernels: readmap_len (a thread per line; the leader line of a read computes
// the read's index and the length of its map line), a scan of the lengths over
// the reads, readmap_write (a thread per read: the bytes).
#pragma once
#include "wk_dtok.hpp"

namespace wk {

struct ReadmapArgs {
    const int32_t* slot_of_subject;  // [n_subjects]
    uint32_t n_subjects;
    const int32_t* slot_order;       // [n_slots] rank of the slot's id string among the slots'
    const uint32_t* shown_off;       // [n_slots + 1]
    const unsigned char* shown;      // text printed per slot
    uint32_t* read_line;             // [n_reads] leader line of the read
    unsigned long long* read_len;    // [n_reads] length of its map line -> exclusive prefix
    unsigned char* out;
    unsigned long long out_cap;
    uint32_t n_reads;
};

__device__ __forceinline__ uint32_t dec_digits(uint32_t v) {
    uint32_t d = 1;
    while (v >= 10u) {
        v /= 10u;
        ++d;
    }
    return d;
}

// The taxa of the read whose leader is line i: distinct slots with counts,
// sorted by (-count, order of the id string).  Returns the number of distinct
// slots (0: a subject outside the table — cannot happen for an accepted block).
__device__ __forceinline__ int readmap_taxa(const DtokArgs& a, const ReadmapArgs& m, uint32_t i, int32_t* slot, uint32_t* count) {
    const uint32_t mate = a.lmeta[i] >> 28;
    int n = 0;
    for (uint32_t j = i; j < a.n_lines; ++j) {
        if (j > i && a.is_start[j]) break;
        if (!(a.is_first[j] & 1u) || (a.lmeta[j] >> 28) != mate) continue;
        const uint32_t s = (uint32_t)a.lsubj[j];
        if (s >= m.n_subjects) return 0;
        const int32_t t = m.slot_of_subject[s];
        int k = 0;
        while (k < n && slot[k] != t) ++k;
        if (k == n) {
            if (n == WK_WEIGHT_MAX_K) return 0;
            slot[n] = t;
            count[n] = 0;
            ++n;
        }
        ++count[k];
    }
    // insertion sort: descending count, then ascending order of the id strings
    for (int x = 1; x < n; ++x) {
        const int32_t t = slot[x];
        const uint32_t c = count[x];
        const int32_t o = m.slot_order[t];
        int y = x - 1;
        while (y >= 0 && (count[y] < c || (count[y] == c && m.slot_order[slot[y]] > o))) {
            slot[y + 1] = slot[y];
            count[y + 1] = count[y];
            --y;
        }
        slot[y + 1] = t;
        count[y + 1] = c;
    }
    return n;
}

// a thread per line; leaders (bit 1 of is_first, dtok_scan_lines_kernel) work
__global__ void __launch_bounds__(kDtokThreads) readmap_len_kernel(DtokArgs a, ReadmapArgs m) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_lines || !(a.is_first[i] & 2u)) return;
    const uint32_t mate = a.lmeta[i] >> 28;
    // the read's index: reads before the run + mates of the run that come first (align.py:327-332)
    uint32_t s = i;
    bool seen[3] = {false, false, false};
    while (!a.is_start[s]) {
        --s;
        if (a.is_first[s] & 1u) {
            const uint32_t q = a.lmeta[s] >> 28;
            if (q < 3u) seen[q] = true;
        }
    }
    for (uint32_t j = i + 1u; j < a.n_lines && !a.is_start[j]; ++j)
        if (a.is_first[j] & 1u) {
            const uint32_t q = a.lmeta[j] >> 28;
            if (q < 3u) seen[q] = true;
        }
    uint32_t before = 0;
    for (uint32_t q = 0; q < mate && q < 3u; ++q) before += seen[q] ? 1u : 0u;
    const uint32_t r = (uint32_t)(a.line_scan[s] >> 32) + before;
    if (r >= m.n_reads) return;
    int32_t slot[WK_WEIGHT_MAX_K];
    uint32_t count[WK_WEIGHT_MAX_K];
    const int n = readmap_taxa(a, m, i, slot, count);
    unsigned long long len = 0;
    if (n > 0) {
        len = (unsigned long long)(a.lmeta[i] & 0x0FFFFFFFu) + (mate ? 2u : 0u) + 1u;  // query (+ "/1"), the line's newline
        if (n == 1) {
            len += 1u + (m.shown_off[slot[0] + 1] - m.shown_off[slot[0]]);
        } else {
            for (int k = 0; k < n; ++k) len += 2u + (m.shown_off[slot[k] + 1] - m.shown_off[slot[k]]) + dec_digits(count[k]);
        }
    }
    m.read_line[r] = i;
    m.read_len[r] = len;
}

// sums of tiles of kDtokThreads values / their exclusive prefixes in place
__global__ void __launch_bounds__(kDtokThreads) u64_tile_sum_kernel(const unsigned long long* __restrict__ v, uint32_t n,
                                                                   unsigned long long* __restrict__ tile_sum) {
    __shared__ unsigned long long wsum[kDtokThreads / kWave];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long s = wave_sum(i < n ? v[i] : 0ull);
    if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x / kWave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (uint32_t w = 0; w < kDtokThreads / kWave; ++w) t += wsum[w];
        tile_sum[blockIdx.x] = t;
    }
}

__global__ void __launch_bounds__(kDtokThreads) u64_tile_prefix_kernel(unsigned long long* __restrict__ v, uint32_t n,
                                                                      const unsigned long long* __restrict__ tile_off) {
    __shared__ unsigned long long scan[kDtokThreads];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long x = i < n ? v[i] : 0ull;
    scan[threadIdx.x] = x;
    __syncthreads();
    for (uint32_t d = 1; d < kDtokThreads; d <<= 1) {
        const unsigned long long u = threadIdx.x >= d ? scan[threadIdx.x - d] : 0ull;
        __syncthreads();
        scan[threadIdx.x] += u;
        __syncthreads();
    }
    if (i < n) v[i] = tile_off[blockIdx.x] + scan[threadIdx.x] - x;
}

// a thread per read: its line at read_len[r] (now the exclusive prefix)
__global__ void __launch_bounds__(kDtokThreads) readmap_write_kernel(DtokArgs a, ReadmapArgs m) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m.n_reads) return;
    const uint32_t i = m.read_line[r];
    int32_t slot[WK_WEIGHT_MAX_K];
    uint32_t count[WK_WEIGHT_MAX_K];
    const int n = readmap_taxa(a, m, i, slot, count);
    if (n <= 0) return;
    unsigned long long at = m.read_len[r];
    const uint32_t qn = a.lmeta[i] & 0x0FFFFFFFu, mate = a.lmeta[i] >> 28;
    unsigned char* o = m.out;
    const unsigned char* q = a.text + a.line_start[i];
    for (uint32_t k = 0; k < qn; ++k) o[at + k] = q[k];
    at += qn;
    if (mate) {
        o[at++] = '/';
        o[at++] = (unsigned char)('0' + mate);
    }
    for (int k = 0; k < n; ++k) {
        o[at++] = '\t';
        const uint32_t lo = m.shown_off[slot[k]], hi = m.shown_off[slot[k] + 1];
        for (uint32_t p = lo; p < hi; ++p) o[at++] = m.shown[p];
        if (n > 1) {
            o[at++] = ':';
            const uint32_t d = dec_digits(count[k]);
            uint32_t v = count[k];
            for (uint32_t p = d; p-- > 0;) {
                o[at + p] = (unsigned char)('0' + v % 10u);
                v /= 10u;
            }
            at += d;
        }
    }
    o[at] = '\n';
}

}  // namespace wk
