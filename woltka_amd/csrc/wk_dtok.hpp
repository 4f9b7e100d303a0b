// wk_dtok.hpp — SAM tokenizer on the device (plain flavour).
//
// The host tokenizer (wk_tokenize.cpp) spends ~60 ns of CPU per alignment
// record; a node's ranks share its CPUs, so with the text parsed on the host
// the end-to-end rate of N GPUs is the rate of one.  Here the host only moves
// bytes — pread into pinned memory, one copy to HBM — and the GPU does what
// align.parse_sam_file + plain_mapper do per line (woltka/align.py:258-347,
// 47-115): split QNAME / FLAG / RNAME, skip unmapped records before the
// QNAME-change test, group runs of equal QNAME into up to three reads by mate
// ((FLAG >> 6) & 3), keep each read's subjects as a set, and look the subject
// up in the dictionary.  The output is the packed record stream of
// wk_weigh.hpp (subject | position << 23 | size << 27), appended straight to
// the sample's accumulated records (wk_words_*): the text never becomes
// anything else on the host.
//
// Kernels (a block of text = lines ending at a run boundary, cut by the host):
//   dtok_count / dtok_lines   positions of the line starts (two passes + scan)
//   dtok_parse                a thread per line: fields, FLAG, dictionary probe
//   dtok_runs                 a thread per line: does its QNAME start a run?
//   dtok_first / dtok_emit    a thread per line: first line of its read with its
//                             subject? -> position and size in the read -> word
// Anything the kernels are not sure to treat like the reference — a short or
// malformed line, both mate bits, a read of more than 16 subjects — sets a flag
// and the host tokenizer takes the block instead.  Subjects the dictionary
// does not know yet are listed; the host interns them in text order (so the
// subject indices are those the host tokenizer would have assigned) and the
// parse runs again.
#pragma once
#include "wk_device.hpp"
#include "wk_strata.hpp"
#include "wk_weigh.hpp"

namespace wk {

constexpr uint32_t kDtokThreads = 256;
constexpr uint32_t kDtokTile = kDtokThreads * 16;  // bytes per workgroup and round

// flags of a block (OR-ed into DtokState::flags)
constexpr uint32_t kDtokShortLine = 1;   // fewer than four fields / FLAG not a number
constexpr uint32_t kDtokBothMates = 2;   // FLAG with both mate bits
constexpr uint32_t kDtokBigRead = 4;     // a read of more than WK_WEIGHT_MAX_K subjects
constexpr uint32_t kDtokUnknownFull = 8; // more unknown subjects than the list holds
constexpr uint32_t kDtokLongName = 16;   // a QNAME longer than the per-line word can say
constexpr uint32_t kDtokBadNumber = 32;  // POS / CIGAR text the kernels leave to the host's Python-exact parsers

// per-line result of dtok_parse
constexpr int32_t kLineUnknown = -1;   // subject not in the dictionary
constexpr int32_t kLineUnmapped = -2;  // RNAME "*"
constexpr int32_t kLineBad = -3;
constexpr int32_t kLineExcluded = -4;  // a subject of the exclusion set (the submap's value for its names): drops its whole run

struct DtokState {  // device scalars of one block
    uint32_t flags;
    uint32_t n_unknown;
    unsigned long long n_out;    // words emitted by dtok_emit
    unsigned long long n_reads;  // reads (non-empty mate groups) emitted
    unsigned long long n_lines;  // (dtok_fused_kernel) newlines of the block
    unsigned int done;           // (dtok_fused_kernel) workgroups through: the last one reports and clears
    unsigned int pad_;
};

struct DictSlot {
    unsigned long long hash;
    int32_t id;  // -1 = empty
    uint32_t off;
};

// The same dictionary in a form that costs two requests per probe instead of fifteen: slots of 8 bytes {high half of
// the name's hash, id}, and the names by id in 16-byte records (up to 15 bytes + the length in the last byte; longer
// names: 0xFF there and their arena offset in the first word).  It matters when few subjects take most of the lines
// (config 5: 5 000 genomes under a Zipf law): every wave then asks for the same few cache lines, which one L2
// channel serves one request at a time -- with a byte-wise compare against the arena (a request per byte and wave)
// dtok_parse took 330 us per block on such text against 80 us on config 3's.
struct DictSlot8 {
    uint32_t hash_hi;
    int32_t id;  // -1 = empty
};

struct DtokArgs {
    const unsigned char* text;  // [n] + 64 readable bytes behind
    uint32_t n;
    uint32_t fmt;  // 0 SAM, 1 simple map, 2 BLAST tabular, 3 PAF (WK_FMT_*; the simple map in the plain flavour only)
    const uint32_t* line_start;  // [n_lines + 1], line_start[n_lines] = n (+1 if the last line has no newline)
    uint32_t n_lines;
    int32_t* lsubj;    // [n_lines]
    uint32_t* lmeta;   // [n_lines] QNAME length | mate << 28
    const DictSlot* dict;
    uint32_t dict_mask;
    const unsigned char* arena;  // [len:4][bytes] per name
    const DictSlot8* dict8;      // (the same slots, 8 bytes each)
    const uint4* names16;        // by id
    uint2* unknown;              // (offset, length) of RNAMEs not in the dictionary
    uint32_t unknown_cap;
    unsigned char* is_start;     // [n_lines]
    unsigned char* is_first;     // [n_lines] first line of its read that names its subject
    DtokState* state;
    uint32_t* out;      // packed records
    uint32_t out_cap;
    StreamSet streams;  // (weighted histogram: the records by slice of the subject table, wk_weigh.hpp)
    unsigned long long* cursor_backup;  // [kMaxStreams] the streams' cursors before this block's emission (or null)
    // "ex" flavour (coord-match): per line POS - 1, reference end, aligned length
    int32_t* lbeg;
    int32_t* lend;
    uint32_t* llen;
    unsigned long long* line_scan;  // [n_lines] exclusive prefix of (hits | leaders << 32) over the lines
    const int32_t* gmap;            // subject id -> genome index of the gene tables (-1: no genes)
    uint32_t n_gmap;
    int32_t* o_genome;              // the staged hits (wk_ordinal_stage's arrays) ...
    int32_t* o_beg;
    int32_t* o_end;
    uint32_t* o_len;
    int32_t* o_hoff;                // ... and the reads' offsets
    uint32_t o_hit_base;            // (hits staged in front of this block's, wk_dtok_stage_hits_append: o_* point behind them, the offsets count from them)
    int32_t* o_group;               // (with a strata map on the device) the reads' (sample, stratum) groups
    // plain flavour: the tokenizer's ids -> the subject indices the records carry, when they differ (`--trim-sub`:
    // several names, one subject; workflow.py:840-841)
    const int32_t* submap;
    uint32_t n_submap;
};

// lsubj[i] = submap[lsubj[i]] for the mapped lines, in front of the kernels that look at the subjects (after the host
// has seen the names the parse listed): a read's subjects are a set of what the names are trimmed to
__global__ void __launch_bounds__(kDtokThreads) dtok_submap_kernel(DtokArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_lines) return;
    const int32_t id = a.lsubj[i];
    if (id < 0) return;
    if ((uint32_t)id < a.n_submap)
        a.lsubj[i] = a.submap[id];
    else
        atomicOr(&a.state->flags, kDtokBadNumber);  // (a name the host has not mapped: the host tokenizer's block)
}

// `--exclude` (align.plain_mapper's `excl`, align.py:47-115, and the filtering parsers, align.py:438-470): a query that
// hits a subject of the set is dropped whole, all its mates.  Behind dtok_runs (the runs were told with every mapped
// line in place) and dtok_submap (the set's names carry kLineExcluded): a line of such a subject marks the line its
// run starts with (is_first[], free until dtok_first), then every line of a marked run turns into a line without a
// subject -- which the kernels behind skip, and which ends no run.
__global__ void __launch_bounds__(kDtokThreads) dtok_excl_mark_kernel(DtokArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_lines || a.lsubj[i] != kLineExcluded) return;
    uint32_t j = i;
    while (!a.is_start[j]) --j;  // (a mapped line at or before i starts the run)
    a.is_first[j] = 0xEE;
}
__global__ void __launch_bounds__(kDtokThreads) dtok_excl_drop_kernel(DtokArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_lines) return;
    const int32_t s = a.lsubj[i];
    if (s < 0 && s != kLineExcluded) return;
    uint32_t j = i;
    while (!a.is_start[j]) --j;
    if (a.is_first[j] == 0xEE) a.lsubj[i] = kLineUnmapped;
}


__device__ __forceinline__ uint32_t count_newlines16(const uint4 v) {
    auto cnt = [](uint32_t w) {
        const uint32_t x = w ^ 0x0A0A0A0Au;                       // bytes equal to '\n' become 0
        const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);  // 0x80 where the byte was 0
        return (uint32_t)__popc(z);
    };
    return cnt(v.x) + cnt(v.y) + cnt(v.z) + cnt(v.w);
}

// newlines per tile of kDtokTile bytes
__global__ void __launch_bounds__(kDtokThreads) dtok_count_kernel(const unsigned char* __restrict__ text, uint32_t n,
                                                                 unsigned long long* __restrict__ tile_count) {
    __shared__ uint32_t wsum[kDtokThreads / kWave];
    const uint32_t tile = blockIdx.x;
    // (the 64 bytes behind the text are zeroed here, by the last tile: no fill launch per block.  The
    // kernels that look behind `n` -- vector loads of a line's tail -- run behind this one on the stream)
    if (tile == gridDim.x - 1u && threadIdx.x < 64u) const_cast<unsigned char*>(text)[n + threadIdx.x] = 0;
    const uint32_t p = tile * kDtokTile + threadIdx.x * 16u;
    uint32_t c = 0;
    if (p + 16u <= n) {
        c = count_newlines16(*reinterpret_cast<const uint4*>(text + p));
    } else if (p < n) {
        for (uint32_t i = p; i < n; ++i) c += text[i] == '\n';
    }
    unsigned long long s = wave_sum((unsigned long long)c);
    if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x / kWave] = (uint32_t)s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (uint32_t w = 0; w < kDtokThreads / kWave; ++w) t += wsum[w];
        tile_count[tile] = t;
    }
}

// line_start[1 + k] = position behind the k-th newline (line_start[0] = 0 is
// written by the host side); tile_off = exclusive scan of tile_count
__global__ void __launch_bounds__(kDtokThreads) dtok_lines_kernel(const unsigned char* __restrict__ text, uint32_t n,
                                                                 const unsigned long long* __restrict__ tile_off,
                                                                 uint32_t* __restrict__ line_start, DtokState* state = nullptr) {
    __shared__ uint32_t wtot[kDtokThreads / kWave];
    const uint32_t tile = blockIdx.x;
    if (tile == 0u && threadIdx.x == 0u) {
        line_start[0] = 0u;  // line 0 starts at 0
        if (state) *state = DtokState{0u, 0u, 0ull, 0ull};  // (the block's scalars, for the first parse)
    }
    const uint32_t p = tile * kDtokTile + threadIdx.x * 16u;
    // (16 aligned bytes in one load: the text has 64 zero bytes behind it, so a load that starts inside it never
    // leaves the buffer, and what lies behind `n` is no newline)
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (p < n) v = *reinterpret_cast<const uint4*>(text + p);
    auto marks = [](uint32_t w) {  // 0x80 in every byte that is '\n'
        const uint32_t x = w ^ 0x0A0A0A0Au;
        return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
    };
    uint32_t z[4] = {marks(v.x), marks(v.y), marks(v.z), marks(v.w)};
    if (p + 16u > n) {  // the block's last bytes: nothing behind n counts (the pad is zero, but be exact)
#pragma unroll
        for (uint32_t w = 0; w < 4; ++w)
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k)
                if (p + 4u * w + k >= n) z[w] &= ~(0x80u << (8u * k));
    }
    const uint32_t c = (uint32_t)(__popc(z[0]) + __popc(z[1]) + __popc(z[2]) + __popc(z[3]));
    // inclusive scan: shuffles inside the wave, the waves' totals through LDS (one barrier)
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t inc = c;
#pragma unroll
    for (uint32_t d = 1; d < kWave; d <<= 1) {
        const uint32_t up = __shfl_up(inc, d, kWave);
        if (lane >= d) inc += up;
    }
    if (lane == kWave - 1) wtot[wave] = inc;
    __syncthreads();
    uint32_t before = 0;
#pragma unroll
    for (uint32_t w = 0; w < kDtokThreads / kWave; ++w) before += w < wave ? wtot[w] : 0u;
    uint32_t at = (uint32_t)tile_off[tile] + before + inc - c + 1u;  // (+1: line 0 starts at 0)
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) {
        uint32_t m = z[w];
        while (m) {
            const uint32_t bit = (uint32_t)__ffs((int)m) - 1u;  // 7, 15, 23 or 31
            line_start[at++] = p + 4u * w + (bit >> 3) + 1u;
            m &= m - 1u;
        }
    }
}

// the host's hash of a name (wkh::hash_bytes), byte for byte
__device__ __forceinline__ unsigned long long dtok_hash(const unsigned char* p, uint32_t n) {
    unsigned long long h = 0xcbf29ce484222325ull ^ ((unsigned long long)n * 0x9E3779B97F4A7C15ull);
    while (n >= 8) {
        unsigned long long v = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) v |= (unsigned long long)p[i] << (8 * i);
        h = (h ^ v) * 0x100000001b3ull;
        h ^= h >> 29;
        p += 8;
        n -= 8;
    }
    unsigned long long v = 0;
    for (uint32_t i = 0; i < n; ++i) v |= (unsigned long long)p[i] << (8 * i);
    h = (h ^ v) * 0x100000001b3ull;
    h ^= h >> 32;
    h *= 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}

__device__ __forceinline__ unsigned long long dtok_load64(const unsigned char* p) {
    unsigned long long w;
    __builtin_memcpy(&w, p, 8);  // (gfx950: one load whatever the alignment, LDS or global)
    return w;
}
__device__ __forceinline__ unsigned long long dtok_low_bytes(unsigned long long w, uint32_t k) {
    return k >= 8u ? w : (w & ((1ull << (8u * k)) - 1ull));
}

// the subject text[rb, rb + rn) in the dictionary: its id, or kLineUnknown (and the name listed for the host).
// (The text may be read up to 7 bytes past the name: a tab and the rest of its line follow, or the text's pad.)
__device__ __forceinline__ int32_t dtok_subject(const DtokArgs& a, uint32_t rb, uint32_t rn) {
    const unsigned char* name = a.text + rb;
    const unsigned long long hv = dtok_hash(name, rn);
    unsigned long long n0 = 0, n1 = 0;
    if (rn <= 15u) {
        n0 = dtok_low_bytes(dtok_load64(name), rn);
        n1 = (rn > 8u ? dtok_low_bytes(dtok_load64(name + 8), rn - 8u) : 0ull) | ((unsigned long long)rn << 56);
    }
    uint32_t h = (uint32_t)hv & a.dict_mask;
    int32_t id = kLineUnknown;
    for (;;) {
        const uint2 slot = reinterpret_cast<const uint2*>(a.dict8)[h];
        if ((int32_t)slot.y < 0) break;
        if (slot.x == (uint32_t)(hv >> 32)) {
            const uint4 nm = a.names16[(int32_t)slot.y];
            bool same;
            if (rn <= 15u) {
                same = ((((unsigned long long)nm.y << 32) | nm.x) == n0) && ((((unsigned long long)nm.w << 32) | nm.z) == n1);
            } else {
                same = (nm.w >> 24) == 0xFFu;
                if (same) {
                    const unsigned char* full = a.arena + nm.x;  // [len:4][bytes]
                    same = ((uint32_t)full[0] | ((uint32_t)full[1] << 8) | ((uint32_t)full[2] << 16) | ((uint32_t)full[3] << 24)) == rn;
                    for (uint32_t k = 0; same && k < rn; ++k) same = full[4 + k] == name[k];
                }
            }
            if (same) {
                id = (int32_t)slot.y;
                break;
            }
        }
        h = (h + 1u) & a.dict_mask;
    }
    if (id == kLineUnknown) {
        const uint32_t at = atomicAdd(&a.state->n_unknown, 1u);
        if (at < a.unknown_cap)
            a.unknown[at] = make_uint2(rb, rn);
        else
            atomicOr(&a.state->flags, kDtokUnknownFull);
    }
    return id;
}

// ---- "ex" rows of BLAST tabular text and PAF (coord-match) ---------------------------
// int() / float() of a field as far as the kernels go: blanks around, a sign, decimal
// digits (a float: digits with a point and an exponent).  Anything else Python may or
// may not accept (underscores, "nan", other white space ...): false, and the block goes to
// the host's parsers.
__device__ __forceinline__ bool dtok_blank(unsigned char ch) { return ch == ' ' || (ch >= '\t' && ch <= '\r'); }

__device__ inline bool dtok_int_field(const unsigned char* __restrict__ t, uint32_t b, uint32_t e, long long& v) {
    while (b < e && dtok_blank(t[b])) ++b;
    while (e > b && dtok_blank(t[e - 1u])) --e;
    bool neg = false;
    if (b < e && (t[b] == '-' || t[b] == '+')) neg = t[b++] == '-';
    if (b >= e || e - b > 10u) return false;
    long long x = 0;
    for (; b < e; ++b) {
        const uint32_t d = (uint32_t)t[b] - (uint32_t)'0';
        if (d > 9u) return false;
        x = x * 10 + (long long)d;
    }
    v = neg ? -x : x;
    return true;
}

__device__ inline bool dtok_float_field(const unsigned char* __restrict__ t, uint32_t b, uint32_t e) {
    while (b < e && dtok_blank(t[b])) ++b;
    while (e > b && dtok_blank(t[e - 1u])) --e;
    if (b < e && (t[b] == '-' || t[b] == '+')) ++b;
    uint32_t digits = 0;
    while (b < e && (uint32_t)t[b] - (uint32_t)'0' <= 9u) ++b, ++digits;
    if (b < e && t[b] == '.') {
        ++b;
        while (b < e && (uint32_t)t[b] - (uint32_t)'0' <= 9u) ++b, ++digits;
    }
    if (digits == 0u) return false;
    if (b < e && (t[b] == 'e' || t[b] == 'E')) {
        ++b;
        if (b < e && (t[b] == '-' || t[b] == '+')) ++b;
        uint32_t ed = 0;
        while (b < e && (uint32_t)t[b] - (uint32_t)'0' <= 9u) ++b, ++ed;
        if (ed == 0u) return false;
    }
    return b == e;
}

// One line of BLAST tabular text (align.parse_b6o_file_ex, align.py:807-856: `x =
// line.split('\t')`; qseqid, sseqid, length, score = x[0], x[1], int(x[3]), float(x[11]);
// start, end = sorted(int(x[8]), int(x[9])); a line of fewer than twelve fields is
// skipped -- unless x[3] is there and no number: int() raises before x[11] is missed) or
// of PAF (align.parse_paf_file_ex, align.py:1046-1095: (x[5], int(x[11]), int(x[10]),
// int(x[7]), int(x[8])), a line that fails either way skipped).  Start as the staged hits
// hold it (0-based), end, aligned length.
__device__ inline void dtok_row_ex(const DtokArgs& a, uint32_t i, uint32_t lo, uint32_t hi) {
    uint32_t f[13];
    uint32_t nf = 1;
    f[0] = lo;
    for (uint32_t p = lo; p < hi && nf < 13u; ++p)
        if (a.text[p] == '\t') f[nf++] = p + 1u;
    auto fe = [&](uint32_t k) { return k + 1u < nf ? f[k + 1u] - 1u : hi; };
    const bool b6o = a.fmt == 2u;
    auto bad = [&] {
        a.lsubj[i] = kLineBad;
        a.lmeta[i] = 0;
        atomicOr(&a.state->flags, kDtokBadNumber);
    };
    long long n = 0, x = 0, y = 0, sc = 0;
    if (nf < 12u) {
        if (b6o && nf >= 4u && !dtok_int_field(a.text, f[3], fe(3), n)) return bad();  // (the host knows whether int() raises)
        a.lsubj[i] = kLineUnmapped;
        a.lmeta[i] = 0;
        return;
    }
    const uint32_t qn = fe(0) - lo;
    if (qn >= (1u << 28)) {
        a.lsubj[i] = kLineBad;
        a.lmeta[i] = 0;
        atomicOr(&a.state->flags, kDtokLongName);
        return;
    }
    bool ok;
    if (b6o)
        ok = dtok_int_field(a.text, f[3], fe(3), n) && dtok_float_field(a.text, f[11], fe(11)) &&
             dtok_int_field(a.text, f[8], fe(8), x) && dtok_int_field(a.text, f[9], fe(9), y);
    else
        ok = dtok_int_field(a.text, f[11], fe(11), sc) && dtok_int_field(a.text, f[10], fe(10), n) &&
             dtok_int_field(a.text, f[7], fe(7), x) && dtok_int_field(a.text, f[8], fe(8), y);
    long long beg = x, end = y;
    if (b6o) {
        beg = (x < y ? x : y) - 1;
        end = x < y ? y : x;
    }
    ok = ok && n >= 0 && n <= 2147483647ll && beg >= -2147483647ll && beg <= 2147483647ll && end >= -2147483647ll && end <= 2147483647ll;
    if (!ok) return bad();
    const uint32_t sub = b6o ? 1u : 5u;
    a.lmeta[i] = qn;
    a.lsubj[i] = dtok_subject(a, f[sub], fe(sub) - f[sub]);
    a.lbeg[i] = (int32_t)beg;
    a.lend[i] = (int32_t)end;
    a.llen[i] = (uint32_t)n;
}

// a thread per line: QNAME / FLAG / RNAME, mate, subject id; kEx: also POS and
// CIGAR -> start, end, aligned length (align.py:376-398, 572-583).  The simple
// map (align.parse_map_file, align.py:621-674: query <tab> subject, the subject
// right-stripped, lines without a tab ignored) and BLAST tabular rows
// (align.parse_b6o_file, align.py:753-803: `qseqid, sseqid, _ = line.split('\t',
// 2)`, lines of fewer fields ignored) are split here too, and so are PAF rows
// (align.parse_paf_file, align.py:984-1045: `qname, _, _, _, _, tname, _ =
// line.split('\t', 6)`, lines of fewer than seven fields ignored); an ignored line
// does not end a run of equal queries, like an unmapped SAM record.
// The text of a workgroup's lines staged in LDS: the 256 lines of a workgroup are one contiguous piece
// of text (10 KB of trimmed SAM), loaded with coalesced 16-byte loads; the threads then walk their lines
// byte by byte in LDS instead of issuing a global load per byte and lane (64 lanes x 42-byte stride: one
// cache line per lane and instruction).  Returns the pointer to index with ABSOLUTE text positions
// (the stage shifted back by the piece's start), or the global text when the piece does not fit.
constexpr uint32_t kDtokStage = 24 * 1024;
__device__ __forceinline__ const unsigned char* dtok_stage_lines(const DtokArgs& a, unsigned char* stage, uint32_t first_line, uint32_t end_line,
                                                                 uint32_t* lo_out) {
    // (uniform over the workgroup: every thread calls this, there is a barrier inside)
    const uint32_t lo = a.line_start[first_line] & ~15u;
    uint32_t hi = a.line_start[end_line];
    hi = hi > a.n ? a.n : hi;
    *lo_out = lo;
    if (hi <= lo || hi - lo > kDtokStage - 16u) return nullptr;
    for (uint32_t off = threadIdx.x * 16u; off < hi - lo; off += kDtokThreads * 16u)
        *reinterpret_cast<uint4*>(stage + off) = *reinterpret_cast<const uint4*>(a.text + lo + off);  // (16 bytes may pass `hi`: the text's pad)
    __syncthreads();
    // (as a flat address: the subtraction must not happen in the 32-bit LDS address space, where it wraps --
    // the cast to a generic pointer comes first, the arithmetic is done on the 64-bit integer)
    const unsigned char* flat = stage;
    return reinterpret_cast<const unsigned char*>(reinterpret_cast<uintptr_t>(flat) - (uintptr_t)lo);
}

template <bool kEx>
__global__ void __launch_bounds__(kDtokThreads) dtok_parse_kernel(DtokArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char stage[kDtokStage];
    {
        const uint32_t i0 = blockIdx.x * blockDim.x;
        const uint32_t i1 = min(i0 + blockDim.x, a.n_lines);
        uint32_t lo0;
        const unsigned char* staged = i0 < i1 ? dtok_stage_lines(a, stage, i0, i1, &lo0) : nullptr;
        if (staged) a.text = staged;  // (positions stay absolute: the unknown subjects' offsets, the lines' ends)
    }
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_lines) return;
    const uint32_t lo = a.line_start[i];
    uint32_t hi = a.line_start[i + 1];  // behind the line's newline (or n + 1 for a last line without one)
    hi = hi > lo ? hi - 1u : lo;        // the newline itself / the end of the text
    if (hi > a.n) hi = a.n;
    if constexpr (kEx) {
        if (a.fmt != 0u) return dtok_row_ex(a, i, lo, hi);
    }
    if constexpr (!kEx) {
        if (a.fmt != 0u) {
            // the query ends at the first tab; the subject is the field behind tab
            // number `sub` (the 2nd field, PAF: the 6th) and ends at the next tab
            const uint32_t sub = a.fmt == 3u ? 4u : 0u;
            uint32_t t0 = hi, ts = hi, t1 = hi, seen = 0;
            for (uint32_t p = lo; p < hi; ++p)
                if (a.text[p] == '\t') {
                    if (seen == 0u) t0 = p;
                    if (seen == sub) ts = p;
                    if (seen == sub + 1u) {
                        t1 = p;
                        break;
                    }
                    ++seen;
                }
            const uint32_t qn = t0 - lo;
            if (t0 == hi || (a.fmt >= 2u && t1 == hi)) {  // not a row of the format
                a.lsubj[i] = kLineUnmapped;
                a.lmeta[i] = 0;
                return;
            }
            if (qn >= (1u << 28)) {
                a.lsubj[i] = kLineBad;
                a.lmeta[i] = 0;
                atomicOr(&a.state->flags, kDtokLongName);
                return;
            }
            const uint32_t rb = ts + 1u;
            uint32_t re = t1;
            if (a.fmt == 1u)  // subject.rstrip()
                while (re > rb) {
                    const unsigned char ch = a.text[re - 1u];
                    if (ch == ' ' || ch == '\r' || ch == '\n' || ch == '\t' || ch == '\v' || ch == '\f')
                        --re;
                    else
                        break;
                }
            if (a.fmt == 1u && re > rb) {
                // a byte str.rstrip() may have an opinion on (\x1c-\x1f, the UTF-8
                // spaces), or a \r inside the row: the host reads the block like Python
                const unsigned char ch = a.text[re - 1u];
                bool odd = ch >= 0x80 || (ch >= 0x1c && ch <= 0x1f);
                for (uint32_t p = lo; p < re && !odd; ++p) odd = a.text[p] == '\r';
                if (odd) {
                    a.lsubj[i] = kLineBad;
                    a.lmeta[i] = 0;
                    atomicOr(&a.state->flags, kDtokShortLine);
                    return;
                }
            }
            a.lmeta[i] = qn;
            a.lsubj[i] = dtok_subject(a, rb, re - rb);
            return;
        }
    }
    // the first three (six) tabs
    constexpr int kTabs = kEx ? 6 : 3;
    uint32_t tab[kTabs];
    int nt = 0;
    for (uint32_t p = lo; p < hi && nt < kTabs; ++p)
        if (a.text[p] == '\t') tab[nt++] = p;
    if (nt < kTabs) {  // not `qname, flag, rname, _ = line.split('\t', 3)` (align.py:313; 6 for the "ex" parser)
        a.lsubj[i] = kLineBad;
        a.lmeta[i] = 0;
        atomicOr(&a.state->flags, kDtokShortLine);
        return;
    }
    uint32_t flag = 0;
    bool digits = tab[1] > tab[0] + 1u;  // (an empty FLAG: int('') raises, align.py:322)
    for (uint32_t p = tab[0] + 1u; p < tab[1]; ++p) {
        const uint32_t d = (uint32_t)a.text[p] - (uint32_t)'0';
        digits &= d <= 9u;
        flag = flag * 10u + d;
    }
    const uint32_t qn = tab[0] - lo;
    if (!digits || tab[1] - tab[0] > 7u || qn >= (1u << 28)) {
        a.lsubj[i] = kLineBad;
        a.lmeta[i] = 0;
        atomicOr(&a.state->flags, !digits || tab[1] - tab[0] > 7u ? kDtokShortLine : kDtokLongName);
        return;
    }
    const uint32_t rb = tab[1] + 1u, rn = tab[2] - rb;
    if (rn == 1u && a.text[rb] == '*') {  // unmapped: skipped before anything else (align.py:318-319)
        a.lsubj[i] = kLineUnmapped;
        a.lmeta[i] = qn;
        return;
    }
    const uint32_t mate = (flag >> 6) & 3u;
    if (mate == 3u) atomicOr(&a.state->flags, kDtokBothMates);
    a.lmeta[i] = qn | (mate << 28);
    a.lsubj[i] = dtok_subject(a, rb, rn);
    if constexpr (kEx) {
        // POS: [+-]digits (anything else int() may or may not accept: the host decides)
        uint32_t p = tab[2] + 1u;
        const uint32_t pe = tab[3];
        bool neg = false, ok = true;
        if (p < pe && (a.text[p] == '-' || a.text[p] == '+')) neg = a.text[p++] == '-';
        ok = p < pe && pe - p <= 10u;
        long long pos = 0;
        for (; ok && p < pe; ++p) {
            const uint32_t d = (uint32_t)a.text[p] - (uint32_t)'0';
            ok = d <= 9u;
            pos = pos * 10 + (long long)d;
        }
        if (neg) pos = -pos;
        // CIGAR: (digits op)+, ops M = X (aligned), D N (reference only), I S H P (neither)
        unsigned long long aligned = 0, extra = 0, num = 0;
        bool have = false;
        for (uint32_t q = tab[4] + 1u; ok && q < tab[5]; ++q) {
            const unsigned char ch = a.text[q];
            if (ch >= '0' && ch <= '9') {
                num = num * 10ull + (unsigned long long)(ch - '0');
                have = true;
                ok = num < (1ull << 40);
            } else if (ch == 'M' || ch == '=' || ch == 'X') {
                ok = have;
                aligned += num;
                num = 0;
                have = false;
            } else if (ch == 'D' || ch == 'N') {
                ok = have;
                extra += num;
                num = 0;
                have = false;
            } else if (ch == 'I' || ch == 'S' || ch == 'H' || ch == 'P') {
                num = 0;
                have = false;
            } else {
                ok = false;
            }
        }
        ok = ok && !have;  // (digits without an operation: the host's parser has its own opinion)
        const long long beg = pos - 1, end = pos - 1 + (long long)(aligned + extra);
        ok = ok && beg >= -2147483647ll && end <= 2147483647ll && aligned < (1ull << 32);
        if (!ok) {
            atomicOr(&a.state->flags, kDtokBadNumber);
            return;
        }
        a.lbeg[i] = (int32_t)beg;
        a.lend[i] = (int32_t)end;
        a.llen[i] = (uint32_t)aligned;
    }
}

// ---- "ex" flavour: the lines' hits as the staged chunk of the coord-match --------
// A hit = a mapped line of aligned length > 0 (ordinal.py:231).  The hits of a
// read — run of equal QNAME, mate — are contiguous in the output, reads ordered
// by (run, mate), hits in text order: what ordinal_mapper builds (ordinal.py:
// 219-237 over parse_sam_file_ex's pools).  Hits only move inside their run, so
// a hit's place = hits before its run + hits of lower mates in the run + earlier
// hits of its own read; a read's index likewise from its first hit.

// is the line a hit / the first hit of its read?  -> tile totals for the scan.
// kEx: a hit is a mapped line of aligned length > 0; else (plain flavour, ordered
// emission) a line dtok_first_kernel marked: the first of its read with its subject.
template <bool kEx>
__global__ void __launch_bounds__(kDtokThreads) dtok_hits_kernel(DtokArgs a, unsigned long long* __restrict__ tile_count) {
    __shared__ unsigned long long wsum[kDtokThreads / kWave];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0;
    auto hit = [&](uint32_t x) { return kEx ? (a.lsubj[x] >= 0 && a.llen[x] > 0u) : (a.is_first[x] != 0); };
    if (i < a.n_lines && hit(i)) {
        const uint32_t m = a.lmeta[i] >> 28;
        bool leader = true;
        if (!a.is_start[i]) {
            uint32_t j = i;
            do {
                --j;
                if (hit(j) && (a.lmeta[j] >> 28) == m) {
                    leader = false;
                    break;
                }
            } while (!a.is_start[j]);
        }
        v = 1ull | (leader ? 1ull << 32 : 0ull);
    }
    if (i < a.n_lines) a.line_scan[i] = v;  // (replaced by its exclusive prefix in dtok_scan_lines_kernel)
    const unsigned long long s = wave_sum(v);
    if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x / kWave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (uint32_t w = 0; w < kDtokThreads / kWave; ++w) t += wsum[w];
        tile_count[blockIdx.x] = t;
    }
}

// line_scan[i] (flags) -> exclusive prefix over all lines; is_first[i] keeps the flags (bit 0 hit, bit 1 leader)
__global__ void __launch_bounds__(kDtokThreads) dtok_scan_lines_kernel(DtokArgs a, const unsigned long long* __restrict__ tile_off) {
    __shared__ unsigned long long scan[kDtokThreads];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long v = i < a.n_lines ? a.line_scan[i] : 0ull;
    scan[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t d = 1; d < kDtokThreads; d <<= 1) {
        const unsigned long long u = threadIdx.x >= d ? scan[threadIdx.x - d] : 0ull;
        __syncthreads();
        scan[threadIdx.x] += u;
        __syncthreads();
    }
    if (i < a.n_lines) {
        a.line_scan[i] = tile_off[blockIdx.x] + scan[threadIdx.x] - v;
        a.is_first[i] = (unsigned char)((v & 1ull) | ((v >> 32) << 1));
    }
}

// kStrata: the read's group from the strata map on the device (wk_strata.hpp;
// classify.counter_strat, classify.py:216-249: -1 = not in the map, skipped)
template <bool kStrata>
__global__ void __launch_bounds__(kDtokThreads) dtok_place_kernel(DtokArgs a, StrataArgs strata) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_lines || !(a.is_first[i] & 1u)) return;
    const uint32_t m = a.lmeta[i] >> 28;
    // the run: back to its start, forward to its end
    uint32_t s = i, before = 0, mine = 0, groups_before = 0;
    bool seen[3] = {false, false, false};
    while (!a.is_start[s]) {
        --s;
        if (a.is_first[s] & 1u) {
            const uint32_t q = a.lmeta[s] >> 28;
            before += q < m ? 1u : 0u;
            mine += q == m ? 1u : 0u;
            if (q < 3u) seen[q] = true;
        }
    }
    for (uint32_t j = i + 1u; j < a.n_lines && !a.is_start[j]; ++j)
        if (a.is_first[j] & 1u) {
            const uint32_t q = a.lmeta[j] >> 28;
            before += q < m ? 1u : 0u;
            if (q < 3u) seen[q] = true;
        }
    const unsigned long long base = a.line_scan[s];  // hits / reads before the run
    const uint32_t at = (uint32_t)base + before + mine;
    const int32_t sid = a.lsubj[i];
    a.o_genome[at] = (uint32_t)sid < a.n_gmap ? a.gmap[sid] : -1;
    a.o_beg[at] = a.lbeg[i];
    a.o_end[at] = a.lend[i];
    a.o_len[at] = a.llen[i];
    if (a.is_first[i] & 2u) {  // the read's first hit: its offset
        for (uint32_t q = 0; q < m && q < 3u; ++q) groups_before += seen[q] ? 1u : 0u;
        const uint32_t r = (uint32_t)(base >> 32) + groups_before;
        a.o_hoff[r] = (int32_t)(at + a.o_hit_base);
        if constexpr (kStrata) a.o_group[r] = strata_lookup(strata, a.text + a.line_start[i], a.lmeta[i] & 0x0FFFFFFFu, m);
    }
}

// a thread per line: does the line start a run of equal QNAMEs?  (Compared
// with the previous *mapped* line: unmapped records do not split a run.)
__global__ void __launch_bounds__(kDtokThreads) dtok_runs_kernel(DtokArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // (a block the kernels give up on must leave the streams as they were: their cursors are put aside
    // here, in front of the emission on the same stream)
    if (a.cursor_backup && blockIdx.x == 0u && threadIdx.x < (uint32_t)kMaxStreams) a.cursor_backup[threadIdx.x] = a.streams.cursor[threadIdx.x];
    // the workgroup's lines and the line before them staged in LDS (dtok_stage_lines): the QNAMEs compared are
    // those of neighbouring lines
    __shared__ __attribute__((aligned(16))) unsigned char stage[kDtokStage];
    const unsigned char* staged = nullptr;
    uint32_t lo0 = 0;
    {
        const uint32_t i0 = blockIdx.x * blockDim.x;
        const uint32_t i1 = min(i0 + blockDim.x, a.n_lines);
        if (i0 < i1) staged = dtok_stage_lines(a, stage, i0 > 0u ? i0 - 1u : 0u, i1, &lo0);
    }
    if (i >= a.n_lines) return;
    unsigned char start = 0;
    if (a.lsubj[i] >= 0) {
        int64_t j = (int64_t)i - 1;
        while (j >= 0 && a.lsubj[j] < 0) --j;  // (only kLineUnmapped can be there: other codes send the block to the host)
        if (j < 0) {
            start = 1;
        } else {
            const uint32_t qn = a.lmeta[i] & 0x0FFFFFFFu, pn = a.lmeta[j] & 0x0FFFFFFFu;
            bool same = qn == pn;
            const uint32_t ls = a.line_start[i], ps = a.line_start[j];
            const unsigned char* x = (staged ? staged : a.text) + ls;
            const unsigned char* y = (staged && ps >= lo0 ? staged : a.text) + ps;  // (a line further back: from the text itself)
            for (uint32_t k = 0; same && k < qn; ++k) same = x[k] == y[k];
            start = same ? 0 : 1;
        }
    }
    a.is_start[i] = start;
}

// a thread per mapped line: is it the first line of its read (run, mate) that
// names its subject?  (The plain parsers collect subject sets, align.py:309.)
// Walks back to the run's start.
__global__ void __launch_bounds__(kDtokThreads) dtok_first_kernel(DtokArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_lines) return;
    const int32_t s = a.lsubj[i];
    unsigned char first = 0;
    if (s >= 0) {
        const uint32_t m = a.lmeta[i] >> 28;
        first = 1;
        if (!a.is_start[i]) {
            uint32_t j = i;
            do {
                --j;
                if (a.lsubj[j] == s && (a.lmeta[j] >> 28) == m) {
                    first = 0;
                    break;
                }
            } while (!a.is_start[j]);  // (a mapped line before i starts the run: j never passes 0)
        }
    }
    a.is_first[i] = first;
}

// a thread per line: the first lines of a read (dtok_first_kernel) are its
// records — position = first lines of the same read before it, size = all of
// them — and go out as packed words, appended in no particular order (the
// histogram does not care).
__global__ void __launch_bounds__(kDtokThreads) dtok_emit_kernel(DtokArgs a) {
    // kScatterItems lines per thread (a workgroup: kScatterItems x 256 consecutive lines, lane-interleaved):
    // their records leave with one reservation per stream (scatter_by_slice_n)
    const uint32_t first_line = blockIdx.x * (kDtokThreads * kScatterItems) + threadIdx.x;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    bool rec[kScatterItems];
    uint32_t word[kScatterItems];
    bool big = false;
    uint32_t n_rec = 0, n_reads = 0;
#pragma unroll
    for (uint32_t r = 0; r < kScatterItems; ++r) {
        const uint32_t i = first_line + r * kDtokThreads;
        rec[r] = false;
        word[r] = 0;
        uint32_t pos = 0;
        if (i < a.n_lines && a.is_first[i]) {
            rec[r] = true;
            const uint32_t m = a.lmeta[i] >> 28;
            if (!a.is_start[i]) {
                uint32_t j = i;
                do {
                    --j;
                    pos += (a.is_first[j] && (a.lmeta[j] >> 28) == m) ? 1u : 0u;
                } while (!a.is_start[j]);
            }
            uint32_t size = pos + 1u;
            for (uint32_t j = i + 1u; j < a.n_lines && !a.is_start[j]; ++j)
                size += (a.is_first[j] && (a.lmeta[j] >> 28) == m) ? 1u : 0u;
            big |= size > (uint32_t)WK_WEIGHT_MAX_K;
            word[r] = (uint32_t)a.lsubj[i] | ((pos & 15u) << kWordSubjBits) | ((size & 31u) << kWordSizeShift);
        }
        n_rec += (uint32_t)__popcll(__ballot(rec[r]));
        n_reads += (uint32_t)__popcll(__ballot(rec[r] && pos == 0u));
    }
    if (big) atomicOr(&a.state->flags, kDtokBigRead);
    // the block's totals: one pair of adds per workgroup
    __shared__ uint32_t w_rec[kDtokThreads / kWave], w_reads[kDtokThreads / kWave];
    const uint32_t wave = threadIdx.x / kWave;
    if (lane == 0) {
        w_rec[wave] = n_rec;
        w_reads[wave] = n_reads;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t n = 0, q = 0;
        for (uint32_t w = 0; w < kDtokThreads / kWave; ++w) {
            n += w_rec[w];
            q += w_reads[w];
        }
        if (n) atomicAdd(&a.state->n_out, (unsigned long long)n);
        if (q) atomicAdd(&a.state->n_reads, (unsigned long long)q);
    }
    scatter_by_slice_n<kDtokThreads, kScatterItems>(a.streams, rec, word);
}

// dtok_first + dtok_emit in one kernel (the unordered emission: records for the weighted histogram).  A
// workgroup takes kScatterItems x 256 consecutive lines; what the walks along a run read -- does the line
// start a run, its mate, its subject -- sits in LDS for those lines and kFeHalo lines on either side, the
// first-line flags are worked out there (no is_first[] round trip through HBM) and the walks for position
// and size run in LDS too.  A walk that leaves the staged window (a run of more than kFeHalo lines across
// the tile's edge) reads the global arrays and works the flag of a line out on the spot: same result, slower.
constexpr uint32_t kFeHalo = 96;
constexpr uint32_t kFeTile = kDtokThreads * kScatterItems;
constexpr uint32_t kFeWindow = kFeTile + 2 * kFeHalo;

struct RunView {
    const DtokArgs& a;
    const unsigned char* pk;  // per staged line: bit 0 starts a run, bit 1 first line of its read with its subject, bits 2-3 mate
    const int32_t* sj;        // per staged line: subject
    uint32_t w0, w1;          // staged lines [w0, w1)
    __device__ __forceinline__ bool in(uint32_t j) const { return j >= w0 && j < w1; }
    __device__ __forceinline__ bool start(uint32_t j) const { return in(j) ? (pk[j - w0] & 1u) != 0u : a.is_start[j] != 0; }
    __device__ __forceinline__ uint32_t mate(uint32_t j) const { return in(j) ? (uint32_t)(pk[j - w0] >> 2) & 3u : a.lmeta[j] >> 28; }
    __device__ __forceinline__ int32_t subj(uint32_t j) const { return in(j) ? sj[j - w0] : a.lsubj[j]; }
    // (dtok_first_kernel's rule) the first line of its run and mate that names its subject
    __device__ __forceinline__ bool first_slow(uint32_t j) const {
        const int32_t s = subj(j);
        if (s < 0) return false;
        if (start(j)) return true;
        const uint32_t m = mate(j);
        uint32_t k = j;
        do {
            --k;
            if (subj(k) == s && mate(k) == m) return false;
        } while (!start(k));  // (a mapped line before j starts the run: k never passes 0)
        return true;
    }
    __device__ __forceinline__ bool first(uint32_t j) const { return in(j) ? (pk[j - w0] & 2u) != 0u : first_slow(j); }
};

__global__ void __launch_bounds__(kDtokThreads) dtok_first_emit_kernel(DtokArgs a) {
    __shared__ unsigned char pk[kFeWindow];
    __shared__ int32_t sj[kFeWindow];
    const uint32_t l0 = blockIdx.x * kFeTile;
    const uint32_t w0 = l0 > kFeHalo ? l0 - kFeHalo : 0u;
    const uint32_t w1 = min(a.n_lines, l0 + kFeTile + kFeHalo);
    for (uint32_t j = w0 + threadIdx.x; j < w1; j += kDtokThreads) {
        pk[j - w0] = (unsigned char)((a.is_start[j] ? 1u : 0u) | ((a.lmeta[j] >> 28) << 2));
        sj[j - w0] = a.lsubj[j];
    }
    __syncthreads();
    const RunView v{a, pk, sj, w0, w1};
    // the first-line flags of the staged lines (a thread writes its own lines' bytes only)
    static_assert((kFeWindow + kDtokThreads - 1) / kDtokThreads <= 32, "a thread's flags fit one word");
    uint32_t fmask = 0;
    {
        uint32_t q = 0;
        for (uint32_t j = w0 + threadIdx.x; j < w1; j += kDtokThreads, ++q) fmask |= v.first_slow(j) ? 1u << q : 0u;
    }
    __syncthreads();  // (every walk has read the bytes as they were)
    {
        uint32_t q = 0;
        for (uint32_t j = w0 + threadIdx.x; j < w1; j += kDtokThreads, ++q)
            if (fmask >> q & 1u) pk[j - w0] |= 2u;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & (kWave - 1);
    bool rec[kScatterItems];
    uint32_t word[kScatterItems];
    bool big = false;
    uint32_t n_rec = 0, n_reads = 0;
#pragma unroll
    for (uint32_t r = 0; r < kScatterItems; ++r) {
        const uint32_t i = l0 + r * kDtokThreads + threadIdx.x;
        rec[r] = false;
        word[r] = 0;
        uint32_t pos = 0;
        if (i < a.n_lines && v.first(i)) {
            rec[r] = true;
            const uint32_t m = v.mate(i);
            if (!v.start(i)) {
                uint32_t j = i;
                do {
                    --j;
                    pos += (v.first(j) && v.mate(j) == m) ? 1u : 0u;
                } while (!v.start(j));
            }
            uint32_t size = pos + 1u;
            for (uint32_t j = i + 1u; j < a.n_lines && !v.start(j); ++j) size += (v.first(j) && v.mate(j) == m) ? 1u : 0u;
            big |= size > (uint32_t)WK_WEIGHT_MAX_K;
            word[r] = (uint32_t)v.subj(i) | ((pos & 15u) << kWordSubjBits) | ((size & 31u) << kWordSizeShift);
        }
        n_rec += (uint32_t)__popcll(__ballot(rec[r]));
        n_reads += (uint32_t)__popcll(__ballot(rec[r] && pos == 0u));
    }
    if (big) atomicOr(&a.state->flags, kDtokBigRead);
    __shared__ uint32_t w_rec[kDtokThreads / kWave], w_reads[kDtokThreads / kWave];
    const uint32_t wave = threadIdx.x / kWave;
    if (lane == 0) {
        w_rec[wave] = n_rec;
        w_reads[wave] = n_reads;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t n = 0, q = 0;
        for (uint32_t w = 0; w < kDtokThreads / kWave; ++w) {
            n += w_rec[w];
            q += w_reads[w];
        }
        if (n) atomicAdd(&a.state->n_out, (unsigned long long)n);
        if (q) atomicAdd(&a.state->n_reads, (unsigned long long)q);
    }
    scatter_by_slice_n<kDtokThreads, kScatterItems>(a.streams, rec, word);
}

// Plain flavour, ordered emission: the records of a read contiguous and in
// position order (what the free-rank stream needs, wk_free.hpp) — placed like the
// hits above: records before the run + records of lower mates in the run + the
// record's position in its read.
__global__ void __launch_bounds__(kDtokThreads) dtok_place_words_kernel(DtokArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_lines || !(a.is_first[i] & 1u)) return;
    const uint32_t m = a.lmeta[i] >> 28;
    uint32_t s = i, before = 0, pos = 0, size = 1;
    while (!a.is_start[s]) {
        --s;
        if (a.is_first[s] & 1u) {
            const uint32_t q = a.lmeta[s] >> 28;
            before += q < m ? 1u : 0u;
            pos += q == m ? 1u : 0u;
        }
    }
    for (uint32_t j = i + 1u; j < a.n_lines && !a.is_start[j]; ++j)
        if (a.is_first[j] & 1u) {
            const uint32_t q = a.lmeta[j] >> 28;
            before += q < m ? 1u : 0u;
            size += q == m ? 1u : 0u;
        }
    size += pos;
    if (size > (uint32_t)WK_WEIGHT_MAX_K) {
        atomicOr(&a.state->flags, kDtokBigRead);
        return;
    }
    const uint32_t at = (uint32_t)a.line_scan[s] + before + pos;
    if (at < a.out_cap) a.out[at] = (uint32_t)a.lsubj[i] | (pos << kWordSubjBits) | (size << kWordSizeShift);
}

}  // namespace wk
