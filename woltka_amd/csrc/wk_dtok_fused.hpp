// wk_dtok_fused.hpp — the plain SAM tokenizer as ONE kernel.
//
// wk_dtok.hpp does a block of text in six launches (count -> tile_scan -> lines ->
// parse -> runs -> first_emit): the text is read three times and every kernel
// hands per-line arrays to the next through HBM (5.8x the text in fabric traffic,
// profiles/r05_e2e_lca_profile.json).  Here a workgroup keeps a tile of text in
// LDS and does everything align.parse_sam_file + plain_mapper do to its lines
// (woltka/align.py:258-347, 47-115) in that one residency: line starts, the
// first three tabs, FLAG -> mate, RNAME -> dictionary, runs of equal QNAME, the
// subject *sets* of a run's reads, and the packed records (subject | position <<
// 23 | size << 27, wk_weigh.hpp).  Per-line arrays never leave LDS; the text is
// read from HBM once.
//
// Tiles and ownership.  The text of a block is cut into tiles of at most kFzTile bytes
// (so many that every workgroup gets the same number of them: FusedArgs::tile).  A
// run of equal QNAMEs belongs to the tile its first line starts in; the workgroup
// of a tile therefore looks kFzBack bytes back (the line before its first one:
// does that one continue a run?) and kFzFwd bytes ahead (the rest of its last
// run).  A run that does not end inside the window, a line that starts in the tile
// and does not end inside the window, more than kFzLines lines in a
// window, or a line before the tile that cannot be found in the window's back
// part is nothing this kernel guesses about: the first three set kDtokSpill (the
// block is done again by the six kernels, which have no such limits), the last
// is looked up in global memory.
//
// Records.  Workgroups are persistent (a few per CU, tiles taken round-robin)
// and keep up to kFzCap records per slice of the subject table in LDS; a full
// buffer leaves with ONE returning atomic on the stream's cursor and coalesced
// stores.  (A reservation per tile and slice would be 12 k atomics on one cache
// line per 64 MB block: at ~11 ns each, serialised, 130 us -- more than the rest
// of the kernel.)  The histogram does not care about the order of the records.
//
// Anything the six kernels would leave to the host tokenizer (a line of fewer than
// four fields, a FLAG that is no number, both mate bits, a read of more than 16
// subjects) sets the same flags here; subjects the dictionary does not know are
// listed the same way.  The caller rolls the streams back and runs the unfused
// kernels on such a block.
#pragma once
#include "wk_dtok.hpp"

namespace wk {

constexpr uint32_t kFzThreads = 512;
constexpr uint32_t kFzWaves = kFzThreads / kWave;
constexpr uint32_t kFzTile = 16384;
constexpr uint32_t kFzBack = 1024;
constexpr uint32_t kFzFwd = 3072;
constexpr uint32_t kFzWin = kFzTile + kFzBack + kFzFwd;
constexpr uint32_t kFzChunks = kFzWin / 16 + 1;   // (+1: the byte behind a text without a last newline)
constexpr uint32_t kFzChunksPerWave = (kFzChunks + kFzWaves - 1) / kFzWaves;
constexpr uint32_t kFzRounds = (kFzChunksPerWave + kWave - 1) / kWave;
constexpr uint32_t kFzLines = 1024;   // lines of a window
constexpr uint32_t kFzStreams = 4;    // slices of the subject table (more: the unfused kernels)
constexpr uint32_t kFzCap = 1024;     // records kept per slice (>= kFzLines: a tile's records always fit an empty buffer)
static_assert(kFzCap >= kFzLines, "a tile's records fit an empty buffer");
static_assert(kFzWin + 32 < 65536, "window offsets fit 16 bits");

constexpr uint32_t kDtokSpill = 64;   // the fused kernel's limits (see above): the unfused kernels take the block

struct FusedArgs {
    const unsigned char* text;  // [n] + 64 readable bytes behind (zero)
    uint32_t n;
    uint32_t open_end;          // the text's last byte is no newline: a line ends at n
    uint32_t n_tiles;
    uint32_t tile;              // bytes per tile (a multiple of 16, <= kFzTile): the block in equal shares of the workgroups' rounds
    const struct DictSlot8* dict8;
    const uint4* names16;       // by id
    uint32_t dict_mask;
    const unsigned char* arena;
    uint2* unknown;
    uint32_t unknown_cap;
    DtokState* state;
    DtokState* host_state;               // pinned host memory: the block's scalars, written by the last workgroup
    unsigned long long* backup_next;     // [kMaxStreams] the streams' cursors behind this block (= in front of the next)
    uint32_t* host_seq;                  // pinned host memory or null: `seq` is stored there (system scope, release) once host_state is written
    uint32_t seq;
    StreamSet streams;
    const int32_t* submap;      // tokenizer id -> subject index, when they differ (`--trim-sub`), or null
    uint32_t n_submap;
    uint32_t ablate;            // (measurement, wk_tune "fz_ablate": phases left out -- results are wrong then)
};

// the streams' cursors put aside and the block's scalars cleared, in front of the fused kernel on its stream
__global__ void dtok_fused_begin_kernel(unsigned long long* __restrict__ backup, const unsigned long long* __restrict__ cursor, DtokState* state) {
    if (threadIdx.x < (uint32_t)kMaxStreams) backup[threadIdx.x] = cursor[threadIdx.x];
    if (threadIdx.x == 0) *state = DtokState{0u, 0u, 0ull, 0ull, 0ull, 0u, 0u};
}

// 16 aligned bytes of the text, read once: kept out of the way of what the L2 should hold (the dictionary)
__device__ __forceinline__ uint4 fz_load_stream(const unsigned char* p) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ unsigned long long fz_load64(const unsigned char* p) {
    unsigned long long w;
    __builtin_memcpy(&w, p, 8);  // (gfx950: one ds_read_b64 / global_load_dwordx2 whatever the alignment)
    return w;
}
__device__ __forceinline__ uint32_t fz_load32(const unsigned char* p) {
    uint32_t w;
    __builtin_memcpy(&w, p, 4);
    return w;
}
// 0x80 in every byte of w that equals c
__device__ __forceinline__ unsigned long long fz_eq_bytes(unsigned long long w, unsigned char c) {
    const unsigned long long x = w ^ (0x0101010101010101ull * c);
    return ~(((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x | 0x7F7F7F7F7F7F7F7Full);
}
__device__ __forceinline__ uint32_t fz_marks32(uint32_t w, uint32_t c4) {
    const uint32_t x = w ^ c4;
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}
// the low `k` bytes of w (k <= 8)
__device__ __forceinline__ unsigned long long fz_low_bytes(unsigned long long w, uint32_t k) {
    return k >= 8u ? w : (w & ((1ull << (8u * k)) - 1ull));
}

// wkh::hash_bytes (dtok_hash), eight bytes per load; `p` may be read up to 7 bytes past the name
__device__ __forceinline__ unsigned long long fz_hash(const unsigned char* p, uint32_t n) {
    unsigned long long h = 0xcbf29ce484222325ull ^ ((unsigned long long)n * 0x9E3779B97F4A7C15ull);
    while (n >= 8u) {
        h = (h ^ fz_load64(p)) * 0x100000001b3ull;
        h ^= h >> 29;
        p += 8;
        n -= 8u;
    }
    const unsigned long long v = n ? fz_low_bytes(fz_load64(p), n) : 0ull;
    h = (h ^ v) * 0x100000001b3ull;
    h ^= h >> 32;
    h *= 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}

// are the n bytes at x and y equal?  (both readable 7 bytes past their ends)
__device__ __forceinline__ bool fz_same(const unsigned char* x, const unsigned char* y, uint32_t n) {
    while (n >= 8u) {
        if (fz_load64(x) != fz_load64(y)) return false;
        x += 8;
        y += 8;
        n -= 8u;
    }
    return n == 0u || fz_low_bytes(fz_load64(x) ^ fz_load64(y), n) == 0ull;
}

// The dictionary as this kernel probes it: DictSlot8 + the names by id (wk_dtok.hpp).  (On config 3's text, 100 k
// subjects met evenly, the probe's time did not depend on the slots' size -- 16-byte slots + arena, 32-byte slots with
// the name inside, 8-byte slots + names by id all gave 27 us of a block's 125: it is two dependent trips to the L2 /
// the fabric either way.  The compact form is kept for text whose lines name few subjects, see there.)
// The subject name[0, rn) (in LDS) in the dictionary: its id, or kLineUnknown and the name listed for the host.  In
// two halves, so that the first slot's trip to memory is under way while the caller does something else.
struct FzProbe {
    unsigned long long hv, n0, n1;
    uint32_t h;
    uint2 slot;
};
__device__ __forceinline__ FzProbe fz_probe_begin(const FusedArgs& a, const unsigned char* name, uint32_t rn) {
    FzProbe p;
    p.hv = fz_hash(name, rn);
    p.n0 = p.n1 = 0ull;
    if (rn <= 15u) {
        p.n0 = fz_low_bytes(fz_load64(name), rn);
        p.n1 = (rn > 8u ? fz_low_bytes(fz_load64(name + 8), rn - 8u) : 0ull) | ((unsigned long long)rn << 56);
    }
    p.h = (uint32_t)p.hv & a.dict_mask;
    p.slot = reinterpret_cast<const uint2*>(a.dict8)[p.h];
    return p;
}
__device__ __forceinline__ int32_t fz_probe_end(const FusedArgs& a, FzProbe p, const unsigned char* name, uint32_t rn, uint32_t abs_off) {
    for (;;) {
        const int32_t id = (int32_t)p.slot.y;
        if (id < 0) break;
        if (p.slot.x == (uint32_t)(p.hv >> 32)) {
            const uint4 nm = a.names16[id];
            const unsigned long long s0 = ((unsigned long long)nm.y << 32) | nm.x, s1 = ((unsigned long long)nm.w << 32) | nm.z;
            if (rn <= 15u) {
                if (s0 == p.n0 && s1 == p.n1) return id;
            } else if ((nm.w >> 24) == 0xFFu) {
                const unsigned char* full = a.arena + nm.x;  // [len:4][bytes], 16 zero bytes behind the arena
                if (fz_load32(full) == rn && fz_same(name, full + 4, rn)) return id;
            }
        }
        p.h = (p.h + 1u) & a.dict_mask;
        p.slot = reinterpret_cast<const uint2*>(a.dict8)[p.h];
    }
    const uint32_t at = atomicAdd(&a.state->n_unknown, 1u);
    if (at < a.unknown_cap)
        a.unknown[at] = make_uint2(abs_off, rn);
    else
        atomicOr(&a.state->flags, kDtokUnknownFull);
    return kLineUnknown;
}

// The mapped line before text position `at` (a line start) that no window holds: its QNAME compared with
// the n bytes at `q`.  true = the line at `at` starts a run.  (Global memory, a byte at a time: a tile whose
// kFzBack bytes in front hold no complete mapped line -- long or unmapped lines.)
__device__ bool fz_starts_run_slow(const unsigned char* __restrict__ text, uint32_t at, const unsigned char* q, uint32_t qn) {
    uint32_t pos = at;
    while (pos > 0u) {
        const uint32_t e = pos - 1u;  // the newline that ends the line before
        uint32_t s = e;
        while (s > 0u && text[s - 1u] != '\n') --s;
        uint32_t tab[3], nt = 0;
        for (uint32_t p = s; p < e && nt < 3u; ++p)
            if (text[p] == '\t') tab[nt++] = p;
        if (nt == 3u && !(tab[2] - tab[1] == 2u && text[tab[1] + 1u] == '*')) {
            if (tab[0] - s != qn) return true;
            for (uint32_t k = 0; k < qn; ++k)
                if (text[s + k] != q[k]) return true;
            return false;
        }
        pos = s;  // (unmapped, or no row at all -- its tile sends the block to the host): the line before
    }
    return true;
}

// per-line word in LDS
constexpr uint32_t kFiSubj = (1u << 23) - 1u;   // subject index (all ones: not in the dictionary)
constexpr uint32_t kFiMateShift = 24;
constexpr uint32_t kFiMapped = 1u << 26;
constexpr uint32_t kFiStart = 1u << 27;          // starts a run of equal QNAMEs
constexpr uint32_t kFiFirst = 1u << 28;          // first line of its read (run, mate) that names its subject
constexpr uint32_t kFiExcl = 1u << 29;           // names a subject of the exclusion set
constexpr uint32_t kFiDropped = 1u << 30;        // (on the line that starts a run) a line of the run does: the run is dropped whole
constexpr uint32_t kFiRead = kFiMapped | (3u << kFiMateShift);          // same read of a run: same mate (and mapped)
constexpr uint32_t kFiKey = kFiRead | kFiSubj;                          // ... and the same subject
constexpr uint32_t kFzPad = 8;                   // words in front of / behind the lines' words (walks read eight at a time)

// the four 0x80 marks of a word (bits 7, 15, 23, 31) as bits 0-3
__device__ __forceinline__ uint32_t fz_nibble(uint32_t z) { return ((z >> 7) * 0x00204081u >> 21) & 15u; }

__global__ void __launch_bounds__(kFzThreads) __attribute__((amdgpu_waves_per_eu(6, 6))) dtok_fused_kernel(FusedArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char txt[kFzChunks * 16 + 32];
    __shared__ uint16_t ls[kFzLines + 2];                                    // line starts (window offsets); ls[k + 1] - 1 = the newline of line k
    __shared__ uint16_t f_qn[kFzLines], f_rb[kFzLines], f_rn[kFzLines];      // QNAME length, RNAME offset and length
    __shared__ __attribute__((aligned(16))) uint32_t info_[kFzPad + kFzLines + kFzPad + 8];
    __shared__ uint32_t rbuf[kFzStreams][kFzCap];
    __shared__ uint32_t rcnt[kFzStreams];
    __shared__ unsigned long long newc_packed;   // records of the tile at hand, per slice: 16 bits each
    static_assert(kFzStreams <= 4 && kFzLines < 65536, "four 16-bit counts");
    __shared__ unsigned long long gbase[kFzStreams];
    __shared__ uint32_t wtot[kFzWaves];
    __shared__ uint32_t own[2];                                              // first owned line, first line of the next tile's runs
    __shared__ uint32_t wg_flags;
    uint32_t* const info = info_ + kFzPad;

    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const uint32_t n_streams = a.streams.n_streams;
    if (tid < kFzStreams) rcnt[tid] = 0u;
    if (tid == 0) newc_packed = 0ull;
    if (tid < kFzPad) info_[tid] = kFiStart;   // (a walk back stops here at the latest; it never gets here: a run starts at or behind the first owned line)
    if (tid == 0) wg_flags = 0u;
    uint32_t my_flags = 0, my_rec = 0, my_reads = 0, my_lines = 0;
    const uint32_t text_end = a.n + (a.open_end ? 1u : 0u);  // (a text without a last newline: as if one followed)

    // all of one slice's buffer to its stream (every thread calls this)
    auto flush = [&](uint32_t k) {
        __syncthreads();
        const uint32_t cnt = (a.ablate & 64u) ? 0u : rcnt[k];
        if (tid == 0) gbase[k] = cnt ? atomicAdd(&a.streams.cursor[k], (unsigned long long)cnt) : 0ull;
        __syncthreads();
        const unsigned long long base = gbase[k];
        for (uint32_t i = tid; i < cnt; i += kFzThreads) {
            if (base + i < a.streams.cap)
                a.streams.out[k][base + i] = rbuf[k][i];
            else
                my_flags |= kDtokSpill;
        }
        __syncthreads();
        if (tid == 0) rcnt[k] = 0u;
        __syncthreads();
    };

    for (uint32_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const uint32_t t0 = tile * a.tile;
        const uint32_t t1 = min(t0 + a.tile, a.n);
        const uint32_t w0 = t0 >= kFzBack ? t0 - kFzBack : 0u;
        const uint32_t w1 = min(t0 + a.tile + kFzFwd, text_end);  // text positions [w0, w1) are looked at
        const bool to_end = w1 == text_end;                  // every line from here to the end of the text is whole
        __syncthreads();  // (the tile before is through with the arrays)
        if (tid < 2u) own[tid] = 0xFFFFFFFFu;

        // ---- the window into LDS; newlines per 16-byte chunk ----
        // (a wave takes kFzChunksPerWave consecutive chunks: no barrier inside the scan below)
        uint4 v[kFzRounds];
#pragma unroll
        for (uint32_t r = 0; r < kFzRounds; ++r) {  // (all loads under way before the first is looked at.  Loading a tile ahead, the
            // registers kept through the tile, bought nothing: three workgroups per CU take turns at the memory anyway)
            const uint32_t cw = r * kWave + lane;
            const uint32_t p = w0 + (wave * kFzChunksPerWave + cw) * 16u;
            v[r] = make_uint4(0u, 0u, 0u, 0u);
            if (cw < kFzChunksPerWave && p < a.n && p < w1) v[r] = fz_load_stream(a.text + p);  // (may pass n: the text's pad)
        }
        uint32_t marks[kFzRounds];
        uint32_t nl_mine = 0;
#pragma unroll
        for (uint32_t r = 0; r < kFzRounds; ++r) {
            const uint32_t cw = r * kWave + lane;
            const uint32_t c = wave * kFzChunksPerWave + cw;
            const uint32_t p = w0 + c * 16u;
            marks[r] = 0u;
            if (cw < kFzChunksPerWave && c < kFzChunks) {
                *reinterpret_cast<uint4*>(txt + c * 16u) = v[r];
                if (p < w1) {
                    uint32_t m = fz_nibble(fz_marks32(v[r].x, 0x0A0A0A0Au)) | (fz_nibble(fz_marks32(v[r].y, 0x0A0A0A0Au)) << 4) |
                                 (fz_nibble(fz_marks32(v[r].z, 0x0A0A0A0Au)) << 8) | (fz_nibble(fz_marks32(v[r].w, 0x0A0A0A0Au)) << 12);
                    if (p + 16u > a.n) m &= p >= a.n ? 0u : (1u << (a.n - p)) - 1u;  // (nothing behind n is text)
                    if (p >= t0 && p < t1) my_lines += (uint32_t)__popc(m);          // (the block's lines: counted where they end)
                    if (a.open_end && a.n >= p && a.n < p + 16u) m |= 1u << (a.n - p);  // (n itself ends an open last line)
                    if (p + 16u > w1) m &= (1u << (w1 - p)) - 1u;
                    marks[r] = m;
                    nl_mine += (uint32_t)__popc(m);
                }
            }
        }
        {
            const uint32_t s = (uint32_t)wave_sum((unsigned long long)nl_mine);
            if (lane == 0) wtot[wave] = s;
        }
        __syncthreads();
        uint32_t before = 0, total_nl = 0;
#pragma unroll
        for (uint32_t w = 0; w < kFzWaves; ++w) {
            before += w < wave ? wtot[w] : 0u;
            total_nl += wtot[w];
        }
        // lines of the window: line k = [ls[k], ls[k + 1] - 1), k < total_nl whole (line 0 only when the window starts the text)
        const bool too_many = total_nl + 1u > kFzLines || (a.ablate & (32u | 512u));
        if (too_many) {
            if (!(a.ablate & 512u)) my_flags |= kDtokSpill;
        } else {
            if (tid == 0) ls[0] = 0;
            uint32_t line = before;  // newlines in front of this wave's chunks
#pragma unroll
            for (uint32_t r = 0; r < kFzRounds; ++r) {
                const uint32_t c = wave * kFzChunksPerWave + r * kWave + lane;
                const uint32_t x = (uint32_t)__popc(marks[r]);
                uint32_t inc = x;
#pragma unroll
                for (uint32_t d = 1; d < (uint32_t)kWave; d <<= 1) {
                    const uint32_t up = __shfl_up(inc, d, kWave);
                    if (lane >= d) inc += up;
                }
                uint32_t at = line + inc - x;
                uint32_t m = marks[r];
                while (m) {
                    const uint32_t b = (uint32_t)__ffs((int)m) - 1u;
                    ls[++at] = (uint16_t)(c * 16u + b + 1u);
                    m &= m - 1u;
                }
                line += __shfl(inc, kWave - 1, kWave);
            }
        }
        __syncthreads();
        // A line that STARTS in this tile and does not end inside the window is a line nobody sees whole: its tile is
        // the one that would own a run it starts, and the tiles behind it find no run start in it.  (Lines of more
        // than kFzFwd bytes -- SEQ / QUAL of a long read kept; text that comes through the column trim has none.)
        if (!too_many && (total_nl > 0u || w0 == 0u)) {
            const uint32_t trail = w0 + (uint32_t)ls[total_nl];   // the first byte behind the window's last newline
            if (trail >= t0 && trail < t1) my_flags |= kDtokSpill;
        }
        const uint32_t first_line = w0 == 0u ? 0u : 1u;
        const uint32_t n_lines = too_many ? 0u : total_nl;  // whole lines: [first_line, n_lines)
        if (tid < kFzPad) info[n_lines + tid] = kFiStart;   // (a walk ahead stops behind the last whole line)
        if (tid == 0 && first_line) info[0] = 0u;

        // ---- a thread per line: three tabs, FLAG, RNAME ----
        for (uint32_t k = first_line + tid; k < n_lines; k += kFzThreads) {
            const uint32_t s = ls[k], e = (uint32_t)ls[k + 1] - 1u;
            uint32_t tab[3] = {0, 0, 0}, nt = 0;
            for (uint32_t p = s; p < e && nt < 3u && !(a.ablate & 256u); p += 32u) {  // (32 bytes under way at a time: the usual line needs no second round)
                unsigned long long w[4];
#pragma unroll
                for (uint32_t i = 0; i < 4; ++i) w[i] = fz_load64(txt + p + 8u * i);
#pragma unroll
                for (uint32_t i = 0; i < 4; ++i) {
                    unsigned long long z = fz_eq_bytes(w[i], '\t');
                    while (z && nt < 3u) {
                        const uint32_t q = p + 8u * i + (((uint32_t)__ffsll((long long)z) - 1u) >> 3);
                        if (q < e) tab[nt++] = q;
                        z &= z - 1ull;
                    }
                }
            }
            uint32_t word = 0;
            if (a.ablate & 16u) {
            } else if (nt < 3u) {  // not `qname, flag, rname, _ = line.split('\t', 3)` (align.py:313)
                my_flags |= kDtokShortLine;
            } else {
                const uint32_t fl = tab[1] - tab[0] - 1u;
                const unsigned long long fw = fz_load64(txt + tab[0] + 1u);
                uint32_t flag = 0;
                bool digits = fl >= 1u && fl <= 6u;
                for (uint32_t i = 0; i < fl && i < 6u; ++i) {
                    const uint32_t d = ((uint32_t)(fw >> (8u * i)) & 0xFFu) - (uint32_t)'0';
                    digits &= d <= 9u;
                    flag = flag * 10u + d;
                }
                const uint32_t rb = tab[1] + 1u, rn = tab[2] - rb;
                if (!digits) {
                    my_flags |= kDtokShortLine;
                } else if (rn == 1u && txt[rb] == '*') {
                    // unmapped: skipped before anything else (align.py:318-319)
                } else {
                    const uint32_t mate = (flag >> 6) & 3u;
                    if (mate == 3u) my_flags |= kDtokBothMates;
                    word = kFiMapped | (mate << kFiMateShift);
                    f_qn[k] = (uint16_t)(tab[0] - s);
                    f_rb[k] = (uint16_t)rb;
                    f_rn[k] = (uint16_t)rn;
                }
            }
            info[k] = word;
        }
        __syncthreads();

        // ---- a mapped line at or behind t0: does it start a run (QNAME against the mapped line before it)?  Its subject?
        // (the dictionary slot is on its way while the QNAMEs are compared; the lines behind the last owned run are
        // looked up for nothing -- a sixth of the window)
        for (uint32_t k = first_line + tid; k < n_lines; k += kFzThreads) {
            if (!(info[k] & kFiMapped) || w0 + ls[k] < t0) continue;
            const unsigned char* name = txt + f_rb[k];
            const uint32_t rn = f_rn[k];
            FzProbe probe;
            if (!(a.ablate & 1u)) probe = fz_probe_begin(a, name, rn);
            uint32_t j = k;
            bool found = false;
            while (j > first_line) {
                --j;
                if (info[j] & kFiMapped) {
                    found = true;
                    break;
                }
            }
            bool start;
            if (a.ablate & 8u)
                start = true;
            else if (found)
                start = f_qn[j] != f_qn[k] || !fz_same(txt + ls[k], txt + ls[j], f_qn[k]);
            else if (w0 == 0u)
                start = true;
            else
                start = fz_starts_run_slow(a.text, w0 + ls[first_line], txt + ls[k], f_qn[k]);
            int32_t sid = (a.ablate & 1u) ? (int32_t)(fz_load32(name + 4) % 1000u) : fz_probe_end(a, probe, name, rn, w0 + f_rb[k]);
            bool excluded = false;
            if (a.submap && sid >= 0) {
                if ((uint32_t)sid < a.n_submap) {
                    sid = a.submap[sid];
                    if (sid == kLineExcluded) {  // (`--exclude`: the run goes, all its mates; align.py:47-115)
                        excluded = true;
                        sid = 0;
                    }
                } else {  // (a name the host has not mapped yet: the block is done again)
                    my_flags |= kDtokSpill;
                    sid = -1;
                }
            }
            // (this thread's own word; the others look at its mapped bit only)
            info[k] |= (start ? kFiStart : 0u) | (excluded ? kFiExcl : 0u) | (sid < 0 ? kFiSubj : ((uint32_t)sid & kFiSubj));
            if (start) atomicMin(&own[w0 + ls[k] < t1 ? 0 : 1], k);
        }
        __syncthreads();
        // owned lines: from the first run that starts in the tile to the first run that starts behind it
        uint32_t ka = own[0], kb = own[1];
        if (ka == 0xFFFFFFFFu) {
            ka = kb = 0u;  // no run starts in this tile
        } else if (kb == 0xFFFFFFFFu) {
            if (to_end) {
                kb = n_lines;
            } else {  // the tile's last run may go on behind the window
                my_flags |= kDtokSpill;
                ka = kb = 0u;
            }
        }
        // ---- first line of its read (run, mate) that names its subject (the plain parsers keep sets, align.py:309) ----
        // (walks read eight lines' words at a time: one trip to the LDS per eight lines instead of two per line)
        for (uint32_t k = ka + tid; k < kb; k += kFzThreads) {
            const uint32_t mk = info[k];
            if (!(mk & kFiMapped)) continue;
            if (mk & kFiExcl) {  // the line its run starts with learns that the run is dropped (read behind the barrier)
                uint32_t j = k;
                while (!(info[j] & kFiStart)) --j;
                atomicOr(&info[j], kFiDropped);
                continue;
            }
            bool dup = false;
            if (!(mk & kFiStart) && !(a.ablate & 2u)) {
                bool done = false;
                for (uint32_t j = k; !done; j -= 8u) {
                    uint32_t w[8];
#pragma unroll
                    for (uint32_t i = 0; i < 8; ++i) w[i] = info[(int32_t)j - 1 - (int32_t)i];
#pragma unroll
                    for (uint32_t i = 0; i < 8; ++i) {
                        dup |= !done && ((w[i] ^ mk) & kFiKey) == 0u;
                        done |= (w[i] & kFiStart) != 0u;
                    }
                }
            }
            // (bit 28 of this thread's own word; the walks above look at the other bits of other lines' words)
            if (!dup) atomicOr(&info[k], kFiFirst);  // (an atomic: a line of an excluded subject may be marking this word as its run's start)
        }
        __syncthreads();
        // ---- records: position and size inside the read ----
        for (uint32_t k0 = ka; k0 < kb; k0 += kFzThreads) {  // (uniform trip count: barriers inside)
            const uint32_t k = k0 + tid;
            bool rec = false;
            uint32_t word = 0, sl = 0, at = 0;
            if (k < kb && (info[k] & kFiFirst) && !(a.ablate & 4u)) {
                const uint32_t mk = info[k];
                uint32_t pos = 0, size = 1;
                bool dropped = (mk & kFiStart) && (mk & kFiDropped);
                if (!(a.ablate & (2u | 128u))) {
                    if (!(mk & kFiStart)) {
                        bool done = false;
                        for (uint32_t j = k; !done; j -= 8u) {
                            uint32_t w[8];
#pragma unroll
                            for (uint32_t i = 0; i < 8; ++i) w[i] = info[(int32_t)j - 1 - (int32_t)i];
#pragma unroll
                            for (uint32_t i = 0; i < 8; ++i) {
                                pos += (!done && (w[i] & kFiFirst) && ((w[i] ^ mk) & kFiRead) == 0u) ? 1u : 0u;
                                dropped |= !done && (w[i] & kFiStart) && (w[i] & kFiDropped);
                                done |= (w[i] & kFiStart) != 0u;
                            }
                        }
                    }
                    bool done = false;
                    for (uint32_t j = k + 1u; !done; j += 8u) {
                        uint32_t w[8];
#pragma unroll
                        for (uint32_t i = 0; i < 8; ++i) w[i] = info[j + i];
#pragma unroll
                        for (uint32_t i = 0; i < 8; ++i) {
                            done |= (w[i] & kFiStart) != 0u;
                            size += (!done && (w[i] & kFiFirst) && ((w[i] ^ mk) & kFiRead) == 0u) ? 1u : 0u;
                        }
                    }
                    size += pos;
                }
                if (size > (uint32_t)WK_WEIGHT_MAX_K) my_flags |= kDtokBigRead;
                const uint32_t s = mk & kFiSubj;
                if (s != kFiSubj && !dropped) {  // (kFiSubj: a subject the dictionary does not know -- the block is done again anyway)
                    rec = true;
                    word = s | ((pos & 15u) << kWordSubjBits) | ((size & 31u) << kWordSizeShift);
                    sl = s / kSliceBins;
                    if (sl >= n_streams) sl = n_streams - 1u;  // (a subject beyond the table: the histogram reports it)
                    ++my_rec;
                    my_reads += pos == 0u ? 1u : 0u;
                }
            }
            // a place in the slice's buffer: ONE LDS atomic per wave (the slices' counts are 16-bit fields of one word)
            {
                unsigned long long add = 0;
                uint32_t mine_before = 0;
                for (uint32_t s2 = 0; s2 < n_streams; ++s2) {
                    const unsigned long long m = __ballot(rec && sl == s2);
                    add |= (unsigned long long)__popcll(m) << (16u * s2);
                    if (rec && sl == s2) mine_before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                }
                if (add) {  // (wave-uniform)
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(&newc_packed, add);
                    base = __shfl(base, 0, kWave);
                    at = (uint32_t)(base >> (16u * sl)) & 0xFFFFu;
                    at += mine_before;
                }
            }
            __syncthreads();
            const unsigned long long newc = newc_packed;
            for (uint32_t s2 = 0; s2 < n_streams; ++s2)
                if (rcnt[s2] + ((uint32_t)(newc >> (16u * s2)) & 0xFFFFu) > kFzCap) flush(s2);  // (uniform: every thread reads the same counters)
            if (rec) rbuf[sl][rcnt[sl] + at] = word;
            __syncthreads();
            if (tid < n_streams) rcnt[tid] += (uint32_t)(newc >> (16u * tid)) & 0xFFFFu;
            if (tid == 0) newc_packed = 0ull;
            __syncthreads();
        }
    }
    for (uint32_t s2 = 0; s2 < n_streams; ++s2) flush(s2);
    // the block's totals and flags: one set of adds per workgroup
    if (my_flags) atomicOr(&wg_flags, my_flags);
    const unsigned long long rec_w = wave_sum((unsigned long long)my_rec), reads_w = wave_sum((unsigned long long)my_reads),
                             lines_w = wave_sum((unsigned long long)my_lines);
    __shared__ unsigned long long w_rec[kFzWaves], w_reads[kFzWaves], w_lines[kFzWaves];
    if (lane == 0) {
        w_rec[wave] = rec_w;
        w_reads[wave] = reads_w;
        w_lines[wave] = lines_w;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long nrec = 0, nreads = 0, nlines = 0;
        for (uint32_t w = 0; w < kFzWaves; ++w) {
            nrec += w_rec[w];
            nreads += w_reads[w];
            nlines += w_lines[w];
        }
        if (nrec) atomicAdd(&a.state->n_out, nrec);
        if (nreads) atomicAdd(&a.state->n_reads, nreads);
        if (nlines) atomicAdd(&a.state->n_lines, nlines);
        if (wg_flags) atomicOr(&a.state->flags, wg_flags);
        // The last workgroup through does what two more launches used to: the block's scalars to pinned host memory
        // (the host reads them once the stream has been waited for), the streams' cursors put aside as the next
        // block's "before", the scalars cleared for it.
        __threadfence();
        own[0] = atomicAdd(&a.state->done, 1u) == gridDim.x - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (own[0]) {
        __threadfence();
        if (tid < (uint32_t)kMaxStreams)
            a.backup_next[tid] = __hip_atomic_load(&a.streams.cursor[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            DtokState st{};
            st.flags = __hip_atomic_load(&a.state->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            st.n_unknown = __hip_atomic_load(&a.state->n_unknown, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            st.n_out = __hip_atomic_load(&a.state->n_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            st.n_reads = __hip_atomic_load(&a.state->n_reads, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            st.n_lines = __hip_atomic_load(&a.state->n_lines, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *a.host_state = st;
            *a.state = DtokState{0u, 0u, 0ull, 0ull, 0ull, 0u, 0u};
            __threadfence_system();
            if (a.host_seq) __hip_atomic_store(a.host_seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace wk
