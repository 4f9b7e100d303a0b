// wk_inflate.cpp — gzip input inflated natively, in parallel inside one file (host).
//
// The reference reads compressed alignments through `gzip -cdfq` children or
// Python's gzip module (file.readzip, woltka/file.py:62-129); decompression
// alone is 63 % of its run time on its published workload (doc/perform.md:
// 44-46).  A DEFLATE stream is sequential by construction -- every match may
// point into the 32 KB before it -- and one `gzip -d` delivers 0.3-0.5 GB/s of
// text, a fortieth of what the device tokenizer takes.  Here:
//
//  * a table-driven decoder (64-bit bit buffer refilled by one unaligned load,
//    11-bit first-level table for literals / lengths with the extra-bit count
//    packed into the entry, 8-bit for distances, word-wise match copies);
//  * one stream decoded by many threads: the compressed bytes are cut into
//    chunks, every chunk but the first FINDS a block start behind its cut
//    (bit by bit: BFINAL = 0, BTYPE = 2, a code-length code and two codes that
//    are complete) and decodes from there with an unknown window -- into
//    16-bit cells, where a value >= 0x8000 says "the byte at position p of the
//    32 KB before this chunk".  A chunk stops at the block start its successor
//    found; the windows are resolved in order (32 KB each), the cells of all
//    chunks then in parallel, straight into the caller's buffer.  Chunks whose
//    ends do not meet (a start that was none; no start found) are decoded again
//    by the next wave from the last position that is certain;
//  * chains of members that state their size (BGZF's 'BC' subfield, this
//    package's 'WK' one): one task per member, no windows to resolve;
//  * CRC-32 and ISIZE of every member verified (pieces' CRCs combined).
//
// Nothing here knows about alignments: bytes in, bytes out.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>  // crc32_combine

#include "../../include/woltka_hip.h"

namespace {

constexpr int kLitBits = 11, kDistBits = 8, kClBits = 7;
constexpr uint32_t F_LIT = 0x8000u, F_EOB = 0x4000u, F_SUB = 0x2000u;
constexpr int kWin = 32768;
constexpr size_t kOutMargin = 320;  // free cells a symbol may need (258 + a copy's overshoot)

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t rev_bits(uint32_t c, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) {
        r = (r << 1) | (c & 1u);
        c >>= 1;
    }
    return r;
}

// entry: bits 0-5 codeword length (SUB: index bits of the second level), bits 8-12 extra bits,
// bit 13 SUB, bit 14 EOB, bit 15 LIT, bits 16-31 value (literal / base / start of the second level);
// length 0 without SUB: no such code (or a symbol the format does not define)
inline uint32_t lit_payload(int s) {
    if (s < 256) return F_LIT | ((uint32_t)s << 16);
    if (s == 256) return F_EOB;
    if (s < 286) return ((uint32_t)kLenBase[s - 257] << 16) | ((uint32_t)kLenExtra[s - 257] << 8);
    return 0xFFFFFFFFu;  // 286, 287: take part in the code, may not appear
}
inline uint32_t dist_payload(int s) {
    if (s < 30) return ((uint32_t)kDistBase[s] << 16) | ((uint32_t)kDistExtra[s] << 8);
    return 0xFFFFFFFFu;
}

// Canonical Huffman code of `lens` as a two-level table.  false: over-subscribed, or incomplete in a way
// zlib refuses (anything but a single code of one bit), or no code at all when `need_one`.
template <typename Payload>
bool build_table(uint32_t* table, int cap, int P, const uint8_t* lens, int n, Payload payload, bool need_one) {
    int count[16] = {0};
    for (int s = 0; s < n; ++s) ++count[lens[s]];
    count[0] = 0;
    int total = 0, left = 1, maxlen = 0;
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return false;
        total += count[l];
        if (count[l]) maxlen = l;
    }
    std::memset(table, 0, sizeof(uint32_t) << P);
    if (total == 0) return !need_one;
    if (left > 0 && !(total == 1 && maxlen == 1)) return false;
    uint32_t next[16];
    uint32_t code = 0;
    for (int l = 1; l <= 15; ++l) {
        code = (code + (uint32_t)count[l - 1]) << 1;
        next[l] = code;
    }
    const uint32_t pmask = (1u << P) - 1u;
    uint8_t submax_small[1 << kLitBits];
    const bool any_long = maxlen > P;
    if (any_long) std::memset(submax_small, 0, (size_t)1 << P);
    uint32_t codes[320];
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t r = rev_bits(next[l]++, l);
        codes[s] = r;
        if (l <= P) {
            uint32_t e = payload(s);
            e = e == 0xFFFFFFFFu ? 0u : (e | (uint32_t)l);
            for (uint32_t i = r; i <= pmask; i += 1u << l) table[i] = e;
        } else {
            uint8_t& m = submax_small[r & pmask];
            if (l - P > m) m = (uint8_t)(l - P);
        }
    }
    if (!any_long) return true;
    int free_at = 1 << P;
    for (uint32_t p = 0; p <= pmask; ++p) {
        const int m = submax_small[p];
        if (!m) continue;
        if (free_at + (1 << m) > cap) return false;
        std::memset(table + free_at, 0, sizeof(uint32_t) << m);
        table[p] = F_SUB | (uint32_t)m | ((uint32_t)free_at << 16);
        free_at += 1 << m;
    }
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (l <= P) continue;
        const uint32_t r = codes[s];
        const uint32_t head = table[r & pmask];
        const int m = (int)(head & 0x3Fu);
        uint32_t* sub = table + (head >> 16);
        uint32_t e = payload(s);
        e = e == 0xFFFFFFFFu ? 0u : (e | (uint32_t)l);
        for (uint32_t i = r >> P; i < (1u << m); i += 1u << (l - P)) sub[i] = e;
    }
    return true;
}

enum Ret { R_BLOCK = 0, R_NEED_OUT = 1, R_END = 2, R_ERR = 3, R_TRUNC = 4 };

struct Decoder {
    uint32_t lit[(1 << kLitBits) + 4608];
    uint32_t dst[(1 << kDistBits) + 3968];
    const uint8_t* base = nullptr;  // the mapped file
    const uint8_t* in = nullptr;
    const uint8_t* in_end = nullptr;
    uint64_t bb = 0;
    int bc = 0;  // valid bits in bb
    bool in_block = false, bfinal = false, final_done = false;
    int btype = 0;
    uint32_t stored_left = 0;
    int blocks_done = 0;

    void start(const uint8_t* file, size_t n, int64_t bit) {
        base = file;
        in_end = file + n;
        in = file + (bit >> 3);
        bb = 0;
        bc = 0;
        in_block = bfinal = final_done = false;
        stored_left = 0;
        blocks_done = 0;
        const int skip = (int)(bit & 7);
        if (skip) {
            if (in < in_end) {
                bb = (uint64_t)*in++ >> skip;
                bc = 8 - skip;
            }
        }
    }
    inline int64_t bitpos() const { return (int64_t)(in - base) * 8 - bc; }
    inline void refill_slow() {
        while (bc <= 56 && in < in_end) {
            bb |= (uint64_t)*in++ << bc;
            bc += 8;
        }
    }
    inline void refill() {
        if (in_end - in >= 8) {
            uint64_t w;
            std::memcpy(&w, in, 8);
            bb |= w << bc;
            in += (63 - bc) >> 3;
            bc |= 56;
        } else {
            refill_slow();
        }
    }
    // n <= 32 bits, or -1 when the input ends first
    inline int64_t take(int n) {
        if (bc < n) {
            refill();
            if (bc < n) return -1;
        }
        const uint64_t v = bb & ((1ull << n) - 1ull);
        bb >>= n;
        bc -= n;
        return (int64_t)v;
    }

    bool fixed_tables() {
        uint8_t l[288];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        uint8_t d[32];
        for (int i = 0; i < 32; ++i) d[i] = 5;
        return build_table(lit, (int)(sizeof lit / 4), kLitBits, l, 288, lit_payload, true) &&
               build_table(dst, (int)(sizeof dst / 4), kDistBits, d, 32, dist_payload, false);
    }

    // the header of a dynamic block (behind its three bits); R_BLOCK = tables built
    Ret dynamic_header() {
        const int64_t h = take(14);
        if (h < 0) return R_TRUNC;
        const int hlit = (int)(h & 31) + 257, hdist = (int)((h >> 5) & 31) + 1, hclen = (int)((h >> 10) & 15) + 4;
        if (hlit > 286 || hdist > 30) return R_ERR;
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        for (int i = 0; i < hclen; ++i) {
            const int64_t v = take(3);
            if (v < 0) return R_TRUNC;
            cl[order[i]] = (uint8_t)v;
        }
        uint32_t ct[1 << kClBits];
        if (!build_table(ct, 1 << kClBits, kClBits, cl, 19, [](int s) { return (uint32_t)s << 16; }, true)) return R_ERR;
        uint8_t lens[320];
        int i = 0;
        const int n = hlit + hdist;
        while (i < n) {
            if (bc < 14) {
                refill();
                if (bc < 1) return R_TRUNC;
            }
            const uint32_t e = ct[bb & ((1u << kClBits) - 1u)];
            const int l = (int)(e & 0x3Fu);
            if (!l) return R_ERR;
            if (l > bc) return R_TRUNC;
            bb >>= l;
            bc -= l;
            const int s = (int)(e >> 16);
            if (s < 16) {
                lens[i++] = (uint8_t)s;
                continue;
            }
            int rep, val = 0;
            if (s == 16) {
                if (i == 0) return R_ERR;
                val = lens[i - 1];
                if (bc < 2) return R_TRUNC;
                rep = 3 + (int)(bb & 3u);
                bb >>= 2;
                bc -= 2;
            } else if (s == 17) {
                if (bc < 3) return R_TRUNC;
                rep = 3 + (int)(bb & 7u);
                bb >>= 3;
                bc -= 3;
            } else {
                if (bc < 7) return R_TRUNC;
                rep = 11 + (int)(bb & 127u);
                bb >>= 7;
                bc -= 7;
            }
            if (i + rep > n) return R_ERR;
            while (rep--) lens[i++] = (uint8_t)val;
        }
        if (lens[256] == 0) return R_ERR;
        if (!build_table(lit, (int)(sizeof lit / 4), kLitBits, lens, hlit, lit_payload, true)) return R_ERR;
        if (!build_table(dst, (int)(sizeof dst / 4), kDistBits, lens + hlit, hdist, dist_payload, false)) return R_ERR;
        return R_BLOCK;
    }

    // Decode from the current position: whole blocks, until a block starts at or behind bit `stop`
    // (R_BLOCK), the output is nearly full (R_NEED_OUT: call again with room), or the stream's last
    // block has ended (R_END).  `floor`: the first cell a match may reach back to.
    template <typename T>
    Ret run(T*& out_ref, T* out_end, const T* floor, int64_t stop) {
        T* out = out_ref;
        Ret ret;
        for (;;) {
            if (!in_block) {
                if (final_done) {
                    ret = R_END;
                    break;
                }
                if (bitpos() >= stop) {
                    ret = R_BLOCK;
                    break;
                }
                const int64_t h = take(3);
                if (h < 0) {
                    ret = R_TRUNC;
                    break;
                }
                bfinal = (h & 1) != 0;
                btype = (int)(h >> 1);
                if (btype == 0) {
                    const int drop = bc & 7;
                    bb >>= drop;
                    bc -= drop;
                    const int64_t v = take(32);
                    if (v < 0) {
                        ret = R_TRUNC;
                        break;
                    }
                    const uint32_t len = (uint32_t)v & 0xFFFFu, nlen = (uint32_t)v >> 16;
                    if ((len ^ 0xFFFFu) != nlen) {
                        ret = R_ERR;
                        break;
                    }
                    stored_left = len;
                } else if (btype == 1) {
                    if (!fixed_tables()) {
                        ret = R_ERR;
                        break;
                    }
                } else if (btype == 2) {
                    const Ret r = dynamic_header();
                    if (r != R_BLOCK) {
                        ret = r;
                        break;
                    }
                } else {
                    ret = R_ERR;
                    break;
                }
                in_block = true;
            }
            if (btype == 0) {
                bool starved = false;
                while (stored_left) {
                    if (out_end - out < 1) break;
                    if (bc >= 8) {
                        *out++ = (T)(bb & 0xFFu);
                        bb >>= 8;
                        bc -= 8;
                        --stored_left;
                        continue;
                    }
                    // (bc == 0 here: the block was byte-aligned and whole bytes were taken; what the
                    // bit buffer holds above its count are bits of the byte at `in`, void once `in` moves)
                    bb = 0;
                    size_t k = std::min<size_t>(stored_left, (size_t)(in_end - in));
                    k = std::min<size_t>(k, (size_t)(out_end - out));
                    if (!k) {
                        starved = in >= in_end;
                        break;
                    }
                    if (sizeof(T) == 1) {
                        std::memcpy(out, in, k);
                    } else {
                        for (size_t i = 0; i < k; ++i) out[i] = (T)in[i];
                    }
                    out += k;
                    in += k;
                    stored_left -= (uint32_t)k;
                }
                if (stored_left) {
                    ret = starved ? R_TRUNC : R_NEED_OUT;
                    break;
                }
                in_block = false;
                ++blocks_done;
                if (bfinal) final_done = true;
                continue;
            }
            // Huffman-coded block
            constexpr uint32_t lmask = (1u << kLitBits) - 1u, dmask = (1u << kDistBits) - 1u;
            ret = R_BLOCK;
            bool eob = false;
            while (!eob) {
                if ((size_t)(out_end - out) < kOutMargin) {
                    ret = R_NEED_OUT;
                    break;
                }
                refill();
                uint32_t e = lit[bb & lmask];
                if (e & F_LIT) {
                    int n = (int)(e & 0x3Fu);
                    bb >>= n;
                    bc -= n;
                    *out++ = (T)(e >> 16);
                    e = lit[bb & lmask];
                    if (e & F_LIT) {
                        n = (int)(e & 0x3Fu);
                        bb >>= n;
                        bc -= n;
                        *out++ = (T)(e >> 16);
                        if (bc < 0) {
                            ret = R_TRUNC;
                            break;
                        }
                        continue;
                    }
                }
                if (e & F_SUB) e = lit[(e >> 16) + ((uint32_t)(bb >> kLitBits) & ((1u << (e & 0x3Fu)) - 1u))];
                int n = (int)(e & 0x3Fu);
                if (!n) {
                    ret = bc <= 0 ? R_TRUNC : R_ERR;
                    break;
                }
                if (e & F_LIT) {  // (a literal with a long code)
                    bb >>= n;
                    bc -= n;
                    *out++ = (T)(e >> 16);
                    if (bc < 0) {
                        ret = R_TRUNC;
                        break;
                    }
                    continue;
                }
                if (e & F_EOB) {
                    bb >>= n;
                    bc -= n;
                    if (bc < 0) {
                        ret = R_TRUNC;
                        break;
                    }
                    eob = true;
                    break;
                }
                int x = (int)((e >> 8) & 0x1Fu);
                const uint32_t len = (e >> 16) + ((uint32_t)(bb >> n) & ((1u << x) - 1u));
                bb >>= n + x;
                bc -= n + x;
                if (bc < 32) refill();
                uint32_t d = dst[bb & dmask];
                if (d & F_SUB) d = dst[(d >> 16) + ((uint32_t)(bb >> kDistBits) & ((1u << (d & 0x3Fu)) - 1u))];
                n = (int)(d & 0x3Fu);
                if (!n) {
                    ret = bc <= 0 ? R_TRUNC : R_ERR;
                    break;
                }
                x = (int)((d >> 8) & 0x1Fu);
                const uint32_t dist = (d >> 16) + ((uint32_t)(bb >> n) & ((1u << x) - 1u));
                bb >>= n + x;
                bc -= n + x;
                if (bc < 0) {
                    ret = R_TRUNC;
                    break;
                }
                if ((size_t)(out - floor) < dist) {
                    ret = R_ERR;
                    break;
                }
                const T* s = out - dist;
                T* const end = out + len;
                constexpr uint32_t W = 8 / sizeof(T);
                if (dist >= 2 * W) {
                    do {
                        std::memcpy(out, s, 16);
                        out += 2 * W;
                        s += 2 * W;
                    } while (out < end);
                } else if (dist >= W) {
                    do {
                        std::memcpy(out, s, 8);
                        out += W;
                        s += W;
                    } while (out < end);
                } else if (dist == 1) {
                    const T v = *s;
                    for (T* p = out; p < end; ++p) *p = v;
                } else {
                    do {
                        *out++ = *s++;
                    } while (out < end);
                }
                out = end;
            }
            if (!eob) break;
            in_block = false;
            ++blocks_done;
            if (bfinal) final_done = true;
        }
        out_ref = out;
        return ret;
    }
};

// ---- gzip member header (RFC 1952) ----------------------------------------------------
struct MemberHead {
    size_t size = 0;        // bytes of the header
    int64_t stated = -1;    // size of the whole member where a subfield states it ('BC' / 'WK')
};
// 0: a header; 1: not a gzip header; 2: truncated
int parse_member_head(const uint8_t* p, size_t n, MemberHead& h) {
    if (n < 10) return n >= 2 && (p[0] != 0x1f || p[1] != 0x8b) ? 1 : 2;
    if (p[0] != 0x1f || p[1] != 0x8b) return 1;
    if (p[2] != 8 || (p[3] & 0xE0)) return 1;
    const int flg = p[3];
    size_t at = 10;
    h.stated = -1;
    if (flg & 4) {
        if (at + 2 > n) return 2;
        const size_t xlen = p[at] | ((size_t)p[at + 1] << 8);
        at += 2;
        if (at + xlen > n) return 2;
        size_t q = at;
        while (q + 4 <= at + xlen) {
            const size_t sl = p[q + 2] | ((size_t)p[q + 3] << 8);
            if (q + 4 + sl > at + xlen) break;
            if (p[q] == 'B' && p[q + 1] == 'C' && sl == 2) h.stated = (int64_t)(p[q + 4] | ((uint32_t)p[q + 5] << 8)) + 1;
            if (p[q] == 'W' && p[q + 1] == 'K' && sl == 4) {
                uint32_t v;
                std::memcpy(&v, p + q + 4, 4);
                h.stated = (int64_t)v;
            }
            q += 4 + sl;
        }
        at += xlen;
    }
    if (flg & 8) {
        while (at < n && p[at]) ++at;
        if (at >= n) return 2;
        ++at;
    }
    if (flg & 16) {
        while (at < n && p[at]) ++at;
        if (at >= n) return 2;
        ++at;
    }
    if (flg & 2) {
        if (at + 2 > n) return 2;
        at += 2;
    }
    h.size = at;
    return 0;
}

uint32_t crc_bytes(const uint8_t* p, size_t n) { return wk_crc32(0, reinterpret_cast<const char*>(p), (int64_t)n); }

// a member that ended inside a chunk: where (cells of the chunk's output), and its trailer
struct MemberEnd {
    size_t out_at;
    uint32_t crc, isize;
};

// Storage of a chunk's output: never zeroed, taken from / given back to a process-wide pool (a wave
// of chunks is tens of MB each: fresh pages for every wave would cost as much as decoding into them).
struct RawPool {
    std::mutex mu;
    std::vector<std::pair<void*, size_t>> free_list;
    void* take(size_t bytes, size_t* got) {
        {
            std::lock_guard<std::mutex> l(mu);
            size_t best = free_list.size();
            for (size_t i = 0; i < free_list.size(); ++i)
                if (free_list[i].second >= bytes && (best == free_list.size() || free_list[i].second < free_list[best].second)) best = i;
            if (best < free_list.size()) {
                void* p = free_list[best].first;
                *got = free_list[best].second;
                free_list.erase(free_list.begin() + (long)best);
                return p;
            }
        }
        *got = bytes;
        return std::malloc(bytes);
    }
    void give(void* p, size_t bytes) {
        if (!p) return;
        std::lock_guard<std::mutex> l(mu);
        if (free_list.size() < 96) {
            free_list.emplace_back(p, bytes);
            return;
        }
        std::free(p);
    }
    ~RawPool() {
        for (auto& f : free_list) std::free(f.first);
    }
};
RawPool g_pool;

template <typename T>
struct Cells {
    T* p = nullptr;
    size_t n = 0, cap_bytes = 0;
    Cells() = default;
    Cells(const Cells&) = delete;
    Cells& operator=(const Cells&) = delete;
    ~Cells() { g_pool.give(p, cap_bytes); }
    T* data() { return p; }
    const T* data() const { return p; }
    size_t size() const { return n; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
    bool reserve_uninit(size_t count) {  // contents dropped
        if (count * sizeof(T) > cap_bytes) {
            g_pool.give(p, cap_bytes);
            p = static_cast<T*>(g_pool.take(count * sizeof(T), &cap_bytes));
        }
        n = p ? cap_bytes / sizeof(T) : 0;
        return p != nullptr;
    }
    bool grow(size_t count) {  // contents kept
        if (count * sizeof(T) <= cap_bytes) return true;
        void* q = std::realloc(p, count * sizeof(T));
        if (!q) return false;
        p = static_cast<T*>(q);
        cap_bytes = count * sizeof(T);
        n = count;
        return true;
    }
};

struct Chunk {
    int64_t start_bit = -1, end_bit = -1;
    bool known = false;         // decoded from a known window into bytes
    bool data_end = false;      // the file's data ended inside this chunk
    bool failed = false;
    std::string err;
    Cells<uint8_t> bytes;          // known: [kWin history | output]
    Cells<uint16_t> cells;         // else:  [kWin markers | output]
    size_t n_out = 0;
    std::vector<MemberEnd> ends;
    std::vector<uint8_t> window;   // the kWin bytes before this chunk (resolved), for cells
    std::vector<uint8_t> lut;      // cell -> byte: literals as they are, markers through `window`
    size_t taken = 0;              // cells already handed to the reader
    size_t ends_checked = 0;       // members of `ends` whose trailers the reader has compared
};

template <typename T>
struct Buf;
template <>
struct Buf<uint8_t> {
    static Cells<uint8_t>& of(Chunk& c) { return c.bytes; }
};
template <>
struct Buf<uint16_t> {
    static Cells<uint16_t>& of(Chunk& c) { return c.cells; }
};

inline bool quick_header_ok(const uint8_t* base, size_t n, int64_t bit) {
    const size_t byte = (size_t)(bit >> 3);
    if (byte + 8 > n) return false;
    uint64_t w;
    std::memcpy(&w, base + byte, 8);
    const uint32_t v = (uint32_t)(w >> (bit & 7));
    // BFINAL = 0, BTYPE = 2 (bits 1-2 = 10b), HLIT <= 29, HDIST <= 29
    return (v & 7u) == 4u && ((v >> 3) & 31u) <= 29u && ((v >> 8) & 31u) <= 29u;
}

struct Gunzip {
    int fd = -1;
    const uint8_t* file = nullptr;
    size_t size = 0;
    int threads = 1;
    std::string err;
    // position of the producer: the deflate data of the current member from `pos_bit`, the resolved
    // window before it (`hist` valid bytes: 0 at a member's start)
    std::thread producer;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::unique_ptr<Chunk>> ready;
    size_t ready_cells = 0;
    bool done = false, stop = false;
    std::string perr;
    // reader side
    uint32_t crc_run = 0;
    uint64_t len_run = 0;
    bool chain = false;  // members that state their size
    size_t chain_at = 0;
    size_t chunk_bytes = 1 << 20;

    ~Gunzip() {
        {
            std::lock_guard<std::mutex> l(mu);
            stop = true;
        }
        cv.notify_all();
        if (producer.joinable()) producer.join();
        if (file) munmap(const_cast<uint8_t*>(file), size);
        if (fd >= 0) close(fd);
    }
};

template <typename Fn>
void parallel_for(int n, int threads, Fn fn) {
    if (n <= 0) return;
    const int T = std::max(1, std::min(threads, n));
    if (T == 1) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&] {
            for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i);
        });
    for (auto& x : th) x.join();
}

// One chunk of a wave.  `start_bit` >= 0: decode from there with the window `hist_bytes` (known);
// else find a block start in [search_from, search_to) first.  Decodes until a block starts at or
// behind `limit_bit`, then on until `*next_start` (published by the next chunk; -2 = it found none).
template <typename T>
void decode_chunk(Gunzip* g, Chunk& c, int64_t start_bit, const uint8_t* hist, size_t hist_n, int64_t search_from, int64_t search_to,
                  int64_t limit_bit, std::atomic<int64_t>* my_start, std::atomic<int64_t>* next_start) {
    const uint8_t* file = g->file;
    const size_t n = g->size;
    Cells<T>& buf = Buf<T>::of(c);
    std::unique_ptr<Decoder> dp(new Decoder);
    Decoder& d = *dp;
    auto publish = [&](int64_t v) {
        if (my_start) my_start->store(v, std::memory_order_release);
    };
    int64_t from = search_from;
    for (;;) {  // (again from the next bit when a start turns out to be none)
        int64_t at = start_bit;
        if (at < 0) {
            at = -1;
            for (int64_t b = from; b < search_to; ++b)
                if (quick_header_ok(file, n, b)) {
                    d.start(file, n, b);
                    (void)d.take(3);
                    if (d.dynamic_header() == R_BLOCK) {
                        at = b;
                        break;
                    }
                }
            if (at < 0) {
                c.failed = true;
                publish(-2);
                return;
            }
        }
        // output: kWin cells of history, then the data
        const int64_t span = std::min<int64_t>(limit_bit, (int64_t)n * 8) - at;
        size_t cap = kWin + (size_t)std::max<int64_t>(0, span / 8) * 10 + (1 << 17);
        if (!buf.reserve_uninit(cap)) {
            c.failed = true;
            c.err = "out of memory";
            publish(-2);
            return;
        }
        if (sizeof(T) == 1) {
            std::memset(buf.data(), 0, kWin - hist_n);
            if (hist_n) std::memcpy(reinterpret_cast<uint8_t*>(buf.data()) + (kWin - hist_n), hist, hist_n);
        } else {
            for (int i = 0; i < kWin; ++i) buf[i] = (T)(0x8000u | (uint32_t)i);
        }
        d.start(file, n, at);
        c.ends.clear();
        size_t floor_at = sizeof(T) == 1 ? kWin - hist_n : 0;
        T* out = buf.data() + kWin;
        bool bad_start = false, published = start_bit >= 0;
        int64_t stop = limit_bit;
        for (;;) {
            T* out_end = buf.data() + buf.size();
            const Ret r = d.run<T>(out, out_end, buf.data() + floor_at, stop);
            if (r == R_NEED_OUT) {
                const size_t used = (size_t)(out - buf.data());
                if (!buf.grow(buf.size() * 2)) {
                    c.failed = true;
                    c.err = "out of memory";
                    break;
                }
                out = buf.data() + used;
                continue;
            }
            if (!published && (d.blocks_done >= 1 || r == R_BLOCK || r == R_END)) {
                // the first block ended where another header parsed (or the stream ended): a start
                publish(at);
                published = true;
            }
            if (r == R_ERR || r == R_TRUNC) {
                if (!published) {
                    bad_start = true;
                    break;
                }
                c.failed = true;
                c.err = r == R_ERR ? "invalid deflate data" : "the compressed data end inside a block";
                break;
            }
            if (r == R_END) {
                // member trailer, then another member or the end of the data
                const int drop = d.bc & 7;
                d.bb >>= drop;
                d.bc -= drop;
                const uint8_t* p = d.in - d.bc / 8;
                if ((size_t)(file + n - p) < 8) {
                    c.failed = true;
                    c.err = "the gzip trailer is missing";
                    break;
                }
                MemberEnd me;
                me.out_at = (size_t)(out - (buf.data() + kWin));
                std::memcpy(&me.crc, p, 4);
                std::memcpy(&me.isize, p + 4, 4);
                c.ends.push_back(me);
                p += 8;
                // (zero padding behind a member is skipped like gzip does)
                const uint8_t* q = p;
                while (q < file + n && *q == 0) ++q;
                MemberHead mh;
                const int hr = q < file + n ? parse_member_head(q, (size_t)(file + n - q), mh) : 1;
                if (hr != 0) {
                    // (a header cut short, or bytes that are no header: trailing garbage is ignored, as
                    // `gzip -cdfq` does with a warning)
                    c.data_end = true;
                    c.end_bit = (int64_t)n * 8;
                    break;
                }
                const int64_t nb = (int64_t)(q + mh.size - file) * 8;
                d.start(file, n, nb);
                floor_at = (size_t)(out - buf.data());
                // (a new member is a block start like any other for the chunk behind)
                continue;
            }
            // R_BLOCK: at a block start at or behind `stop`
            const int64_t here = d.bitpos();
            if (!next_start) {
                c.end_bit = here;
                break;
            }
            int64_t ns = next_start->load(std::memory_order_acquire);
            while (ns == -1) {
                std::this_thread::yield();
                ns = next_start->load(std::memory_order_acquire);
            }
            if (ns == -2 || here >= ns) {
                c.end_bit = here;  // (met, missed, or nothing to meet)
                break;
            }
            stop = ns;
        }
        if (bad_start) {
            if (start_bit >= 0) {
                c.failed = true;
                c.err = "invalid deflate data";
                return;
            }
            from = at + 1;
            continue;
        }
        if (!published) publish(-2);
        c.start_bit = at;
        c.n_out = (size_t)(out - (buf.data() + kWin));
        return;
    }
}

// The producer: wave after wave until the data end.
void produce(Gunzip* g) {
    const uint8_t* file = g->file;
    const size_t n = g->size;
    std::vector<uint8_t> window(kWin, 0);
    size_t hist = 0;
    auto fail = [&](const std::string& m) {
        std::lock_guard<std::mutex> l(g->mu);
        g->perr = m;
        g->done = true;
        g->cv.notify_all();
    };
    MemberHead mh;
    const int hr = parse_member_head(file, n, mh);
    if (hr != 0) return fail(hr == 1 ? "not a gzip file" : "the gzip header is cut short");
    int64_t pos = (int64_t)mh.size * 8;
    // (two threads decode less than one: every chunk but a wave's first is decoded into cells and
    // resolved, about twice the work per byte)
    const int T = g->threads < 3 ? 1 : g->threads;
    for (;;) {
        {
            std::unique_lock<std::mutex> l(g->mu);
            g->cv.wait(l, [&] { return g->stop || g->ready_cells < (size_t)T * g->chunk_bytes * 16; });
            if (g->stop) return;
        }
        const int64_t base_byte = pos >> 3;
        const int64_t left = (int64_t)n - base_byte;
        // a wave: three chunks per thread, taken in order (a thread that reaches its chunk's end waits for
        // the start the next chunk finds -- always a chunk some thread has taken before; with one thread
        // there is nothing to wait for: one chunk)
        const int64_t want = T == 1 ? 1 : 3 * (int64_t)T;
        int K = (int)std::max<int64_t>(1, std::min<int64_t>(want, (left + (int64_t)g->chunk_bytes - 1) / (int64_t)g->chunk_bytes));
        std::vector<std::unique_ptr<Chunk>> wave(K);
        std::vector<std::atomic<int64_t>> starts(K + 1);
        for (auto& s : starts) s.store(-1);
        starts[0].store(pos);
        starts[K].store(-2);
        for (int k = 0; k < K; ++k) wave[k].reset(new Chunk);
        const std::vector<uint8_t> win0 = window;
        const size_t hist0 = hist;
        parallel_for(K, T, [&](int k) {
            const int64_t lo = (base_byte + (int64_t)k * (int64_t)g->chunk_bytes) * 8;
            const int64_t hi = std::min<int64_t>((int64_t)n * 8, (base_byte + (int64_t)(k + 1) * (int64_t)g->chunk_bytes) * 8);
            Chunk& c = *wave[k];
            // (a chunk stops at the first block start at or behind its end -- never, when that is the file's)
            const int64_t limit = hi >= (int64_t)n * 8 ? (int64_t)n * 8 + 64 : hi;
            if (k == 0) {
                c.known = true;
                decode_chunk<uint8_t>(g, c, pos, win0.data() + (kWin - hist0), hist0, 0, 0, limit, nullptr, k + 1 < K ? &starts[1] : nullptr);
            } else {
                decode_chunk<uint16_t>(g, c, -1, nullptr, 0, lo, hi, limit, &starts[k], k + 1 < K ? &starts[k + 1] : nullptr);
            }
        });
        // stitch: chunk k + 1 counts when it starts where chunk k ended
        int good = 0;
        for (int k = 0; k < K; ++k) {
            Chunk& c = *wave[k];
            if (c.failed) {
                if (k == 0) return fail(c.err.empty() ? "invalid deflate data" : c.err);
                break;
            }
            if (k > 0 && c.start_bit != wave[k - 1]->end_bit) break;
            good = k + 1;
            if (c.data_end) break;
        }
        // windows in order; the position behind the last good chunk
        bool data_end = false;
        for (int k = 0; k < good; ++k) {
            Chunk& c = *wave[k];
            // a member that ended inside: history restarts there
            size_t keep_from = 0;  // cells of this chunk's output a later window may not reach before
            if (!c.ends.empty()) keep_from = c.ends.back().out_at;
            if (!c.known) {
                c.window = window;
                c.lut.resize(65536);
                for (int i = 0; i < 256; ++i) c.lut[i] = (uint8_t)i;
                std::memset(c.lut.data() + 256, 0, 0x8000 - 256);
                std::memcpy(c.lut.data() + 0x8000, window.data(), kWin);
            }
            // the kWin bytes before the next chunk = tail of (window ++ output)
            const size_t no = c.n_out;
            std::vector<uint8_t> nw(kWin, 0);
            const size_t take = std::min<size_t>(no, kWin);
            for (size_t i = 0; i < take; ++i) {
                const size_t at = no - take + i;
                uint8_t v;
                if (c.known) {
                    v = c.bytes[kWin + at];
                } else {
                    const uint16_t x = c.cells[kWin + at];
                    v = (x & 0x8000u) ? window[x & 0x7FFFu] : (uint8_t)x;
                }
                nw[kWin - take + i] = v;
            }
            if (take < (size_t)kWin) std::memcpy(nw.data(), window.data() + take, kWin - take);
            hist = c.ends.empty() ? std::min<size_t>(kWin, hist + no) : std::min<size_t>(kWin, no - keep_from);
            window.swap(nw);
            pos = c.end_bit;
            data_end = c.data_end;
        }
        {
            std::lock_guard<std::mutex> l(g->mu);
            for (int k = 0; k < good; ++k) {
                g->ready_cells += wave[k]->n_out;
                g->ready.push_back(std::move(wave[k]));
            }
            if (data_end) g->done = true;
            g->cv.notify_all();
        }
        if (data_end) return;
        if (pos >= (int64_t)n * 8) return fail("the compressed data end inside a member");
        // (a member that ended exactly at the wave's end: `pos` is the next member's first block; hist is 0)
    }
}

}  // namespace

struct wk_gunzip {
    Gunzip g;
};

extern "C" {

// Open `path` (a regular gzip file) for inflating on up to `n_threads` threads.  NULL on failure
// (`err`, if given, says why): not a regular file, not gzip.  (file.readzip, woltka/file.py:62-129.)
wk_gunzip* wk_gunzip_open(const char* path, int n_threads, char* err, size_t err_cap) {
    auto say = [&](const char* m) {
        if (err && err_cap) snprintf(err, err_cap, "%s", m);
        return (wk_gunzip*)nullptr;
    };
    if (!path) return say("no path");
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return say("cannot open the file");
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 18) {
        close(fd);
        return say("not a regular gzip file");
    }
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) {
        close(fd);
        return say("cannot map the file");
    }
    const uint8_t* p = static_cast<const uint8_t*>(m);
    MemberHead mh;
    if (parse_member_head(p, (size_t)st.st_size, mh) != 0) {
        munmap(m, (size_t)st.st_size);
        close(fd);
        return say("not a gzip file");
    }
    wk_gunzip* h = new wk_gunzip;
    Gunzip& g = h->g;
    g.fd = fd;
    g.file = p;
    g.size = (size_t)st.st_size;
    g.threads = std::max(1, std::min(n_threads, 64));
    (void)madvise(m, g.size, MADV_SEQUENTIAL);
    // chunks: a wave of `threads` of them covers <= 1/2 of a small file, 1 MB of compressed bytes at most
    g.chunk_bytes = std::max<size_t>(1 << 16, std::min<size_t>(g.threads == 1 ? 1 << 20 : 1 << 19, g.size / (size_t)(6 * g.threads) + 1));
    if (mh.stated > 0) {
        g.chain = true;
        g.chain_at = 0;
    } else {
        g.producer = std::thread(produce, &g);
    }
    return h;
}

const char* wk_gunzip_error(wk_gunzip* h) { return h ? h->g.err.c_str() : "no handle"; }

void wk_gunzip_close(wk_gunzip* h) { delete h; }

// The next bytes of the inflated text into dst[0, cap): returns their number (0 at the end of the
// data), or -1 (wk_gunzip_error says why: damaged data, a CRC-32 or a size that does not match).
// Several threads resolve / inflate into `dst` at once.  cap >= 64 KB.
int64_t wk_gunzip_read(wk_gunzip* h, char* dst_c, int64_t cap) {
    if (!h || !dst_c || cap < (1 << 16)) return -1;
    Gunzip& g = h->g;
    uint8_t* dst = reinterpret_cast<uint8_t*>(dst_c);
    if (!g.err.empty()) return -1;
    if (g.chain) {
        // members that state their size: as many as fit, one task each
        struct Task {
            size_t at, size, head;
            uint32_t isize;
            size_t out;
        };
        std::vector<Task> tasks;
        size_t fill = 0, at = g.chain_at;
        while (at < g.size) {
            while (at < g.size && g.file[at] == 0) ++at;
            if (at >= g.size) break;
            MemberHead mh;
            const int hr = parse_member_head(g.file + at, g.size - at, mh);
            if (hr == 1) {
                at = g.size;  // trailing garbage
                break;
            }
            if (hr == 2 || mh.stated < (int64_t)mh.size + 8 || at + (size_t)mh.stated > g.size) {
                if (fill == 0) {  // (nothing to hand out but empty members: 0 bytes would read as the end of the data)
                    if (hr == 0 && mh.stated < 0) {
                        g.err = "a gzip member without a stated size inside a chain of members that state theirs";
                        return -1;
                    }
                    g.err = "a gzip member is cut short";
                    return -1;
                }
                break;
            }
            uint32_t isize;
            std::memcpy(&isize, g.file + at + mh.stated - 4, 4);
            if (fill + isize > (size_t)cap) {
                if (fill == 0) {
                    g.err = "a gzip member larger than the read buffer";
                    return -1;
                }
                break;
            }
            tasks.push_back(Task{at, (size_t)mh.stated, mh.size, isize, fill});
            fill += isize;
            at += (size_t)mh.stated;
        }
        std::atomic<int> bad{-1};
        parallel_for((int)tasks.size(), g.threads, [&](int i) {
            const Task& t = tasks[i];
            std::unique_ptr<Decoder> d(new Decoder);
            // (inflated in place: the member's own bytes are its only history)
            d->start(g.file, t.at + t.size - 8, (int64_t)(t.at + t.head) * 8);
            // the decoder wants a margin behind its output: the last symbols go through a bounce buffer
            uint8_t* out = dst + t.out;
            uint8_t* const want_end = out + t.isize;
            bool ok = true;
            Ret r = R_NEED_OUT;
            if ((size_t)t.isize > kOutMargin) r = d->run<uint8_t>(out, want_end, dst + t.out, INT64_MAX);
            if (r == R_NEED_OUT) {
                // tail: decode into a scratch that carries the last kWin bytes as history
                const size_t have = (size_t)(out - (dst + t.out));
                const size_t hist = std::min<size_t>(have, kWin);
                std::vector<uint8_t> tmp(kWin + kOutMargin * 4 + 66000);
                std::memcpy(tmp.data() + kWin - hist, out - hist, hist);
                uint8_t* o2 = tmp.data() + kWin;
                for (;;) {
                    r = d->run<uint8_t>(o2, tmp.data() + tmp.size(), tmp.data() + kWin - hist, INT64_MAX);
                    if (r != R_NEED_OUT) break;
                    const size_t used = (size_t)(o2 - tmp.data());
                    if (used > kWin + (size_t)t.isize + 66000) break;  // more than the trailer says
                    tmp.resize(tmp.size() * 2);
                    o2 = tmp.data() + used;
                }
                const size_t more = (size_t)(o2 - (tmp.data() + kWin));
                if (r != R_END || have + more != t.isize) {
                    ok = false;
                } else {
                    std::memcpy(out, tmp.data() + kWin, more);
                    out += more;
                }
            } else if (r != R_END || out != want_end) {
                ok = false;
            }
            if (ok) {
                uint32_t crc;
                std::memcpy(&crc, g.file + t.at + t.size - 8, 4);
                ok = crc == crc_bytes(dst + t.out, t.isize);
            }
            if (!ok) {
                int cur = bad.load();
                while ((cur < 0 || i < cur) && !bad.compare_exchange_weak(cur, i)) {
                }
            }
        });
        if (bad.load() >= 0) {
            g.err = "a gzip member does not inflate to what its trailer says (size, CRC-32)";
            return -1;
        }
        g.chain_at = at;
        return (int64_t)fill;
    }
    // one stream: pieces of the ready chunks, resolved in parallel
    struct Piece {
        Chunk* c;
        size_t a, b, out;
        uint32_t crc;
    };
    std::vector<Piece> pieces;
    std::vector<std::unique_ptr<Chunk>> hold;
    size_t fill = 0;
    while (fill < (size_t)cap) {
        std::unique_lock<std::mutex> l(g.mu);
        g.cv.wait(l, [&] { return !g.ready.empty() || g.done; });
        if (g.ready.empty()) {
            if (!g.perr.empty() && fill == 0) {
                g.err = g.perr;
                return -1;
            }
            break;
        }
        Chunk* c = g.ready.front().get();
        const size_t room = (size_t)cap - fill, left = c->n_out - c->taken;
        const size_t k = std::min(room, left);
        // cut at the ends of members (their CRCs are checked piece by piece) and into pieces of <= 4 MB
        size_t a = c->taken;
        const size_t stop_at = a + k;
        while (a < stop_at) {
            size_t b = std::min(stop_at, a + ((size_t)4 << 20));
            for (const MemberEnd& me : c->ends)
                if (me.out_at > a && me.out_at < b) b = me.out_at;
            pieces.push_back(Piece{c, a, b, fill + (a - c->taken), 0});
            a = b;
        }
        fill += k;
        c->taken += k;
        if (c->taken == c->n_out) {
            g.ready_cells -= c->n_out;
            hold.push_back(std::move(g.ready.front()));
            g.ready.pop_front();
            g.cv.notify_all();
        } else {
            break;  // the buffer is full
        }
        if (fill >= (size_t)cap / 2 && g.ready.empty()) break;  // (do not wait for the producer with a buffer half full)
    }
    parallel_for((int)pieces.size(), g.threads, [&](int i) {
        Piece& p = pieces[i];
        uint8_t* o = dst + p.out;
        if (p.c->known) {
            std::memcpy(o, p.c->bytes.data() + kWin + p.a, p.b - p.a);
        } else {
            // (without a branch per cell: markers and literals alternate unpredictably in text)
            const uint16_t* s = p.c->cells.data() + kWin + p.a;
            const uint8_t* lut = p.c->lut.data();
            const size_t m = p.b - p.a;
            size_t j = 0;
            for (; j + 4 <= m; j += 4) {
                const uint8_t a0 = lut[s[j]], a1 = lut[s[j + 1]], a2 = lut[s[j + 2]], a3 = lut[s[j + 3]];
                o[j] = a0;
                o[j + 1] = a1;
                o[j + 2] = a2;
                o[j + 3] = a3;
            }
            for (; j < m; ++j) o[j] = lut[s[j]];
        }
        p.crc = crc_bytes(o, p.b - p.a);
    });
    // CRCs in order; the members that ended at a piece's edge are checked against their trailers
    bool crc_ok = true;
    auto check_ends = [&](Chunk* c, size_t upto) {
        while (c->ends_checked < c->ends.size() && c->ends[c->ends_checked].out_at <= upto) {
            const MemberEnd& me = c->ends[c->ends_checked++];
            if (me.crc != g.crc_run || me.isize != (uint32_t)g.len_run) crc_ok = false;
            g.crc_run = 0;
            g.len_run = 0;
        }
    };
    for (const Piece& p : pieces) {
        check_ends(p.c, p.a);
        g.crc_run = (uint32_t)crc32_combine(g.crc_run, p.crc, (z_off_t)(p.b - p.a));
        g.len_run += p.b - p.a;
        check_ends(p.c, p.b);
    }
    for (auto& c : hold) check_ends(c.get(), c->n_out);  // (chunks without output)
    if (!crc_ok) {
        g.err = "CRC check failed in a gzip member";
        return -1;
    }
    return (int64_t)fill;
}

}  // extern "C"
