// wk_deflate.cpp — gzip members of read-map text, fast (host).
//
// `--outmap` (file.write_readmap, woltka/file.py:469-500, through
// file.openzip: gzip by default, cli.py:173-176) writes one line per read;
// config 5's first pass writes 20 M of them per sample.  With the lines
// formatted on the device (wk_readmap.hpp) the compressor is what is left on
// the host's CPUs, and zlib at level 4 spends ~10 ns per byte there.  This is
// a one-pass LZ77 (4-byte hash, one candidate, greedy) with dynamic Huffman
// codes per block of 32 k tokens — the shape of the fastest levels of the
// usual deflate libraries — written for this text: lines that repeat most of
// the line before.  The output is a standard gzip member (RFC 1952 / 1951);
// its size is carried in the 'WK' extra subfield like pgzip.member's, so the
// second pass inflates the members of a map in parallel.  Only the bytes
// differ from what zlib would write; any gzip reader reads them.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include <immintrin.h>
#include <zlib.h>

#include <atomic>
#include <thread>

#include "../../include/woltka_hip.h"

namespace {

// ---- CRC-32 (gzip polynomial) ------------------------------------------------
struct CrcTables {
    uint32_t t[8][256];
    CrcTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFFu];
    }
};
const CrcTables kCrc;

uint32_t crc32_slice8(uint32_t crc, const unsigned char* p, size_t n) {
    crc = ~crc;
    while (n >= 8) {
        uint64_t v;
        std::memcpy(&v, p, 8);
        v ^= crc;
        crc = kCrc.t[7][v & 0xFF] ^ kCrc.t[6][(v >> 8) & 0xFF] ^ kCrc.t[5][(v >> 16) & 0xFF] ^ kCrc.t[4][(v >> 24) & 0xFF] ^
              kCrc.t[3][(v >> 32) & 0xFF] ^ kCrc.t[2][(v >> 40) & 0xFF] ^ kCrc.t[1][(v >> 48) & 0xFF] ^ kCrc.t[0][v >> 56];
        p += 8;
        n -= 8;
    }
    while (n--) crc = kCrc.t[0][(crc ^ *p++) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}

// Carry-less-multiply folding, 64 bytes per round (Gopal et al., "Fast CRC
// computation for generic polynomials using PCLMULQDQ"; the constants are x^n
// mod P for the reflected gzip polynomial).
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_clmul(uint32_t crc, const unsigned char* p, size_t n) {
    if (n < 64) return crc32_slice8(crc, p, n);
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5k0 = _mm_set_epi64x(0x0000000000ll, 0x0163cd6124ll);
    const __m128i poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    __m128i x1 = _mm_loadu_si128((const __m128i*)(p + 0x00));
    __m128i x2 = _mm_loadu_si128((const __m128i*)(p + 0x10));
    __m128i x3 = _mm_loadu_si128((const __m128i*)(p + 0x20));
    __m128i x4 = _mm_loadu_si128((const __m128i*)(p + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)~crc));
    p += 64;
    n -= 64;
    while (n >= 64) {
        __m128i x5 = _mm_clmulepi64_si128(x1, k1k2, 0x00);
        __m128i x6 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        __m128i x7 = _mm_clmulepi64_si128(x3, k1k2, 0x00);
        __m128i x8 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11);
        x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11);
        x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), _mm_loadu_si128((const __m128i*)(p + 0x00)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), _mm_loadu_si128((const __m128i*)(p + 0x10)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), _mm_loadu_si128((const __m128i*)(p + 0x20)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), _mm_loadu_si128((const __m128i*)(p + 0x30)));
        p += 64;
        n -= 64;
    }
    // fold the four lanes into one
    __m128i x5 = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (n >= 16) {
        x2 = _mm_loadu_si128((const __m128i*)p);
        x5 = _mm_clmulepi64_si128(x1, k3k4, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        p += 16;
        n -= 16;
    }
    // 128 -> 64 bits
    x2 = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    const __m128i mask32 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, mask32);
    x1 = _mm_clmulepi64_si128(x1, k5k0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    // Barrett reduction 64 -> 32 bits
    x2 = _mm_and_si128(x1, mask32);
    x2 = _mm_clmulepi64_si128(x2, poly, 0x10);
    x2 = _mm_and_si128(x2, mask32);
    x2 = _mm_clmulepi64_si128(x2, poly, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    uint32_t c = ~(uint32_t)_mm_extract_epi32(x1, 1);
    return n ? crc32_slice8(c, p, n) : c;
}

bool have_clmul() {
    static const bool yes = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    return yes;
}

// ---- deflate -----------------------------------------------------------------
struct BitWriter {
    unsigned char* p;
    unsigned char* end;
    uint64_t acc = 0;
    int n = 0;
    bool overflow = false;
    inline void put(uint32_t code, int bits) {  // bits <= 32
        acc |= (uint64_t)code << n;
        n += bits;
        if (n >= 32) {
            if (p + 4 <= end) {
                std::memcpy(p, &acc, 4);
                p += 4;
            } else {
                overflow = true;
            }
            acc >>= 32;
            n -= 32;
        }
    }
    inline void align() {
        while (n > 0) {
            if (p < end)
                *p++ = (unsigned char)acc;
            else
                overflow = true;
            acc >>= 8;
            n -= 8;
        }
        acc = 0;
        n = 0;
    }
};

constexpr int kLitLen = 286, kDist = 30, kCodeLen = 19;
constexpr int kMaxTokens = 1 << 15;
constexpr int kHashBits = 15;

struct LenCode {
    unsigned char sym[259];    // length -> symbol - 257
    unsigned char extra[29];
    uint16_t base[29];
    unsigned char dsym[512];   // distance -> symbol (two-level, like zlib's d_code)
    unsigned char dextra[30];
    uint16_t dbase[30];
    LenCode() {
        static const int lb[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const int le[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        for (int s = 0; s < 29; ++s) {
            base[s] = (uint16_t)lb[s];
            extra[s] = (unsigned char)le[s];
        }
        for (int len = 3; len <= 258; ++len) {
            int s = 28;
            while (lb[s] > len) --s;
            sym[len] = (unsigned char)s;
        }
        int d = 1;
        for (int s = 0; s < 30; ++s) {
            const int e = s < 2 ? 0 : (s >> 1) - 1;
            dextra[s] = (unsigned char)e;
            dbase[s] = (uint16_t)d;
            d += 1 << e;
        }
        for (int dist = 1; dist <= 256; ++dist) {
            int s = 29;
            while (dbase[s] > dist) --s;
            dsym[dist - 1] = (unsigned char)s;
        }
        for (int hi = 2; hi < 256; ++hi) {  // distances 257..32768 by (dist - 1) >> 7
            const int dist = (hi << 7) + 1;
            int s = 29;
            while (dbase[s] > dist) --s;
            dsym[256 + hi] = (unsigned char)s;
        }
        dsym[256] = dsym[257] = 0;  // (never indexed: (dist - 1) >> 7 >= 2 there)
    }
    inline int dist_sym(uint32_t dist) const { return dist <= 256 ? dsym[dist - 1] : dsym[256 + ((dist - 1) >> 7)]; }
};
const LenCode kLen;

// Huffman code lengths (<= limit) of `n` symbols by frequency: the two-queue
// construction over the sorted symbols; a tree deeper than the limit is rebuilt
// from halved frequencies (a block has at most 32 k tokens, so this is rare).
void huff_lengths(const uint32_t* freq_in, int n, int limit, unsigned char* len) {
    uint32_t freq[kLitLen];
    for (int i = 0; i < n; ++i) freq[i] = freq_in[i];
    for (;;) {
        int order[kLitLen], m = 0;
        for (int i = 0; i < n; ++i) {
            len[i] = 0;
            if (freq[i]) order[m++] = i;
        }
        if (m == 0) return;
        if (m == 1) {
            len[order[0]] = 1;
            return;
        }
        std::sort(order, order + m, [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
        // nodes: leaves 0..m-1 (sorted), internal m..2m-2
        uint64_t w[2 * kLitLen];
        int parent[2 * kLitLen];
        for (int i = 0; i < m; ++i) w[i] = freq[order[i]];
        int leaf = 0, inner = m, next = m;
        auto take = [&]() {
            if (leaf < m && (inner >= next || w[leaf] <= w[inner])) return leaf++;
            return inner++;
        };
        while (next < 2 * m - 1) {
            const int a = take(), b = take();
            w[next] = w[a] + w[b];
            parent[a] = parent[b] = next;
            ++next;
        }
        int depth[2 * kLitLen];
        depth[2 * m - 2] = 0;
        int deepest = 0;
        for (int i = 2 * m - 3; i >= 0; --i) {
            depth[i] = depth[parent[i]] + 1;
            if (i < m && depth[i] > deepest) deepest = depth[i];
        }
        if (deepest <= limit) {
            for (int i = 0; i < m; ++i) len[order[i]] = (unsigned char)depth[i];
            return;
        }
        for (int i = 0; i < n; ++i)
            if (freq[i]) freq[i] = (freq[i] + 1) >> 1;
        // (frequencies of 1 stay 1: the spread shrinks until the tree fits)
        bool flat = true;
        for (int i = 0; i < n; ++i) flat &= freq[i] <= 1;
        if (flat) {  // all equal and still too deep cannot happen for n <= 2^limit
            for (int i = 0; i < n; ++i) len[i] = freq[i] ? (unsigned char)limit : 0;
            return;
        }
    }
}

inline uint32_t reverse_bits(uint32_t v, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) {
        r = (r << 1) | (v & 1u);
        v >>= 1;
    }
    return r;
}

// canonical codes (RFC 1951 3.2.2), bit-reversed for the LSB-first writer
void huff_codes(const unsigned char* len, int n, uint16_t* code) {
    int count[16] = {0};
    for (int i = 0; i < n; ++i) ++count[len[i]];
    count[0] = 0;
    uint32_t next[16];
    uint32_t c = 0;
    for (int b = 1; b < 16; ++b) {
        c = (c + (uint32_t)count[b - 1]) << 1;
        next[b] = c;
    }
    for (int i = 0; i < n; ++i) code[i] = len[i] ? (uint16_t)reverse_bits(next[len[i]]++, len[i]) : 0;
}

struct Token {
    uint16_t len;   // 0: literal `dist`; else match length
    uint16_t dist;  // literal byte, or distance - 1
};

struct Deflater {
    std::vector<uint32_t> head;
    std::vector<Token> tok;
    Deflater() : head(1u << kHashBits), tok(kMaxTokens) {}

    void block(BitWriter& bw, const unsigned char* src, size_t from, size_t to, int n_tok, uint32_t* lfreq, uint32_t* dfreq,
               bool last) {
        lfreq[256] = 1;
        // (a decoder wants a complete code: at least two symbols per alphabet)
        int used = 0;
        for (int i = 0; i < kDist; ++i) used += dfreq[i] != 0;
        if (used < 2) {
            dfreq[0] += dfreq[0] ? 0 : 1;
            dfreq[1] += dfreq[1] ? 0 : 1;
        }
        unsigned char llen[kLitLen], dlen[kDist];
        huff_lengths(lfreq, kLitLen, 15, llen);
        huff_lengths(dfreq, kDist, 15, dlen);
        int hlit = kLitLen, hdist = kDist;
        while (hlit > 257 && llen[hlit - 1] == 0) --hlit;
        while (hdist > 1 && dlen[hdist - 1] == 0) --hdist;
        // the code lengths, run-length coded (16: repeat previous 3-6, 17: zeros 3-10, 18: zeros 11-138)
        unsigned char all[kLitLen + kDist];
        std::memcpy(all, llen, (size_t)hlit);
        std::memcpy(all + hlit, dlen, (size_t)hdist);
        const int n_all = hlit + hdist;
        struct Cl {
            unsigned char sym, extra;
        };
        Cl cl[kLitLen + kDist];
        int n_cl = 0;
        uint32_t cfreq[kCodeLen] = {0};
        for (int i = 0; i < n_all;) {
            int run = 1;
            while (i + run < n_all && all[i + run] == all[i]) ++run;
            const int v = all[i];
            int left = run;
            if (v == 0) {
                while (left >= 11) {
                    const int r = std::min(left, 138);
                    cl[n_cl++] = Cl{18, (unsigned char)(r - 11)};
                    ++cfreq[18];
                    left -= r;
                }
                if (left >= 3) {
                    cl[n_cl++] = Cl{17, (unsigned char)(left - 3)};
                    ++cfreq[17];
                    left = 0;
                }
            } else if (left >= 4) {
                cl[n_cl++] = Cl{(unsigned char)v, 0};
                ++cfreq[v];
                --left;
                while (left >= 3) {
                    const int r = std::min(left, 6);
                    cl[n_cl++] = Cl{16, (unsigned char)(r - 3)};
                    ++cfreq[16];
                    left -= r;
                }
            }
            while (left-- > 0) {
                cl[n_cl++] = Cl{(unsigned char)v, 0};
                ++cfreq[v];
            }
            i += run;
        }
        unsigned char clen[kCodeLen];
        huff_lengths(cfreq, kCodeLen, 7, clen);
        static const int kOrder[kCodeLen] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        int hclen = kCodeLen;
        while (hclen > 4 && clen[kOrder[hclen - 1]] == 0) --hclen;
        // cost of the dynamic block against a stored one
        uint64_t bits = 3 + 5 + 5 + 4 + 3ull * hclen;
        for (int i = 0; i < n_cl; ++i) bits += clen[cl[i].sym] + (cl[i].sym == 16 ? 2 : cl[i].sym == 17 ? 3 : cl[i].sym == 18 ? 7 : 0);
        for (int s = 0; s < kLitLen; ++s) bits += (uint64_t)lfreq[s] * (llen[s] + (s >= 257 ? kLen.extra[s - 257] : 0));
        for (int s = 0; s < kDist; ++s) bits += (uint64_t)dfreq[s] * (dlen[s] + kLen.dextra[s]);
        const size_t raw = to - from;
        if (bits / 8 + 1 > raw + 5 * (raw / 65535 + 1)) {
            size_t at = from;
            do {  // stored blocks of at most 65535 bytes (also for an empty last block)
                const size_t n = std::min<size_t>(65535, to - at);
                const bool fin = last && at + n == to;
                bw.put(fin ? 1u : 0u, 3);
                bw.align();
                bw.put((uint32_t)n, 16);
                bw.put((uint32_t)n ^ 0xFFFFu, 16);
                bw.align();
                if (bw.p + n <= bw.end) {
                    std::memcpy(bw.p, src + at, n);
                    bw.p += n;
                } else {
                    bw.overflow = true;
                }
                at += n;
            } while (at < to);
            return;
        }
        uint16_t lcode[kLitLen], dcode[kDist], ccode[kCodeLen];
        huff_codes(llen, kLitLen, lcode);
        huff_codes(dlen, kDist, dcode);
        huff_codes(clen, kCodeLen, ccode);
        bw.put((last ? 1u : 0u) | (2u << 1), 3);
        bw.put((uint32_t)(hlit - 257), 5);
        bw.put((uint32_t)(hdist - 1), 5);
        bw.put((uint32_t)(hclen - 4), 4);
        for (int i = 0; i < hclen; ++i) bw.put(clen[kOrder[i]], 3);
        for (int i = 0; i < n_cl; ++i) {
            bw.put(ccode[cl[i].sym], clen[cl[i].sym]);
            if (cl[i].sym == 16)
                bw.put(cl[i].extra, 2);
            else if (cl[i].sym == 17)
                bw.put(cl[i].extra, 3);
            else if (cl[i].sym == 18)
                bw.put(cl[i].extra, 7);
        }
        const Token* t = tok.data();
        for (int i = 0; i < n_tok; ++i) {
            if (t[i].len == 0) {
                bw.put(lcode[t[i].dist], llen[t[i].dist]);
            } else {
                const int ls = kLen.sym[t[i].len];
                // symbol + extra bits of the length in one put (<= 15 + 5 bits)
                bw.put((uint32_t)lcode[257 + ls] | ((uint32_t)(t[i].len - kLen.base[ls]) << llen[257 + ls]), llen[257 + ls] + kLen.extra[ls]);
                const uint32_t dist = (uint32_t)t[i].dist + 1u;
                const int ds = kLen.dist_sym(dist);
                bw.put((uint32_t)dcode[ds] | ((dist - kLen.dbase[ds]) << dlen[ds]), dlen[ds] + kLen.dextra[ds]);
            }
        }
        bw.put(lcode[256], llen[256]);
    }

    // raw deflate stream of src[0, n) into [out, out + cap); returns its size or -1
    int64_t run(const unsigned char* src, size_t n, unsigned char* out, size_t cap) {
        BitWriter bw{out, out + cap};
        std::fill(head.begin(), head.end(), 0u);  // position + 1; 0 = none
        uint32_t lfreq[kLitLen], dfreq[kDist];
        auto reset = [&]() {
            std::memset(lfreq, 0, sizeof lfreq);
            std::memset(dfreq, 0, sizeof dfreq);
        };
        reset();
        Token* t = tok.data();
        int n_tok = 0;
        size_t ip = 0, block_from = 0;
        const size_t safe = n >= 12 ? n - 12 : 0;  // 4-byte loads and 8-byte compares stay inside
        while (ip < n) {
            bool matched = false;
            if (ip < safe) {
                uint32_t v;
                std::memcpy(&v, src + ip, 4);
                const uint32_t h = (v * 2654435761u) >> (32 - kHashBits);
                const uint32_t cand1 = head[h];
                head[h] = (uint32_t)ip + 1u;
                if (cand1) {
                    const size_t cand = cand1 - 1u;
                    const size_t dist = ip - cand;
                    uint32_t u;
                    std::memcpy(&u, src + cand, 4);
                    if (dist <= 32768 && u == v) {
                        size_t len = 4;
                        const size_t most = std::min<size_t>(258, n - ip);
                        while (len + 8 <= most) {
                            uint64_t x, y;
                            std::memcpy(&x, src + ip + len, 8);
                            std::memcpy(&y, src + cand + len, 8);
                            const uint64_t d = x ^ y;
                            if (d) {
                                len += (size_t)(__builtin_ctzll(d) >> 3);
                                goto done;
                            }
                            len += 8;
                        }
                        while (len < most && src[ip + len] == src[cand + len]) ++len;
                    done:
                        t[n_tok++] = Token{(uint16_t)len, (uint16_t)(dist - 1)};
                        ++lfreq[257 + kLen.sym[len]];
                        ++dfreq[kLen.dist_sym((uint32_t)dist)];
                        // (one more position inside the match: the next line's tail finds it)
                        if (ip + len < safe) {
                            uint32_t w;
                            std::memcpy(&w, src + ip + len - 3, 4);
                            head[(w * 2654435761u) >> (32 - kHashBits)] = (uint32_t)(ip + len - 3) + 1u;
                        }
                        ip += len;
                        matched = true;
                    }
                }
            }
            if (!matched) {
                t[n_tok++] = Token{0, src[ip]};
                ++lfreq[src[ip]];
                ++ip;
            }
            if (n_tok == kMaxTokens) {
                block(bw, src, block_from, ip, n_tok, lfreq, dfreq, ip == n);
                if (bw.overflow) return -1;
                block_from = ip;
                n_tok = 0;
                reset();
            }
        }
        if (block_from < n || n == 0) block(bw, src, block_from, n, n_tok, lfreq, dfreq, true);
        bw.align();
        if (bw.overflow) return -1;
        return (int64_t)(bw.p - out);
    }
};

}  // namespace

extern "C" {

uint32_t wk_crc32(uint32_t crc, const char* data, int64_t n) {
    if (!data || n <= 0) return crc;
    const unsigned char* p = reinterpret_cast<const unsigned char*>(data);
    return have_clmul() ? crc32_clmul(crc, p, (size_t)n) : crc32_slice8(crc, p, (size_t)n);
}

int64_t wk_gz_bound(int64_t n) {
    if (n < 0) return -1;
    // A block is closed every kMaxTokens tokens (>= kMaxTokens bytes of input
    // each but the last) and costs at most its bytes + 5 per stored piece of
    // 65535 + the alignment of its first and last byte; head and trailer: 28.
    const int64_t blocks = n / kMaxTokens + 1;
    return n + 5 * (n / 65535 + blocks) + 2 * blocks + 64;
}

// One gzip member holding data[0, n): header with the 'WK' extra subfield (the
// member's total size, so that a reader can skip from member to member), the
// deflate stream, CRC-32 and size.  Returns the member's size, or -1 when `cap`
// is too small (wk_gz_bound(n) always suffices).
int64_t wk_gz_member(const char* data, int64_t n, char* out, int64_t cap) {
    static const unsigned char head[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x08, 0x00, 'W', 'K', 0x04, 0x00};
    if (n < 0 || n >= (1ll << 32) || !out || (n > 0 && !data) || cap < 28) return -1;
    thread_local Deflater d;
    std::memcpy(out, head, 16);
    const unsigned char* src = reinterpret_cast<const unsigned char*>(data);
    const int64_t body = d.run(src, (size_t)n, reinterpret_cast<unsigned char*>(out) + 20, (size_t)(cap - 28));
    if (body < 0) return -1;
    const uint64_t size = 20ull + (uint64_t)body + 8ull;
    if (size >= (1ull << 32)) return -1;
    const uint32_t size32 = (uint32_t)size, crc = wk_crc32(0, data, n), isize = (uint32_t)n;
    std::memcpy(out + 16, &size32, 4);
    std::memcpy(out + 20 + body, &crc, 4);
    std::memcpy(out + 24 + body, &isize, 4);
    return (int64_t)size;
}

// Inflate members written by wk_gz_member — blob[lo[i], hi[i]) each, found by
// their 'WK' size fields — on `n_threads` threads straight into out[off[i],
// off[i + 1]) (the caller sums the members' ISIZE fields): the stratified
// second pass reads a sample's read map this way (workflow.read_strata,
// workflow.py:912-938).  Returns 0, or -(1 + i) for the first member that is
// not what its trailer says (size, CRC-32) or does not inflate.
int64_t wk_gz_inflate_members(const char* blob, const int64_t* lo, const int64_t* hi, int64_t n, char* out, const int64_t* off,
                              int n_threads) {
    if (n < 0 || (n > 0 && (!blob || !lo || !hi || !out || !off))) return -1;
    std::atomic<int64_t> next{0}, bad{n};
    auto work = [&]() {
        for (int64_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
            const int64_t a = lo[i] + 20, b = hi[i] - 8, want = off[i + 1] - off[i];
            bool ok = b >= a && want >= 0;
            if (ok) {
                z_stream z;
                std::memset(&z, 0, sizeof z);
                ok = inflateInit2(&z, -15) == Z_OK;
                if (ok) {
                    // (members are smaller than 4 GB: wk_gz_member refuses larger ones)
                    z.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(blob + a));
                    z.avail_in = (uInt)(b - a);
                    z.next_out = reinterpret_cast<Bytef*>(out + off[i]);
                    z.avail_out = (uInt)want;
                    const int rc = inflate(&z, Z_FINISH);
                    ok = rc == Z_STREAM_END && (int64_t)z.total_out == want;
                    inflateEnd(&z);
                }
            }
            if (ok) {
                uint32_t crc, isize;
                std::memcpy(&crc, blob + b, 4);
                std::memcpy(&isize, blob + b + 4, 4);
                ok = isize == (uint32_t)want && crc == wk_crc32(0, out + off[i], want);
            }
            if (!ok) {
                int64_t cur = bad.load();
                while (i < cur && !bad.compare_exchange_weak(cur, i)) {
                }
            }
        }
    };
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads > 0 ? n_threads : 1, n));
    if (T == 1) {
        work();
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(work);
        for (auto& x : th) x.join();
    }
    return bad.load() < n ? -(1 + bad.load()) : 0;
}

}  // extern "C"
