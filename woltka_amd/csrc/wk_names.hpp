// wk_names.hpp — byte-string hashing and string -> dense id tables of the host
// side (alignment tokenizer, hierarchy ingest).  Host only.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace wkh {

inline uint64_t hash_bytes(const char* p, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull ^ (n * 0x9E3779B97F4A7C15ull);
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        h = (h ^ v) * 0x100000001b3ull;
        h ^= h >> 29;
        p += 8;
        n -= 8;
    }
    uint64_t v = 0;
    memcpy(&v, p, n);
    h = (h ^ v) * 0x100000001b3ull;
    // (an avalanche at the end: names shorter than eight bytes go through the one multiply above only, and the low
    // bits of a product depend on the low bits of its factors alone -- 5 000 names "G000000" .. "G004999" had 500
    // distinct home slots in a table of 16 384, 22 probes per look-up on average, 180 at most: the 4x slower
    // dtok_parse on config 5's text that round 5 left "not understood".  With this: 1.2 probes, 10 at most.)
    h ^= h >> 32;
    h *= 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}

// End of [b, e) after Python's str.rstrip() on the UTF-8 text: every character
// str.isspace() knows -- \t \n \v \f \r, \x1c-\x1f, space, U+0085, U+00A0,
// U+1680, U+2000-200A, U+2028, U+2029, U+202F, U+205F, U+3000.
inline const char* py_rstrip(const char* b, const char* e) {
    for (;;) {
        if (e <= b) return e;
        const unsigned char c = (unsigned char)e[-1];
        if (c == ' ' || (c >= '\t' && c <= '\r') || (c >= 0x1c && c <= 0x1f)) {
            --e;
            continue;
        }
        if (c < 0x80) return e;
        if (e - b >= 2 && (unsigned char)e[-2] == 0xC2 && (c == 0x85 || c == 0xA0)) {
            e -= 2;
            continue;
        }
        if (e - b >= 3) {
            const unsigned char a0 = (unsigned char)e[-3], a1 = (unsigned char)e[-2];
            const bool sp = (a0 == 0xE1 && a1 == 0x9A && c == 0x80) ||
                            (a0 == 0xE2 && a1 == 0x80 && ((c >= 0x80 && c <= 0x8A) || c == 0xA8 || c == 0xA9 || c == 0xAF)) ||
                            (a0 == 0xE2 && a1 == 0x81 && c == 0x9F) || (a0 == 0xE3 && a1 == 0x80 && c == 0x80);
            if (sp) {
                e -= 3;
                continue;
            }
        }
        return e;
    }
}

// string -> id table; names live in an arena (stable across calls), ids are
// dense in order of insertion
struct NameTable {
    std::vector<int32_t> slot;  // id or -1
    std::vector<uint64_t> hash;
    std::vector<uint32_t> off, len;  // per id
    std::string arena;
    size_t mask = 0;
    NameTable() { rehash(1 << 12); }
    void rehash(size_t n) {
        slot.assign(n, -1);
        mask = n - 1;
        for (int32_t id = 0; id < (int32_t)off.size(); ++id) {
            size_t h = hash[id] & mask;
            while (slot[h] >= 0) h = (h + 1) & mask;
            slot[h] = id;
        }
    }
    void reserve(size_t n_names, size_t n_bytes) {
        size_t want = slot.size();
        while (want < 2 * n_names + 2) want <<= 1;
        if (want != slot.size()) rehash(want);
        hash.reserve(n_names);
        off.reserve(n_names);
        len.reserve(n_names);
        arena.reserve(n_bytes);
    }
    int32_t find(const char* p, size_t n, uint64_t hv) const {
        size_t h = hv & mask;
        for (;;) {
            const int32_t id = slot[h];
            if (id < 0) return -1;
            if (hash[id] == hv && len[id] == n && memcmp(arena.data() + off[id], p, n) == 0) return id;
            h = (h + 1) & mask;
        }
    }
    int32_t add(const char* p, size_t n, uint64_t hv) {
        if ((off.size() + 1) * 2 > slot.size()) rehash(slot.size() * 2);
        const int32_t id = (int32_t)off.size();
        off.push_back((uint32_t)arena.size());
        len.push_back((uint32_t)n);
        hash.push_back(hv);
        arena.append(p, n);
        size_t h = hv & mask;
        while (slot[h] >= 0) h = (h + 1) & mask;
        slot[h] = id;
        return id;
    }
    int32_t intern(const char* p, size_t n, uint64_t hv) {
        const int32_t id = find(p, n, hv);
        return id >= 0 ? id : add(p, n, hv);
    }
    int32_t size() const { return (int32_t)off.size(); }
    const char* ptr(int32_t id) const { return arena.data() + off[id]; }
};

}  // namespace wkh
