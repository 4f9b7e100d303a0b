// wk_coords.cpp — native reader of gene coordinate files (host side).
//
// ordinal.load_gene_coords + encode_genes of the reference (woltka/ordinal.py:
// 338-473) without a Python object per line: ">name" / "# name" lines start a
// nucleotide (a doubled marker is a super-group label and is ignored), other
// lines are "gene <tab> beg <tab> end" (1-based, inclusive, either strand
// order); start0 = min(beg, end) - 1, end = max(beg, end) (ordinal.py:459-465);
// the genes of a nucleotide are sorted by start0, stably; a name seen again
// replaces its earlier genes but keeps its place; `isdup` = some gene id was
// seen twice (ordinal.py:413-417).  Text whose reading depends on Python's str
// / int rules beyond plain ASCII digits — non-ASCII white space, "1_000", a
// bare '\r', a marker as the very last byte — is refused (WK_E_STATE): the
// Python reader (woltka_amd/ordinal.py) takes the file and raises or accepts
// like the reference.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/woltka_hip.h"
#include "wk_names.hpp"

using wkh::hash_bytes;
using wkh::NameTable;

struct wk_coords {
    std::vector<int32_t> goff, start0, end, findex;  // findex: the gene's place in its nucleotide's lines (the index encode_genes puts into its codes, ordinal.py:459-465)
    std::string genome_blob, gene_blob;
    std::vector<int64_t> genome_off, gene_off;
    int isdup = 0;
    std::string err;
};

namespace {

inline bool py_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }

// int(text) for optional white space, optional sign, ASCII digits; false = not
// that simple (the Python reader decides)
inline bool simple_int(const char* p, const char* e, long long& v) {
    while (p < e && py_space((unsigned char)*p)) ++p;
    while (e > p && py_space((unsigned char)e[-1])) --e;
    bool neg = false;
    if (p < e && (*p == '-' || *p == '+')) neg = *p++ == '-';
    if (p >= e || e - p > 18) return false;
    long long x = 0;
    for (; p < e; ++p) {
        if (*p < '0' || *p > '9') return false;
        x = x * 10 + (*p - '0');
    }
    v = neg ? -x : x;
    return true;
}

struct Gene {
    const char* name;
    uint32_t nlen;
    long long lo, hi;
};

struct Nucl {
    const char* name;
    uint32_t nlen;
    std::vector<Gene> genes;
};

// Does some gene id occur twice among `ids` (ordinal.py:413-417 keeps a set of them while it reads)?  Half a
// million ids through one hash table as they are read cost more than everything else the reader does (a random
// access into megabytes per line).  Here the ids are dealt into a thousand piles by the top bits of their hashes
// -- two passes over arrays, no random access -- and every pile gets a table of its own that stays in the cache.
struct IdRef {
    const char* p;
    uint32_t n;
    uint64_t h;  // hash_bytes(p, n), taken while the line is being read
};
bool any_id_twice(const std::vector<IdRef>& ids) {
    const size_t n = ids.size();
    if (n < 2) return false;
    constexpr int kBits = 10;
    constexpr size_t kPiles = (size_t)1 << kBits;
    std::vector<uint32_t> first(kPiles + 1, 0);
    for (size_t i = 0; i < n; ++i) first[(size_t)(ids[i].h >> (64 - kBits)) + 1] += 1;
    for (size_t k = 0; k < kPiles; ++k) first[k + 1] += first[k];
    std::vector<uint32_t> pile(n), at(first.begin(), first.end() - 1);
    for (size_t i = 0; i < n; ++i) pile[at[(size_t)(ids[i].h >> (64 - kBits))]++] = (uint32_t)i;
    std::vector<uint32_t> slots;  // id number + 1, 0 = free
    for (size_t k = 0; k < kPiles; ++k) {
        const uint32_t lo = first[k], hi = first[k + 1];
        if (hi - lo < 2) continue;
        size_t cap = 16;
        while (cap < (size_t)(hi - lo) * 2) cap <<= 1;
        slots.assign(cap, 0u);
        for (uint32_t x = lo; x < hi; ++x) {
            const uint32_t i = pile[x];
            size_t s = (size_t)(ids[i].h >> 8) & (cap - 1);
            while (slots[s]) {
                const uint32_t j = slots[s] - 1u;
                if (ids[j].h == ids[i].h && ids[j].n == ids[i].n && memcmp(ids[j].p, ids[i].p, ids[i].n) == 0) return true;
                s = (s + 1) & (cap - 1);
            }
            slots[s] = i + 1u;
        }
    }
    return false;
}

}  // namespace

extern "C" {

int wk_coords_parse(const char* buf, int64_t len, wk_coords** out) {
    if (!out || len < 0 || (len > 0 && !buf)) return WK_E_ARG;
    wk_coords* c = new (std::nothrow) wk_coords();
    if (!c) return WK_E_HIP;
    *out = c;
    std::vector<Nucl> nucls;
    NameTable index;  // nucleotide name -> position in `nucls`
    std::vector<IdRef> ids;  // the gene ids of all lines, in text order (any_id_twice)
    ids.reserve((size_t)len / 16 + 16);
    int cur = -1;
    const char* p = buf;
    const char* e = buf + len;
    while (p < e) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
        const char* le = nl ? nl : e;
        const char* line = p;
        p = nl ? nl + 1 : e;
        const char* cr = (const char*)memchr(line, '\r', (size_t)(le - line));
        if (cr && !(cr + 1 == le && nl)) return WK_E_STATE;
        if (le > line && (unsigned char)le[-1] >= 0x80) return WK_E_STATE;
        // `line` of the Python loop = [line, le) + "\n" (if nl); line[0] is '\n' for an empty one
        const char c0 = line < le ? *line : '\n';
        if (c0 == '>' || c0 == '#') {
            // line[1]: the next byte, '\n' at the line's end, nothing at the end of the file
            if (line + 1 >= le && !nl) return WK_E_STATE;  // (IndexError in the reference)
            const char c1 = line + 1 < le ? line[1] : '\n';
            if (c1 == c0) continue;
            const char* nb = line + 1;
            const char* ne = le;
            if (nb < ne && (unsigned char)*nb >= 0x80) return WK_E_STATE;
            while (nb < ne && py_space((unsigned char)*nb)) ++nb;
            while (ne > nb && py_space((unsigned char)ne[-1])) --ne;
            const uint64_t hv = hash_bytes(nb, (size_t)(ne - nb));
            int32_t id = index.find(nb, (size_t)(ne - nb), hv);
            if (id < 0) {
                id = index.add(nb, (size_t)(ne - nb), hv);
                nucls.push_back(Nucl{nb, (uint32_t)(ne - nb), {}});
            } else {
                nucls[(size_t)id].genes.clear();  // `coords[nucl] = []`
            }
            cur = id;
            continue;
        }
        // gene, beg, end = line.rstrip().split('\t')
        const char* re = le;
        while (re > line && py_space((unsigned char)re[-1])) --re;
        const char* t1 = (const char*)memchr(line, '\t', (size_t)(re - line));
        const char* t2 = t1 ? (const char*)memchr(t1 + 1, '\t', (size_t)(re - t1 - 1)) : nullptr;
        if (!t1 || !t2 || memchr(t2 + 1, '\t', (size_t)(re - t2 - 1))) {
            c->err = "Cannot extract coordinates from line: \"" + std::string(line, (size_t)(le - line)) + (nl ? "\n" : "") + "\".";
            return WK_E_ARG;
        }
        if (cur < 0) return WK_E_STATE;  // coordinates before any nucleotide: the Python reader's business
        long long b = 0, en = 0;
        if (!simple_int(t1 + 1, t2, b) || !simple_int(t2 + 1, re, en)) return WK_E_STATE;
        nucls[(size_t)cur].genes.push_back(Gene{line, (uint32_t)(t1 - line), std::min(b, en) - 1, std::max(b, en)});
        ids.push_back(IdRef{line, (uint32_t)(t1 - line), hash_bytes(line, (size_t)(t1 - line))});
    }
    if (nucls.empty()) {
        c->err = "No coordinate was read from file.";
        return WK_E_ARG;
    }
    c->isdup = any_id_twice(ids) ? 1 : 0;
    {
        // (room for everything below in one go)
        const size_t n_genes = ids.size();
        c->start0.reserve(n_genes);
        c->end.reserve(n_genes);
        c->findex.reserve(n_genes);
        c->gene_off.reserve(n_genes + 1);
        c->goff.reserve(nucls.size() + 1);
        c->genome_off.reserve(nucls.size() + 1);
        size_t id_bytes = 0;
        for (const IdRef& r : ids) id_bytes += r.n;
        c->gene_blob.reserve(id_bytes);
    }
    c->goff.push_back(0);
    c->genome_off.push_back(0);
    c->gene_off.push_back(0);
    std::vector<uint32_t> order;
    for (const Nucl& n : nucls) {
        order.resize(n.genes.size());
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return n.genes[x].lo < n.genes[y].lo; });
        for (uint32_t i : order) {
            const Gene& g = n.genes[i];
            if (g.hi > 2147483647ll || g.lo < -1) {
                c->err = "Gene coordinates beyond 2^31 - 1 are not supported by the device tables.";
                return WK_E_ARG;
            }
            c->start0.push_back((int32_t)g.lo);
            c->findex.push_back((int32_t)i);
            c->end.push_back((int32_t)g.hi);
            c->gene_blob.append(g.name, g.nlen);
            c->gene_off.push_back((int64_t)c->gene_blob.size());
        }
        c->goff.push_back((int32_t)c->start0.size());
        c->genome_blob.append(n.name, n.nlen);
        c->genome_off.push_back((int64_t)c->genome_blob.size());
    }
    return WK_OK;
}

const char* wk_coords_error(const wk_coords* c) { return c ? c->err.c_str() : ""; }

int wk_coords_sizes(const wk_coords* c, int32_t* n_genomes, int32_t* n_genes, int64_t* genome_bytes, int64_t* gene_bytes,
                    int* isdup) {
    if (!c) return WK_E_ARG;
    if (n_genomes) *n_genomes = (int32_t)c->goff.size() - 1;
    if (n_genes) *n_genes = (int32_t)c->start0.size();
    if (genome_bytes) *genome_bytes = (int64_t)c->genome_blob.size();
    if (gene_bytes) *gene_bytes = (int64_t)c->gene_blob.size();
    if (isdup) *isdup = c->isdup;
    return WK_OK;
}

int wk_coords_fetch(const wk_coords* c, int32_t* goff, int32_t* start0, int32_t* end, int32_t* findex, char* genome_blob,
                    int64_t* genome_off, char* gene_blob, int64_t* gene_off) {
    if (!c) return WK_E_ARG;
    if (goff) memcpy(goff, c->goff.data(), c->goff.size() * 4);
    if (start0 && !c->start0.empty()) memcpy(start0, c->start0.data(), c->start0.size() * 4);
    if (end && !c->end.empty()) memcpy(end, c->end.data(), c->end.size() * 4);
    if (findex && !c->findex.empty()) memcpy(findex, c->findex.data(), c->findex.size() * 4);
    if (genome_blob && !c->genome_blob.empty()) memcpy(genome_blob, c->genome_blob.data(), c->genome_blob.size());
    if (genome_off) memcpy(genome_off, c->genome_off.data(), c->genome_off.size() * 8);
    if (gene_blob && !c->gene_blob.empty()) memcpy(gene_blob, c->gene_blob.data(), c->gene_blob.size());
    if (gene_off) memcpy(gene_off, c->gene_off.data(), c->gene_off.size() * 8);
    return WK_OK;
}

void wk_coords_free(wk_coords* c) { delete c; }

int wk_blob_join(const char* blob, const int64_t* off, int64_t n, char sep, char* out) {
    if (n < 0 || !off || !out || (n > 0 && !blob && off[n] > off[0])) return WK_E_ARG;
    char* w = out;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t len = off[i + 1] - off[i];
        if (len < 0) return WK_E_ARG;
        memcpy(w, blob + off[i], (size_t)len);
        w += len;
        *w++ = sep;
    }
    return WK_OK;
}


// The body of a one-sample TSV table (table.write_tsv after table.prep_table of
// the reference, woltka/table.py:29-66, 247-283): the n keys of `keys` (joined by
// '\n') in ascending order — byte order of UTF-8 = Python's order of str —, each
// as "key \t value \n"; rows whose value is 0 are left out.
int wk_table_body(const char* keys, int64_t keys_len, const int64_t* values, int64_t n, int threads, char* out, int64_t cap,
                  int64_t* out_len, int64_t* n_rows) {
    if (n < 0 || keys_len < 0 || !out_len || !n_rows || (n > 0 && (!keys || !values)) || (cap > 0 && !out)) return WK_E_ARG;
    *out_len = *n_rows = 0;
    if (n == 0) return WK_OK;
    std::vector<uint32_t> off((size_t)n + 1);
    {
        int64_t k = 0;
        off[0] = 0;
        for (int64_t i = 0; i < keys_len; ++i)
            if (keys[i] == '\n') {
                if (++k >= n) return WK_E_ARG;  // more separators than keys
                off[(size_t)k] = (uint32_t)(i + 1);
            }
        if (k != n - 1 || keys_len >= (1ll << 32)) return WK_E_ARG;
        off[(size_t)n] = (uint32_t)(keys_len + 1);
    }
    auto less = [&](uint32_t a, uint32_t b) {
        const uint32_t la = off[a + 1] - off[a] - 1, lb = off[b + 1] - off[b] - 1;
        const int c = memcmp(keys + off[a], keys + off[b], std::min(la, lb));
        return c < 0 || (c == 0 && la < lb);
    };
    std::vector<uint32_t> order((size_t)n);
    std::iota(order.begin(), order.end(), 0u);
    int T = (int)std::max<int64_t>(1, std::min<int64_t>(threads > 0 ? threads : 8, n >> 14));
    if (T == 1) {
        std::sort(order.begin(), order.end(), less);
    } else {
        std::vector<size_t> cut((size_t)T + 1);
        for (int t = 0; t <= T; ++t) cut[(size_t)t] = (size_t)n * (size_t)t / (size_t)T;
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t] { std::sort(order.begin() + (ptrdiff_t)cut[(size_t)t], order.begin() + (ptrdiff_t)cut[(size_t)t + 1], less); });
        for (auto& x : th) x.join();
        for (int step = 1; step < T; step *= 2) {  // pairwise merges, a thread each
            std::vector<std::thread> mt;
            for (int t = 0; t + step < T; t += 2 * step)
                mt.emplace_back([&, t, step] {
                    const size_t hi = cut[(size_t)std::min(T, t + 2 * step)];
                    std::inplace_merge(order.begin() + (ptrdiff_t)cut[(size_t)t], order.begin() + (ptrdiff_t)cut[(size_t)(t + step)],
                                       order.begin() + (ptrdiff_t)hi, less);
                });
            for (auto& x : mt) x.join();
        }
    }
    char* w = out;
    char* const end = out + cap;
    int64_t rows = 0;
    for (uint32_t i : order) {
        const int64_t v = values[i];
        if (v == 0) continue;
        const uint32_t len = off[i + 1] - off[i] - 1;
        if (w + len + 24 > end) return WK_E_CAPACITY;
        memcpy(w, keys + off[i], len);
        w += len;
        *w++ = '\t';
        char digits[24];
        int d = 0;
        uint64_t u = v < 0 ? (uint64_t)(-(v + 1)) + 1u : (uint64_t)v;
        do {
            digits[d++] = (char)('0' + u % 10);
            u /= 10;
        } while (u);
        if (v < 0) *w++ = '-';
        while (d) *w++ = digits[--d];
        *w++ = '\n';
        rows += 1;
    }
    *out_len = (int64_t)(w - out);
    *n_rows = rows;
    return WK_OK;
}


}  // extern "C"

namespace {

// offsets of n strings joined by '\n' (none of them holds one); false if the blob is anything else
bool split_lines(const char* blob, int64_t len, int64_t n, std::vector<uint32_t>& off) {
    off.assign((size_t)n + 1, 0);
    if (n == 0) return len == 0;
    if (len >= (1ll << 32)) return false;
    int64_t k = 0;
    for (int64_t i = 0; i < len; ++i)
        if (blob[i] == '\n') {
            if (++k >= n) return false;
            off[(size_t)k] = (uint32_t)(i + 1);
        }
    if (k != n - 1) return false;
    off[(size_t)n] = (uint32_t)(len + 1);
    return true;
}

// rank[i] = place of string i among the n strings in byte order (= Python's order of str for UTF-8).
// Half a million gene ids take a single thread ~60 ms to sort: parts are sorted on threads of their own and merged
// pairwise, the strings' first eight bytes (big-endian) next to their numbers so that most comparisons touch no text.
void rank_strings(const char* blob, const std::vector<uint32_t>& off, int64_t n, std::vector<uint32_t>& rank) {
    struct Key {
        uint64_t head;
        uint32_t id;
    };
    std::vector<Key> order((size_t)n);
    auto head_of = [&](uint32_t i) {
        const uint32_t len = off[i + 1] - off[i] - 1;
        unsigned char b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        memcpy(b, blob + off[i], std::min<uint32_t>(len, 8));
        uint64_t h = 0;
        for (int k = 0; k < 8; ++k) h = h << 8 | b[k];
        return h;
    };
    auto less = [&](const Key& x, const Key& y) {
        if (x.head != y.head) return x.head < y.head;
        // (equal heads: both shorter than 8 bytes and the same, a NUL byte in a name, or the text decides)
        const uint32_t la = off[x.id + 1] - off[x.id] - 1, lb = off[y.id + 1] - off[y.id] - 1;
        const int c = memcmp(blob + off[x.id], blob + off[y.id], std::min(la, lb));
        return c < 0 || (c == 0 && la < lb);
    };
    int parts = 1;
    if (n >= (1 << 16)) {
        const unsigned hw = std::thread::hardware_concurrency();
        parts = (int)std::min<int64_t>(std::max(1u, std::min(hw, 8u)), n >> 14);
        int p2 = 1;
        while (p2 * 2 <= parts) p2 *= 2;
        parts = p2;
    }
    std::vector<int64_t> cut((size_t)parts + 1);
    for (int t = 0; t <= parts; ++t) cut[(size_t)t] = n * t / parts;
    auto sort_part = [&](int t) {
        for (int64_t i = cut[(size_t)t]; i < cut[(size_t)t + 1]; ++i) order[(size_t)i] = Key{head_of((uint32_t)i), (uint32_t)i};
        std::sort(order.begin() + cut[(size_t)t], order.begin() + cut[(size_t)t + 1], less);
    };
    if (parts == 1) {
        sort_part(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < parts; ++t) th.emplace_back(sort_part, t);
        for (std::thread& x : th) x.join();
        for (int width = 1; width < parts; width *= 2) {
            th.clear();
            for (int t = 0; t + width < parts; t += 2 * width)
                th.emplace_back([&, t, width] {
                    std::inplace_merge(order.begin() + cut[(size_t)t], order.begin() + cut[(size_t)(t + width)],
                                       order.begin() + cut[(size_t)std::min(parts, t + 2 * width)], less);
                });
            for (std::thread& x : th) x.join();
        }
    }
    rank.assign((size_t)n, 0);
    for (int64_t i = 0; i < n; ++i) rank[order[(size_t)i].id] = (uint32_t)i;
}

inline char* put_i64(char* w, int64_t v) {
    char digits[24];
    int d = 0;
    uint64_t u = v < 0 ? (uint64_t)(-(v + 1)) + 1u : (uint64_t)v;
    do {
        digits[d++] = (char)('0' + u % 10);
        u /= 10;
    } while (u);
    if (v < 0) *w++ = '-';
    while (d) *w++ = digits[--d];
    return w;
}

}  // namespace

extern "C" {

// The body of a TSV table with n_cols sample columns and (optionally) stratified
// row names (table.prep_table + table.write_tsv, woltka/table.py:29-136, 247-283):
// row r is named `prefix|name` — prefix_of_row[r] < 0 or an empty prefix: just
// `name` — and holds values[r * n_cols .. + n_cols).  Rows come out sorted like
// the reference's `sorted(allkeys(profile))` over (stratum, feature) tuples /
// plain ids: by prefix, then by name, in byte order (rows without prefix as if
// their prefix were the name's — the caller never mixes the two kinds); rows
// that are all zero are left out.  `prefixes` / `names`: the strings joined by
// '\n'.  `out` needs, per row, its two strings + 2 + 21 n_cols bytes.
int wk_table_rows(const char* prefixes, int64_t prefixes_len, int32_t n_prefixes, const char* names, int64_t names_len,
                  int32_t n_names, const int32_t* prefix_of_row, const int32_t* name_of_row, const int64_t* values, int64_t n_rows,
                  int32_t n_cols, char* out, int64_t cap, int64_t* out_len, int64_t* rows_written) {
    if (n_rows < 0 || n_cols < 0 || n_prefixes < 0 || n_names < 0 || !out_len || !rows_written ||
        (n_rows > 0 && (!name_of_row || (n_cols > 0 && !values))) || (cap > 0 && !out))
        return WK_E_ARG;
    *out_len = *rows_written = 0;
    std::vector<uint32_t> poff, noff, prank, nrank;
    if (!split_lines(prefixes, prefixes_len, n_prefixes, poff) || !split_lines(names, names_len, n_names, noff)) return WK_E_ARG;
    rank_strings(prefixes, poff, n_prefixes, prank);
    rank_strings(names, noff, n_names, nrank);
    std::vector<std::pair<uint64_t, uint32_t>> order((size_t)n_rows);
    for (int64_t r = 0; r < n_rows; ++r) {
        const int32_t p = prefix_of_row ? prefix_of_row[r] : -1, m = name_of_row[r];
        if (p >= n_prefixes || m < 0 || m >= n_names) return WK_E_ARG;
        const uint64_t hi = p < 0 ? 0ull : 1ull + prank[(size_t)p];
        order[(size_t)r] = {hi << 32 | nrank[(size_t)m], (uint32_t)r};
    }
    std::sort(order.begin(), order.end());
    char* w = out;
    char* const end = out + cap;
    int64_t rows = 0;
    for (const auto& kv : order) {
        const uint32_t r = kv.second;
        const int64_t* v = values + (size_t)r * (size_t)n_cols;
        bool any = false;
        for (int32_t c = 0; c < n_cols; ++c) any |= v[c] != 0;
        if (!any) continue;
        const int32_t p = prefix_of_row ? prefix_of_row[r] : -1, m = name_of_row[r];
        const uint32_t pl = p < 0 ? 0u : poff[(size_t)p + 1] - poff[(size_t)p] - 1, nl = noff[(size_t)m + 1] - noff[(size_t)m] - 1;
        if (w + pl + nl + 2 + (size_t)21 * (size_t)n_cols + 1 > end) return WK_E_CAPACITY;
        if (pl) {
            memcpy(w, prefixes + poff[(size_t)p], pl);
            w += pl;
            *w++ = '|';
        }
        memcpy(w, names + noff[(size_t)m], nl);
        w += nl;
        *w++ = '\t';  // (write_tsv joins the sample block as one field: the tab is there without samples too)
        for (int32_t c = 0; c < n_cols; ++c) {
            if (c) *w++ = '\t';
            w = put_i64(w, v[c]);
        }
        *w++ = '\n';
        rows += 1;
    }
    *out_len = (int64_t)(w - out);
    *rows_written = rows;
    return WK_OK;
}

}  // extern "C"
