// wk_hierarchy.cpp — native ingest of classification hierarchies (host side).
//
// Replaces, for the files that carry millions of lines, the per-line Python of
// the reference's hierarchy readers and what follows them:
//   tree.read_nodes  (woltka/tree.py:73-101)   nodes.dmp / "id <tab> parent [<tab> rank]"
//   tree.read_names  (woltka/tree.py:48-70)    names.dmp / "id <tab> name"
//   file.read_map_1st (woltka/file.py:388-406) "subject <tab> taxon" maps
//   util.update_dict (woltka/util.py:46-75)    merging the files' dicts, conflicts are errors
//   tree.fill_root   (woltka/tree.py:302-388)  one root; missing parents join the tree
// and the flattening to the pre-order arrays wk_set_tree takes.  The reference
// builds three Python dicts (child -> parent, node -> rank, node -> name) with
// one entry per line; here the text is parsed by all threads into one sharded
// symbol table, the dict semantics are kept (a repeated key inside a file takes
// its last value, a key that two files disagree on is an error), and Python
// sees the dicts as lazy views that look single entries up (hierarchy.py).
//
// Text the native readers do not take — a byte >= 0x80 at a line's end (rstrip
// of Unicode white space), a bare '\r' (universal newlines), a "\t|" in the
// middle of a field — is refused with WK_E_ARG before anything is changed; the
// caller then reads that file with the Python reader and hands the pairs over
// (wk_hier_update).
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/woltka_hip.h"
#include "wk_names.hpp"

using wkh::hash_bytes;
using wkh::NameTable;

namespace {

constexpr int kShards = 64;
constexpr uint32_t kNone = 0xFFFFFFFFu;   // tree[x] = None
constexpr uint32_t kUnset = 0xFFFFFFFEu;  // x is not a key of the tree

inline int shard_of(uint64_t h) { return (int)(h >> 58); }
inline uint32_t make_ref(int shard, int32_t local) { return ((uint32_t)local << 6) | (uint32_t)shard; }
inline int ref_shard(uint32_t r) { return (int)(r & 63u); }
inline int32_t ref_local(uint32_t r) { return (int32_t)(r >> 6); }

// str.rstrip() / str.isspace() on ASCII
inline bool py_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }

enum : uint8_t { HAS_PARENT = 1, HAS_NONE = 2, HAS_RANK = 4, HAS_TITLE = 8, KEY_RESOLVED = 16 /* kref is set: no name to intern */ };

struct Ent {
    const char* k;
    const char* v;  // parent name, or title text
    uint32_t kn, vn;
    int32_t rank;  // local (per parse thread), then global rank code - 1
    uint8_t has;
    uint64_t kh, vh;
    uint32_t kref, vref;
};

struct Shard {
    NameTable names;
    std::vector<uint32_t> parent;  // ref | kNone | kUnset
    std::vector<int32_t> rank;     // code >= 1, 0 = no rank entry
    std::vector<int64_t> t_off;    // title (namedic value) in `titles`, -1 = none
    std::vector<uint32_t> t_len;
    std::string titles;
    std::vector<int32_t> p_file, r_file, t_file;  // update (file) that set the field last, -1 = never
    std::vector<int64_t> dense;                   // node number in input order (finish), -1 = not in the tree
    // values a field had before the current update touched it (util.update_dict compares)
    // (+ where in the update the key came first: update_dict reports the first
    // conflicting key in the order of the other dict)
    struct SavedP { int32_t k; uint32_t old; uint64_t at; };
    struct SavedR { int32_t k; int32_t old; uint64_t at; };
    struct SavedT { int32_t k; std::string old; uint64_t at; };
    std::vector<SavedP> saved_p;
    std::vector<SavedR> saved_r;
    std::vector<SavedT> saved_t;
    void grow() {
        const size_t n = (size_t)names.size();
        if (parent.size() < n) {
            parent.resize(n, kUnset);
            rank.resize(n, 0);
            t_off.resize(n, -1);
            t_len.resize(n, 0);
            p_file.resize(n, -1);
            r_file.resize(n, -1);
            t_file.resize(n, -1);
        }
    }
};

// WK_HIER_TIMING=1 in the environment: phase times on stderr (measurement)
struct Lap {
    const bool on = getenv("WK_HIER_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void operator()(const char* what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "  [wk_hier] %-18s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

template <typename F>
void run_parallel(int threads, int n, F&& fn) {
    if (threads <= 1 || n <= 1) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    std::vector<std::thread> th;
    std::atomic<int> next{0};
    const int w = std::min(threads, n);
    th.reserve(w);
    for (int t = 0; t < w; ++t)
        th.emplace_back([&] {
            for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i);
        });
    for (auto& x : th) x.join();
}

}  // namespace

struct wk_hier {
    int n_threads = 1;
    std::string err;
    Shard shard[kShards];
    NameTable ranks;  // rank vocabulary: code = id + 1
    int32_t n_updates = 0;
    bool finished = false;
    // after wk_hier_finish
    int64_t n_nodes = 0;
    uint32_t root_ref = kNone;
    bool has_root = false;
    std::vector<int32_t> parent, last, rank_code, depth;  // pre-order
    std::vector<uint32_t> ref_of_pre;                      // pre-order id -> symbol
    std::vector<int64_t> pre_of_dense;
    std::vector<int64_t> rank_used;  // symbols carrying each rank code (code - 1)

    int32_t find_ref(const char* p, size_t n, uint32_t* ref) const {
        const uint64_t h = hash_bytes(p, n);
        const int s = shard_of(h);
        const int32_t id = shard[s].names.find(p, n, h);
        if (id < 0) return -1;
        *ref = make_ref(s, id);
        return 0;
    }
    const char* sym_name(uint32_t ref, uint32_t* len) const {
        const Shard& sh = shard[ref_shard(ref)];
        const int32_t id = ref_local(ref);
        *len = sh.names.len[id];
        return sh.names.ptr(id);
    }
};

namespace {

int hfail(wk_hier* h, int code, const std::string& msg) {
    h->err = msg;
    return code;
}

// Apply one update_dict(dic, other): `ents[t]` are the pairs of `other` in
// text order (thread ranges in order); a key repeated inside the update takes
// its last value (the dict the reader built), a key that an earlier update set
// to something else is a conflict.
int apply_update(wk_hier* h, std::vector<std::vector<Ent>>& ents) {
    const int T = (int)ents.size();
    const int upd = h->n_updates++;
    Lap lap;
    // buckets of entry indices by key shard / by parent-name shard
    std::vector<std::vector<uint32_t>> kb((size_t)T * kShards), vb((size_t)T * kShards);
    run_parallel(h->n_threads, T, [&](int t) {
        std::vector<Ent>& E = ents[t];
        std::vector<uint32_t> kc(kShards, 0), vc(kShards, 0);
        for (const Ent& e : E) {
            kc[shard_of(e.kh)] += 1;
            if (e.has & HAS_PARENT) vc[shard_of(e.vh)] += 1;
        }
        for (int s = 0; s < kShards; ++s) {
            kb[(size_t)t * kShards + s].reserve(kc[s]);
            vb[(size_t)t * kShards + s].reserve(vc[s]);
        }
        for (uint32_t i = 0; i < (uint32_t)E.size(); ++i) {
            kb[(size_t)t * kShards + shard_of(E[i].kh)].push_back(i);
            if (E[i].has & HAS_PARENT) vb[(size_t)t * kShards + shard_of(E[i].vh)].push_back(i);
        }
    });
    lap("bucket");
    // phase 1: intern keys and parent names (a shard is owned by one thread)
    run_parallel(h->n_threads, kShards, [&](int s) {
        Shard& sh = h->shard[s];
        size_t want = 0, bytes = 0;
        for (int t = 0; t < T; ++t)
            for (uint32_t i : kb[(size_t)t * kShards + s])
                if (!(ents[t][i].has & KEY_RESOLVED)) {
                    want += 1;
                    bytes += ents[t][i].kn;
                }
        sh.names.reserve((size_t)sh.names.size() + want, sh.names.arena.size() + bytes);
        for (int t = 0; t < T; ++t) {
            std::vector<Ent>& E = ents[t];
            for (uint32_t i : kb[(size_t)t * kShards + s])
                if (!(E[i].has & KEY_RESOLVED)) E[i].kref = make_ref(s, sh.names.intern(E[i].k, E[i].kn, E[i].kh));
        }
        for (int t = 0; t < T; ++t) {
            std::vector<Ent>& E = ents[t];
            for (uint32_t i : vb[(size_t)t * kShards + s]) E[i].vref = make_ref(s, sh.names.intern(E[i].v, E[i].vn, E[i].vh));
        }
        sh.grow();
    });
    // phase 2: the fields, in text order per key
    lap("intern");
    // per shard: the earliest conflicting key of each field (stamp = position of
    // the key's first pair in the update)
    constexpr uint64_t kNo = ~0ull;
    std::vector<uint64_t> bad_at((size_t)kShards * 3, kNo);
    std::vector<int32_t> bad_key((size_t)kShards * 3, -1);
    run_parallel(h->n_threads, kShards, [&](int s) {
        Shard& sh = h->shard[s];
        sh.saved_p.clear();
        sh.saved_r.clear();
        sh.saved_t.clear();
        for (int t = 0; t < T; ++t) {
            const std::vector<Ent>& E = ents[t];
            for (uint32_t i : kb[(size_t)t * kShards + s]) {
                const Ent& e = E[i];
                const int32_t k = ref_local(e.kref);
                const uint64_t at = ((uint64_t)t << 32) | i;
                if (e.has & (HAS_PARENT | HAS_NONE)) {
                    if (sh.p_file[k] != upd) {
                        if (sh.p_file[k] >= 0) sh.saved_p.push_back({k, sh.parent[k], at});
                        sh.p_file[k] = upd;
                    }
                    sh.parent[k] = (e.has & HAS_NONE) ? kNone : e.vref;
                }
                if (e.has & HAS_RANK) {
                    if (sh.r_file[k] != upd) {
                        if (sh.r_file[k] >= 0) sh.saved_r.push_back({k, sh.rank[k], at});
                        sh.r_file[k] = upd;
                    }
                    sh.rank[k] = e.rank + 1;
                }
                if (e.has & HAS_TITLE) {
                    if (sh.t_file[k] != upd) {
                        if (sh.t_file[k] >= 0) sh.saved_t.push_back({k, std::string(sh.titles.data() + sh.t_off[k], sh.t_len[k]), at});
                        sh.t_file[k] = upd;
                    }
                    sh.t_off[k] = (int64_t)sh.titles.size();
                    sh.t_len[k] = e.vn;
                    sh.titles.append(e.v, e.vn);
                }
            }
        }
        auto note = [&](int field, int32_t k, uint64_t at) {
            if (at < bad_at[(size_t)s * 3 + field]) {
                bad_at[(size_t)s * 3 + field] = at;
                bad_key[(size_t)s * 3 + field] = k;
            }
        };
        for (const auto& sv : sh.saved_p)
            if (sh.parent[sv.k] != sv.old) note(0, sv.k, sv.at);
        for (const auto& sv : sh.saved_r)
            if (sh.rank[sv.k] != sv.old) note(1, sv.k, sv.at);
        for (const auto& sv : sh.saved_t)
            if (sh.t_len[sv.k] != sv.old.size() || memcmp(sh.titles.data() + sh.t_off[sv.k], sv.old.data(), sv.old.size()) != 0)
                note(2, sv.k, sv.at);
    });
    lap("fields");
    // the reader's dicts are merged one after the other: tree, then ranks (workflow.py:754-757)
    for (int field = 0; field < 3; ++field) {
        int cs = -1;
        for (int s = 0; s < kShards; ++s)
            if (bad_at[(size_t)s * 3 + field] != kNo && (cs < 0 || bad_at[(size_t)s * 3 + field] < bad_at[(size_t)cs * 3 + field])) cs = s;
        if (cs < 0) continue;
        const Shard& sh = h->shard[cs];
        const int32_t k = bad_key[(size_t)cs * 3 + field];
        return hfail(h, WK_E_STATE, "Conflicting values found for \"" + std::string(sh.names.ptr(k), sh.names.len[k]) + "\".");
    }
    return WK_OK;
}

struct ParseOut {
    std::vector<Ent> ents;
    NameTable ranks;
    int status = 0;  // 0 ok, 1 = refuse (python reader), 2 = short line (IndexError)
};

// Fields of line [p, e) after `line.rstrip().replace('\t|', '').split('\t')`
// (woltka/tree.py:67,95): up to `want` fields, *nf = fields seen (capped at
// want).  false = a "\t|" sits between two parts of one field (the Python
// reader takes the file).
inline bool dmp_fields(const char* p, const char* e, int want, const char** fb, const char** fe, int* nf) {
    int n = 0;
    const char* start = p;  // start of the field's content
    const char* cend = p;   // end of its content
    bool content = false, gap = false;
    const char* c = p;
    while (c < e) {
        if (*c == '\t') {
            if (c + 1 < e && c[1] == '|') {  // removed
                if (content)
                    gap = true;
                c += 2;
                if (!content) start = cend = c;
                continue;
            }
            fb[n] = start;
            fe[n] = content ? cend : start;
            if (++n == want) {
                *nf = n;
                return true;
            }
            c += 1;
            start = cend = c;
            content = gap = false;
            continue;
        }
        if (gap) return false;
        if (!content) {
            start = c;
            content = true;
        }
        c += 1;
        cend = c;
    }
    fb[n] = start;
    fe[n] = content ? cend : start;
    *nf = n + 1;
    return true;
}

void parse_range(int kind, const char* b, const char* e, bool has_rank_arg, ParseOut& out) {
    static const char kSci[] = "scientific name";
    out.ents.reserve((size_t)((e - b) / 24 + 16));
    const char* p = b;
    while (p < e) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
        const char* le = nl ? nl : e;  // line without its newline
        const char* line = p;
        p = nl ? nl + 1 : e;
        // universal newlines: "\r\n" is one line end; any other '\r' splits lines in Python
        const char* cr = (const char*)memchr(line, '\r', (size_t)(le - line));
        if (cr && !(cr + 1 == le && nl)) {
            out.status = 1;
            return;
        }
        Ent en{};
        if (kind == WK_HIER_MAP) {
            // key, found, rest = line.partition('\t'); value = rest.partition('\t')[0].rstrip()
            const char* t1 = (const char*)memchr(line, '\t', (size_t)(le - line));
            if (!t1) continue;
            const char* vb = t1 + 1;
            const char* t2 = (const char*)memchr(vb, '\t', (size_t)(le - vb));
            const char* ve = t2 ? t2 : le;
            if (ve > vb && (unsigned char)ve[-1] >= 0x80) {
                out.status = 1;
                return;
            }
            while (ve > vb && py_space((unsigned char)ve[-1])) --ve;
            en.k = line;
            en.kn = (uint32_t)(t1 - line);
            en.v = vb;
            en.vn = (uint32_t)(ve - vb);
            en.has = HAS_PARENT;
        } else {
            const char* re = le;
            if (re > line && (unsigned char)re[-1] >= 0x80) {
                out.status = 1;
                return;
            }
            while (re > line && py_space((unsigned char)re[-1])) --re;
            const char* fb[4];
            const char* fe[4];
            int nf = 0;
            const int want = kind == WK_HIER_NAMES ? 4 : 3;
            if (!dmp_fields(line, re, want, fb, fe, &nf)) {
                out.status = 1;
                return;
            }
            if (nf < 2) {  // x[1]: IndexError
                out.status = 2;
                return;
            }
            en.k = fb[0];
            en.kn = (uint32_t)(fe[0] - fb[0]);
            en.v = fb[1];
            en.vn = (uint32_t)(fe[1] - fb[1]);
            if (kind == WK_HIER_NAMES) {
                if (nf >= 4 && !((size_t)(fe[3] - fb[3]) == sizeof kSci - 1 && memcmp(fb[3], kSci, sizeof kSci - 1) == 0)) continue;
                en.has = HAS_TITLE;
            } else {
                en.has = HAS_PARENT;
                if (nf >= 3) {
                    const size_t rn = (size_t)(fe[2] - fb[2]);
                    const uint64_t rh = hash_bytes(fb[2], rn);
                    en.rank = out.ranks.intern(fb[2], rn, rh);
                    en.has |= HAS_RANK;
                }
            }
        }
        en.kh = hash_bytes(en.k, en.kn);
        if (en.has & HAS_PARENT) en.vh = hash_bytes(en.v, en.vn);
        out.ents.push_back(en);
    }
    (void)has_rank_arg;
}

}  // namespace

extern "C" {

int wk_hier_create(int n_threads, wk_hier** out) {
    if (!out) return WK_E_ARG;
    wk_hier* h = new (std::nothrow) wk_hier();
    if (!h) return WK_E_HIP;
    if (n_threads <= 0) {
        n_threads = (int)std::thread::hardware_concurrency();
        if (n_threads <= 0) n_threads = 1;
    }
    h->n_threads = std::min(n_threads, 256);
    *out = h;
    return WK_OK;
}

void wk_hier_destroy(wk_hier* h) { delete h; }

const char* wk_hier_last_error(const wk_hier* h) { return h ? h->err.c_str() : "null hierarchy"; }

int wk_hier_add_text(wk_hier* h, int kind, const char* buf, int64_t len, const char* rank) {
    if (!h || len < 0 || (len > 0 && !buf)) return WK_E_ARG;
    if (kind != WK_HIER_NODES && kind != WK_HIER_MAP && kind != WK_HIER_NAMES) return hfail(h, WK_E_ARG, "unknown hierarchy file kind");
    if (h->finished) return hfail(h, WK_E_STATE, "hierarchy is already finished");
    int T = (int)std::max<int64_t>(1, std::min<int64_t>(h->n_threads, len >> 16));
    std::vector<const char*> cut((size_t)T + 1);
    const char* e = buf + len;
    cut[0] = buf;
    cut[T] = e;
    for (int i = 1; i < T; ++i) {
        const char* p = buf + len * i / T;
        if (p < cut[i - 1]) p = cut[i - 1];
        const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
        cut[i] = nl ? nl + 1 : e;
    }
    std::vector<ParseOut> po((size_t)T);
    Lap lap;
    run_parallel(h->n_threads, T, [&](int t) { parse_range(kind, cut[t], cut[t + 1], rank != nullptr, po[t]); });
    lap("parse");
    for (int t = 0; t < T; ++t) {
        if (po[t].status == 1) return hfail(h, WK_E_ARG, "text is left to the Python reader");
        if (po[t].status == 2) return hfail(h, WK_E_RANGE, "list index out of range");
    }
    // rank vocabulary: thread-local codes -> global codes, in text order
    for (int t = 0; t < T; ++t) {
        const NameTable& f = po[t].ranks;
        if (!f.size()) continue;
        std::vector<int32_t> remap((size_t)f.size());
        for (int32_t k = 0; k < f.size(); ++k) remap[k] = h->ranks.intern(f.ptr(k), f.len[k], f.hash[k]);
        for (Ent& en : po[t].ents)
            if (en.has & HAS_RANK) en.rank = remap[en.rank];
    }
    std::vector<std::vector<Ent>> ents((size_t)T);
    for (int t = 0; t < T; ++t) ents[t].swap(po[t].ents);
    int rc = apply_update(h, ents);
    if (rc) return rc;
    if (kind == WK_HIER_MAP && rank) {
        // update_dict(rankdic, {k: rank for k in set(map_.values())}) (workflow.py:803-805):
        // the values of the file's final dict, i.e. what its keys point at now
        const size_t rn = strlen(rank);
        const int32_t code = h->ranks.intern(rank, rn, hash_bytes(rank, rn));
        std::vector<std::vector<Ent>> rents((size_t)T);
        run_parallel(h->n_threads, T, [&](int t) {
            rents[t].reserve(ents[t].size());
            for (const Ent& en : ents[t]) {
                // (the parent a key ended up with; entries overwritten later in the file name a value the dict dropped)
                const Shard& ks = h->shard[ref_shard(en.kref)];
                const uint32_t pv = ks.parent[ref_local(en.kref)];
                if (pv != en.vref) continue;
                Ent r{};
                r.kref = pv;  // (no pointer into a name arena: interning may move it)
                r.kh = (uint64_t)ref_shard(pv) << 58;
                r.rank = code;
                r.has = HAS_RANK | KEY_RESOLVED;
                rents[t].push_back(r);
            }
        });
        rc = apply_update(h, rents);
    }
    return rc;
}

// update_dict(dic, other) with `other` given as arrays: field = WK_HIER_PARENT
// (values may be None: is_none[i]), WK_HIER_RANK or WK_HIER_NAME.
int wk_hier_update(wk_hier* h, int field, const char* kblob, const int64_t* koff, const char* vblob, const int64_t* voff,
                   const uint8_t* is_none, int64_t n) {
    if (!h || n < 0 || (n > 0 && (!koff || !voff))) return WK_E_ARG;
    if (field != WK_HIER_PARENT && field != WK_HIER_RANK && field != WK_HIER_NAME) return hfail(h, WK_E_ARG, "unknown hierarchy field");
    if (h->finished) return hfail(h, WK_E_STATE, "hierarchy is already finished");
    static const char kEmpty[1] = {0};
    if (!kblob) kblob = kEmpty;
    if (!vblob) vblob = kEmpty;
    int T = (int)std::max<int64_t>(1, std::min<int64_t>(h->n_threads, n >> 12));
    std::vector<std::vector<Ent>> ents((size_t)T);
    std::vector<NameTable> lranks((size_t)T);
    run_parallel(h->n_threads, T, [&](int t) {
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        ents[t].reserve((size_t)(hi - lo));
        for (int64_t i = lo; i < hi; ++i) {
            Ent en{};
            en.k = kblob + koff[i];
            en.kn = (uint32_t)(koff[i + 1] - koff[i]);
            en.kh = hash_bytes(en.k, en.kn);
            en.v = vblob + voff[i];
            en.vn = (uint32_t)(voff[i + 1] - voff[i]);
            if (field == WK_HIER_PARENT) {
                if (is_none && is_none[i]) {
                    en.has = HAS_NONE;
                } else {
                    en.has = HAS_PARENT;
                    en.vh = hash_bytes(en.v, en.vn);
                }
            } else if (field == WK_HIER_RANK) {
                en.has = HAS_RANK;
                en.rank = lranks[t].intern(en.v, en.vn, hash_bytes(en.v, en.vn));
            } else {
                en.has = HAS_TITLE;
            }
            ents[t].push_back(en);
        }
    });
    if (field == WK_HIER_RANK)
        for (int t = 0; t < T; ++t) {
            const NameTable& f = lranks[t];
            std::vector<int32_t> remap((size_t)f.size());
            for (int32_t k = 0; k < f.size(); ++k) remap[k] = h->ranks.intern(f.ptr(k), f.len[k], f.hash[k]);
            for (Ent& en : ents[t]) en.rank = remap[en.rank];
        }
    return apply_update(h, ents);
}

// tree.fill_root (tree.py:302-388) + the pre-order flattening.
int wk_hier_finish(wk_hier* h, int64_t* n_nodes, int32_t* n_ranks) {
    if (!h) return WK_E_ARG;
    if (h->finished) return hfail(h, WK_E_STATE, "hierarchy is already finished");
    // parents that are not keys of the tree join it as crowns (`tree[node] = None`)
    Lap lap;
    std::vector<std::vector<uint32_t>> missing(kShards);
    run_parallel(h->n_threads, kShards, [&](int s) {
        const Shard& sh = h->shard[s];
        for (int32_t k = 0; k < sh.names.size(); ++k) {
            const uint32_t p = sh.parent[k];
            if (p == kUnset || p == kNone) continue;
            if (h->shard[ref_shard(p)].parent[ref_local(p)] == kUnset) missing[s].push_back(p);
        }
    });
    for (int s = 0; s < kShards; ++s)
        for (uint32_t p : missing[s]) h->shard[ref_shard(p)].parent[ref_local(p)] = kNone;
    // crowns: their own parent, or None
    std::vector<std::vector<uint32_t>> crowns(kShards);
    std::vector<int64_t> in_tree(kShards, 0);
    run_parallel(h->n_threads, kShards, [&](int s) {
        const Shard& sh = h->shard[s];
        for (int32_t k = 0; k < sh.names.size(); ++k) {
            const uint32_t p = sh.parent[k];
            if (p == kUnset) continue;
            in_tree[s] += 1;
            if (p == kNone || p == make_ref(s, k)) crowns[s].push_back(make_ref(s, k));
        }
    });
    int64_t n = 0, n_crowns = 0;
    for (int s = 0; s < kShards; ++s) {
        n += in_tree[s];
        n_crowns += (int64_t)crowns[s].size();
    }
    h->has_root = false;
    if (n_crowns == 1) {
        for (int s = 0; s < kShards; ++s)
            if (!crowns[s].empty()) h->root_ref = crowns[s][0];
        h->shard[ref_shard(h->root_ref)].parent[ref_local(h->root_ref)] = h->root_ref;
        h->has_root = true;
    } else if (n_crowns > 1) {
        // a new root named by the smallest positive integer that is not a key
        char num[24];
        for (long long i = 1;; ++i) {
            const int ln = snprintf(num, sizeof num, "%lld", i);
            uint32_t r;
            if (h->find_ref(num, (size_t)ln, &r) == 0 && h->shard[ref_shard(r)].parent[ref_local(r)] != kUnset) continue;
            const uint64_t hv = hash_bytes(num, (size_t)ln);
            Shard& sh = h->shard[shard_of(hv)];
            const int32_t id = sh.names.intern(num, (size_t)ln, hv);
            sh.grow();
            h->root_ref = make_ref(shard_of(hv), id);
            sh.parent[id] = h->root_ref;
            in_tree[shard_of(hv)] += 1;
            n += 1;
            break;
        }
        for (int s = 0; s < kShards; ++s)
            for (uint32_t c : crowns[s]) h->shard[ref_shard(c)].parent[ref_local(c)] = h->root_ref;
        h->has_root = true;
    }
    h->n_nodes = n;
    h->finished = true;
    if (n_nodes) *n_nodes = n;
    if (n_ranks) *n_ranks = h->ranks.size();
    // rank usage (what `set(rankdic.values())` holds, workflow.py:665-669)
    h->rank_used.assign((size_t)h->ranks.size(), 0);
    for (int s = 0; s < kShards; ++s)
        for (int32_t c : h->shard[s].rank)
            if (c > 0) h->rank_used[(size_t)c - 1] += 1;
    if (n == 0) return WK_OK;
    if (!h->has_root) {
        // no node without a parent among the keys, none that is its own: every walk of fill_root ends in a cycle, it
        // adds nothing and returns None (tree.py:358-360) — build_hierarchy goes on with that.  The dict views hold the
        // nodes, the numbered tree is empty: every subject is a name outside it.
        h->n_nodes = 0;
        if (n_nodes) *n_nodes = 0;
        return WK_OK;
    }
    if (n > (int64_t)WK_MAX_FEATURE) return hfail(h, WK_E_RANGE, "too many hierarchy nodes");
    // node numbers in input order: shard-major, order of first appearance inside a shard
    std::vector<int64_t> base(kShards + 1, 0);
    for (int s = 0; s < kShards; ++s) base[s + 1] = base[s] + in_tree[s];
    run_parallel(h->n_threads, kShards, [&](int s) {
        Shard& sh = h->shard[s];
        sh.dense.assign((size_t)sh.names.size(), -1);
        int64_t d = base[s];
        for (int32_t k = 0; k < sh.names.size(); ++k)
            if (sh.parent[k] != kUnset) sh.dense[k] = d++;
    });
    std::vector<int64_t> par((size_t)n);
    std::vector<uint32_t> ref_of_dense((size_t)n);
    std::vector<int32_t> code_of_dense((size_t)n);
    run_parallel(h->n_threads, kShards, [&](int s) {
        const Shard& sh = h->shard[s];
        for (int32_t k = 0; k < sh.names.size(); ++k) {
            const int64_t d = sh.dense[k];
            if (d < 0) continue;
            const uint32_t p = sh.parent[k];
            par[(size_t)d] = h->shard[ref_shard(p)].dense[ref_local(p)];
            ref_of_dense[(size_t)d] = make_ref(s, k);
            code_of_dense[(size_t)d] = sh.rank[k];
        }
    });
    lap("fill_root+number");
    int64_t root = h->shard[ref_shard(h->root_ref)].dense[ref_local(h->root_ref)];
    std::vector<int64_t> pre((size_t)n), size((size_t)n), depth((size_t)n);
    int64_t bad = -1;
    int rc = wk_preorder(par.data(), n, root, pre.data(), size.data(), depth.data(), &bad);
    if (rc == WK_E_STATE) {
        // A cycle beside the rooted part.  tree.fill_root lets it stand (its walk stops at a node it has tested,
        // tree.py:329-353) and the reference only fails — by never returning — once a read walks into it
        // (tree.py:418-429).  Here the nodes that cannot reach the root stay in the dict views and leave the numbered
        // tree: a subject among them is a name outside the tree (it assigns nothing), everything else is classified
        // as the reference does.  (state: 1 reaches the root, 2 does not, 3 on the walk in progress)
        std::vector<uint8_t> st((size_t)n, 0);
        st[(size_t)root] = 1;
        std::vector<int64_t> walk;
        for (int64_t d = 0; d < n; ++d) {
            if (st[(size_t)d]) continue;
            walk.clear();
            int64_t cur = d;
            while (st[(size_t)cur] == 0) {
                st[(size_t)cur] = 3;
                walk.push_back(cur);
                cur = par[(size_t)cur];
            }
            const uint8_t res = st[(size_t)cur] == 1 ? 1 : 2;
            for (int64_t v : walk) st[(size_t)v] = res;
        }
        std::vector<int64_t> newid((size_t)n, -1);
        int64_t m = 0;
        for (int64_t d = 0; d < n; ++d)
            if (st[(size_t)d] == 1) newid[(size_t)d] = m++;
        std::vector<int64_t> par2((size_t)m);
        std::vector<uint32_t> ref2((size_t)m);
        std::vector<int32_t> code2((size_t)m);
        for (int64_t d = 0; d < n; ++d) {
            const int64_t v = newid[(size_t)d];
            if (v < 0) continue;
            par2[(size_t)v] = newid[(size_t)par[(size_t)d]];  // (the parent of a node that reaches the root does too)
            ref2[(size_t)v] = ref_of_dense[(size_t)d];
            code2[(size_t)v] = code_of_dense[(size_t)d];
        }
        for (int s = 0; s < kShards; ++s)
            for (int64_t& d : h->shard[s].dense)
                if (d >= 0) d = newid[(size_t)d];
        par.swap(par2);
        ref_of_dense.swap(ref2);
        code_of_dense.swap(code2);
        root = newid[(size_t)root];
        n = m;
        h->n_nodes = n;
        if (n_nodes) *n_nodes = n;
        pre.resize((size_t)n);
        size.resize((size_t)n);
        depth.resize((size_t)n);
        rc = wk_preorder(par.data(), n, root, pre.data(), size.data(), depth.data(), &bad);
    }
    lap("preorder");
    if (rc == WK_E_STATE) {
        uint32_t ln = 0;
        const char* nm = h->sym_name(ref_of_dense[(size_t)bad], &ln);
        return hfail(h, WK_E_STATE, "Node \"" + std::string(nm, ln) + "\" cannot reach the root (cyclic hierarchy).");
    }
    if (rc) return hfail(h, WK_E_STATE, "Hierarchy must have exactly one root.");
    h->parent.resize((size_t)n);
    h->last.resize((size_t)n);
    h->rank_code.resize((size_t)n);
    h->depth.resize((size_t)n);
    h->ref_of_pre.resize((size_t)n);
    h->pre_of_dense.swap(pre);
    const std::vector<int64_t>& P = h->pre_of_dense;
    const int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(h->n_threads, n >> 14));
    run_parallel(h->n_threads, chunks, [&](int c) {
        const int64_t lo = n * c / chunks, hi = n * (c + 1) / chunks;
        for (int64_t d = lo; d < hi; ++d) {
            const int64_t v = P[(size_t)d];
            h->parent[(size_t)v] = (int32_t)P[(size_t)par[(size_t)d]];
            h->last[(size_t)v] = (int32_t)(v + size[(size_t)d] - 1);
            h->rank_code[(size_t)v] = code_of_dense[(size_t)d];
            h->depth[(size_t)v] = (int32_t)depth[(size_t)d];
            h->ref_of_pre[(size_t)v] = ref_of_dense[(size_t)d];
        }
    });
    lap("arrays");
    return WK_OK;
}

int wk_hier_arrays(const wk_hier* h, int32_t* parent, int32_t* last, int32_t* rank_code, int32_t* depth) {
    if (!h || !h->finished) return WK_E_STATE;
    const size_t b = (size_t)h->parent.size() * 4;
    if (parent && b) memcpy(parent, h->parent.data(), b);
    if (last && b) memcpy(last, h->last.data(), b);
    if (rank_code && b) memcpy(rank_code, h->rank_code.data(), b);
    if (depth && b) memcpy(depth, h->depth.data(), b);
    return WK_OK;
}

int wk_hier_root(const wk_hier* h, int32_t* root_id) {
    if (!h || !h->finished || !root_id) return WK_E_STATE;
    *root_id = (h->has_root && h->n_nodes > 0) ? 0 : -1;
    return WK_OK;
}

// Pre-order node ids of names (-1: not a node of the tree).
int wk_hier_lookup(const wk_hier* h, const char* blob, const int64_t* off, int64_t n, int32_t* out) {
    if (!h || !h->finished || n < 0 || (n > 0 && (!off || !out))) return WK_E_ARG;
    static const char kEmpty[1] = {0};
    if (!blob) blob = kEmpty;
    const int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(h->n_threads, n >> 13));
    run_parallel(h->n_threads, chunks, [&](int c) {
        const int64_t lo = n * c / chunks, hi = n * (c + 1) / chunks;
        for (int64_t i = lo; i < hi; ++i) {
            uint32_t r;
            out[i] = -1;
            if (h->find_ref(blob + off[i], (size_t)(off[i + 1] - off[i]), &r)) continue;
            const Shard& sh = h->shard[ref_shard(r)];
            if (sh.dense.empty()) continue;
            const int64_t d = sh.dense[ref_local(r)];
            if (d >= 0) out[i] = (int32_t)h->pre_of_dense[(size_t)d];
        }
    });
    return WK_OK;
}

// Names of pre-order node ids: off[n + 1] always, bytes into blob when given
// (cap checked).
int wk_hier_node_names(const wk_hier* h, const int32_t* ids, int64_t n, char* blob, int64_t cap, int64_t* off) {
    if (!h || !h->finished || n < 0 || (n > 0 && (!ids || !off)) || !off) return WK_E_ARG;
    int64_t w = 0;
    off[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (ids[i] < 0 || ids[i] >= h->n_nodes) return WK_E_ARG;
        uint32_t ln = 0;
        const char* p = h->sym_name(h->ref_of_pre[(size_t)ids[i]], &ln);
        if (blob) {
            if (w + ln > cap) return WK_E_CAPACITY;
            memcpy(blob + w, p, ln);
        }
        w += ln;
        off[i + 1] = w;
    }
    return WK_OK;
}

// One entry of a dict view: field = WK_HIER_PARENT (tree[key], after
// fill_root), WK_HIER_RANK (rankdic[key]) or WK_HIER_NAME (namedic[key]).
// *len = -1: key absent; else the value's length (bytes copied when they fit).
int wk_hier_get(const wk_hier* h, int field, const char* key, int64_t klen, char* out, int64_t cap, int64_t* len) {
    if (!h || !h->finished || klen < 0 || !len) return WK_E_ARG;
    *len = -1;
    uint32_t r;
    if (h->find_ref(key ? key : "", (size_t)klen, &r)) return WK_OK;
    const Shard& sh = h->shard[ref_shard(r)];
    const int32_t k = ref_local(r);
    const char* p = nullptr;
    uint32_t ln = 0;
    if (field == WK_HIER_PARENT) {
        if (sh.parent[k] == kUnset) return WK_OK;
        if (sh.parent[k] == kNone) return WK_E_STATE;  // (cannot happen after fill_root with a root)
        p = h->sym_name(sh.parent[k], &ln);
    } else if (field == WK_HIER_RANK) {
        if (sh.rank[k] <= 0) return WK_OK;
        p = h->ranks.ptr(sh.rank[k] - 1);
        ln = h->ranks.len[sh.rank[k] - 1];
    } else if (field == WK_HIER_NAME) {
        if (sh.t_off[k] < 0) return WK_OK;
        p = sh.titles.data() + sh.t_off[k];
        ln = sh.t_len[k];
    } else {
        return WK_E_ARG;
    }
    *len = ln;
    if (out && (int64_t)ln <= cap && ln) memcpy(out, p, ln);
    return WK_OK;
}

int64_t wk_hier_size(const wk_hier* h, int field) {
    if (!h) return 0;
    int64_t n = 0;
    for (int s = 0; s < kShards; ++s) {
        const Shard& sh = h->shard[s];
        if (field == WK_HIER_PARENT)
            for (uint32_t p : sh.parent) n += p != kUnset;
        else if (field == WK_HIER_RANK)
            for (int32_t c : sh.rank) n += c > 0;
        else
            for (int64_t o : sh.t_off) n += o >= 0;
    }
    return n;
}

// All keys of a dict view: off[size + 1] always, bytes when blob is given.
int wk_hier_keys(const wk_hier* h, int field, char* blob, int64_t cap, int64_t* off) {
    if (!h || !off) return WK_E_ARG;
    int64_t w = 0, i = 0;
    off[0] = 0;
    for (int s = 0; s < kShards; ++s) {
        const Shard& sh = h->shard[s];
        for (int32_t k = 0; k < sh.names.size(); ++k) {
            const bool has = field == WK_HIER_PARENT ? sh.parent[k] != kUnset
                             : field == WK_HIER_RANK ? sh.rank[k] > 0
                                                     : sh.t_off[k] >= 0;
            if (!has) continue;
            const uint32_t ln = sh.names.len[k];
            if (blob) {
                if (w + ln > cap) return WK_E_CAPACITY;
                memcpy(blob + w, sh.names.ptr(k), ln);
            }
            w += ln;
            off[++i] = w;
        }
    }
    return WK_OK;
}

// Rank vocabulary: names (code = index + 1) and how many keys carry each.
int wk_hier_ranks(const wk_hier* h, char* blob, int64_t cap, int64_t* off, int64_t* used) {
    if (!h || !off) return WK_E_ARG;
    int64_t w = 0;
    off[0] = 0;
    for (int32_t i = 0; i < h->ranks.size(); ++i) {
        const uint32_t ln = h->ranks.len[i];
        if (blob) {
            if (w + ln > cap) return WK_E_CAPACITY;
            memcpy(blob + w, h->ranks.ptr(i), ln);
        }
        w += ln;
        off[i + 1] = w;
        if (used) used[i] = (size_t)i < h->rank_used.size() ? h->rank_used[(size_t)i] : 0;
    }
    return WK_OK;
}

}  // extern "C"
