// wk_tokenize.cpp — native multi-threaded SAM tokenizer / packer (host side).
//
// Replaces the per-line Python of the reference's SAM parsers + plain_mapper /
// ordinal_mapper packing (woltka/align.py:258-406, 550-583; ordinal.py:219-237)
// for the benchmarked input format.  Same semantics:
//   * leading '@' lines are the header; records with RNAME '*' are skipped
//     before the QNAME-change test (align.py:295-300, 318-319);
//   * a run of consecutive mapped records with the same QNAME yields up to
//     three reads — unpaired, /1, /2 — chosen by (FLAG >> 6) & 3
//     (align.py:322-333); both mate bits set is an error;
//   * with an exclusion set, a run is dropped entirely once one of its records
//     names an excluded subject (align.py:443-469);
//   * "extra" flavour: POS-1, CIGAR -> (aligned length, reference span)
//     (align.py:376-398, 572-583); zero-length hits are dropped
//     (ordinal.py:231).
// Subjects are interned into dense indices in order of first appearance (the
// indices wk_set_subjects expects).  A block of text is cut into byte ranges at
// run boundaries, one range per thread; ranges are tokenised independently and
// concatenated, so the output does not depend on the thread count.
#include <immintrin.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/woltka_hip.h"
#include "wk_names.hpp"

namespace {

using wkh::hash_bytes;
using wkh::NameTable;

struct Record {
    int32_t subj;  // global id >= 0, or -(1 + local new-name id)
    int32_t beg, end;
    uint32_t len;
};

// Read-only lookup structure over the subject dictionary for the tokenizer
// threads: one 16-byte slot per name {hash, id, offset of the name's bytes}, the
// bytes behind a 4-byte length in an arena of their own — a hit costs one slot
// and one short compare (NameTable: slot -> hash[] -> len[] -> arena).
struct FastDict {
    struct Slot {
        uint64_t hash;
        int32_t id;  // -1 = empty
        uint32_t off;
    };
    std::vector<Slot> slot;
    std::string arena;
    size_t mask = 0;
    int32_t count = 0;
    FastDict() { resize(1 << 12); }
    void resize(size_t n) {
        std::vector<Slot> old;
        old.swap(slot);
        slot.assign(n, Slot{0, -1, 0});
        mask = n - 1;
        for (const Slot& x : old)
            if (x.id >= 0) place(x);
    }
    void place(const Slot& x) {
        size_t h = x.hash & mask;
        while (slot[h].id >= 0) h = (h + 1) & mask;
        slot[h] = x;
    }
    void insert(const char* p, size_t n, uint64_t hv, int32_t id) {
        if ((size_t)(count + 1) * 2 > slot.size()) resize(slot.size() * 2);
        const uint32_t ln = (uint32_t)n;
        const uint32_t off = (uint32_t)arena.size();
        arena.append(reinterpret_cast<const char*>(&ln), 4);
        arena.append(p, n);
        place(Slot{hv, id, off});
        count += 1;
    }
    void prefetch(uint64_t hv) const { __builtin_prefetch(&slot[hv & mask]); }
    int32_t find(const char* p, size_t n, uint64_t hv) const {
        size_t h = hv & mask;
        for (;;) {
            const Slot& x = slot[h];
            if (x.id < 0) return -1;
            if (x.hash == hv) {
                uint32_t ln;
                memcpy(&ln, arena.data() + x.off, 4);
                if (ln == n && memcmp(arena.data() + x.off + 4, p, n) == 0) return x.id;
            }
            h = (h + 1) & mask;
        }
    }
    void clear() {
        slot.assign(slot.size(), Slot{0, -1, 0});
        arena.clear();
        count = 0;
    }
};

// What one tokenizer thread produces for its byte range; kept between calls so
// that the buffers are allocated (and their pages touched) once.
struct Local {
    // records of emitted reads, read-major: plain flavour `subj` only, "ex"
    // flavour `rec`
    std::vector<int32_t> subj;   // global id >= 0, or -(1 + local new-name id)
    std::vector<Record> rec;
    std::vector<int32_t> rend;   // per read: end offset into the records
    std::vector<uint64_t> qname; // per read: (offset << 24) | (len << 2) | mate
    std::vector<int32_t> group;  // per read: stratum id or -1 (want_groups)
    std::string gkeys;             // ... their keys (read id + mate suffix), looked up together at the end of the range
    std::vector<uint32_t> gkey_off;
    std::vector<int32_t> sample; // per read: sample id >= 0, or -(1 + local new-name id) (want_samples)
    NameTable fresh_samples;
    NameTable fresh;             // names not yet in the global table
    int error = 0;               // 1 = both mate bits, 2 = malformed line
    size_t error_at = 0;
    // SAM, "ex" flavour with an exclusion set: what parse_sam_file_ex_ft's
    // variables hold at the end of the range (align.py:509-547) — the last
    // query, whether it is kept, and the lines whose records sit in `pool`:
    // those of the last run that was not excluded at its first line, up to the
    // record that excluded it (a later run excluded at its first line leaves
    // the pool alone)
    bool any_run = false, fin_keep = true, have_pool = false;
    const char* fin_q = nullptr;
    size_t fin_qn = 0;
    std::vector<std::pair<const char*, const char*>> pool_lines;
    int64_t n_big = 0;  // reads with more than WK_WEIGHT_MAX_K records
    void reset() {
        subj.clear();
        rec.clear();
        rend.clear();
        qname.clear();
        group.clear();
        gkeys.clear();
        gkey_off.clear();
        sample.clear();
        if (fresh.size()) fresh = NameTable();
        if (fresh_samples.size()) fresh_samples = NameTable();
        error = 0;
        error_at = 0;
        any_run = false;
        fin_keep = true;
        have_pool = false;
        fin_q = nullptr;
        fin_qn = 0;
        pool_lines.clear();
        n_big = 0;
    }
    size_t n_records(bool extra) const { return extra ? rec.size() : subj.size(); }
};

struct Line {
    const char* q;
    size_t qn;
    const char* r;
    size_t rn;
    int flag;
    const char* pos;
    const char* pos_end;
    const char* cigar;
    size_t cn;
    bool ok;
    // map / b6o / paf rows of the "ex" flavour carry their numbers directly
    int32_t beg, end;
    uint32_t len;
    bool bad_number;  // a field Python's int() would refuse (the reference raises; SAM: the FLAG, once the line is kept)
    // the line itself (without its newline) and the hash of its subject name
    const char* line;
    const char* le;
    uint64_t rh;
};

constexpr size_t kScanSlack = 64;  // bytes the vector scanner wants ahead of a line start

inline bool is_unmapped(const Line& L) { return L.rn == 1 && L.r[0] == '*'; }

// decimal integer like Python's int() on a clean field: optional sign, digits,
// single underscores between digits, blanks around.  Python's integers have no
// width: a value of more than 18 significant digits SATURATES here at +-kHuge
// (every caller tests the range its field has, so the text is a loud error and
// never a wrapped number), `wrap` = keep the low 64 bits instead (what
// `int(flag) >> 6 & 3` looks at, exact for any length).  More than 4300 digits:
// int() itself refuses (sys.get_int_max_str_digits()).
constexpr long kHuge = 1l << 62;
template <bool wrap = false>
inline bool parse_int(const char* p, const char* e, long& v) {
    auto blank = [](char c) { return c == ' ' || (c >= '\t' && c <= '\r'); };
    while (p < e && blank(*p)) ++p;  // int() ignores surrounding whitespace
    while (e > p && blank(e[-1])) --e;
    bool neg = false;
    if (p < e && (*p == '-' || *p == '+')) neg = *p++ == '-';
    if (p >= e) return false;
    unsigned long x = 0;
    bool huge = false;
    long digits = 0;
    for (const char* b = p; p < e; ++p) {
        // (int() takes single underscores between digits)
        if (*p == '_' && p > b && p[-1] != '_' && p + 1 < e && p[1] >= '0' && p[1] <= '9') continue;
        if (*p < '0' || *p > '9') return false;
        ++digits;
        const unsigned long d = (unsigned long)(*p - '0');
        if (wrap) {
            x = x * 10ul + d;
        } else if (x > ((unsigned long)kHuge - 9ul) / 10ul) {
            huge = true;
        } else {
            x = x * 10ul + d;
        }
    }
    if (digits > 4300) return false;
    if (wrap)
        v = (long)(neg ? 0ul - x : x);
    else
        v = huge ? (neg ? -kHuge : kHuge) : (neg ? -(long)x : (long)x);
    return true;
}

inline bool fits_i32(long v) { return v >= INT32_MIN && v <= INT32_MAX; }

// FLAG as int() reads it (align.py:322): the value's low bits (the mate bits are all that is looked at)
inline bool parse_flag(const char* b, const char* e, int& flag) {
    long v;
    if (!parse_int<true>(b, e, v)) return false;
    flag = (int)(v & 0x7FFFFFFFl);
    return true;
}

// split the first 3 (or 6) tab-separated fields of [p, e)
inline Line parse_line(const char* p, const char* e, bool extra) {
    Line L{};
    const char* t1 = (const char*)memchr(p, '\t', e - p);
    if (!t1) return L;
    const char* t2 = (const char*)memchr(t1 + 1, '\t', e - t1 - 1);
    if (!t2) return L;
    const char* t3 = (const char*)memchr(t2 + 1, '\t', e - t2 - 1);
    if (!t3) return L;
    L.q = p;
    L.qn = t1 - p;
    L.r = t2 + 1;
    L.rn = t3 - t2 - 1;
    int f = 0;
    // (an unmapped record is skipped before its FLAG is looked at, align.py:318-322;
    // an empty FLAG, or text int() refuses, raises)
    if (!is_unmapped(L) && !parse_flag(t1 + 1, t2, f)) L.bad_number = true;  // (raises where the FLAG is converted)
    L.flag = f;
    if (extra) {
        const char* t4 = (const char*)memchr(t3 + 1, '\t', e - t3 - 1);
        if (!t4) return L;
        const char* t5 = (const char*)memchr(t4 + 1, '\t', e - t4 - 1);
        if (!t5) return L;
        const char* t6 = (const char*)memchr(t5 + 1, '\t', e - t5 - 1);
        if (!t6) return L;
        L.pos = t3 + 1;
        L.pos_end = t4;
        L.cigar = t5 + 1;
        L.cn = t6 - t5 - 1;
    }
    L.ok = true;
    return L;
}



// Would Python's float() take the text?  (ASCII forms: blanks around, a sign, then
// "inf" / "infinity" / "nan" in any case, or digits with single underscores between
// them, a point, an exponent.  strtod takes more -- hex floats, "nan(...)" -- and
// knows no underscores.)
inline bool py_float_ok(const char* p, const char* e) {
    auto blank = [](char c) { return c == ' ' || (c >= '\t' && c <= '\r'); };
    while (p < e && blank(*p)) ++p;
    while (e > p && blank(e[-1])) --e;
    if (p < e && (*p == '-' || *p == '+')) ++p;
    auto word = [&](const char* w) {
        size_t n = strlen(w);
        if ((size_t)(e - p) != n) return false;
        for (size_t i = 0; i < n; ++i)
            if ((p[i] | 0x20) != w[i]) return false;
        return true;
    };
    if (word("inf") || word("infinity") || word("nan")) return true;
    auto digits = [&]() {  // digit (["_"] digit)*: how many digits
        int n = 0;
        while (p < e) {
            if (*p >= '0' && *p <= '9') {
                ++n;
                ++p;
            } else if (*p == '_' && n > 0 && p + 1 < e && p[1] >= '0' && p[1] <= '9') {
                ++p;
            } else {
                break;
            }
        }
        return n;
    };
    int n = digits();
    if (p < e && *p == '.') {
        ++p;
        n += digits();
    }
    if (n == 0) return false;
    if (p < e && (*p == 'e' || *p == 'E')) {
        ++p;
        if (p < e && (*p == '-' || *p == '+')) ++p;
        if (digits() == 0) return false;
    }
    return p == e;
}

// Row extractors of the simple formats (align.py: parse_map_file :621,
// parse_b6o_file :753 / _ex :807, parse_paf_file :984 / _ex :1046): a line
// that does not qualify is skipped (ok = false), like the reference's
// `except IndexError: continue`.
inline Line parse_row(int fmt, const char* p, const char* e, bool extra) {
    if (fmt == WK_FMT_SAM) return parse_line(p, e, extra);
    Line L{};
    const char* f[13];  // field starts; f[i + 1] - 1 is the tab that ends field i
    int nf = 0;
    f[0] = p;
    const int want = fmt == WK_FMT_MAP ? 2 : (extra ? 12 : (fmt == WK_FMT_B6O ? 3 : 7));
    for (const char* c = p; nf < want;) {
        const char* t = (const char*)memchr(c, '\t', e - c);
        if (!t) {
            f[++nf] = e + 1;  // last field runs to the end of the line
            break;
        }
        f[++nf] = t + 1;
        c = t + 1;
    }
    // nf = number of fields seen (up to `want`); field i = [f[i], f[i + 1] - 1)
    auto fb = [&](int i) { return f[i]; };
    auto fe = [&](int i) { return f[i + 1] - 1; };
    if (fmt == WK_FMT_MAP) {
        if (nf < 2) return L;  // no tab
        L.q = fb(0);
        L.qn = fe(0) - fb(0);
        const char* rb = fb(1);
        const char* re = wkh::py_rstrip(rb, fe(1));  // subject.rstrip() (align.py:653)
        L.r = rb;
        L.rn = re - rb;
        L.ok = true;
        return L;
    }
    if (nf < want) {
        // (parse_b6o_file_ex evaluates int(x[3]) before it misses x[11], align.py:832: a
        // short line whose fourth field is no number raises instead of being skipped)
        long n3;
        if (extra && fmt == WK_FMT_B6O && nf >= 4 && !parse_int(fb(3), fe(3), n3)) L.bad_number = true;
        return L;
    }
    const int sub = fmt == WK_FMT_B6O ? 1 : 5;
    L.q = fb(0);
    L.qn = fe(0) - fb(0);
    L.r = fb(sub);
    L.rn = fe(sub) - fb(sub);
    if (!extra) {
        // (b6o: "a\tb" without a third column is not a row; the third split
        // part exists only behind a second tab)
        L.ok = true;
        return L;
    }
    long a, b, n;
    if (fmt == WK_FMT_B6O) {
        // length = int(x[3]); start, end = sorted(int(x[8]), int(x[9])); x[11] must exist
        if (!parse_int(fb(3), fe(3), n) || !parse_int(fb(8), fe(8), a) || !parse_int(fb(9), fe(9), b)) {
            L.bad_number = true;
            return L;
        }
        if (!py_float_ok(fb(11), fe(11))) {  // score = float(x[11]) must parse as well (align.py:832)
            L.bad_number = true;
            return L;
        }
        // (coordinates are held in 32 bits: anything wider is a loud error)
        if (!fits_i32(a) || !fits_i32(b) || !fits_i32((a < b ? a : b) - 1) || n > (long)UINT32_MAX || n < -(long)UINT32_MAX) {
            L.bad_number = true;
            return L;
        }
        L.len = (uint32_t)n;
        L.beg = (int32_t)((a < b ? a : b) - 1);
        L.end = (int32_t)(a < b ? b : a);
    } else {
        // length = int(x[10]), start = int(x[7]), end = int(x[8]), score = int(x[11])
        long sc;
        if (!parse_int(fb(10), fe(10), n) || !parse_int(fb(7), fe(7), a) || !parse_int(fb(8), fe(8), b) ||
            !parse_int(fb(11), fe(11), sc))
            return L;  // ValueError is caught there: the row is skipped
        if (!fits_i32(a) || !fits_i32(b) || n > (long)UINT32_MAX || n < -(long)UINT32_MAX) {
            L.bad_number = true;  // (numbers, but wider than the 32 bits they are held in)
            return L;
        }
        L.len = (uint32_t)n;
        L.beg = (int32_t)a;
        L.end = (int32_t)b;
    }
    L.ok = true;
    return L;
}

inline bool is_row(int fmt, const Line& L) { return L.ok && !(fmt == WK_FMT_SAM && is_unmapped(L)); }

// align.cigar_to_lens (align.py:550-583).  false where the reference raises:
// int() of the characters collected since the last operation runs only for
// M = X D N, and refuses an empty string or anything int() does.
inline bool cigar_lens(const char* c, size_t n, int64_t& aligned, int64_t& span) {
    constexpr int64_t kWide = 1ll << 40;  // sums saturate here: far beyond the 32 bits the caller accepts
    int64_t a = 0, x = 0;
    size_t s = 0;  // start of the text collected since the last operation
    for (size_t i = 0; i < n; ++i) {
        const char ch = c[i];
        const bool lens = ch == 'M' || ch == '=' || ch == 'X', skip = ch == 'D' || ch == 'N';
        if (lens || skip) {
            long v;
            if (!parse_int(c + s, c + i, v)) return false;  // int(n), with what int() takes (sign, blanks, underscores)
            int64_t& acc = skip ? x : a;
            acc = std::max<int64_t>(-kWide, std::min<int64_t>(kWide, acc + std::max<long>(-kWide, std::min<long>(kWide, v))));
            s = i + 1;
        } else if (ch == 'I' || ch == 'H' || ch == 'P' || ch == 'S') {
            s = i + 1;
        }
    }
    aligned = a;
    span = a + x;
    return true;
}

inline const char* next_line(const char* p, const char* e) {
    const char* nl = (const char*)memchr(p, '\n', e - p);
    return nl ? nl + 1 : e;
}

}  // namespace

// WOLTKA_TOK_TIMING=1 in the environment: time per phase, summed over the
// tokenizer's life and printed when it is destroyed (measurement)
struct TokLap {
    static bool enabled() {
        static const bool on = getenv("WOLTKA_TOK_TIMING") != nullptr;
        return on;
    }
    double* acc;
    std::chrono::steady_clock::time_point t;
    explicit TokLap(double* a) : acc(a) {
        if (enabled()) t = std::chrono::steady_clock::now();
    }
    void operator()(int phase) {
        if (!enabled()) return;
        const auto n = std::chrono::steady_clock::now();
        acc[phase] += std::chrono::duration<double, std::milli>(n - t).count();
        t = n;
    }
};
enum { LAP_CUTS, LAP_TOKENIZE, LAP_MERGE, LAP_FETCH, LAP_N };

// Worker threads that live as long as the tokenizer: a block is tokenised in
// two or three parallel steps, and a step of a few milliseconds does not pay
// for creating its threads.
class WorkPool {
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable wake, done;
    const std::function<void(int)>* job = nullptr;
    int n_items = 0, next = 0, running = 0;
    uint64_t epoch = 0;
    bool stop = false;

    void loop() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            wake.wait(lk, [&] { return stop || epoch != seen; });
            if (stop) return;
            seen = epoch;
            while (next < n_items) {
                const int i = next++;
                lk.unlock();
                (*job)(i);
                lk.lock();
            }
            if (--running == 0) done.notify_all();
        }
    }

  public:
    explicit WorkPool(int n) {
        for (int i = 0; i < n; ++i) workers.emplace_back([this] { loop(); });
    }
    ~WorkPool() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        wake.notify_all();
        for (auto& w : workers) w.join();
    }
    int size() const { return (int)workers.size(); }
    // fn(0) .. fn(n - 1), each once, on the workers; returns when all are done
    void run(int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        if (n == 1 || workers.empty()) {
            for (int i = 0; i < n; ++i) fn(i);
            return;
        }
        std::unique_lock<std::mutex> lk(m);
        job = &fn;
        n_items = n;
        next = 0;
        running = (int)workers.size();
        ++epoch;
        wake.notify_all();
        done.wait(lk, [&] { return running == 0; });
        job = nullptr;
    }
};

// One shard of the read id -> stratum table of a sample (tens of millions of
// entries: every probe is a cache miss).  A slot carries 32 bits of the hash next
// to the entry's index, an entry its key's place and its label: a hit costs the
// slot, the entry and the key (three lines), a miss usually the slot alone.
struct StrataShard {
    struct Ent {
        uint32_t off, len;
        int32_t label;
    };
    std::vector<uint64_t> slot;  // (tag << 32) | (entry + 1); 0 = empty
    std::vector<Ent> ent;
    std::string arena;
    size_t mask = 0;
    StrataShard() { rehash(1 << 10); }
    static uint32_t tag_of(uint64_t hv) { return (uint32_t)(hv >> 26); }
    void rehash(size_t n) {
        slot.assign(n, 0ull);
        mask = n - 1;
        for (size_t id = 0; id < ent.size(); ++id) {
            const uint64_t hv = hash_bytes(arena.data() + ent[id].off, ent[id].len);
            size_t h = hv & mask;
            while (slot[h]) h = (h + 1) & mask;
            slot[h] = ((uint64_t)tag_of(hv) << 32) | (uint64_t)(id + 1);
        }
    }
    void reserve(size_t n, size_t bytes) {
        size_t want = slot.size();
        while (want < 2 * n + 2) want <<= 1;
        if (want != slot.size()) rehash(want);
        ent.reserve(n);
        arena.reserve(bytes);
    }
    int32_t find_entry(const char* p, size_t n, uint64_t hv) const {
        const uint32_t tag = tag_of(hv);
        for (size_t h = hv & mask;; h = (h + 1) & mask) {
            const uint64_t v = slot[h];
            if (!v) return -1;
            if ((uint32_t)(v >> 32) != tag) continue;
            const Ent& e = ent[(uint32_t)v - 1u];
            if (e.len == n && memcmp(arena.data() + e.off, p, n) == 0) return (int32_t)((uint32_t)v - 1u);
        }
    }
    int32_t find(const char* p, size_t n, uint64_t hv) const {
        const int32_t id = find_entry(p, n, hv);
        return id < 0 ? -1 : ent[(size_t)id].label;
    }
    void put(const char* p, size_t n, uint64_t hv, int32_t label) {  // a repeated key keeps its last label, like dict()
        const int32_t id = find_entry(p, n, hv);
        if (id >= 0) {
            ent[(size_t)id].label = label;
            return;
        }
        if ((ent.size() + 1) * 2 > slot.size()) rehash(slot.size() * 2);
        ent.push_back(Ent{(uint32_t)arena.size(), (uint32_t)n, label});
        arena.append(p, n);
        size_t h = hv & mask;
        while (slot[h]) h = (h + 1) & mask;
        slot[h] = ((uint64_t)tag_of(hv) << 32) | (uint64_t)ent.size();
    }
    void prefetch_slot(uint64_t hv) const { __builtin_prefetch(&slot[hv & mask]); }
    void prefetch_entry(uint64_t hv) const {  // (the first slot of the probe sequence: where a key usually is)
        const uint64_t v = slot[hv & mask];
        if (v) __builtin_prefetch(&ent[(uint32_t)v - 1u]);
    }
    void prefetch_key(uint64_t hv) const {
        const uint64_t v = slot[hv & mask];
        if (v) __builtin_prefetch(arena.data() + ent[(uint32_t)v - 1u].off);
    }
};

struct wk_tok {
    int n_threads = 1;
    NameTable names;     // global subject dictionary (sidx = id)
    FastDict dict;       // the same names, laid out for the tokenizer threads' lookups
    NameTable exclude;
    std::string err;
    WorkPool* pool = nullptr;
    // results of the last call, per thread range, until they are fetched
    std::vector<Local> loc;
    int n_loc = 0;
    bool last_extra = false;
    int last_want = 0;
    std::vector<std::vector<int32_t>> remap, sremap;  // fresh name ids -> global ids, per range
    std::vector<int64_t> rbase, qbase;
    int64_t tot_reads = 0, tot_rec = 0, tot_big = 0;
    int (*produce_sam)(const char*&, const char*, bool, const FastDict&, Line*, int) = nullptr;
    // optional translation of subject ids at fetch time (wk_tok_set_subject_map):
    // the coord-match wants genome indices, not dictionary ids
    std::vector<int32_t> subj_map;
    bool use_map = false;
    double lap_ms[LAP_N] = {0, 0, 0, 0};
    int64_t lap_calls = 0, lap_bytes = 0;
    ~wk_tok() {
        if (TokLap::enabled() && lap_calls)
            fprintf(stderr, "[wk_tok] %d threads, %lld calls, %.1f MB: cuts %.1f ms, tokenize %.1f ms, merge %.1f ms, fetch %.1f ms\n",
                    n_threads, (long long)lap_calls, lap_bytes / 1e6, lap_ms[LAP_CUTS], lap_ms[LAP_TOKENIZE], lap_ms[LAP_MERGE],
                    lap_ms[LAP_FETCH]);
        delete pool;
    }
    int32_t reported = 0;  // subjects already handed to the caller
    bool in_header = false; // still inside the leading '@' lines of a file
    // state of parse_sam_file_ex_ft's variables after the text seen so far (see Local)
    bool tail_keep = true;
    std::string tail_this, tail_lines;
    // stratification of the current sample: read id -> stratum (file.read_map_uniq
    // + workflow.read_strata, file.py:368-385, workflow.py:912-938)
    // (sharded by the top hash bits so that a map of tens of millions of reads is
    // built by all threads: each shard is owned by one thread while loading)
    static constexpr int kStrataShards = 64;
    // (two of them: the table of the next sample can be built — wk_tok_strata_select
    // — while the current one is in use, then swapped in)
    struct StrataSet {
        StrataShard shard[kStrataShards];
        NameTable labels;
    };
    StrataSet strata_sets[2];
    int strata_cur = 0, strata_target = 0;  // the set lookups use; the set clear / load / labels work on
    const StrataShard* strata_shards() const { return strata_sets[strata_cur].shard; }
    int32_t strata_find(const char* p, size_t n) const {
        const uint64_t hv = hash_bytes(p, n);
        return strata_shards()[hv >> 58].find(p, n, hv);
    }
    // demultiplexing (workflow.demultiplex, workflow.py:844-909): sample = text
    // before the first '_' of the read id, if anything follows it
    NameTable samples;
    int32_t samples_reported = 0;
};

namespace {

// the vector scanners of SAM lines (wk_tok_scan.inc), one per instruction set
#define WK_SCAN_W 16
#define WK_SCAN_NAME produce_sam_sse2
#define WK_SCAN_ATTR
#include "wk_tok_scan.inc"
#undef WK_SCAN_W
#undef WK_SCAN_NAME
#undef WK_SCAN_ATTR
#define WK_SCAN_W 32
#define WK_SCAN_NAME produce_sam_avx2
#define WK_SCAN_ATTR __attribute__((target("avx2")))
#include "wk_tok_scan.inc"
#undef WK_SCAN_W
#undef WK_SCAN_NAME
#undef WK_SCAN_ATTR

// the column trim of SAM text (wk_trim.inc), one per instruction set
#define WK_TRIM_W 16
#define WK_TRIM_NAME trim_sam_sse2
#define WK_TRIM_ATTR
#include "wk_trim.inc"
#undef WK_TRIM_W
#undef WK_TRIM_NAME
#undef WK_TRIM_ATTR
#define WK_TRIM_W 32
#define WK_TRIM_NAME trim_sam_avx2
#define WK_TRIM_ATTR __attribute__((target("avx2")))
#include "wk_trim.inc"
#undef WK_TRIM_W
#undef WK_TRIM_NAME
#undef WK_TRIM_ATTR

// Lines of any format through the memchr parsers (parse_row): what the vector
// scanner leaves over at the end of a range, and every format but SAM.
int produce_rows(int fmt, const char*& pos, const char* e, bool extra, const FastDict& dict, Line* out, int max) {
    const char* p = pos;
    int n = 0;
    while (n < max && p < e) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
        const char* le = nl ? nl : e;
        Line L = parse_row(fmt, p, le, extra);
        L.line = p;
        L.le = le;
        p = nl ? nl + 1 : e;
        if (L.ok) {
            if (fmt == WK_FMT_SAM && is_unmapped(L)) continue;
            L.rh = hash_bytes(L.r, L.rn);
            dict.prefetch(L.rh);
        }
        out[n++] = L;
    }
    pos = p;
    return n;
}

inline bool same_name(const char* a, size_t an, const char* b, size_t bn) {
    return an == bn && memcmp(a, b, an) == 0;
}

// The strata of a range's reads, looked up together: the table of a sample does
// not fit any cache, so the lookups run as a pipeline — hash and ask for the
// slot's line, a few keys later for the entry's, then for the key's, and only
// then compare — instead of one miss after the other per read.
static void resolve_groups(const wk_tok* T, Local& out) {
    const size_t n = out.group.size();
    if (n == 0) return;
    constexpr size_t D = 8, R = 32;  // keys between two steps; ring of hashes
    uint64_t hv[R];
    out.gkey_off.push_back((uint32_t)out.gkeys.size());
    const char* keys = out.gkeys.data();
    const StrataShard* const shards = T->strata_shards();
    for (size_t i = 0; i < n + 3 * D; ++i) {
        if (i < n) {
            const uint64_t h = hash_bytes(keys + out.gkey_off[i], out.gkey_off[i + 1] - out.gkey_off[i]);
            hv[i % R] = h;
            shards[h >> 58].prefetch_slot(h);
        }
        if (i >= D && i - D < n) shards[hv[(i - D) % R] >> 58].prefetch_entry(hv[(i - D) % R]);
        if (i >= 2 * D && i - 2 * D < n) shards[hv[(i - 2 * D) % R] >> 58].prefetch_key(hv[(i - 2 * D) % R]);
        if (i >= 3 * D) {
            const size_t j = i - 3 * D;
            const uint64_t h = hv[j % R];
            out.group[j] = shards[h >> 58].find(keys + out.gkey_off[j], out.gkey_off[j + 1] - out.gkey_off[j], h);
        }
    }
    out.gkey_off.pop_back();
}

template <bool kExtra>
void tokenize_range(const wk_tok* T, int fmt, const char* base, const char* b, const char* e, int extra_bits, bool want_names,
                    bool want_groups, bool want_samples, Local& out) {
    const bool keep_empty = (extra_bits & 2) != 0;
    const bool filt = T->exclude.size() > 0;
    const bool track_pool = kExtra && filt && fmt == WK_FMT_SAM;
    static const char* const kSuffix[3] = {"", "/1", "/2"};
    // room for the range's records up front (a trimmed SAM line is ~40 bytes)
    {
        const size_t guess = (size_t)(e - b) / 36 + 1024;
        if (kExtra)
            out.rec.reserve(guess);
        else
            out.subj.reserve(guess);
        out.rend.reserve(guess / 2 + 1024);
    }
    // current run state
    const char* cur = nullptr;
    size_t cur_n = 0;
    bool keep = true;
    std::vector<Record> pool_x[3];  // "ex" flavour
    std::vector<int32_t> pool_s[3]; // plain flavour: subject ids
    auto emit_read = [&](int m) {
        if (want_names) out.qname.push_back(((uint64_t)(cur - base) << 24) | ((uint64_t)cur_n << 2) | (uint64_t)m);
        if (want_samples) {
            // query.partition('_'): sample = left part if the right part
            // (which includes a mate suffix) is not empty, else ''
            const char* us = (const char*)memchr(cur, '_', cur_n);
            size_t sn = 0;
            if (us && ((size_t)(us - cur) + 1 < cur_n || m != 0)) sn = (size_t)(us - cur);
            const uint64_t hv = hash_bytes(cur, sn);
            int32_t id = T->samples.find(cur, sn, hv);
            if (id < 0) {
                int32_t f = out.fresh_samples.find(cur, sn, hv);
                if (f < 0) f = out.fresh_samples.add(cur, sn, hv);
                id = -(1 + f);
            }
            out.sample.push_back(id);
        }
        if (want_groups) {  // stratum of read id = QNAME + mate suffix
            out.gkey_off.push_back((uint32_t)out.gkeys.size());
            out.gkeys.append(cur, cur_n);
            out.gkeys.append(kSuffix[m]);
            out.group.push_back(-1);  // (resolve_groups, at the end of the range)
        }
    };
    auto flush = [&]() {
        if (!cur || !keep) {
            for (int m = 0; m < 3; ++m) {
                pool_x[m].clear();
                pool_s[m].clear();
            }
            return;
        }
        for (int m = 0; m < 3; ++m) {
            if (kExtra) {
                auto& v = pool_x[m];
                if (v.empty()) continue;
                out.rec.insert(out.rec.end(), v.begin(), v.end());
                out.rend.push_back((int32_t)out.rec.size());
                out.n_big += v.size() > (size_t)WK_WEIGHT_MAX_K;
                emit_read(m);
                v.clear();
            } else {
                auto& v = pool_s[m];
                if (v.empty()) continue;
                size_t w = v.size();
                if (w > 1) {
                    // the plain parsers collect subject *sets* (align.py:258-330):
                    // duplicates go here, so the device never has to look for them
                    if (w <= 64) {
                        w = 1;
                        for (size_t i = 1; i < v.size(); ++i) {
                            bool dup = false;
                            for (size_t j = 0; j < w && !dup; ++j) dup = v[j] == v[i];
                            if (!dup) v[w++] = v[i];
                        }
                    } else {
                        std::sort(v.begin(), v.end());
                        w = (size_t)(std::unique(v.begin(), v.end()) - v.begin());
                    }
                }
                out.subj.insert(out.subj.end(), v.begin(), v.begin() + (ptrdiff_t)w);
                out.rend.push_back((int32_t)out.subj.size());
                out.n_big += w > (size_t)WK_WEIGHT_MAX_K;
                emit_read(m);
                v.clear();
            }
        }
    };
    constexpr int kBatch = 16;
    Line batch[kBatch];
    const char* p = b;
    bool vector_scan = fmt == WK_FMT_SAM && T->produce_sam != nullptr;
    while (p < e) {
        int n = 0;
        if (vector_scan) {
            n = T->produce_sam(p, e, kExtra, T->dict, batch, kBatch);
            if (n == 0 && p < e && (size_t)(e - p) < kScanSlack + 64) vector_scan = false;  // the last lines of the range
            if (n == 0 && vector_scan) {
                // (a line longer than the scanner could finish near the end: the memchr parser takes over)
                vector_scan = false;
            }
        }
        if (!vector_scan && n == 0) n = produce_rows(fmt, p, e, kExtra, T->dict, batch, kBatch);
        for (int li = 0; li < n; ++li) {
            const Line& L = batch[li];
            const char* line = L.line;
            const char* le = L.le;
            if (!L.ok) {
                // (an empty line inside a SAM body fails `line.split('\t', 3)`
                // like any other short line, align.py:313; the other formats skip it)
                if (le == line && fmt != WK_FMT_SAM) continue;
                if (fmt != WK_FMT_SAM && !L.bad_number) continue;  // not a row of this format
                out.error = 2;
                out.error_at = (size_t)(line - base);
                return;
            }
            bool run_start = false;
            if (!(cur && same_name(L.q, L.qn, cur, cur_n))) {
                flush();
                cur = L.q;
                cur_n = L.qn;
                keep = true;
                run_start = true;
            } else if (!keep) {
                continue;
            }
            const uint64_t hv = L.rh;
            if (filt && T->exclude.find(L.r, L.rn, hv) >= 0) {
                keep = false;
                continue;
            }
            if (fmt == WK_FMT_SAM && L.bad_number) {
                // int(flag) of a line that is kept (the filtering parsers never convert
                // the FLAG of a line they drop, align.py:438-470, 511-540)
                out.error = 2;
                out.error_at = (size_t)(line - base);
                return;
            }
            if (track_pool) {
                if (run_start) {  // `pool = ([], [], [])` (align.py:526)
                    out.pool_lines.clear();
                    out.have_pool = true;
                }
                out.pool_lines.emplace_back(line, le);
            }
            const int mate = fmt == WK_FMT_SAM ? (L.flag >> 6) & 3 : 0;
            // (both mate bits index a 3-tuple with 3, align.py:339 -- in the "ex" parser
            // after int(pos) and cigar_to_lens have had their say, align.py:382-391)
            if (mate == 3 && !(kExtra && fmt == WK_FMT_SAM)) {
                out.error = 1;
                out.error_at = (size_t)(line - base);
                return;
            }
            int32_t id = T->dict.find(L.r, L.rn, hv);
            if (id < 0) {
                int32_t f = out.fresh.find(L.r, L.rn, hv);
                if (f < 0) f = out.fresh.add(L.r, L.rn, hv);
                id = -(1 + f);
            }
            if (!kExtra) {
                pool_s[mate].push_back(id);
                continue;
            }
            Record rc{};
            rc.subj = id;
            if (fmt == WK_FMT_SAM) {
                // int(pos) and cigar_to_lens raise on text that is not a number
                // (align.py:382-385); a negative POS is a number
                long pos = 0;
                int64_t aligned = 0, span = 0;
                if (!parse_int(L.pos, L.pos_end, pos) || !cigar_lens(L.cigar, L.cn, aligned, span)) {
                    out.error = 2;
                    out.error_at = (size_t)(line - base);
                    return;
                }
                if (mate == 3) {
                    out.error = 1;
                    out.error_at = (size_t)(line - base);
                    return;
                }
                // numbers, but wider than the 32 bits coordinates are held in
                // (a stated limit, DESIGN: a loud error, never a wrapped value)
                if (!fits_i32(pos - 1) || aligned < -(int64_t)UINT32_MAX || aligned > (int64_t)UINT32_MAX || span < INT32_MIN ||
                    span > INT32_MAX || !fits_i32(pos - 1 + span)) {
                    out.error = 3;
                    out.error_at = (size_t)(line - base);
                    return;
                }
                if (aligned == 0 && !keep_empty) continue;  // ordinal.py:231 (range.py keeps them)
                rc.beg = (int32_t)(pos - 1);
                rc.end = (int32_t)(pos - 1 + span);
                rc.len = (uint32_t)aligned;
            } else {
                if (L.len == 0 && !keep_empty) continue;
                rc.beg = L.beg;
                rc.end = L.end;
                rc.len = L.len;
            }
            pool_x[mate].push_back(rc);
        }
    }
    if (track_pool && cur) {
        out.any_run = true;
        out.fin_keep = keep;
        out.fin_q = cur;
        out.fin_qn = cur_n;
    }
    flush();
    if (want_groups) resolve_groups(T, out);
}

// first mapped line at or after p whose QNAME differs from the previous mapped line's
const char* run_boundary(int fmt, bool extra, const char* base, const char* p, const char* e) {
    // previous mapped line before p
    const char* prev_q = nullptr;
    size_t prev_n = 0;
    const char* s = p;
    while (s > base) {
        const char* ls = s - 1;  // points at '\n' ending the previous line
        const char* q = ls;
        while (q > base && q[-1] != '\n') --q;
        Line L = parse_row(fmt, q, ls, extra);
        s = q;
        if (is_row(fmt, L)) {
            prev_q = L.q;
            prev_n = L.qn;
            break;
        }
    }
    if (!prev_q) return p;
    while (p < e) {
        const char* nl = (const char*)memchr(p, '\n', e - p);
        const char* le = nl ? nl : e;
        Line L = parse_row(fmt, p, le, extra);
        if (is_row(fmt, L)) {
            if (!(L.qn == prev_n && memcmp(L.q, prev_q, prev_n) == 0)) return p;
        }
        p = nl ? nl + 1 : e;
    }
    return e;
}

// The part of a block that can be tokenised now: behind the leading '@' lines
// of a SAM file (`in_header`: still inside them; updated) up to — unless the
// block is final — the start of the last run of equal query ids (it may
// continue in the next block).  false: nothing complete yet (b = what was
// consumed: header lines).
bool block_span(int fmt, bool ex, const char* buf, int64_t len, bool final_block, bool& in_header, const char*& b, const char*& stop) {
    b = buf;
    const char* e = buf + len;
    while (in_header && b < e) {
        if (*b != '@') {
            in_header = false;
            break;
        }
        const char* nl = (const char*)memchr(b, '\n', e - b);
        if (!nl && !final_block) break;  // partial header line: wait for more text
        b = nl ? nl + 1 : e;
    }
    // only whole lines; unless final, stop before the last run (it may continue)
    stop = e;
    if (final_block) return true;
    const char* last_nl = nullptr;
    for (const char* p = e; p > b; --p)
        if (p[-1] == '\n') {
            last_nl = p;
            break;
        }
    if (!last_nl || in_header) return false;
    stop = last_nl;
    // start of the last run: walk back over lines while the QNAME stays the same
    const char* run_start = nullptr;
    const char* q_last = nullptr;
    size_t qn_last = 0;
    const char* s = stop;
    while (s > b) {
        const char* ls = s - 1;
        const char* q = ls;
        while (q > b && q[-1] != '\n') --q;
        Line L = parse_row(fmt, q, ls, ex);
        if (is_row(fmt, L)) {
            if (!q_last) {
                q_last = L.q;
                qn_last = L.qn;
                run_start = q;
            } else if (L.qn == qn_last && memcmp(L.q, q_last, qn_last) == 0) {
                run_start = q;
            } else {
                break;
            }
        }
        s = q;
    }
    if (run_start) stop = run_start;
    return true;
}

}  // namespace

extern "C" {

int wk_tok_create(int n_threads, wk_tok** out) {
    if (!out) return WK_E_ARG;
    wk_tok* t = new (std::nothrow) wk_tok();
    if (!t) return WK_E_HIP;
    if (n_threads <= 0) {
        n_threads = (int)std::thread::hardware_concurrency();
        if (n_threads <= 0) n_threads = 1;
    }
    t->n_threads = std::min(n_threads, 256);  // (the default the host picks is lower: classify.tokenizer_threads)
    t->pool = new (std::nothrow) WorkPool(t->n_threads > 1 ? t->n_threads : 0);
    if (!t->pool) {
        delete t;
        return WK_E_HIP;
    }
    t->produce_sam = __builtin_cpu_supports("avx2") ? produce_sam_avx2 : produce_sam_sse2;
    if (getenv("WOLTKA_TOK_SCALAR")) t->produce_sam = nullptr;  // measurement: the memchr parsers only
    *out = t;
    return WK_OK;
}

void wk_tok_destroy(wk_tok* t) { delete t; }

const char* wk_tok_last_error(const wk_tok* t) { return t ? t->err.c_str() : "null tokenizer"; }

int wk_tok_sam_tail(wk_tok* t, char* buf, int64_t cap, int64_t* len) {
    if (!t || !len || cap < 0 || (cap > 0 && !buf)) return WK_E_ARG;
    // parse_sam_file_ex_ft's last three statements yield `pool` under the last
    // query's name without looking at `keep` (align.py:542-547): when the last
    // query of the file was dropped, that is one more yield — of whatever the
    // pool still holds.  The text returned here makes the tokenizer produce
    // exactly those reads: the pool's lines with the last query's name.
    std::string text;
    if (!t->tail_keep) {
        const char* p = t->tail_lines.data();
        const char* e = p + t->tail_lines.size();
        while (p < e) {
            const char* nl = (const char*)memchr(p, '\n', e - p);
            const char* le = nl ? nl : e;
            const char* tab = (const char*)memchr(p, '\t', le - p);
            if (tab) {
                text += t->tail_this;
                text.append(tab, le);
                text.push_back('\n');
            }
            p = nl ? nl + 1 : e;
        }
    }
    *len = (int64_t)text.size();
    if ((int64_t)text.size() > cap) {
        if (cap == 0) return WK_OK;  // size query
        t->err = "buffer too small for the tail text";
        return WK_E_CAPACITY;
    }
    if (!text.empty()) memcpy(buf, text.data(), text.size());
    return WK_OK;
}

int wk_tok_set_exclude(wk_tok* t, const char* blob, const int32_t* off, int32_t n) {
    if (!t || n < 0 || (n > 0 && (!blob || !off))) return WK_E_ARG;
    t->exclude = NameTable();
    for (int32_t i = 0; i < n; ++i) {
        const char* p = blob + off[i];
        const size_t len = (size_t)(off[i + 1] - off[i]);
        const uint64_t hv = hash_bytes(p, len);
        if (t->exclude.find(p, len, hv) < 0) t->exclude.add(p, len, hv);
    }
    return WK_OK;
}

// Offset of the first line at or after `pos` that starts a new run of equal
// query ids (a cut there never splits a read): the unit of byte-range sharding
// of one large file over several processes.  0 stays 0, len stays len.
int wk_tok_boundary(int fmt, int extra, const char* buf, int64_t len, int64_t pos, int64_t* out) {
    if (!buf || len < 0 || !out || fmt < WK_FMT_SAM || fmt > WK_FMT_PAF) return WK_E_ARG;
    if (pos <= 0) {
        *out = 0;
        return WK_OK;
    }
    if (pos >= len) {
        *out = len;
        return WK_OK;
    }
    if (fmt == WK_FMT_MAP) extra = 0;
    const char* e = buf + len;
    const char* p = next_line(buf + pos - 1, e);  // first line start at or after pos
    *out = run_boundary(fmt, (extra & 1) != 0, buf, p, e) - buf;
    return WK_OK;
}

int wk_tok_sam(wk_tok* t, const char* buf, int64_t len, int first_block, int final_block, int extra, int want_names,
               int64_t* consumed, int64_t* n_reads, int64_t* n_records) {
    return wk_tok_text(t, WK_FMT_SAM, buf, len, first_block, final_block, extra, want_names, consumed, n_reads, n_records);
}

int wk_tok_text(wk_tok* t, int fmt, const char* buf, int64_t len, int first_block, int final_block, int extra,
                int want_names, int64_t* consumed, int64_t* n_reads, int64_t* n_records) {
    if (!t || !buf || len < 0 || !consumed || !n_reads || !n_records) return WK_E_ARG;
    if (fmt < WK_FMT_SAM || fmt > WK_FMT_PAF) {
        t->err = "unknown alignment format code";
        return WK_E_ARG;
    }
    if (fmt == WK_FMT_MAP) extra = 0;  // no "ex" flavour (align.py:236)
    const bool ex = (extra & 1) != 0;
    // header: leading '@' lines (align.py:295-300); it may span several blocks
    if (first_block) {
        t->in_header = fmt == WK_FMT_SAM;
        t->tail_keep = true;
        t->tail_this.clear();
        t->tail_lines.clear();
    }
    const char* b = buf;
    const char* stop = buf + len;
    if (!block_span(fmt, ex, buf, len, final_block != 0, t->in_header, b, stop)) {
        *consumed = b - buf;
        *n_reads = *n_records = 0;
        t->n_loc = 0;
        t->tot_reads = t->tot_rec = t->tot_big = 0;
        t->last_extra = ex;
        t->last_want = want_names;
        return WK_OK;
    }
    TokLap lap(t->lap_ms);
    t->lap_calls += 1;
    t->lap_bytes += stop - b;
    const int64_t span = stop - b;
    int T = t->n_threads;
    if (span < (int64_t)T * (1 << 16)) T = (int)std::max<int64_t>(1, span >> 16);
    std::vector<const char*> cut(T + 1);
    cut[0] = b;
    cut[T] = stop;
    for (int i = 1; i < T; ++i) {
        const char* p = b + span * i / T;
        if (p < cut[i - 1]) p = cut[i - 1];
        p = (p > b) ? next_line(p - 1, stop) : b;  // to a line start
        cut[i] = run_boundary(fmt, ex, b, p, stop);
    }
    for (int i = 1; i <= T; ++i)
        if (cut[i] < cut[i - 1]) cut[i] = cut[i - 1];
    if ((int)t->loc.size() < T) t->loc.resize((size_t)T);
    std::vector<Local>& loc = t->loc;
    t->n_loc = T;
    t->last_extra = ex;
    t->last_want = want_names;
    const std::function<void(int)> work = [&](int i) {
        loc[i].reset();
        if (ex)
            tokenize_range<true>(t, fmt, buf, cut[i], cut[i + 1], extra, (want_names & 1) != 0, (want_names & 2) != 0,
                                 (want_names & 4) != 0, loc[i]);
        else
            tokenize_range<false>(t, fmt, buf, cut[i], cut[i + 1], extra, (want_names & 1) != 0, (want_names & 2) != 0,
                                  (want_names & 4) != 0, loc[i]);
    };
    lap(LAP_CUTS);
    t->pool->run(T, work);
    lap(LAP_TOKENIZE);
    for (int i = 0; i < T; ++i)
        if (loc[i].error) {
            // (the line is named by its query: where it lies in the block says nothing to the user, and differs
            // with the way the blocks are cut and trimmed)
            const char* ln = buf + loc[i].error_at;
            const char* le = buf + len;
            int qn = 0;
            while (ln + qn < le && qn < 80 && ln[qn] != '\t' && ln[qn] != '\n') ++qn;
            char msg[240];
            snprintf(msg, sizeof msg,
                     loc[i].error == 1   ? "SAM flag with both mate bits set in a line of query '%.*s'"
                     : loc[i].error == 3 ? "coordinate wider than 32 bits in a line of query '%.*s'"
                                         : "malformed alignment line of query '%.*s'",
                     qn, ln);
            t->err = msg;
            t->n_loc = 0;
            t->tot_reads = t->tot_rec = t->tot_big = 0;
            return loc[i].error == 1 ? WK_E_RANGE : WK_E_ARG;
        }
    for (int i = 0; i < T; ++i) {  // (ranges in text order)
        if (!loc[i].any_run) continue;
        t->tail_keep = loc[i].fin_keep;
        t->tail_this.assign(loc[i].fin_q, loc[i].fin_qn);
        if (loc[i].have_pool) {
            t->tail_lines.clear();
            for (const auto& ln : loc[i].pool_lines) {
                t->tail_lines.append(ln.first, ln.second);
                t->tail_lines.push_back('\n');
            }
        }
    }
    // merge fresh names in thread order (= order of first appearance in the text)
    // first appearance order must not depend on the thread count: names that were
    // fresh in several ranges were added by the earliest range, which is also
    // where they first appear in the text.  Within one range `fresh` ids follow
    // the text order.  (A name fresh in range i is absent from all earlier ranges
    // only if those did not see it at all.)
    if ((int)t->remap.size() < T) {
        t->remap.resize((size_t)T);
        t->sremap.resize((size_t)T);
    }
    for (int i = 0; i < T; ++i) {
        const NameTable& f = loc[i].fresh;
        std::vector<int32_t>& rm = t->remap[i];
        rm.resize((size_t)f.size());
        for (int32_t k = 0; k < f.size(); ++k) {
            const char* p = f.arena.data() + f.off[k];
            int32_t id = t->names.find(p, f.len[k], f.hash[k]);
            if (id < 0) {
                id = t->names.add(p, f.len[k], f.hash[k]);
                t->dict.insert(p, f.len[k], f.hash[k], id);
            }
            rm[k] = id;
        }
    }
    for (int i = 0; i < T; ++i) {
        const NameTable& f = loc[i].fresh_samples;
        std::vector<int32_t>& rm = t->sremap[i];
        rm.resize((size_t)f.size());
        for (int32_t k = 0; k < f.size(); ++k) {
            const char* p = f.arena.data() + f.off[k];
            int32_t id = t->samples.find(p, f.len[k], f.hash[k]);
            if (id < 0) id = t->samples.add(p, f.len[k], f.hash[k]);
            rm[k] = id;
        }
    }
    int64_t tot_reads = 0, tot_rec = 0, tot_big = 0;
    t->rbase.assign((size_t)T + 1, 0);
    t->qbase.assign((size_t)T + 1, 0);
    for (int i = 0; i < T; ++i) {
        t->rbase[i + 1] = t->rbase[i] + (int64_t)loc[i].n_records(ex);
        t->qbase[i + 1] = t->qbase[i] + (int64_t)loc[i].rend.size();
        tot_big += loc[i].n_big;
    }
    tot_reads = t->qbase[T];
    tot_rec = t->rbase[T];
    if (tot_rec >= (1ll << 31)) {
        t->err = "more than 2^31 records in one block; pass smaller blocks";
        t->n_loc = 0;
        return WK_E_RANGE;
    }
    t->tot_reads = tot_reads;
    t->tot_rec = tot_rec;
    t->tot_big = tot_big;
    lap(LAP_MERGE);
    *consumed = stop - buf;
    *n_reads = tot_reads;
    *n_records = tot_rec;
    return WK_OK;
}

// The results of the last wk_tok_text, scattered from the thread ranges' own
// buffers straight into the caller's arrays (pinned staging buffers, for
// instance) by all threads.
static int fetch_results(wk_tok* t, int32_t* subj, uint32_t* packed, int32_t* off, int32_t* beg, int32_t* end, uint32_t* len,
                         uint64_t* qname, int32_t* group, int32_t* sample) {
    const int T = t->n_loc;
    if (off) off[0] = 0;
    if (T == 0) return WK_OK;
    const bool ex = t->last_extra;
    std::vector<Local>& loc = t->loc;
    const std::function<void(int)> work = [&](int i) {
        const Local& L = loc[i];
        const int64_t rb = t->rbase[i], qb = t->qbase[i];
        const std::vector<int32_t>& rm = t->remap[i];
        const int32_t* map = t->use_map ? t->subj_map.data() : nullptr;
        const int32_t map_n = (int32_t)t->subj_map.size();
        if (ex) {
            for (size_t k = 0; k < L.rec.size(); ++k) {
                const Record& rc = L.rec[k];
                if (subj) {
                    const int32_t id = rc.subj >= 0 ? rc.subj : rm[(size_t)(-(rc.subj + 1))];
                    subj[rb + (int64_t)k] = !map ? id : (id < map_n ? map[id] : -1);
                }
                if (beg) beg[rb + (int64_t)k] = rc.beg;
                if (end) end[rb + (int64_t)k] = rc.end;
                if (len) len[rb + (int64_t)k] = rc.len;
            }
        } else if (subj) {
            for (size_t k = 0; k < L.subj.size(); ++k) {
                const int32_t v = L.subj[k];
                const int32_t id = v >= 0 ? v : rm[(size_t)(-(v + 1))];
                subj[rb + (int64_t)k] = !map ? id : (id < map_n ? map[id] : -1);
            }
        }
        if (packed && !ex) {
            // subject index | position in the read << 23 | size of the read << 27
            // (both 0 for a read of more than WK_WEIGHT_MAX_K records: the
            // weighted histogram leaves it out)
            size_t lo = 0;
            for (size_t r = 0; r < L.rend.size(); ++r) {
                const size_t hi = (size_t)L.rend[r];
                const uint32_t n = (uint32_t)(hi - lo);
                const bool small = n <= (uint32_t)WK_WEIGHT_MAX_K;
                const uint32_t tag = small ? n << 27 : 0u;
                for (size_t k = lo; k < hi; ++k) {
                    const int32_t v = L.subj[k];
                    packed[rb + (int64_t)k] =
                        (uint32_t)(v >= 0 ? v : rm[(size_t)(-(v + 1))]) | tag | (small ? (uint32_t)(k - lo) << 23 : 0u);
                }
                lo = hi;
            }
        }
        if (off)
            for (size_t k = 0; k < L.rend.size(); ++k) off[qb + (int64_t)k + 1] = (int32_t)(rb + L.rend[k]);
        if (qname && (t->last_want & 1)) memcpy(qname + qb, L.qname.data(), L.qname.size() * 8);
        if (group && (t->last_want & 2)) memcpy(group + qb, L.group.data(), L.group.size() * 4);
        if (sample && (t->last_want & 4)) {
            const std::vector<int32_t>& sm = t->sremap[i];
            for (size_t k = 0; k < L.sample.size(); ++k)
                sample[qb + (int64_t)k] = L.sample[k] >= 0 ? L.sample[k] : sm[(size_t)(-(L.sample[k] + 1))];
        }
    };
    TokLap lap(t->lap_ms);
    t->pool->run(T, work);
    lap(LAP_FETCH);
    return WK_OK;
}

// [offset, offset + len) of file `fd` into dst, read by all worker threads
// (pread on slices): a block of a few hundred MB arrives at memory speed instead
// of one thread's copy rate, and — unlike a memory map of the file — leaves no
// page of the input to be mapped and unmapped by the tokenizer threads.
int wk_tok_read(wk_tok* t, int fd, int64_t offset, char* dst, int64_t len, int64_t* got) {
    if (!t || fd < 0 || offset < 0 || len < 0 || (len > 0 && !dst) || !got) return WK_E_ARG;
    *got = 0;
    if (len == 0) return WK_OK;
    const int64_t slice = 4ll << 20;
    const int n = (int)((len + slice - 1) / slice);
    std::vector<int64_t> done((size_t)n, 0);
    std::atomic<int> failed{0};
    const std::function<void(int)> work = [&](int i) {
        const int64_t lo = (int64_t)i * slice, hi = std::min(len, lo + slice);
        int64_t at = lo;
        while (at < hi) {
            const ssize_t r = pread(fd, dst + at, (size_t)(hi - at), (off_t)(offset + at));
            if (r < 0) {
                failed.store(1);
                break;
            }
            if (r == 0) break;  // end of file
            at += r;
        }
        done[(size_t)i] = at - lo;
    };
    t->pool->run(n, work);
    if (failed.load()) {
        t->err = "reading the alignment file failed";
        return WK_E_ARG;
    }
    int64_t total = 0;
    for (int i = 0; i < n; ++i) {
        total += done[(size_t)i];
        if (done[(size_t)i] < std::min(len, (int64_t)(i + 1) * slice) - (int64_t)i * slice) break;  // short slice: the file ends here
    }
    *got = total;
    return WK_OK;
}

// SAM text [begin, begin + want) of a source -- `begin` a line start; the source: memory all threads see (a mapped
// file) or an open file, read by the threads piece by piece (pread into a buffer of a piece's size that stays in the
// thread's cache: no page of the file is mapped, and the bytes cross the memory bus once) -- trimmed to its first
// `keep` columns (wk_trim.inc) into dst by all tokenizer threads: pieces of 1 MB, each trimmed into its thread's
// scratch (a tenth of the piece, as a rule) and copied to its place behind the piece before it, whose length it waits
// for -- not for its copy.  Whole lines only: *consumed = bytes of input taken (up to the start of the first line
// without a newline in the range, or -- the source ends with the range -- all of it), *got = bytes written.  A piece
// that would pass `cap` ends the call early the same way.
static int trim_pieces(wk_tok* t, const char* mem, int fd, int64_t src_len, int64_t begin, int64_t want, int keep_tabs, char* dst, int64_t cap,
                       int64_t* consumed, int64_t* got) {
    *consumed = *got = 0;
    if (want == 0) return WK_OK;
    typedef size_t (*trim_fn)(const char*, const char*, const char*, int, bool, char*, const char*, const char**, bool*);
    static const trim_fn trim = __builtin_cpu_supports("avx2") ? trim_sam_avx2 : trim_sam_sse2;
    const int64_t piece = 1ll << 20;
    const int n = (int)((want + piece - 1) / piece);
    const int64_t limit = begin + want;   // (absolute positions from here on)
    const bool at_eof = limit == src_len;
    // off[j]: where piece j's bytes go (-1: not known yet, -2: no room for a piece before it)
    std::vector<std::atomic<int64_t>> off((size_t)n + 1);
    for (auto& x : off) x.store(-1, std::memory_order_relaxed);
    off[0].store(0, std::memory_order_relaxed);
    std::vector<int64_t> stop((size_t)n, 0);   // where piece j ended: the start of the first line it did not take
    std::vector<char> done((size_t)n, 0);      // piece j took every line that starts in it
    std::atomic<int> failed{0};
    const std::function<void(int)> work = [&](int j) {
        static thread_local std::vector<char> scratch, inbuf;
        const int64_t pb = begin + (int64_t)j * piece, pe = std::min(limit, pb + piece);
        // the bytes [lo, hi) at hand: all of a mapped source; of a file what was read (the piece, the byte in front
        // of it and some more for the line that runs behind it; read again, larger, when that line is longer)
        int64_t lo = mem ? begin : (j > 0 ? pb - 1 : pb), hi = mem ? limit : std::min(limit, pe + (64 << 10));
        const char* base = mem ? mem + lo : nullptr;
        auto read_window = [&]() -> bool {
            if (mem) return true;
            inbuf.resize((size_t)(hi - lo) + 64);
            int64_t at = 0;
            while (at < hi - lo) {
                const ssize_t r = pread(fd, inbuf.data() + at, (size_t)(hi - lo - at), (off_t)(lo + at));
                if (r <= 0) return false;
                at += r;
            }
            base = inbuf.data();
            return true;
        };
        if (!read_window()) {
            failed.store(1);
            hi = lo;
        }
        // the first line that starts in the piece
        int64_t s = pb;
        if (j > 0 && hi > lo) {
            for (;;) {
                const char* nl = (const char*)memchr(base + (pb - 1 - lo), '\n', (size_t)(hi - (pb - 1)));
                if (nl) {
                    s = lo + (nl - base) + 1;
                    break;
                }
                if (hi >= limit) {
                    s = limit;
                    break;
                }
                hi = std::min(limit, hi + (hi - lo));   // (a line longer than the window: look further)
                if (!read_window()) {
                    failed.store(1);
                    s = limit;
                    break;
                }
            }
        }
        size_t len = 0;
        int64_t next = s;
        if (s < pe && hi > lo) {
            if (scratch.size() < (size_t)piece + 4096) scratch.resize((size_t)piece + 4096);
            for (;;) {
                bool full = false;
                const char* nx = nullptr;
                // lines that start in [next, pe): each to its newline, wherever in front of `hi` that is
                len += trim(base + (next - lo), base + (pe - lo), base + (hi - lo), keep_tabs, at_eof && hi == limit, scratch.data() + len,
                            scratch.data() + scratch.size(), &nx, &full);
                next = lo + (nx - base);
                if (full) {  // (a line that leaves whole and is longer than the scratch: room for it)
                    scratch.resize(scratch.size() * 2);
                    continue;
                }
                if (next >= pe || hi >= limit) break;
                hi = std::min(limit, hi + std::max<int64_t>(hi - lo, 1 << 20));   // (the line runs behind the window)
                if (!read_window()) {
                    failed.store(1);
                    break;
                }
            }
        }
        stop[(size_t)j] = next;
        done[(size_t)j] = (next >= pe || (at_eof && next >= limit)) ? 1 : 0;
        // my place: behind the piece before me
        int64_t at;
        while ((at = off[(size_t)j].load(std::memory_order_acquire)) == -1) std::this_thread::yield();
        if (at == -2 || at + (int64_t)len > cap) {
            off[(size_t)j + 1].store(-2, std::memory_order_release);
            return;
        }
        off[(size_t)j + 1].store(at + (int64_t)len, std::memory_order_release);
        if (len) memcpy(dst + at, scratch.data(), len);
    };
    t->pool->run(n, work);
    if (failed.load()) {
        t->err = "reading the alignment file failed";
        return WK_E_ARG;
    }
    // what was taken: pieces in order, up to the first that did not take all its lines (a line without newline in the
    // range) or found no room
    int64_t out_len = 0, upto = begin;
    for (int j = 0; j < n; ++j) {
        const int64_t nxt = off[(size_t)j + 1].load(std::memory_order_relaxed);
        if (nxt == -2) break;  // no room for piece j (or one before it)
        out_len = nxt;
        upto = stop[(size_t)j];
        if (!done[(size_t)j]) break;
    }
    *consumed = upto - begin;
    *got = out_len;
    return WK_OK;
}

int wk_tok_trim(wk_tok* t, const char* src, int fd, int64_t src_len, int64_t begin, int64_t want, int keep_tabs, char* dst, int64_t cap,
                int64_t* consumed, int64_t* got) {
    if (!t || (!src && fd < 0) || src_len < 0 || begin < 0 || want < 0 || begin + want > src_len || keep_tabs < 1 || keep_tabs > 32 || !dst ||
        cap < 0 || !consumed || !got)
        return WK_E_ARG;
    return trim_pieces(t, src, fd, src_len, begin, want, keep_tabs, dst, cap, consumed, got);
}

int wk_tok_set_subject_map(wk_tok* t, const int32_t* map, int32_t n) {
    if (!t || n < 0 || (n > 0 && !map)) return WK_E_ARG;
    t->use_map = map != nullptr;
    t->subj_map.assign(map, map + n);
    return WK_OK;
}

int wk_tok_fetch(wk_tok* t, int32_t* subj, int32_t* off, int32_t* beg, int32_t* end, uint32_t* len, uint64_t* qname) {
    if (!t) return WK_E_ARG;
    return fetch_results(t, subj, nullptr, off, beg, end, len, qname, nullptr, nullptr);
}

int wk_tok_fetch_packed(wk_tok* t, uint32_t* packed, int32_t* off, uint64_t* qname, int64_t* n_big) {
    if (!t) return WK_E_ARG;
    if (t->last_extra && t->n_loc) {
        t->err = "packed records exist for the plain flavour only";
        return WK_E_STATE;
    }
    if (t->names.size() > (1 << 23)) {
        t->err = "more than 2^23 subjects: the packed record has 23 bits for the subject index";
        return WK_E_RANGE;
    }
    if (n_big) *n_big = t->n_loc ? t->tot_big : 0;
    return fetch_results(t, nullptr, packed, off, nullptr, nullptr, nullptr, qname, nullptr, nullptr);
}

int wk_tok_fetch_groups(wk_tok* t, int32_t* group) {
    if (!t) return WK_E_ARG;
    return fetch_results(t, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, group, nullptr);
}

int wk_tok_fetch_samples(wk_tok* t, int32_t* sample) {
    if (!t) return WK_E_ARG;
    return fetch_results(t, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, sample);
}

// Names of the samples first seen since the last call: sizes with blob == NULL
int wk_tok_new_samples(wk_tok* t, char* blob, int64_t* off, int32_t* n_new) {
    if (!t || !n_new) return WK_E_ARG;
    *n_new = t->samples.size() - t->samples_reported;
    if (!off) return WK_OK;
    int64_t w = 0;
    int32_t k = 0;
    off[0] = 0;
    for (int32_t i = t->samples_reported; i < t->samples.size(); ++i, ++k) {
        if (blob) memcpy(blob + w, t->samples.arena.data() + t->samples.off[i], t->samples.len[i]);
        w += t->samples.len[i];
        off[k + 1] = w;
    }
    if (blob || *n_new == 0) t->samples_reported = t->samples.size();
    return WK_OK;
}

int wk_tok_strata_clear(wk_tok* t) {
    if (!t) return WK_E_ARG;
    for (int i = 0; i < wk_tok::kStrataShards; ++i) {
        t->strata_sets[t->strata_target].shard[i] = StrataShard();
    }
    t->strata_sets[t->strata_target].labels = NameTable();
    return WK_OK;
}

// Lines "read id <tab> label" with exactly two columns; the label loses its
// trailing white space; a repeated read id keeps its last label (dict()).
int wk_tok_strata_load(wk_tok* t, const char* buf, int64_t len, int64_t* n_entries, int32_t* n_labels) {
    if (!t || (len > 0 && !buf) || len < 0) return WK_E_ARG;
    constexpr int S = wk_tok::kStrataShards;
    struct Entry {
        const char* key;
        uint32_t kn;
        int32_t label;  // id in the parsing thread's local label table
        uint64_t kh;
    };
    // phase 1: line-aligned ranges parsed in parallel into per-shard buckets
    int T = (int)std::max<int64_t>(1, std::min<int64_t>(t->n_threads, len >> 16));
    std::vector<const char*> cut(T + 1);
    const char* e = buf + len;
    cut[0] = buf;
    cut[T] = e;
    for (int i = 1; i < T; ++i) {
        const char* p = buf + len * i / T;
        if (p < cut[i - 1]) p = cut[i - 1];
        cut[i] = (p > buf) ? next_line(p - 1, e) : buf;
    }
    std::vector<std::vector<Entry>> bucket((size_t)T * S);
    std::vector<NameTable> labels(T);
    auto parse = [&](int ti) {
        const char* p = cut[ti];
        const char* stop = cut[ti + 1];
        NameTable& lt = labels[ti];
        std::vector<Entry>* mine = &bucket[(size_t)ti * S];
        while (p < stop) {
            const char* nl = (const char*)memchr(p, '\n', stop - p);
            const char* le = nl ? nl + 1 : stop;  // line including its newline
            // (the reference reads text with universal newlines: a \r without a
            // \n behind it ends a line as well)
            if (const char* cr = (const char*)memchr(p, '\r', le - p))
                if (cr + 1 < le && cr[1] != '\n') le = cr + 1;
            const char* tab = (const char*)memchr(p, '\t', le - p);
            if (tab && !memchr(tab + 1, '\t', le - tab - 1)) {
                const char* vb = tab + 1;
                const char* ve = wkh::py_rstrip(vb, le);  // value.rstrip() (file.py:384)
                const size_t kn = (size_t)(tab - p);
                const uint64_t kh = hash_bytes(p, kn);
                const uint64_t lh = hash_bytes(vb, (size_t)(ve - vb));
                int32_t lab = lt.find(vb, (size_t)(ve - vb), lh);
                if (lab < 0) lab = lt.add(vb, (size_t)(ve - vb), lh);
                mine[kh >> 58].push_back(Entry{p, (uint32_t)kn, lab, kh});
            }
            p = le;
        }
    };
    auto run = [&](int n, auto&& fn) {
        if (T == 1) {
            for (int i = 0; i < n; ++i) fn(i);
            return;
        }
        std::vector<std::thread> th;
        std::atomic<int> next{0};
        for (int w = 0; w < std::min(T, n); ++w)
            th.emplace_back([&] {
                for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i);
            });
        for (auto& x : th) x.join();
    };
    run(T, parse);
    // labels: local ids -> global ids, merged in text order
    std::vector<std::vector<int32_t>> remap(T);
    for (int ti = 0; ti < T; ++ti) {
        const NameTable& f = labels[ti];
        remap[ti].resize(f.size());
        for (int32_t k = 0; k < f.size(); ++k) {
            const char* p = f.arena.data() + f.off[k];
            NameTable& all = t->strata_sets[t->strata_target].labels;
            int32_t id = all.find(p, f.len[k], f.hash[k]);
            if (id < 0) id = all.add(p, f.len[k], f.hash[k]);
            remap[ti][k] = id;
        }
    }
    // phase 2: every shard takes its buckets in text order (a repeated read id
    // keeps its last label, like dict())
    run(S, [&](int sh) {
        StrataShard& shard = t->strata_sets[t->strata_target].shard[sh];
        size_t more = 0, bytes = 0;
        for (int ti = 0; ti < T; ++ti)
            for (const Entry& en : bucket[(size_t)ti * S + sh]) {
                more += 1;
                bytes += en.kn;
            }
        shard.reserve(shard.ent.size() + more, shard.arena.size() + bytes);
        for (int ti = 0; ti < T; ++ti) {
            const std::vector<Entry>& list = bucket[(size_t)ti * S + sh];
            for (size_t q = 0; q < list.size(); ++q) {
                if (q + 8 < list.size()) shard.prefetch_slot(list[q + 8].kh);
                shard.put(list[q].key, list[q].kn, list[q].kh, remap[ti][list[q].label]);
            }
        }
    });
    if (n_entries) {
        int64_t n = 0;
        for (int sh = 0; sh < S; ++sh) n += (int64_t)t->strata_sets[t->strata_target].shard[sh].ent.size();
        *n_entries = n;
    }
    if (n_labels) *n_labels = t->strata_sets[t->strata_target].labels.size();
    return WK_OK;
}

// Label names: blob (NULL = size query) and off[n_labels + 1]
int wk_tok_strata_labels(wk_tok* t, char* blob, int64_t* off) {
    if (!t || !off) return WK_E_ARG;
    const NameTable& labels = t->strata_sets[t->strata_target].labels;
    int64_t w = 0;
    off[0] = 0;
    for (int32_t i = 0; i < labels.size(); ++i) {
        if (blob) memcpy(blob + w, labels.arena.data() + labels.off[i], labels.len[i]);
        w += labels.len[i];
        off[i + 1] = w;
    }
    return WK_OK;
}

int wk_tok_strata_select(wk_tok* t, int other) {
    if (!t) return WK_E_ARG;
    t->strata_target = other ? t->strata_cur ^ 1 : t->strata_cur;
    return WK_OK;
}

int wk_tok_strata_swap(wk_tok* t) {
    if (!t) return WK_E_ARG;
    t->strata_cur ^= 1;
    t->strata_target = t->strata_cur;
    return WK_OK;
}

// Read-map text (file.write_readmap, file.py:469-500): one line per assigned read,
// "read id <tab> name" or "read id <tab> name:count <tab> ..." for reads split over
// several features (their (feature, count) lists arrive pre-sorted).  Two passes per
// thread range: sizes, then bytes; `out` NULL returns the size only.
int wk_format_readmap(const char* text, const uint64_t* qname, const int32_t* assign, int64_t n_reads,
                      const int64_t* m_off, const int32_t* m_feat, const int32_t* m_count, const char* names_blob,
                      const int64_t* names_off, int32_t n_names, int unassigned, int n_threads, char* out, int64_t cap,
                      int64_t* written) {
    if (!text || !qname || !assign || n_reads < 0 || !names_off || !written) return WK_E_ARG;
    static const char* const kSuffix[3] = {"", "/1", "/2"};
    static const size_t kSuffixLen[3] = {0, 2, 2};
    static const char kUnassigned[] = "Unassigned";
    if (n_threads <= 0) n_threads = (int)std::min<unsigned>(64, std::max(1u, std::thread::hardware_concurrency()));
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, n_reads / 4096 + 1));
    // index of each MULTI read into m_off: running count over the reads
    std::vector<int64_t> mbase(T + 1, 0), size(T, 0), start(T + 1, 0);
    auto range = [&](int i, int64_t& lo, int64_t& hi) {
        lo = n_reads * i / T;
        hi = n_reads * (i + 1) / T;
    };
    auto digits = [](int32_t v) {
        int d = 1;
        while (v >= 10) {
            v /= 10;
            ++d;
        }
        return d;
    };
    auto name_len = [&](int32_t f) -> int64_t {
        return (f >= 0 && f < n_names) ? names_off[f + 1] - names_off[f] : 0;
    };
    // pass 0: MULTI reads per range
    {
        std::vector<std::thread> th;
        for (int i = 0; i < T; ++i)
            th.emplace_back([&, i] {
                int64_t lo, hi, c = 0;
                range(i, lo, hi);
                for (int64_t r = lo; r < hi; ++r) c += assign[r] == WK_ASSIGN_MULTI;
                mbase[i + 1] = c;
            });
        for (auto& x : th) x.join();
        for (int i = 0; i < T; ++i) mbase[i + 1] += mbase[i];
    }
    auto walk = [&](int i, char* dst) -> int64_t {
        int64_t lo, hi, w = 0, m = mbase[i];
        range(i, lo, hi);
        char num[16];
        for (int64_t r = lo; r < hi; ++r) {
            const int32_t a = assign[r];
            const bool multi = a == WK_ASSIGN_MULTI;
            if (!(a >= 0 || multi || (a == WK_ASSIGN_NONE && unassigned))) continue;
            const uint64_t d = qname[r];
            const size_t qo = (size_t)(d >> 24), qn = (size_t)((d >> 2) & 0x3FFFFF), mate = (size_t)(d & 3);
            if (dst) {
                memcpy(dst + w, text + qo, qn);
                memcpy(dst + w + qn, kSuffix[mate], kSuffixLen[mate]);
            }
            w += (int64_t)(qn + kSuffixLen[mate]);
            if (multi) {
                for (int64_t k = m_off[m]; k < m_off[m + 1]; ++k) {
                    const int32_t f = m_feat[k];
                    const int64_t nl = name_len(f);
                    const int dg = digits(m_count[k]);
                    if (dst) {
                        dst[w] = '\t';
                        memcpy(dst + w + 1, names_blob + names_off[f], (size_t)nl);
                        dst[w + 1 + nl] = ':';
                        int32_t v = m_count[k];
                        for (int q = dg - 1; q >= 0; --q) {
                            num[q] = (char)('0' + v % 10);
                            v /= 10;
                        }
                        memcpy(dst + w + 2 + nl, num, (size_t)dg);
                    }
                    w += 2 + nl + dg;
                }
                ++m;
            } else if (a >= 0) {
                const int64_t nl = name_len(a);
                if (dst) {
                    dst[w] = '\t';
                    memcpy(dst + w + 1, names_blob + names_off[a], (size_t)nl);
                }
                w += 1 + nl;
            } else {
                if (dst) {
                    dst[w] = '\t';
                    memcpy(dst + w + 1, kUnassigned, sizeof kUnassigned - 1);
                }
                w += 1 + (int64_t)(sizeof kUnassigned - 1);
            }
            if (dst) dst[w] = '\n';
            w += 1;
        }
        return w;
    };
    {
        std::vector<std::thread> th;
        for (int i = 0; i < T; ++i) th.emplace_back([&, i] { size[i] = walk(i, nullptr); });
        for (auto& x : th) x.join();
    }
    for (int i = 0; i < T; ++i) start[i + 1] = start[i] + size[i];
    *written = start[T];
    if (!out) return WK_OK;
    if (cap < start[T]) return WK_E_CAPACITY;
    {
        std::vector<std::thread> th;
        for (int i = 0; i < T; ++i) th.emplace_back([&, i] { walk(i, out + start[i]); });
        for (auto& x : th) x.join();
    }
    return WK_OK;
}

int wk_tok_subjects(wk_tok* t, int32_t* n_total, int32_t* n_new, int64_t* new_bytes) {
    if (!t || !n_total || !n_new || !new_bytes) return WK_E_ARG;
    *n_total = t->names.size();
    *n_new = t->names.size() - t->reported;
    int64_t bytes = 0;
    for (int32_t i = t->reported; i < t->names.size(); ++i) bytes += t->names.len[i];
    *new_bytes = bytes;
    return WK_OK;
}

int wk_tok_new_subjects(wk_tok* t, char* blob, int32_t* off) {
    if (!t || !off) return WK_E_ARG;
    int64_t w = 0;
    int32_t k = 0;
    off[0] = 0;
    for (int32_t i = t->reported; i < t->names.size(); ++i, ++k) {
        if (blob) memcpy(blob + w, t->names.arena.data() + t->names.off[i], t->names.len[i]);
        w += t->names.len[i];
        off[k + 1] = (int32_t)w;
    }
    t->reported = t->names.size();
    return WK_OK;
}

// DFS pre-order numbering of a rooted tree given as a parent array (host helper
// of the hierarchy flattening, woltka_amd/hierarchy.py; the reference walks its
// child -> parent dict per query instead, tree.py:391-566).  par[v] = parent of
// v, exactly the root has par[r] == r; siblings keep their input order.
// Outputs pre[v] (pre-order number), size[v] (subtree size), depth[v].
// Returns WK_OK, WK_E_ARG (no / several roots, root != expected, parent out of
// range) or WK_E_STATE with *bad = a node that cannot reach the root.
int wk_preorder(const int64_t* par, int64_t n, int64_t expected_root, int64_t* pre, int64_t* size, int64_t* depth,
                int64_t* bad) {
    if (!par || n <= 0 || !pre || !size || !depth) return WK_E_ARG;
    if (n >= (1ll << 31) - 2) return WK_E_ARG;  // (node ids are int32 on the device anyway)
    // The walk is a chain of dependent cache misses (2 M nodes: 74 ms with 64-bit arrays of their own for the
    // children's offsets, numbers, sizes and depths -- four lines per node).  Here: 32-bit offsets, the three
    // results of a node next to each other while the walk runs (one line per node, written out in order
    // afterwards), leaves numbered without a trip through the stack, and the records of the siblings two ahead
    // asked for early.
    int64_t root = -1;
    std::vector<uint32_t> first((size_t)n + 2, 0);  // children CSR: counts at [p + 2], then offsets (see below)
    for (int64_t v = 0; v < n; ++v) {
        const int64_t p = par[v];
        if (p < 0 || p >= n) return WK_E_ARG;
        if (p == v) {
            if (root >= 0) return WK_E_ARG;
            root = v;
        } else {
            first[(size_t)p + 2] += 1;
        }
    }
    if (root < 0 || (expected_root >= 0 && root != expected_root)) return WK_E_ARG;
    for (int64_t v = 2; v <= n + 1; ++v) first[(size_t)v] += first[(size_t)v - 1];
    // (first[p + 1] = where p's children begin; filling them moves it to where they end = where those of p + 1
    // begin: afterwards p's children are kids[first[p] .. first[p + 1]))
    std::vector<uint32_t> kids((size_t)n > 0 ? (size_t)n - 1 : 0);
    for (int64_t v = 0; v < n; ++v)
        if (par[v] != v) kids[first[(size_t)par[v] + 1]++] = (uint32_t)v;  // ascending v: input order
    struct Rec {
        uint32_t pre, size, depth;
    };
    constexpr uint32_t kUnseen = 0xFFFFFFFFu;
    std::vector<Rec> rec((size_t)n, Rec{0u, 0u, kUnseen});
    struct Frame {
        uint32_t node, it, end;
    };
    std::vector<Frame> stack;
    stack.reserve(128);
    uint32_t counter = 0;
    rec[(size_t)root] = Rec{counter++, 0u, 0u};
    stack.push_back(Frame{(uint32_t)root, first[(size_t)root], first[(size_t)root + 1]});
    while (!stack.empty()) {
        Frame& f = stack.back();
        if (f.it < f.end) {
            if (f.it + 2u < f.end) {
                const uint32_t ahead = kids[(size_t)f.it + 2u];
                __builtin_prefetch(&rec[ahead], 1);
                __builtin_prefetch(&first[ahead]);
            }
            const uint32_t c = kids[f.it++];
            const uint32_t d = rec[f.node].depth + 1u;
            const uint32_t lo = first[c], hi = first[(size_t)c + 1];
            if (lo == hi) {  // a leaf: numbered on the spot
                rec[c] = Rec{counter++, 1u, d};
            } else {
                rec[c] = Rec{counter++, 0u, d};
                stack.push_back(Frame{c, lo, hi});  // (f is dangling from here on)
            }
        } else {
            rec[f.node].size = counter - rec[f.node].pre;
            stack.pop_back();
        }
    }
    if ((int64_t)counter != n) {
        for (int64_t v = 0; v < n; ++v)
            if (rec[(size_t)v].depth == kUnseen) {
                if (bad) *bad = v;
                return WK_E_STATE;
            }
    }
    for (int64_t v = 0; v < n; ++v) {
        pre[v] = rec[(size_t)v].pre;
        size[v] = rec[(size_t)v].size;
        depth[v] = rec[(size_t)v].depth;
    }
    return WK_OK;
}

// [*begin, *stop) of a block of SAM text that can be tokenised now: behind the
// leading '@' lines (in_header: the block starts inside them — the first block of
// a file, or one after a header-only block) up to the start of the last run of
// equal query ids unless `final_block`.  Stateless: what wk_tok_text does with a
// block before tokenising it, for callers that feed the device tokenizer.
int wk_tok_sam_span(const char* buf, int64_t len, int final_block, int in_header, int64_t* begin, int64_t* stop,
                    int* in_header_after) {
    if (!buf || len < 0 || !begin || !stop || !in_header_after) return WK_E_ARG;
    bool hdr = in_header != 0;
    const char* b = buf;
    const char* s = buf + len;
    const bool ok = block_span(WK_FMT_SAM, false, buf, len, final_block != 0, hdr, b, s);
    *begin = b - buf;
    *stop = ok ? s - buf : b - buf;
    *in_header_after = hdr ? 1 : 0;
    return ok ? WK_OK : WK_E_STATE;
}

// The same for any format the device tokenizer takes (WK_FMT_*): no header lines
// outside SAM.
int wk_tok_span(int fmt, int extra, const char* buf, int64_t len, int final_block, int in_header, int64_t* begin, int64_t* stop,
                int* in_header_after) {
    if (!buf || len < 0 || !begin || !stop || !in_header_after || fmt < WK_FMT_SAM || fmt > WK_FMT_PAF) return WK_E_ARG;
    bool hdr = fmt == WK_FMT_SAM && in_header != 0;
    const char* b = buf;
    const char* s = buf + len;
    const bool ok = block_span(fmt, extra != 0, buf, len, final_block != 0, hdr, b, s);
    *begin = b - buf;
    *stop = ok ? s - buf : b - buf;
    *in_header_after = hdr ? 1 : 0;
    return ok ? WK_OK : WK_E_STATE;
}

// The header state wk_tok_text continues from (blocks the device tokenizer took
// are not seen by it).
int wk_tok_set_header_state(wk_tok* t, int in_header) {
    if (!t) return WK_E_ARG;
    t->in_header = in_header != 0;
    return WK_OK;
}

// ---- internal: what the device tokenizer (woltka_hip.hip) needs of a wk_tok ------

int wkx_tok_device_ok(const wk_tok* t) { return t->exclude.size() == 0 ? 1 : 0; }

int32_t wkx_tok_n_names(const wk_tok* t) { return t->names.size(); }

void wkx_tok_name(const wk_tok* t, int32_t id, const char** p, uint32_t* len, uint64_t* hash) {
    *p = t->names.ptr(id);
    *len = t->names.len[(size_t)id];
    *hash = t->names.hash[(size_t)id];
}

int32_t wkx_tok_intern(wk_tok* t, const char* p, uint32_t len) {
    const uint64_t hv = hash_bytes(p, len);
    int32_t id = t->names.find(p, len, hv);
    if (id < 0) {
        id = t->names.add(p, len, hv);
        t->dict.insert(p, len, hv, id);
    }
    return id;
}

}  // extern "C"
