// wk_strata.hpp — the read -> stratum map of `--stratify`, joined on the device.
//
// workflow.read_strata (woltka/workflow.py:912-938) reads a sample's map into
// a dict — file.read_map_uniq (file.py:368-385): lines `read <tab> label` with
// exactly two columns, the label right-stripped, a repeated read keeps its last
// label — and classify.counter_strat (classify.py:216-249) looks every query up
// in it: queries that are not there are skipped.  Config 5's second pass does
// that for 20 M reads per sample.  Here the map's text goes to the device as it
// is and becomes an open-addressing table there: a slot holds the 64-bit hash
// of a read id and the line that won it (the last one); the reads of the
// alignment blocks (wk_dtok.hpp, "ex" flavour) probe it with the hash of their
// QNAME + mate suffix and compare the bytes of the winning line's key, so the
// join is exact — the hash only finds the candidate.  Labels are interned the
// same way in a small table; the host gives each label its (sample, stratum)
// group id.
//
// Anything the kernels cannot decide like the reference sets a flag and the
// host's join takes the sample (wk_tok_strata_*): two different read ids (or
// labels) with the same 64-bit hash, more labels than the label table holds.
#pragma once
#include "wk_device.hpp"

namespace wk {

constexpr uint32_t kStrataThreads = 256;
constexpr uint32_t kStrataLabelSlots = 1u << 17;
constexpr uint32_t kStrataNoPair = 0xFFFFFFFFu;
constexpr unsigned long long kLabelEmpty = ~0ull;

constexpr uint32_t kStrataCollision = 1;   // different keys / labels, equal hashes
constexpr uint32_t kStrataLabelsFull = 2;  // more distinct labels than half the label table
constexpr uint32_t kStrataOddBytes = 4;    // a \r inside a line, or a label ending in a byte str.rstrip() may have an opinion on
                                           // (\x1c-\x1f, UTF-8 spaces): the host's join reads such text like Python

struct StrataSlot {
    unsigned long long hash;  // 0: empty
    uint32_t line1;           // 1 + the last line with this key
    uint32_t pad;
};

struct LabelSlot {
    unsigned long long hash;  // ~0: empty
    uint32_t rep;             // first line with this label
    uint32_t pad;
};

struct StrataArgs {
    const unsigned char* text;
    uint32_t n;
    const uint32_t* line_start;  // [n_lines + 1]
    uint32_t n_lines;
    uint32_t* line_tab;              // [n_lines] key length (the tab's offset in the line); kStrataNoPair: not a pair
    uint32_t* line_vlen;             // [n_lines] length of the right-stripped label
    unsigned long long* line_hash;   // [n_lines] hash of the key
    uint32_t* line_label;            // [n_lines] label slot
    StrataSlot* slots;
    uint32_t mask;
    LabelSlot* labels;
    uint2* label_text;               // [kStrataLabelSlots] (offset, length) of the label's text
    const int32_t* label_group;      // [kStrataLabelSlots] group id the host gave the label
    uint32_t* state;                 // [0] flags [1] pairs [2] labels
};

// streaming 64-bit hash of a byte string (8 bytes per round, FNV-style mixing, avalanche at the end)
struct NameHash {
    unsigned long long h = 0xcbf29ce484222325ull, acc = 0;
    uint32_t k = 0, n = 0;
    __device__ __forceinline__ void put(unsigned char b) {
        acc |= (unsigned long long)b << (8u * k);
        ++n;
        if (++k == 8u) {
            h = (h ^ acc) * 0x100000001b3ull;
            h ^= h >> 29;
            acc = 0;
            k = 0;
        }
    }
    __device__ __forceinline__ unsigned long long done() {
        h = (h ^ acc) * 0x100000001b3ull;
        h ^= (unsigned long long)n * 0x9E3779B97F4A7C15ull;
        h ^= h >> 32;
        h *= 0xff51afd7ed558ccdull;
        h ^= h >> 33;
        return h;
    }
};

// a thread per line of the map: key / label extents, hashes, table inserts
__global__ void __launch_bounds__(kStrataThreads) strata_parse_kernel(StrataArgs s) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n_lines) return;
    const uint32_t lo = s.line_start[i];
    uint32_t hi = s.line_start[i + 1];  // behind the newline (n + 1 for a last line without one)
    hi = hi > lo ? hi - 1u : lo;
    if (hi > s.n) hi = s.n;
    uint32_t tab = hi;
    NameHash kh;
    for (uint32_t p = lo; p < hi; ++p) {
        const unsigned char b = s.text[p];
        if (b == '\t') {
            tab = p;
            break;
        }
        if (b == '\r') atomicOr(&s.state[0], kStrataOddBytes);  // (universal newlines: the line ends here)
        kh.put(b);
    }
    bool pair = tab < hi;
    for (uint32_t p = tab + 1u; pair && p < hi; ++p) pair = s.text[p] != '\t';  // exactly two columns (file.py:383)
    if (!pair) {
        s.line_tab[i] = kStrataNoPair;
        return;
    }
    uint32_t ve = hi;  // value.rstrip() (file.py:384)
    while (ve > tab + 1u) {
        const unsigned char b = s.text[ve - 1u];
        if (b == '\r' || b == ' ' || b == '\v' || b == '\f' || b == '\n')
            --ve;
        else
            break;
    }
    if (ve > tab + 1u) {
        const unsigned char b = s.text[ve - 1u];
        if (b >= 0x80 || (b >= 0x1c && b <= 0x1f)) atomicOr(&s.state[0], kStrataOddBytes);
    }
    NameHash lh;
    for (uint32_t p = tab + 1u; p < ve; ++p) {
        if (s.text[p] == '\r') atomicOr(&s.state[0], kStrataOddBytes);
        lh.put(s.text[p]);
    }
    unsigned long long hk = kh.done(), hl = lh.done();
    if (hk == 0ull) hk = 1ull;
    if (hl == kLabelEmpty) hl = kLabelEmpty - 1ull;
    s.line_tab[i] = tab - lo;
    s.line_vlen[i] = ve - (tab + 1u);
    s.line_hash[i] = hk;
    // label
    // (a map of 20 M reads names a few thousand labels: once a label sits in its slot every later line
    // finds it with a plain load -- a compare-and-swap per line on the slot of a common label, a
    // running minimum per line on its representative and one add per line to the pair counter were
    // 47 ms per map of config 5, profiles/r05_e2e_twopass2_kernel_stats.csv)
    uint32_t q = (uint32_t)hl & (kStrataLabelSlots - 1u);
    for (uint32_t tries = 0;; ++tries) {
        unsigned long long old = *reinterpret_cast<volatile unsigned long long*>(&s.labels[q].hash);
        if (old == kLabelEmpty) old = atomicCAS(&s.labels[q].hash, kLabelEmpty, hl);
        if (old == kLabelEmpty) {
            if (atomicAdd(&s.state[2], 1u) >= kStrataLabelSlots / 2u) atomicOr(&s.state[0], kStrataLabelsFull);
            break;
        }
        if (old == hl) break;
        if (tries >= kStrataLabelSlots) {
            atomicOr(&s.state[0], kStrataLabelsFull);
            break;
        }
        q = (q + 1u) & (kStrataLabelSlots - 1u);
    }
    if (*reinterpret_cast<volatile uint32_t*>(&s.labels[q].rep) > i) atomicMin(&s.labels[q].rep, i);
    s.line_label[i] = q;
    // key: the last line wins (dict())
    uint32_t h = (uint32_t)(hk ^ (hk >> 32)) & s.mask;
    for (;;) {
        const unsigned long long old = atomicCAS(&s.slots[h].hash, 0ull, hk);
        if (old == 0ull || old == hk) break;
        h = (h + 1u) & s.mask;
    }
    atomicMax(&s.slots[h].line1, i + 1u);
    // the pairs of the map: one add per wave
    const unsigned long long pairs = __ballot(true);
    if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)pairs) - 1)) atomicAdd(&s.state[1], (uint32_t)__popcll(pairs));
}

__device__ __forceinline__ bool same_bytes(const unsigned char* a, const unsigned char* b, uint32_t n) {
    for (uint32_t k = 0; k < n; ++k)
        if (a[k] != b[k]) return false;
    return true;
}

// a thread per line: equal hashes must be equal strings; representatives note their label's text
__global__ void __launch_bounds__(kStrataThreads) strata_verify_kernel(StrataArgs s) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n_lines || s.line_tab[i] == kStrataNoPair) return;
    const uint32_t lo = s.line_start[i], kn = s.line_tab[i], vn = s.line_vlen[i];
    const unsigned long long hk = s.line_hash[i];
    uint32_t h = (uint32_t)(hk ^ (hk >> 32)) & s.mask;
    while (s.slots[h].hash != hk) h = (h + 1u) & s.mask;
    const uint32_t w = s.slots[h].line1 - 1u;
    if (w != i) {
        const uint32_t wl = s.line_start[w];
        if (s.line_tab[w] != kn || !same_bytes(s.text + lo, s.text + wl, kn)) atomicOr(&s.state[0], kStrataCollision);
    }
    const uint32_t q = s.line_label[i], rep = s.labels[q].rep;
    if (rep == i) {
        s.label_text[q] = make_uint2(lo + kn + 1u, vn);
    } else {
        const uint32_t rl = s.line_start[rep];
        if (s.line_vlen[rep] != vn || !same_bytes(s.text + lo + kn + 1u, s.text + rl + s.line_tab[rep] + 1u, vn))
            atomicOr(&s.state[0], kStrataCollision);
    }
}

// group id of the read named qname[0, qn) + ("/1" | "/2" for mate 1 | 2), -1 if the map does not hold it
__device__ __forceinline__ int32_t strata_lookup(const StrataArgs& s, const unsigned char* qname, uint32_t qn, uint32_t mate) {
    NameHash nh;
    for (uint32_t k = 0; k < qn; ++k) nh.put(qname[k]);
    if (mate) {
        nh.put('/');
        nh.put((unsigned char)('0' + mate));
    }
    unsigned long long hk = nh.done();
    if (hk == 0ull) hk = 1ull;
    const uint32_t len = qn + (mate ? 2u : 0u);
    uint32_t h = (uint32_t)(hk ^ (hk >> 32)) & s.mask;
    for (;;) {
        const StrataSlot slot = s.slots[h];
        if (slot.hash == 0ull) return -1;
        if (slot.hash == hk) {
            const uint32_t w = slot.line1 - 1u;
            if (s.line_tab[w] != len) return -1;
            const unsigned char* key = s.text + s.line_start[w];
            if (!same_bytes(key, qname, qn)) return -1;
            if (mate && (key[qn] != '/' || key[qn + 1u] != (unsigned char)('0' + mate))) return -1;
            return s.label_group[s.line_label[w]];
        }
        h = (h + 1u) & s.mask;
    }
}

}  // namespace wk
