// wk_free.hpp — `--rank free` over the packed record stream.
//
// classify.assign_free (woltka/classify.py:54-78) followed by classify.counter
// (classify.py:144-171), for a job set that is one free-rank job: a read with
// one subject goes to the subject's parent (to the subject itself under
// --subok), a read with several to their lowest common ancestor (tree.find_lca,
// tree.py:513-566), nowhere if that is the root or a subject is not in the tree.
//
// The records arrive as packed words (wk_weigh.hpp) whose subject field holds,
// here, the *rank of the subject's node among the distinct subject nodes in
// pre-order* — or kFreeMissing for a name that is not a node (the field is
// rewritten when a chunk is appended, words_to_ranks_kernel; when a later chunk
// brings new subjects the ranks of the records accumulated so far are renumbered,
// ranks_renumber_kernel).  Pre-order ids turn find_lca into "lowest ancestor of
// the smallest id whose subtree holds the largest" (DESIGN.md §2), ranks keep the
// order of the ids, so a read needs the minimum and the maximum of its records'
// ranks and nothing else — and the tables of the look-ups are indexed by rank
// (round 3's records carried node ids: two gathers of rank blocks per read, two
// of the three cache lines a read's look-ups moved).  The position and size
// in every word make the stream self-describing: the last record of a read
// says where the read began — no offsets, no per-read gathers of candidate
// rows (round 2's evaluator: 225 M 16-byte row gathers, 1.5 ms).
//
// The same stream serves one rank job under an option that looks at whole reads
// (classify.assign_rank, classify.py:81-141: --uniq, --above, --major): the
// records then hold each subject's ancestor at the rank (kFreeMissing: none), a
// read whose records agree goes there, and one whose records differ goes to
// their LCA (--above; nowhere if a record has no ancestor at the rank), to the
// value that reaches the threshold (--major: one vote per subject, "none" is a
// value that can win and then assigns nothing) or nowhere (--uniq).
#pragma once
#include "wk_classify.hpp"
#include "wk_device.hpp"
#include "wk_weigh.hpp"

namespace wk {

constexpr uint32_t kFreeMissing = kWordSubjMask;  // feature field of a subject that is not in the tree
constexpr uint32_t kFreeThreads = 1024;

struct FreeArgs {
    const uint32_t* words;  // [n_records] feature | position << 23 | size << 27
    uint32_t n_records;
    // the lowest common ancestor without a walk (DESIGN_HISTORY §3.1c) over the distinct
    // subject nodes d_0 < d_1 < ... in pre-order, which the records name by their
    // index: sparse[k][i] = the smallest among LCA(d_j, d_j+1), j in [i, i + 2^k);
    // parent_d[i] = parent of d_i, self_d[i] = d_i — all three as *result ids*: the
    // possible results (subject nodes and their ancestors) numbered in pre-order,
    // the root 0
    const int32_t* sparse;     // [levels][sparse_m]
    const int32_t* parent_d;   // [sparse_m]
    const int32_t* self_d;     // [sparse_m]
    uint32_t sparse_m;
    uint32_t job, group;
    uint32_t subok, unassigned;
    // a rank job under --uniq / --above / --major instead of the free-rank job:
    uint32_t by_rank, above;
    double major;  // > 0.5 (a value that reaches it is the only one that can), or 0
    uint32_t count_stats;  // reads and records into stat_block (the first of several jobs over the same records)
    // reads per result id ([n_results]: 'Unassigned'), all zero between launches:
    // free_counts_kernel moves them to the count table and clears them
    uint32_t* dense;  // [n_results + 1]
    uint32_t n_results;
    unsigned long long* stat_block;
    // results the workgroup's cache had no room for, as a list per wave: log[wave * log_cap ...], log_cnt[wave] of them
    uint32_t* log;
    uint32_t* log_cnt;
    uint32_t log_cap;
    // (free_stream_kernel<., true>: several jobs over one accumulation) the records hold subject indices, which
    // this job reads through its own table: rank_of_subject[s] < 0 = the subject has no node (at the rank)
    const int32_t* rank_of_subject;
    uint32_t n_subjects;
    int* err;
};

constexpr uint32_t kFreeMiss = 128;     // per-wave ring of results on their way to the log: < 64 left over + <= 64 new
constexpr uint32_t kFreeQueue = 256;    // per-wave queue of reads to evaluate: < 64 left over + <= 128 new
constexpr uint32_t kFreeBlock = 256;    // records a wave loads at a time
constexpr uint32_t kFreeAdvance = 240;  // ... of which it owns the last 240
// LDS of a wave: its queue and its ring of uncached results
__host__ __device__ constexpr uint32_t free_wave_lds() { return kFreeQueue * 8 + kFreeMiss * 4; }

// A wave takes blocks of 256 records — one 16-byte load per lane, the next
// block's issued before this one is looked at — of which it owns the last 240
// (a read of <= 16 records lies inside the block that owns its last record), and
// works on a block in three steps (the stream is bound by vector instructions,
// not by its bytes):
//   1. a lane looks at its own four records: ids, places in their reads; the
//      smallest and largest id of the lane's last run of records of one read, and
//      — four shuffles — what the one to four lanes before it hold of the read its
//      first record continues (a read of <= 16 records began at most four lanes
//      back).  Round 3 listed the read ends in LDS and had a lane walk back over
//      each read's records there: 0.21 ms of config 3's 0.95;
//   2. along its four records a lane keeps the running minimum and maximum and, at
//      a record that ends a read, queues what the read needs: {smallest, largest,
//      what to do};
//   3. 64 queued reads at a time: the table gathers and the counting, every lane
//      busy, as in a kernel with one read per lane but without that kernel's
//      gathers of the reads' records.
// kMajor: the instance for a rank job under --major (the vote along the records;
// the look-up is the value's own node or nothing).
template <bool kMajor, bool kTranslate = false>
__global__ void __launch_bounds__(kFreeThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) free_stream_kernel(FreeArgs a, uint32_t lds_slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long acc[2];
    __shared__ uint32_t need[32];  // --major: the votes a read of so many records asks for
    // The counting: results are node ids and the job is one, so a slot of the
    // workgroup's LDS cache is {node, reads} in 8 bytes.  What the cache cannot
    // hold went to a dense array of counters in HBM by a fire-and-forget atomic
    // in round 3: 17.5 M atomics at config 3, 0.56 GB of write traffic and 0.4 of
    // the kernel's 0.95 ms.  Now such a result is appended to a list of the
    // wave's own — 64 at a time, one 256-byte store — and free_log_kernel counts
    // the lists in LDS afterwards, a slice of the result ids per workgroup.
    // (Tried and not faster than the atomics: one array per XCD, picked by
    // HW_REG_XCC_ID, with atomics of workgroup scope.)
    uint32_t* const dense = a.dense;
    uint32_t* const ckeys = reinterpret_cast<uint32_t*>(smem);
    uint32_t* const ccnt = ckeys + lds_slots;
    const uint32_t cshift = (uint32_t)__clz((int)lds_slots) + 1u;
    for (uint32_t i = threadIdx.x; i < lds_slots; i += blockDim.x) {
        ckeys[i] = 0xFFFFFFFFu;
        ccnt[i] = 0u;
    }
    if (threadIdx.x < 2) acc[threadIdx.x] = 0ull;
    if (kMajor && threadIdx.x < 32u) {  // (the size field has five bits)
        // the smallest v with `v >= size * th` as classify.majority compares them
        // (classify.py:300-317, in fp64), none: a number no read reaches
        uint32_t v = 0;
        while (v <= 17u && !((double)v >= (double)threadIdx.x * a.major)) ++v;
        need[threadIdx.x] = v <= 17u ? v : 0xFFFFFFFFu;
    }
    __syncthreads();
    // (plain LDS pointers: a volatile one turns the accesses into flat ones, each with a wait)
    unsigned char* const mine = smem + (size_t)lds_slots * 8 + (size_t)(threadIdx.x >> 6) * free_wave_lds();
    uint2* const queue = reinterpret_cast<uint2*>(mine);
    uint32_t* const ring = reinterpret_cast<uint32_t*>(mine + kFreeQueue * 8);

    const uint32_t lane = threadIdx.x & (kWave - 1);
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t wave0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    // the queue, the rings and the staged block are the wave's own: its LDS accesses
    // complete in order, the fences keep the compiler from moving them across
    auto settle = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    uint32_t* const my_log = a.log + (size_t)wave0 * a.log_cap;
    uint32_t ring_head = 0, ring_tail = 0, logged = 0;  // (wave-uniform)
    // every lane calls; `have`: this lane has a result
    auto count = [&](bool have, uint32_t node) {
        bool miss = have;
        if (have) {
            uint32_t h = (node * 0x9E3779B1u) >> cshift;
#pragma unroll
            for (int probe = 0; probe < 2; ++probe) {
                uint32_t k = ckeys[h];
                if (k == 0xFFFFFFFFu) {
                    k = atomicCAS(&ckeys[h], 0xFFFFFFFFu, node);
                    if (k == 0xFFFFFFFFu) k = node;
                }
                if (k == node) {
                    atomicAdd(&ccnt[h], 1u);
                    miss = false;
                    break;
                }
                h = (h + 1u) & (lds_slots - 1u);
            }
        }
        const unsigned long long mm = __ballot(miss);
        if (mm == 0ull) return;
        if (miss) ring[(ring_tail + (uint32_t)__popcll(mm & below)) & (kFreeMiss - 1)] = node;
        ring_tail += (uint32_t)__popcll(mm);
        if (ring_tail - ring_head >= (uint32_t)kWave) {
            settle();
            my_log[logged + lane] = ring[(ring_head + lane) & (kFreeMiss - 1)];
            ring_head += kWave;
            logged += kWave;
        }
    };
    // 64 queued reads, one per lane.  (Two per lane, with the gathers of both in
    // flight together, needed more registers than the 64 that eight waves per SIMD
    // leave and was slower, 0.75 against 0.65 ms at config 3; one round of gathers
    // whatever the read needs — the node's rank block twice, two table entries —
    // instead of the branches below: 0.70 ms.)
    auto evaluate = [&](uint32_t head, uint32_t n) {
        const bool on = lane < n;
        const uint2 e = on ? queue[(head + lane) & (kFreeQueue - 1)] : make_uint2(0u, 0u);
        const uint32_t mn = e.x & kWordSubjMask, mx = e.y & kWordSubjMask;
        // (a missing subject carries the largest value of the field: it is the maximum)
        // what to do: 0 = nothing to look up, 1 = parent of node lo, 2 = LCA of lo and hi, 3 = the node lo
        uint32_t kind = 0, lo = mn, hi = mx;
        if constexpr (kMajor) {  // (the block loop has voted: both are where the read goes)
            kind = mn != kFreeMissing ? 3u : 0u;
            hi = 0u;
        } else if (a.by_rank) {
            uint32_t to = kFreeMissing;  // where the read goes without a look at the tree
            bool lca = false;
            if (mn == mx) {
                to = mn;
            } else if (a.above) {
                lca = mx != kFreeMissing;
            }
            kind = lca ? 2u : (to != kFreeMissing ? 3u : 0u);
            lo = lca ? mn : to;
            hi = lca ? mx : 0u;
        } else if (mx == kFreeMissing) {
            kind = 0u;
        } else if (e.y >> kWordSizeShift == 1u) {
            kind = a.subok ? 3u : 1u;
            hi = 0u;
        } else if (mn != mx) {
            kind = 2u;
        } else {  // the same node several times (cannot happen with sets): itself, None if the root
            kind = 3u;
            hi = 1u;
        }
        int32_t res = -1;
        if (on && kind != 0u) {
            const uint32_t p = lo;
            if (kind == 1u) {
                res = a.parent_d[p];
            } else if (kind == 3u) {
                res = a.self_d[p];
                if (res == 0 && hi != 0u) res = -1;  // (several records, all the root: None)
            } else {
                const uint32_t q = hi;
                const uint32_t k = 31u - (uint32_t)__clz((int)(q - p));  // q > p: distinct nodes
                const int32_t* row = a.sparse + (size_t)k * a.sparse_m;
                const int32_t x = row[p], y = row[q - (1u << k)];
                const int32_t anc = x < y ? x : y;
                res = anc == 0 ? -1 : anc;
            }
        }
        if (on && res < 0 && a.unassigned) res = (int32_t)a.n_results;
        count(res >= 0, (uint32_t)res);
    };
    // block b looks at records [240 b - 16, 240 b + 240) and owns the last 240
    const uint32_t n_blocks = (a.n_records + kFreeAdvance - 1) / kFreeAdvance;
    // (one load whatever the place, so that the compiler can count the loads in
    // flight and wait for this block's only, not for the next one's too: the
    // records before the stream's start and behind its end are masked afterwards —
    // the buffer has 64 bytes of room behind the last record)
    auto load_block = [&](uint32_t b) -> uint4 {
        const int64_t r0 = (int64_t)b * kFreeAdvance - 16 + 4 * (int64_t)lane;
        const bool inside = r0 >= 0 && r0 < (int64_t)a.n_records;
        uint4 v = *reinterpret_cast<const uint4*>(a.words + (inside ? r0 : 0));
        const uint32_t left = inside ? (uint32_t)((int64_t)a.n_records - r0) : 0u;
        v.x = left > 0u ? v.x : 0u;
        v.y = left > 1u ? v.y : 0u;
        v.z = left > 2u ? v.z : 0u;
        v.w = left > 3u ? v.w : 0u;
        return v;
    };
    // (kTranslate) subject indices -> this job's ranks, what words_to_ranks_kernel writes for a single job: the
    // table (4 bytes per subject) stays in the caches; the look-ups of a block are issued one block ahead of its
    // use and its words two blocks ahead, so that neither wait is on the wave's path
    auto translate_word = [&](uint32_t w) -> uint32_t {
        if ((w >> kWordSizeShift) == 0u) return w;  // (masked out: no record)
        const uint32_t sidx = w & kWordSubjMask;
        uint32_t f = kFreeMissing;
        if (sidx < a.n_subjects) {
            const int32_t x = a.rank_of_subject[sidx];
            f = x >= 0 ? (uint32_t)x : kFreeMissing;
        } else {
            atomicOr(a.err, kErrFeatureRange);
        }
        return (w & ~kWordSubjMask) | f;
    };
    auto translate = [&](uint4 v) -> uint4 {
        return make_uint4(translate_word(v.x), translate_word(v.y), translate_word(v.z), translate_word(v.w));
    };
    uint32_t my_records = 0;      // (wave-uniform, as are these: a wave's share of < 2^32 records)
    uint32_t head = 0, tail = 0;  // (tail: the reads met so far)
    uint4 cur = load_block(wave0);
    uint4 raw = make_uint4(0u, 0u, 0u, 0u);  // (kTranslate) the next block's words as they are
    if constexpr (kTranslate) {
        cur = translate(cur);
        raw = load_block(wave0 + waves);
    }
    for (uint32_t b = wave0; b < n_blocks; b += waves) {
        uint4 nxt;
        if constexpr (kTranslate) {
            const uint4 raw2 = load_block(b + 2u * waves);
            nxt = translate(raw);
            raw = raw2;
        } else {
            nxt = load_block(b + waves);
        }
        // 1. a lane's own four records: ids, places in their reads
        const uint32_t w4[4] = {cur.x, cur.y, cur.z, cur.w};
        uint32_t f[4], pos[4], size[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[j] = w4[j] & kWordSubjMask;
            pos[j] = (w4[j] >> kWordSubjBits) & 15u;
            size[j] = w4[j] >> kWordSizeShift;
        }
        // the smallest and the largest id of the lane's last run of records of one
        // read (from the last record that begins a read; all four if none does) ...
        uint32_t tmn = f[3], tmx = f[3];
        bool open = pos[3] != 0u;
#pragma unroll
        for (int j = 2; j >= 0; --j) {
            tmn = open && f[j] < tmn ? f[j] : tmn;
            tmx = open && f[j] > tmx ? f[j] : tmx;
            open = open && pos[j] != 0u;
        }
        // ... of the d lanes before this one make what the read of the lane's first
        // record brings along: it began 1 .. 15 records back, in lane - d, and holds
        // every record of the lanes between
        const uint32_t d = (pos[0] + 3u) >> 2;
        uint32_t rmn = 0xFFFFFFFFu, rmx = 0u;
#pragma unroll
        for (uint32_t k = 1; k <= 4u; ++k) {
            const uint32_t smn = (uint32_t)__shfl_up((int)tmn, k), smx = (uint32_t)__shfl_up((int)tmx, k);
            rmn = d >= k && smn < rmn ? smn : rmn;
            rmx = d >= k && smx > rmx ? smx : rmx;
        }
        // 2. the running minimum and maximum along the four records; a record that
        // ends a read is queued: {smallest id,
        // largest id | the record's place and size fields}
        uint2 ent[4];
        unsigned long long ends[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool begins = pos[j] == 0u;
            rmn = begins || f[j] < rmn ? f[j] : rmn;
            rmx = begins || f[j] > rmx ? f[j] : rmx;
            // (a record past the stream's end has size 0, and the 16 records before the owned ones are lanes 0-3's)
            ends[j] = __ballot(lane >= 4u && pos[j] + 1u == size[j]);
            ent[j] = make_uint2(rmn, (w4[j] & ~kWordSubjMask) | rmx);
        }
        if constexpr (kMajor) {
            // --major (classify.majority, classify.py:300-317; one vote per record): the
            // only value that can reach a threshold above one half is the one a
            // Boyer-Moore count leaves, and such counts merge -- (value, lead) of two
            // parts of a read: equal values add their leads, different ones cancel --
            // so the candidate comes along the records the way the minimum and the
            // maximum do: the lane's last run, the d lanes before, the four records.
            // (Rounds 3-4 staged the block in LDS and had the lane of a read's end walk
            // its records twice, a record at a time: 0.43 of the stream's 0.85 ms at
            // config 3, whatever the unrolling.)
            auto step = [](uint32_t& c, uint32_t& l, uint32_t x) {
                const bool same = x == c;
                c = !same && l == 0u ? x : c;
                l = same || l == 0u ? l + 1u : l - 1u;
            };
            uint32_t tc = f[3], tl = 1u;
            bool run = pos[3] != 0u;
#pragma unroll
            for (int j = 2; j >= 0; --j) {
                uint32_t c2 = tc, l2 = tl;
                step(c2, l2, f[j]);
                tc = run ? c2 : tc;
                tl = run ? l2 : tl;
                run = run && pos[j] != 0u;
            }
            uint32_t rc = 0u, rl = 0u;
#pragma unroll
            for (uint32_t k = 1; k <= 4u; ++k) {
                const uint32_t sc = (uint32_t)__shfl_up((int)tc, k), sl = (uint32_t)__shfl_up((int)tl, k);
                const bool same = sc == rc, more = rl >= sl;
                const uint32_t nl = same ? rl + sl : (more ? rl - sl : sl - rl);
                const uint32_t nc = same || more ? rc : sc;
                rc = d >= k ? nc : rc;
                rl = d >= k ? nl : rl;
            }
            uint32_t cand[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t c2 = rc, l2 = rl;
                step(c2, l2, f[j]);
                rc = pos[j] == 0u ? f[j] : c2;
                rl = pos[j] == 0u ? 1u : l2;
                cand[j] = rc;
            }
            // its votes.  The read of a lane's first record ends in this lane at record
            // j0 (if it does): the lanes before, whose last runs are its other records,
            // fetch its candidate from here -- a read that goes on behind a lane's last
            // record ends 1 - 4 lanes further -- count it in their runs, and this lane
            // adds up what the d lanes before counted
            const uint32_t j0 = size[0] - 1u - pos[0];  // (a masked record: size 0, no lane asks)
            const uint32_t headc = j0 == 0u ? cand[0] : (j0 == 1u ? cand[1] : (j0 == 2u ? cand[2] : cand[3]));
            const uint32_t ahead = (size[3] - pos[3] + 2u) >> 2;  // lanes until the read of the last record ends: ceil(records left / 4)
            uint32_t hc = 0u;
#pragma unroll
            for (uint32_t k = 1; k <= 4u; ++k) {
                const uint32_t v = (uint32_t)__shfl_down((int)headc, k);
                hc = ahead == k ? v : hc;
            }
            uint32_t tv = f[3] == hc ? 1u : 0u;
            run = pos[3] != 0u;
#pragma unroll
            for (int j = 2; j >= 0; --j) {
                tv += run && f[j] == hc ? 1u : 0u;
                run = run && pos[j] != 0u;
            }
            uint32_t before = 0u;
#pragma unroll
            for (uint32_t k = 1; k <= 4u; ++k) {
                const uint32_t v = (uint32_t)__shfl_up((int)tv, k);
                before += d >= k ? v : 0u;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t votes = pos[j] > (uint32_t)j ? before : 0u;
#pragma unroll
                for (int i = 0; i <= j; ++i) votes += (uint32_t)(j - i) <= pos[j] && f[i] == cand[j] ? 1u : 0u;
                const uint32_t mn = ent[j].x, mx = ent[j].y & kWordSubjMask;
                const uint32_t to = mn == mx ? mn : (votes >= need[size[j]] ? cand[j] : kFreeMissing);
                ent[j] = make_uint2(to, (w4[j] & ~kWordSubjMask) | to);
            }
        }
        my_records += min((uint32_t)kFreeAdvance, a.n_records - b * kFreeAdvance);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // (two records' worth at a time: < 64 left over + <= 128 new fit the queue)
#pragma unroll
            for (int j = 2 * h; j < 2 * h + 2; ++j) {
                const uint32_t at = tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(ends[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ends[j], 0u));
                if ((ends[j] >> lane) & 1ull) queue[at & (kFreeQueue - 1)] = ent[j];
                tail += (uint32_t)__popcll(ends[j]);
            }
            settle();
            // 3.
            while (tail - head >= (uint32_t)kWave) {
                evaluate(head, (uint32_t)kWave);
                head += (uint32_t)kWave;
                settle();
            }
        }
        cur = nxt;
    }
    settle();
    if (tail != head) evaluate(head, tail - head);
    settle();
    if (lane < ring_tail - ring_head) my_log[logged + lane] = ring[(ring_head + lane) & (kFreeMiss - 1)];
    if (lane == 0) a.log_cnt[wave0] = logged + (ring_tail - ring_head);
    if (lane == 0) {
        atomicAdd(&acc[0], (unsigned long long)tail);
        atomicAdd(&acc[1], (unsigned long long)my_records);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < lds_slots; i += blockDim.x)
        if (ccnt[i]) atomicAdd(&dense[ckeys[i]], ccnt[i]);
    if (threadIdx.x == 0 && a.count_stats) {
        a.stat_block[2 * blockIdx.x] += acc[0];
        a.stat_block[2 * blockIdx.x + 1] += acc[1];
    }
}

// The lists of results the caches had no room for, counted: a workgroup takes a
// slice of kLogBins result ids (32-bit counters in LDS) and one of n_parts
// shares of the waves' lists, and leaves its counters as they are in `partial`
// — no atomics on the way out; free_counts_kernel adds the shares up.  The
// workgroups of one share sit on one XCD (blockIdx % 8), so the share's lists
// come from HBM once and from that XCD's L2 for the other slices.
constexpr uint32_t kLogBins = 36864;  // 144 KB
constexpr uint32_t kListSharesMax = 32;  // shares of the lists: a multiple of 8, as many as fill the CUs once (one workgroup per CU: the bins)
constexpr uint32_t kLogThreads = 1024;

struct FreeLogArgs {
    const uint32_t* log;
    const uint32_t* log_cnt;
    uint32_t log_cap, n_waves;
    uint32_t n_slices, n_parts;
    uint32_t* partial;    // [n_parts][n_slices][kLogBins]
    uint32_t* part_used;  // [n_parts][n_slices] 1: the share met results of the slice
};

__global__ void __launch_bounds__(kLogThreads) free_log_kernel(FreeLogArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t bins[];
    __shared__ uint32_t any;
    const uint32_t r = blockIdx.x >> 3, slice = r % a.n_slices, part = (r / a.n_slices) * 8u + (blockIdx.x & 7u);
    const uint32_t w_lo = (uint32_t)((unsigned long long)part * a.n_waves / a.n_parts);
    const uint32_t w_hi = (uint32_t)((unsigned long long)(part + 1u) * a.n_waves / a.n_parts);
    const uint32_t slot = part * a.n_slices + slice;
    // (nothing listed in this share — every result found room in the caches, as under --uniq at a high rank: done)
    uint32_t listed = 0;
    for (uint32_t w = w_lo + threadIdx.x; w < w_hi; w += kLogThreads) listed |= a.log_cnt[w];
    if (!__syncthreads_or((int)listed)) {
        if (threadIdx.x == 0) a.part_used[slot] = 0u;
        return;
    }
    if (threadIdx.x == 0) any = 0u;
    for (uint32_t i = threadIdx.x; i < kLogBins; i += kLogThreads) bins[i] = 0u;
    __syncthreads();
    const uint32_t base = slice * kLogBins;
    const uint32_t lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    bool met = false;
    // (a wave takes a list at a time, eight 16-byte loads in flight per lane: with one, the kernel waited for memory
    // 0.17 ms at config 3's 17.5 M entries)
    constexpr uint32_t kDeep = 8;
    for (uint32_t w = w_lo + wv; w < w_hi; w += kLogThreads / kWave) {
        const uint32_t cnt = a.log_cnt[w];
        const uint32_t* src = a.log + (size_t)w * a.log_cap;
        for (uint32_t i0 = 0; i0 < cnt; i0 += 4u * kWave * kDeep) {
            uint4 v[kDeep];
#pragma unroll
            for (uint32_t k = 0; k < kDeep; ++k) {
                const uint32_t i = i0 + 4u * kWave * k + 4u * lane;
                // (a list's room is a multiple of 64 entries, and the lists are one allocation: a load that
                // starts inside the list stays inside the allocation)
                v[k] = i < cnt ? *reinterpret_cast<const uint4*>(src + i) : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
            }
#pragma unroll
            for (uint32_t k = 0; k < kDeep; ++k) {
                const uint32_t i = i0 + 4u * kWave * k + 4u * lane;
                const uint32_t id[4] = {v[k].x - base, v[k].y - base, v[k].z - base, v[k].w - base};
#pragma unroll
                for (uint32_t j = 0; j < 4u; ++j)
                    if (i + j < cnt && id[j] < kLogBins) {
                        atomicAdd(&bins[id[j]], 1u);
                        met = true;
                    }
            }
        }
    }
    if (met) any = 1u;
    __syncthreads();
    if (threadIdx.x == 0) a.part_used[slot] = any;
    if (!any) return;
    uint32_t* out = a.partial + (size_t)slot * kLogBins;
    for (uint32_t i = threadIdx.x; i < kLogBins; i += kLogThreads) out[i] = bins[i];
}

// the dense counters + the shares of free_log_kernel -> the count table (weight L
// per read, DESIGN_HISTORY §3.2); the dense counters are cleared on the way
__global__ void __launch_bounds__(256) free_counts_kernel(uint32_t* __restrict__ dense, uint32_t n_results,
                                                          const int32_t* __restrict__ result_node, uint32_t job, uint32_t group,
                                                          const uint32_t* __restrict__ partial, const uint32_t* __restrict__ part_used,
                                                          uint32_t n_slices, uint32_t n_parts, CountTable table) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_results) return;
    unsigned long long n = dense[i];
    if (n) dense[i] = 0u;
    const uint32_t slice = i / kLogBins, bin = i - slice * kLogBins;
    for (uint32_t part = 0; part < n_parts; ++part) {
        const uint32_t slot = part * n_slices + slice;
        if (part_used[slot]) n += partial[(size_t)slot * kLogBins + bin];
    }
    if (!n) return;
    table_add(table, make_key(job, 0u, group, i == n_results ? (uint32_t)WK_FEATURE_UNASSIGNED : (uint32_t)result_node[i]), n * WK_WEIGHT_L);
}

// subject indices -> ranks of their nodes (`src` may be `dst`: chunks appended to
// a single-job accumulation are rewritten in place); rank_of_subject[s] < 0: the
// subject has no node
__global__ void __launch_bounds__(256) words_to_ranks_kernel(const uint32_t* src, uint32_t* dst, uint32_t n,
                                                             const int32_t* __restrict__ rank_of_subject, uint32_t n_subjects,
                                                             int* __restrict__ err) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t w = src[i], s = w & kWordSubjMask;
    uint32_t f = kFreeMissing;
    if (s < n_subjects) {
        const int32_t x = rank_of_subject[s];
        f = x >= 0 ? (uint32_t)x : kFreeMissing;
    } else if (w >> kWordSizeShift) {
        atomicOr(err, kErrFeatureRange);
    }
    dst[i] = (w & ~kWordSubjMask) | f;
}

// the subject set has grown: ranks of the records accumulated so far -> ranks among the larger set
__global__ void __launch_bounds__(256) ranks_renumber_kernel(uint32_t* words, uint32_t n, const int32_t* __restrict__ new_of_old,
                                                             uint32_t n_old) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t w = words[i], r = w & kWordSubjMask;
    if (r < n_old) words[i] = (w & ~kWordSubjMask) | (uint32_t)new_of_old[r];
}

}  // namespace wk
