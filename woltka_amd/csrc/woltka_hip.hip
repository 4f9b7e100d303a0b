// woltka_hip.hip — C-ABI implementation of include/woltka_hip.h for gfx950.
// Host-side resource management + kernel launches.  No CPU compute path: every
// entry point that does work launches HIP kernels on the context's stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <ctime>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/woltka_hip.h"
#include "../../include/woltka_hip_measure.h"
#include "wk_classify.hpp"
#include "wk_device.hpp"
#include "wk_dtok.hpp"
#include "wk_dtok_fused.hpp"
#include "wk_free.hpp"
#include "wk_ordinal.hpp"
#include "wk_stripe.hpp"
#include "wk_readmap.hpp"
#include "wk_tok_internal.h"
#include "wk_weigh.hpp"

using namespace wk;

namespace {

thread_local std::string g_last_error;  // for calls without a context

// growable device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            cap = 0;
            if (e != hipSuccess) return e;
        }
        size_t want = bytes + bytes / 4 + 256;  // headroom against re-allocation churn
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            e = hipMalloc(&p, bytes);
            want = bytes;
            if (e != hipSuccess) {
                p = nullptr;
                return e;
            }
        }
        cap = want;
        return hipSuccess;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T* as() const {
        return reinterpret_cast<T*>(p);
    }
};

// Kernel timing (wk_profile_kernels): the timed launches of one API call form a
// chain of events — one at the call's start, one behind every timed family — and
// a family's time runs from the event before it to its own.  The brackets of a
// call therefore add up to the call's time on the stream (a pair of events of
// its own around every small kernel read 30 us for 10 us kernels).
struct KernelTimer {
    hipEvent_t b = nullptr;     // behind the family's launches
    hipEvent_t from = nullptr;  // the chain's event before them (not owned)
    bool valid = false;
};

}  // namespace

struct wk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    hipDeviceProp_t prop;

    // hierarchy
    DevBuf nodes, rank_code;
    int32_t n_nodes = 0;
    DevBuf rank_tab[WK_MAX_JOBS * 4];
    bool rank_tab_valid[WK_MAX_JOBS * 4] = {};
    int32_t rank_tab_code[WK_MAX_JOBS * 4] = {};
    int64_t rank_tab_nodes[WK_MAX_JOBS * 4] = {};  // nodes that carry the slot's rank (distinct results a job can have)
    std::vector<int32_t> rank_code_host;           // host copy of the rank codes (for those counts)
    // host copies of the tree and the subject table + what `--rank free` looks up
    // instead of walking (ClassifyArgs::free_sparse): subjects ranked by pre-order
    // id, LCAs of rank-adjacent subjects as a sparse table; rebuilt when either changes
    std::vector<int32_t> parent_host, last_host, subj_feat_host;
    int tree_serial = 0, subj_serial = 0, free_tree = -1, free_subj = -1;
    DevBuf f_rank, f_sparse;
    // tables of the per-read stream (wk_free.hpp) for one job: over the nodes its
    // records can name — the subjects' nodes (kind 1, `--rank free`) or their
    // ancestors at the job's rank (kind 2)
    struct StreamTables {
        DevBuf dsparse, dparent, dself, rnode;
        // the distinct nodes the stream's records can name, ascending — a record names one by its index — and per
        // subject that index (-1: the subject has no node); what they were made for
        std::vector<int32_t> dn_host;
        DevBuf subj_rank;
        int rk_kind = -1, rk_slot = -1, rk_tree = -1, rk_subj = -1, rk_rank = -1;
        DevBuf subj_node;                    // kind 2: ancestor at the rank per subject (-1: none)
        std::vector<int32_t> subj_node_host;
        int node_slot = -1, node_tree = -1, node_rank = -1;  // what subj_node_host was made for
        uint32_t dm = 0, results = 0;
        int kind = -1, slot = -1, tree = -1, subj = -1, rank = -1;  // what the tables were made for
    };
    StreamTables st[WK_MAX_JOBS];
    DevBuf w_tmp;  // (unused since the stream translates subject indices itself: free_stream_kernel<., true>)
    DevBuf w_renum;  // old rank -> new rank when the subject table grows under accumulated records
    DevBuf f_dense;  // reads per result node of the free-rank stream
    DevBuf f_log, f_log_cnt, f_partial, f_part_used;  // results the stream's LDS caches had no room for, per wave; their counts per share (free_log_kernel)
    int f_dense_tree = -1;
    uint32_t f_m = 0;
    int use_free_sparse = 1;
    int log_parts_opt = 0;                         // 0 = auto, else 256 / 1024

    // compact subject table (optional)
    DevBuf subj_feat, subj_rows, dense_slab, plog, plog_cnt;
    int use_plog = 1;                     // partitioned miss log (auto: large chunks without dense bins)
    int64_t plog_max_bytes = 4ll << 30;   // upper bound of the miss-log buffer
    int32_t n_subjects = 0;
    int32_t max_subject_feature = -1;
    int use_dense = 1;
    int32_t rows_w = 0;
    std::vector<int> rows_sig;  // rank slots the rows were built for (+ n_subjects)
    bool subj_indexed = false;  // staged chunk carries subject indices

    // genes
    DevBuf gene4, g_grid, g_first, g_goff, g_shift;  // gene tables (wk_set_genes; wk_ordinal.hpp)
    bool genes_set = false;
    bool genes_by_index = false;   // gene lists carry gene table indices (wk_ordinal_pair_genes), translated to features for the classification
    int gene_index_opt = 0;        // asked for by the next wk_set_genes (option "gene_index_pairs")
    DevBuf g_feature, o_pairs_feat;
    int grid_density = 2;  // grid cells per gene (rounded up to a power of two per genome)
    int32_t n_genomes = 0, n_genes = 0;

    // count table
    DevBuf tkeys, tvals;
    uint64_t slots = 0;

    // staged classify chunk
    DevBuf c_subj, c_qoff, c_group;
    const int32_t* cur_subj = nullptr;  // points into c_subj or o_pairs
    const int32_t* cur_qoff = nullptr;
    int64_t n_reads = 0, n_records = 0;
    bool has_group = false, subj_is_set = false, chunk_valid = false;
    int32_t group_base = 0;  // group of every read of a chunk staged without a group array

    // staged ordinal chunk
    DevBuf o_genome, o_beg, o_end, o_len, o_hoff, o_cnt, o_ub, o_first2, o_poff, o_pairs, o_qoff, o_tile_sum, o_tile_off;
    int64_t n_hits = 0, o_reads = 0;
    double th = 0.8;
    bool ord_valid = false;
    // hits binned by genome stripe (wk_stripe.hpp): the stripes of the gene tables, the sorted chunk
    DevBuf g_gene_off, g_stripe_of, g_stripes;
    std::vector<StripeInfo> stripes_host;
    bool stripes_usable = false;
    int use_fused = 1;         // (0: every block through the six kernels of wk_dtok.hpp; wk_tune "dtok_fused")
    uint32_t fused_ablate = 0; // (measurement, wk_tune "fz_ablate")
    std::atomic<bool> fused_streak{false}; // the block scanned last went through the one kernel: copies are not followed by a newline count
    double dt_lpb = 0.0;       // lines per byte of that block
    int fused_per_cu = 3;      // persistent workgroups per CU of dtok_fused_kernel (wk_tune "dtok_fused_per_cu")
    int64_t fused_blocks = 0, fused_fallbacks = 0;   // blocks the fused kernel did / handed back
    hipEvent_t res_ev = nullptr;
    int use_stripes = 1;       // (0: the gather kernels of wk_ordinal.hpp for every read; measurement)
    int64_t chunks_sorted = 0, chunks_gathered = 0;   // (measurement: wk_ordinal_chunk_counts)
    bool acc_open = false;     // the staged hits are blocks piled up by wk_dtok_stage_hits_append, not counted yet
    int64_t stripes_min_hits = 4000000;  // chunks below this keep the gather kernels (wk_tune "stripes_min")
    DevBuf sb_cnt, sb_tot, sb_base, sb_binned, sb_units, sb_over, sb_stat;
    DevBuf r_genome, r_beg, r_end, r_len, r_hoff;
    bool sb_valid = false;     // the staged chunk has been sorted
    int64_t sb_single = 0, sb_rest_reads = 0, sb_rest_hits = 0;
    uint32_t sb_n_units = 0;

    // misc device scalars: [0]=err(int) [3]=total pairs [4]=compact counter [5]=log cursor
    DevBuf scalars;
    DevBuf stat_block;  // kStatBlocks x (reads, records), see flush_stats()
    DevBuf log;         // contribution log of WK_F_SIZED jobs, cursor = scalars[5]
    int64_t log_cap = 0;
    DevBuf assign_out, fetch_k, fetch_v;
    int64_t stat_pairs = 0;

    // timing
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool timer_closed = false;
    bool profile = false;
    std::map<std::string, KernelTimer> ktimers;
    hipEvent_t kt_step = nullptr;   // recorded where a timed API call starts
    hipEvent_t kt_tail = nullptr;   // last event of the chain
    int kt_depth = 0;
    double lap_s[4] = {0, 0, 0, 0};   // (wk_tune "lap_print") seconds inside wk_dtok_copy / scan / waits of scan / emit
    double lap_x[6] = {0, 0, 0, 0, 0, 0};  // ... of the scan: until the buffer is claimed / line starts queued / parse queued / emission queued / finish / names
    double lap_c[5] = {0, 0, 0, 0, 0};  // ... of the copy: streams + events / device buffer / copy queued / count queued; [4] the longest device-buffer step
    double lap_copy_ms = 0;           // ... and the copies' own durations (events around each on the copy stream)
    int64_t lap_copy_bytes = 0;
    hipEvent_t copy_ev0[192] = {};    // (kTextBufs) start of a block's copy
    hipEvent_t copy_evm[192] = {};    // ... its end (the newline count behind it runs on a stream of its own)

    int lds_slots = 8192;  // LDS front-cache slots per workgroup (16 B each = 128 KiB)
    int threads = 1024;    // workgroup size of the direct classify kernel
    int use_lds = 1;
    int blocks_per_cu = 1;
    int ablate = 0;  // honoured only by -DWK_ABLATE measurement builds
    int tiled = 0;  // LDS-staged classify kernel (0: direct one-thread-per-read kernel)
    // two-class split (single-candidate pass, then the generic pass over the
    // rest): 0 = off, 1/2 = on whenever the chunk qualifies
    int use_split = 1;
    int single_blocks_per_cu = 1;
    int use_count_kernel = 1;  // count-first pass as count_subjects_kernel (statically pipelined)
    DevBuf left_mask, left_list, first_slab;
    // weighted subject histogram (wk_weigh.hpp): 0 = off, 1 = auto, 2 = whenever applicable
    int use_weigh = 1;
    int tally_per_cu = 3;    // tally workgroups (512 threads) per CU
    int tally_slots = 1024;  // hash-cache slots of a tally workgroup (a power of two)
    int use_tally = 1;      // wk_ordinal_count: genes tallied per read straight from the matches
    int match_lds = 1;      // per-genome words of the coordinate grid in LDS when they fit
    bool listed_only = false;  // wk_classify_staged evaluates only the reads of left_mask
    int bins_ring = 4;  // measurement knob: load stages in flight of weigh_bins_kernel
    DevBuf w_slab, w_hi, w_invalid;
    size_t w_hi_clean = 0;        // leading entries of w_hi known to be zero
    // read size per record of the staged chunk + reads the histogram does not
    // cover + totals: [0] derived at staging, [1] derived again with the
    // validity bits of the current subject rows (when some subject has one set)
    DevBuf c_rk[2], rk_left[2], rk_totals;
    bool rk_valid[2] = {false, false};
    int64_t rk_reads[2] = {0, 0}, rk_records[2] = {0, 0};
    int64_t stage_serial = 0, rows_serial = 0, rk1_stage = -1, rk1_rows = -1;
    int64_t stat_extra_reads = 0, stat_extra_records = 0;  // statistics added on the host
    bool rows_any_invalid = false;  // some subject lacks an ancestor at a rank column of the current rows
    // packed records of the native tokenizer accumulated over the chunks of one
    // sample (wk_words_*): one launch of the weighted histogram at the end
    DevBuf c_words;   // (per-read stream, w_mode != 0: one buffer, reads contiguous)
    // weighted histogram (w_mode 0): the records by slice of the subject table (wk_weigh.hpp), appended through device
    // cursors; unsliced (more than kMaxStreams slices) = one stream for the team kernel
    DevBuf w_stream[kMaxStreams], w_cursor, w_stage, w_backup, w_backup2;
    void* w_backup_cur = nullptr;   // the cursors in front of the block emitted last (what a block that is not kept restores)
    bool fz_chain = false;          // the one-kernel tokenizer's last launch left the next block's "before" and cleared scalars behind
    int fz_parity = 0;              // ... in w_backup (0) / w_backup2 (1) / w_backup3 (2): `fz_bk(i)`
    DevBuf w_backup3;
    // blocks of the one-kernel tokenizer whose verdict has not been read yet (wk_dtok_scan_emit_begin / _end): at most
    // two, oldest first -- the second one's kernel is queued behind the first one's, so that the device never waits
    // for the host between blocks
    struct LagSlot {
        int buf = -1;                // text buffer (-1: a resident block)
        const char* src = nullptr;   // the block's tag (put back when the block is handed back)
        uint32_t n = 0;
        bool open_end = false;
        int ring = 0;                // fz_bk(ring) = the cursors in front of the block
        uint32_t lines_est = 0;
        int host_slot = 0;           // slot of the pinned scratch its scalars land in
        hipEvent_t ev = nullptr;
        uint32_t seq = 0;
    };
    LagSlot lag[2];
    int lag_count = 0;
    hipEvent_t lag_ev[2] = {nullptr, nullptr};
    int lag_next_ev = 0;
    int fz_back_streak = 0, fz_skip = 0;   // blocks in a row the one kernel handed back; blocks it is left out for
    bool count_ahead = true;        // wk_set_option "dtok_count_ahead"
    bool lag_poll = true;           // a block's end is seen in pinned memory instead of waited for through an event (WOLTKA_LAG_POLL=0: the event; 127.7 against 126.3 us per block, tools/lag_probe.py)
    uint32_t lag_seq = 0;
    bool lag_enabled = true;        // (WOLTKA_NO_LAG=1: every block's verdict is read before the next is launched)
    bool fz_no_chain = false;       // (measurement, WOLTKA_FZ_NO_CHAIN=1: the small kernel in front of every block)
    int w_streams = 0;              // streams the open accumulation writes to
    bool w_sliced = false;
    bool w_counts_known = false;    // w_count holds the cursors' values
    unsigned long long w_count[kMaxStreams] = {};
    int64_t w_records = 0, w_reads = 0;  // accumulated so far
    std::vector<wk_job> w_jobs;          // the job set they will be classified under
    int32_t w_group = 0;
    bool w_open = false;
    // host memory of others registered for asynchronous copies (wk_host_register): a copy may not span two of them
    struct HostReg {
        const char* p;
        size_t n;
    };
    std::mutex reg_mu;
    std::vector<HostReg> regs;
    int w_mode = 0;  // 0: subject indices for the weighted histogram; the stream of wk_free.hpp: 1: feature ids (one free-rank job), 2: ancestors at the job's rank (one rank job under --uniq / --above / --major), 3: subject indices, rewritten per job when the sample is classified (several such jobs)
    int rank_serial = 0;              // bumped by wk_build_rank_table
    int free_per_cu = 2, free_threads = 1024, free_slots = 4096;  // launch shape of the free-rank stream (measurement knobs; DESIGN_HISTORY §3.1d)  // measurement knobs of the free-rank stream: workgroups per CU, windows in flight per wave
    int32_t max_gene_feature = 0;
    int use_range_log = 1;  // (0: the hashed miss log for the gene tally too; measurement)
    int words_keep = 0;  // measurement: wk_words_flush leaves the accumulated records in place
    static constexpr int kStageSlots = 8;
    hipEvent_t slot_ev[kStageSlots] = {};
    bool slot_busy[kStageSlots] = {};
    std::vector<void*> host_blocks;      // wk_host_alloc (guarded by host_mu: the host layer pins buffers on several threads)
    std::mutex host_mu;
    // device tokenizer (wk_dtok.hpp): the block scanned last and the dictionary mirror
    // the block being scanned + the blocks copied ahead of it.  Three are enough while the scans keep up with
    // the link; the host layer's reader may start before anything can be scanned (the hierarchy is still
    // being read) and then runs as far ahead as it likes -- HBM is what a 288 GB device has to spare; the
    // buffers are allocated as they are first used, the lowest free one first
    static constexpr int kTextBufs = 192;
    static_assert(sizeof(copy_ev0) / sizeof(copy_ev0[0]) == kTextBufs, "copy_ev0 has kTextBufs entries");
    // (text buffers are cut from slabs of kSlabBufs: one hipMalloc per GB instead of one per block while the reader
    // runs ahead -- every allocation holds the runtime's lock against the launches of the scans; a block larger than a
    // slab's share gets a buffer of its own, d_textbuf[k])
    static constexpr int kSlabBufs = 16;
    static constexpr size_t kTextStride = ((size_t)66 << 20) + 256;
    DevBuf d_textslab[kTextBufs / kSlabBufs];
    unsigned char* d_textptr[kTextBufs] = {};
    DevBuf d_textbuf[kTextBufs], d_tiles, d_tile_off, d_lines, d_lsubj, d_lmeta, d_start, d_first, d_unknown, d_state, d_dict, d_dict2, d_names16, d_arena, d_submap;
    bool dt_mapped = false;       // the block scanned last has had its lines' ids translated (dtok_submap_kernel)
    bool dt_submap_on = false;    // a map has been given (it may be empty still)
    bool dt_has_excl = false;     // the map holds kLineExcluded: `--exclude` on the device text route
    uint32_t dt_submap_n = 0;     // wk_dtok_subject_map (0: the tokenizer's ids are the subject indices)
    DevBuf d_lbeg, d_lend, d_llen, d_lscan, d_gmap;  // "ex" flavour
    bool dt_extra = false;
    // measurement (woltka_hip_measure.h): blocks of text that are resident on the device already
    // (wk_text_upload) -- the scan of such a block copies nothing
    struct ResidentText {
        const char* host;
        uint32_t n;
        unsigned char* dev;
        unsigned long long* tile_off;  // the block's newline offsets per tile, counted at upload
        unsigned long long n_newlines;
    };
    std::vector<ResidentText> resident;
    const unsigned char* dt_text = nullptr;  // the text of the block scanned last
    int dt_fmt = 0;   // WK_FMT_* of the blocks (wk_dtok_format)
    uint32_t dt_n = 0, dt_lines = 0, dt_dict_mask = 0;
    int32_t dt_dict_names = -1;  // names of the tokenizer the mirror holds
    const wk_tok* dt_dict_tok = nullptr;
    bool dt_ready = false;
    // the text of a block is copied on a stream of its own into one of two
    // buffers while the kernels work on the other (wk_dtok_copy)
    hipStream_t copy_stream = nullptr, count_stream = nullptr;
    hipEvent_t copy_ev[kTextBufs] = {};
    const char* copy_src[kTextBufs] = {};
    DevBuf d_tiles_k[kTextBufs], d_tile_off_k[kTextBufs];          // newlines per tile of a block copied ahead, counted behind its copy
    unsigned long long* copy_newlines = nullptr;   // [kTextBufs] pinned: their totals
    bool copy_counted[kTextBufs] = {};
    uint32_t copy_n[kTextBufs] = {};
    int dt_cur = 0;
    // blocks are scanned in the order they were copied: a copy's number tells two buffers with the same tag apart
    unsigned long long copy_seq[kTextBufs] = {}, copy_seq_next = 1;
    // a block copied by wk_dtok_copy_ahead: the host bytes are not kept until the scan (the reader reuses its pinned
    // buffer once the copy is through); what the scan wants from them comes back from the device when it does
    bool copy_detached[kTextBufs] = {};
    char copy_last[kTextBufs] = {};   // the block's last byte (a last line without newline)
    bool dt_detached = false;         // the block scanned last: its host bytes are gone
    // (wk_dtok_expect) bytes of text the open sample will have brought in all; the records that makes, told from
    // the first block's lines per byte: the sample's record buffers are sized once instead of doubled eight times
    int64_t dt_expect_bytes = 0, w_expect = 0;
    // Small results of a block (its DtokState, totals) come back through pinned memory that a one-wave kernel
    // writes (small_back): a hipMemcpyAsync of a few bytes queues on the DMA engine behind the 64 MB text copy
    // that is under way there, and the scan loop then runs at the pace of the link instead of the kernels'
    unsigned char* host_back = nullptr;   // [kBackSlots][kBackBytes] pinned, then copy_newlines[kTextBufs]
    static constexpr int kBackSlots = 4, kBackBytes = 256;
    // A text buffer is free, holds a block copied ahead (tagged by copy_src / copy_n), or is the one the
    // kernels of the block scanned last read (until the next scan begins).  wk_dtok_copy may be called
    // from another thread than the scans (the host layer's reader thread issues the copies as soon as a
    // block is cut): the states and tags are guarded by copy_mu.
    enum : unsigned char { kBufFree = 0, kBufCopied = 1, kBufScanning = 2, kBufPending = 3 };
    unsigned char buf_state[kTextBufs] = {};
    std::mutex copy_mu;
    std::mutex slab_mu;   // a slab is allocated by whoever needs one of its buffers first
    // read maps formatted on the device (wk_readmap.hpp): per job the taxon slot of every subject, the slots' order
    // and shown text; per block the reads' leader lines, line lengths / offsets and the text itself
    struct MapTables {
        DevBuf slot_of_subject, slot_order, shown_off, shown;
        int32_t n_subjects = 0, n_slots = 0;
    };
    MapTables rm[WK_MAX_JOBS];
    DevBuf rm_line, rm_len, rm_text;
    // strata map of the current sample on the device (wk_strata.hpp)
    DevBuf s_text, s_lines, s_tab, s_vlen, s_hash, s_label, s_slots, s_labels, s_label_text, s_label_group, s_state, s_tiles, s_tile_off;
    uint32_t s_n = 0, s_n_lines = 0, s_mask = 0;
    bool s_ready = false;   // wk_strata_load built the tables
    bool s_use = false;     // wk_strata_groups gave the labels their groups: wk_dtok_stage_hits joins
    bool dt_keep_reads = false;   // wk_dtok_keep_reads: wk_dtok_emit places the records in read order and keeps the block
    bool dt_emitted = false;      // the block scanned last was emitted with its reads kept
    uint32_t dt_emit_reads = 0;
    int64_t rm_bytes = 0;         // text of the last wk_dtok_readmap
    int range_parts_opt = 0;   // partitions of the dense gene log (0: as few as the merge's LDS array allows; measurement)
    int use_streams = 1;   // (0: the records of all slices in one stream, round 3's team kernel; measurement)
    int use_subject_bins = 1;
    int use_hot_bins = 1;  // hot-subject bins for subject tables beyond the LDS  // count-first mode of the split for small subject tables
};

namespace {

int fail(wk_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
        g_last_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                       \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return fail((ctx), WK_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),  \
                        __FILE__, __LINE__);                                                     \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
};

constexpr int kStatBlocks = 16384;  // >= the largest classify grid (256 CUs x 32 + slack)

unsigned long long* scalar_u64(wk_ctx* c, int idx) { return c->scalars.as<unsigned long long>() + idx; }
int* scalar_err(wk_ctx* c) { return c->scalars.as<int>(); }

// Device memory for text buffer k, at least `need` bytes.
hipError_t text_buffer(wk_ctx* c, int k, size_t need) {
    if (need <= wk_ctx::kTextStride) {
        DevBuf& slab = c->d_textslab[k / wk_ctx::kSlabBufs];
        std::lock_guard<std::mutex> lock(c->slab_mu);
        if (!slab.p) {
            const hipError_t e = slab.reserve(wk_ctx::kTextStride * wk_ctx::kSlabBufs);
            if (e != hipSuccess) return e;
        }
        c->d_textptr[k] = slab.as<unsigned char>() + (size_t)(k % wk_ctx::kSlabBufs) * wk_ctx::kTextStride;
        return hipSuccess;
    }
    const hipError_t e = c->d_textbuf[k].reserve(need);
    if (e == hipSuccess) c->d_textptr[k] = c->d_textbuf[k].as<unsigned char>();
    return e;
}

// `bytes` (<= 256, a multiple of 4) from device memory to pinned host memory, by the device itself
__global__ void small_back_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ host, uint32_t words) {
    if (threadIdx.x < words) host[threadIdx.x] = src[threadIdx.x];
}

// (one word to device memory without a trip through the DMA queue)
__global__ void store_u32_kernel(uint32_t* dst, uint32_t value) { *dst = value; }

// Queue the copy of a small result into slot `slot` of the pinned scratch; read it with small_back_get once the
// stream has been waited for.
hipError_t small_back(wk_ctx* c, int slot, const void* dev, size_t bytes) {
    if (!c->host_back || slot < 0 || slot >= wk_ctx::kBackSlots || bytes > (size_t)wk_ctx::kBackBytes || (bytes & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(small_back_kernel, dim3(1), dim3(64), 0, c->stream, static_cast<const uint32_t*>(dev),
                       reinterpret_cast<uint32_t*>(c->host_back + (size_t)slot * wk_ctx::kBackBytes), (uint32_t)(bytes / 4));
    return hipGetLastError();
}
void small_back_get(wk_ctx* c, int slot, void* out, size_t bytes) {
    std::memcpy(out, c->host_back + (size_t)slot * wk_ctx::kBackBytes, bytes);
}

struct Lap {
    double* acc;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit Lap(double* a) : acc(a) {}
    ~Lap() { *acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

// start of an API call that times its kernels: a fresh head of the chain
void ktimer_step(wk_ctx* c) {
    if (!c->profile) return;
    if (!c->kt_step && hipEventCreate(&c->kt_step) != hipSuccess) {
        c->kt_step = nullptr;
        return;
    }
    (void)hipEventRecord(c->kt_step, c->stream);
    c->kt_tail = c->kt_step;
}
// (API calls nest — wk_ordinal_count runs wk_classify_staged —: the outermost one heads the chain)
struct KtScope {
    wk_ctx* c;
    explicit KtScope(wk_ctx* ctx) : c(ctx) {
        if (c->kt_depth++ == 0) ktimer_step(c);
    }
    ~KtScope() { --c->kt_depth; }
};
KernelTimer* ktimer_begin(wk_ctx* c, const char* family) {
    if (!c->profile) return nullptr;
    if (!c->kt_tail) ktimer_step(c);
    if (!c->kt_tail) return nullptr;
    KernelTimer& t = c->ktimers[family];
    if (!t.b && hipEventCreate(&t.b) != hipSuccess) {
        t.b = nullptr;
        return nullptr;
    }
    t.valid = false;
    // (a family launched twice in a call: its second bracket would start at its own event, which is about to be recorded again)
    t.from = c->kt_tail == t.b ? c->kt_step : c->kt_tail;
    return &t;
}
void ktimer_end(wk_ctx* c, KernelTimer* t) {
    if (!t) return;
    (void)hipEventRecord(t->b, c->stream);
    c->kt_tail = t->b;
    t->valid = true;
}

int check_device_errors(wk_ctx* c) {
    int e = 0;
    HIP_TRY(c, hipMemcpyAsync(&e, scalar_err(c), sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (e == 0) return WK_OK;
    HIP_TRY(c, hipMemsetAsync(scalar_err(c), 0, sizeof(int), c->stream));
    if (e & kErrTableFull) return fail(c, WK_E_TABLE_FULL, "count table is full (%llu slots); call wk_counts_reserve with more slots", (unsigned long long)c->slots);
    if (e & kErrKRange) return fail(c, WK_E_RANGE, "a read has more than %d candidate features", WK_MAX_K);
    if (e & kErrFeatureRange) return fail(c, WK_E_RANGE, "feature id outside [0, %d]", WK_MAX_FEATURE);
    if (e & kErrGroupRange) return fail(c, WK_E_RANGE, "group id outside [0, %d)", 1 << WK_KEY_GROUP_BITS);
    if (e & kErrPairOverflow) return fail(c, WK_E_RANGE, "more than 2^31 read-gene pairs in one chunk");
    return fail(c, WK_E_HIP, "unknown device error flag %d", e);
}

int upload(wk_ctx* c, DevBuf& b, const void* src, size_t bytes) {
    HIP_TRY(c, b.reserve(bytes ? bytes : 1));
    if (bytes) HIP_TRY(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    return WK_OK;
}

// reads per thread and round in the per-read first pass (classify_single_kernel)
constexpr int kPerReadItems = 2;

int grid_for(int64_t n, int threads, int max_blocks) {
    int64_t b = (n + threads - 1) / threads;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

__global__ void __launch_bounds__(256) table_count_kernel(const unsigned long long* __restrict__ keys,
                                                          uint64_t slots, unsigned long long* __restrict__ n) {
    unsigned long long mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (uint64_t)gridDim.x * blockDim.x)
        mine += (keys[i] != kEmptyKey) ? 1ull : 0ull;
    mine = wave_sum(mine);
    if ((threadIdx.x & (kWave - 1)) == 0 && mine) atomicAdd(n, mine);
}

__global__ void __launch_bounds__(256) table_compact_kernel(const unsigned long long* __restrict__ keys,
                                                            const unsigned long long* __restrict__ vals,
                                                            uint64_t slots, unsigned long long* __restrict__ cursor,
                                                            unsigned long long* __restrict__ out_k,
                                                            long long* __restrict__ out_v) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (slots + stride - 1) / stride;
    const int lane = threadIdx.x & (kWave - 1);
    for (uint64_t it = 0; it < rounds; ++it) {  // wave-uniform trip count: ballots see all lanes
        const uint64_t i = it * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const unsigned long long k = (i < slots) ? keys[i] : kEmptyKey;
        const bool live = (k != kEmptyKey);
        const unsigned long long m = __ballot(live);
        if (m == 0) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(cursor, (unsigned long long)__popcll(m));
        base = __shfl(base, 0, kWave);
        if (live) {
            const unsigned long long pos = base + __popcll(m & ((1ull << lane) - 1ull));
            out_k[pos] = k;
            out_v[pos] = (long long)vals[i];
        }
    }
}

}  // namespace

// tree.find_rank (tree.py:467-510) for the subjects, on the host: ancestor (or
// self) of each subject's node with the rank code of `slot`, -1 if none — what
// rank_table_kernel holds for all nodes.  Kept up to date as subjects are added.
static int ensure_subject_ancestors(wk_ctx* c, wk_ctx::StreamTables& T, int slot) {
    std::vector<int32_t>& t = T.subj_node_host;
    if (T.node_slot != slot || T.node_tree != c->tree_serial || T.node_rank != c->rank_serial) t.clear();
    if ((int32_t)t.size() > c->n_subjects) t.clear();
    const size_t have = t.size();
    if (have == (size_t)c->n_subjects && T.node_slot == slot) return WK_OK;
    const int32_t code = c->rank_tab_code[slot];
    t.resize((size_t)c->n_subjects, -1);
    for (size_t s = have; s < t.size(); ++s) {
        int32_t u = c->subj_feat_host[s], res = -1;
        if (u < c->n_nodes)
            for (;;) {
                if (c->rank_code_host[u] == code) {
                    res = u;
                    break;
                }
                const int32_t p = c->parent_host[u];
                if (p == u) break;
                u = p;
            }
        t[s] = res;
    }
    T.node_slot = slot;
    T.node_tree = c->tree_serial;
    T.node_rank = c->rank_serial;
    const int rc = upload(c, T.subj_node, t.data(), t.size() * 4);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return WK_OK;
}

// The nodes the records of the per-read stream (wk_free.hpp) can name — the
// subjects' nodes (mode 1, `--rank free`) or their ancestors at the job's rank
// (mode 2) — and every subject's place among them.  `before` (optional) receives
// the list as it was when a list for the same job over the same tree is replaced
// (the subject table has grown): records written under it are renumbered.
static int ensure_rank_table(wk_ctx* c, wk_ctx::StreamTables& T, int mode, int slot, std::vector<int32_t>* before) {
    if (mode == 2) {
        const int rc = ensure_subject_ancestors(c, T, slot);
        if (rc) return rc;
    }
    if (T.rk_kind == mode && T.rk_slot == slot && T.rk_tree == c->tree_serial && T.rk_subj == c->subj_serial && T.rk_rank == c->rank_serial)
        return WK_OK;
    const int32_t n_nodes = c->n_nodes;
    const std::vector<int32_t>& of = mode == 2 ? T.subj_node_host : c->subj_feat_host;
    std::vector<int32_t> dn;  // distinct nodes, ascending
    dn.reserve(of.size());
    for (int32_t v : of)
        if (v >= 0 && v < n_nodes) dn.push_back(v);
    std::sort(dn.begin(), dn.end());
    dn.erase(std::unique(dn.begin(), dn.end()), dn.end());
    std::vector<int32_t> rank_of(of.size(), -1);
    for (size_t s = 0; s < of.size(); ++s)
        if (of[s] >= 0 && of[s] < n_nodes) rank_of[s] = (int32_t)(std::lower_bound(dn.begin(), dn.end(), of[s]) - dn.begin());
    const int rc = upload(c, T.subj_rank, rank_of.data(), rank_of.size() * 4);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // (the vector is about to go out of scope)
    const bool same_job = T.rk_kind == mode && T.rk_slot == slot && T.rk_tree == c->tree_serial && T.rk_rank == c->rank_serial;
    if (before) {
        before->clear();
        if (same_job) before->swap(T.dn_host);
    }
    T.dn_host.swap(dn);
    T.rk_kind = mode;
    T.rk_slot = slot;
    T.rk_tree = c->tree_serial;
    T.rk_subj = c->subj_serial;
    T.rk_rank = c->rank_serial;
    return WK_OK;
}

// Tables of the per-read stream over those nodes: the parents, the table of neighbours' LCAs.
static int ensure_stream_tables(wk_ctx* c, wk_ctx::StreamTables& T, int mode, int slot) {
    {
        const int rc = ensure_rank_table(c, T, mode, slot, nullptr);
        if (rc) return rc;
    }
    if (T.kind == mode && T.slot == slot && T.tree == c->tree_serial && T.subj == c->subj_serial && T.rank == c->rank_serial)
        return WK_OK;
    const int32_t n_nodes = c->n_nodes;
    const std::vector<int32_t>& dn = T.dn_host;
    const uint32_t md = (uint32_t)dn.size();
    uint32_t dlevels = 1;
    while (md > 1 && (2u << (dlevels - 1)) <= md - 1) dlevels += 1;
    const size_t drow = std::max<uint32_t>(md, 1u);
    // ... in terms of *result ids*: the nodes a read can be assigned to are the
    // subject nodes and their ancestors (a few hundred thousand of a 2 M-node
    // tree), numbered in pre-order — so that the smallest id is still the
    // shallowest ancestor, and the root is 0 — to keep the stream's counters
    // (one per possible result) small enough to stay in the L2s
    std::vector<int32_t> rid((size_t)std::max(n_nodes, 1), -1);
    for (int32_t v : dn)
        for (int32_t u = v; rid[(size_t)u] < 0; u = c->parent_host[u]) {
            rid[(size_t)u] = 0;
            if (c->parent_host[u] == u) break;
        }
    std::vector<int32_t> rnode;
    for (int32_t v = 0; v < n_nodes; ++v)
        if (rid[(size_t)v] == 0) {
            rid[(size_t)v] = (int32_t)rnode.size();
            rnode.push_back(v);
        }
    if (rnode.empty()) rnode.push_back(0);
    std::vector<int32_t> dsparse(drow * dlevels, 0x7FFFFFFF), dparent(drow, -1), dself(drow, -1);
    for (uint32_t i = 0; i < md; ++i) {
        dparent[i] = rid[(size_t)c->parent_host[dn[i]]];
        dself[i] = rid[(size_t)dn[i]];
    }
    for (uint32_t i = 0; i + 1 < md; ++i) {
        int32_t u = dn[i];
        while (c->last_host[u] < dn[i + 1]) u = c->parent_host[u];
        dsparse[i] = rid[(size_t)u];
    }
    for (uint32_t k = 1; k < dlevels; ++k)
        for (uint32_t i = 0; i + (1u << k) <= md - 1; ++i)
            dsparse[k * drow + i] = std::min(dsparse[(k - 1) * drow + i], dsparse[(k - 1) * drow + i + (1u << (k - 1))]);
    int rc;
    if ((rc = upload(c, T.dsparse, dsparse.data(), dsparse.size() * 4))) return rc;
    if ((rc = upload(c, T.dparent, dparent.data(), dparent.size() * 4))) return rc;
    if ((rc = upload(c, T.dself, dself.data(), dself.size() * 4))) return rc;
    if ((rc = upload(c, T.rnode, rnode.data(), rnode.size() * 4))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // the vectors are about to go out of scope
    T.results = (uint32_t)rnode.size();
    T.dm = (uint32_t)drow;
    T.kind = mode;
    T.slot = slot;
    T.tree = c->tree_serial;
    T.subj = c->subj_serial;
    T.rank = c->rank_serial;
    return WK_OK;
}

// Tables behind ClassifyArgs::free_sparse for the current tree and subject
// table (host side: 100 k subjects x ~17 levels).
static int ensure_free_tables(wk_ctx* c) {
    if (c->free_tree == c->tree_serial && c->free_subj == c->subj_serial) return WK_OK;
    const int32_t n = c->n_subjects, n_nodes = c->n_nodes;
    const std::vector<int32_t>& feat = c->subj_feat_host;
    std::vector<int32_t> order;
    order.reserve((size_t)n);
    for (int32_t s = 0; s < n; ++s)
        if (feat[s] < n_nodes) order.push_back(s);
    std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return feat[x] < feat[y] || (feat[x] == feat[y] && x < y); });
    const uint32_t m = (uint32_t)order.size();
    std::vector<int32_t> rank((size_t)std::max(n, 1), -1);
    for (uint32_t i = 0; i < m; ++i) rank[order[i]] = (int32_t)i;
    uint32_t levels = 1;
    while (m > 1 && (2u << (levels - 1)) <= m - 1) levels += 1;  // the longest range of pairs has m - 1 of them
    const size_t row = std::max<uint32_t>(m, 1u);
    std::vector<int32_t> sparse(row * levels, 0x7FFFFFFF);
    for (uint32_t i = 0; i + 1 < m; ++i) {  // LCA of neighbours: the lowest ancestor of the first whose subtree holds the second
        int32_t u = feat[order[i]];
        const int32_t hi = feat[order[i + 1]];
        while (c->last_host[u] < hi) u = c->parent_host[u];
        sparse[i] = u;
    }
    for (uint32_t k = 1; k < levels; ++k)
        for (uint32_t i = 0; i + (1u << k) <= m - 1; ++i)
            sparse[k * row + i] = std::min(sparse[(k - 1) * row + i], sparse[(k - 1) * row + i + (1u << (k - 1))]);
    int rc;
    if ((rc = upload(c, c->f_rank, rank.data(), rank.size() * 4))) return rc;
    if ((rc = upload(c, c->f_sparse, sparse.data(), sparse.size() * 4))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // the vectors are about to go out of scope
    c->f_m = (uint32_t)row;
    c->free_tree = c->tree_serial;
    c->free_subj = c->subj_serial;
    return WK_OK;
}

// Read size per record of the staged chunk (wk_weigh.hpp): marks at the read
// starts, then spread over the records.  `check`: leave out reads that name a
// subject with its validity bit set.
static int derive_read_sizes(wk_ctx* c, bool check, const int32_t* qoff, const int32_t* subj, int64_t n_reads, int64_t n_rec,
                             unsigned char* rk, unsigned long long* left, unsigned long long* totals) {
    HIP_TRY(c, hipMemsetAsync(totals, 0, 16, c->stream));
    HIP_TRY(c, hipMemsetAsync(rk + n_rec, 0, 8, c->stream));  // the sizes are read four at a time
    const int64_t n_tiles = (n_reads + kSizeTile - 1) / kSizeTile;
    const dim3 grid((unsigned)std::min<int64_t>(n_tiles, (int64_t)c->prop.multiProcessorCount * 2));
    if (check)
        hipLaunchKernelGGL(read_sizes_kernel<true>, grid, dim3(kSizeTile), 0, c->stream, qoff, (uint32_t)n_reads, subj,
                           c->w_invalid.as<uint32_t>(), (uint32_t)c->n_subjects, rk, left, totals);
    else
        hipLaunchKernelGGL(read_sizes_kernel<false>, grid, dim3(kSizeTile), 0, c->stream, qoff, (uint32_t)n_reads, subj,
                           (const uint32_t*)nullptr, 0u, rk, left, totals);
    HIP_TRY(c, hipGetLastError());
    return WK_OK;
}

// Subject rows {feature, ancestor at rank column 0, 1, ...} for a job set over
// the current subject table (rebuilt when the subjects or the rank tables
// changed), the jobs' columns, and which subjects the weighted histogram cannot
// take (c->rows_any_invalid).
static int build_subject_rows(wk_ctx* c, const wk_job* jobs, int32_t n_jobs, ClassifyArgs& a) {
    // (re)build the compact subject rows for this set of rank tables
    if (c->n_subjects <= 0 && c->n_records > 0) return fail(c, WK_E_STATE, "chunk carries subject indices but no subject table is set (wk_set_subjects)");
    std::vector<int> sig;
    RowCols cols{};
    for (int j = 0; j < n_jobs; ++j) {
        if (jobs[j].mode != WK_MODE_RANK) continue;
        int col = -1;
        for (int q = 0; q < cols.n_cols; ++q)
            if (sig[q] == jobs[j].rank_slot) col = q;
        if (col < 0) {
            col = cols.n_cols++;
            sig.push_back(jobs[j].rank_slot);
            cols.anc[col] = c->rank_tab[jobs[j].rank_slot].as<int32_t>();
        }
        a.jobs[j].col = col;
    }
    // `--rank free`: one more column with the subject's rank among the
    // subjects of the tree, when it fits the 4-column row
    bool any_free = false;
    for (int j = 0; j < n_jobs; ++j) any_free |= jobs[j].mode == WK_MODE_FREE;
    cols.by_subject = -1;
    int free_col = -1;
    // (the table has n log n entries: subject tables beyond 4 M keep the walk)
    if (any_free && c->use_free_sparse && cols.n_cols < 3 && c->n_subjects > 0 && c->n_subjects <= (1 << 22) && c->n_nodes > 0) {
        int rc2 = ensure_free_tables(c);
        if (rc2) return rc2;
        free_col = cols.n_cols++;
        sig.push_back(-2);
        sig.push_back(c->free_tree);
        sig.push_back(c->free_subj);
        cols.anc[free_col] = nullptr;
        cols.by_subject = free_col;
        cols.subject_col = c->f_rank.as<int32_t>();
        a.free_sparse = c->f_sparse.as<int32_t>();
        a.free_m = (int32_t)c->f_m;
        a.free_col = free_col;
    }
    int w = 4;  // {feature, <= 3 rank columns}: the kernel's single-pass fast path
    while (w < 1 + cols.n_cols) w <<= 1;
    sig.push_back(-1);
    sig.push_back(c->n_subjects);
    if (sig != c->rows_sig || w != c->rows_w) {
        HIP_TRY(c, c->subj_rows.reserve((size_t)std::max(c->n_subjects, 1) * w * sizeof(int32_t)));
        if (c->n_subjects > 0) {
            hipLaunchKernelGGL(subject_rows_kernel, dim3((c->n_subjects + 255) / 256), dim3(256), 0, c->stream,
                               c->subj_feat.as<int32_t>(), c->n_subjects, c->n_nodes, cols, w,
                               c->subj_rows.as<int32_t>());
            HIP_TRY(c, hipGetLastError());
        }
        c->rows_sig = sig;
        c->rows_w = w;
        // which subjects the weighted histogram cannot take (wk_weigh.hpp)
        c->rows_any_invalid = false;
        c->rows_serial += 1;
        if (c->n_subjects > 0) {
            HIP_TRY(c, c->w_invalid.reserve(((size_t)c->n_subjects / 32 + 2) * 4));
            HIP_TRY(c, hipMemsetAsync(scalar_u64(c, 6), 0, 8, c->stream));
            hipLaunchKernelGGL(subject_invalid_kernel, dim3((c->n_subjects + 255) / 256), dim3(256), 0, c->stream,
                               c->subj_rows.as<int32_t>(), w, cols.n_cols, c->n_subjects,
                               c->w_invalid.as<uint32_t>(), reinterpret_cast<uint32_t*>(scalar_u64(c, 6)));
            HIP_TRY(c, hipGetLastError());
            uint32_t any = 0;
            HIP_TRY(c, hipMemcpyAsync(&any, scalar_u64(c, 6), 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            c->rows_any_invalid = any != 0;
        }
    }
    a.rows = c->subj_rows.as<int32_t>();
    a.row_w = w;
    a.n_subjects = c->n_subjects;
    return WK_OK;
}

extern "C" {

static int refresh_word_ranks(wk_ctx* c, int64_t count);
static int streams_needed(const wk_ctx* c) { return std::max(1, (int)(((int64_t)c->n_subjects + kSliceBins - 1) / kSliceBins)); }

// the accumulation is empty again
static int words_reset(wk_ctx* c) {
    c->w_records = c->w_reads = 0;
    c->w_expect = 0;
    c->w_open = false;
    c->w_counts_known = false;
    c->fz_chain = false;
    if (c->w_cursor.p) HIP_TRY(c, hipMemsetAsync(c->w_cursor.p, 0, kMaxStreams * 8, c->stream));
    return WK_OK;
}

int wk_words_flush(wk_ctx* c);
int wk_words_begin(wk_ctx* c, const wk_job* jobs, int32_t n_jobs, int32_t group, int* ok);

int wk_abi_version(void) { return WK_ABI_VERSION; }

#ifndef WK_BUILD_ID
#define WK_BUILD_ID "unknown"
#endif
// (the tag is also what the build recipe looks for in the file's bytes)
static const char kBuildTag[] = "WK_BUILD_ID=" WK_BUILD_ID;
const char* wk_build_id(void) { return kBuildTag + 12; }

int wk_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

const char* wk_last_error(const wk_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int wk_device_pci_bus_id(int device, char* buf, size_t cap) {
    if (!buf || cap < 16) return fail(nullptr, WK_E_ARG, "bad arguments");
    const hipError_t e = hipDeviceGetPCIBusId(buf, (int)cap, device);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(nullptr, WK_E_HIP, "hipDeviceGetPCIBusId(%d) failed: %s", device, hipGetErrorString(e));
    }
    return WK_OK;
}

int wk_create(int device, wk_ctx** out) {
    if (!out) return fail(nullptr, WK_E_ARG, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, WK_E_HIP, "no HIP device available (%s); libwoltka_hip has no CPU path",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= ndev) return fail(nullptr, WK_E_ARG, "device %d out of range [0,%d)", device, ndev);
    wk_ctx* c = new (std::nothrow) wk_ctx();
    if (!c) return fail(nullptr, WK_E_HIP, "out of host memory");
    c->device = device;
    DeviceGuard guard(device);
    if ((e = hipGetDeviceProperties(&c->prop, device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        // (the text route's two streams with it: made by the first wk_dtok_copy they cost the reader 40 ms of a call's
        // first 100 -- it runs beside the hierarchy's and the pread pool's threads by then --, here 6)
        (e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->count_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&c->t0)) != hipSuccess || (e = hipEventCreate(&c->t1)) != hipSuccess ||
        (e = hipHostMalloc(reinterpret_cast<void**>(&c->host_back),
                           (size_t)wk_ctx::kBackSlots * wk_ctx::kBackBytes + (size_t)wk_ctx::kTextBufs * 8,
                           hipHostMallocDefault)) != hipSuccess ||
        (e = c->scalars.reserve(128)) != hipSuccess ||
        (e = hipMemsetAsync(c->scalars.p, 0, 128, c->stream)) != hipSuccess ||
        (e = c->stat_block.reserve((size_t)kStatBlocks * 16)) != hipSuccess ||
        (e = hipMemsetAsync(c->stat_block.p, 0, (size_t)kStatBlocks * 16, c->stream)) != hipSuccess) {
        int rc = fail(nullptr, WK_E_HIP, "context setup failed: %s", hipGetErrorString(e));
        wk_destroy(c);
        return rc;
    }
    c->copy_newlines = reinterpret_cast<unsigned long long*>(c->host_back + (size_t)wk_ctx::kBackSlots * wk_ctx::kBackBytes);
    // the LDS front cache needs more than the default 64 KiB dynamic LDS limit
    // (160 KiB per CU minus the kernels' few bytes of static LDS)
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stripe_match_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStripeMatchLds)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&free_stream_kernel<false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&free_stream_kernel<true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&free_stream_kernel<false, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&free_stream_kernel<true, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&free_log_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLogBins * 4))) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&classify_kernel<true, false, 0>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&classify_kernel<true, false, 1>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&classify_kernel<true, false, 2>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&classify_kernel<true, true, 0>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&classify_kernel<true, true, 1>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&classify_kernel<true, true, 2>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&classify_single_kernel<true, true, 4>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&count_subjects_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&classify_single_kernel<false, true, kPerReadItems>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&classify_single_kernel<false, false, kPerReadItems>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&classify_single_kernel<true, true, 2, false, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&classify_single_kernel<true, false, 2, false, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&classify_single_kernel<false, true, kPerReadItems, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&classify_single_kernel<false, false, kPerReadItems, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&weigh_bins_kernel<4>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBinsMaxLds)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&weigh_bins_kernel<4, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBinsMaxLds)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&weigh_bins_kernel<6, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBinsMaxLds)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&weigh_bins_kernel<8, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBinsMaxLds)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&weigh_bins_kernel<3>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBinsMaxLds)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&weigh_bins_kernel<6>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBinsMaxLds)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&weigh_bins_kernel<8>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBinsMaxLds)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&weigh_streams_kernel<4>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBinsMaxLds)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&weigh_streams_kernel<6>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBinsMaxLds)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&weigh_streams_kernel<8>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBinsMaxLds)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&weigh_merge_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&classify_tiled_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&match_hits_kernel<true, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&match_hits_kernel<true, false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&range_merge_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(&partition_merge_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)) != hipSuccess) {
        int rc = fail(nullptr, WK_E_HIP, "hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e));
        wk_destroy(c);
        return rc;
    }
    // (measurement: WOLTKA_NO_FUSED=1 keeps every block of the text route on the six kernels of wk_dtok.hpp, for
    // whole `woltka classify` calls that cannot reach wk_tune)
    if (const char* nf = getenv("WOLTKA_NO_FUSED")) c->use_fused = (nf[0] && nf[0] != '0') ? 0 : 1;
    if (const char* nc = getenv("WOLTKA_FZ_NO_CHAIN")) c->fz_no_chain = nc[0] && nc[0] != '0';
    if (const char* nl = getenv("WOLTKA_NO_LAG")) c->lag_enabled = !(nl[0] && nl[0] != '0');
    if (const char* lp = getenv("WOLTKA_LAG_POLL")) c->lag_poll = lp[0] && lp[0] != '0';
    if (const char* sm = getenv("WOLTKA_STRIPES_MIN"))
        if (sm[0]) c->stripes_min_hits = std::max<long long>(0, atoll(sm));
    *out = c;
    return WK_OK;
}

void wk_destroy(wk_ctx* c) {
    if (!c) return;
    DeviceGuard guard(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    DevBuf* bufs[] = {&c->nodes, &c->rank_code, &c->gene4, &c->g_grid, &c->g_first, &c->g_goff, &c->g_shift,
                      &c->tkeys, &c->tvals, &c->c_subj, &c->c_qoff, &c->c_group, &c->o_genome, &c->o_beg,
                      &c->o_end, &c->o_len, &c->o_hoff, &c->o_cnt, &c->o_ub, &c->o_first2, &c->o_poff, &c->o_pairs, &c->o_qoff,
                      &c->o_tile_sum, &c->o_tile_off, &c->scalars, &c->stat_block, &c->log, &c->subj_feat, &c->subj_rows, &c->dense_slab, &c->plog, &c->plog_cnt, &c->left_mask, &c->left_list, &c->first_slab, &c->w_slab, &c->w_hi, &c->w_invalid, &c->c_rk[0], &c->c_rk[1], &c->rk_left[0], &c->rk_left[1], &c->rk_totals, &c->f_rank, &c->f_sparse, &c->f_dense, &c->w_renum, &c->f_log, &c->f_log_cnt, &c->f_partial, &c->f_part_used, &c->w_tmp, &c->assign_out, &c->fetch_k, &c->fetch_v};
    for (DevBuf* b : bufs) b->release();
    for (wk_ctx::StreamTables& T : c->st)
        for (DevBuf* b : {&T.dsparse, &T.dparent, &T.dself, &T.rnode, &T.subj_node, &T.subj_rank}) b->release();
    c->c_words.release();
    for (DevBuf& b : c->w_stream) b.release();
    c->w_backup2.release();
    c->w_backup3.release();
    for (hipEvent_t& e : c->lag_ev)
        if (e) (void)hipEventDestroy(e);
    for (int q = 0; q < wk_ctx::kTextBufs; ++q) {
        c->d_tiles_k[q].release();
        c->d_tile_off_k[q].release();
        c->d_textbuf[q].release();
        if (q % wk_ctx::kSlabBufs == 0) c->d_textslab[q / wk_ctx::kSlabBufs].release();
    }
    for (DevBuf* b : {&c->d_tiles, &c->d_tile_off, &c->d_lines, &c->d_lsubj, &c->d_lmeta, &c->d_start, &c->d_first, &c->d_unknown, &c->d_lbeg, &c->d_lend, &c->d_llen, &c->d_lscan, &c->d_gmap,
                      &c->d_state, &c->d_dict, &c->d_dict2, &c->d_names16, &c->d_arena, &c->d_submap})
        b->release();
    for (wk_ctx::ResidentText& r : c->resident) {
        (void)hipFree(r.dev);
        (void)hipFree(r.tile_off);
    }
    c->resident.clear();
    for (void* hp : c->host_blocks) (void)hipHostFree(hp);
    if (c->host_back) (void)hipHostFree(c->host_back);
    for (const wk_ctx::HostReg& r : c->regs) (void)hipHostUnregister(const_cast<char*>(r.p));
    c->regs.clear();
    for (hipEvent_t ev : c->copy_ev)
        if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : c->copy_ev0)
        if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : c->copy_evm)
        if (ev) (void)hipEventDestroy(ev);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->count_stream) (void)hipStreamDestroy(c->count_stream);
    if (c->res_ev) (void)hipEventDestroy(c->res_ev);
    for (hipEvent_t ev : c->slot_ev)
        if (ev) (void)hipEventDestroy(ev);
    for (DevBuf& b : c->rank_tab) b.release();
    for (auto& kv : c->ktimers)
        if (kv.second.b) (void)hipEventDestroy(kv.second.b);
    if (c->kt_step) (void)hipEventDestroy(c->kt_step);
    if (c->t0) (void)hipEventDestroy(c->t0);
    if (c->t1) (void)hipEventDestroy(c->t1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int wk_device_name(const wk_ctx* c, char* buf, size_t cap) {
    if (!c || !buf || cap == 0) return WK_E_ARG;
    snprintf(buf, cap, "%s (%s, %d CUs)", c->prop.name, c->prop.gcnArchName, c->prop.multiProcessorCount);
    return WK_OK;
}

int wk_sync(wk_ctx* c) {
    if (!c) return WK_E_ARG;
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->count_stream) HIP_TRY(c, hipStreamSynchronize(c->count_stream));  // (newline counts behind copies / resident blocks)
    return WK_OK;
}

int wk_set_option(wk_ctx* c, const char* name, int64_t value) {
    if (!c || !name) return WK_E_ARG;
    if (!strcmp(name, "gene_index_pairs")) {  // the next wk_set_genes: gene lists by gene table index (wk_ordinal_pair_genes)
        c->gene_index_opt = value != 0;
        return WK_OK;
    }
    if (!strcmp(name, "dtok_count_ahead")) {  // 0: no newline count behind the copies of blocks (wk_dtok_copy*): the scans count the blocks that need it
        c->count_ahead = value != 0;
        return WK_OK;
    }
    return fail(c, WK_E_ARG, "unknown option '%s' (launch shapes and ablation switches: wk_tune)", name);
}

// Launch shapes, ablation switches and what bench.py needs to time repeated passes:
// measurement only — nothing in woltka_amd/ calls it, results never depend on it.
int wk_tune(wk_ctx* c, const char* name, int64_t value) {
    if (!c || !name) return WK_E_ARG;
    if (!strcmp(name, "lds_slots")) {
        if (value < 64 || value > 8192 || (value & (value - 1))) return fail(c, WK_E_ARG, "lds_slots must be a power of two in [64, 8192]");
        c->lds_slots = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "blocks_per_cu")) {
        if (value < 1 || value > 32) return fail(c, WK_E_ARG, "blocks_per_cu must be in [1, 32]");
        c->blocks_per_cu = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "threads")) {
        if (value < 64 || value > 1024 || (value % 64)) return fail(c, WK_E_ARG, "threads must be a multiple of 64 in [64, 1024]");
        c->threads = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "ablate")) {
        c->ablate = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "free_sparse")) {  // 0: `--rank free` walks up the tree per read
        c->use_free_sparse = (int)value;
        c->rows_sig.clear();
        return WK_OK;
    }
    if (!strcmp(name, "tally_per_cu")) {
        if (value < 1 || value > 4) return fail(c, WK_E_ARG, "tally_per_cu must be in [1, 4]");
        c->tally_per_cu = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "tally_slots")) {
        if (value < 256 || value > 4096 || (value & (value - 1))) return fail(c, WK_E_ARG, "tally_slots must be a power of two in [256, 4096]");
        c->tally_slots = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "dtok_fused")) {
        if (value != 0 && value != 1) return fail(c, WK_E_ARG, "dtok_fused must be 0 or 1");
        c->use_fused = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "fz_ablate")) {
        c->fused_ablate = (uint32_t)value;
        return WK_OK;
    }
    if (!strcmp(name, "dtok_fused_per_cu")) {
        if (value < 1 || value > 8) return fail(c, WK_E_ARG, "dtok_fused_per_cu must be in [1, 8]");
        c->fused_per_cu = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "stripes_min")) {
        if (value < 0) return fail(c, WK_E_ARG, "stripes_min must be >= 0");
        c->stripes_min_hits = value;
        return WK_OK;
    }
    if (!strcmp(name, "stripes")) {
        if (value != 0 && value != 1) return fail(c, WK_E_ARG, "stripes must be 0 or 1");
        c->use_stripes = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "tally")) {
        c->use_tally = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "match_lds")) {
        c->match_lds = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "grid_density")) {  // takes effect at the next wk_set_genes
        if (value < 1 || value > 8) return fail(c, WK_E_ARG, "grid_density must be in [1, 8]");
        c->grid_density = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "plog")) {  // 0 = off, 1 = auto (large chunks), 2 = always
        c->use_plog = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "log_parts")) {  // 0 = auto, 256 or 1024
        if (value != 0 && (value < 64 || value > 1024 || (value & (value - 1)))) return fail(c, WK_E_ARG, "log_parts must be 0 or a power of two in [64, 1024]");
        c->log_parts_opt = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "plog_max_bytes")) {
        if (value < (1 << 20)) return fail(c, WK_E_ARG, "plog_max_bytes must be at least 1 MiB");
        c->plog_max_bytes = value;
        return WK_OK;
    }
    if (!strcmp(name, "dense")) {
        c->use_dense = value ? 1 : 0;
        return WK_OK;
    }
    if (!strcmp(name, "split")) {  // 0 = off, 1 = auto, 2 = always (when the chunk qualifies)
        if (value < 0 || value > 2) return fail(c, WK_E_ARG, "split must be 0, 1 or 2");
        c->use_split = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "hot_bins")) {
        c->use_hot_bins = value ? 1 : 0;
        return WK_OK;
    }
    if (!strcmp(name, "subject_bins")) {
        c->use_subject_bins = value ? 1 : 0;
        return WK_OK;
    }
    if (!strcmp(name, "count_kernel")) {
        c->use_count_kernel = value ? 1 : 0;
        return WK_OK;
    }
    if (!strcmp(name, "single_blocks_per_cu")) {
        if (value < 1 || value > 8) return fail(c, WK_E_ARG, "single_blocks_per_cu must be in [1, 8]");
        c->single_blocks_per_cu = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "lap_print")) {
        fprintf(stderr, "[wk] seconds inside wk_dtok_copy %.3f, wk_dtok_scan(_emit) %.3f of which waiting for the stream %.3f, for the copy %.3f\n", c->lap_s[0],
                c->lap_s[1], c->lap_s[2], c->lap_s[3]);
        fprintf(stderr, "[wk] copy calls, host side: streams + events %.3f, device buffer %.3f (longest %.3f), copy queued %.3f, count queued %.3f\n", c->lap_c[0],
                c->lap_c[1], c->lap_c[4], c->lap_c[2], c->lap_c[3]);
        for (double& x : c->lap_c) x = 0;
        if (c->lap_copy_ms > 0)
            fprintf(stderr, "[wk] the copies themselves: %.1f ms for %.2f GB = %.1f GB/s (events around each copy + newline count on the copy stream)\n",
                    c->lap_copy_ms, (double)c->lap_copy_bytes / 1e9, (double)c->lap_copy_bytes / 1e6 / c->lap_copy_ms);
        fprintf(stderr, "[wk] scan, host side: claim %.3f, line starts queued %.3f, parse queued %.3f, emission queued %.3f, finish %.3f, names %.3f\n",
                c->lap_x[0], c->lap_x[1], c->lap_x[2], c->lap_x[3], c->lap_x[4], c->lap_x[5]);
        for (double& x : c->lap_x) x = 0;
        c->lap_s[0] = c->lap_s[1] = c->lap_s[2] = c->lap_s[3] = 0;
        c->lap_copy_ms = 0;
        c->lap_copy_bytes = 0;
        return WK_OK;
    }
    if (!strcmp(name, "range_parts")) {  // partitions of the dense gene log (a power of two; 0 = auto)
        c->range_parts_opt = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "streams")) {  // 0: the records of all slices in one stream (round 3's team kernel)
        c->use_streams = value != 0;
        return WK_OK;
    }
    if (!strcmp(name, "weigh")) {  // 0 = off, 1 = auto (large multi-hit chunks), 2 = whenever the jobs allow it
        if (value < 0 || value > 2) return fail(c, WK_E_ARG, "weigh must be 0, 1 or 2");
        c->use_weigh = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "free_per_cu")) {
        if (value < 1 || value > 8) return fail(c, WK_E_ARG, "free_per_cu must be in [1, 8]");
        c->free_per_cu = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "range_log")) {
        c->use_range_log = value != 0;
        return WK_OK;
    }
    if (!strcmp(name, "free_threads")) {
        if (value != 256 && value != 512 && value != 1024) return fail(c, WK_E_ARG, "free_threads must be 256, 512 or 1024");
        c->free_threads = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "free_slots")) {
        if (value < 256 || value > 8192 || (value & (value - 1))) return fail(c, WK_E_ARG, "free_slots must be a power of two in [256, 8192]");
        c->free_slots = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "words_keep")) {
        c->words_keep = value ? 1 : 0;
        return WK_OK;
    }
    if (!strcmp(name, "bins_ring")) {
        c->bins_ring = (int)value;
        return WK_OK;
    }
    if (!strcmp(name, "tiled")) {
        c->tiled = value ? 1 : 0;
        return WK_OK;
    }
    if (!strcmp(name, "use_lds")) {
        c->use_lds = value ? 1 : 0;
        return WK_OK;
    }
    return fail(c, WK_E_ARG, "unknown tuning knob '%s'", name);
}

// ---- static state ----------------------------------------------------------

int wk_set_tree(wk_ctx* c, const int32_t* parent, const int32_t* last, const int32_t* rank_code, int32_t n) {
    if (!c) return WK_E_ARG;
    if (n < 0 || (n > 0 && (!parent || !last || !rank_code))) return fail(c, WK_E_ARG, "bad tree arguments");
    if ((int64_t)n > (int64_t)WK_MAX_FEATURE) return fail(c, WK_E_RANGE, "too many hierarchy nodes");
    DeviceGuard guard(c->device);
    {
        const int rcw = wk_words_flush(c);  // (records accumulated under the old tree)
        if (rcw) return rcw;
    }
    for (int32_t v = 0; v < n; ++v) {  // validate the pre-order contract the kernels rely on
        const bool ok = (v == 0) ? (parent[0] == 0) : (parent[v] >= 0 && parent[v] < v);
        if (!ok || last[v] < v || last[v] >= n) return fail(c, WK_E_ARG, "node %d violates the pre-order contract (parent %d, last %d)", v, parent[v], last[v]);
    }
    std::vector<Node> packed((size_t)n);
    for (int32_t v = 0; v < n; ++v) packed[v] = Node{parent[v], last[v]};
    int rc;
    if ((rc = upload(c, c->nodes, packed.data(), (size_t)n * sizeof(Node)))) return rc;
    if ((rc = upload(c, c->rank_code, rank_code, (size_t)n * sizeof(int32_t)))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // `packed` is about to go out of scope
    c->rank_code_host.assign(rank_code, rank_code + n);
    c->parent_host.assign(parent, parent + n);
    c->last_host.assign(last, last + n);
    c->tree_serial += 1;
    c->n_nodes = n;
    c->rows_sig.clear();
    for (bool& v : c->rank_tab_valid) v = false;
    return WK_OK;
}

int wk_build_rank_table(wk_ctx* c, int32_t slot, int32_t code) {
    if (!c) return WK_E_ARG;
    if (slot < 0 || slot >= (int)(sizeof c->rank_tab / sizeof c->rank_tab[0])) return fail(c, WK_E_ARG, "rank slot %d out of range", slot);
    if (c->n_nodes <= 0) return fail(c, WK_E_STATE, "no hierarchy uploaded (wk_set_tree)");
    DeviceGuard guard(c->device);
    HIP_TRY(c, c->rank_tab[slot].reserve((size_t)c->n_nodes * sizeof(int32_t)));
    KernelTimer* kt = ktimer_begin(c, "rank_table");
    hipLaunchKernelGGL(rank_table_kernel, dim3((c->n_nodes + 255) / 256), dim3(256), 0, c->stream,
                       c->nodes.as<Node>(), c->rank_code.as<int32_t>(), c->n_nodes, code,
                       c->rank_tab[slot].as<int32_t>());
    ktimer_end(c, kt);
    HIP_TRY(c, hipGetLastError());
    c->rank_tab_valid[slot] = true;
    c->rank_tab_code[slot] = code;
    c->rank_serial += 1;
    c->rank_tab_nodes[slot] = std::count(c->rank_code_host.begin(), c->rank_code_host.end(), code);
    c->rows_sig.clear();
    return WK_OK;
}

int wk_get_rank_table(wk_ctx* c, int32_t slot, int32_t* out) {
    if (!c || !out) return WK_E_ARG;
    if (slot < 0 || slot >= (int)(sizeof c->rank_tab / sizeof c->rank_tab[0]) || !c->rank_tab_valid[slot])
        return fail(c, WK_E_STATE, "rank slot %d has not been built", slot);
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipMemcpyAsync(out, c->rank_tab[slot].p, (size_t)c->n_nodes * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return WK_OK;
}

int wk_set_genes(wk_ctx* c, const int32_t* genome_off, int32_t n_genomes, const int32_t* start0,
                 const int32_t* end, const int32_t* gene_feature, int32_t n_genes) {
    if (!c) return WK_E_ARG;
    if (n_genomes < 0 || n_genes < 0 || !genome_off || (n_genes > 0 && (!start0 || !end || !gene_feature)))
        return fail(c, WK_E_ARG, "bad gene table arguments");
    if (genome_off[0] != 0 || genome_off[n_genomes] != n_genes) return fail(c, WK_E_ARG, "genome_off must run from 0 to n_genes");
    // gene records {start0, end, largest end before the gene, feature}: the
    // third word stops the backward walk of a hit
    std::vector<int32_t> packed((size_t)n_genes * 4);
    for (int32_t g = 0; g < n_genomes; ++g) {
        if (genome_off[g + 1] < genome_off[g]) return fail(c, WK_E_ARG, "genome_off is not monotone at genome %d", g);
        int32_t m = INT32_MIN;
        for (int32_t i = genome_off[g]; i < genome_off[g + 1]; ++i) {
            if (i > genome_off[g] && start0[i] < start0[i - 1]) return fail(c, WK_E_ARG, "genes of genome %d are not sorted by start", g);
            if (end[i] < start0[i]) return fail(c, WK_E_ARG, "gene %d has end < start", i);
            if (gene_feature[i] < 0 || gene_feature[i] > WK_MAX_FEATURE) return fail(c, WK_E_RANGE, "gene feature id out of range");
            packed[4 * (size_t)i] = start0[i];
            packed[4 * (size_t)i + 1] = end[i];
            packed[4 * (size_t)i + 2] = m;
            packed[4 * (size_t)i + 3] = c->gene_index_opt ? i : gene_feature[i];
            m = end[i] > m ? end[i] : m;
        }
    }
    // grid per genome: `cells` (a power of two, about grid_density per gene)
    // equal ranges of width 2^shift from the smallest start0 on;
    // grid[off + c] = first gene whose cell is >= c, for c = 0 .. cells
    std::vector<int32_t> first((size_t)n_genomes + 1, 0), goff((size_t)n_genomes + 1, 0);
    std::vector<unsigned char> shift((size_t)n_genomes + 1, 0);
    std::vector<int32_t> grid;
    grid.reserve((size_t)n_genes * 2 * (size_t)c->grid_density + (size_t)n_genomes * 2 + 1);
    for (int32_t g = 0; g < n_genomes; ++g) {
        const int32_t lo = genome_off[g], n = genome_off[g + 1] - lo;
        goff[g] = (int32_t)grid.size();
        if (n == 0) continue;  // no cells: no hit of this genome matches
        first[g] = start0[lo];
        const uint64_t span = (uint64_t)((int64_t)start0[lo + n - 1] - (int64_t)start0[lo]);
        uint64_t cells = 1;
        while (cells < (uint64_t)n * (uint64_t)c->grid_density) cells <<= 1;
        int sh = 0;
        while ((span >> sh) >= cells) sh += 1;
        shift[g] = (unsigned char)sh;
        if (grid.size() + cells + 1 >= (1ull << 31)) return fail(c, WK_E_RANGE, "gene tables too large for the coordinate grid");
        int32_t i = lo;
        for (uint64_t cell = 0; cell <= cells; ++cell) {
            while (i < lo + n && (((uint64_t)((int64_t)start0[i] - (int64_t)start0[lo])) >> sh) < cell) i += 1;
            grid.push_back(i);
        }
    }
    goff[n_genomes] = (int32_t)grid.size();
    if (grid.empty()) grid.push_back(0);
    DeviceGuard guard(c->device);
    int rc;
    if ((rc = upload(c, c->gene4, packed.data(), packed.size() * sizeof(int32_t)))) return rc;
    if ((rc = upload(c, c->g_grid, grid.data(), grid.size() * sizeof(int32_t)))) return rc;
    if ((rc = upload(c, c->g_first, first.data(), first.size() * sizeof(int32_t)))) return rc;
    if ((rc = upload(c, c->g_goff, goff.data(), goff.size() * sizeof(int32_t)))) return rc;
    if ((rc = upload(c, c->g_shift, shift.data(), shift.size()))) return rc;
    if ((rc = upload(c, c->g_feature, gene_feature, (size_t)n_genes * 4))) return rc;
    // genome stripes for the sorted coord-match (wk_stripe.hpp): consecutive genomes while their genes, the
    // cells of their grids and their per-genome words fit the LDS; a genome that does not fit alone has no
    // stripe (its hits keep the gather kernels)
    {
        std::vector<int32_t> stripe_of((size_t)std::max(n_genomes, 1), -1);
        c->stripes_host.clear();
        auto genes_of = [&](int32_t g) { return genome_off[g + 1] - genome_off[g]; };
        auto cells_of = [&](int32_t g) { return goff[g + 1] - goff[g]; };
        int32_t g = 0;
        while (g < n_genomes) {
            if ((uint32_t)genes_of(g) > kStripeGenes || (uint32_t)cells_of(g) > kStripeCells) {
                ++g;
                continue;
            }
            StripeInfo si{};
            si.gene_lo = genome_off[g];
            si.genome_lo = g;
            si.cell_lo = goff[g];
            const int32_t sid = (int32_t)c->stripes_host.size();
            while (g < n_genomes && (uint32_t)(si.n_genes + genes_of(g)) <= kStripeGenes && (uint32_t)(si.n_cells + cells_of(g)) <= kStripeCells &&
                   (uint32_t)si.n_genomes < kStripeGenomes) {
                si.n_genes += genes_of(g);
                si.n_cells += cells_of(g);
                si.n_genomes += 1;
                stripe_of[g] = sid;
                ++g;
            }
            c->stripes_host.push_back(si);
        }
        c->stripes_usable = !c->stripes_host.empty() && n_genes > 0;
        if (c->stripes_host.empty()) c->stripes_host.push_back(StripeInfo{});
        if ((rc = upload(c, c->g_stripe_of, stripe_of.data(), stripe_of.size() * 4))) return rc;
        if ((rc = upload(c, c->g_stripes, c->stripes_host.data(), c->stripes_host.size() * sizeof(StripeInfo)))) return rc;
        if ((rc = upload(c, c->g_gene_off, genome_off, ((size_t)n_genomes + 1) * 4))) return rc;
    }
    c->genes_by_index = c->gene_index_opt != 0;
    c->genes_set = true;
    c->sb_valid = false;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->n_genomes = n_genomes;
    c->n_genes = n_genes;
    c->max_gene_feature = 0;
    for (int32_t i = 0; i < n_genes; ++i) c->max_gene_feature = std::max(c->max_gene_feature, gene_feature[i]);
    return WK_OK;
}

int wk_set_subjects(wk_ctx* c, const int32_t* feature_of_subject, int32_t n) {
    if (!c) return WK_E_ARG;
    if (n < 0 || (n > 0 && !feature_of_subject)) return fail(c, WK_E_ARG, "bad subject table arguments");
    int32_t mx = -1;
    for (int32_t s = 0; s < n; ++s) {
        if (feature_of_subject[s] < 0 || feature_of_subject[s] > WK_MAX_FEATURE)
            return fail(c, WK_E_RANGE, "subject %d: feature id outside [0, %d]", s, WK_MAX_FEATURE);
        mx = std::max(mx, feature_of_subject[s]);
    }
    c->max_subject_feature = mx;
    DeviceGuard guard(c->device);
    int rc;
    // accumulated packed records name subjects of the old table: they stay
    // valid only if the new table extends it
    if (c->w_open && c->w_records > 0 &&
        (n < c->n_subjects || memcmp(feature_of_subject, c->subj_feat_host.data(), (size_t)c->n_subjects * 4) != 0) &&
        (rc = wk_words_flush(c)))
        return rc;
    rc = upload(c, c->subj_feat, feature_of_subject, (size_t)n * sizeof(int32_t));
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->n_subjects = n;
    c->subj_feat_host.assign(feature_of_subject, feature_of_subject + n);
    c->subj_serial += 1;
    c->rows_sig.clear();  // rows are rebuilt on the next classify call
    return WK_OK;
}

// ---- count table -------------------------------------------------------------

int wk_counts_clear(wk_ctx* c) {
    if (!c) return WK_E_ARG;
    if (!c->slots) return fail(c, WK_E_STATE, "count table not reserved");
    DeviceGuard guard(c->device);
    if (!c->words_keep) {  // accumulated records are counts that have not been added yet
        c->w_records = c->w_reads = 0;
        c->w_open = false;
    }
    HIP_TRY(c, hipMemsetAsync(c->tkeys.p, 0xFF, c->slots * sizeof(uint64_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->tvals.p, 0, c->slots * sizeof(uint64_t), c->stream));
    return WK_OK;
}

int wk_counts_reserve(wk_ctx* c, int64_t min_slots) {
    if (!c) return WK_E_ARG;
    if (min_slots < 1) min_slots = 1;
    uint64_t s = 1024;
    while (s < (uint64_t)min_slots) s <<= 1;
    DeviceGuard guard(c->device);
    if (s != c->slots) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->tkeys.release();
        c->tvals.release();
        HIP_TRY(c, c->tkeys.reserve(s * sizeof(uint64_t)));
        HIP_TRY(c, c->tvals.reserve(s * sizeof(uint64_t)));
        c->slots = s;
    }
    return wk_counts_clear(c);
}

int wk_counts_fetch(wk_ctx* c, uint64_t* keys, int64_t* counts, int64_t cap, int64_t* n) {
    if (!c || !n) return WK_E_ARG;
    if (!c->slots) return fail(c, WK_E_STATE, "count table not reserved");
    DeviceGuard guard(c->device);
    int rc = c->words_keep ? WK_OK : wk_words_flush(c);  // records still accumulated (wk_words_append) are counted now
    if (rc) return rc;
    if ((rc = check_device_errors(c))) return rc;
    HIP_TRY(c, hipMemsetAsync(scalar_u64(c, 4), 0, sizeof(unsigned long long), c->stream));
    const int blocks = grid_for((int64_t)c->slots, 256, 2048);
    hipLaunchKernelGGL(table_count_kernel, dim3(blocks), dim3(256), 0, c->stream, c->tkeys.as<unsigned long long>(),
                       c->slots, scalar_u64(c, 4));
    HIP_TRY(c, hipGetLastError());
    unsigned long long used = 0;
    HIP_TRY(c, hipMemcpyAsync(&used, scalar_u64(c, 4), sizeof used, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n = (int64_t)used;
    if ((int64_t)used > cap || (used && (!keys || !counts))) return fail(c, WK_E_CAPACITY, "output capacity %lld < %llu entries", (long long)cap, used);
    if (used == 0) return WK_OK;
    HIP_TRY(c, c->fetch_k.reserve(used * sizeof(uint64_t)));
    HIP_TRY(c, c->fetch_v.reserve(used * sizeof(int64_t)));
    HIP_TRY(c, hipMemsetAsync(scalar_u64(c, 4), 0, sizeof(unsigned long long), c->stream));
    KernelTimer* kt = ktimer_begin(c, "compact");
    hipLaunchKernelGGL(table_compact_kernel, dim3(blocks), dim3(256), 0, c->stream, c->tkeys.as<unsigned long long>(),
                       c->tvals.as<unsigned long long>(), c->slots, scalar_u64(c, 4),
                       c->fetch_k.as<unsigned long long>(), c->fetch_v.as<long long>());
    ktimer_end(c, kt);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(keys, c->fetch_k.p, used * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(counts, c->fetch_v.p, used * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return WK_OK;
}

int wk_log_reserve(wk_ctx* c, int64_t n_entries) {
    if (!c) return WK_E_ARG;
    if (n_entries < 1) return fail(c, WK_E_ARG, "log capacity must be positive");
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, c->log.reserve((size_t)n_entries * 16));
    c->log_cap = n_entries;
    HIP_TRY(c, hipMemsetAsync(scalar_u64(c, 5), 0, 8, c->stream));
    return WK_OK;
}

int wk_log_fetch(wk_ctx* c, int32_t* out, int64_t cap, int64_t* n) {
    if (!c || !n) return WK_E_ARG;
    DeviceGuard guard(c->device);
    int rc = check_device_errors(c);
    if (rc) return rc;
    unsigned long long used = 0;
    HIP_TRY(c, hipMemcpyAsync(&used, scalar_u64(c, 5), 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n = (int64_t)used;
    if ((int64_t)used > c->log_cap) {
        HIP_TRY(c, hipMemsetAsync(scalar_u64(c, 5), 0, 8, c->stream));
        return fail(c, WK_E_CAPACITY, "contribution log overflowed: %llu entries needed, %lld reserved", used, (long long)c->log_cap);
    }
    if ((int64_t)used > cap || (used && !out)) return fail(c, WK_E_CAPACITY, "output capacity %lld < %llu entries", (long long)cap, used);
    if (used) HIP_TRY(c, hipMemcpyAsync(out, c->log.p, (size_t)used * 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemsetAsync(scalar_u64(c, 5), 0, 8, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return WK_OK;
}

// ---- classify ----------------------------------------------------------------

int wk_chunk_stage(wk_ctx* c, const int32_t* subj, const int32_t* qoff, int64_t n_reads, const int32_t* group,
                   int subj_flags) {
    if (!c) return WK_E_ARG;
    if (n_reads < 0 || !qoff) return fail(c, WK_E_ARG, "bad chunk arguments");
    const int64_t n_rec = qoff[n_reads];
    if (qoff[0] != 0 || n_rec < 0 || (n_rec > 0 && !subj)) return fail(c, WK_E_ARG, "qoff must start at 0 and end at n_records");
    const bool uniform = (subj_flags & WK_GROUP_UNIFORM) != 0;
    if (uniform && (!group || group[0] < 0 || group[0] >= (1 << WK_KEY_GROUP_BITS)))
        return fail(c, WK_E_ARG, "WK_GROUP_UNIFORM needs one group id in [0, %d)", 1 << WK_KEY_GROUP_BITS);
    DeviceGuard guard(c->device);
    int rc;
    // (64 bytes of slack behind the records: the weighted histogram reads them 16 bytes at a time)
    HIP_TRY(c, c->c_subj.reserve((size_t)n_rec * sizeof(int32_t) + 64));
    if ((rc = upload(c, c->c_subj, subj, (size_t)n_rec * sizeof(int32_t)))) return rc;
    if ((rc = upload(c, c->c_qoff, qoff, ((size_t)n_reads + 1) * sizeof(int32_t)))) return rc;
    if (group && !uniform && (rc = upload(c, c->c_group, group, (size_t)n_reads * sizeof(int32_t)))) return rc;
    // the read size of every record (what the record histogram of wk_weigh.hpp
    // streams next to the subject indices) + the reads it does not cover
    c->rk_valid[0] = c->rk_valid[1] = false;
    c->stage_serial += 1;
    unsigned long long totals[2] = {0, 0};
    if (c->use_weigh && (subj_flags & WK_SUBJ_INDEXED) && (subj_flags & WK_SUBJ_IS_SET) && n_reads > 0 &&
        n_reads < (1ll << 30) && n_rec < (1ll << 30)) {
        HIP_TRY(c, c->c_rk[0].reserve((size_t)n_rec + 64));
        HIP_TRY(c, c->rk_left[0].reserve((size_t)((n_reads + 63) / 64) * 8));
        HIP_TRY(c, c->rk_totals.reserve(32));
        KernelTimer* kt = ktimer_begin(c, "read_sizes");
        if ((rc = derive_read_sizes(c, false, c->c_qoff.as<int32_t>(), c->c_subj.as<int32_t>(), n_reads, n_rec,
                                    c->c_rk[0].as<unsigned char>(), c->rk_left[0].as<unsigned long long>(),
                                    c->rk_totals.as<unsigned long long>())))
            return rc;
        ktimer_end(c, kt);
        HIP_TRY(c, hipMemcpyAsync(totals, c->rk_totals.p, 16, hipMemcpyDeviceToHost, c->stream));
        c->rk_valid[0] = true;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // host buffers are only valid during the call
    c->rk_reads[0] = (int64_t)totals[0];
    c->rk_records[0] = (int64_t)totals[1];
    c->cur_subj = c->c_subj.as<int32_t>();
    c->cur_qoff = c->c_qoff.as<int32_t>();
    c->n_reads = n_reads;
    c->n_records = n_rec;
    c->has_group = group != nullptr && !uniform;
    c->group_base = uniform ? group[0] : 0;
    c->subj_is_set = (subj_flags & WK_SUBJ_IS_SET) != 0;
    c->subj_indexed = (subj_flags & WK_SUBJ_INDEXED) != 0;
    c->chunk_valid = true;
    return WK_OK;
}

int wk_classify_staged(wk_ctx* c, const wk_job* jobs, int32_t n_jobs, int32_t* out_assign) {
    if (!c) return WK_E_ARG;
    if (!jobs || n_jobs < 1 || n_jobs > WK_MAX_JOBS) return fail(c, WK_E_ARG, "n_jobs must be in [1,%d]", WK_MAX_JOBS);
    if (!c->chunk_valid) return fail(c, WK_E_STATE, "no chunk staged");
    if (!c->slots) return fail(c, WK_E_STATE, "count table not reserved (wk_counts_reserve)");
    DeviceGuard guard(c->device);
    KtScope kt_scope(c);

    ClassifyArgs a{};
    a.subj = c->cur_subj;
    a.qoff = c->cur_qoff;
    a.group = c->has_group ? c->c_group.as<int32_t>() : nullptr;
    a.group_base = c->has_group ? 0 : c->group_base;
    a.n_reads = c->n_reads;
    a.nodes = c->n_nodes ? c->nodes.as<Node>() : nullptr;
    a.n_nodes = c->n_nodes;
    a.n_jobs = n_jobs;
    a.subj_is_set = c->subj_is_set ? 1 : 0;
    for (int j = 0; j < n_jobs; ++j) {
        const wk_job& jb = jobs[j];
        JobDev d{};
        d.mode = jb.mode;
        d.flags = jb.flags;
        d.major = jb.major;
        d.anc = nullptr;
        if (jb.mode == WK_MODE_FREE || jb.mode == WK_MODE_RANK) {
            if (c->n_nodes <= 0) return fail(c, WK_E_STATE, "job %d needs a hierarchy (wk_set_tree)", j);
        } else if (jb.mode != WK_MODE_NONE) {
            return fail(c, WK_E_ARG, "job %d: unknown mode %d", j, jb.mode);
        }
        if (jb.mode == WK_MODE_RANK) {
            if (jb.rank_slot < 0 || jb.rank_slot >= (int)(sizeof c->rank_tab / sizeof c->rank_tab[0]) || !c->rank_tab_valid[jb.rank_slot])
                return fail(c, WK_E_STATE, "job %d: rank slot %d has not been built", j, jb.rank_slot);
            d.anc = c->rank_tab[jb.rank_slot].as<int32_t>();
            if (jb.major < 0.0 || jb.major > 1.0) return fail(c, WK_E_ARG, "job %d: major must be a fraction in [0,1]", j);
        }
        a.jobs[j] = d;
    }
    if (c->subj_indexed) {
        const int rc_rows = build_subject_rows(c, jobs, n_jobs, a);
        if (rc_rows) return rc_rows;
    } else {
        // chunk of feature ids: rank columns for the first pass of the split
        // (at most three distinct ranks, like a subject row)
        int slots[3];
        int n_cols = 0;
        for (int j = 0; j < n_jobs && n_cols >= 0; ++j) {
            if (jobs[j].mode != WK_MODE_RANK) continue;
            int col = -1;
            for (int q = 0; q < n_cols; ++q)
                if (slots[q] == jobs[j].rank_slot) col = q;
            if (col < 0) {
                if (n_cols == 3) {
                    n_cols = -1;  // too many ranks: no split for this launch
                    break;
                }
                col = n_cols++;
                slots[col] = jobs[j].rank_slot;
                a.col_anc[col] = c->rank_tab[jobs[j].rank_slot].as<int32_t>();
            }
            a.jobs[j].col = col;
        }
        a.n_cols = n_cols;
    }
    if (out_assign) {
        HIP_TRY(c, c->assign_out.reserve((size_t)n_jobs * (size_t)(c->n_reads ? c->n_reads : 1) * sizeof(int32_t)));
        a.out_assign = c->assign_out.as<int32_t>();
    }
    a.stat_block = c->stat_block.as<unsigned long long>();
    a.ablate = (uint32_t)c->ablate;
    a.log = c->log.as<int32_t>();
    a.log_cursor = scalar_u64(c, 5);
    a.log_cap = c->log_cap;
    for (int j = 0; j < n_jobs; ++j)
        if ((jobs[j].flags & WK_F_SIZED) && c->log_cap <= 0)
            return fail(c, WK_E_STATE, "job %d is size-normalised but no log is reserved (wk_log_reserve)", j);
    a.table = CountTable{c->tkeys.as<unsigned long long>(), c->tvals.as<unsigned long long>(), c->slots - 1, scalar_err(c)};

    if (c->n_reads > 0) {
        KernelTimer* kt = ktimer_begin(c, "classify");
        if (c->use_lds && c->tiled && !c->subj_indexed) {
            // persistent workgroups, each with its own tile buffers + LDS front cache
            const size_t lds = (size_t)(kTileWindow + 4 + kTileReads + 4) * sizeof(int32_t) + (size_t)c->lds_slots * 16;
            const int64_t n_tiles = (c->n_reads + kTileReads - 1) / kTileReads;
            const int blocks = (int)std::min<int64_t>(std::min<int64_t>(n_tiles, kStatBlocks), (int64_t)c->prop.multiProcessorCount * c->blocks_per_cu);
            hipLaunchKernelGGL(classify_tiled_kernel, dim3(blocks), dim3(kTileThreads), lds, c->stream, a,
                               (uint32_t)c->lds_slots, c->n_records);
        } else if (c->use_lds) {
            bool sized = false;
            for (int j = 0; j < n_jobs; ++j) sized |= (jobs[j].flags & WK_F_SIZED) != 0;
            // two-class split (32-bit byte offsets in the first pass bound the
            // sizes).  It pays even when only half of the reads have a single
            // candidate (config 3: 10.6 -> 7.7 ms); a chunk without any pays one
            // streaming pass over the offsets.
            const bool feature_chunk = !c->subj_indexed;
            // (ext: the caller evaluates only the reads of its own bit mask — wk_ordinal_count's leftovers)
            const bool ext = c->listed_only;
            const bool split = !ext && c->use_split && c->n_reads < (1ll << 30) && c->n_records < (1ll << 30) &&
                               (feature_chunk ? a.n_cols >= 0 : (a.row_w == 4 && c->n_subjects < (1 << 28)));
            // ... and with a small subject table the first pass only histograms
            // subject indices; the assigners run once per subject afterwards
            const bool subject_ok = split && !feature_chunk && c->use_subject_bins && !out_assign && !c->has_group && !sized;
            const bool by_subject = subject_ok && c->n_subjects <= 28672;
            // a larger table: the first 24,576 subject indices (first appearance
            // order: the abundant ones) in bins, the others per read
            const bool hot_subjects = subject_ok && !by_subject && c->use_subject_bins >= 1 && c->use_hot_bins;
            constexpr int kHotBins = 24576;
            // weighted subject histogram (wk_weigh.hpp): plain assigners only, every
            // read a set of subject indices, one group, no per-read output
            bool weigh = !ext && c->use_weigh && c->subj_indexed && c->subj_is_set && !c->has_group && !out_assign && !sized &&
                         c->n_reads < (1ll << 30) && c->n_records < (1ll << 30) && c->n_subjects > 0;
            for (int j = 0; j < n_jobs && weigh; ++j) {
                if (jobs[j].mode == WK_MODE_NONE)
                    weigh = !(jobs[j].flags & WK_F_UNIQ);
                else if (jobs[j].mode == WK_MODE_RANK)
                    weigh = !(jobs[j].flags & (WK_F_UNIQ | WK_F_ABOVE)) && !(jobs[j].major > 0.0);
                else
                    weigh = false;
            }
            // auto: chunks that are worth the fixed cost (slab rows of every
            // workgroup) and are not all single-candidate reads (the count-first
            // pass above is the faster special case of those)
            if (weigh && c->use_weigh == 1)
                weigh = c->n_reads >= (1 << 16) && c->n_records > c->n_reads + c->n_reads / 64;
            uint32_t w_bins = 0, w_slices = 0, w_teams = 0, w_xcd = 8;
            weigh = weigh && c->rk_valid[0] && a.subj == c->c_subj.as<int32_t>();
            if (weigh) {
                const uint32_t cus = (uint32_t)c->prop.multiProcessorCount;
                if (cus % w_xcd) w_xcd = 1;
                const int64_t cap = (int64_t)kBinsMaxLds / 4 - 96;
                w_slices = (uint32_t)((c->n_subjects + cap - 1) / cap);
                w_bins = ((uint32_t)c->n_subjects + w_slices - 1) / w_slices;  // slice = w_bins consecutive subject indices
                w_teams = (cus / w_xcd) / w_slices;
                if (!w_teams) weigh = false;
            }
            const int max_blocks = std::min(kStatBlocks, c->prop.multiProcessorCount * c->blocks_per_cu);
            const int blocks = grid_for(c->n_reads, c->threads, max_blocks);
            // dense bins: small id space, subject-indexed chunk, no size-normalised job
            int64_t bins = 0;
            int lds_slots = c->lds_slots;
            if (c->use_dense && c->subj_indexed && !by_subject && !hot_subjects && !weigh) {
                const int64_t b = std::max<int64_t>(c->n_nodes, (int64_t)c->max_subject_feature + 1);
                if (!sized && b * n_jobs <= 28672) {  // <= 112 KiB of bins + 32 KiB hash cache = 144 KiB LDS
                    bins = b;
                    lds_slots = std::min(lds_slots, 2048);
                    // two workgroups per CU fit if each stays below 80 KiB
                    if (c->blocks_per_cu >= 2 && b * n_jobs * 4 + 1024 * 16 <= 80 * 1024) lds_slots = 1024;
                }
            }
            size_t lds = (size_t)lds_slots * 16;
            // partitioned miss log: worth its fixed cost (one more merge
            // launch) only when many keys can miss the LDS cache
            uint32_t plog_cap = 0;
            // contributions that can reach the log: all of them, or — when the
            // single-candidate reads are counted per subject — those of the
            // multi-hit reads (each has >= 2 records, so at most twice the
            // excess of records over reads when no read is empty; an estimate,
            // the streams overflow into the count table)
            const int64_t contrib = by_subject ? 3 * std::max<int64_t>(c->n_records - c->n_reads, 0) + 1024
                                               : c->n_records + c->n_reads;
            if (!bins && !weigh && (c->use_plog == 2 || (c->use_plog == 1 && contrib >= (1 << 22)))) {
                // partitions: the merge counts a partition in one LDS table of
                // 8192 slots, so 256 partitions hold ~1.3 M distinct keys at a
                // comfortable load; fewer partitions keep a workgroup's open
                // log lines in L2 until they are full.  Distinct keys of a
                // launch <= sum over jobs of the ids the job can emit.
                int64_t distinct = 0;
                for (int j = 0; j < n_jobs; ++j) {
                    const int64_t ids = jobs[j].mode == WK_MODE_RANK   ? c->rank_tab_nodes[jobs[j].rank_slot]
                                        : jobs[j].mode == WK_MODE_FREE ? (int64_t)c->n_nodes
                                        : c->subj_indexed              ? (int64_t)c->n_subjects
                                        : c->n_genes > 0               ? (int64_t)c->n_genes
                                                                       : (int64_t)WK_MAX_FEATURE;
                    distinct += ids + 1;
                }
                const uint32_t log_parts = c->log_parts_opt ? (uint32_t)c->log_parts_opt
                                           : (!c->has_group && distinct <= 256 * 5120) ? 256u : kLogPartsMax;
                a.log_parts = log_parts;
                const int64_t streams = (int64_t)blocks * log_parts;
                // room for 3x the expected entries per stream if every contribution missed
                int64_t cap = 3 * (contrib * (int64_t)n_jobs / streams + 1) + 16;
                cap = std::min<int64_t>(cap, c->plog_max_bytes / 8 / streams);
                if (!sized && cap >= 16) {
                    plog_cap = (uint32_t)cap;
                    lds = (size_t)lds_slots * 16 + log_parts * 4;  // + the stream cursors
                    HIP_TRY(c, c->plog.reserve((size_t)streams * plog_cap * 8));
                    HIP_TRY(c, c->plog_cnt.reserve((size_t)streams * 4));
                    a.plog = c->plog.as<unsigned long long>();
                    a.plog_cnt = c->plog_cnt.as<uint32_t>();
                    a.plog_cap = plog_cap;
                }
            }
            if (bins) {
                lds += (size_t)bins * n_jobs * 4;
                HIP_TRY(c, c->dense_slab.reserve((size_t)blocks * bins * n_jobs * 4));
                a.dense_bins = (uint32_t)bins;
                a.dense_total = (uint32_t)(bins * n_jobs);
                a.dense_slab = c->dense_slab.as<uint32_t>();
            }
            if (weigh) {
                // ---- weighted subject histogram + per-subject merge; the reads it
                // does not cover go to the second pass below through a mask
                const uint32_t n_words = (uint32_t)((c->n_reads + 63) / 64);
                const uint32_t list_seg = (((n_words + 15u) / 16u + (uint32_t)blocks - 1u) / (uint32_t)blocks) * 1024u;
                const uint32_t n_teams = w_xcd * w_teams;
                HIP_TRY(c, c->left_list.reserve((size_t)blocks * list_seg * 4));
                HIP_TRY(c, c->w_slab.reserve((size_t)w_slices * n_teams * w_bins * 4));
                if ((size_t)c->n_subjects > c->w_hi_clean) {
                    HIP_TRY(c, c->w_hi.reserve((size_t)c->n_subjects * 4 + ((size_t)c->n_subjects * 4) / 2));
                    HIP_TRY(c, hipMemsetAsync(c->w_hi.p, 0, c->w_hi.cap, c->stream));
                    c->w_hi_clean = c->w_hi.cap / 4;
                }
                // which derivation of the read sizes: the one of the staging, or —
                // some subject lacks an ancestor at a rank in use — the one that
                // also leaves out the reads naming such a subject (once per staged
                // chunk and set of rank columns)
                int v = 0;
                if (c->rows_any_invalid) {
                    v = 1;
                    if (!c->rk_valid[1] || c->rk1_stage != c->stage_serial || c->rk1_rows != c->rows_serial) {
                        ktimer_end(c, kt);
                        kt = ktimer_begin(c, "read_sizes");
                        HIP_TRY(c, c->c_rk[1].reserve((size_t)c->n_records + 64));
                        HIP_TRY(c, c->rk_left[1].reserve((size_t)n_words * 8));
                        {
                            const int rc1 = derive_read_sizes(c, true, a.qoff, a.subj, c->n_reads, c->n_records,
                                                              c->c_rk[1].as<unsigned char>(),
                                                              c->rk_left[1].as<unsigned long long>(),
                                                              c->rk_totals.as<unsigned long long>() + 2);
                            if (rc1) return rc1;
                        }
                        ktimer_end(c, kt);
                        unsigned long long totals[2] = {0, 0};
                        HIP_TRY(c, hipMemcpyAsync(totals, c->rk_totals.as<unsigned char>() + 16, 16, hipMemcpyDeviceToHost, c->stream));
                        HIP_TRY(c, hipStreamSynchronize(c->stream));
                        c->rk_reads[1] = (int64_t)totals[0];
                        c->rk_records[1] = (int64_t)totals[1];
                        c->rk_valid[1] = true;
                        c->rk1_stage = c->stage_serial;
                        c->rk1_rows = c->rows_serial;
                        kt = ktimer_begin(c, "classify");
                    }
                }
                BinsArgs ba{};
                ba.subj = a.subj;
                ba.rk = c->c_rk[v].as<unsigned char>();
                ba.n_records = (uint32_t)c->n_records;
                ba.n_subjects = (uint32_t)c->n_subjects;
                ba.bins = w_bins;
                ba.n_slices = w_slices;
                ba.teams_per_xcd = w_teams;
                ba.n_xcd = w_xcd;
                ba.slab = c->w_slab.as<uint32_t>();
                ba.hi = c->w_hi.as<uint32_t>();
                ba.err = scalar_err(c);
                const dim3 wgrid((unsigned)c->prop.multiProcessorCount);
                const size_t wlds = ((size_t)w_bins + 96) * 4;
                kt = ktimer_begin(c, "classify");  // (again: the bracket starts at the launch, not at the host's decisions above)
                if (c->bins_ring == 6)
                    hipLaunchKernelGGL((weigh_bins_kernel<6>), wgrid, dim3(kWeighThreads), wlds, c->stream, ba);
                else if (c->bins_ring == 8)
                    hipLaunchKernelGGL((weigh_bins_kernel<8>), wgrid, dim3(kWeighThreads), wlds, c->stream, ba);
                else if (c->bins_ring == 3)
                    hipLaunchKernelGGL((weigh_bins_kernel<3>), wgrid, dim3(kWeighThreads), wlds, c->stream, ba);
                else
                    hipLaunchKernelGGL((weigh_bins_kernel<4>), wgrid, dim3(kWeighThreads), wlds, c->stream, ba);
                // (the reads and records it covers were counted with the sizes)
                c->stat_extra_reads += c->rk_reads[v];
                c->stat_extra_records += c->rk_records[v];
                ktimer_end(c, kt);
                kt = ktimer_begin(c, "weigh_merge");
                WeighMergeArgs wm{};
                wm.slab = ba.slab;
                wm.hi = ba.hi;
                wm.n_subjects = ba.n_subjects;
                wm.bins = w_bins;
                wm.n_teams = n_teams;
                wm.rows = a.rows;
                wm.row_w = a.row_w;
                wm.n_jobs = n_jobs;
                for (int j = 0; j < n_jobs; ++j) {
                    wm.mode[j] = a.jobs[j].mode;
                    wm.col[j] = a.jobs[j].col;
                }
                wm.group = (uint32_t)a.group_base;
                wm.table = a.table;
                hipLaunchKernelGGL(weigh_merge_kernel, dim3((ba.n_subjects + kMergeSubjects - 1u) / kMergeSubjects), dim3(1024), (size_t)4096 * 16,
                                   c->stream, wm, 4096u);
                ktimer_end(c, kt);
                kt = ktimer_begin(c, "leftover");
                a.left_mask = c->rk_left[v].as<unsigned long long>();
                a.n_mask_words = n_words;
                a.list_seg = list_seg;
                a.read_list = c->left_list.as<uint32_t>();
                a.resume = 0;
                a.slab16 = 0;
            } else if (split) {
                // ---- first pass: single-candidate reads -----------------------------
                const uint32_t n_words = (uint32_t)((c->n_reads + 63) / 64);
                const uint32_t list_seg = (((n_words + 15u) / 16u + (uint32_t)blocks - 1u) / (uint32_t)blocks) * 1024u;
                HIP_TRY(c, c->left_mask.reserve((size_t)n_words * 8));
                HIP_TRY(c, c->left_list.reserve((size_t)blocks * list_seg * 4));
                unsigned long long* mask = c->left_mask.as<unsigned long long>();
                ClassifyArgs first = a;
                first.plog = nullptr;  // (kept when the first pass counts per read, below)
                first.plog_cnt = nullptr;
                first.plog_cap = 0;
                int blocks1 = blocks;
                if (by_subject) {
                    // its own grid and LDS layout: no hash cache to speak of, bins = subjects
                    blocks1 = grid_for((c->n_reads + 3) / 4, 1024,
                                       std::min(kStatBlocks, c->prop.multiProcessorCount * c->single_blocks_per_cu));
                    HIP_TRY(c, c->first_slab.reserve((size_t)blocks1 * ((size_t)c->n_subjects + 64) * 4));  // (16-bit layout pads units to 64 columns)
                    first.dense_bins = (uint32_t)c->n_subjects;
                    first.dense_total = (uint32_t)c->n_subjects;
                    first.dense_by_subject = 1;
                    first.dense_slab = c->first_slab.as<uint32_t>();
                    // reads per workgroup: its rounds x 4096
                    const int64_t rounds = ((c->n_reads + 4095) / 4096 + blocks1 - 1) / blocks1;
                    first.slab16 = rounds * 4096 <= 65535 ? 1 : 0;
                    if (c->use_count_kernel)
                        hipLaunchKernelGGL(count_subjects_kernel, dim3(blocks1), dim3(1024),
                                           64 * 16 + (size_t)c->n_subjects * 4, c->stream, first, 64u, mask);
                    else
                        hipLaunchKernelGGL((classify_single_kernel<true, true, 4>), dim3(blocks1), dim3(1024),
                                           64 * 16 + (size_t)c->n_subjects * 4, c->stream, first, 64u, mask);
                } else if (hot_subjects) {
                    // hash cache + log cursors like the second pass (shared streams),
                    // plus the hot subjects' bins behind them
                    HIP_TRY(c, c->first_slab.reserve((size_t)blocks * ((size_t)kHotBins + 64) * 4));
                    first.plog = a.plog;
                    first.plog_cnt = a.plog_cnt;
                    first.plog_cap = a.plog_cap;
                    first.dense_bins = kHotBins;
                    first.dense_total = kHotBins;
                    first.dense_by_subject = 1;
                    first.dense_slab = c->first_slab.as<uint32_t>();
                    const int64_t per_wg = ((c->n_reads + (int64_t)blocks * c->threads * 2 - 1) / ((int64_t)blocks * c->threads * 2)) * c->threads * 2;
                    first.slab16 = per_wg <= 65535 ? 1 : 0;
                    const int hot_slots = std::min(lds_slots, 2048);
                    const size_t lds1 = (size_t)hot_slots * 16 + (a.plog ? a.log_parts * 4 : 0) + (size_t)kHotBins * 4;
                    if (n_jobs == 1)
                        hipLaunchKernelGGL((classify_single_kernel<true, true, 2, false, true>), dim3(blocks), dim3(c->threads), lds1,
                                           c->stream, first, (uint32_t)hot_slots, mask);
                    else
                        hipLaunchKernelGGL((classify_single_kernel<true, false, 2, false, true>), dim3(blocks), dim3(c->threads), lds1,
                                           c->stream, first, (uint32_t)hot_slots, mask);
                } else {
                    // per-read evaluation with the same counting levels as the second
                    // pass; dense bins go to a slab of their own, log streams are shared
                    size_t lds1 = lds;
                    if (bins) {
                        HIP_TRY(c, c->first_slab.reserve((size_t)blocks * ((size_t)bins * n_jobs + 64) * 4));
                        first.dense_slab = c->first_slab.as<uint32_t>();
                        const int64_t per_wg = ((c->n_reads + (int64_t)blocks * c->threads - 1) / ((int64_t)blocks * c->threads)) * c->threads;
                        first.slab16 = per_wg <= 65535 ? 1 : 0;
                    } else {
                        first.plog = a.plog;
                        first.plog_cnt = a.plog_cnt;
                        first.plog_cap = a.plog_cap;
                    }
                    if (feature_chunk && n_jobs == 1)
                        hipLaunchKernelGGL((classify_single_kernel<false, true, kPerReadItems, true>), dim3(blocks), dim3(c->threads), lds1,
                                           c->stream, first, (uint32_t)lds_slots, mask);
                    else if (feature_chunk)
                        hipLaunchKernelGGL((classify_single_kernel<false, false, kPerReadItems, true>), dim3(blocks), dim3(c->threads), lds1,
                                           c->stream, first, (uint32_t)lds_slots, mask);
                    else if (n_jobs == 1)
                        hipLaunchKernelGGL((classify_single_kernel<false, true, kPerReadItems>), dim3(blocks), dim3(c->threads), lds1,
                                           c->stream, first, (uint32_t)lds_slots, mask);
                    else
                        hipLaunchKernelGGL((classify_single_kernel<false, false, kPerReadItems>), dim3(blocks), dim3(c->threads), lds1,
                                           c->stream, first, (uint32_t)lds_slots, mask);
                }
                ktimer_end(c, kt);
                kt = ktimer_begin(c, "leftover");
                // ---- second pass: merges the first pass's bins, lists and walks the rest
                a.left_mask = mask;
                a.n_mask_words = n_words;
                a.list_seg = list_seg;
                a.read_list = c->left_list.as<uint32_t>();
                if (by_subject || hot_subjects || bins) {
                    a.first_slab = c->first_slab.as<uint32_t>();
                    a.first_rows = (uint32_t)blocks1;
                    a.first_total = first.dense_total;
                    a.first_by_subject = (by_subject || hot_subjects) ? 1 : 0;
                    a.first_slab16 = first.slab16;
                }
                a.resume = (!by_subject && !bins && plog_cap) ? 1 : 0;
                a.slab16 = 0;
                if (bins) {  // the second pass counts in the hash cache only
                    a.dense_bins = 0;
                    a.dense_total = 0;
                    a.dense_slab = nullptr;
                    lds = (size_t)lds_slots * 16;
                    bins = 0;
                }
            }
            // one evaluator per kind of candidates (see classify_kernel's kPath)
            const int path = (a.rows != nullptr && a.row_w == 4) ? 0 : a.rows != nullptr ? 1 : 2;
            const dim3 grid(blocks), block(c->threads);
            if (ext) {
                const uint32_t n_words = (uint32_t)((c->n_reads + 63) / 64);
                const uint32_t list_seg = (((n_words + 15u) / 16u + (uint32_t)blocks - 1u) / (uint32_t)blocks) * 1024u;
                HIP_TRY(c, c->left_list.reserve((size_t)blocks * list_seg * 4));
                a.left_mask = c->left_mask.as<unsigned long long>();
                a.n_mask_words = n_words;
                a.list_seg = list_seg;
                a.read_list = c->left_list.as<uint32_t>();
                a.resume = 0;
                a.slab16 = 0;
            }
            const bool listed = split || weigh || ext;
            if (listed && path == 0)
                hipLaunchKernelGGL((classify_kernel<true, true, 0>), grid, block, lds, c->stream, a, (uint32_t)lds_slots);
            else if (listed && path == 1)
                hipLaunchKernelGGL((classify_kernel<true, true, 1>), grid, block, lds, c->stream, a, (uint32_t)lds_slots);
            else if (listed)
                hipLaunchKernelGGL((classify_kernel<true, true, 2>), grid, block, lds, c->stream, a, (uint32_t)lds_slots);
            else if (path == 0)
                hipLaunchKernelGGL((classify_kernel<true, false, 0>), grid, block, lds, c->stream, a, (uint32_t)lds_slots);
            else if (path == 1)
                hipLaunchKernelGGL((classify_kernel<true, false, 1>), grid, block, lds, c->stream, a, (uint32_t)lds_slots);
            else
                hipLaunchKernelGGL((classify_kernel<true, false, 2>), grid, block, lds, c->stream, a, (uint32_t)lds_slots);
            if (plog_cap) {
                ktimer_end(c, kt);
                kt = ktimer_begin(c, "partition_merge");
                hipLaunchKernelGGL(partition_merge_kernel, dim3(a.log_parts), dim3(1024), (size_t)8192 * 16, c->stream,
                                   c->plog.as<unsigned long long>(), c->plog_cnt.as<uint32_t>(), (uint32_t)blocks,
                                   plog_cap, 8192u, a.table);
            }
            if (bins) {
                ktimer_end(c, kt);
                kt = ktimer_begin(c, "dense_merge");
                const uint32_t nb = (uint32_t)(bins * n_jobs);
                hipLaunchKernelGGL(dense_merge_kernel, dim3((nb + 63) / 64), dim3(1024), 0, c->stream,
                                   c->dense_slab.as<uint32_t>(), (uint32_t)blocks, (uint32_t)n_jobs, (uint32_t)bins, (uint32_t)a.group_base, a.table);
            }
        } else {
            const int blocks = grid_for(c->n_reads, 256, c->prop.multiProcessorCount * 8);
            hipLaunchKernelGGL(classify_kernel<false>, dim3(blocks), dim3(256), 0, c->stream, a, 0u);
        }
        ktimer_end(c, kt);
        HIP_TRY(c, hipGetLastError());
    }
    if (out_assign) {
        HIP_TRY(c, hipMemcpyAsync(out_assign, c->assign_out.p, (size_t)n_jobs * (size_t)c->n_reads * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return WK_OK;
}

int wk_classify_chunk(wk_ctx* c, const wk_job* jobs, int32_t n_jobs, const int32_t* subj, const int32_t* qoff,
                      int64_t n_reads, const int32_t* group, int subj_flags, int32_t* out_assign) {
    int rc = wk_chunk_stage(c, subj, qoff, n_reads, group, subj_flags);
    if (rc) return rc;
    return wk_classify_staged(c, jobs, n_jobs, out_assign);
}

// ---- packed records accumulated over the chunks of a sample ---------------------

int wk_host_alloc(wk_ctx* c, size_t bytes, void** out) {
    if (!c || !out) return WK_E_ARG;
    DeviceGuard guard(c->device);
    void* p = nullptr;
    HIP_TRY(c, hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault));
    {
        std::lock_guard<std::mutex> lock(c->host_mu);
        c->host_blocks.push_back(p);
    }
    *out = p;
    return WK_OK;
}

int wk_host_free(wk_ctx* c, void* p) {
    if (!c || !p) return WK_E_ARG;
    {
        std::lock_guard<std::mutex> lock(c->host_mu);
        auto it = std::find(c->host_blocks.begin(), c->host_blocks.end(), p);
        if (it == c->host_blocks.end()) return fail(c, WK_E_ARG, "not a block of wk_host_alloc");
        c->host_blocks.erase(it);
    }
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipHostFree(p));
    return WK_OK;
}

// Can the weighted histogram alone classify records under these jobs?  Plain
// assigners only (see wk_weigh.hpp) and — checked against the current subject
// table — every subject with an ancestor at every rank in use.
static int words_jobs_ok(wk_ctx* c, const wk_job* jobs, int32_t n_jobs, ClassifyArgs& a, bool* ok, int* mode = nullptr) {
    *ok = false;
    if (mode) *mode = 0;
    if (!c->use_weigh || c->n_subjects <= 0 || c->n_subjects > (int32_t)kWordSubjMask + 1) return WK_OK;
    // Jobs that look at whole reads — `--rank free` (classify.assign_free), a rank under --uniq, --above or --major above
    // one half (classify.assign_rank) — go to the per-read stream over node ids (wk_free.hpp): one such job with the
    // records rewritten as they are appended, several with the records rewritten per job when the sample is classified.
    {
        int n_stream = 0;
        for (int j = 0; j < n_jobs; ++j) {
            const wk_job& jb = jobs[j];
            if (jb.flags & WK_F_SIZED) continue;
            if (jb.mode == WK_MODE_FREE)
                n_stream += 1;
            else if (jb.mode == WK_MODE_RANK && (jb.major > 0.5 || (jb.major <= 0.0 && (jb.flags & (WK_F_UNIQ | WK_F_ABOVE)))))
                n_stream += 1;
        }
        if (n_stream == n_jobs && n_jobs >= 1) {
            if (c->n_nodes <= 0 || (uint32_t)c->n_nodes >= kFreeMissing) return WK_OK;
            for (int j = 0; j < n_jobs; ++j) {
                const wk_job& jb = jobs[j];
                if (jb.mode == WK_MODE_FREE) {
                    if (jb.flags & WK_F_SUBOK)  // a subject that is no node is its own result under --subok: not a feature the stream can carry
                        for (int32_t f : c->subj_feat_host)
                            if (f >= c->n_nodes) return WK_OK;
                } else if (jb.rank_slot < 0 || jb.rank_slot >= (int)(sizeof c->rank_tab / sizeof c->rank_tab[0]) ||
                           !c->rank_tab_valid[jb.rank_slot]) {
                    return fail(c, WK_E_STATE, "job %d: rank slot %d has not been built", j, jb.rank_slot);
                }
            }
            if (mode) *mode = n_jobs > 1 ? 3 : jobs[0].mode == WK_MODE_FREE ? 1 : 2;
            *ok = true;
            return WK_OK;
        }
    }
    a = ClassifyArgs{};
    a.n_jobs = n_jobs;
    for (int j = 0; j < n_jobs; ++j) {
        const wk_job& jb = jobs[j];
        if (jb.flags & (WK_F_UNIQ | WK_F_SIZED)) return WK_OK;
        if (jb.mode == WK_MODE_RANK) {
            if ((jb.flags & WK_F_ABOVE) || jb.major > 0.0) return WK_OK;
            if (c->n_nodes <= 0) return fail(c, WK_E_STATE, "job %d needs a hierarchy (wk_set_tree)", j);
            if (jb.rank_slot < 0 || jb.rank_slot >= (int)(sizeof c->rank_tab / sizeof c->rank_tab[0]) || !c->rank_tab_valid[jb.rank_slot])
                return fail(c, WK_E_STATE, "job %d: rank slot %d has not been built", j, jb.rank_slot);
        } else if (jb.mode != WK_MODE_NONE) {
            return WK_OK;
        }
        JobDev d{};
        d.mode = jb.mode;
        d.flags = jb.flags;
        d.major = jb.major;
        a.jobs[j] = d;
    }
    const int rc = build_subject_rows(c, jobs, n_jobs, a);
    if (rc) return rc;
    *ok = !c->rows_any_invalid;
    return WK_OK;
}

static bool same_jobs(const std::vector<wk_job>& have, const wk_job* jobs, int32_t n) {
    if ((int32_t)have.size() != n) return false;
    for (int32_t j = 0; j < n; ++j)
        if (have[j].mode != jobs[j].mode || have[j].rank_slot != jobs[j].rank_slot || have[j].flags != jobs[j].flags ||
            have[j].major != jobs[j].major)
            return false;
    return true;
}

int wk_words_flush(wk_ctx* c) {
    if (!c) return WK_E_ARG;
    if (c->lag_count) return fail(c, WK_E_STATE, "blocks whose verdict has not been read (wk_dtok_scan_emit_end)");
    if (!c->w_open || c->w_records == 0) return words_reset(c);
    if (!c->slots) return fail(c, WK_E_STATE, "count table not reserved (wk_counts_reserve)");
    DeviceGuard guard(c->device);
    KtScope kt_scope(c);
    if (c->w_mode != 0) {
        // ---- free-rank jobs and rank jobs that look at whole reads: per job, the stream over node ids, then its dense
        // counters into the count table.  One job: the records hold its node ids since they were appended; several:
        // they hold subject indices, which the stream reads through the job's table (free_stream_kernel<., true>).
        const int blocks = std::min(kStatBlocks, c->prop.multiProcessorCount * c->free_per_cu);
        const CountTable table{c->tkeys.as<unsigned long long>(), c->tvals.as<unsigned long long>(), c->slots - 1, scalar_err(c)};
        if (c->w_mode != 3) {  // (subjects registered since the last chunk was appended)
            const int rcr = refresh_word_ranks(c, c->w_records);
            if (rcr) return rcr;
        }
        for (size_t j = 0; j < c->w_jobs.size(); ++j) {
            const wk_job& jb = c->w_jobs[j];
            const int kind = jb.mode == WK_MODE_FREE ? 1 : 2;
            wk_ctx::StreamTables& T = c->st[j];
            {
                const int rcf = ensure_stream_tables(c, T, kind, kind == 2 ? jb.rank_slot : -1);
                if (rcf) return rcf;
            }
            FreeArgs fa{};
            fa.words = c->c_words.as<uint32_t>();
            const bool translate = c->w_mode == 3;  // (several jobs: the records hold subject indices, read through this job's table)
            fa.rank_of_subject = translate ? T.subj_rank.as<int32_t>() : nullptr;
            fa.n_subjects = (uint32_t)c->n_subjects;
            fa.err = scalar_err(c);
            fa.sparse = T.dsparse.as<int32_t>();
            fa.parent_d = T.dparent.as<int32_t>();
            fa.self_d = T.dself.as<int32_t>();
            fa.sparse_m = T.dm;
            fa.n_records = (uint32_t)c->w_records;
            fa.job = (uint32_t)j;
            fa.group = (uint32_t)c->w_group;
            fa.subok = (jb.flags & WK_F_SUBOK) ? 1u : 0u;
            fa.unassigned = (jb.flags & WK_F_UNASSIGNED) ? 1u : 0u;
            fa.by_rank = kind == 2 ? 1u : 0u;
            fa.above = (jb.flags & WK_F_ABOVE) ? 1u : 0u;
            fa.major = kind == 2 ? jb.major : 0.0;
            fa.count_stats = j == 0 ? 1u : 0u;  // (reads and records are counted once)
            // the dense counters of the results (zero between launches: free_counts_kernel clears what it moves)
            const size_t dense_bytes = ((size_t)T.results + 1) * 4;
            if (c->f_dense.cap < dense_bytes || c->f_dense_tree != c->tree_serial) {  // (otherwise zero)
                HIP_TRY(c, c->f_dense.reserve(dense_bytes));
                HIP_TRY(c, hipMemsetAsync(c->f_dense.p, 0, c->f_dense.cap, c->stream));
                c->f_dense_tree = c->tree_serial;
            }
            fa.dense = c->f_dense.as<uint32_t>();
            fa.n_results = T.results;
            fa.stat_block = c->stat_block.as<unsigned long long>();
            const uint32_t slots = (uint32_t)c->free_slots;
            const size_t lds = (size_t)slots * 8 + (size_t)(c->free_threads / 64) * free_wave_lds();
            // a wave's list of uncached results: room for every read it can meet (240 per block of records)
            const uint32_t n_waves = (uint32_t)blocks * (uint32_t)(c->free_threads / 64);
            const uint32_t n_blocks = (fa.n_records + kFreeAdvance - 1) / kFreeAdvance;
            fa.log_cap = ((n_blocks + n_waves - 1) / n_waves * kFreeAdvance + 63u) / 64u * 64u;
            const uint32_t n_slices = (T.results + 1 + kLogBins - 1) / kLogBins;
            HIP_TRY(c, c->f_log.reserve((size_t)n_waves * fa.log_cap * 4));
            HIP_TRY(c, c->f_log_cnt.reserve((size_t)n_waves * 4));
            // (shares x slices = workgroups of free_log_kernel, one per CU: as many as run at once)
            const uint32_t n_parts = std::max(8u, std::min(kListSharesMax, ((uint32_t)c->prop.multiProcessorCount / n_slices) & ~7u));
            HIP_TRY(c, c->f_partial.reserve((size_t)n_parts * n_slices * kLogBins * 4));
            HIP_TRY(c, c->f_part_used.reserve((size_t)n_parts * n_slices * 4));
            fa.log = c->f_log.as<uint32_t>();
            fa.log_cnt = c->f_log_cnt.as<uint32_t>();
            KernelTimer* kt = ktimer_begin(c, "classify");
            if (fa.major > 0.0 && translate)
                hipLaunchKernelGGL((free_stream_kernel<true, true>), dim3(blocks), dim3(c->free_threads), lds, c->stream, fa, slots);
            else if (fa.major > 0.0)
                hipLaunchKernelGGL((free_stream_kernel<true, false>), dim3(blocks), dim3(c->free_threads), lds, c->stream, fa, slots);
            else if (translate)
                hipLaunchKernelGGL((free_stream_kernel<false, true>), dim3(blocks), dim3(c->free_threads), lds, c->stream, fa, slots);
            else
                hipLaunchKernelGGL((free_stream_kernel<false, false>), dim3(blocks), dim3(c->free_threads), lds, c->stream, fa, slots);
            ktimer_end(c, kt);
            FreeLogArgs la{};
            la.log = fa.log;
            la.log_cnt = fa.log_cnt;
            la.log_cap = fa.log_cap;
            la.n_waves = n_waves;
            la.n_slices = n_slices;
            la.n_parts = n_parts;
            la.partial = c->f_partial.as<uint32_t>();
            la.part_used = c->f_part_used.as<uint32_t>();
            kt = ktimer_begin(c, "free_log");
            hipLaunchKernelGGL(free_log_kernel, dim3(n_parts * n_slices), dim3(kLogThreads), (size_t)kLogBins * 4, c->stream, la);
            ktimer_end(c, kt);
            kt = ktimer_begin(c, "free_counts");
            hipLaunchKernelGGL(free_counts_kernel, dim3((T.results + 256) / 256), dim3(256), 0, c->stream, fa.dense, fa.n_results,
                               T.rnode.as<int32_t>(), fa.job, fa.group, la.partial, la.part_used, n_slices, n_parts, table);
            ktimer_end(c, kt);
            HIP_TRY(c, hipGetLastError());
        }
        if (c->words_keep) return WK_OK;
        return words_reset(c);
    }
    ClassifyArgs a{};
    bool ok = false;
    int rc = words_jobs_ok(c, c->w_jobs.data(), (int32_t)c->w_jobs.size(), a, &ok);
    if (rc) return rc;
    // (`ok` may be false by now: a subject without an ancestor at a rank in use
    // joined the table after these records.  wk_words_begin stops accepting
    // records the moment that happens, so every subject the accumulated records
    // name has a valid row — rows of old subjects do not change when the table
    // grows — and the new ones carry no weight.)
    (void)ok;
    const int32_t n_jobs = (int32_t)c->w_jobs.size();
    const uint32_t cus = (uint32_t)c->prop.multiProcessorCount;
    // how many records every stream holds (known once per accumulation state)
    if (!c->w_counts_known) {
        HIP_TRY(c, hipMemcpyAsync(c->w_count, c->w_cursor.p, kMaxStreams * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->w_counts_known = true;
    }
    if ((size_t)c->n_subjects > c->w_hi_clean) {
        HIP_TRY(c, c->w_hi.reserve((size_t)c->n_subjects * 4 + ((size_t)c->n_subjects * 4) / 2));
        HIP_TRY(c, hipMemsetAsync(c->w_hi.p, 0, c->w_hi.cap, c->stream));
        c->w_hi_clean = c->w_hi.cap / 4;
    }
    WeighMergeArgs wm{};
    KernelTimer* kt = nullptr;
    if (c->w_sliced) {
        // ---- a stream per slice of the subject table, each read once; workgroups in proportion to the streams' lengths
        StreamBinsArgs sa{};
        const int S = c->w_streams;
        unsigned long long total = 0;
        for (int k = 0; k < S; ++k) total += c->w_count[k];
        uint32_t wg[kMaxStreams] = {}, used = 0;
        for (int k = 0; k < S; ++k)
            if (c->w_count[k]) {
                wg[k] = (uint32_t)std::max<unsigned long long>(1, c->w_count[k] * cus / total);
                used += wg[k];
            }
        while (used > cus) {  // (the minimum of one workgroup per stream pushed the sum over)
            int big = 0;
            for (int k = 1; k < S; ++k)
                if (wg[k] > wg[big]) big = k;
            --wg[big];
            --used;
        }
        while (used < cus && total) {  // what the rounding left: to the stream with the most records per workgroup
            int best = -1;
            for (int k = 0; k < S; ++k)
                if (wg[k] && (best < 0 || c->w_count[k] * wg[best] > c->w_count[best] * wg[k])) best = k;
            ++wg[best];
            ++used;
        }
        sa.n_streams = (uint32_t)S;
        sa.n_subjects = (uint32_t)c->n_subjects;
        uint32_t first = 0;
        for (int k = 0; k < S; ++k) {
            if (c->w_count[k] >= (1ull << 30)) return fail(c, WK_E_RANGE, "more than 2^30 records in one stream");
            sa.words[k] = c->w_stream[k].as<uint32_t>();
            sa.count[k] = (uint32_t)c->w_count[k];
            sa.wg_first[k] = first;
            first += wg[k];
        }
        for (int k = S; k <= kMaxStreams; ++k) sa.wg_first[k] = first;
        const uint32_t n_wg = first;
        HIP_TRY(c, c->w_slab.reserve((size_t)std::max(n_wg, 1u) * kSliceBins * 4));
        sa.slab = c->w_slab.as<uint32_t>();
        sa.hi = c->w_hi.as<uint32_t>();
        sa.err = scalar_err(c);
        kt = ktimer_begin(c, "classify");
        if (n_wg) {
            if (c->bins_ring == 6)
                hipLaunchKernelGGL((weigh_streams_kernel<6>), dim3(n_wg), dim3(kWeighThreads), kBinsMaxLds, c->stream, sa);
            else if (c->bins_ring == 8)
                hipLaunchKernelGGL((weigh_streams_kernel<8>), dim3(n_wg), dim3(kWeighThreads), kBinsMaxLds, c->stream, sa);
            else
                hipLaunchKernelGGL((weigh_streams_kernel<4>), dim3(n_wg), dim3(kWeighThreads), kBinsMaxLds, c->stream, sa);
        }
        ktimer_end(c, kt);
        wm.slab = sa.slab;
        wm.hi = sa.hi;
        wm.streams = (uint32_t)S;
        for (int k = 0; k <= kMaxStreams; ++k) wm.wg_first[k] = sa.wg_first[k];
        wm.bins = kSliceBins;
        wm.n_teams = 0;
    } else {
        // ---- more slices than streams: one stream, a team of workgroups per tile (round 3's launch)
        uint32_t w_xcd = 8;
        if (cus % w_xcd) w_xcd = 1;
        const int64_t cap = (int64_t)kBinsMaxLds / 4 - 96;
        const uint32_t w_slices = (uint32_t)((c->n_subjects + cap - 1) / cap);
        const uint32_t w_bins = ((uint32_t)c->n_subjects + w_slices - 1) / w_slices;
        const uint32_t w_teams = (cus / w_xcd) / w_slices;
        if (!w_teams) return fail(c, WK_E_RANGE, "subject table too large for the weighted histogram");
        const uint32_t n_teams = w_xcd * w_teams;
        HIP_TRY(c, c->w_slab.reserve((size_t)w_slices * n_teams * w_bins * 4));
        BinsArgs ba{};
        ba.subj = c->w_stream[0].as<int32_t>();
        ba.rk = nullptr;
        ba.n_records = (uint32_t)c->w_count[0];
        ba.n_subjects = (uint32_t)c->n_subjects;
        ba.bins = w_bins;
        ba.n_slices = w_slices;
        ba.teams_per_xcd = w_teams;
        ba.n_xcd = w_xcd;
        ba.slab = c->w_slab.as<uint32_t>();
        ba.hi = c->w_hi.as<uint32_t>();
        ba.err = scalar_err(c);
        const dim3 wgrid(cus);
        const size_t wlds = ((size_t)w_bins + 96) * 4;
        kt = ktimer_begin(c, "classify");
        if (c->bins_ring == 6)
            hipLaunchKernelGGL((weigh_bins_kernel<6, true>), wgrid, dim3(kWeighThreads), wlds, c->stream, ba);
        else if (c->bins_ring == 8)
            hipLaunchKernelGGL((weigh_bins_kernel<8, true>), wgrid, dim3(kWeighThreads), wlds, c->stream, ba);
        else
            hipLaunchKernelGGL((weigh_bins_kernel<4, true>), wgrid, dim3(kWeighThreads), wlds, c->stream, ba);
        ktimer_end(c, kt);
        wm.slab = ba.slab;
        wm.hi = ba.hi;
        wm.bins = w_bins;
        wm.n_teams = n_teams;
    }
    kt = ktimer_begin(c, "weigh_merge");
    wm.n_subjects = (uint32_t)c->n_subjects;
    wm.rows = a.rows;
    wm.row_w = a.row_w;
    wm.n_jobs = n_jobs;
    for (int j = 0; j < n_jobs; ++j) {
        wm.mode[j] = a.jobs[j].mode;
        wm.col[j] = a.jobs[j].col;
    }
    wm.group = (uint32_t)c->w_group;
    wm.table = CountTable{c->tkeys.as<unsigned long long>(), c->tvals.as<unsigned long long>(), c->slots - 1, scalar_err(c)};
    hipLaunchKernelGGL(weigh_merge_kernel, dim3((wm.n_subjects + kMergeSubjects - 1u) / kMergeSubjects), dim3(1024), (size_t)4096 * 16, c->stream, wm, 4096u);
    ktimer_end(c, kt);
    HIP_TRY(c, hipGetLastError());
    c->stat_extra_reads += c->w_reads;
    c->stat_extra_records += c->w_records;
    if (c->words_keep) return WK_OK;  // (bench: repeated passes over the resident batch)
    return words_reset(c);
}

int wk_words_begin(wk_ctx* c, const wk_job* jobs, int32_t n_jobs, int32_t group, int* ok) {
    if (!c || !ok) return WK_E_ARG;
    *ok = 0;
    if (!jobs || n_jobs < 1 || n_jobs > WK_MAX_JOBS) return fail(c, WK_E_ARG, "n_jobs must be in [1,%d]", WK_MAX_JOBS);
    if (group < 0 || group >= (1 << WK_KEY_GROUP_BITS)) return fail(c, WK_E_ARG, "group id outside [0, %d)", 1 << WK_KEY_GROUP_BITS);
    DeviceGuard guard(c->device);
    // records accumulated under other jobs / another group are classified first
    if (c->w_open && c->w_records > 0 && (group != c->w_group || !same_jobs(c->w_jobs, jobs, n_jobs))) {
        const int rc = wk_words_flush(c);
        if (rc) return rc;
    }
    ClassifyArgs a{};
    bool good = false;
    int mode = 0;
    int rc = words_jobs_ok(c, jobs, n_jobs, a, &good, &mode);
    if (rc) return rc;
    if (!good) {
        // (what is there was appended while every subject was valid: classify it now)
        if (c->w_open && c->w_records > 0 && (rc = wk_words_flush(c))) return rc;
        c->w_open = false;
        return WK_OK;
    }
    if (!c->w_open || c->w_records == 0) {  // a fresh accumulation
        if ((rc = words_reset(c))) return rc;
        c->w_sliced = mode == 0 && c->use_streams && streams_needed(c) <= kMaxStreams;
        c->w_streams = 0;
    }
    c->w_jobs.assign(jobs, jobs + n_jobs);
    c->w_group = group;
    c->w_mode = mode;
    c->w_open = true;
    *ok = 1;
    return WK_OK;
}

static StreamSet stream_set(wk_ctx* c) {
    StreamSet s{};
    size_t cap = ~(size_t)0;
    for (int k = 0; k < c->w_streams; ++k) {
        s.out[k] = c->w_stream[k].as<uint32_t>();
        cap = std::min(cap, c->w_stream[k].cap / 4);
    }
    s.cursor = c->w_cursor.as<unsigned long long>();
    s.n_streams = (uint32_t)c->w_streams;
    s.cap = (uint32_t)std::min<size_t>(cap, 0xFFFFFFFFu);
    return s;
}

// Single-job accumulation of the per-read stream: the first `count` records name their nodes by rank among the
// subjects' nodes.  When the subject table has grown since they were written, they are renumbered.
static int refresh_word_ranks(wk_ctx* c, int64_t count) {
    wk_ctx::StreamTables& T = c->st[0];
    std::vector<int32_t> before;
    const int rc = ensure_rank_table(c, T, c->w_mode, c->w_mode == 2 ? c->w_jobs[0].rank_slot : -1, &before);
    if (rc) return rc;
    if (count <= 0 || before.empty() || before == T.dn_host) return WK_OK;
    std::vector<int32_t> new_of_old(before.size());
    for (size_t i = 0; i < before.size(); ++i) {
        const auto it = std::lower_bound(T.dn_host.begin(), T.dn_host.end(), before[i]);
        if (it == T.dn_host.end() || *it != before[i]) return fail(c, WK_E_STATE, "a subject changed its node under accumulated records");
        new_of_old[i] = (int32_t)(it - T.dn_host.begin());
    }
    const int rcu = upload(c, c->w_renum, new_of_old.data(), new_of_old.size() * 4);
    if (rcu) return rcu;
    hipLaunchKernelGGL(ranks_renumber_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, c->c_words.as<uint32_t>(),
                       (uint32_t)count, c->w_renum.as<int32_t>(), (uint32_t)before.size());
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // (the vector is about to go out of scope)
    return WK_OK;
}

// ... and the subject fields of words [first, first + n), subject indices as the tokenizers write them, become such ranks.
static int words_translate(wk_ctx* c, int64_t first, int64_t n) {
    if ((c->w_mode != 1 && c->w_mode != 2) || n <= 0) return WK_OK;  // (several stream jobs: rewritten per job at the flush)
    const int rc = refresh_word_ranks(c, first);
    if (rc) return rc;
    hipLaunchKernelGGL(words_to_ranks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                       c->c_words.as<uint32_t>() + first, c->c_words.as<uint32_t>() + first, (uint32_t)n,
                       c->st[0].subj_rank.as<int32_t>(), (uint32_t)c->n_subjects, scalar_err(c));
    HIP_TRY(c, hipGetLastError());
    return WK_OK;
}

// Room for n_more records behind the accumulated ones.
static int words_room(wk_ctx* c, int64_t n_more) {
    if (c->w_mode == 0) {
        // every stream may take all of them (its cursor is only known to the device: at most w_records)
        if (!c->w_cursor.p) {
            HIP_TRY(c, c->w_cursor.reserve(kMaxStreams * 8));
            HIP_TRY(c, hipMemsetAsync(c->w_cursor.p, 0, kMaxStreams * 8, c->stream));
        }
        const int want = c->w_sliced ? streams_needed(c) : 1;
        c->w_streams = std::max(c->w_streams, want);
        const size_t need = (size_t)(c->w_records + n_more) * 4 + 64;
        const size_t room = std::max(need * 2, (size_t)std::max<int64_t>(c->w_expect, 0) * 4 + 64);
        for (int k = 0; k < c->w_streams; ++k) {
            DevBuf& b = c->w_stream[k];
            if (need <= b.cap) continue;
            DevBuf bigger;
            HIP_TRY(c, bigger.reserve(room));
            const size_t keep = std::min((size_t)c->w_records * 4, b.cap);
            if (keep > 0) HIP_TRY(c, hipMemcpyAsync(bigger.p, b.p, keep, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            b.release();
            b = bigger;
        }
        return WK_OK;
    }
    const size_t need = (size_t)(c->w_records + n_more) * 4 + 64;
    if (need <= c->c_words.cap) return WK_OK;
    // grow: a new buffer (twice the need, or what the sample is expected to bring) takes over what is there
    DevBuf bigger;
    HIP_TRY(c, bigger.reserve(std::max(need * 2, (size_t)std::max<int64_t>(c->w_expect, 0) * 4 + 64)));
    if (c->w_records > 0) HIP_TRY(c, hipMemcpyAsync(bigger.p, c->c_words.p, (size_t)c->w_records * 4, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->c_words.release();
    c->c_words = bigger;
    return WK_OK;
}

// The accumulated records would pass 2^30 (the histogram addresses bytes with
// 32 bits): classify what is there and open the same job set again.
static int words_roll(wk_ctx* c, int64_t n_more) {
    const bool outgrown = c->w_mode == 0 && c->w_sliced && streams_needed(c) > kMaxStreams;  // (continues unsliced)
    if (c->w_records + n_more < (1ll << 30) && !outgrown) return WK_OK;
    const std::vector<wk_job> jobs = c->w_jobs;
    const int32_t group = c->w_group;
    int rc = wk_words_flush(c);
    if (rc) return rc;
    int ok = 0;
    if ((rc = wk_words_begin(c, jobs.data(), (int32_t)jobs.size(), group, &ok))) return rc;
    if (!ok) return fail(c, WK_E_STATE, "job set no longer accepted");
    if (n_more >= (1ll << 30)) return fail(c, WK_E_RANGE, "more than 2^30 records in one block");
    return WK_OK;
}

int wk_words_append(wk_ctx* c, const uint32_t* words, int64_t n_records, int64_t n_reads, int slot) {
    if (!c) return WK_E_ARG;
    if (n_records < 0 || n_reads < 0 || (n_records > 0 && !words)) return fail(c, WK_E_ARG, "bad packed record arguments");
    if (!c->w_open) return fail(c, WK_E_STATE, "wk_words_begin has not accepted a job set");
    if (c->lag_count) return fail(c, WK_E_STATE, "blocks whose verdict has not been read (wk_dtok_scan_emit_end)");
    c->fz_chain = false;
    if (slot < -1 || slot >= wk_ctx::kStageSlots) return fail(c, WK_E_ARG, "slot must be -1 or in [0, %d)", wk_ctx::kStageSlots);
    DeviceGuard guard(c->device);
    {
        const int rcr = words_roll(c, n_records);
        if (rcr) return rcr;
    }
    {
        const int rcw = words_room(c, n_records);
        if (rcw) return rcw;
    }
    if (n_records > 0 && c->w_mode == 0) {
        HIP_TRY(c, c->w_stage.reserve((size_t)n_records * 4));
        HIP_TRY(c, hipMemcpyAsync(c->w_stage.p, words, (size_t)n_records * 4, hipMemcpyHostToDevice, c->stream));
    } else if (n_records > 0) {
        HIP_TRY(c, hipMemcpyAsync(c->c_words.as<uint32_t>() + c->w_records, words, (size_t)n_records * 4, hipMemcpyHostToDevice, c->stream));
    }
    if (slot >= 0) {
        if (!c->slot_ev[slot]) HIP_TRY(c, hipEventCreateWithFlags(&c->slot_ev[slot], hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(c->slot_ev[slot], c->stream));
        c->slot_busy[slot] = true;
    }
    if (n_records > 0 && c->w_mode == 0) {  // ... to the streams of their slices
        hipLaunchKernelGGL(words_partition_kernel, dim3((unsigned)((n_records + 256 * kScatterItems - 1) / (256 * kScatterItems))), dim3(256), 0, c->stream,
                           c->w_stage.as<uint32_t>(), (uint32_t)n_records, stream_set(c));
        HIP_TRY(c, hipGetLastError());
        c->w_counts_known = false;
    }
    {
        const int rct = words_translate(c, c->w_records, n_records);
        if (rct) return rct;
    }
    if (slot < 0) HIP_TRY(c, hipStreamSynchronize(c->stream));  // the caller's buffer is only valid during the call
    c->w_records += n_records;
    c->w_reads += n_reads;
    return WK_OK;
}

int wk_words_wait(wk_ctx* c, int slot) {
    if (!c || slot < 0 || slot >= wk_ctx::kStageSlots) return WK_E_ARG;
    if (!c->slot_busy[slot]) return WK_OK;
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipEventSynchronize(c->slot_ev[slot]));
    c->slot_busy[slot] = false;
    return WK_OK;
}

int wk_words_pending(wk_ctx* c, int64_t* n_records, int64_t* n_reads) {
    if (!c) return WK_E_ARG;
    if (n_records) *n_records = c->w_open ? c->w_records : 0;
    if (n_reads) *n_reads = c->w_open ? c->w_reads : 0;
    return WK_OK;
}

// ---- SAM tokenizer on the device --------------------------------------------------

// The dictionary of `tok` as the kernels probe it: open addressing, 16-byte slots
// {hash, id, offset}, the names behind 4-byte lengths in one arena.
static int dtok_mirror_dict(wk_ctx* c, const wk_tok* tok) {
    const int32_t n = wkx_tok_n_names(tok);
    if (c->dt_dict_tok == tok && c->dt_dict_names == n) return WK_OK;
    uint32_t slots = 1024;
    while (slots < 2u * (uint32_t)n + 2u) slots <<= 1;
    std::vector<DictSlot> tab(slots, DictSlot{0ull, -1, 0u});
    std::string arena;
    arena.reserve((size_t)n * 16 + 16);
    for (int32_t id = 0; id < n; ++id) {
        const char* p;
        uint32_t len;
        uint64_t hv;
        wkx_tok_name(tok, id, &p, &len, &hv);
        const uint32_t off = (uint32_t)arena.size();
        arena.append(reinterpret_cast<const char*>(&len), 4);
        arena.append(p, len);
        uint32_t h = (uint32_t)hv & (slots - 1);
        while (tab[h].id >= 0) h = (h + 1) & (slots - 1);
        tab[h] = DictSlot{(unsigned long long)hv, id, off};
    }
    arena.append(16, '\0');
    int rc;
    if ((rc = upload(c, c->d_dict, tab.data(), tab.size() * sizeof(DictSlot)))) return rc;
    if ((rc = upload(c, c->d_arena, arena.data(), arena.size()))) return rc;
    // (the same table for dtok_fused_kernel: 8-byte slots, the names by id)
    std::vector<DictSlot8> tab8(slots, DictSlot8{0u, -1});
    std::vector<unsigned char> names16((size_t)std::max(n, 1) * 16, 0);
    for (uint32_t h = 0; h < slots; ++h) {
        if (tab[h].id < 0) continue;
        tab8[h] = DictSlot8{(uint32_t)(tab[h].hash >> 32), tab[h].id};
        uint32_t len;
        std::memcpy(&len, arena.data() + tab[h].off, 4);
        unsigned char* rec = names16.data() + (size_t)tab[h].id * 16;
        if (len <= 15) {
            std::memcpy(rec, arena.data() + tab[h].off + 4, len);
            rec[15] = (unsigned char)len;
        } else {
            std::memcpy(rec, &tab[h].off, 4);
            rec[15] = 0xFF;
        }
    }
    if ((rc = upload(c, c->d_dict2, tab8.data(), tab8.size() * sizeof(DictSlot8)))) return rc;
    if ((rc = upload(c, c->d_names16, names16.data(), names16.size()))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->dt_dict_mask = slots - 1;
    c->dt_dict_names = n;
    c->dt_dict_tok = tok;
    return WK_OK;
}

static DtokArgs dtok_args(wk_ctx* c) {
    DtokArgs a{};
    a.text = c->dt_text;
    a.n = c->dt_n;
    a.fmt = (uint32_t)c->dt_fmt;
    a.line_start = c->d_lines.as<uint32_t>();
    a.n_lines = c->dt_lines;
    a.lsubj = c->d_lsubj.as<int32_t>();
    a.lmeta = c->d_lmeta.as<uint32_t>();
    a.dict = c->d_dict.as<DictSlot>();
    a.dict_mask = c->dt_dict_mask;
    a.arena = c->d_arena.as<unsigned char>();
    a.dict8 = c->d_dict2.as<DictSlot8>();
    a.names16 = c->d_names16.as<uint4>();
    a.unknown = c->d_unknown.as<uint2>();
    a.unknown_cap = (uint32_t)(c->d_unknown.cap / 8);
    a.is_start = c->d_start.as<unsigned char>();
    a.is_first = c->d_first.as<unsigned char>();
    a.state = c->d_state.as<DtokState>();
    a.lbeg = c->d_lbeg.as<int32_t>();
    a.lend = c->d_lend.as<int32_t>();
    a.llen = c->d_llen.as<uint32_t>();
    a.line_scan = c->d_lscan.as<unsigned long long>();
    a.submap = c->dt_submap_on ? c->d_submap.as<int32_t>() : nullptr;
    a.n_submap = c->dt_submap_n;
    return a;
}

// Start copying text[begin, stop) of a block to the device on the copy stream;
// the wk_dtok_scan of the same block finds it there.  At most one block ahead of
// the one being scanned (two buffers).
// hipMemcpyAsync host -> device of [src, src + n), cut where registered ranges
// (wk_host_register) begin and end: one copy may not span two registrations.
static hipError_t copy_text_async(wk_ctx* c, void* dst, const char* src, size_t n, hipStream_t stream) {
    std::vector<size_t> cuts{0, n};
    {
        std::lock_guard<std::mutex> lock(c->reg_mu);
        for (const wk_ctx::HostReg& r : c->regs) {
            if (r.p > src && r.p < src + n) cuts.push_back((size_t)(r.p - src));
            if (r.p + r.n > src && r.p + r.n < src + n) cuts.push_back((size_t)(r.p + r.n - src));
        }
    }
    std::sort(cuts.begin(), cuts.end());
    for (size_t i = 0; i + 1 < cuts.size(); ++i) {
        if (cuts[i + 1] == cuts[i]) continue;
        const hipError_t e = hipMemcpyAsync((char*)dst + cuts[i], src + cuts[i], cuts[i + 1] - cuts[i], hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

int wk_host_register(wk_ctx* c, const void* p, size_t bytes) {
    if (!c || !p || !bytes) return WK_E_ARG;
    DeviceGuard guard(c->device);
    const hipError_t e = hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterReadOnly);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(c, WK_E_HIP, "hipHostRegister(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    }
    std::lock_guard<std::mutex> lock(c->reg_mu);
    c->regs.push_back(wk_ctx::HostReg{(const char*)p, bytes});
    return WK_OK;
}

int wk_host_unregister(wk_ctx* c, const void* p) {
    if (!c || !p) return WK_E_ARG;
    {
        std::lock_guard<std::mutex> lock(c->reg_mu);
        auto it = std::find_if(c->regs.begin(), c->regs.end(), [&](const wk_ctx::HostReg& r) { return r.p == (const char*)p; });
        if (it == c->regs.end()) return fail(c, WK_E_ARG, "not a registered range");
        c->regs.erase(it);
    }
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipHostUnregister(const_cast<void*>(p)));
    return WK_OK;
}

static int dtok_copy_impl(wk_ctx* c, const char* text, int64_t begin, int64_t stop, bool detached, int32_t* ticket) {
    if (!c || !text || begin < 0 || stop < begin) return WK_E_ARG;
    if (ticket) *ticket = -1;
    Lap lap(&c->lap_s[0]);
    const int64_t n64 = stop - begin;
    if (n64 == 0 || n64 >= (1ll << 31) - 64) return WK_OK;
    auto lap_t = std::chrono::steady_clock::now();
    DeviceGuard guard(c->device);
    int k = -1;
    {
        std::lock_guard<std::mutex> lock(c->copy_mu);
        if (!c->copy_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        if (!c->count_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->count_stream, hipStreamNonBlocking));
        for (int q = 0; q < wk_ctx::kTextBufs && k < 0; ++q)
            if (c->buf_state[q] == wk_ctx::kBufFree) k = q;
        if (k >= 0) {
            c->buf_state[k] = wk_ctx::kBufCopied;
            c->copy_src[k] = nullptr;  // (tagged below, once the copy is queued)
            c->copy_seq[k] = c->copy_seq_next++;
        }
    }
    if (k < 0) return fail(c, WK_E_STATE, "more than %d blocks copied ahead of the scan", wk_ctx::kTextBufs - 1);
    auto lap_mark = [&](int i) {
        const auto now = std::chrono::steady_clock::now();
        const double dt = std::chrono::duration<double>(now - lap_t).count();
        c->lap_c[i] += dt;
        if (i == 1) c->lap_c[4] = std::max(c->lap_c[4], dt);
        lap_t = now;
    };
    if (!c->copy_ev[k]) HIP_TRY(c, hipEventCreate(&c->copy_ev[k]));
    if (!c->copy_ev0[k]) HIP_TRY(c, hipEventCreate(&c->copy_ev0[k]));
    if (!c->copy_evm[k]) HIP_TRY(c, hipEventCreate(&c->copy_evm[k]));
    const uint32_t n = (uint32_t)n64;
    // (at least a full block's worth from the start: a file's first blocks are small, and growing a buffer
    // three times means three hipFree / hipMalloc pairs per buffer while the dictionary is cold)
    lap_mark(0);
    HIP_TRY(c, text_buffer(c, k, (size_t)n + 64));
    lap_mark(1);
    HIP_TRY(c, hipEventRecord(c->copy_ev0[k], c->copy_stream));
    // (only the copy on this stream: the 64 zero bytes behind the text are a fill
    // kernel, which wk_dtok_scan launches on its own stream behind the copy's event)
    HIP_TRY(c, copy_text_async(c, c->d_textptr[k], text + begin, (size_t)n, c->copy_stream));
    HIP_TRY(c, hipEventRecord(c->copy_evm[k], c->copy_stream));
    lap_mark(2);
    // ... and, behind the copy, the count of the block's newlines -- on a stream of its own, so that the next
    // block's copy follows this one without a gap; the total lands in pinned memory (written by the kernel: no
    // trip through the DMA queue): wk_dtok_scan finds the number on the host instead of waiting for it
    HIP_TRY(c, hipStreamWaitEvent(c->count_stream, c->copy_evm[k], 0));
    c->copy_counted[k] = false;
    if (!c->fused_streak && c->count_ahead) {  // (blocks that go through the one kernel need no count: wk_dtok_fused.hpp)
        const uint32_t n_tiles = (n + kDtokTile - 1) / kDtokTile;
        HIP_TRY(c, c->d_tiles_k[k].reserve((size_t)n_tiles * 8));
        HIP_TRY(c, c->d_tile_off_k[k].reserve((size_t)n_tiles * 8));
        hipLaunchKernelGGL(dtok_count_kernel, dim3(n_tiles), dim3(kDtokThreads), 0, c->count_stream, c->d_textptr[k], n,
                           c->d_tiles_k[k].as<unsigned long long>());
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, c->count_stream, c->d_tiles_k[k].as<unsigned long long>(),
                           c->d_tile_off_k[k].as<unsigned long long>(), (int64_t)n_tiles, &c->copy_newlines[k]);
        HIP_TRY(c, hipGetLastError());
        c->copy_counted[k] = true;
    }
    HIP_TRY(c, hipEventRecord(c->copy_ev[k], c->count_stream));
    lap_mark(3);
    {
        std::lock_guard<std::mutex> lock(c->copy_mu);
        c->copy_n[k] = n;
        c->copy_last[k] = text[stop - 1];
        c->copy_detached[k] = detached;
        c->copy_src[k] = text + begin;
    }
    if (ticket) *ticket = k;
    return WK_OK;
}

int wk_dtok_copy(wk_ctx* c, const char* text, int64_t begin, int64_t stop) {
    return dtok_copy_impl(c, text, begin, stop, false, nullptr);
}

// The same copy for a reader that does not keep the host bytes until the block is scanned: `ticket` names the copy for
// wk_dtok_copy_wait, after which text[begin, stop) may be overwritten.  The scan of the block is called with the same
// (text, begin, stop) as ever -- the pointer is only the block's tag then; where it wants the bytes themselves (names
// of subjects the dictionary does not hold) they are fetched from the device, and wk_dtok_text_back hands the host
// layer the block's text when the kernels leave the block to the host tokenizer.
int wk_dtok_copy_ahead(wk_ctx* c, const char* text, int64_t begin, int64_t stop, int32_t* ticket) {
    if (!ticket) return WK_E_ARG;
    return dtok_copy_impl(c, text, begin, stop, true, ticket);
}

int wk_dtok_copy_wait(wk_ctx* c, int32_t ticket) {
    if (!c) return WK_E_ARG;
    if (ticket < 0) return WK_OK;  // (an empty block: nothing was copied)
    if (ticket >= wk_ctx::kTextBufs || !c->copy_evm[ticket]) return fail(c, WK_E_ARG, "not a ticket of wk_dtok_copy_ahead");
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipEventSynchronize(c->copy_evm[ticket]));
    return WK_OK;
}

// The text blocks scanned from now on belong to `text_bytes` bytes of one sample (0: unknown again): the buffers of its
// records are sized for all of them when the first block is emitted.
int wk_dtok_expect(wk_ctx* c, int64_t text_bytes) {
    if (!c || text_bytes < 0) return WK_E_ARG;
    c->dt_expect_bytes = text_bytes;
    c->w_expect = 0;
    return WK_OK;
}

// Forget the blocks copied ahead that no scan has asked for (the file goes another way after all).
int wk_dtok_copy_drop(wk_ctx* c) {
    if (!c) return WK_E_ARG;
    DeviceGuard guard(c->device);
    if (c->copy_stream) HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    if (c->count_stream) HIP_TRY(c, hipStreamSynchronize(c->count_stream));
    std::lock_guard<std::mutex> lock(c->copy_mu);
    for (int q = 0; q < wk_ctx::kTextBufs; ++q)
        if (c->buf_state[q] == wk_ctx::kBufCopied) {
            c->buf_state[q] = wk_ctx::kBufFree;
            c->copy_src[q] = nullptr;
            c->copy_counted[q] = false;
        }
    // (the slabs a reader far ahead of the scans took -- up to 12 GB -- go back to the device once none of their
    // buffers is in use: what follows the file, count tables and records, must not compete with parked text.
    // The first slab stays: a reader that keeps close to the scans lives in it)
    {
        std::lock_guard<std::mutex> slabs(c->slab_mu);
        for (int sl = 1; sl < wk_ctx::kTextBufs / wk_ctx::kSlabBufs; ++sl) {
            if (!c->d_textslab[sl].p) continue;
            bool idle = true;
            for (int q = sl * wk_ctx::kSlabBufs; q < (sl + 1) * wk_ctx::kSlabBufs; ++q) idle = idle && c->buf_state[q] == wk_ctx::kBufFree;
            if (!idle) continue;
            HIP_TRY(c, hipStreamSynchronize(c->stream));  // (the scan of a block that lived there)
            for (int q = sl * wk_ctx::kSlabBufs; q < (sl + 1) * wk_ctx::kSlabBufs; ++q)
                if (c->d_textptr[q] >= c->d_textslab[sl].as<unsigned char>() &&
                    c->d_textptr[q] < c->d_textslab[sl].as<unsigned char>() + wk_ctx::kTextStride * wk_ctx::kSlabBufs)
                    c->d_textptr[q] = nullptr;
            if (c->dt_text >= c->d_textslab[sl].as<unsigned char>() && c->dt_text < c->d_textslab[sl].as<unsigned char>() + c->d_textslab[sl].cap)
                c->dt_text = nullptr;  // (wk_dtok_text_back of that block: refused from now on)
            c->d_textslab[sl].release();
        }
    }
    return WK_OK;
}

// How many blocks a reader may copy ahead of the scans on this device as it is now: half of the memory that is free
// (slabs already taken for text count as free: they are reused), in text buffers of one block each.
int wk_dtok_ahead_room(wk_ctx* c, int32_t* n_blocks) {
    if (!c || !n_blocks) return WK_E_ARG;
    DeviceGuard guard(c->device);
    size_t free_b = 0, total_b = 0;
    HIP_TRY(c, hipMemGetInfo(&free_b, &total_b));
    size_t held = 0;
    {
        std::lock_guard<std::mutex> slabs(c->slab_mu);
        for (const DevBuf& b : c->d_textslab) held += b.p ? b.cap : 0;
    }
    const size_t room = (free_b + held) / 2 / wk_ctx::kTextStride;
    *n_blocks = (int32_t)std::min<size_t>(room, (size_t)wk_ctx::kTextBufs - 1);
    return WK_OK;
}

// The text of the block scanned last, as the device holds it: n = stop - begin bytes into `out`.
int wk_dtok_text_back(wk_ctx* c, char* out, int64_t n) {
    if (!c || !out || n < 0) return WK_E_ARG;
    if (!c->dt_text || (int64_t)c->dt_n != n) return fail(c, WK_E_STATE, "no scanned block of %lld bytes on the device", (long long)n);
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipMemcpyAsync(out, c->dt_text, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return WK_OK;
}

// (measurement) text[begin, stop) to the device, to stay: a later wk_dtok_scan / wk_dtok_scan_emit of the same
// bytes finds them there and copies nothing.
int wk_text_upload(wk_ctx* c, const char* text, int64_t begin, int64_t stop) {
    if (!c || !text || begin < 0 || stop <= begin || stop - begin >= (1ll << 31) - 64) return WK_E_ARG;
    DeviceGuard guard(c->device);
    const uint32_t n = (uint32_t)(stop - begin);
    const uint32_t n_tiles = (n + kDtokTile - 1) / kDtokTile;
    wk_ctx::ResidentText r{};
    r.host = text + begin;
    r.n = n;
    void *dev = nullptr, *tiles = nullptr, *off = nullptr;
    HIP_TRY(c, hipMalloc(&dev, (size_t)n + 64));
    HIP_TRY(c, hipMalloc(&tiles, (size_t)n_tiles * 8));
    HIP_TRY(c, hipMalloc(&off, (size_t)n_tiles * 8));
    HIP_TRY(c, hipMemcpyAsync(dev, r.host, n, hipMemcpyHostToDevice, c->stream));
    // (the pad behind the text: dtok_count_kernel's last tile; the total: tile_scan_kernel stores it)
    hipLaunchKernelGGL(dtok_count_kernel, dim3(n_tiles), dim3(kDtokThreads), 0, c->stream, (const unsigned char*)dev, n,
                       (unsigned long long*)tiles);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, c->stream, (const unsigned long long*)tiles, (unsigned long long*)off,
                       (int64_t)n_tiles, scalar_u64(c, 3));
    HIP_TRY(c, hipMemcpyAsync(&r.n_newlines, scalar_u64(c, 3), 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    (void)hipFree(tiles);
    r.dev = (unsigned char*)dev;
    r.tile_off = (unsigned long long*)off;
    c->resident.push_back(r);
    return WK_OK;
}

// (measurement) the rate of pinned host -> device copies of `bytes` each on this box, the way the text
// route makes them (hipMemcpyAsync from pinned memory on a stream of their own): bytes per second over
// `reps` copies back to back, timed with events.
int wk_h2d_rate(wk_ctx* c, int64_t bytes, int reps, double* bytes_per_s) {
    if (!c || bytes <= 0 || reps <= 0 || !bytes_per_s) return WK_E_ARG;
    DeviceGuard guard(c->device);
    void *h[2] = {nullptr, nullptr}, *d[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() {
        for (int i = 0; i < 2; ++i) {
            if (h[i]) (void)hipHostFree(h[i]);
            if (d[i]) (void)hipFree(d[i]);
        }
        if (st) (void)hipStreamDestroy(st);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipHostMalloc(&h[i], (size_t)bytes, hipHostMallocDefault);
        if (e == hipSuccess) std::memset(h[i], 0x41 + i, (size_t)bytes);
        if (e == hipSuccess) e = hipMalloc(&d[i], (size_t)bytes);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipMemcpyAsync(d[i], h[i], (size_t)bytes, hipMemcpyHostToDevice, st);  // warm
    if (e == hipSuccess) e = hipEventRecord(e0, st);
    for (int i = 0; i < reps && e == hipSuccess; ++i) e = hipMemcpyAsync(d[i & 1], h[i & 1], (size_t)bytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    cleanup();
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(c, WK_E_HIP, "wk_h2d_rate: %s", hipGetErrorString(e));
    }
    *bytes_per_s = ms > 0.f ? (double)bytes * reps / (ms * 1e-3) : 0.0;
    return WK_OK;
}

int wk_ordinal_chunk_counts(wk_ctx* c, int64_t* sorted, int64_t* gathered) {
    if (!c || !sorted || !gathered) return WK_E_ARG;
    *sorted = c->chunks_sorted;
    *gathered = c->chunks_gathered;
    return WK_OK;
}

// (measurement) blocks of the text route the fused kernel did / handed back to the six kernels since the context exists
int wk_dtok_fused_counts(wk_ctx* c, int64_t* done, int64_t* handed_back) {
    if (!c || !done || !handed_back) return WK_E_ARG;
    *done = c->fused_blocks;
    *handed_back = c->fused_fallbacks;
    return WK_OK;
}

int wk_text_clear(wk_ctx* c) {
    if (!c) return WK_E_ARG;
    DeviceGuard guard(c->device);
    for (wk_ctx::ResidentText& r : c->resident) {
        (void)hipFree(r.dev);
        (void)hipFree(r.tile_off);
    }
    c->resident.clear();
    return WK_OK;
}

// The tokenizer's ids are not the subject indices of the records (wk_set_subjects): map[id] is (`--trim-sub`, where
// several names are one subject).  n = 0: they are again.  Applies to the plain flavour's blocks from the next
// wk_dtok_emit / wk_dtok_scan_emit on; a block that meets an id beyond the map is the host tokenizer's.
int wk_dtok_subject_map(wk_ctx* c, const int32_t* map, int32_t n) {
    if (!c || n < 0 || (n > 0 && !map)) return WK_E_ARG;
    DeviceGuard guard(c->device);
    c->dt_submap_on = map != nullptr;  // (an empty map that is there: no name has been met yet)
    if (map && n == 0) HIP_TRY(c, c->d_submap.reserve(64));
    bool excl = false;
    if (n > 0) {
        for (int32_t i = 0; i < n; ++i) {
            if (map[i] == kLineExcluded)
                excl = true;
            else if (map[i] < 0)
                return fail(c, WK_E_ARG, "a subject map holds subject indices (or -4: excluded)");
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));  // (a kernel may be reading the map that is there)
        const int rc = upload(c, c->d_submap, map, (size_t)n * 4);
        if (rc) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    c->dt_submap_n = (uint32_t)n;
    c->dt_has_excl = excl;
    return WK_OK;
}

int wk_dtok_format(wk_ctx* c, int fmt) {
    if (!c) return WK_E_ARG;
    if (fmt != WK_FMT_SAM && fmt != WK_FMT_MAP && fmt != WK_FMT_B6O && fmt != WK_FMT_PAF)
        return fail(c, WK_E_ARG, "the device tokenizer takes SAM, simple maps, BLAST tabular text and PAF");
    c->dt_fmt = fmt;
    return WK_OK;
}

// a block the one kernel handed back: see dtok_scan_impl
static void fz_handed_back(wk_ctx* c) {
    c->fz_back_streak = std::min(c->fz_back_streak + 1, 6);
    if (c->fz_back_streak >= 2) c->fz_skip = 1 << (c->fz_back_streak - 1);
}

static DevBuf& fz_bk(wk_ctx* c, int i) { return i == 0 ? c->w_backup : i == 1 ? c->w_backup2 : c->w_backup3; }

static int dtok_emit_launch(wk_ctx* c, bool* ordered_out, unsigned long long* totals);
static int dtok_emit_finish(wk_ctx* c, bool keep, bool ordered, DtokState st, unsigned long long totals, int64_t* n_reads,
                            int64_t* n_records);

// `emit` (may be null): with the words of this sample open (wk_words_begin), the emission is queued right behind the
// parse — the usual block brings no subject the dictionary does not know — and the host waits once for both; *emit = 1
// when the block's records have been appended that way (then emitted[0..1] = reads, records).
static int dtok_scan_impl(wk_ctx* c, wk_tok* tok, const char* text, int64_t begin, int64_t stop, int extra, int64_t* n_lines, int* status,
                          int* emit, int64_t* emitted) {
    if (!c || !tok || !text || begin < 0 || stop < begin || !n_lines || !status) return WK_E_ARG;
    if (emit) *emit = 0;
    if (c->lag_count) return fail(c, WK_E_STATE, "blocks whose verdict has not been read (wk_dtok_scan_emit_end)");
    Lap lap(&c->lap_s[1]);
    *status = 1;
    *n_lines = 0;
    c->dt_ready = false;
    c->dt_extra = extra != 0;
    // (an exclusion set: the plain flavour takes it as kLineExcluded entries of the subject map, wk_dtok_subject_map;
    // the "ex" parsers' way with it -- align.py:481-547 yields a stale pool at the end of a file -- stays the host's)
    if (!wkx_tok_device_ok(tok) && (extra || !c->dt_submap_on)) return WK_OK;
    if (extra && c->dt_fmt == WK_FMT_MAP) return WK_OK;  // (a simple map has no "ex" flavour, align.py:236)
    const int64_t n64 = stop - begin;
    if (n64 >= (1ll << 31) - 64) return WK_OK;
    DeviceGuard guard(c->device);
    KtScope kt_scope(c);
    const uint32_t n = (uint32_t)n64;
    c->dt_n = n;
    c->dt_lines = 0;
    if (n == 0) {
        *status = 0;
        c->dt_ready = true;
        return WK_OK;
    }
    const char* src = text + begin;
    // the block's text: copied ahead by wk_dtok_copy (then the kernels only wait
    // for that copy), or copied now
    int k = -1;
    const unsigned char* resident = nullptr;
    const wk_ctx::ResidentText* res = nullptr;
    for (const wk_ctx::ResidentText& r : c->resident)
        if (r.host == src && r.n == n) {
            resident = r.dev;
            res = &r;
        }
    bool counted = false, counted_or_ahead = false;
    auto lap_t = std::chrono::steady_clock::now();
    auto lap_mark = [&](int i) {
        const auto now = std::chrono::steady_clock::now();
        c->lap_x[i] += std::chrono::duration<double>(now - lap_t).count();
        lap_t = now;
    };
    {
        std::lock_guard<std::mutex> lock(c->copy_mu);
        // (the buffer of the block scanned before is free from here on: its kernels have been waited for)
        for (int q = 0; q < wk_ctx::kTextBufs; ++q)
            if (c->buf_state[q] == wk_ctx::kBufScanning) c->buf_state[q] = wk_ctx::kBufFree;
        // (the oldest copy with this tag: a reader that reuses its pinned buffers may have copied two blocks
        // of one length from one address)
        for (int q = 0; q < wk_ctx::kTextBufs && !resident; ++q)
            if (c->buf_state[q] == wk_ctx::kBufCopied && c->copy_src[q] == src && c->copy_n[q] == n &&
                (k < 0 || c->copy_seq[q] < c->copy_seq[k]))
                k = q;
        if (!resident && k < 0) {  // not copied ahead: a free buffer, copied into below
            for (int q = 0; q < wk_ctx::kTextBufs && k < 0; ++q)
                if (c->buf_state[q] == wk_ctx::kBufFree) k = q;
            if (k < 0) return fail(c, WK_E_STATE, "every text buffer holds a block copied ahead");
            c->copy_counted[k] = false;
            c->copy_src[k] = nullptr;
            c->copy_detached[k] = false;
            k = -1 - k;  // (marks "copy now")
        } else if (!resident) {
            counted = c->copy_counted[k];
            counted_or_ahead = true;
        }
        c->dt_detached = !resident && k >= 0 && c->copy_detached[k];
        if (!resident) {
            const int kk = k >= 0 ? k : -1 - k;
            c->buf_state[kk] = wk_ctx::kBufScanning;
            c->copy_src[kk] = nullptr;  // (the buffer's tag is used up)
            c->copy_counted[kk] = false;
        }
    }
    if (resident) {
        // (measurement: the block is on the device already, 64 zero bytes behind it)
    } else if (k >= 0) {
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->copy_ev[k], 0));  // (the pad: zeroed by the count behind the copy)
    } else {
        k = -1 - k;
        HIP_TRY(c, text_buffer(c, k, (size_t)n + 64));
        HIP_TRY(c, copy_text_async(c, c->d_textptr[k], src, n, c->stream));  // (the pad: zeroed by the count below)
    }
    if (!resident) c->dt_cur = k;
    lap_mark(0);
    c->dt_text = resident ? resident : c->d_textptr[k];
    // line starts: newlines per tile -> offsets -> positions
    const uint32_t n_tiles = (n + kDtokTile - 1) / kDtokTile;
    HIP_TRY(c, c->d_state.reserve(sizeof(DtokState) + 64));
    // (one kernel for the block, wk_dtok_fused.hpp: decided here, the words' mode checked again below)
    bool fuse = emit && !extra && c->use_fused && c->dt_fmt == WK_FMT_SAM && c->w_open && c->w_mode == 0 && !c->dt_keep_reads &&
                wkx_tok_n_names(tok) < (1 << 23) - 1;
    // (text the one kernel keeps handing back -- runs longer than its window, say -- pays for both ways block after
    // block: after two blocks in a row it is left out for 2, 4, ... 32 blocks before it is tried again)
    if (fuse && c->fz_skip > 0) {
        --c->fz_skip;
        fuse = false;
    }
    // The six kernels need the block's newlines per tile before anything else; the one kernel counts its lines itself.
    // A block copied ahead while blocks went through the one kernel (`fused_streak`) was not counted behind its copy:
    // it is counted here only if it turns out to need the six kernels after all -- or if the sample's record buffers
    // are still to be sized from its lines per byte (wk_dtok_expect).
    const bool ahead = !resident && counted_or_ahead;
    KernelTimer* kt = ktimer_begin(c, "dtok_lines");
    unsigned long long n_newlines = 0;
    const unsigned long long* tile_off = nullptr;
    bool have_count = false;
    auto count_now = [&]() -> int {
        HIP_TRY(c, c->d_tiles.reserve((size_t)n_tiles * 8));
        HIP_TRY(c, c->d_tile_off.reserve((size_t)n_tiles * 8));
        hipLaunchKernelGGL(dtok_count_kernel, dim3(n_tiles), dim3(kDtokThreads), 0, c->stream, c->dt_text, n,
                           c->d_tiles.as<unsigned long long>());
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_tiles.as<unsigned long long>(),
                           c->d_tile_off.as<unsigned long long>(), (int64_t)n_tiles, scalar_u64(c, 3));
        HIP_TRY(c, hipMemcpyAsync(&n_newlines, scalar_u64(c, 3), 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        tile_off = c->d_tile_off.as<unsigned long long>();
        have_count = true;
        return WK_OK;
    };
    const bool want_count = !fuse || (c->dt_expect_bytes > 0 && c->w_expect == 0);
    if (res) {
        // (the product counts a block's newlines behind its copy -- when the block is to take the six kernels -- and
        // never waits for them; here the same two kernels run in front of the scan, their result was taken at upload)
        if (!fuse) {
            HIP_TRY(c, c->d_tiles.reserve((size_t)n_tiles * 8));
            HIP_TRY(c, c->d_tile_off.reserve((size_t)n_tiles * 8));
            hipLaunchKernelGGL(dtok_count_kernel, dim3(n_tiles), dim3(kDtokThreads), 0, c->stream, c->dt_text, n,
                               c->d_tiles.as<unsigned long long>());
            hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_tiles.as<unsigned long long>(),
                               c->d_tile_off.as<unsigned long long>(), (int64_t)n_tiles, scalar_u64(c, 3));
        }
        n_newlines = res->n_newlines;
        tile_off = res->tile_off;
        have_count = true;
    } else if (counted) {  // (counted behind the copy: the number is on the host once the copy's event has passed)
        {
            Lap wait(&c->lap_s[3]);
            HIP_TRY(c, hipEventSynchronize(c->copy_ev[k]));
        }
        {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, c->copy_ev0[k], c->copy_evm[k]) == hipSuccess) {
                c->lap_copy_ms += ms;
                c->lap_copy_bytes += n;
            } else {
                (void)hipGetLastError();
            }
        }
        n_newlines = c->copy_newlines[k];
        tile_off = c->d_tile_off_k[k].as<unsigned long long>();
        have_count = true;
    } else if (want_count) {
        const int rc = count_now();
        if (rc) return rc;
    }
    // a last line without newline (the byte was noted when the block was copied ahead: the host bytes of such a
    // block may be gone)
    const bool open_end = (ahead ? c->copy_last[k] : src[n - 1]) != '\n';
    // (not counted: no more lines than bytes / 7 -- a mapped record is "q\tf\tr\t\n" at least -- and, once the
    // sample has shown its lines per byte, a quarter more than that; the kernel reports records that find no room)
    uint32_t lines = (uint32_t)n_newlines + (open_end ? 1u : 0u);
    if (!have_count) {
        double est = (double)n / 7.0 + 1.0;
        if (c->dt_lpb > 0.0) est = std::min(est, (double)n * c->dt_lpb * 1.25 + 65536.0);
        lines = (uint32_t)est;
    }
    if (c->d_unknown.cap < (size_t)(1 << 20) * 8) HIP_TRY(c, c->d_unknown.reserve((size_t)(1 << 20) * 8));
    // The usual block of a sample whose words are open -- SAM text, records for the weighted histogram, every subject
    // known -- goes through ONE kernel (wk_dtok_fused.hpp).  A block it hands back (a subject the dictionary does not
    // hold, a line for the host tokenizer, a run or a line beyond its window) leaves the streams as they were and
    // takes the six kernels below.
    if (fuse) {
        c->dt_lines = lines;
        int rc = dtok_mirror_dict(c, tok);
        if (rc) return rc;
        if (c->dt_expect_bytes > 0 && c->w_expect == 0) {   // (have_count: see want_count)
            const double expect = (double)c->dt_expect_bytes * ((double)lines / (double)n) * 1.08 + (double)lines;
            c->w_expect = (int64_t)std::min(expect, (double)(1ll << 30));
        }
        if ((rc = words_roll(c, lines))) return rc;
        if ((rc = words_room(c, lines))) return rc;
        FusedArgs fa{};
        fa.streams = stream_set(c);
        if (c->w_mode == 0 && fa.streams.n_streams <= kFzStreams) {   // (words_roll may have reopened the job set)
            fa.text = c->dt_text;
            fa.n = n;
            fa.open_end = open_end ? 1u : 0u;
            // (tiles of equal size, as many as make every workgroup's share the same number of rounds: 4096 tiles
            // of 16 KB over 768 workgroups are 5.33 rounds paid as 6 -- 6 rounds of 14.2 KB tiles do the same work in
            // 0.89 of the time)
            const unsigned wgs = (unsigned)(c->prop.multiProcessorCount * c->fused_per_cu);
            {
                const uint32_t rounds = (uint32_t)(((uint64_t)n + (uint64_t)wgs * kFzTile - 1) / ((uint64_t)wgs * kFzTile));
                uint32_t tile = (uint32_t)(((uint64_t)n + (uint64_t)wgs * rounds - 1) / ((uint64_t)wgs * rounds));
                tile = (tile + 15u) & ~15u;
                fa.tile = std::min<uint32_t>(kFzTile, std::max<uint32_t>(tile, 4096u));
            }
            fa.n_tiles = (n + fa.tile - 1) / fa.tile;
            fa.dict8 = c->d_dict2.as<DictSlot8>();
            fa.names16 = c->d_names16.as<uint4>();
            fa.dict_mask = c->dt_dict_mask;
            fa.arena = c->d_arena.as<unsigned char>();
            fa.unknown = c->d_unknown.as<uint2>();
            fa.unknown_cap = (uint32_t)(c->d_unknown.cap / 8);
            fa.state = c->d_state.as<DtokState>();
            fa.ablate = c->fused_ablate;
            fa.submap = c->dt_submap_on ? c->d_submap.as<int32_t>() : nullptr;
            fa.n_submap = c->dt_submap_n;
            c->w_counts_known = false;
            HIP_TRY(c, c->w_backup.reserve(kMaxStreams * 8));
            HIP_TRY(c, c->w_backup2.reserve(kMaxStreams * 8));
            // ONE launch per block while blocks follow one another through this kernel: its last workgroup leaves the
            // cursors as they are behind the block in the other of two buffers -- the next block's "before" --, the
            // block's scalars in pinned memory and cleared ones on the device.  (Anything else that moves the cursors
            // or the scalars in between breaks the chain: the small kernel in front, then.)
            HIP_TRY(c, c->w_backup3.reserve(kMaxStreams * 8));
            DevBuf& before = fz_bk(c, c->fz_parity);
            DevBuf& after = fz_bk(c, (c->fz_parity + 1) % 3);
            fa.backup_next = after.as<unsigned long long>();
            fa.host_state = reinterpret_cast<DtokState*>(c->host_back);   // (slot 0 of the pinned scratch)
            c->w_backup_cur = before.p;
            if (res) ktimer_end(c, kt);   // (the count behind a resident block's "copy": its own family)
            KernelTimer* kf = ktimer_begin(c, "dtok_fused");
            if (!c->fz_chain || c->fz_no_chain)
                hipLaunchKernelGGL(dtok_fused_begin_kernel, dim3(1), dim3(64), 0, c->stream, before.as<unsigned long long>(),
                                   (const unsigned long long*)fa.streams.cursor, fa.state);
            const unsigned grid = std::min<unsigned>(fa.n_tiles, wgs);
            hipLaunchKernelGGL(dtok_fused_kernel, dim3(grid), dim3(kFzThreads), 0, c->stream, fa);
            ktimer_end(c, kf);
            HIP_TRY(c, hipGetLastError());
            lap_mark(2);
            DtokState st{};
            {
                Lap wait(&c->lap_s[2]);
                HIP_TRY(c, hipStreamSynchronize(c->stream));
            }
            small_back_get(c, 0, &st, sizeof st);
            lap_t = std::chrono::steady_clock::now();
            const bool keep = st.flags == 0 && st.n_unknown == 0;
            int64_t nr = 0, nrec = 0;
            c->dt_emitted = false;
            if ((rc = dtok_emit_finish(c, keep, false, st, 0, &nr, &nrec))) return rc;   // (not kept: the cursors go back)
            lap_mark(4);
            c->fused_streak = keep;
            c->fz_chain = keep;   // (not kept: dtok_emit_finish has put the cursors back)
            if (keep) {
                c->fz_parity = (c->fz_parity + 1) % 3;
                c->fz_back_streak = 0;
                ++c->fused_blocks;
                lines = (uint32_t)st.n_lines + (open_end ? 1u : 0u);
                c->dt_lines = lines;
                c->dt_lpb = (double)lines / (double)n;
                *status = 0;
                *n_lines = lines;
                *emit = 1;
                emitted[0] = nr;
                emitted[1] = nrec;
                c->dt_ready = false;
                return WK_OK;
            }
            ++c->fused_fallbacks;
            fz_handed_back(c);
            kt = ktimer_begin(c, "dtok_lines");
        }
    }
    if (!have_count) {  // (the six kernels after all)
        const int rc = count_now();
        if (rc) return rc;
        lines = (uint32_t)n_newlines + (open_end ? 1u : 0u);
    }
    c->fused_streak = false;
    // (sized for a full block of short lines from the first block on: a file's first blocks are small, and
    // every growth is a hipFree -- which waits for the device -- and a hipMalloc per array)
    const size_t cap_lines = std::max<size_t>(lines, (size_t)1 << 21);
    HIP_TRY(c, c->d_lines.reserve((cap_lines + 2) * 4));
    HIP_TRY(c, c->d_lsubj.reserve((cap_lines + 1) * 4));
    HIP_TRY(c, c->d_lmeta.reserve((cap_lines + 1) * 4));
    HIP_TRY(c, c->d_start.reserve(cap_lines + 64));
    HIP_TRY(c, c->d_first.reserve(cap_lines + 64));
    if (extra) {
        HIP_TRY(c, c->d_lbeg.reserve((cap_lines + 1) * 4));
        HIP_TRY(c, c->d_lend.reserve((cap_lines + 1) * 4));
        HIP_TRY(c, c->d_llen.reserve((cap_lines + 1) * 4));
        HIP_TRY(c, c->d_lscan.reserve((cap_lines + 1) * 8));
    }
    // (line 0 and the block's scalars are written by the lines kernel itself)
    hipLaunchKernelGGL(dtok_lines_kernel, dim3(n_tiles), dim3(kDtokThreads), 0, c->stream, c->dt_text, n,
                       tile_off, c->d_lines.as<uint32_t>(), c->d_state.as<DtokState>());
    if (open_end) {
        const uint32_t end = n + 1;  // as if a newline followed the text
        HIP_TRY(c, hipMemcpyAsync(c->d_lines.as<uint32_t>() + lines, &end, 4, hipMemcpyHostToDevice, c->stream));
    }
    ktimer_end(c, kt);
    HIP_TRY(c, hipGetLastError());
    c->dt_lines = lines;
    lap_mark(1);
    // parse; subjects the dictionary does not know are interned in text order, then once more
    for (int round = 0; round < 3; ++round) {
        int rc = dtok_mirror_dict(c, tok);
        if (rc) return rc;
        if (round > 0) HIP_TRY(c, hipMemsetAsync(c->d_state.p, 0, sizeof(DtokState), c->stream));  // (round 0: by dtok_lines_kernel)
        const DtokArgs a = dtok_args(c);
        c->dt_mapped = false;
        kt = ktimer_begin(c, "dtok_parse");
        if (extra)
            hipLaunchKernelGGL(dtok_parse_kernel<true>, dim3((lines + kDtokThreads - 1) / kDtokThreads), dim3(kDtokThreads), 0, c->stream, a);
        else
            hipLaunchKernelGGL(dtok_parse_kernel<false>, dim3((lines + kDtokThreads - 1) / kDtokThreads), dim3(kDtokThreads), 0, c->stream, a);
        ktimer_end(c, kt);
        HIP_TRY(c, hipGetLastError());
        lap_mark(2);
        const bool speculate = emit && round == 0 && !extra && c->w_open && lines > 0;
        bool ordered = false;
        unsigned long long totals = 0;
        if (speculate && (rc = dtok_emit_launch(c, &ordered, &totals))) return rc;
        lap_mark(3);
        DtokState st{};
        static_assert(sizeof(DtokState) <= (size_t)wk_ctx::kBackBytes && sizeof(DtokState) % 4 == 0, "DtokState fits a slot");
        HIP_TRY(c, small_back(c, 0, c->d_state.p, sizeof st));
        {
            Lap wait(&c->lap_s[2]);
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        small_back_get(c, 0, &st, sizeof st);
        if (speculate && ordered) small_back_get(c, 1, &totals, 8);
        lap_t = std::chrono::steady_clock::now();
        if (speculate) {
            const bool keep = st.flags == 0 && st.n_unknown == 0;
            int64_t nr = 0, nrec = 0;
            if ((rc = dtok_emit_finish(c, keep, ordered, st, totals, &nr, &nrec))) return rc;
            lap_mark(4);
            if (keep) {
                *status = 0;
                *n_lines = lines;
                *emit = 1;
                emitted[0] = nr;
                emitted[1] = nrec;
                c->dt_ready = false;  // (nothing left for wk_dtok_emit)
                return WK_OK;
            }
        }
        if (st.flags) return WK_OK;  // status 1: the host tokenizer takes the block
        if (st.n_unknown == 0) {
            *status = 0;
            *n_lines = lines;
            c->dt_ready = true;
            return WK_OK;
        }
        lap_t = std::chrono::steady_clock::now();
        std::vector<uint2> unk(st.n_unknown);
        HIP_TRY(c, hipMemcpyAsync(unk.data(), c->d_unknown.p, (size_t)st.n_unknown * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        std::sort(unk.begin(), unk.end(), [](const uint2& x, const uint2& y) { return x.x < y.x; });
        if (!c->dt_detached) {
            for (const uint2& u : unk) (void)wkx_tok_intern(tok, src + u.x, u.y);
        } else {
            // the names come back from the device: one by one while they are few, with the whole block otherwise
            const bool whole = unk.size() > 128;
            size_t need = 0;
            for (const uint2& u : unk) need += u.y;
            std::vector<char> back(whole ? (size_t)n : need);
            if (whole) {
                HIP_TRY(c, hipMemcpyAsync(back.data(), c->dt_text, n, hipMemcpyDeviceToHost, c->stream));
            } else {
                size_t at = 0;
                for (const uint2& u : unk) {
                    if (u.y) HIP_TRY(c, hipMemcpyAsync(back.data() + at, c->dt_text + u.x, u.y, hipMemcpyDeviceToHost, c->stream));
                    at += u.y;
                }
            }
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            size_t at = 0;
            for (const uint2& u : unk) {
                (void)wkx_tok_intern(tok, back.data() + (whole ? (size_t)u.x : at), u.y);
                at += u.y;
            }
        }
        lap_mark(5);
    }
    return fail(c, WK_E_STATE, "device tokenizer: subjects still unknown after interning them");
}

// ---- the verdict of a block read one block late ------------------------------------------------------------------
// wk_dtok_scan_emit waits for a block's kernel before the next block's is launched: the device idles while the host
// wakes up, reads the verdict and launches again (14 of a block's 138 us with the text resident).  _begin launches a
// block's one-kernel tokenizer and returns; _end waits for the OLDEST block launched that way and says what became of
// it.  With two blocks under way the second one's kernel starts the moment the first one's ends.  A block the kernel
// hands back (*status = 2) has left no record -- nor has the block launched behind it, which is forgotten: the caller
// scans both again the synchronous way (their text is where it was).
int wk_dtok_scan_emit_begin(wk_ctx* c, wk_tok* tok, const char* text, int64_t begin, int64_t stop, int* started) {
    if (!c || !tok || !text || begin < 0 || stop < begin || !started) return WK_E_ARG;
    *started = 0;
    const int64_t n64 = stop - begin;
    if (!c->lag_enabled || !c->use_fused || c->lag_count >= 2 || c->fz_skip > 0 || c->dt_fmt != WK_FMT_SAM || !c->w_open || c->w_mode != 0 || c->dt_keep_reads ||
        n64 <= 0 || n64 >= (1ll << 31) - 64 || wkx_tok_n_names(tok) >= (1 << 23) - 1)
        return WK_OK;
    if (!wkx_tok_device_ok(tok) && !c->dt_submap_on) return WK_OK;
    if (c->dt_expect_bytes > 0 && c->w_expect == 0) return WK_OK;   // (the sample's buffers are sized from a counted block)
    if (c->lag_count > 0 && !c->fz_chain) return WK_OK;
    DeviceGuard guard(c->device);
    const uint32_t n = (uint32_t)n64;
    const char* src = text + begin;
    const wk_ctx::ResidentText* res = nullptr;
    for (const wk_ctx::ResidentText& r : c->resident)
        if (r.host == src && r.n == n) res = &r;
    int k = -1;
    uint32_t lines = 0;
    bool open_end;
    {
        std::lock_guard<std::mutex> lock(c->copy_mu);
        if (c->lag_count == 0)
            for (int q = 0; q < wk_ctx::kTextBufs; ++q)
                if (c->buf_state[q] == wk_ctx::kBufScanning) c->buf_state[q] = wk_ctx::kBufFree;
        if (!res) {
            for (int q = 0; q < wk_ctx::kTextBufs; ++q)
                if (c->buf_state[q] == wk_ctx::kBufCopied && c->copy_src[q] == src && c->copy_n[q] == n && (k < 0 || c->copy_seq[q] < c->copy_seq[k]))
                    k = q;
            if (k < 0) return WK_OK;   // (not copied ahead: the synchronous call copies it)
        }
    }
    if (res) {
        open_end = src[n - 1] != '\n';
        lines = (uint32_t)res->n_newlines + (open_end ? 1u : 0u);
    } else {
        if (c->dt_lpb <= 0.0) return WK_OK;
        open_end = c->copy_last[k] != '\n';
        lines = (uint32_t)std::min((double)n / 7.0 + 1.0, (double)n * c->dt_lpb * 1.25 + 65536.0);
    }
    // room for the records of every block under way (none of them is in w_records yet); a buffer that would have to
    // grow while a kernel appends to it, or a sample that would have to be rolled, waits for the verdicts
    int64_t under_way = lines;
    for (int i = 0; i < c->lag_count; ++i) under_way += c->lag[i].lines_est;
    if (c->w_records + under_way >= (1ll << 30) - (1 << 20)) return WK_OK;
    if (c->w_sliced && streams_needed(c) > kMaxStreams) return WK_OK;
    if (c->lag_count == 0) {
        const int rcw = words_room(c, under_way);
        if (rcw) return rcw;
    } else {
        const int want = c->w_sliced ? streams_needed(c) : 1;
        const size_t need = (size_t)(c->w_records + under_way) * 4 + 64;
        bool fits = c->w_cursor.p != nullptr && c->w_streams >= want;
        for (int q = 0; fits && q < c->w_streams; ++q) fits = need <= c->w_stream[q].cap;
        if (!fits) return WK_OK;
    }
    int rc = dtok_mirror_dict(c, tok);
    if (rc) return rc;
    FusedArgs fa{};
    fa.streams = stream_set(c);
    if (fa.streams.n_streams > kFzStreams) return WK_OK;
    if (!res) {
        std::lock_guard<std::mutex> lock(c->copy_mu);
        c->buf_state[k] = wk_ctx::kBufPending;
        c->copy_src[k] = nullptr;
        c->copy_counted[k] = false;
    }
    if (!res) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->copy_ev[k], 0));
    HIP_TRY(c, c->d_state.reserve(sizeof(DtokState) + 64));
    if (c->d_unknown.cap < (size_t)(1 << 20) * 8) HIP_TRY(c, c->d_unknown.reserve((size_t)(1 << 20) * 8));
    for (int i = 0; i < 3; ++i) HIP_TRY(c, fz_bk(c, i).reserve(kMaxStreams * 8));
    const unsigned wgs = (unsigned)(c->prop.multiProcessorCount * c->fused_per_cu);
    fa.text = res ? res->dev : c->d_textptr[k];
    fa.n = n;
    fa.open_end = open_end ? 1u : 0u;
    {
        const uint32_t rounds = (uint32_t)(((uint64_t)n + (uint64_t)wgs * kFzTile - 1) / ((uint64_t)wgs * kFzTile));
        uint32_t tile = (uint32_t)(((uint64_t)n + (uint64_t)wgs * rounds - 1) / ((uint64_t)wgs * rounds));
        tile = (tile + 15u) & ~15u;
        fa.tile = std::min<uint32_t>(kFzTile, std::max<uint32_t>(tile, 4096u));
    }
    fa.n_tiles = (n + fa.tile - 1) / fa.tile;
    fa.dict8 = c->d_dict2.as<DictSlot8>();
    fa.names16 = c->d_names16.as<uint4>();
    fa.dict_mask = c->dt_dict_mask;
    fa.arena = c->d_arena.as<unsigned char>();
    fa.unknown = c->d_unknown.as<uint2>();
    fa.unknown_cap = (uint32_t)(c->d_unknown.cap / 8);
    fa.state = c->d_state.as<DtokState>();
    fa.ablate = c->fused_ablate;
    fa.submap = c->dt_submap_on ? c->d_submap.as<int32_t>() : nullptr;
    fa.n_submap = c->dt_submap_n;
    wk_ctx::LagSlot& L = c->lag[c->lag_count];
    L.buf = res ? -1 : k;
    L.src = src;
    L.n = n;
    L.open_end = open_end;
    L.ring = c->fz_parity;
    L.lines_est = lines;
    L.host_slot = 2 + c->lag_next_ev;
    if (!c->lag_ev[c->lag_next_ev]) HIP_TRY(c, hipEventCreateWithFlags(&c->lag_ev[c->lag_next_ev], hipEventDisableTiming));
    L.ev = c->lag_ev[c->lag_next_ev];
    c->lag_next_ev ^= 1;
    fa.backup_next = fz_bk(c, (L.ring + 1) % 3).as<unsigned long long>();
    fa.host_state = reinterpret_cast<DtokState*>(c->host_back + (size_t)L.host_slot * wk_ctx::kBackBytes);
    static_assert(sizeof(DtokState) <= 128 && wk_ctx::kBackBytes >= 132, "the slot's second half holds the block's sequence number");
    L.seq = 0;
    if (c->lag_poll) {
        L.seq = ++c->lag_seq ? c->lag_seq : ++c->lag_seq;
        fa.host_seq = reinterpret_cast<uint32_t*>(c->host_back + (size_t)L.host_slot * wk_ctx::kBackBytes + 128);
        fa.seq = L.seq;
    }
    c->w_counts_known = false;
    KtScope kt_scope(c);   // (wk_profile_kernels: a block queued behind another one is bracketed from that one's end to its own)
    KernelTimer* kf = ktimer_begin(c, "dtok_fused");
    if (!c->fz_chain || c->fz_no_chain)
        hipLaunchKernelGGL(dtok_fused_begin_kernel, dim3(1), dim3(64), 0, c->stream, fz_bk(c, L.ring).as<unsigned long long>(),
                           (const unsigned long long*)fa.streams.cursor, fa.state);
    hipLaunchKernelGGL(dtok_fused_kernel, dim3(std::min<unsigned>(fa.n_tiles, wgs)), dim3(kFzThreads), 0, c->stream, fa);
    ktimer_end(c, kf);
    HIP_TRY(c, hipGetLastError());
    if (!c->lag_poll) HIP_TRY(c, hipEventRecord(L.ev, c->stream));
    // (the next block is launched as if this one will be kept: its kernel leaves what that needs behind)
    c->fz_chain = true;
    c->fz_parity = (c->fz_parity + 1) % 3;
    ++c->lag_count;
    *started = 1;
    return WK_OK;
}

// *status: 0 = the oldest block under way has been appended (*n_reads, *n_records, *n_lines), 2 = the kernel handed it
// back: no block under way has left a record, none is under way any more, and the synchronous calls find their text.
int wk_dtok_scan_emit_end(wk_ctx* c, int64_t* n_lines, int* status, int64_t* n_reads, int64_t* n_records) {
    if (!c || !n_lines || !status || !n_reads || !n_records) return WK_E_ARG;
    if (c->lag_count == 0) return fail(c, WK_E_STATE, "no block under way (wk_dtok_scan_emit_begin)");
    DeviceGuard guard(c->device);
    const wk_ctx::LagSlot L = c->lag[0];
    if (L.seq) {
        // (the kernel's last workgroup stores the number behind the block's scalars: looked for every few
        // microseconds -- there is a whole kernel's time of slack --, the stream waited for if it does not show)
        const volatile uint32_t* seen = reinterpret_cast<const volatile uint32_t*>(c->host_back + (size_t)L.host_slot * wk_ctx::kBackBytes + 128);
        const auto t0 = std::chrono::steady_clock::now();
        bool there = false;
        for (int spin = 0;; ++spin) {
            if (__atomic_load_n(const_cast<const uint32_t*>(reinterpret_cast<const volatile uint32_t*>(seen)), __ATOMIC_ACQUIRE) == L.seq) {
                there = true;
                break;
            }
            if (spin < 64) continue;
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.01) break;
            struct timespec ts = {0, 5000};
            nanosleep(&ts, nullptr);
        }
        if (!there) HIP_TRY(c, hipStreamSynchronize(c->stream));
    } else {
        HIP_TRY(c, hipEventSynchronize(L.ev));
    }
    DtokState st{};
    small_back_get(c, L.host_slot, &st, sizeof st);
    const bool keep = st.flags == 0 && st.n_unknown == 0;
    *n_lines = *n_reads = *n_records = 0;
    if (keep) {
        c->w_backup_cur = fz_bk(c, L.ring).p;
        c->dt_emitted = false;
        const int rc = dtok_emit_finish(c, true, false, st, 0, n_reads, n_records);
        if (rc) return rc;
        const uint32_t lines = (uint32_t)st.n_lines + (L.open_end ? 1u : 0u);
        c->dt_lines = lines;
        c->dt_lpb = (double)lines / (double)L.n;
        c->fused_streak = true;
        c->fz_back_streak = 0;
        ++c->fused_blocks;
        *n_lines = lines;
        *status = 0;
        if (L.buf >= 0) {
            std::lock_guard<std::mutex> lock(c->copy_mu);
            c->buf_state[L.buf] = wk_ctx::kBufFree;
        }
        c->lag[0] = c->lag[1];
        --c->lag_count;
        return WK_OK;
    }
    // handed back: every kernel under way through, the cursors as they were in front of this block -- which takes the
    // records of the block behind it along --, the blocks' text findable again
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->w_cursor.p, fz_bk(c, L.ring).p, kMaxStreams * 8, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    {
        std::lock_guard<std::mutex> lock(c->copy_mu);
        for (int i = 0; i < c->lag_count; ++i) {
            const wk_ctx::LagSlot& X = c->lag[i];
            if (X.buf < 0) continue;
            c->buf_state[X.buf] = wk_ctx::kBufCopied;
            c->copy_src[X.buf] = X.src;
            c->copy_n[X.buf] = X.n;
            c->copy_counted[X.buf] = false;
        }
    }
    c->lag_count = 0;
    c->fz_chain = false;
    c->fused_streak = false;
    ++c->fused_fallbacks;
    fz_handed_back(c);
    *status = 2;
    return WK_OK;
}

int wk_dtok_scan(wk_ctx* c, wk_tok* tok, const char* text, int64_t begin, int64_t stop, int extra, int64_t* n_lines, int* status) {
    return dtok_scan_impl(c, tok, text, begin, stop, extra, n_lines, status, nullptr, nullptr);
}

int wk_dtok_scan_emit(wk_ctx* c, wk_tok* tok, const char* text, int64_t begin, int64_t stop, int64_t* n_lines, int* status, int* emitted,
                      int64_t* n_reads, int64_t* n_records) {
    if (!emitted || !n_reads || !n_records) return WK_E_ARG;
    int64_t out[2] = {0, 0};
    const int rc = dtok_scan_impl(c, tok, text, begin, stop, 0, n_lines, status, emitted, out);
    *n_reads = out[0];
    *n_records = out[1];
    return rc;
}

// The emission of a scanned block in two halves, so that wk_dtok_scan_emit can queue it behind the parse without
// waiting in between: `launch` queues the kernels (and, where the records are placed by prefix sums, the copy of their
// totals), `finish` — once the stream has been waited for — accepts or discards what they wrote.
static int dtok_emit_launch(wk_ctx* c, bool* ordered_out, unsigned long long* totals) {
    c->fz_chain = false;
    if (c->dt_expect_bytes > 0 && c->w_expect == 0 && c->dt_n > 0) {
        // (a record per line at most; 8 % on top of the first block's rate, never more than the 2^30 a pass holds)
        const double lines = (double)c->dt_expect_bytes * ((double)c->dt_lines / (double)c->dt_n) * 1.08 + (double)c->dt_lines;
        c->w_expect = (int64_t)std::min(lines, (double)(1ll << 30));
    }
    int rc = words_roll(c, c->dt_lines);
    if (rc) return rc;
    if ((rc = words_room(c, c->dt_lines))) return rc;
    DtokArgs a = dtok_args(c);
    if (c->w_mode == 0) {
        a.streams = stream_set(c);
        c->w_counts_known = false;
        // (a block the kernels give up on must leave the streams as they were)
        HIP_TRY(c, c->w_backup.reserve(kMaxStreams * 8));
        a.cursor_backup = c->w_backup.as<unsigned long long>();  // (copied by dtok_runs_kernel)
        c->w_backup_cur = c->w_backup.p;
    } else {
        a.out = c->c_words.as<uint32_t>() + c->w_records;
        a.out_cap = c->dt_lines;
    }
    const dim3 grid((c->dt_lines + kDtokThreads - 1) / kDtokThreads);
    const dim3 emit_grid((c->dt_lines + kDtokThreads * kScatterItems - 1) / (kDtokThreads * kScatterItems));
    KernelTimer* kt = ktimer_begin(c, "dtok_emit");
    hipLaunchKernelGGL(dtok_runs_kernel, grid, dim3(kDtokThreads), 0, c->stream, a);
    if (a.submap && !c->dt_mapped) {  // (once per scanned block; behind the runs: a line of an excluded subject starts and continues runs like any other)
        hipLaunchKernelGGL(dtok_submap_kernel, grid, dim3(kDtokThreads), 0, c->stream, a);
        if (c->dt_has_excl) {
            HIP_TRY(c, hipMemsetAsync(c->d_first.p, 0, c->dt_lines, c->stream));
            hipLaunchKernelGGL(dtok_excl_mark_kernel, grid, dim3(kDtokThreads), 0, c->stream, a);
            hipLaunchKernelGGL(dtok_excl_drop_kernel, grid, dim3(kDtokThreads), 0, c->stream, a);
        }
        c->dt_mapped = true;
    }
    const bool ordered = c->w_mode != 0 || c->dt_keep_reads;
    *ordered_out = ordered;
    c->dt_emitted = false;
    if (ordered) hipLaunchKernelGGL(dtok_first_kernel, grid, dim3(kDtokThreads), 0, c->stream, a);
    if (ordered) {
        // the per-read stream wants the records of a read next to each other, in
        // position order: placed by prefix sums instead of appended wave by wave
        const uint32_t n_tiles = grid.x;
        HIP_TRY(c, c->d_tiles.reserve((size_t)n_tiles * 8));
        HIP_TRY(c, c->d_tile_off.reserve((size_t)n_tiles * 8));
        HIP_TRY(c, c->d_lscan.reserve(((size_t)c->dt_lines + 1) * 8));
        a.line_scan = c->d_lscan.as<unsigned long long>();
        HIP_TRY(c, hipMemsetAsync(scalar_u64(c, 3), 0, 8, c->stream));
        hipLaunchKernelGGL(dtok_hits_kernel<false>, grid, dim3(kDtokThreads), 0, c->stream, a, c->d_tiles.as<unsigned long long>());
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_tiles.as<unsigned long long>(),
                           c->d_tile_off.as<unsigned long long>(), (int64_t)n_tiles, scalar_u64(c, 3));
        hipLaunchKernelGGL(dtok_scan_lines_kernel, grid, dim3(kDtokThreads), 0, c->stream, a, c->d_tile_off.as<unsigned long long>());
        if (c->w_mode == 0)  // (read maps wanted, records for the histogram: by slice, in no particular order)
            hipLaunchKernelGGL(dtok_emit_kernel, emit_grid, dim3(kDtokThreads), 0, c->stream, a);
        else
            hipLaunchKernelGGL(dtok_place_words_kernel, grid, dim3(kDtokThreads), 0, c->stream, a);
        (void)totals;  // (slot 1 of the pinned scratch: read by the caller once the stream has been waited for)
        HIP_TRY(c, small_back(c, 1, scalar_u64(c, 3), 8));
    } else {
        // (first-line flags and emission in one kernel, the runs' lines staged in LDS)
        hipLaunchKernelGGL(dtok_first_emit_kernel, emit_grid, dim3(kDtokThreads), 0, c->stream, a);
    }
    ktimer_end(c, kt);
    HIP_TRY(c, hipGetLastError());
    return WK_OK;
}

// (the stream has been waited for) `keep` false: the streams' cursors go back to where they were
static int dtok_emit_finish(wk_ctx* c, bool keep, bool ordered, DtokState st, unsigned long long totals, int64_t* n_reads,
                            int64_t* n_records) {
    if (ordered) {
        st.n_out = totals & 0xFFFFFFFFull;
        st.n_reads = totals >> 32;
    }
    if (!keep) {
        if (c->w_mode == 0) {
            c->fz_chain = false;
            HIP_TRY(c, hipMemcpyAsync(c->w_cursor.p, c->w_backup_cur, kMaxStreams * 8, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        return WK_OK;
    }
    c->dt_emitted = c->dt_keep_reads;
    c->dt_emit_reads = (uint32_t)st.n_reads;
    const int rc = words_translate(c, c->w_records, (int64_t)st.n_out);
    if (rc) return rc;
    c->w_records += (int64_t)st.n_out;
    c->w_reads += (int64_t)st.n_reads;
    *n_reads = (int64_t)st.n_reads;
    *n_records = (int64_t)st.n_out;
    return WK_OK;
}

int wk_dtok_emit(wk_ctx* c, int64_t* n_reads, int64_t* n_records, int* status) {
    if (!c || !n_reads || !n_records || !status) return WK_E_ARG;
    *status = 1;
    *n_reads = *n_records = 0;
    if (!c->dt_ready || c->dt_extra) return fail(c, WK_E_STATE, "no block scanned for the plain flavour (wk_dtok_scan)");
    if (!c->w_open) return fail(c, WK_E_STATE, "wk_words_begin has not accepted a job set");
    c->dt_ready = false;
    if (c->dt_lines == 0) {
        *status = 0;
        return WK_OK;
    }
    DeviceGuard guard(c->device);
    KtScope kt_scope(c);
    bool ordered = false;
    unsigned long long totals = 0;
    int rc = dtok_emit_launch(c, &ordered, &totals);
    if (rc) return rc;
    DtokState st{};
    HIP_TRY(c, small_back(c, 0, c->d_state.p, sizeof st));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    small_back_get(c, 0, &st, sizeof st);
    if (ordered) small_back_get(c, 1, &totals, 8);
    // (a read of more than 16 subjects: nothing counts as appended)
    if ((rc = dtok_emit_finish(c, st.flags == 0, ordered, st, totals, n_reads, n_records))) return rc;
    if (st.flags == 0) *status = 0;
    return WK_OK;
}

// ---- strata map on the device (wk_strata.hpp) ---------------------------------------

static StrataArgs strata_args(wk_ctx* c) {
    StrataArgs s{};
    s.text = c->s_text.as<unsigned char>();
    s.n = c->s_n;
    s.line_start = c->s_lines.as<uint32_t>();
    s.n_lines = c->s_n_lines;
    s.line_tab = c->s_tab.as<uint32_t>();
    s.line_vlen = c->s_vlen.as<uint32_t>();
    s.line_hash = c->s_hash.as<unsigned long long>();
    s.line_label = c->s_label.as<uint32_t>();
    s.slots = c->s_slots.as<StrataSlot>();
    s.mask = c->s_mask;
    s.labels = c->s_labels.as<LabelSlot>();
    s.label_text = c->s_label_text.as<uint2>();
    s.label_group = c->s_label_group.as<int32_t>();
    s.state = c->s_state.as<uint32_t>();
    return s;
}

int wk_strata_clear(wk_ctx* c) {
    if (!c) return WK_E_ARG;
    c->s_ready = c->s_use = false;
    return WK_OK;
}

int wk_strata_load(wk_ctx* c, const char* text, int64_t n64, int64_t* n_pairs, int32_t* n_labels, int* status) {
    if (!c || n64 < 0 || (n64 > 0 && !text) || !n_pairs || !n_labels || !status) return WK_E_ARG;
    *status = 1;
    *n_pairs = 0;
    *n_labels = 0;
    c->s_ready = c->s_use = false;
    if (n64 >= (1ll << 32) - 64) return WK_OK;  // (offsets are 32 bits: the host's join takes larger maps)
    DeviceGuard guard(c->device);
    KtScope kt_scope(c);
    const uint32_t n = (uint32_t)n64;
    c->s_n = n;
    c->s_n_lines = 0;
    HIP_TRY(c, c->s_state.reserve(64));
    HIP_TRY(c, hipMemsetAsync(c->s_state.p, 0, 64, c->stream));
    HIP_TRY(c, c->s_labels.reserve((size_t)kStrataLabelSlots * sizeof(LabelSlot)));
    HIP_TRY(c, c->s_label_text.reserve((size_t)kStrataLabelSlots * 8));
    HIP_TRY(c, c->s_label_group.reserve((size_t)kStrataLabelSlots * 4));
    HIP_TRY(c, hipMemsetAsync(c->s_labels.p, 0xFF, (size_t)kStrataLabelSlots * sizeof(LabelSlot), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->s_label_group.p, 0xFF, (size_t)kStrataLabelSlots * 4, c->stream));
    if (n == 0) {
        HIP_TRY(c, c->s_slots.reserve(1024 * sizeof(StrataSlot)));
        HIP_TRY(c, hipMemsetAsync(c->s_slots.p, 0, 1024 * sizeof(StrataSlot), c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->s_mask = 1023;
        c->s_ready = true;
        *status = 0;
        return WK_OK;
    }
    HIP_TRY(c, c->s_text.reserve((size_t)n + 64));
    HIP_TRY(c, copy_text_async(c, c->s_text.p, text, n, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->s_text.as<unsigned char>() + n, 0, 64, c->stream));
    // line starts, like a block of alignment text
    const uint32_t n_tiles = (n + kDtokTile - 1) / kDtokTile;
    HIP_TRY(c, c->s_tiles.reserve((size_t)n_tiles * 8));
    HIP_TRY(c, c->s_tile_off.reserve((size_t)n_tiles * 8));
    HIP_TRY(c, hipMemsetAsync(scalar_u64(c, 3), 0, 8, c->stream));
    hipLaunchKernelGGL(dtok_count_kernel, dim3(n_tiles), dim3(kDtokThreads), 0, c->stream, c->s_text.as<unsigned char>(), n,
                       c->s_tiles.as<unsigned long long>());
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, c->stream, c->s_tiles.as<unsigned long long>(),
                       c->s_tile_off.as<unsigned long long>(), (int64_t)n_tiles, scalar_u64(c, 3));
    unsigned long long n_newlines = 0;
    HIP_TRY(c, hipMemcpyAsync(&n_newlines, scalar_u64(c, 3), 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const bool open_end = text[n - 1] != '\n';
    const uint32_t lines = (uint32_t)n_newlines + (open_end ? 1u : 0u);
    HIP_TRY(c, c->s_lines.reserve(((size_t)lines + 2) * 4));
    HIP_TRY(c, c->s_tab.reserve(((size_t)lines + 1) * 4));
    HIP_TRY(c, c->s_vlen.reserve(((size_t)lines + 1) * 4));
    HIP_TRY(c, c->s_hash.reserve(((size_t)lines + 1) * 8));
    HIP_TRY(c, c->s_label.reserve(((size_t)lines + 1) * 4));
    uint64_t slots = 1024;
    while (slots < 2ull * lines) slots <<= 1;
    HIP_TRY(c, c->s_slots.reserve((size_t)slots * sizeof(StrataSlot)));
    HIP_TRY(c, hipMemsetAsync(c->s_slots.p, 0, (size_t)slots * sizeof(StrataSlot), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->s_lines.p, 0, 4, c->stream));
    hipLaunchKernelGGL(dtok_lines_kernel, dim3(n_tiles), dim3(kDtokThreads), 0, c->stream, c->s_text.as<unsigned char>(), n,
                       c->s_tile_off.as<unsigned long long>(), c->s_lines.as<uint32_t>());
    if (open_end) {
        const uint32_t end = n + 1;
        HIP_TRY(c, hipMemcpyAsync(c->s_lines.as<uint32_t>() + lines, &end, 4, hipMemcpyHostToDevice, c->stream));
    }
    c->s_n_lines = lines;
    c->s_mask = (uint32_t)(slots - 1);
    const StrataArgs s = strata_args(c);
    const dim3 grid((lines + kStrataThreads - 1) / kStrataThreads);
    KernelTimer* kt = ktimer_begin(c, "strata");
    hipLaunchKernelGGL(strata_parse_kernel, grid, dim3(kStrataThreads), 0, c->stream, s);
    hipLaunchKernelGGL(strata_verify_kernel, grid, dim3(kStrataThreads), 0, c->stream, s);
    ktimer_end(c, kt);
    HIP_TRY(c, hipGetLastError());
    uint32_t st[4] = {0, 0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(st, c->s_state.p, sizeof st, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (st[0]) return WK_OK;  // status 1: the host's join takes the sample
    c->s_ready = true;
    *n_pairs = st[1];
    *n_labels = (int32_t)st[2];
    *status = 0;
    return WK_OK;
}

int wk_strata_labels(wk_ctx* c, int32_t* slot, int64_t* text_off, int32_t* text_len, int32_t cap, int32_t* n_out) {
    if (!c || !n_out || cap < 0 || (cap > 0 && (!slot || !text_off || !text_len))) return WK_E_ARG;
    *n_out = 0;
    if (!c->s_ready) return fail(c, WK_E_STATE, "no strata map on the device (wk_strata_load)");
    DeviceGuard guard(c->device);
    std::vector<LabelSlot> tab(kStrataLabelSlots);
    std::vector<uint2> txt(kStrataLabelSlots);
    HIP_TRY(c, hipMemcpyAsync(tab.data(), c->s_labels.p, tab.size() * sizeof(LabelSlot), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(txt.data(), c->s_label_text.p, txt.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // in the order their first lines appear in the map
    std::vector<std::pair<uint32_t, uint32_t>> order;  // (first line, slot)
    for (uint32_t q = 0; q < kStrataLabelSlots; ++q)
        if (tab[q].hash != kLabelEmpty) order.emplace_back(tab[q].rep, q);
    std::sort(order.begin(), order.end());
    if ((int64_t)order.size() > cap) return fail(c, WK_E_CAPACITY, "need room for %zu labels", order.size());
    for (size_t k = 0; k < order.size(); ++k) {
        const uint32_t q = order[k].second;
        slot[k] = (int32_t)q;
        text_off[k] = txt[q].x;
        text_len[k] = (int32_t)txt[q].y;
    }
    *n_out = (int32_t)order.size();
    return WK_OK;
}

int wk_strata_groups(wk_ctx* c, const int32_t* slot, const int32_t* group, int32_t n) {
    if (!c || n < 0 || (n > 0 && (!slot || !group))) return WK_E_ARG;
    if (!c->s_ready) return fail(c, WK_E_STATE, "no strata map on the device (wk_strata_load)");
    std::vector<int32_t> all(kStrataLabelSlots, -1);
    for (int32_t k = 0; k < n; ++k) {
        if (slot[k] < 0 || (uint32_t)slot[k] >= kStrataLabelSlots) return fail(c, WK_E_ARG, "label slot outside the table");
        if (group[k] >= (1 << WK_KEY_GROUP_BITS)) return fail(c, WK_E_ARG, "group id outside [0, %d)", 1 << WK_KEY_GROUP_BITS);
        all[slot[k]] = group[k];
    }
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipMemcpyAsync(c->s_label_group.p, all.data(), all.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->s_use = true;
    return WK_OK;
}

// ---- read maps on the device (wk_readmap.hpp) -------------------------------------

int wk_dtok_keep_reads(wk_ctx* c, int on) {
    if (!c) return WK_E_ARG;
    c->dt_keep_reads = on != 0;
    if (!on) c->dt_emitted = false;
    return WK_OK;
}

int wk_readmap_tables(wk_ctx* c, int32_t job, const int32_t* slot_of_subject, int32_t n_subjects, const int32_t* slot_order,
                      const uint32_t* shown_off, const char* shown, int32_t n_slots) {
    if (!c) return WK_E_ARG;
    if (job < 0 || job >= WK_MAX_JOBS) return fail(c, WK_E_ARG, "job must be in [0, %d)", WK_MAX_JOBS);
    if (n_subjects < 0 || n_slots < 0 || (n_subjects > 0 && !slot_of_subject) || (n_slots > 0 && (!slot_order || !shown_off)))
        return fail(c, WK_E_ARG, "bad read-map tables");
    for (int32_t s = 0; s < n_subjects; ++s)
        if (slot_of_subject[s] < 0 || slot_of_subject[s] >= n_slots) return fail(c, WK_E_ARG, "subject %d has no taxon slot", s);
    DeviceGuard guard(c->device);
    wk_ctx::MapTables& T = c->rm[job];
    static const uint32_t zero = 0;
    int rc;
    if ((rc = upload(c, T.slot_of_subject, slot_of_subject, (size_t)n_subjects * 4))) return rc;
    if ((rc = upload(c, T.slot_order, slot_order, (size_t)n_slots * 4))) return rc;
    if ((rc = upload(c, T.shown_off, n_slots > 0 ? shown_off : &zero, ((size_t)n_slots + 1) * 4))) return rc;
    const size_t n_text = n_slots > 0 ? shown_off[n_slots] : 0;
    if (n_text > 0 && !shown) return fail(c, WK_E_ARG, "bad read-map tables");
    if ((rc = upload(c, T.shown, shown, n_text))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // (the caller's buffers are only valid during the call)
    T.n_subjects = n_subjects;
    T.n_slots = n_slots;
    return WK_OK;
}

int wk_dtok_readmap(wk_ctx* c, int32_t job, int64_t* n_bytes) {
    if (!c || !n_bytes) return WK_E_ARG;
    *n_bytes = 0;
    c->rm_bytes = 0;
    if (job < 0 || job >= WK_MAX_JOBS) return fail(c, WK_E_ARG, "job must be in [0, %d)", WK_MAX_JOBS);
    if (!c->dt_emitted) return fail(c, WK_E_STATE, "no block emitted with its reads kept (wk_dtok_keep_reads, wk_dtok_emit)");
    const wk_ctx::MapTables& T = c->rm[job];
    if (T.n_subjects < c->n_subjects) return fail(c, WK_E_STATE, "read-map tables of job %d cover %d of %d subjects", job, T.n_subjects, c->n_subjects);
    const uint32_t n_reads = c->dt_emit_reads;
    if (n_reads == 0 || c->dt_lines == 0) return WK_OK;
    DeviceGuard guard(c->device);
    KtScope kt_scope(c);
    HIP_TRY(c, c->rm_line.reserve((size_t)n_reads * 4));
    HIP_TRY(c, c->rm_len.reserve((size_t)n_reads * 8));
    DtokArgs a = dtok_args(c);
    ReadmapArgs m{};
    m.slot_of_subject = T.slot_of_subject.as<int32_t>();
    m.n_subjects = (uint32_t)T.n_subjects;
    m.slot_order = T.slot_order.as<int32_t>();
    m.shown_off = T.shown_off.as<uint32_t>();
    m.shown = T.shown.as<unsigned char>();
    m.read_line = c->rm_line.as<uint32_t>();
    m.read_len = c->rm_len.as<unsigned long long>();
    m.n_reads = n_reads;
    const dim3 lgrid((c->dt_lines + kDtokThreads - 1) / kDtokThreads), rgrid((n_reads + kDtokThreads - 1) / kDtokThreads);
    HIP_TRY(c, c->d_tiles.reserve((size_t)rgrid.x * 8));
    HIP_TRY(c, c->d_tile_off.reserve((size_t)rgrid.x * 8));
    HIP_TRY(c, hipMemsetAsync(c->rm_len.p, 0, (size_t)n_reads * 8, c->stream));
    HIP_TRY(c, hipMemsetAsync(scalar_u64(c, 3), 0, 8, c->stream));
    KernelTimer* kt = ktimer_begin(c, "readmap");
    hipLaunchKernelGGL(readmap_len_kernel, lgrid, dim3(kDtokThreads), 0, c->stream, a, m);
    hipLaunchKernelGGL(u64_tile_sum_kernel, rgrid, dim3(kDtokThreads), 0, c->stream, m.read_len, n_reads, c->d_tiles.as<unsigned long long>());
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_tiles.as<unsigned long long>(),
                       c->d_tile_off.as<unsigned long long>(), (int64_t)rgrid.x, scalar_u64(c, 3));
    hipLaunchKernelGGL(u64_tile_prefix_kernel, rgrid, dim3(kDtokThreads), 0, c->stream, m.read_len, n_reads, c->d_tile_off.as<unsigned long long>());
    HIP_TRY(c, hipGetLastError());
    unsigned long long total = 0;
    HIP_TRY(c, hipMemcpyAsync(&total, scalar_u64(c, 3), 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (total >= (1ull << 40)) return fail(c, WK_E_RANGE, "read-map text of one block beyond 2^40 bytes");
    if (total > 0) {
        HIP_TRY(c, c->rm_text.reserve((size_t)total + 64));
        m.out = c->rm_text.as<unsigned char>();
        m.out_cap = total;
        hipLaunchKernelGGL(readmap_write_kernel, rgrid, dim3(kDtokThreads), 0, c->stream, a, m);
        HIP_TRY(c, hipGetLastError());
    }
    ktimer_end(c, kt);
    c->rm_bytes = (int64_t)total;
    *n_bytes = (int64_t)total;
    return WK_OK;
}

int wk_dtok_readmap_fetch(wk_ctx* c, char* out, int64_t cap) {
    if (!c) return WK_E_ARG;
    if (c->rm_bytes == 0) return WK_OK;
    if (!out || cap < c->rm_bytes) return fail(c, WK_E_CAPACITY, "need %lld bytes", (long long)c->rm_bytes);
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipMemcpyAsync(out, c->rm_text.p, (size_t)c->rm_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return WK_OK;
}

// The scanned block's hits ("ex" flavour) as the staged chunk of the coord-match:
// what wk_ordinal_stage would have been given.  `append`: behind the hits staged by the calls before that
// nothing has counted yet (a read's hits never span two blocks: the chunk of both is what wk_ordinal_stage
// would have been given for their reads together).
static hipError_t grow_keep(wk_ctx* c, DevBuf& b, size_t need, size_t keep) {
    if (need <= b.cap) return hipSuccess;
    if (keep == 0 || !b.p) return b.reserve(need);
    DevBuf bigger;
    hipError_t e = bigger.reserve(need * 2);
    if (e != hipSuccess && (e = bigger.reserve(need)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(bigger.p, b.p, std::min(keep, b.cap), hipMemcpyDeviceToDevice, c->stream)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return e;
    b.release();
    b = bigger;
    return hipSuccess;
}

static int dtok_stage_impl(wk_ctx* c, const int32_t* genome_of_subject, int32_t n_subjects, double th, bool append, int64_t* n_reads,
                           int64_t* n_hits, int* status) {
    if (!c || !n_reads || !n_hits || !status || n_subjects < 0 || (n_subjects > 0 && !genome_of_subject)) return WK_E_ARG;
    *status = 1;
    *n_reads = *n_hits = 0;
    if (!c->dt_ready || !c->dt_extra) return fail(c, WK_E_STATE, "no block scanned for the \"ex\" flavour (wk_dtok_scan)");
    if (!(th > 0.0)) return fail(c, WK_E_ARG, "overlap threshold must be positive");
    if (c->acc_open && !append) return fail(c, WK_E_STATE, "staged hits not counted yet (wk_ordinal_count)");
    if (c->acc_open && th != c->th) return fail(c, WK_E_STATE, "staged hits of another overlap threshold not counted yet");
    c->dt_ready = false;
    DeviceGuard guard(c->device);
    KtScope kt_scope(c);
    const uint32_t lines = c->dt_lines;
    const bool behind = append && c->acc_open && c->ord_valid;
    const int64_t base_h = behind ? c->n_hits : 0, base_r = behind ? c->o_reads : 0;
    if (base_h + (int64_t)lines >= (1ll << 31) - 64) return fail(c, WK_E_RANGE, "more than 2^31 hits staged");
    int rc;
    if ((rc = upload(c, c->d_gmap, genome_of_subject, (size_t)std::max(n_subjects, 1) * 4))) return rc;
    const size_t cap_h = (size_t)base_h + lines + 1, keep_h = (size_t)base_h * 4;
    HIP_TRY(c, grow_keep(c, c->o_genome, cap_h * 4, keep_h));
    HIP_TRY(c, grow_keep(c, c->o_beg, cap_h * 4, keep_h));
    HIP_TRY(c, grow_keep(c, c->o_end, cap_h * 4, keep_h));
    HIP_TRY(c, grow_keep(c, c->o_len, cap_h * 4, keep_h));
    HIP_TRY(c, grow_keep(c, c->o_hoff, ((size_t)base_r + lines + 2) * 4, (size_t)base_r * 4));
    unsigned long long totals = 0;
    if (lines > 0) {
        DtokArgs a = dtok_args(c);
        a.gmap = c->d_gmap.as<int32_t>();
        a.n_gmap = (uint32_t)n_subjects;
        a.o_genome = c->o_genome.as<int32_t>() + base_h;
        a.o_beg = c->o_beg.as<int32_t>() + base_h;
        a.o_end = c->o_end.as<int32_t>() + base_h;
        a.o_len = c->o_len.as<uint32_t>() + base_h;
        a.o_hoff = c->o_hoff.as<int32_t>() + base_r;
        a.o_hit_base = (uint32_t)base_h;
        const uint32_t n_tiles = (lines + kDtokThreads - 1) / kDtokThreads;
        HIP_TRY(c, c->d_tiles.reserve((size_t)n_tiles * 8));
        HIP_TRY(c, c->d_tile_off.reserve((size_t)n_tiles * 8));
        HIP_TRY(c, hipMemsetAsync(scalar_u64(c, 3), 0, 8, c->stream));
        const dim3 grid(n_tiles);
        KernelTimer* kt = ktimer_begin(c, "dtok_emit");
        hipLaunchKernelGGL(dtok_runs_kernel, grid, dim3(kDtokThreads), 0, c->stream, a);
        hipLaunchKernelGGL(dtok_hits_kernel<true>, grid, dim3(kDtokThreads), 0, c->stream, a, c->d_tiles.as<unsigned long long>());
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_tiles.as<unsigned long long>(),
                           c->d_tile_off.as<unsigned long long>(), (int64_t)n_tiles, scalar_u64(c, 3));
        hipLaunchKernelGGL(dtok_scan_lines_kernel, grid, dim3(kDtokThreads), 0, c->stream, a, c->d_tile_off.as<unsigned long long>());
        if (c->s_use) {
            HIP_TRY(c, grow_keep(c, c->c_group, ((size_t)base_r + lines + 1) * 4, (size_t)base_r * 4));
            a.o_group = c->c_group.as<int32_t>() + base_r;
            hipLaunchKernelGGL(dtok_place_kernel<true>, grid, dim3(kDtokThreads), 0, c->stream, a, strata_args(c));
        } else {
            hipLaunchKernelGGL(dtok_place_kernel<false>, grid, dim3(kDtokThreads), 0, c->stream, a, StrataArgs{});
        }
        ktimer_end(c, kt);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, small_back(c, 1, scalar_u64(c, 3), 8));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        small_back_get(c, 1, &totals, 8);
    }
    const int64_t hits = (int64_t)(totals & 0xFFFFFFFFull), reads = (int64_t)(totals >> 32);
    // (the offsets' last entry; the stream orders it in front of whatever reads them)
    hipLaunchKernelGGL(store_u32_kernel, dim3(1), dim3(1), 0, c->stream, c->o_hoff.as<uint32_t>() + base_r + reads, (uint32_t)(base_h + hits));
    HIP_TRY(c, hipGetLastError());
    c->n_hits = base_h + hits;
    c->o_reads = base_r + reads;
    c->th = th;
    c->has_group = c->s_use && (lines > 0 || (behind && c->has_group));
    c->group_base = 0;
    c->rk_valid[0] = c->rk_valid[1] = false;
    c->ord_valid = true;
    c->sb_valid = false;
    c->chunk_valid = false;
    c->acc_open = append;
    *n_reads = reads;
    *n_hits = hits;
    *status = 0;
    return WK_OK;
}

int wk_dtok_stage_hits(wk_ctx* c, const int32_t* genome_of_subject, int32_t n_subjects, double th, int64_t* n_reads,
                       int64_t* n_hits, int* status) {
    return dtok_stage_impl(c, genome_of_subject, n_subjects, th, false, n_reads, n_hits, status);
}

static bool ordinal_tally_jobs(const wk_ctx* c, const wk_job* jobs, int32_t n_jobs) {
    bool tally = c->use_tally && !c->genes_by_index && c->slots > 0;
    for (int j = 0; j < n_jobs && tally; ++j)
        tally = jobs[j].mode == WK_MODE_NONE && !(jobs[j].flags & (WK_F_UNIQ | WK_F_SIZED));
    return tally;
}

// wk_dtok_stage_hits, the block's hits placed BEHIND those of the blocks staged this way since the last count:
// a 64 MB block of text brings 1.5 M hits, the sorted match (wk_stripe.hpp) pays from a few million on.
// *may_wait = 1: the caller may stage the next block before it counts (`jobs` are of the kind the sorted match
// serves, the pile is still below what it wants); 0: count now.  Whatever is piled up is one chunk for
// wk_ordinal_count / wk_ordinal_match, with one group for all of it (wk_set_uniform_group) unless a strata map
// on the device names the reads' groups.
int wk_dtok_stage_hits_append(wk_ctx* c, const int32_t* genome_of_subject, int32_t n_subjects, double th, const wk_job* jobs,
                              int32_t n_jobs, int64_t* n_reads, int64_t* n_hits, int* status, int* may_wait) {
    if (!c || !may_wait || n_jobs < 0 || (n_jobs > 0 && !jobs)) return WK_E_ARG;
    *may_wait = 0;
    const int rc = dtok_stage_impl(c, genome_of_subject, n_subjects, th, true, n_reads, n_hits, status);
    if (rc || *status != 0) return rc;
    *may_wait = n_jobs > 0 && n_jobs <= WK_MAX_JOBS && !c->s_use && ordinal_tally_jobs(c, jobs, n_jobs) && c->use_stripes &&
                        c->stripes_usable && c->stripes_host.size() <= kStripeMax && c->n_hits < c->stripes_min_hits
                    ? 1
                    : 0;
    return WK_OK;
}

// ---- ordinal -------------------------------------------------------------------

int wk_ordinal_stage(wk_ctx* c, const int32_t* genome, const int32_t* beg, const int32_t* end, const uint32_t* len,
                     int64_t n_hits, const int32_t* hoff, int64_t n_reads, const int32_t* group, double th) {
    if (!c) return WK_E_ARG;
    if (n_hits < 0 || n_reads < 0 || !hoff || (n_hits > 0 && (!genome || !beg || !end || !len)))
        return fail(c, WK_E_ARG, "bad ordinal chunk arguments");
    if (n_hits >= (1ll << 31)) return fail(c, WK_E_RANGE, "more than 2^31 hits in one chunk");
    if (hoff[0] != 0 || hoff[n_reads] != n_hits) return fail(c, WK_E_ARG, "hoff must run from 0 to n_hits");
    if (!(th > 0.0)) return fail(c, WK_E_ARG, "overlap threshold must be positive");
    if (c->acc_open) return fail(c, WK_E_STATE, "staged hits not counted yet (wk_ordinal_count)");
    DeviceGuard guard(c->device);
    int rc;
    if ((rc = upload(c, c->o_genome, genome, (size_t)n_hits * 4))) return rc;
    if ((rc = upload(c, c->o_beg, beg, (size_t)n_hits * 4))) return rc;
    if ((rc = upload(c, c->o_end, end, (size_t)n_hits * 4))) return rc;
    if ((rc = upload(c, c->o_len, len, (size_t)n_hits * 4))) return rc;
    if ((rc = upload(c, c->o_hoff, hoff, ((size_t)n_reads + 1) * 4))) return rc;
    if (group && (rc = upload(c, c->c_group, group, (size_t)n_reads * 4))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->n_hits = n_hits;
    c->o_reads = n_reads;
    c->th = th;
    c->has_group = group != nullptr;
    c->group_base = 0;
    c->rk_valid[0] = c->rk_valid[1] = false;
    c->ord_valid = true;
    c->sb_valid = false;
    c->chunk_valid = false;
    return WK_OK;
}

// Launch of match_hits_kernel: per-genome words in LDS when they fit.
// The dense log of the gene tally (wk_ordinal.hpp): genes go to the partition their low bits name, a partition is summed in
// an LDS array of 64-bit weights indexed by the high bits.  As few partitions as that array allows: every (workgroup,
// partition) stream has a line open in L2, and 768 x 256 of them (25 MB) did not stay there — the scattered 4-byte stores
// went out as partial lines, 1.08 GB written for 0.34 GB of entries (profiles/r03_ordinal_pmc_summary.txt).
static bool range_log_shape(const wk_ctx* c, uint32_t* parts, uint32_t* span) {
    constexpr uint32_t kSpanMax = 16384;  // 128 KiB of LDS
    uint32_t p = c->range_parts_opt > 0 ? (uint32_t)c->range_parts_opt : 16u;
    while (p < kLogPartsMax && (uint32_t)(c->max_gene_feature / (int32_t)p) + 1u > kSpanMax) p <<= 1;
    *parts = p;
    *span = (uint32_t)(c->max_gene_feature / (int32_t)p) + 1u;
    return *span <= kSpanMax && c->max_gene_feature < (1 << 28);
}

static int launch_match_hits(wk_ctx* c, bool counts) {
    MatchArgs a{};
    a.genome = c->o_genome.as<int32_t>();
    a.beg = c->o_beg.as<int32_t>();
    a.end = c->o_end.as<int32_t>();
    a.len = c->o_len.as<uint32_t>();
    a.n_hits = c->n_hits;
    a.th = c->th;
    a.gene4 = c->gene4.as<int4>();
    a.grid = c->g_grid.as<int32_t>();
    a.gfirst = c->g_first.as<int32_t>();
    a.goff = c->g_goff.as<int32_t>();
    a.gshift = c->g_shift.as<unsigned char>();
    a.n_genomes = c->n_genomes;
    a.ablate = c->ablate;
    const int64_t n_tiles = (c->n_hits + kMatchTile - 1) / kMatchTile;
    const int64_t rounds = (n_tiles + kMatchGroup - 1) / kMatchGroup;
    // 9 bytes per genome; three workgroups of 512 threads per CU while the
    // registers (80 without the counts) and the LDS copies allow
    const size_t info = ((size_t)c->n_genomes * 2 + 1) * 4 + (size_t)c->n_genomes + 16;
    const bool in_lds = c->match_lds && info <= (size_t)150 * 1024;
    const int by_regs = counts ? 2 : 3;
    const int per_cu = !in_lds ? by_regs : std::max(1, std::min(by_regs, (int)((size_t)156 * 1024 / info)));
    const dim3 grid((unsigned)std::min<int64_t>(rounds, (int64_t)c->prop.multiProcessorCount * per_cu));
    const dim3 block(kMatchThreads * kMatchGroup);
    int2* f2 = c->o_first2.as<int2>();
    int32_t* start = c->o_ub.as<int32_t>();
    int32_t* cnt = c->o_cnt.as<int32_t>();
    unsigned long long* ts = c->o_tile_sum.as<unsigned long long>();
    if (in_lds && counts)
        hipLaunchKernelGGL((match_hits_kernel<true, true>), grid, block, info, c->stream, a, f2, start, cnt, ts);
    else if (in_lds)
        hipLaunchKernelGGL((match_hits_kernel<true, false>), grid, block, info, c->stream, a, f2, start, cnt, ts);
    else if (counts)
        hipLaunchKernelGGL((match_hits_kernel<false, true>), grid, block, 0, c->stream, a, f2, start, cnt, ts);
    else
        hipLaunchKernelGGL((match_hits_kernel<false, false>), grid, block, 0, c->stream, a, f2, start, cnt, ts);
    HIP_TRY(c, hipGetLastError());
    return WK_OK;
}

// Gene lists per read from the matches (offset scan, lists, read offsets);
// `matched`: match_hits_kernel already ran with counts.
static int materialise_gene_lists(wk_ctx* c) {
    const int64_t n_hits = c->n_hits;
    const int64_t n_tiles = (n_hits + kMatchTile - 1) / kMatchTile;
    HIP_TRY(c, c->o_poff.reserve((size_t)(n_hits ? n_hits : 1) * 4));
    HIP_TRY(c, c->o_tile_off.reserve((size_t)(n_tiles ? n_tiles : 1) * 8));
    HIP_TRY(c, c->o_qoff.reserve(((size_t)c->o_reads + 1) * 4));
    unsigned long long total = 0;
    HIP_TRY(c, hipMemsetAsync(scalar_u64(c, 3), 0, 8, c->stream));
    if (n_hits > 0) {
        KernelTimer* kt = ktimer_begin(c, "scan");
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, c->stream, c->o_tile_sum.as<unsigned long long>(),
                           c->o_tile_off.as<unsigned long long>(), n_tiles, scalar_u64(c, 3));
        ktimer_end(c, kt);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(&total, scalar_u64(c, 3), 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (total >= (1ull << 31)) return fail(c, WK_E_RANGE, "more than 2^31 read-gene pairs in one chunk; stage fewer hits");
        HIP_TRY(c, c->o_pairs.reserve((size_t)(total ? total : 1) * 4));
        MatchArgs a{};
        a.beg = c->o_beg.as<int32_t>();
        a.end = c->o_end.as<int32_t>();
        a.len = c->o_len.as<uint32_t>();
        a.n_hits = n_hits;
        a.th = c->th;
        a.gene4 = c->gene4.as<int4>();
        kt = ktimer_begin(c, "match_write");
        hipLaunchKernelGGL(match_write_kernel, dim3((unsigned)n_tiles), dim3(kMatchThreads), 0, c->stream, a,
                           c->o_cnt.as<int32_t>(), c->o_ub.as<int32_t>(), c->o_first2.as<int2>(), c->o_tile_off.as<unsigned long long>(), c->o_poff.as<int32_t>(),
                           c->o_pairs.as<int32_t>());
        ktimer_end(c, kt);
        HIP_TRY(c, hipGetLastError());
    } else {
        HIP_TRY(c, c->o_pairs.reserve(4));
    }
    hipLaunchKernelGGL(read_offsets_kernel, dim3((unsigned)((c->o_reads + 1 + 255) / 256)), dim3(256), 0, c->stream,
                       c->o_hoff.as<int32_t>(), c->o_poff.as<int32_t>(), c->o_reads, n_hits, scalar_u64(c, 3),
                       c->o_qoff.as<int32_t>());
    HIP_TRY(c, hipGetLastError());
    c->stat_pairs += (int64_t)total;
    // the per-read gene lists become the current classify chunk
    c->cur_subj = c->o_pairs.as<int32_t>();
    if (c->genes_by_index) {  // (the lists name genes by table index: their features for the classification)
        HIP_TRY(c, c->o_pairs_feat.reserve((size_t)(total ? total : 1) * 4));
        if (total > 0) {
            hipLaunchKernelGGL(gather_i32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                               c->o_pairs.as<int32_t>(), c->g_feature.as<int32_t>(), (int64_t)total, (int64_t)c->n_genes,
                               c->o_pairs_feat.as<int32_t>());
            HIP_TRY(c, hipGetLastError());
        }
        c->cur_subj = c->o_pairs_feat.as<int32_t>();
    }
    c->cur_qoff = c->o_qoff.as<int32_t>();
    c->n_reads = c->o_reads;
    c->n_records = (int64_t)total;
    c->subj_is_set = false;  // several hits of a read may match the same gene
    c->subj_indexed = false; // gene lists carry feature ids
    c->chunk_valid = true;
    return WK_OK;
}

static int reserve_match_buffers(wk_ctx* c) {
    const int64_t n_hits = c->n_hits;
    const int64_t n_tiles = (n_hits + kMatchTile - 1) / kMatchTile;
    HIP_TRY(c, c->o_cnt.reserve((size_t)(n_hits ? n_hits : 1) * 4));
    HIP_TRY(c, c->o_ub.reserve((size_t)(n_hits ? n_hits : 1) * 4));
    HIP_TRY(c, c->o_first2.reserve((size_t)(n_hits ? n_hits : 1) * 8));
    HIP_TRY(c, c->o_tile_sum.reserve((size_t)(n_tiles ? n_tiles : 1) * 8));
    return WK_OK;
}

int wk_ordinal_match(wk_ctx* c) {
    if (!c) return WK_E_ARG;
    if (!c->ord_valid) return fail(c, WK_E_STATE, "no ordinal chunk staged");
    if (!c->genes_set) return fail(c, WK_E_STATE, "no gene tables uploaded (wk_set_genes)");
    DeviceGuard guard(c->device);
    KtScope kt_scope(c);
    c->acc_open = false;
    int rc = reserve_match_buffers(c);
    if (rc) return rc;
    if (c->n_hits > 0) {
        KernelTimer* kt = ktimer_begin(c, "match_count");
        if ((rc = launch_match_hits(c, true))) return rc;
        ktimer_end(c, kt);
    }
    return materialise_gene_lists(c);
}

int wk_ordinal_hit_offsets(wk_ctx* c, int32_t* poff, int64_t cap) {
    if (!c || !poff) return WK_E_ARG;
    if (!c->chunk_valid || (c->cur_subj != c->o_pairs.as<int32_t>() && c->cur_subj != c->o_pairs_feat.as<int32_t>()))
        return fail(c, WK_E_STATE, "no gene lists staged (wk_ordinal_match)");
    if (cap < c->n_hits + 1) return fail(c, WK_E_CAPACITY, "need %lld offsets", (long long)c->n_hits + 1);
    DeviceGuard guard(c->device);
    if (c->n_hits) HIP_TRY(c, hipMemcpyAsync(poff, c->o_poff.p, (size_t)c->n_hits * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    poff[c->n_hits] = (int32_t)c->n_records;
    return WK_OK;
}

int wk_ordinal_pair_genes(wk_ctx* c, int32_t* out, int64_t cap) {
    if (!c || (cap > 0 && !out)) return WK_E_ARG;
    if (!c->genes_by_index) return fail(c, WK_E_STATE, "the gene lists carry features (option gene_index_pairs before wk_set_genes)");
    if (!c->chunk_valid || c->cur_subj != c->o_pairs_feat.as<int32_t>()) return fail(c, WK_E_STATE, "no gene lists staged (wk_ordinal_match)");
    if (cap < c->n_records) return fail(c, WK_E_CAPACITY, "need %lld entries", (long long)c->n_records);
    DeviceGuard guard(c->device);
    if (c->n_records) HIP_TRY(c, hipMemcpyAsync(out, c->o_pairs.p, (size_t)c->n_records * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return WK_OK;
}

int wk_set_uniform_group(wk_ctx* c, int32_t group) {
    if (!c) return WK_E_ARG;
    if (group < 0 || group >= (1 << WK_KEY_GROUP_BITS)) return fail(c, WK_E_ARG, "group id outside [0, %d)", 1 << WK_KEY_GROUP_BITS);
    c->has_group = false;
    c->group_base = group;
    return WK_OK;
}

// The genes of the staged chunk's reads counted straight from the matches (rank none, one group): the
// gather kernels of wk_ordinal.hpp over c->o_* (match_hits -> first2 -> ordinal_tally -> log -> merge).
static int tally_count(wk_ctx* c, const wk_job* jobs, int32_t n_jobs) {
    int rc = reserve_match_buffers(c);
    if (rc) return rc;
    // (without the per-hit counts: they are only needed for reads the tally
    // leaves over, and then the matching runs once more)
    KernelTimer* kt = ktimer_begin(c, "match_count");
    if ((rc = launch_match_hits(c, false))) return rc;
    ktimer_end(c, kt);
    const int blocks = std::min(kStatBlocks, c->prop.multiProcessorCount * c->tally_per_cu);
    const int64_t n_words = (c->o_reads + 63) / 64;
    TallyArgs t{};
    t.hoff = c->o_hoff.as<int32_t>();
    t.first2 = c->o_first2.as<int2>();
    t.n_reads = c->o_reads;
    t.n_jobs = n_jobs;
    for (int j = 0; j < n_jobs; ++j) t.job_index[j] = j;
    t.group = c->group_base;
    t.table = CountTable{c->tkeys.as<unsigned long long>(), c->tvals.as<unsigned long long>(), c->slots - 1, scalar_err(c)};
    // one job over gene ids with at most 4096 per partition: the dense log (wk_ordinal.hpp)
    uint32_t range_parts = 0, range_span = 0;
    const bool by_range = range_log_shape(c, &range_parts, &range_span) && c->use_range_log && n_jobs == 1;
    // distinct keys: the genes (256 merge tables of 8192 slots hold ~1.3 M at a comfortable load)
    if (by_range)
        t.log_parts = range_parts;
    else
        t.log_parts = c->log_parts_opt ? (uint32_t)c->log_parts_opt : ((int64_t)c->n_genes * n_jobs <= 256 * 5120 ? 256u : kLogPartsMax);
    const int64_t streams = (int64_t)blocks * t.log_parts;
    const size_t entry = by_range ? 4 : 8;
    // every gene of a hit is one entry: hits with a gene rarely have two
    int64_t cap = 3 * (c->n_hits * (int64_t)n_jobs / streams + 1) + 16;
    cap = std::max<int64_t>(16, std::min<int64_t>(cap, c->plog_max_bytes / (int64_t)entry / streams));
    t.plog_cap = (uint32_t)cap;
    HIP_TRY(c, c->plog.reserve((size_t)streams * t.plog_cap * entry));
    HIP_TRY(c, c->plog_cnt.reserve((size_t)streams * 4));
    HIP_TRY(c, c->left_mask.reserve((size_t)n_words * 8));
    t.plog = c->plog.as<unsigned long long>();
    t.plog32 = c->plog.as<uint32_t>();
    t.plog_cnt = c->plog_cnt.as<uint32_t>();
    t.stat_block = c->stat_block.as<unsigned long long>();
    t.left_mask = c->left_mask.as<unsigned long long>();
    t.n_left = scalar_u64(c, 7);
    HIP_TRY(c, hipMemsetAsync(scalar_u64(c, 7), 0, 16, c->stream));
    kt = ktimer_begin(c, "classify");
    if (by_range) {
        // LDS per workgroup: log cursors + 12 KiB queue + 16 KiB of gene sets
        const size_t tally_lds = (size_t)t.log_parts * 4 + (size_t)kTallyQueue * 4 + (size_t)kTallySlots * kTallyThreads * 4;
        hipLaunchKernelGGL(ordinal_tally_kernel<true>, dim3(blocks), dim3(kTallyThreads), tally_lds, c->stream, t, 0u);
        ktimer_end(c, kt);
        kt = ktimer_begin(c, "partition_merge");
        hipLaunchKernelGGL(range_merge_kernel, dim3(t.log_parts, kRangeMergeSplit), dim3(1024), (size_t)8 * range_span, c->stream, t.plog32,
                           t.plog_cnt, (uint32_t)blocks, t.plog_cap, range_span, 0u, (uint32_t)t.group, t.table);
        ktimer_end(c, kt);
    } else {
        // LDS per workgroup (three per CU): 16 KiB hash cache + log cursors + 12 KiB queue + 16 KiB of gene sets
        const uint32_t tally_slots = (uint32_t)std::max(256, std::min(c->tally_slots, 4096));
        const size_t tally_lds = (size_t)tally_slots * 16 + (size_t)t.log_parts * 4 + (size_t)kTallyQueue * 4 + (size_t)kTallySlots * kTallyThreads * 4;
        hipLaunchKernelGGL(ordinal_tally_kernel<false>, dim3(blocks), dim3(kTallyThreads), tally_lds, c->stream, t, tally_slots);
        ktimer_end(c, kt);
        kt = ktimer_begin(c, "partition_merge");
        hipLaunchKernelGGL(partition_merge_kernel, dim3(t.log_parts), dim3(1024), (size_t)8192 * 16, c->stream,
                           c->plog.as<unsigned long long>(), c->plog_cnt.as<uint32_t>(), (uint32_t)blocks, t.plog_cap, 8192u, t.table);
        ktimer_end(c, kt);
    }
    HIP_TRY(c, hipGetLastError());
    unsigned long long left_pairs[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(left_pairs, scalar_u64(c, 7), 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->chunk_valid = false;
    if (left_pairs[0] == 0) {
        c->stat_pairs += (int64_t)left_pairs[1];
        return WK_OK;
    }
    // reads the tally left out (many genes per hit / per read): their gene
    // lists through the generic evaluator
    if ((rc = launch_match_hits(c, true))) return rc;
    if ((rc = materialise_gene_lists(c))) return rc;
    c->listed_only = true;
    rc = wk_classify_staged(c, jobs, n_jobs, nullptr);
    c->listed_only = false;
    return rc;
}


// ---- the sorted coord-match (wk_stripe.hpp) ------------------------------------------------------
static StripeSortArgs stripe_sort_args(wk_ctx* c, uint32_t n_tiles) {
    StripeSortArgs a{};
    a.genome = c->o_genome.as<int32_t>();
    a.beg = c->o_beg.as<int32_t>();
    a.end = c->o_end.as<int32_t>();
    a.len = c->o_len.as<uint32_t>();
    a.hoff = c->o_hoff.as<int32_t>();
    a.n_reads = c->o_reads;
    a.stripe_of = c->g_stripe_of.as<int32_t>();
    a.n_genomes = c->n_genomes;
    a.n_stripes = (uint32_t)c->stripes_host.size();
    a.n_tiles = n_tiles;
    a.cnt = c->sb_cnt.as<uint32_t>();
    a.row_base = c->sb_base.as<unsigned long long>();
    a.binned = c->sb_binned.as<int4>();
    a.r_genome = c->r_genome.as<int32_t>();
    a.r_beg = c->r_beg.as<int32_t>();
    a.r_end = c->r_end.as<int32_t>();
    a.r_len = c->r_len.as<uint32_t>();
    a.r_hoff = c->r_hoff.as<int32_t>();
    return a;
}

// The staged chunk sorted: reads of one hit by genome stripe, the others compacted (once per staged chunk).
static int stripe_sort(wk_ctx* c) {
    if (c->sb_valid) return WK_OK;
    const uint32_t n_stripes = (uint32_t)c->stripes_host.size();
    const uint32_t n_tiles = (uint32_t)((c->o_reads + kStripeTileReads - 1) / kStripeTileReads);
    const size_t rows = (size_t)n_stripes + 2;
    HIP_TRY(c, c->sb_cnt.reserve(rows * n_tiles * 4));
    HIP_TRY(c, c->sb_tot.reserve(rows * 8));
    HIP_TRY(c, c->sb_base.reserve(rows * 8));
    // (every read could have one hit, or none: both outputs sized for the chunk)
    HIP_TRY(c, c->sb_binned.reserve((size_t)std::max<int64_t>(c->o_reads, 1) * 16));
    HIP_TRY(c, c->r_genome.reserve((size_t)std::max<int64_t>(c->n_hits, 1) * 4));
    HIP_TRY(c, c->r_beg.reserve((size_t)std::max<int64_t>(c->n_hits, 1) * 4));
    HIP_TRY(c, c->r_end.reserve((size_t)std::max<int64_t>(c->n_hits, 1) * 4));
    HIP_TRY(c, c->r_len.reserve((size_t)std::max<int64_t>(c->n_hits, 1) * 4));
    HIP_TRY(c, c->r_hoff.reserve(((size_t)c->o_reads + 1) * 4));
    const StripeSortArgs a = stripe_sort_args(c, n_tiles);
    KernelTimer* kt = ktimer_begin(c, "stripe_sort");
    hipLaunchKernelGGL(stripe_count_kernel, dim3(n_tiles), dim3(kStripeTileThreads), 0, c->stream, a);
    hipLaunchKernelGGL(stripe_rows_kernel, dim3((unsigned)rows), dim3(1024), 0, c->stream, c->sb_cnt.as<uint32_t>(), n_tiles,
                       c->sb_tot.as<unsigned long long>());
    hipLaunchKernelGGL(stripe_bases_kernel, dim3(1), dim3(64), 0, c->stream, c->sb_tot.as<unsigned long long>(),
                       c->sb_base.as<unsigned long long>(), n_stripes);
    hipLaunchKernelGGL(stripe_scatter_kernel, dim3(n_tiles), dim3(kStripeTileThreads), 0, c->stream, a);
    ktimer_end(c, kt);
    HIP_TRY(c, hipGetLastError());
    std::vector<unsigned long long> tot(rows);
    HIP_TRY(c, hipMemcpyAsync(tot.data(), c->sb_tot.p, rows * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // the pieces of the stripes' hits, one workgroup each
    std::vector<StripeUnit> units;
    unsigned long long first = 0;
    for (uint32_t s = 0; s < n_stripes; ++s) {
        for (unsigned long long at = 0; at < tot[s]; at += kStripePiece)
            units.push_back(StripeUnit{s, (uint32_t)(first + at), (uint32_t)std::min<unsigned long long>(kStripePiece, tot[s] - at), 0u});
        first += tot[s];
    }
    c->sb_single = (int64_t)first;
    c->sb_rest_reads = (int64_t)tot[n_stripes];
    c->sb_rest_hits = (int64_t)tot[n_stripes + 1];
    c->sb_n_units = (uint32_t)units.size();
    if (!units.empty()) {
        int rc = upload(c, c->sb_units, units.data(), units.size() * sizeof(StripeUnit));
        if (rc) return rc;
    }
    const int32_t end = (int32_t)c->sb_rest_hits;  // r_hoff's last entry
    HIP_TRY(c, hipMemcpyAsync(c->r_hoff.as<int32_t>() + c->sb_rest_reads, &end, 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->sb_valid = true;
    return WK_OK;
}

static int stripe_count(wk_ctx* c, const wk_job* jobs, int32_t n_jobs) {
    int rc = stripe_sort(c);
    if (rc) return rc;
    HIP_TRY(c, c->sb_stat.reserve(64));
    HIP_TRY(c, hipMemsetAsync(c->sb_stat.p, 0, 64, c->stream));
    StripeMatchArgs m{};
    m.binned = c->sb_binned.as<int4>();
    m.units = c->sb_units.as<StripeUnit>();
    m.gene4 = c->gene4.as<int4>();
    m.gene_off = c->g_gene_off.as<int32_t>();
    m.stripes = c->g_stripes.as<StripeInfo>();
    m.grid = c->g_grid.as<int32_t>();
    m.gfirst = c->g_first.as<int32_t>();
    m.goff = c->g_goff.as<int32_t>();
    m.gshift = c->g_shift.as<unsigned char>();
    m.th = c->th;
    m.n_jobs = n_jobs;
    for (int j = 0; j < n_jobs; ++j) m.job_index[j] = j;
    m.group = c->group_base;
    m.table = CountTable{c->tkeys.as<unsigned long long>(), c->tvals.as<unsigned long long>(), c->slots - 1, scalar_err(c)};
    const uint32_t over_cap = (uint32_t)std::min<int64_t>(std::max<int64_t>(c->sb_single, 1), 1 << 24);
    HIP_TRY(c, c->sb_over.reserve((size_t)over_cap * 4));
    m.overflow = c->sb_over.as<uint32_t>();
    m.overflow_cap = over_cap;
    m.stat = c->sb_stat.as<unsigned long long>();
    m.stat_block = c->stat_block.as<unsigned long long>();
    if (c->sb_n_units) {
        KernelTimer* kt = ktimer_begin(c, "stripe_match");
        hipLaunchKernelGGL(stripe_match_kernel, dim3(c->sb_n_units), dim3(kStripeMatchThreads), kStripeMatchLds, c->stream, m);
        ktimer_end(c, kt);
        HIP_TRY(c, hipGetLastError());
    }
    unsigned long long stat[3] = {0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(stat, c->sb_stat.p, 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (stat[2] > over_cap) return fail(c, WK_E_CAPACITY, "more than %u hits with three or more genes in one chunk", over_cap);
    if (stat[2]) {
        hipLaunchKernelGGL(stripe_overflow_kernel, dim3((unsigned)stat[2]), dim3(64), 0, c->stream, m, (uint32_t)stat[2]);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(stat, c->sb_stat.p, 24, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    c->stat_pairs += (int64_t)stat[1];
    c->chunk_valid = false;
    if (c->sb_rest_reads == 0) return WK_OK;
    // the reads of several hits (and hits on genomes without a stripe): the gather kernels, over their own arrays
    auto swap_rest = [&]() {
        std::swap(c->o_genome, c->r_genome);
        std::swap(c->o_beg, c->r_beg);
        std::swap(c->o_end, c->r_end);
        std::swap(c->o_len, c->r_len);
        std::swap(c->o_hoff, c->r_hoff);
        std::swap(c->n_hits, c->sb_rest_hits);
        std::swap(c->o_reads, c->sb_rest_reads);
    };
    swap_rest();
    rc = tally_count(c, jobs, n_jobs);
    swap_rest();
    return rc;
}

int wk_ordinal_count(wk_ctx* c, const wk_job* jobs, int32_t n_jobs) {
    if (!c) return WK_E_ARG;
    if (!c->ord_valid) return fail(c, WK_E_STATE, "no ordinal chunk staged");
    if (!c->genes_set) return fail(c, WK_E_STATE, "no gene tables uploaded (wk_set_genes)");
    if (n_jobs < 1 || n_jobs > WK_MAX_JOBS || !jobs) return fail(c, WK_E_ARG, "n_jobs must be in [1, %d]", WK_MAX_JOBS);
    // the genes themselves are counted (rank none, one group, no size
    // normalisation): tallied per read straight from the matches
    const bool tally = ordinal_tally_jobs(c, jobs, n_jobs) && !c->has_group && c->n_hits > 0 && c->o_reads > 0;
    c->acc_open = false;   // (whatever was piled up is counted by this call)
    if (!tally) {
        int rc = wk_ordinal_match(c);
        if (rc) return rc;
        return wk_classify_staged(c, jobs, n_jobs, nullptr);
    }
    DeviceGuard guard(c->device);
    KtScope kt_scope(c);
    // The sorted match pays for chunks that are large against its fixed costs (a scan over stripes x tiles, a
    // unit per stripe that loads the stripe's genes into LDS, one more read-back): from a few million hits on.
    // The text route stages a block's hits at a time (~1.6 M): those keep the gather kernels, at 80 us per
    // block against 370 (profiles/r05_e2e_ordinal_kernel_stats.csv).
    if (c->use_stripes && c->stripes_usable && c->stripes_host.size() <= kStripeMax && c->n_hits >= c->stripes_min_hits) {
        ++c->chunks_sorted;
        return stripe_count(c, jobs, n_jobs);
    }
    ++c->chunks_gathered;
    return tally_count(c, jobs, n_jobs);
}

int wk_chunk_download(wk_ctx* c, int32_t* subj, int64_t subj_cap, int32_t* qoff, int64_t qoff_cap, int64_t* n_records,
                      int64_t* n_reads) {
    if (!c) return WK_E_ARG;
    if (!c->chunk_valid) return fail(c, WK_E_STATE, "no chunk staged");
    if (n_records) *n_records = c->n_records;
    if (n_reads) *n_reads = c->n_reads;
    if (!subj && !qoff) return WK_OK;
    if ((subj && subj_cap < c->n_records) || (qoff && qoff_cap < c->n_reads + 1)) return fail(c, WK_E_CAPACITY, "download buffers too small");
    DeviceGuard guard(c->device);
    if (subj && c->n_records) HIP_TRY(c, hipMemcpyAsync(subj, c->cur_subj, (size_t)c->n_records * 4, hipMemcpyDeviceToHost, c->stream));
    if (qoff) HIP_TRY(c, hipMemcpyAsync(qoff, c->cur_qoff, ((size_t)c->n_reads + 1) * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return WK_OK;
}

// ---- statistics & timing -------------------------------------------------------

int wk_get_stats(wk_ctx* c, wk_stats* out) {
    if (!c || !out) return WK_E_ARG;
    DeviceGuard guard(c->device);
    std::vector<unsigned long long> part((size_t)kStatBlocks * 2);
    HIP_TRY(c, hipMemcpyAsync(part.data(), c->stat_block.p, part.size() * 8, hipMemcpyDeviceToHost, c->stream));
    unsigned long long used = 0;
    if (c->slots) {
        HIP_TRY(c, hipMemsetAsync(scalar_u64(c, 4), 0, 8, c->stream));
        hipLaunchKernelGGL(table_count_kernel, dim3(grid_for((int64_t)c->slots, 256, 2048)), dim3(256), 0, c->stream,
                           c->tkeys.as<unsigned long long>(), c->slots, scalar_u64(c, 4));
        HIP_TRY(c, hipMemcpyAsync(&used, scalar_u64(c, 4), 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    unsigned long long s[2] = {0, 0};
    for (int b = 0; b < kStatBlocks; ++b) {
        s[0] += part[2 * b];
        s[1] += part[2 * b + 1];
    }
    out->n_reads = (int64_t)s[0] + c->stat_extra_reads + (c->w_open ? c->w_reads : 0);
    out->n_records = (int64_t)s[1] + c->stat_extra_records + (c->w_open ? c->w_records : 0);
    out->n_pairs = c->stat_pairs;
    out->table_used = (int64_t)used;
    return WK_OK;
}

int wk_reset_stats(wk_ctx* c) {
    if (!c) return WK_E_ARG;
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipMemsetAsync(c->stat_block.p, 0, (size_t)kStatBlocks * 16, c->stream));
    c->stat_pairs = 0;
    c->stat_extra_reads = c->stat_extra_records = 0;
    return WK_OK;
}

int wk_timer_begin(wk_ctx* c) {
    if (!c) return WK_E_ARG;
    DeviceGuard guard(c->device);
    c->timer_closed = false;
    HIP_TRY(c, hipEventRecord(c->t0, c->stream));
    return WK_OK;
}

int wk_timer_end(wk_ctx* c) {
    if (!c) return WK_E_ARG;
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipEventRecord(c->t1, c->stream));
    c->timer_closed = true;
    return WK_OK;
}

int wk_timer_ms(wk_ctx* c, double* ms) {
    if (!c || !ms) return WK_E_ARG;
    if (!c->timer_closed) return fail(c, WK_E_STATE, "timer region not closed");
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipEventSynchronize(c->t1));
    float f = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&f, c->t0, c->t1));
    *ms = (double)f;
    return WK_OK;
}

int wk_profile_kernels(wk_ctx* c, int enable) {
    if (!c) return WK_E_ARG;
    c->profile = enable != 0;
    return WK_OK;
}

int wk_last_kernel_ms(wk_ctx* c, const char* family, double* ms) {
    if (!c || !family || !ms) return WK_E_ARG;
    auto it = c->ktimers.find(family);
    if (it == c->ktimers.end() || !it->second.valid) return fail(c, WK_E_STATE, "no timed launch of kernel family '%s'", family);
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipEventSynchronize(it->second.b));
    float f = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&f, it->second.from, it->second.b));
    *ms = (double)f;
    return WK_OK;
}

}  // extern "C"
