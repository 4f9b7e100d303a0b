/* woltka_hip_measure.h -- measurement entry points of libwoltka_hip.so.
 *
 * NOT part of the drop-in surface (include/woltka_hip.h): nothing a host layer
 * needs to classify alignments is declared here.  bench.py, tools/ and the
 * tests that hold one route against another use these: launch-shape and
 * ablation knobs (results never depend on a knob), HIP-event timers on the
 * context's own stream, and device-resident text for timing the text route
 * without the host link.
 */
#ifndef WOLTKA_HIP_MEASURE_H
#define WOLTKA_HIP_MEASURE_H
#include "woltka_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- measurement only ---------------------------------------------------
 * Launch shapes, ablation switches and what a benchmark needs to time repeated
 * passes over one resident batch.  Nothing in the host layer (woltka_amd/) depends on it (the
 * route's lap printer under WOLTKA_DTOK_TIMING aside); bench.py (--opt NAME=VALUE), tools/
 * and the tests that hold one route against another do.  Results never depend
 * on a knob.  The knobs: "lds_slots" (LDS front-cache slots per workgroup, power of two
 * in [64, 8192]), "use_lds" (0/1), "tiled" (0/1: LDS-staged classify kernel),
 * "dense" (0/1: dense LDS bins for small id spaces), "plog" (0 off / 1 auto / 2
 * always: partitioned miss log), "plog_max_bytes", "log_parts" (0 auto, or a
 * power of two in [64, 1024]: hash partitions of that log), "threads" (workgroup size of
 * the direct classify kernel), "blocks_per_cu"
 * (classify grid size per CU, 1..32), "split" (0 off / 1 on: single-candidate
 * reads in a first small kernel, the rest compacted into per-workgroup lists
 * for the generic kernel), "subject_bins" (0/1: with a small subject table the
 * first pass histograms subject indices and the assigners run once per
 * subject), "hot_bins" (0/1: for larger tables the first 24,576 subject indices
 * are histogrammed, the others evaluated per read), "count_kernel" (0/1: the
 * subject histogram as its own statically pipelined kernel),
 * "single_blocks_per_cu" (grid of the first pass), "weigh" (0 off / 1 auto / 2
 * whenever the jobs allow it: plain rank jobs as one weighted histogram over
 * subject indices, csrc/wk_weigh.hpp), "bins_ring" (3/4/6/8 tiles in flight in
 * that histogram), "tally" (0/1: wk_ordinal_count counts rank-none jobs
 * straight from the matches), "tally_slots" / "tally_per_cu" (LDS cache slots
 * and workgroups per CU of that kernel), "grid_density" (1..8 grid cells per
 * gene; takes effect at the next wk_set_genes), "match_lds" (0/1: per-genome
 * words of the coordinate grid in LDS), "range_log" (0: the coord-match tally keeps the hashed miss log instead of 4-byte entries by gene stripe,
 * csrc/wk_ordinal.hpp), "free_per_cu" / "free_threads" / "free_slots" (launch shape of the free-rank
 * stream, csrc/wk_free.hpp: workgroups per CU, threads and LDS cache slots per workgroup), "words_keep" (0/1, measurement: wk_words_flush classifies the accumulated
 * packed records but leaves them in place, so that a benchmark can time
 * repeated passes over one resident batch), "free_sparse" (0/1: `--rank free` on
 * chunks of subject indices looks the LCA up in a sparse table over the
 * subjects instead of walking up the tree).
 * "streams" (0: the packed records of all slices of the subject table in one
 * stream, round 3's team kernel), "range_parts" (partitions of the dense gene
 * log; 0 = as few as the merge's LDS array allows), "stripes" (0: the
 * coord-match tally never sorts the hits by genome stripe, csrc/wk_stripe.hpp)
 * and "stripes_min" (chunks of fewer hits keep the gather kernels; default
 * 4,000,000). */
int wk_tune(wk_ctx* ctx, const char* name, int64_t value);

/* ---- measurement ------------------------------------------------------- */
/* HIP-event timing on the context's own stream (the stream every kernel of
 * this library is launched on).  wk_timer_begin/end bracket a region;
 * wk_timer_ms returns the elapsed GPU time of the last closed region.
 * wk_last_kernel_ms returns the duration of the most recent launch of the
 * named kernel family ("classify", "leftover", "weigh_merge", "read_sizes", "dense_merge", "partition_merge",
 * "match_count", "match_write", "scan", "rank_table", "compact"), measured with events around that launch; event
 * recording around individual kernels is enabled by wk_profile_kernels(1). */
int wk_timer_begin(wk_ctx* ctx);
int wk_timer_end(wk_ctx* ctx);
int wk_timer_ms(wk_ctx* ctx, double* ms);
int wk_profile_kernels(wk_ctx* ctx, int enable);
int wk_last_kernel_ms(wk_ctx* ctx, const char* family, double* ms);

/* Resident text: text[begin, stop) -- a block as the host cuts it for
 * wk_dtok_scan -- is copied to the device and stays there; a later scan of the
 * same bytes copies nothing.  bench.py times the device side of the text route
 * this way (every kernel a block goes through, inputs in HBM when the clock
 * starts).  wk_text_clear frees the blocks. */
int wk_text_upload(wk_ctx* ctx, const char* text, int64_t begin, int64_t stop);
int wk_text_clear(wk_ctx* ctx);
/* Blocks of plain SAM text the one-kernel tokenizer (csrc/wk_dtok_fused.hpp) did
 * and blocks it handed back to the six kernels of csrc/wk_dtok.hpp, since the
 * context was created.  wk_tune "dtok_fused" (0/1) switches it; results never
 * depend on it ("dtok_fused_per_cu": its persistent workgroups per CU). */
int wk_dtok_fused_counts(wk_ctx* ctx, int64_t* done, int64_t* handed_back);
/* Chunks wk_ordinal_count matched sorted by genome stripe (csrc/wk_stripe.hpp)
 * and chunks it matched with the gather kernels alone (csrc/wk_ordinal.hpp),
 * since the context was created.  wk_tune "stripes_min" / WOLTKA_STRIPES_MIN:
 * the hits from which a chunk is sorted; results never depend on it. */
int wk_ordinal_chunk_counts(wk_ctx* ctx, int64_t* sorted, int64_t* gathered);
/* The rate (bytes/s) of `reps` pinned host -> device copies of `bytes` each,
 * back to back on a stream of their own: the bound of the end-to-end text
 * route on this box (bench.py's e2e rooflines are quoted against it). */
int wk_h2d_rate(wk_ctx* ctx, int64_t bytes, int reps, double* bytes_per_s);

#ifdef __cplusplus
}
#endif
#endif /* WOLTKA_HIP_MEASURE_H */
