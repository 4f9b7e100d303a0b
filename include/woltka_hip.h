/* woltka_hip.h — C ABI of libwoltka_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the `woltka classify` hot path.  The
 * reference (qiyunzhu/woltka v0.1.7, pure Python) has no FFI of its own; the
 * boundary below is what a ctypes binding added to the reference would call in
 * place of the per-read Python loops.  Each entry point cites the reference
 * code it replaces (paths relative to the reference repository root).
 *
 * Conventions
 * -----------
 *  - Plain C: opaque context, plain pointers + sizes, no torch/C++ types.
 *  - Every function returns 0 on success, <0 on error (WK_E_*); the message is
 *    available from wk_last_error().  The library never falls back to a CPU
 *    path: without a usable HIP device wk_create() fails.
 *  - The caller owns all host buffers; they need to stay valid only for the
 *    duration of the call.  The library owns all device memory.
 *  - Strings never cross the boundary.  The host interns every subject /
 *    taxon / gene name into an int32 *feature id*.  Ids [0, n_nodes) are the
 *    nodes of the classification hierarchy numbered in DFS pre-order (root is
 *    0, parent id < child id); ids >= n_nodes are names that are not part of
 *    the hierarchy (woltka/tree.py: "taxon not in tree").
 *  - A *read* is one (query, mate) unit yielded by the reference's alignment
 *    parsers (woltka/align.py:328-333); its candidates are a CSR segment.
 *
 * Count keys
 * ----------
 * Counts are exact integers.  A read that contributes 1/k to each of k
 * candidates (woltka/classify.py:167-170) increments the counter of the key
 * (job, k, group, feature) by one; the host reconstitutes sum_k n_k / k as an
 * exact rational before rounding (woltka/util.py:323-354).  Key layout
 * (uint64): [63:61] job | [60:49] k (1..4095) | [48:28] group | [27:0] feature.
 */
#ifndef WOLTKA_HIP_H
#define WOLTKA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WK_ABI_VERSION 1

/* error codes */
#define WK_OK 0
#define WK_E_HIP (-1)      /* HIP runtime error (no device, OOM, launch failure) */
#define WK_E_ARG (-2)      /* invalid argument */
#define WK_E_STATE (-3)    /* required table not uploaded */
#define WK_E_CAPACITY (-4) /* caller-provided output buffer too small */
#define WK_E_RANGE (-5)    /* value does not fit the key layout (k > 4095, ...) */
#define WK_E_TABLE_FULL (-6) /* device count table ran out of slots; counts are incomplete */

/* key layout */
#define WK_KEY_FEATURE_BITS 28
#define WK_KEY_GROUP_BITS 21
#define WK_KEY_K_BITS 12
#define WK_KEY_JOB_BITS 3
#define WK_MAX_JOBS 8
#define WK_MAX_K 4095
/* Contributions 1/k with k <= WK_WEIGHT_MAX_K are accumulated under k = 0 as
 * multiples of 1 / WK_WEIGHT_L (L = lcm(1..16), exact in 64 bits for > 2^43
 * reads): one key per (job, group, feature) instead of one per k, so the
 * on-chip caches aggregate far more.  Keys with k >= 1 hold plain counts of
 * 1/k contributions (k = 1 from the dense / per-subject paths, k > 16 always). */
#define WK_WEIGHT_L 720720
#define WK_WEIGHT_MAX_K 16
#define WK_FEATURE_UNASSIGNED 0x0FFFFFFF /* 'Unassigned', workflow.py:1038-1039 */
#define WK_MAX_FEATURE 0x0FFFFFFE

/* assignment modes: which reference assigner a job reproduces */
#define WK_MODE_NONE 0 /* classify.assign_none  (classify.py:32-51)  */
#define WK_MODE_FREE 1 /* classify.assign_free  (classify.py:54-78)  */
#define WK_MODE_RANK 2 /* classify.assign_rank  (classify.py:81-127) */

/* job flags */
#define WK_F_UNIQ 1u       /* --uniq       */
#define WK_F_ABOVE 2u      /* --above      (classify.py:119-123) */
#define WK_F_SUBOK 4u      /* --subok      (classify.py:75)      */
#define WK_F_UNASSIGNED 8u /* --unassigned (workflow.py:1038-1039) */
#define WK_F_SIZED 16u     /* --sizes: contributions go to the (feature, subject)
                              log instead of the count table (classify.py:174-213) */

/* subj_flags of wk_chunk_stage / wk_classify_chunk */
#define WK_SUBJ_IS_SET 1
#define WK_SUBJ_INDEXED 2
#define WK_GROUP_UNIFORM 4 /* `group` points to ONE id that holds for every read */

/* values written to the optional per-read assignment output */
#define WK_ASSIGN_NONE (-1)  /* read not assigned (None)                     */
#define WK_ASSIGN_MULTI (-2) /* read split over several features (a list)    */
#define WK_ASSIGN_EMPTY (-3) /* read has no candidates (skipped altogether)  */

typedef struct wk_ctx wk_ctx;
typedef struct wk_tok wk_tok; /* the native tokenizer (below) */

/* One classification job = one rank of `--rank a,b,c` evaluated in the same
 * pass over the records (the reference loops `for rank in ranks`,
 * workflow.py:333-335). */
typedef struct wk_job {
    int32_t mode;       /* WK_MODE_*                                          */
    int32_t rank_slot;  /* WK_MODE_RANK: slot filled by wk_build_rank_table   */
    uint32_t flags;     /* WK_F_*                                             */
    uint32_t _pad;
    double major;       /* majority threshold as fraction (major/100), 0=off;
                           compared in fp64 exactly as classify.py:317        */
} wk_job;

typedef struct wk_stats {
    int64_t n_reads;      /* reads with >= 1 candidate ("Number of sequences
                             classified", workflow.py:305,344)                */
    int64_t n_records;    /* candidate records consumed                       */
    int64_t n_pairs;      /* ordinal: read-gene matches emitted               */
    int64_t table_used;   /* distinct count keys currently in the table       */
} wk_stats;

/* ---- life cycle -------------------------------------------------------- */
int wk_abi_version(void);
/* Digest of the sources this library was compiled from (csrc/ + this header;
 * computed by the build recipe, __graft_entry__.build_native, which rebuilds
 * on a mismatch): proof that the .so in use is not a stale one. */
const char* wk_build_id(void);
/* Number of HIP devices visible to this process (0 when there is none). */
int wk_device_count(void);
/* PCI address of a device ("0000:c1:00.0") — with it a host layer finds the
 * NUMA node the GPU hangs off (/sys/bus/pci/devices/<address>/numa_node) and
 * keeps the process that feeds it on that node. */
int wk_device_pci_bus_id(int device, char* buf, size_t cap);
/* Create a context on HIP device `device`.  Fails (WK_E_HIP) without a GPU. */
int wk_create(int device, wk_ctx** out);
void wk_destroy(wk_ctx* ctx);
const char* wk_last_error(const wk_ctx* ctx); /* ctx may be NULL */
int wk_device_name(const wk_ctx* ctx, char* buf, size_t cap);
int wk_sync(wk_ctx* ctx); /* wait for all work on the context's stream */
/* Product options.  "gene_index_pairs" (0/1; takes effect at the next
 * wk_set_genes): the gene lists of wk_ordinal_match are kept by gene table index
 * as well (wk_ordinal_pair_genes).  "dtok_count_ahead" (0/1, default 1): 0 =
 * the blocks copied from now on will be scanned for plain SAM records of the
 * weighted histogram -- wk_dtok_copy* launches no newline count behind their
 * copies (the one-kernel tokenizer needs none; a block that takes the six
 * kernels after all is counted when it is scanned).  Results never depend on
 * it.  Unknown names are an error. */
int wk_set_option(wk_ctx* ctx, const char* name, int64_t value);

/* (Measurement entry points -- launch-shape knobs, event timers, resident-text
 * passes for the benchmark -- are declared in woltka_hip_measure.h: they are
 * not part of the drop-in surface.) */

/* ---- static state ------------------------------------------------------ */
/* Flattened hierarchy (replaces the `tree`/`rankdic` dicts produced by
 * workflow.build_hierarchy, workflow.py:698-815, after tree.fill_root,
 * tree.py:302-388).  Nodes are in DFS pre-order: parent[0] == 0 is the root,
 * parent[v] < v otherwise; last[v] is the largest id in v's subtree;
 * rank_code[v] is the host's integer code of rankdic[v] (0 = no rank). */
int wk_set_tree(wk_ctx* ctx, const int32_t* parent, const int32_t* last,
                const int32_t* rank_code, int32_t n_nodes);

/* tree.find_rank (tree.py:467-510) for every node at once: fills slot `slot`
 * with anc[v] = first node on the path v -> root (v itself first) whose
 * rank_code equals `rank_code`, or -1.  Device kernel. */
int wk_build_rank_table(wk_ctx* ctx, int32_t slot, int32_t rank_code);
/* Download a rank table (testing / read-map output). `out` has n_nodes slots. */
int wk_get_rank_table(wk_ctx* ctx, int32_t slot, int32_t* out);

/* Gene coordinate tables (replaces the `coords` dict of encoded int64 queues
 * built by ordinal.load_gene_coords / encode_genes, ordinal.py:338-473).
 * Genome g owns genes [genome_off[g], genome_off[g+1]), sorted by start0.
 * start0 = min(beg,end)-1, end = max(beg,end) (ordinal.py:459-465).
 * gene_feature[i] is the feature id reported for gene i. */
int wk_set_genes(wk_ctx* ctx, const int32_t* genome_off, int32_t n_genomes,
                 const int32_t* start0, const int32_t* end,
                 const int32_t* gene_feature, int32_t n_genes);

/* Optional compact subject table.  Alignment files name a limited set of
 * subjects (genomes) over and over; the host interns them into dense *subject
 * indices* (order of first appearance) and registers feature_of_subject[s] =
 * the feature id of subject s (a hierarchy node id, or an id >= n_nodes).  A
 * chunk staged with WK_SUBJ_INDEXED then carries subject indices in `subj`,
 * and the library keeps one short row per subject (feature id + its ancestor
 * at every requested rank), so that a record costs one gather from a table that
 * stays cache-resident instead of one gather per rank into per-node tables.
 * May be called again with a longer table as new subjects appear. */
int wk_set_subjects(wk_ctx* ctx, const int32_t* feature_of_subject,
                    int32_t n_subjects);

/* ---- count table ------------------------------------------------------- */
/* (Re)allocate the device count table with at least `min_slots` slots and
 * clear it.  Must be called before the first classify call. */
int wk_counts_reserve(wk_ctx* ctx, int64_t min_slots);
int wk_counts_clear(wk_ctx* ctx);
/* Copy all (key, count) pairs to the host. Returns WK_E_CAPACITY and sets *n
 * to the required size if cap is too small. Order is unspecified. */
int wk_counts_fetch(wk_ctx* ctx, uint64_t* keys, int64_t* counts, int64_t cap,
                    int64_t* n);

/* ---- contribution log of size-normalised jobs --------------------------- */
/* A WK_F_SIZED job does not count: every contribution is appended to a log as
 * 4 x int32 {feature, subject feature id, job << 16 | divisor, group}; its
 * value is sizes[subject] / divisor (classify.counter_size, classify.py:
 * 174-213), which the host evaluates.  wk_log_fetch copies the entries logged
 * so far and empties the log; WK_E_CAPACITY (with *n = entries needed) means
 * the log overflowed — reserve more and re-run the chunk. */
int wk_log_reserve(wk_ctx* ctx, int64_t n_entries);
int wk_log_fetch(wk_ctx* ctx, int32_t* out, int64_t cap, int64_t* n);

/* ---- per-chunk work ---------------------------------------------------- */
/* Stage one packed chunk of plain-mapper output in HBM (replaces the
 * (qryque, subque) lists yielded by align.plain_mapper, align.py:47-115).
 *   subj[n_records]   candidate feature id per alignment record
 *   qoff[n_reads + 1] CSR offsets of each read's records (int32: one staged
 *                     chunk holds < 2^31 records; larger inputs are chunked)
 *   group[n_reads]    optional stratum/sample slot per read, -1 = read is not
 *                     in the strata map and is skipped (classify.py:239);
 *                     NULL = group 0 for every read; with WK_GROUP_UNIFORM
 *                     group[0] is the group of every read (one sample per
 *                     input file, workflow.py:304-335: the usual case, and
 *                     the one the per-subject counting paths need)
 * `subj_flags`: WK_SUBJ_IS_SET promises that no read lists the same subject
 * twice (the reference's per-read sets, align.py:309); otherwise the device
 * removes duplicates itself.  WK_SUBJ_INDEXED: `subj` holds subject indices of
 * the table given to wk_set_subjects instead of feature ids. */
int wk_chunk_stage(wk_ctx* ctx, const int32_t* subj, const int32_t* qoff,
                   int64_t n_reads, const int32_t* group, int subj_flags);

/* Run `n_jobs` classification jobs over the staged chunk and add the results
 * to the count table (replaces workflow.assign_readmap, workflow.py:941-1058:
 * assigner -> counter -> sum_dict).  May be called repeatedly on one staged
 * chunk.  `out_assign`, if not NULL, receives n_jobs * n_reads int32
 * (job-major): feature id, or WK_ASSIGN_*. */
int wk_classify_staged(wk_ctx* ctx, const wk_job* jobs, int32_t n_jobs,
                       int32_t* out_assign);

/* ---- packed records, accumulated over the chunks of one sample ------------
 * The native tokenizer hands the plain flavour's records over as one word each
 * (wk_tok_fetch_packed): subject index (23 bits) | position of the record in
 * its read << 23 (4 bits) | size of its read << 27 (1..16).  For the plain
 * assigners — classify.assign_none / assign_rank without --uniq, --major,
 * --above, followed by classify.counter (classify.py:32-51, 81-127, 144-171)
 * — that word is all the device needs (wk_weigh.hpp): the chunks of a sample
 * (workflow.py:304-335 loops over them) are appended to one device buffer as
 * they are tokenised, asynchronously from pinned memory, and classified by one
 * launch when the sample ends (wk_words_flush; wk_counts_fetch flushes too).
 * Two more job sets are taken, by the per-read stream of csrc/wk_free.hpp: one
 * `--rank free` job (classify.assign_free, classify.py:54-78), and one rank job
 * under --uniq, --above or --major above one half (classify.assign_rank,
 * classify.py:81-141 with classify.majority, classify.py:300-317).
 *
 * wk_words_begin declares the jobs and the group (sample) of the records that
 * follow; *ok = 0 means this job set / subject table needs the general path
 * (wk_chunk_stage + wk_classify_staged) — e.g. a subject without an ancestor
 * at a requested rank (its reads change k, classify.py:167-168).  Records
 * accumulated under other jobs or another group are classified first.
 * wk_words_append copies one chunk; with slot >= 0 the copy is only enqueued
 * and `words` (pinned: wk_host_alloc) must stay untouched until
 * wk_words_wait(slot) returns; slot = -1 copies before returning. */
int wk_host_alloc(wk_ctx* ctx, size_t bytes, void** out);
int wk_host_free(wk_ctx* ctx, void* p);
/* Memory of the caller — a read-only mapping of an alignment file — pinned in
 * place (hipHostRegister, read-only), so that wk_dtok_copy takes the text from
 * the page cache without a copy on the host; `p` page-aligned.  A failure
 * (WK_E_HIP) leaves nothing registered: the caller reads the file into pinned
 * buffers instead.  Ranges still registered are released by wk_destroy. */
int wk_host_register(wk_ctx* ctx, const void* p, size_t bytes);
int wk_host_unregister(wk_ctx* ctx, const void* p);
int wk_words_begin(wk_ctx* ctx, const wk_job* jobs, int32_t n_jobs,
                   int32_t group, int* ok);
int wk_words_append(wk_ctx* ctx, const uint32_t* words, int64_t n_records,
                    int64_t n_reads, int slot);
int wk_words_wait(wk_ctx* ctx, int slot);
int wk_words_flush(wk_ctx* ctx);
int wk_words_pending(wk_ctx* ctx, int64_t* n_records, int64_t* n_reads);

/* ---- SAM tokenizer on the device (plain flavour) --------------------------
 * align.parse_sam_file + plain_mapper (align.py:258-347, 47-115) on the GPU
 * (csrc/wk_dtok.hpp): the host only moves the text (pread into pinned memory,
 * one copy to HBM); lines are split, runs of equal QNAME grouped into reads
 * by mate, subjects looked up in `tok`'s dictionary, and the reads' records
 * appended as packed words to the sample's accumulated records (wk_words_*).
 * text[begin, stop) must be whole lines ending at a run boundary
 * (wk_tok_sam_span).
 *
 * wk_dtok_scan copies and parses the block; subjects the dictionary does not
 * know are interned into `tok` in text order (wk_tok_new_subjects reports
 * them: the ids are those the host tokenizer would have assigned).  *status:
 * 0 = parsed, 1 = the block has something the kernels leave to the host
 * tokenizer (a short or malformed line, both mate bits, an exclusion set):
 * tokenise it with wk_tok_text instead.  wk_dtok_emit — after wk_words_begin
 * accepted the jobs for the grown subject table — groups and appends the
 * records; *status 1 = a read of more than WK_WEIGHT_MAX_K subjects: nothing
 * was appended, the host tokenizer takes the block. */
/* Start the copy of a block's text on a copy stream (pinned `text`): the
 * wk_dtok_scan of the same block then only waits for it, and the copy of block
 * i + 1 overlaps the kernels of block i. */
int wk_dtok_copy(wk_ctx* ctx, const char* text, int64_t begin, int64_t stop);
/* The same copy for a reader that runs ahead of the scans and does not keep
 * the host bytes until then (up to 191 blocks: the reader may start while the
 * hierarchy is still being read — workflow.py:84-95 reads it before the first
 * alignment — and HBM holds what it copied meanwhile).  *ticket names the copy:
 * after wk_dtok_copy_wait(ticket) text[begin, stop) may be overwritten.  The
 * block is scanned with the same (text, begin, stop), which is only its tag
 * then; blocks are scanned in the order they were copied.  wk_dtok_text_back
 * copies the text of the block scanned last back to the host (n = stop -
 * begin): what the host tokenizer is given when *status = 1.
 * wk_dtok_copy_drop forgets the blocks copied ahead that no scan asked for. */
int wk_dtok_copy_ahead(wk_ctx* ctx, const char* text, int64_t begin,
                       int64_t stop, int32_t* ticket);
int wk_dtok_copy_wait(wk_ctx* ctx, int32_t ticket);
int wk_dtok_copy_drop(wk_ctx* ctx);
/* `--trim-sub` (workflow.py:840-841: `x.rsplit(sep, 1)[0]`, then a set again):
 * the names the tokenizer meets are not the subjects; map[id of a name] = index
 * of its subject (wk_set_subjects).  The plain flavour's kernels translate a
 * block's lines before they group them into reads.  map NULL: no map (n = 0
 * with a map: one that no name has entered yet).
 * `--exclude` (align.py:47-115, 438-470: a query that hits a subject of the set
 * is dropped whole, all its mates): map[id] = -4 for the names of the set; a
 * tokenizer with an exclusion set is scanned on the device only under such a
 * map (plain flavour; the "ex" parsers' exclusion stays on the host). */
int wk_dtok_subject_map(wk_ctx* ctx, const int32_t* map, int32_t n);
/* How many blocks may be copied ahead on this device as it is now: half of its
 * free memory in text buffers (a reader that starts before the hierarchy is
 * read must not take the memory the count table and the records will want). */
int wk_dtok_ahead_room(wk_ctx* ctx, int32_t* n_blocks);
int wk_dtok_text_back(wk_ctx* ctx, char* out, int64_t n);
/* A hint: the blocks scanned from now on are `text_bytes` bytes of one sample
 * in all (0 = unknown again).  The sample's record buffers are then sized once,
 * from the first block's lines per byte, instead of grown as they fill. */
int wk_dtok_expect(wk_ctx* ctx, int64_t text_bytes);
int wk_dtok_scan(wk_ctx* ctx, wk_tok* tok, const char* text, int64_t begin,
                 int64_t stop, int extra, int64_t* n_lines, int* status);
/* The format of the blocks wk_dtok_scan is given from now on: WK_FMT_SAM
 * (default), WK_FMT_B6O (align.parse_b6o_file / _ex, align.py:753-856),
 * WK_FMT_PAF (align.parse_paf_file / _ex, align.py:984-1095) or -- plain
 * flavour only, it has no other -- WK_FMT_MAP (align.parse_map_file,
 * align.py:621-674): rows `query <tab> subject ...` (PAF: the subject is the
 * 6th field), grouped into runs of equal queries the same way; lines that are
 * not rows of the format are ignored.  With `extra` a BLAST / PAF row also
 * gives start, end and aligned length; number text beyond sign-and-digits
 * sends the block to the host tokenizer (status 1), as POS / CIGAR do. */
int wk_dtok_format(wk_ctx* ctx, int fmt);
int wk_dtok_emit(wk_ctx* ctx, int64_t* n_reads, int64_t* n_records,
                 int* status);
/* wk_dtok_scan and wk_dtok_emit of the plain flavour with one wait instead of
 * two: while the words of this sample are open (wk_words_begin accepted the
 * jobs) the emission is queued right behind the parse — the usual block
 * brings no subject the dictionary does not know — and discarded again if the
 * parse says otherwise.  *emitted = 1: the block's records are appended
 * (*n_reads, *n_records as wk_dtok_emit reports them); 0: the call did what
 * wk_dtok_scan does (*status as there) and wk_dtok_emit is still to come. */
int wk_dtok_scan_emit(wk_ctx* ctx, wk_tok* tok, const char* text, int64_t begin,
                      int64_t stop, int64_t* n_lines, int* status, int* emitted,
                      int64_t* n_reads, int64_t* n_records);
/* The same with the verdict read one block late, so that the device goes from
 * one block's kernel to the next without waiting for the host (the reference
 * has no counterpart: align.py:86-128 yields a chunk at a time).  _begin
 * launches the block's one-kernel tokenizer and returns; *started = 0: the
 * block is not one for this way right now (not copied ahead, not the plain
 * SAM flavour, two blocks under way already, the sample's buffers to be grown
 * or rolled first ...) and nothing has happened -- wk_dtok_scan_emit takes it
 * once the blocks under way have been read.  _end waits for the OLDEST block
 * under way.  *status 0: its records are appended (*n_lines, *n_reads,
 * *n_records as wk_dtok_scan_emit reports them).  *status 2: the kernel handed
 * the block back; neither it nor the block launched behind it has left a
 * record, none is under way any more, and both are found by wk_dtok_scan_emit
 * as blocks copied ahead.  wk_words_flush, wk_words_append and the scans
 * refuse (WK_E_STATE) while a block is under way. */
int wk_dtok_scan_emit_begin(wk_ctx* ctx, wk_tok* tok, const char* text,
                            int64_t begin, int64_t stop, int* started);
int wk_dtok_scan_emit_end(wk_ctx* ctx, int64_t* n_lines, int* status,
                          int64_t* n_reads, int64_t* n_records);
/* ---- strata map on the device (csrc/wk_strata.hpp) -------------------------
 * workflow.read_strata + the lookups of classify.counter_strat (workflow.py:
 * 912-938, file.py:368-385, classify.py:216-249) for samples the device
 * tokenises: the text of the sample's read -> stratum map (lines `read <tab>
 * label`; lines with another number of columns are ignored, the label is
 * right-stripped, a repeated read keeps its last label) goes to the device as
 * it is and is joined there, exactly (hashes find candidates, bytes decide).
 *   wk_strata_load    builds the tables from text[0, n).  *status 1: something
 *                     the kernels leave to the host's join (wk_tok_strata_*): a
 *                     map of 4 GB or more, two ids or labels with equal 64-bit
 *                     hashes, more than 65536 labels.
 *   wk_strata_labels  the labels in order of first appearance: their slot,
 *                     and where their text lies in the map's text.
 *   wk_strata_groups  the (sample, stratum) group id of every label; from then
 *                     on wk_dtok_stage_hits gives each read its group (-1: not
 *                     in the map, not counted) until wk_strata_clear. */
int wk_strata_load(wk_ctx* ctx, const char* text, int64_t n, int64_t* n_pairs,
                   int32_t* n_labels, int* status);
int wk_strata_labels(wk_ctx* ctx, int32_t* slot, int64_t* text_off,
                     int32_t* text_len, int32_t cap, int32_t* n);
int wk_strata_groups(wk_ctx* ctx, const int32_t* slot, const int32_t* group,
                     int32_t n);
int wk_strata_clear(wk_ctx* ctx);

/* ---- read maps formatted on the device (csrc/wk_readmap.hpp) ----------------
 * file.write_readmap (file.py:469-500) for blocks the device tokenised: the
 * lines `query[/mate] <tab> taxon` / `query <tab> taxon:count <tab> ...`
 * (sorted by descending count, then id string) are built on the device from the
 * block's text and only the finished text is fetched.  Plain assigners over
 * subjects that all have a taxon (the job sets wk_words_begin accepts for the
 * weighted histogram).
 *   wk_dtok_keep_reads   on: wk_dtok_emit keeps the block's per-read state for
 *                        wk_dtok_readmap (and places the records in read order).
 *   wk_readmap_tables    job j's tables over the current subject table:
 *                        slot_of_subject[s] = compact index of the taxon of
 *                        subject s at the job's rank (the subject's own
 *                        feature for `--rank none`); slot_order[t] = rank of
 *                        slot t's id string among the slots'; shown_off /
 *                        shown = the text printed for slot t (the id, or its
 *                        name under --name-as-id).  Sent again whenever the
 *                        subject table grows.
 *   wk_dtok_readmap      the text of the block emitted last at job j:
 *                        *n_bytes = its size (0: no lines).
 *   wk_dtok_readmap_fetch copies it to out[0, cap). */
int wk_dtok_keep_reads(wk_ctx* ctx, int on);
int wk_readmap_tables(wk_ctx* ctx, int32_t job, const int32_t* slot_of_subject,
                      int32_t n_subjects, const int32_t* slot_order,
                      const uint32_t* shown_off, const char* shown,
                      int32_t n_slots);
int wk_dtok_readmap(wk_ctx* ctx, int32_t job, int64_t* n_bytes);
int wk_dtok_readmap_fetch(wk_ctx* ctx, char* out, int64_t cap);

/* `extra` != 0 — the "ex" flavour (align.parse_sam_file_ex + ordinal_mapper,
 * align.py:350-406, ordinal.py:167-240): wk_dtok_scan also takes POS and CIGAR
 * (start, end, aligned length per line; text beyond [+-]digits and
 * (digits op)+ goes back to the host: status 1), and wk_dtok_stage_hits
 * leaves the block's hits staged exactly as wk_ordinal_stage would have: hits
 * of zero length dropped, the hits of a read (query, mate) contiguous, reads in
 * the order the parser yields them; genome_of_subject[id] = index into the
 * gene tables for tokenizer subject id (-1: none).  wk_ordinal_count /
 * wk_ordinal_match follow. */
int wk_dtok_stage_hits(wk_ctx* ctx, const int32_t* genome_of_subject,
                       int32_t n_subjects, double th, int64_t* n_reads,
                       int64_t* n_hits, int* status);
/* The same, the block's hits placed BEHIND those of the blocks staged this way
 * since the last wk_ordinal_count / wk_ordinal_match: ordinal.py:243-335
 * (flush_chunk) matches a chunk's reads one by one, so how many blocks make a
 * chunk changes nothing -- and the match sorted by genome stripe (O4,
 * csrc/wk_stripe.hpp) pays from a few million hits on, where a 64 MB block of
 * text brings 1.5 M.  *may_wait = 1: `jobs` are of the kind that match serves
 * and the pile is below what it wants (wk_tune "stripes_min"): the caller may
 * stage the next block first; 0: count now.  The pile is one chunk, with one
 * group for all of it (wk_set_uniform_group).  wk_ordinal_stage and
 * wk_dtok_stage_hits refuse (WK_E_STATE) while a pile has not been counted. */
int wk_dtok_stage_hits_append(wk_ctx* ctx, const int32_t* genome_of_subject,
                              int32_t n_subjects, double th,
                              const wk_job* jobs, int32_t n_jobs,
                              int64_t* n_reads, int64_t* n_hits, int* status,
                              int* may_wait);

/* Convenience: stage + classify in one call from host buffers. */
int wk_classify_chunk(wk_ctx* ctx, const wk_job* jobs, int32_t n_jobs,
                      const int32_t* subj, const int32_t* qoff,
                      int64_t n_reads, const int32_t* group, int subj_flags,
                      int32_t* out_assign);

/* Stage one chunk of ordinal-mapper input (replaces the qrys/lens/begs/ends
 * arrays + sub2idx of ordinal.ordinal_mapper, ordinal.py:204-237).
 *   genome[n_hits]  index into the gene tables, -1 = genome without genes
 *   beg/end[n_hits] 0-based start, exclusive end (align.py:382-398)
 *   len[n_hits]     alignment length; rel = ceil(len * th) in fp64
 *                   (ordinal.py:281)
 *   hoff[n_reads+1] CSR offsets: hits of each read (query, mate)
 *   group[n_reads]  as above, may be NULL */
int wk_ordinal_stage(wk_ctx* ctx, const int32_t* genome, const int32_t* beg,
                     const int32_t* end, const uint32_t* len, int64_t n_hits,
                     const int32_t* hoff, int64_t n_reads,
                     const int32_t* group, double th);

/* Match every staged hit against the genes of its genome (replaces
 * ordinal.flush_chunk + match_read_gene / _quart, ordinal.py:243-335,
 * 476-582, 650-811) and leave the per-read gene sets staged as the current
 * classify chunk (subjects = gene feature ids), ready for
 * wk_classify_staged().  A hit (rs, re, rel) matches gene (gs, ge) iff
 * min(ge, re) - max(gs, rs) >= rel (ordinal.py:555,580). */
int wk_ordinal_match(wk_ctx* ctx);

/* The same in one call for jobs that need no per-read result: matches the
 * staged hits and adds the reads' gene sets to the count table under every
 * job (ordinal.flush_chunk + workflow.assign_readmap, ordinal.py:243-335,
 * workflow.py:941-1058).  When every job is a plain WK_MODE_NONE job and the
 * chunk has one group (the genes themselves are the profile's features:
 * classify.assign_none + classify.counter, classify.py:32-51, 144-171) the
 * genes are counted per read straight from the matches, without gene lists;
 * otherwise this is wk_ordinal_match followed by wk_classify_staged.  The
 * staged classify chunk is not valid afterwards. */
int wk_ordinal_count(wk_ctx* ctx, const wk_job* jobs, int32_t n_jobs);

/* With wk_set_option("gene_index_pairs", 1) before wk_set_genes the gene lists
 * of wk_ordinal_match are kept by gene table index as well (and translated to
 * features for the classification): out[i] = index of the i-th (hit, gene)
 * match, in the order of wk_chunk_download's features — what tells genes apart
 * that share a feature (`--trim-sub` next to `--coords`, workflow.py:318-319)
 * when a read map has to list the queries in the reference's order
 * (ordinal.py:290-335). */
int wk_ordinal_pair_genes(wk_ctx* ctx, int32_t* out, int64_t cap);
/* The chunk staged last (wk_chunk_stage / wk_ordinal_stage) has no per-read
 * groups: every read belongs to group `group` (what WK_GROUP_UNIFORM says for
 * wk_chunk_stage). */
int wk_set_uniform_group(wk_ctx* ctx, int32_t group);

/* After wk_ordinal_match: poff[n_hits + 1], the offsets of every hit's genes in
 * the gene lists wk_chunk_download returns (the genes of a read are the
 * concatenation of its hits' genes) — which hit matched which gene
 * (ordinal.flush_chunk's (read, gene) pairs, ordinal.py:321-332). */
int wk_ordinal_hit_offsets(wk_ctx* ctx, int32_t* poff, int64_t cap);

/* Download the staged classify chunk (testing / read-map output): the
 * candidate lists as currently staged (after wk_ordinal_match: gene feature
 * ids per read, duplicates possible when several hits of a read match the same
 * gene).  Pass NULL buffers to query sizes. */
int wk_chunk_download(wk_ctx* ctx, int32_t* subj, int64_t subj_cap,
                      int32_t* qoff, int64_t qoff_cap, int64_t* n_records,
                      int64_t* n_reads);

int wk_get_stats(wk_ctx* ctx, wk_stats* out);
int wk_reset_stats(wk_ctx* ctx);

/* ---- native SAM tokenizer (host, multi-threaded) ------------------------ */
/* Replaces the per-line Python of align.parse_sam_file / parse_sam_file_ex and
 * the packing loops of plain_mapper / ordinal_mapper (align.py:258-406,
 * 550-583; ordinal.py:219-237) for SAM input.  Host-only: no device is needed.
 * A tokenizer owns the subject dictionary (RNAME -> dense subject index in
 * order of first appearance, the indices wk_set_subjects expects). */
int wk_tok_create(int n_threads /* <= 0: all hardware threads */, wk_tok** out);
void wk_tok_destroy(wk_tok* tok);
const char* wk_tok_last_error(const wk_tok* tok);
/* SAM, "ex" flavour with an exclusion set, after the final block of a file:
 * the text of the reads that parse_sam_file_ex_ft's closing statements yield
 * once more when the last query of the file was dropped (they do not look at
 * `keep`, align.py:542-547): the lines still in its pool, under the last
 * query's name.  Empty when the last query was kept.  Tokenising this text as
 * one more block (exclusion switched off) reproduces the reference's output.
 * cap = 0 asks for the length. */
int wk_tok_sam_tail(wk_tok* tok, char* buf, int64_t cap, int64_t* len);

/* Subjects to exclude (align.py:443-469): names are blob[off[i]..off[i+1]). */
int wk_tok_set_exclude(wk_tok* tok, const char* blob, const int32_t* off,
                       int32_t n);
/* Tokenize one block of SAM text.  `first_block`: the block starts the file
 * (leading '@' header lines are skipped).  Unless `final_block`, parsing stops
 * before the last QNAME run (it may continue in the next block); *consumed is
 * the number of bytes used — feed the rest again, followed by more text.
 * `extra` bit 0: also produce POS-1, reference end and aligned length per
 * record ("ex" flavour; zero-length hits are dropped unless bit 1 is set — the
 * coverage mapper, range.py:21, keeps them).  `want_names`: keep a
 * descriptor of every read's QNAME.  Results are held by the tokenizer until
 * the next call; sizes are returned in *n_reads / *n_records. */
int wk_tok_sam(wk_tok* tok, const char* buf, int64_t len, int first_block,
               int final_block, int extra, int want_names, int64_t* consumed,
               int64_t* n_reads, int64_t* n_records);
/* The same for any of the reference's alignment formats (align.py:153-223):
 * simple maps (parse_map_file, align.py:621), BLAST tabular (parse_b6o_file
 * :753 / _ex :807) and PAF (parse_paf_file :984 / _ex :1046) group runs of
 * equal first columns; lines that are not rows of the format are skipped like
 * the reference's `except IndexError: continue`. */
#define WK_FMT_SAM 0
#define WK_FMT_MAP 1
#define WK_FMT_B6O 2
#define WK_FMT_PAF 3
int wk_tok_text(wk_tok* tok, int fmt, const char* buf, int64_t len,
                int first_block, int final_block, int extra, int want_names,
                int64_t* consumed, int64_t* n_reads, int64_t* n_records);
/* *out = offset of the first line at or after `pos` that starts a new run of
 * equal query ids — where one large file may be cut into byte ranges for
 * several processes without splitting a read. */
int wk_tok_boundary(int fmt, int extra, const char* buf, int64_t len,
                    int64_t pos, int64_t* out);
/* Copy the results out: subj[n_records] (subject indices), off[n_reads + 1]
 * (CSR), beg/end/len[n_records] (extra only), qname[n_reads] =
 * (byte offset in buf << 24) | (length << 2) | mate.  NULL skips an array. */
int wk_tok_fetch(wk_tok* tok, int32_t* subj, int32_t* off, int32_t* beg,
                 int32_t* end, uint32_t* len, uint64_t* qname);
/* [*begin, *stop) of a block of SAM text that can be tokenised now: behind the
 * leading '@' header lines (`in_header`: the block starts inside them, as the
 * first block of a file does) up to — unless `final_block` — the start of the
 * last run of equal query ids, which may continue in the next block (what
 * wk_tok_text does with a block before tokenising it; align.py:295-300, 325).
 * WK_E_STATE: nothing complete yet (*stop == *begin). */
int wk_tok_sam_span(const char* buf, int64_t len, int final_block, int in_header,
                    int64_t* begin, int64_t* stop, int* in_header_after);
/* The header state wk_tok_text continues from, for callers that had the device
 * tokenizer (wk_dtok_*) take the blocks before. */
/* The same for any of the formats (WK_FMT_*; header lines exist in SAM only). */
int wk_tok_span(int fmt, int extra, const char* buf, int64_t len, int final_block,
                int in_header, int64_t* begin, int64_t* stop,
                int* in_header_after);
int wk_tok_set_header_state(wk_tok* tok, int in_header);
/* *got bytes of [offset, offset + len) of the open file `fd` into dst, read by
 * all tokenizer threads (pread on slices); short only at the end of the file. */
int wk_tok_read(wk_tok* tok, int fd, int64_t offset, char* dst, int64_t len,
                int64_t* got);
/* The column trim between the page cache and pinned memory: SAM text
 * [begin, begin + want) -- begin a line start -- of src (memory all threads
 * see, e.g. the mapped file) or, src NULL, of the open file fd, which the
 * threads read piece by piece (src_len: the length of either), with every line
 * cut behind its tab number keep_tabs (3:
 * QNAME, FLAG, RNAME, all `line.split('\t', 3)` looks at, align.py:313; 6 for the
 * coord-match, align.py:376; doc/perform.md:122-128 asks the user to do this
 * beforehand), written to dst by all tokenizer threads.  Lines of fewer tabs and
 * lines that hold a carriage return leave whole.  Whole lines only: *consumed =
 * input bytes taken (short when the range ends inside a line, unless it ends src;
 * or when `cap` is reached), *got = bytes written. */
int wk_tok_trim(wk_tok* tok, const char* src, int fd, int64_t src_len,
                int64_t begin, int64_t want, int keep_tabs, char* dst,
                int64_t cap, int64_t* consumed, int64_t* got);
/* Translate subject indices on their way out (wk_tok_fetch's `subj`):
 * subj[i] = map[dictionary id], -1 for ids >= n.  The coord-match stages genome
 * indices of the gene tables (wk_ordinal_stage), which the host derives from the
 * names of wk_tok_new_subjects; map == NULL switches the translation off. */
int wk_tok_set_subject_map(wk_tok* tok, const int32_t* map, int32_t n);
/* The plain flavour's records in the form the weighted histogram of the device
 * streams (wk_words_append): packed[i] = subject index | position of record i
 * in its read << 23 | size << 27, size = the number of records of its read
 * (the k of classify.counter, classify.py:167-170) when that is <=
 * WK_WEIGHT_MAX_K, else position and size are 0; *n_big = reads with more
 * records than that.  Filled by all tokenizer threads straight from
 * their own buffers (no intermediate copy): `packed`, `off` and `qname` may be
 * pinned staging memory.  WK_E_RANGE beyond 2^23 subjects. */
int wk_tok_fetch_packed(wk_tok* tok, uint32_t* packed, int32_t* off,
                        uint64_t* qname, int64_t* n_big);
/* `want_names` of wk_tok_sam is a bit set: 1 = QNAME descriptors, 2 = stratum of
 * every read (read id = QNAME + "" | "/1" | "/2" looked up in the table loaded
 * with wk_tok_strata_load; -1 = not found: the read is skipped by the
 * stratified counters, classify.py:239).  wk_tok_fetch_groups copies them. */
int wk_tok_fetch_groups(wk_tok* tok, int32_t* group);
/* Bit 4 of `want_names`: demultiplexing (workflow.demultiplex, workflow.py:
 * 844-909) — the sample of a read is the text before the first '_' of its read
 * id if anything follows it, else the empty name; samples are interned in
 * order of first appearance.  wk_tok_fetch_samples copies the per-read sample
 * ids; wk_tok_new_samples returns the names first seen since the last call
 * (call with blob == NULL and off == NULL for the count, then with off[n+1]
 * for the sizes, then with blob). */
int wk_tok_fetch_samples(wk_tok* tok, int32_t* sample);
int wk_tok_new_samples(wk_tok* tok, char* blob, int64_t* off, int32_t* n_new);
/* Stratification map of the current sample (file.read_map_uniq +
 * workflow.read_strata; file.py:368-385, workflow.py:912-938): lines
 * "read id <tab> label" with exactly two columns, appended block by block
 * (blocks must end at line ends).  Labels get stratum ids in order of first
 * appearance; a repeated read id keeps its last label. */
int wk_tok_strata_clear(wk_tok* tok);
int wk_tok_strata_load(wk_tok* tok, const char* buf, int64_t len,
                       int64_t* n_entries, int32_t* n_labels);
int wk_tok_strata_labels(wk_tok* tok, char* blob /* NULL: sizes only */,
                         int64_t* off /* [n_labels + 1] */);
/* A tokenizer holds two such tables.  wk_tok_strata_select(tok, 1) makes
 * clear / load / labels work on the one lookups do not use — the map of the next
 * sample is read (by another thread) while the current sample is tokenised —,
 * select(tok, 0) on the one in use again; wk_tok_strata_swap exchanges them (no
 * tokenising call may be running). */
int wk_tok_strata_select(wk_tok* tok, int other);
int wk_tok_strata_swap(wk_tok* tok);
/* Read-map text (file.write_readmap, file.py:469-500), multi-threaded.  One line
 * per assigned read: "read id <tab> name", or "read id <tab> name:n <tab> ..."
 * for a read split over several features; such reads (assign[r] ==
 * WK_ASSIGN_MULTI) take their (feature, count) lists, already sorted by
 * descending count then feature id string, from m_feat/m_count[m_off[i] ..
 * m_off[i+1]) in read order.  Unassigned reads are written as "Unassigned" when
 * `unassigned`, else skipped.  names_blob/names_off give the text printed for
 * feature id f.  With out == NULL only *written (the size) is computed. */
int wk_format_readmap(const char* text, const uint64_t* qname,
                      const int32_t* assign, int64_t n_reads,
                      const int64_t* m_off, const int32_t* m_feat,
                      const int32_t* m_count, const char* names_blob,
                      const int64_t* names_off, int32_t n_names, int unassigned,
                      int n_threads, char* out, int64_t cap, int64_t* written);
/* ---- gzip members of read-map text (host; csrc/wk_deflate.cpp) ------------- */
/* file.openzip(..., 'at') of the read maps (file.py:30-59 via workflow.py:
 * 1042-1046; gzip by default, cli.py:173-176): one standard gzip member per
 * call, its total size in a 'WK' extra subfield so that the members of a map
 * can be found and inflated in parallel (the stratified second pass,
 * workflow.py:912-938).  wk_gz_member returns the member's size, -1 if `cap`
 * is too small; wk_gz_bound(n) is always enough.  wk_crc32 continues `crc`
 * (0 to start) over data[0, n), the gzip CRC-32. */
int64_t wk_gz_bound(int64_t n);
int64_t wk_gz_member(const char* data, int64_t n, char* out, int64_t cap);
uint32_t wk_crc32(uint32_t crc, const char* data, int64_t n);
/* Inflate n members written by wk_gz_member, blob[lo[i], hi[i]) each, on
 * n_threads threads straight into out[off[i], off[i+1]) (off = running sum of
 * the members' ISIZE fields).  0, or -(1 + i) for the first member whose size /
 * CRC-32 / stream is not what its trailer says. */
int64_t wk_gz_inflate_members(const char* blob, const int64_t* lo,
                              const int64_t* hi, int64_t n, char* out,
                              const int64_t* off, int n_threads);

/* ---- gzip input, inflated natively (host; csrc/wk_inflate.cpp) ------------ */
/* file.readzip's decompressor (woltka/file.py:62-129: `gzip -cdfq` as a child,
 * or the gzip module) for alignment files: a regular `.gz` file is mapped and
 * inflated by this library's own table-driven decoder on up to n_threads
 * threads -- one stream cut into chunks that find a block start each and are
 * decoded with an unknown window, resolved in order; members that state their
 * size (BGZF 'BC', this library's 'WK') one task each.  CRC-32 and ISIZE of
 * every member are verified.  wk_gunzip_open: NULL on failure (`err` says why);
 * wk_gunzip_read: the next bytes of text into dst[0, cap) (cap >= 65536),
 * their number, 0 at the end, -1 on damaged data (wk_gunzip_error). */
typedef struct wk_gunzip wk_gunzip;
wk_gunzip* wk_gunzip_open(const char* path, int n_threads, char* err, size_t err_cap);
int64_t wk_gunzip_read(wk_gunzip* h, char* dst, int64_t cap);
const char* wk_gunzip_error(wk_gunzip* h);
void wk_gunzip_close(wk_gunzip* h);

/* Dictionary growth: total subjects, subjects not yet reported, their bytes. */
int wk_tok_subjects(wk_tok* tok, int32_t* n_total, int32_t* n_new,
                    int64_t* new_bytes);
/* Names of the not-yet-reported subjects: blob (may be NULL) and off[n_new+1];
 * marks them reported. */
int wk_tok_new_subjects(wk_tok* tok, char* blob, int32_t* off);

/* Host helper of the hierarchy flattening: DFS pre-order numbers, subtree sizes
 * and depths of a rooted tree given as a parent array (exactly the root is its
 * own parent; siblings keep their input order).  The device tables of
 * wk_set_tree are these numbers applied to the reference's child -> parent
 * dict (tree.py:302-388).  WK_E_STATE: *bad cannot reach the root (a cycle). */
int wk_preorder(const int64_t* parent, int64_t n, int64_t expected_root /* -1: any */,
                int64_t* pre, int64_t* size, int64_t* depth, int64_t* bad);

/* ---- native hierarchy ingest (host, multi-threaded) ---------------------- */
/* Replaces the per-line Python of the hierarchy readers and what follows them
 * in workflow.build_hierarchy (workflow.py:698-815): tree.read_nodes
 * (tree.py:73-101), tree.read_names (tree.py:48-70), file.read_map_1st
 * (file.py:388-406), util.update_dict (util.py:46-75), tree.fill_root
 * (tree.py:302-388), and the flattening of the dicts to the arrays wk_set_tree
 * takes.  A wk_hier holds the three dicts of the reference — child -> parent,
 * node -> rank, node -> name — as one symbol table; every wk_hier_add_text /
 * wk_hier_update is one util.update_dict(dic, other): a key that an earlier
 * update set to another value fails with WK_E_STATE and the reference's
 * message ("Conflicting values found for ..."), a key repeated inside one file
 * takes its last value. */
typedef struct wk_hier wk_hier;
#define WK_HIER_NODES 0 /* tree.read_nodes: "id | parent | rank |" or "id parent [rank]" */
#define WK_HIER_MAP 1   /* file.read_map_1st: first two columns, key -> parent           */
#define WK_HIER_NAMES 2 /* tree.read_names: names.dmp (scientific names) or "id name"    */
#define WK_HIER_PARENT 0 /* fields: tree[key]    */
#define WK_HIER_RANK 1   /*         rankdic[key] */
#define WK_HIER_NAME 2   /*         namedic[key] */
int wk_hier_create(int n_threads /* <= 0: all hardware threads */, wk_hier** out);
void wk_hier_destroy(wk_hier* h);
const char* wk_hier_last_error(const wk_hier* h);
/* One whole file of text.  `rank` (WK_HIER_MAP only, may be NULL): the rank
 * given to every value of the map ("map as rank", workflow.py:789-805).
 * WK_E_ARG = text the native readers leave to the Python readers (a byte
 * >= 0x80 where str.rstrip() looks, a bare '\r', a "\t|" inside a field):
 * nothing was changed, read the file in Python and use wk_hier_update;
 * WK_E_RANGE = a line with fewer than two fields (the reference's IndexError). */
int wk_hier_add_text(wk_hier* h, int kind, const char* buf, int64_t len,
                     const char* rank);
/* util.update_dict(dic, other) with `other` as arrays: key i =
 * kblob[koff[i]..koff[i+1]), value i likewise (WK_HIER_PARENT: is_none[i] != 0
 * means None; is_none may be NULL). */
int wk_hier_update(wk_hier* h, int field, const char* kblob, const int64_t* koff,
                   const char* vblob, const int64_t* voff,
                   const uint8_t* is_none, int64_t n);
/* tree.fill_root + flattening: parents that are not keys join the tree; one
 * crown becomes the root, several get a new root named by the smallest unused
 * positive integer; nodes are numbered in DFS pre-order.  WK_E_STATE: no root
 * (a pure cycle) or a node that cannot reach the root.  No update afterwards. */
int wk_hier_finish(wk_hier* h, int64_t* n_nodes, int32_t* n_ranks);
/* The arrays of wk_set_tree (+ depth), n_nodes each; NULL skips one. */
int wk_hier_arrays(const wk_hier* h, int32_t* parent, int32_t* last,
                   int32_t* rank_code, int32_t* depth);
/* *root_id = 0 (the root's pre-order id) or -1 for an empty tree. */
int wk_hier_root(const wk_hier* h, int32_t* root_id);
/* Pre-order ids of names (-1: not a node): name i = blob[off[i]..off[i+1]). */
int wk_hier_lookup(const wk_hier* h, const char* blob, const int64_t* off,
                   int64_t n, int32_t* out);
/* Names of pre-order ids: off[n + 1] always; bytes when blob != NULL. */
int wk_hier_node_names(const wk_hier* h, const int32_t* ids, int64_t n,
                       char* blob, int64_t cap, int64_t* off);
/* One entry of a dict: *len = -1 when the key is absent, else the value's
 * length (copied to out when it fits cap). */
int wk_hier_get(const wk_hier* h, int field, const char* key, int64_t klen,
                char* out, int64_t cap, int64_t* len);
int64_t wk_hier_size(const wk_hier* h, int field); /* len(dict) */
/* All keys of a dict: off[size + 1] always; bytes when blob != NULL. */
int wk_hier_keys(const wk_hier* h, int field, char* blob, int64_t cap,
                 int64_t* off);
/* Rank vocabulary (rank code = index + 1; n_ranks from wk_hier_finish) and the
 * number of keys that carry each rank. */
int wk_hier_ranks(const wk_hier* h, char* blob, int64_t cap, int64_t* off,
                  int64_t* used);

/* ---- native gene coordinates reader (host) -------------------------------- */
/* ordinal.load_gene_coords + encode_genes (ordinal.py:338-473): ">name" /
 * "# name" lines start a nucleotide (a doubled marker is ignored), other lines
 * are "gene <tab> beg <tab> end"; start0 = min(beg, end) - 1, end = max(beg,
 * end); genes of a nucleotide stably sorted by start0; isdup = a gene id was
 * seen twice.  WK_E_ARG: the reference's error (message: wk_coords_error);
 * WK_E_STATE: text left to the Python reader (non-ASCII white space, numbers
 * beyond plain digits, a bare '\r', coordinates before the first nucleotide). */
typedef struct wk_coords wk_coords;
int wk_coords_parse(const char* buf, int64_t len, wk_coords** out);
const char* wk_coords_error(const wk_coords* c);
int wk_coords_sizes(const wk_coords* c, int32_t* n_genomes, int32_t* n_genes,
                    int64_t* genome_bytes, int64_t* gene_bytes, int* isdup);
/* goff[n_genomes + 1], start0/end/findex[n_genes] (findex = the gene's place
 * among its nucleotide's lines: the index in encode_genes' codes), names as
 * blob + off[n + 1]. */
int wk_coords_fetch(const wk_coords* c, int32_t* goff, int32_t* start0,
                    int32_t* end, int32_t* findex, char* genome_blob,
                    int64_t* genome_off, char* gene_blob, int64_t* gene_off);
void wk_coords_free(wk_coords* c);
/* The n strings blob[off[i], off[i + 1]) written to `out` one after the other,
 * each followed by `sep` (off[n] - off[0] + n bytes): what a host layer splits
 * into its own string objects in one call. */
int wk_blob_join(const char* blob, const int64_t* off, int64_t n, char sep, char* out);
/* The rows of a one-sample TSV table (table.prep_table + table.write_tsv,
 * woltka/table.py:29-66, 247-283): the n feature ids of `keys` (joined by '\n')
 * in ascending order, each as "id \t value \n", rows of value 0 left out;
 * `out` needs keys_len + 24 n bytes.  *n_rows = the features written. */
int wk_table_body(const char* keys, int64_t keys_len, const int64_t* values, int64_t n, int threads,
                  char* out, int64_t cap, int64_t* out_len, int64_t* n_rows);

/* The rows of a TSV table with n_cols sample columns and, for stratified
 * profiles, `stratum|feature` row names (same reference code): row r is named
 * prefixes[prefix_of_row[r]] | names[name_of_row[r]] (no prefix when
 * prefix_of_row is NULL, its entry negative or the prefix empty) and holds
 * values[r * n_cols ..).  Sorted by (prefix, name) in byte order like
 * `sorted(allkeys(profile))` over (stratum, feature) tuples; all-zero rows are
 * left out.  prefixes / names: the strings joined by '\n'. */
int wk_table_rows(const char* prefixes, int64_t prefixes_len, int32_t n_prefixes,
                  const char* names, int64_t names_len, int32_t n_names,
                  const int32_t* prefix_of_row, const int32_t* name_of_row,
                  const int64_t* values, int64_t n_rows, int32_t n_cols,
                  char* out, int64_t cap, int64_t* out_len, int64_t* rows_written);


#ifdef __cplusplus
}
#endif
#endif /* WOLTKA_HIP_H */
