/*
 * meryl_gpu_count.h -- C ABI of the MI355X-native `meryl count` engine.
 *
 * This is the drop-in boundary for the reference's counting engine
 * (marbl/meryl, paths relative to the reference root):
 *
 *   upstream   bool merylInput::loadBases(char *seq, uint64 maxLength,
 *                                         uint64 &seqLength, bool &endOfSequence)
 *              src/meryl/merylInput.H:67-70, called from
 *              src/meryl/merylOp-countThreads.C:180-182
 *   engine     merylOperation::configureCounting()  src/meryl/merylOp-count.C:300-403
 *              merylOperation::countThreads()       src/meryl/merylOp-countThreads.C:385-474
 *   downstream merylFileWriter::initialize(wPrefix) / numberOfFiles() /
 *              firstPrefixInFile() / lastPrefixInFile()
 *              merylBlockWriter::addBlock(prefix, nKmers, suffixes, counts)
 *              call sites src/meryl/merylOp-countThreads.C:404,453-464,
 *              src/meryl/merylCountArray.C:472-475
 *
 * A maintainer replaces the body of countThreads() with: mgc_open, a loop of
 * mgc_push_bases over merylInput::loadBases, then mgc_finish whose callback
 * calls merylBlockWriter::addBlock -- see INTEGRATION.md.
 *
 * Plain C types only: pointers and sizes, no C++/torch types.  Functions
 * return MGC_OK (0) or a negative MGC_E* code instead of the reference's
 * fprintf(stderr)+exit(1); mgc_last_error() gives the text.
 *
 * Pointers named d_* are DEVICE pointers (HBM of the current HIP device);
 * `stream` is a hipStream_t passed as void* (NULL = the null stream).
 */
#ifndef MERYL_GPU_COUNT_H
#define MERYL_GPU_COUNT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGC_OK            0
#define MGC_EINVAL       -1    /* bad argument (k, mode, NULL pointer, capacity) */
#define MGC_ENOMEM       -2    /* host or device allocation failed */
#define MGC_EHIP         -3    /* a HIP runtime call or kernel failed */
#define MGC_ESTATE       -4    /* call out of order (e.g. finish before count) */
#define MGC_EUNSUPPORTED -5    /* valid in the reference, not implemented here yet */
#define MGC_ETIMEOUT     -6    /* an in-kernel bounded spin expired (never hangs) */
#define MGC_EFORMAT      -7    /* mgc_end_text: the file is not what the device parser handles (see there) */

/* opCount / opCountForward / opCountReverse, src/meryl/merylOp.H:40-42 and
 * src/meryl/merylOp-countThreads.C:241-258 */
#define MGC_MODE_CANONICAL 0
#define MGC_MODE_FORWARD   1
#define MGC_MODE_REVERSE   2

#define MGC_NUM_FILES_BITS 6   /* 64 files: src/meryl/merylOp-count.C:185-187, documentation/source/usage.rst:18 */
#define MGC_NUM_FILES      64

/* ------------------------------------------------------------------------
 * Configuration -- replaces merylOperation::configureCounting
 * (src/meryl/merylOp-count.C:300-403).  Pure host arithmetic.
 * ---------------------------------------------------------------------- */
#define MGC_MAX_COUNT_SUFFIX 32

typedef struct mgc_count_config {
  /* in */
  uint32_t k;                     /* kmerTiny::merSize(), 1..64 */
  int32_t  mode;                  /* MGC_MODE_* */
  uint64_t n_kmers_estimate;      /* n= / guesstimateNumberOfkmersInInput (:317,449) */
  uint64_t memory_allowed;        /* memory= in bytes (merylCommandBuilder.C:299-302) */
  uint32_t threads;               /* threads= (host threads used by mgc_finish) */
  uint32_t count_suffix_length;   /* count-suffix= length (see count_suffix below); forces simple mode (:379-382) */
  uint32_t homopoly_compress;     /* `compress` (merylInput.C:261-262) */
  uint32_t page_size;             /* 0 -> 4096 (getPageSize()) */
  uint32_t sizeof_count_array;    /* 0 -> 3232 (sizeof(merylCountArray), merylCountArray.H:44,71-74) */
  /* out */
  int32_t  use_simple;            /* :368-382 */
  uint32_t w_prefix;              /* wPrefix_ */
  uint64_t n_prefix;              /* nPrefix_ */
  uint32_t w_data;                /* wData_ = 2k - wPrefix */
  uint32_t n_batches;             /* incl. the reference's post-increment quirk (:355-358) */
  uint64_t memory_used;           /* what the "Configured ... mode for %.3f GB" line prints (:398-401) */
  /* in (at the end: added after the first layout) */
  char     count_suffix[MGC_MAX_COUNT_SUFFIX + 4];   /* count-suffix=<bases> (merylCommandBuilder.C:271-272, merylOp.H:139-147):
                                      only k-mers -- the canonical / forward / reverse one that is counted -- ENDING in these
                                      bases are counted (merylOp-countSimple.C:50-58,88-93); count_suffix_length of them,
                                      NUL-terminated; needs k - length >= 3 */
  /* in: value-labelled k-mers (meryl2).  The count hands ONE constant label to the writer when it dumps a block:
   * addCountedBlock(prefix, nKmers, suffix, counts, labels = nullptr, label = _lConstant)
   * (src/meryl2/merylCountArray.C:469-471, src/meryl2/merylOp-countThreads.C:363-367,464-468; `label=#<n>` sets it,
   * src/meryl2/merylCommandBuilder-isAssign.C:124; the global `-l <bits>` fixes the width, src/meryl2/merylGlobals.C:75-77).
   * label_size = 0: unlabelled (meryl v1 behaviour). */
  uint32_t label_size;            /* bits of label per k-mer, 0..64 */
  uint32_t reserved0;
  uint64_t label_constant;
} mgc_count_config;

int mgc_configure_counting(mgc_count_config *cfg);

/* Formats the Canu-parsed line of src/meryl/merylOp-count.C:398-401 into buf. */
int mgc_format_configured_line(const mgc_count_config *cfg, char *buf, size_t buflen);

/* ------------------------------------------------------------------------
 * Device-level operators (stateless; caller owns all memory).  These are the
 * hot path: the HIP kernels that replace insertKmers (merylOp-countThreads.C:
 * 235-280), merylCountArray::add/get (merylCountArray.C:490-847) and
 * countSingleKmers (merylCountArray.C:323-365).  Keys are full k-mers
 * (prefix<<wData | suffix): uint64 for k <= 32 (key_words = 1), 16-byte
 * little-endian {lo, hi} pairs for k in 33..64 (key_words = 2) -- the same
 * 128-bit kmdata the reference uses (src/tests/merylCountArrayTest.C:27-31).
 * ---------------------------------------------------------------------- */

/* Number of partition buckets is 2^bucket_bits, bucket = key >> (2k-bucket_bits);
 * bucket_bits = 6 gives the reference's 64 files, 0 a single bucket. */
#define MGC_MAX_BUCKET_BITS 10

/* Scratch needed by mgc_dev_kmer_histogram / mgc_dev_kmer_partition. */
size_t mgc_dev_partition_workspace_bytes(uint32_t bucket_bits);

/* Pass 1: count k-mer instances per bucket.  Writes d_bucket_counts[2^bucket_bits]
 * (uint64, overwritten) and fills the workspace with the per-workgroup counts
 * pass 2 needs.  Bases are ASCII; any byte that is not ACGTacgt (e.g. the '.'
 * breakers of merylOp-countThreads.C:196,214-215, or N) breaks the k-mer. */
int mgc_dev_kmer_histogram(const uint8_t *d_bases, uint64_t n_bases, uint32_t k, int mode,
                           uint32_t bucket_bits, uint64_t *d_bucket_counts,
                           void *d_workspace, size_t workspace_bytes, void *stream);

/* Pass 2: pack every k-mer instance (2 bits/base, A0 C1 T2 G3, first base most
 * significant), pick fmer/rmer per `mode`, and scatter it into its bucket's
 * region: bucket b occupies d_keys[d_bucket_starts[b] ...).  d_bucket_starts
 * holds 2^bucket_bits uint64 offsets (in keys) chosen by the caller from the
 * pass-1 counts (any layout with enough room per bucket).  Must be called
 * with the same bases/k/mode/bucket_bits and the workspace left by pass 1.
 * Order inside a bucket is unspecified (the sort follows). */
int mgc_dev_kmer_partition(const uint8_t *d_bases, uint64_t n_bases, uint32_t k, int mode,
                           uint32_t bucket_bits, const uint64_t *d_bucket_starts,
                           void *d_keys, void *d_workspace, size_t workspace_bytes, void *stream);

/* mgc_dev_kmer_histogram + the k-mers per top FIFTEEN bits (d_fine_hist: uint64[2^15], zeroed here), one pass over the bases;
 * bucket_bits 6..8.  What a sharded count's senders run: the bucket counts feed the routing plan, the workspace rows the
 * partition, the fifteen-bit histogram -- summed over the ranks -- the owners' first grouping digit (mgc_count_buckets_into). */
int mgc_dev_kmer_histogram_fine(const uint8_t *d_bases, uint64_t n_bases, uint32_t k, int mode, uint32_t bucket_bits,
                                uint64_t *d_bucket_counts, uint64_t *d_fine_hist, void *d_workspace, size_t workspace_bytes, void *stream);

/* LSB radix sort of keys (key_words 1 or 2) on bits [begin_bit, end_bit).
 * Ping-pongs between d_keys and d_alt (both n keys); *result_in_alt tells
 * where the sorted keys ended up. */
size_t mgc_dev_sort_workspace_bytes(uint64_t n);
int mgc_dev_radix_sort(void *d_keys, void *d_alt, uint64_t n, uint32_t key_words,
                       uint32_t begin_bit, uint32_t end_bit,
                       void *d_workspace, size_t workspace_bytes,
                       int *result_in_alt, void *stream);
/* The GROUPING passes the count path uses on a file's top bits (not a sort): the keys come out grouped by bits
 * [begin_bit, end_bit), groups ascending, members in ANY order (unstable).  Defined for ranges of at most two digits
 * (18 bits) and n < 2^30; anything else takes the stable sort above.  Same arguments and workspace.  Replaces the top-bit
 * part of unpackSuffixes + std::sort (merylCountArray.C:276-289,330), which the sub-bucket count does not need ordered. */
int mgc_dev_radix_group(void *d_keys, void *d_alt, uint64_t n, uint32_t key_words,
                        uint32_t begin_bit, uint32_t end_bit,
                        void *d_workspace, size_t workspace_bytes,
                        int *result_in_alt, void *stream);

/* Run-length count of a sorted key array (countSingleKmers' two passes,
 * merylCountArray.C:334-358).  Step 1 returns the number of distinct keys
 * (synchronises the stream); step 2 writes d_unique[n_distinct] and
 * d_counts[n_distinct] (uint32, wraps mod 2^32 like merylCountArray.C:357). */
size_t mgc_dev_rle_workspace_bytes(uint64_t n);
int mgc_dev_rle_count(const void *d_sorted, uint64_t n, uint32_t key_words, void *d_workspace,
                      size_t workspace_bytes, uint64_t *n_distinct, void *stream);
int mgc_dev_rle_emit(const void *d_sorted, uint64_t n, uint32_t key_words, void *d_workspace,
                     size_t workspace_bytes, void *d_unique, uint32_t *d_counts, void *stream);

/* d_block_start[p] = index of the first distinct key with (key >> w_data) >= p,
 * for p in [0, n_prefix]; block p of the database is
 * [d_block_start[p], d_block_start[p+1]) -- the (prefix, nKmers) of addBlock. */
int mgc_dev_block_offsets(const void *d_unique, uint64_t n_distinct, uint32_t key_words, uint32_t w_data,
                          uint64_t n_prefix, uint64_t *d_block_start, void *stream);

/* Merge of two (k-mer, value) streams with distinct ascending keys -- the two-input step of the reference's k-way merge
 * (merylOperation::nextMer, src/meryl/merylOp-nextMer.C:478-641: smallest k-mer over the inputs, values of the inputs that
 * hold it combined) and of merylBlockWriter::finish() folding the batches a memory-limited count spilled.  Step 1 returns
 * the output length (synchronises the stream), step 2 writes d_keys_out / d_counts_out (that many entries), with the
 * same inputs and the workspace step 1 left.  Sums wrap mod 2^32 like the reference's kmvalu arithmetic. */
#define MGC_MERGE_UNION_SUM     0     /* opUnionSum      src/meryl/merylOp-nextMer.C:571-573 (findSumCount :43-48) */
#define MGC_MERGE_UNION_MIN     1     /* opUnionMin      :563-565 */
#define MGC_MERGE_UNION_MAX     2     /* opUnionMax      :567-569 */
#define MGC_MERGE_INTERSECT_SUM 3     /* opIntersectSum  :590-593 (k-mers in BOTH inputs) */
#define MGC_MERGE_INTERSECT_MIN 4     /* opIntersectMin  :580-583 */
#define MGC_MERGE_INTERSECT_MAX 5     /* opIntersectMax  :585-588 */
#define MGC_MERGE_INTERSECT     6     /* opIntersect     :575-578: in both, the FIRST input's value */
#define MGC_MERGE_SUBTRACT      7     /* opSubtract      :595-602 + subtractCount :51-62: in A; a - b while a > b, otherwise dropped */
#define MGC_MERGE_DIFFERENCE    8     /* opDifference    :604-607: in A and not in B */
#define MGC_MERGE_SYMMETRIC_DIFFERENCE 9  /* opSymmetricDifference :609-612: in exactly one input */
#define MGC_MERGE_UNION        10     /* opUnion         :559-561: value = number of inputs holding the k-mer (mgc_db_merge only) */
size_t mgc_dev_merge_workspace_bytes(uint64_t na, uint64_t nb);
int mgc_dev_merge_count(const void *d_keys_a, uint64_t na, const void *d_keys_b, uint64_t nb, uint32_t key_words, int op,
                        void *d_workspace, size_t workspace_bytes, uint64_t *n_out, void *stream);
int mgc_dev_merge_emit(const void *d_keys_a, const uint32_t *d_counts_a, uint64_t na,
                       const void *d_keys_b, const uint32_t *d_counts_b, uint64_t nb, uint32_t key_words, int op,
                       void *d_workspace, size_t workspace_bytes, void *d_keys_out, uint32_t *d_counts_out, void *stream);
/* step 1 with the values at hand: MGC_MERGE_SUBTRACT needs them to know what is written (mgc_dev_merge_count refuses it) */
int mgc_dev_merge_count_values(const void *d_keys_a, const uint32_t *d_counts_a, uint64_t na, const void *d_keys_b,
                               const uint32_t *d_counts_b, uint64_t nb, uint32_t key_words, int op, void *d_workspace,
                               size_t workspace_bytes, uint64_t *n_out, void *stream);

/* One (k-mer, value) stream through a single-input operation of src/meryl/merylOp-nextMer.C:490-557: the value filters
 * (the value passes or the k-mer is dropped) and the arithmetic operations (with the reference's overflow / underflow /
 * divide-by-zero results; a k-mer whose new value is 0 is dropped, :470-474).  Two steps like the merge. */
#define MGC_VALUE_LESS_THAN     0     /* opLessThan     :490-492  value <  constant */
#define MGC_VALUE_GREATER_THAN  1     /* opGreaterThan  :494-496 */
#define MGC_VALUE_AT_LEAST      2     /* opAtLeast      :498-500 */
#define MGC_VALUE_AT_MOST       3     /* opAtMost       :502-504 */
#define MGC_VALUE_EQUAL_TO      4     /* opEqualTo      :506-508 */
#define MGC_VALUE_NOT_EQUAL_TO  5     /* opNotEqualTo   :510-512 */
#define MGC_VALUE_INCREASE      6     /* opIncrease     :514-519 */
#define MGC_VALUE_DECREASE      7     /* opDecrease     :521-526 */
#define MGC_VALUE_MULTIPLY      8     /* opMultiply     :528-533 */
#define MGC_VALUE_DIVIDE        9     /* opDivide       :535-540 */
#define MGC_VALUE_DIVIDE_ROUND 10     /* opDivideRound  :541-550 */
#define MGC_VALUE_MODULO       11     /* opModulo       :552-557 */
size_t mgc_dev_select_workspace_bytes(uint64_t n);
int mgc_dev_select_count(const void *d_keys, const uint32_t *d_values, uint64_t n, uint32_t key_words, int value_op, uint64_t constant,
                         void *d_workspace, size_t workspace_bytes, uint64_t *n_out, void *stream);
int mgc_dev_select_emit(const void *d_keys, const uint32_t *d_values, uint64_t n, uint32_t key_words, int value_op, uint64_t constant,
                        void *d_workspace, size_t workspace_bytes, void *d_keys_out, uint32_t *d_values_out, void *stream);

/* Homopolymer compression of a base stream (the `compress` word: merylInput.C:
 * 237-240,261-268 calls homopolyCompress() on every chunk loadBases returns,
 * carrying the last byte across chunks of a sequence).  Drops every byte equal,
 * ignoring case, to the byte before it; d_out needs n bytes; *n_out = new length.
 * The session applies this itself when cfg.homopoly_compress is set. */
size_t mgc_dev_homopoly_workspace_bytes(uint64_t n);
int mgc_dev_homopoly_compress(const uint8_t *d_in, uint64_t n, uint8_t *d_out, uint64_t *n_out,
                              void *d_workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------
 * Session -- replaces merylOperation::countThreads
 * (src/meryl/merylOp-countThreads.C:385-474).
 * ---------------------------------------------------------------------- */
typedef struct mgc_session mgc_session;

/* cfg must have been through mgc_configure_counting (w_prefix decides the
 * block structure, countThreads.C:404).  device < 0 keeps the current device. */
mgc_session *mgc_open(const mgc_count_config *cfg, int device);
void         mgc_close(mgc_session *s);
const char  *mgc_last_error(const mgc_session *s);   /* s may be NULL: last open/config error */

/* Same contract as merylInput::loadBases's output (merylInput.H:67-70): a run
 * of bases of the current sequence; end_of_sequence != 0 appends the '.'
 * breaker the reference's loader appends (merylOp-countThreads.C:214-215).
 * Bases are copied; the caller may reuse the buffer on return. */
int mgc_push_bases(mgc_session *s, const char *bases, size_t len, int end_of_sequence);

/* Out-of-core input (the analogue of writeBatch's memory-full spill, merylOp-countThreads.C:323-379, and of
 * merylBlockWriter::finish() merging the iterations, :461-464): bases pushed from the host travel through two pinned
 * buffers (asynchronous uploads) into a staging buffer in HBM; when the staged bases -- pushed or parsed from text --
 * reach what one pass can hold, everything up to the last sequence boundary is counted as one BATCH by a worker thread
 * while the caller keeps pushing into a second staging buffer.  A batch's (k-mer, count) result is parked as a sorted
 * RUN -- in HBM while the result budget lasts, in pinned host DRAM otherwise (include/meryl_db.h, mgc_runs_*) -- and the
 * runs are merged ONCE, when the count ends:
 *   - every run still in HBM and their merge fits: one device-resident result, every result call works as after a
 *     single pass;
 *   - otherwise the result is OUT OF CORE (mgc_result_out_of_core() == 1; mgc_get_result_info reports n_distinct = 0
 *     until it has been delivered): mgc_write_database and mgc_finish / mgc_finish_labelled merge the runs chunk by
 *     chunk into the consumer, the calls that hand out the whole result (mgc_copy_result*, mgc_get_result_device)
 *     return MGC_ESTATE.  The result may be larger than HBM; host DRAM bounds it.
 * By default the batch size is derived from the free HBM at the first input (mgc_set_batch_bases overrides it: bases per
 * batch) and the runs may keep 60 % of the HBM that is free when the first batch has been counted
 * (mgc_set_result_budget: bytes; 0 restores the default). */
int mgc_set_batch_bases(mgc_session *s, uint64_t bases_per_batch);
/* Optional, right after mgc_open: about how many bases will be pushed.  The count's largest buffer is then allocated by a
 * helper thread while the caller reads and uploads its input (a first large hipMalloc is slow, and nothing needs the
 * memory before mgc_count).  Ignored when the input is expected to take batches; a wrong estimate only costs time. */
int mgc_prepare(mgc_session *s, uint64_t expected_bases);
int mgc_set_result_budget(mgc_session *s, uint64_t device_bytes);
int mgc_result_out_of_core(const mgc_session *s);

/* Sequence-file TEXT instead of bases: the raw bytes of a FASTA or FASTQ file (after any decompression), in
 * chunks of any size and alignment.  The library stages them through pinned buffers, uploads them and parses
 * them ON THE DEVICE into the same base stream mgc_push_bases builds ('.' between sequences; headers, qualities
 * and line ends dropped) -- the device-side replacement of dnaSeqFile::loadBases behind
 * merylInput::loadBases (src/meryl/merylInput.C:245-271) and of the loader's chunk assembly
 * (src/meryl/merylOp-countThreads.C:138-231).  FASTA may be multi-line; FASTQ must be strict four-line records:
 * every '@' and '+' line start is checked on the device and mgc_end_text returns MGC_EFORMAT if the structure
 * does not hold -- the file's output is then already rolled back and the caller feeds the file through
 * mgc_push_bases instead (include/meryl_seq.h reads multi-line FASTQ).  Text input is batched like pushed bases
 * (a batch may be cut inside a file: if such a file is refused AFTER part of it was counted, mgc_end_text returns
 * MGC_EINVAL instead of MGC_EFORMAT -- nothing can be rolled back then); it may be mixed with mgc_push_bases. */
#define MGC_TEXT_FASTA 1
#define MGC_TEXT_FASTQ 2
int mgc_reserve_text(mgc_session *s, uint64_t text_bytes);       /* optional: expected total, avoids regrowth */
int mgc_begin_text(mgc_session *s, int format);
int mgc_push_text(mgc_session *s, const char *text, size_t len);
int mgc_end_text(mgc_session *s);
/* One whole UNCOMPRESSED FASTA/FASTQ file (format 0 = tell from the first record): begin + push + end in one call, the
 * file read by `reader_threads` threads (0 = default) straight into pinned upload buffers.  Return codes as mgc_end_text. */
int mgc_push_text_file(mgc_session *s, const char *path, int format, int reader_threads);
/* A byte window [begin, end) of such a file -- begin at a record start (mgc_text_record_start), end = where the next reader
 * begins (or anything >= the file size): the ranks of a node count read disjoint windows of the input, each through its
 * own device's link.  mgc_text_record_start: the first record start at or after `offset` (FASTA: a line starting with
 * '>'; FASTQ: a line starting with '@' whose next-but-one line starts with '+'); the file size when there is none;
 * format 0 = sniff.  Host I/O only, no device. */
int mgc_push_text_file_range(mgc_session *s, const char *path, int format, int reader_threads, uint64_t begin, uint64_t end);
int mgc_text_record_start(const char *path, int format, uint64_t offset, uint64_t *start);
/* One whole BGZF file of FASTA/FASTQ text (bgzip: independent gzip members of <= 64 KiB with their compressed size in a 'BC' extra
 * field, SAMv1 4.1): mapped, its blocks inflated by `threads` threads (0 = default) straight into the pinned upload buffers, uploaded
 * and parsed in order.  The reference reads every compressed input through ONE decoder (its second loader thread,
 * src/meryl/merylOp-countThreads.C:162-168).  MGC_EFORMAT: not BGZF / a corrupt block / neither FASTA nor strict FASTQ (nothing of
 * the file stays in the session unless part of it was already counted as a batch).  mgc_is_bgzf_file: 1 when the file starts with a
 * BGZF block (host I/O only). */
int mgc_push_text_bgzf_file(mgc_session *s, const char *path, int format, int threads);
int mgc_is_bgzf_file(const char *path);

/* Bases already resident in HBM (breakers included).  The buffer is borrowed
 * until mgc_count returns.  May be called once per session. */
int mgc_push_bases_device(mgc_session *s, const uint8_t *d_bases, uint64_t n_bases);

/* The base stream staged so far -- everything pushed (host bases, parsed text, files), breakers included, resident
 * on the session's device -- without counting it: for callers that route the bases themselves (the `gpus=` option
 * of the CLI hands slices of it to mgc_count_node).  The view is valid until the next push, mgc_count or
 * mgc_close.  MGC_ESTATE once a batch has been counted out of core (the early bases are gone by then). */
int mgc_staged_bases(mgc_session *s, const uint8_t **d_bases, uint64_t *n_bases);

/* Runs histogram -> partition -> per-file radix sort -> run-length count ->
 * block offsets over everything pushed.  Results stay in HBM. */
int mgc_count(mgc_session *s);

/* Owner side of a sharded (multi-GPU) count: the k-mers are already extracted (canonical 2-bit encoding, uint64 for
 * k <= 32, {lo,hi} for k > 32) and laid out FILE-MAJOR in device memory -- file f (the top six bits of the k-mer,
 * merylOp-countThreads.C:255-262 prefix routing restricted to the 64 output files) occupies file_counts[f] consecutive
 * keys, files ascending; files this rank does not own have count 0.  Runs everything mgc_count runs after the
 * partition (grouping passes, LDS finish, blocks) IN PLACE on d_keys; results are read like after mgc_count.
 * The caller must have completed all writes to d_keys (synchronise the producing stream) before the call. */
int mgc_count_partitioned(mgc_session *s, void *d_keys, const uint64_t *file_counts /*[64]*/, void *reserved);

/* The same with a finer granularity: 2^bucket_bits buckets (6..10 bits: the top bucket_bits bits of the k-mer, i.e. every
 * file cut into 2^(bucket_bits-6) ranges), bucket-major, bucket_counts[2^bucket_bits].  A node of N GPUs routes
 * 64*N buckets so that an owner-side bucket is as large as a single-GPU file (a whole file would be N times larger
 * and need a third grouping pass). */
int mgc_count_buckets(mgc_session *s, void *d_keys, uint32_t bucket_bits, const uint64_t *bucket_counts);
/* ... and the packed result -- distinct k-mers ascending, their counts -- written STRAIGHT into the caller's device buffers when
 * it fits (capacity in k-mers; *n_distinct tells): the waves of a sharded count are counted into one pre-sized result instead of
 * being copied out of the session and concatenated (the reference's threads hand their blocks to ONE writer the same way,
 * merylOp-countThreads.C:452-459).  *n_distinct > capacity: nothing was written there, the result is in the session (read it with
 * mgc_copy_result_device) -- the caller grows its buffer.  The session's result views point into the caller's buffers until the
 * next count. */
int mgc_count_buckets_into(mgc_session *s, void *d_keys, uint32_t bucket_bits, const uint64_t *bucket_counts,
                           void *d_out_keys, uint32_t *d_out_counts, uint64_t capacity, uint64_t *n_distinct,
                           const uint64_t *d_fine_hist /* optional (device, uint64[2^15]): k-mers per top FIFTEEN bits over ALL ranks'
                           reads of this batch (mgc_dev_kmer_histogram_fine, summed) -- with bucket_bits <= 8 the owner's grouping passes
                           take their first digit's histogram from it instead of reading the keys for one */);

typedef struct mgc_result_info {
  uint64_t n_bases;
  uint64_t n_instances;           /* k-mer instances (sum of counts) */
  uint64_t n_distinct;
  uint32_t w_prefix, w_data;
  uint64_t n_prefix;
  uint64_t file_instances[MGC_NUM_FILES];   /* instances per file = the 6-bit histogram */
} mgc_result_info;
int mgc_get_result_info(const mgc_session *s, mgc_result_info *info);

/* Device-to-device copy of the result (distinct k-mers ascending, their counts) into caller-owned device buffers of
 * n_distinct keys / counts; completes before returning.  Either pointer may be NULL. */
int mgc_copy_result_device(mgc_session *s, void *d_keys_out, uint32_t *d_counts_out);

/* Device views of the result (valid until mgc_close / the next mgc_count). */
int mgc_get_result_device(const mgc_session *s, const void **d_unique, const uint32_t **d_counts,
                          const uint64_t **d_block_start, uint32_t *key_words);

/* Copies the result to host arrays sized from mgc_get_result_info
 * (keys_lo/keys_hi/counts: n_distinct; block_start: n_prefix+1).  Any pointer
 * may be NULL; keys_hi is zero-filled for k <= 32. */
int mgc_copy_result(const mgc_session *s, uint64_t *keys_lo, uint64_t *keys_hi, uint32_t *counts,
                    uint64_t *block_start);

/* Delivery in the reference's addBlock convention (merylCountArray.C:472-475;
 * merylOp-countThreads.C:452-459): for every file ff, for every prefix of the
 * file in ascending order -- empty blocks included -- cb(ctx, prefix, nKmers,
 * suffix_lo, suffix_hi, counts).  Suffixes are the low w_data bits of the
 * k-mer split in two uint64 halves (suffix_hi is NULL while k <= 32); the
 * callee must not keep the pointers (caller-owned, as in the reference).
 * Files are delivered from up to `host_threads` threads concurrently, one
 * file per thread, exactly like the reference's `omp parallel for` over files. */
typedef int (*mgc_block_cb)(void *ctx, uint64_t prefix, uint64_t n_kmers,
                            const uint64_t *suffix_lo, const uint64_t *suffix_hi,
                            const uint32_t *counts);
int mgc_finish(mgc_session *s, mgc_block_cb cb, void *ctx, int host_threads);

/* The same in meryl2's addCountedBlock convention (src/meryl2/merylCountArray.C:469-471): `labels` is NULL and
 * `label` is the session's constant label (cfg.label_constant) for every k-mer of the block. */
typedef int (*mgc_block_cb2)(void *ctx, uint64_t prefix, uint64_t n_kmers,
                             const uint64_t *suffix_lo, const uint64_t *suffix_hi,
                             const uint32_t *counts, const uint64_t *labels, uint64_t label);
int mgc_finish_labelled(mgc_session *s, mgc_block_cb2 cb, void *ctx, int host_threads);

/* Per-stage device timings of the last mgc_count (HIP events on the session's
 * stream).  Enable before mgc_count. */
#define MGC_STAGE_HISTOGRAM 0
#define MGC_STAGE_PARTITION 1
#define MGC_STAGE_SORT      2
#define MGC_STAGE_RLE       3
#define MGC_STAGE_BLOCKS    4
#define MGC_NUM_STAGES      5
typedef struct mgc_profile {
  double   stage_ms[MGC_NUM_STAGES];
  uint32_t stage_launches[MGC_NUM_STAGES];
  double   sort_pass_ms_total;     /* sum over the radix scatter-pass kernels only */
  uint32_t sort_pass_launches;
  uint64_t sort_pass_keys;         /* keys moved by those launches (sum of n per launch) */
  double   total_ms;
  double   merge_ms;               /* out-of-core: device merges of batch results into the running result (all batches) */
  uint32_t n_batches;              /* batches the input was counted in (1 = single pass) */
  uint32_t reserved;
  /* the same pass launches split by position: [0] a file's first pass, [1] its later passes -- with their ALGORITHMIC
   * bytes (key bytes read + written; narrowed files: 8 + 4 in the first pass, 4 + 4 in the second) */
  double   pass_ms[2];
  uint64_t pass_bytes[2];
  uint64_t pass_keys[2];
  uint32_t pass_launches[2];
  /* the sub-bucket count kernels (hash_count* / bitmap_count / lds_sort_count: one launch per file, on two alternating
   * streams): sum of the launches' own durations (HIP events on the stream each is launched on), their ALGORITHMIC bytes
   * (the keys read: 4 B narrowed, 8 / 16 B otherwise; the distinct suffixes + counts written) and keys */
  double   finish_ms;
  uint64_t finish_bytes, finish_keys;
  uint32_t finish_launches;
  uint32_t wide_msd_files;         /* files whose WHOLE keys took the high-digit-first grouping passes (mgc_device.h, launch_group_wide) */
  /* round 6: which plans the files took */
  uint32_t stream_files;           /* files on the distinct-sized count (hash_count_stream_kernel) and its coarser sub-buckets */
  uint32_t k96_files;              /* files that lay as 12-byte K96 records (k = 33..51) */
  uint32_t k96_widened_files;      /* ... of which widened back to 16-byte keys (an oversized sub-bucket nothing streams) */
  uint32_t hpc_mixed_files;        /* `compress` buckets grouped by a dense-rank high digit + the plain eight bits of four bases (3^9 sub-buckets) */
  uint64_t stream_retries;         /* sub-buckets with more distinct suffixes than that kernel's table holds (counted by its retry launch) */
  double   probe_ratio;            /* distinct / instances of the probe file that chose between the plans (0: no probe ran) */
  /* what the rest of the count stage lasts (offsets of the sub-buckets in the packed result + the packing kernels of all files), on
   * the session stream behind the count kernels: stage_ms[MGC_STAGE_RLE] - pack_ms = the wall clock of the count kernels themselves */
  double   pack_ms;
  uint64_t hist_bytes;             /* ALGORITHMIC bytes of the histogram kernel (the bases read) ... */
  uint64_t partition_bytes;        /* ... and of the partition (the bases read + the k-mers written in the layout the files take: 5 / 8 / 12 / 16 B) */
} mgc_profile;
int mgc_set_profiling(mgc_session *s, int enable);
int mgc_get_profile(const mgc_session *s, mgc_profile *p);

/* ------------------------------------------------------------------------
 * Bench/test utility: deterministic synthetic reads generated in HBM
 * (byte-identical to oracle/oracle_count.c orc_synth_reads).
 * ---------------------------------------------------------------------- */
int mgc_dev_synth_reads(uint64_t seed, uint64_t genome_len, uint64_t first_read, uint64_t n_reads,
                        uint32_t read_len, uint32_t sub_rate_ppm, uint32_t n_rate_ppm,
                        uint8_t *d_out, void *stream);
/* ... with repeat families in the genome: repeat_ppm of its `repeat_unit`-base blocks show one of
 * `repeat_families` template sequences, family 0 by far the most frequent (byte-identical to orc_synth_reads_ex). */
int mgc_dev_synth_reads_ex(uint64_t seed, uint64_t genome_len, uint64_t first_read, uint64_t n_reads,
                           uint32_t read_len, uint32_t sub_rate_ppm, uint32_t n_rate_ppm,
                           uint32_t repeat_ppm, uint32_t repeat_unit, uint32_t repeat_families,
                           uint8_t *d_out, void *stream);

/* Library/ABI version: major<<16 | minor. */
uint32_t mgc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MERYL_GPU_COUNT_H */
