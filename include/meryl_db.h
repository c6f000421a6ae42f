/*
 * meryl_db.h -- C ABI of the meryl database writer/reader (the downstream
 * side of the count path).
 *
 * Replaces, for the count path only, the reference's
 *   merylFileWriter(name) / ::initialize(wPrefix) / ::getBlockWriter()
 *   merylBlockWriter::addBlock(prefix, nKmers, suffixes, counts) / finish()
 * whose call sites are src/meryl/merylOp.C:262,
 * src/meryl/merylOp-countThreads.C:48,404,453-464 and
 * src/meryl/merylCountArray.C:472-475.  Those classes live in the absent
 * submodule marbl/meryl-utility (utility/src/kmers-v1/kmers-writer*.C), so
 * the BYTE LAYOUT written here is a restatement of that library's published
 * v1 format from its documented shape (documentation/source/usage.rst:13-45,
 * reference.rst:73-77) and is marked PARITY UNPINNED: no reference-written
 * database exists in the tree to diff against.  Every layout assumption is
 * listed in DESIGN.md ("database encoding").  What IS pinned: 64 data files +
 * 64 index files + one master `merylIndex`; prefixSize / suffixSize /
 * numFilesBits=6 / numBlocksBits=prefixSize-6; block header fields
 * prefix nKmers kCode uBits bBits k1 cCode c1 c2 with uBits+bBits=suffixSize.
 */
#ifndef MERYL_DB_H
#define MERYL_DB_H

#include <stddef.h>
#include <stdint.h>

#include "meryl_gpu_count.h"   /* MGC_MERGE_* */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mdb_writer mdb_writer;
typedef struct mdb_reader mdb_reader;
struct mgc_session;

/* Creates directory `path` (merylFileWriter ctor, merylOp.C:262) and fixes the
 * geometry (merylFileWriter::initialize(wPrefix), countThreads.C:404). */
mdb_writer *mdb_writer_open(const char *path, uint32_t k, uint32_t w_prefix);

/* The same with two extensions:
 *  label_size  bits of label stored per k-mer (0..64; 0 = none: the bytes written are then exactly the unlabelled ones).
 *              meryl2 sets it with the global `-l <bits>` (src/meryl2/merylGlobals.C:75-77) and its count hands ONE constant
 *              label to the writer at dump time: addCountedBlock(prefix, nKmers, suffix, counts, labels=nullptr, label)
 *              (src/meryl2/merylCountArray.C:469-471, src/meryl2/merylOp-countThreads.C:363-367,464-468).
 *  part / n_parts  a database written by several writers, one per rank of a sharded (multi-GPU) count: writer `part`
 *              receives the blocks of a contiguous prefix range (ranges ascend with `part`; a range may begin or end in
 *              the middle of a file).  It writes `0xBBBBBB.merylData.part<part>` files and a side file; once every part
 *              is closed, ONE caller runs mdb_merge_parts, which stitches the data files (rename when a file has one
 *              contributor, append otherwise), and writes the 64 per-file indexes and the master index.  The resulting
 *              directory is byte-identical to what a single writer fed the same blocks produces.  n_parts = 1, part = 0
 *              is the plain writer. */
mdb_writer *mdb_writer_open_ex(const char *path, uint32_t k, uint32_t w_prefix, uint32_t label_size,
                               uint32_t part, uint32_t n_parts);
int mdb_merge_parts(const char *path, uint32_t n_parts);

/* merylBlockWriter::addBlock.  May be called concurrently from several threads
 * as long as each FILE (prefix >> (w_prefix-6)) is fed by one thread with
 * ascending prefixes -- the reference's convention (countThreads.C:452-459).
 * suffix_hi may be NULL when 2k - w_prefix <= 64.  Arrays are not retained. */
int mdb_writer_add_block(mdb_writer *w, uint64_t prefix, uint64_t n_kmers,
                         const uint64_t *suffix_lo, const uint64_t *suffix_hi, const uint32_t *counts);

/* merylBlockWriter::addCountedBlock of meryl2 (src/meryl2/merylCountArray.C:469-471): per-k-mer labels, or -- labels ==
 * NULL -- the one constant `label` for every k-mer of the block.  Only the low label_size bits are stored. */
int mdb_writer_add_block_labelled(mdb_writer *w, uint64_t prefix, uint64_t n_kmers,
                                  const uint64_t *suffix_lo, const uint64_t *suffix_hi, const uint32_t *counts,
                                  const uint64_t *labels, uint64_t label);

/* Blocks that are already encoded (the device encoder behind mgc_write_database): `nbytes` bytes of data file ff holding
 * the blocks `entries` list (prefixes ascending and after everything file ff has received so far; entries[i].position
 * is the block's offset inside `bytes`); bytes before the first entry -- or all of them when n_entries is 0 -- continue
 * the block the previous call ended in (a block larger than the caller's copy buffer arrives in pieces).  Same threading rule as mdb_writer_add_block (one thread per file).  The value
 * histogram of such blocks is handed over separately, once or in pieces: (value, occurrences) pairs. */
typedef struct mdb_index_entry { uint64_t prefix, position, n_kmers; } mdb_index_entry;
int mdb_writer_add_encoded(mdb_writer *w, uint32_t ff, const void *bytes, uint64_t nbytes,
                           const mdb_index_entry *entries, uint64_t n_entries);
/* The same in two steps, so that many threads can write ONE file at once: reserve (same ordering rule as above, no data)
 * registers the index entries and returns the file offset the nbytes belong at; mdb_writer_write_at then puts them
 * there with pwrite -- thread-safe, in any order, from any thread. */
int mdb_writer_reserve_encoded(mdb_writer *w, uint32_t ff, uint64_t nbytes, const mdb_index_entry *entries,
                               uint64_t n_entries, uint64_t *file_offset);
int mdb_writer_write_at(mdb_writer *w, uint32_t ff, uint64_t file_offset, const void *bytes, uint64_t nbytes);
int mdb_writer_add_histogram(mdb_writer *w, const uint64_t *values, const uint64_t *occurrences, uint64_t n_pairs);

/* merylBlockWriter::finish() + ~merylFileWriter(): per-file indexes, master
 * index with the value histogram.  Frees the writer.  0 on success. */
int mdb_writer_close(mdb_writer *w);
/* Frees a writer without writing the indexes (error paths: nothing that looks like a finished database is left behind). */
void mdb_writer_discard(mdb_writer *w);

const char *mdb_last_error(void);

typedef struct mdb_info {
  uint32_t k;
  uint32_t prefix_size, suffix_size, num_files_bits, num_blocks_bits, flags;
  uint64_t num_unique, num_distinct, num_total;   /* histogram header */
  uint64_t hist_len;                              /* number of (value, occurrences) pairs */
  uint32_t label_size;                            /* bits of label per k-mer (flags bits 8..15) */
  uint32_t reserved;
} mdb_info;

mdb_reader *mdb_reader_open(const char *path);
int  mdb_reader_info(const mdb_reader *r, mdb_info *info);
/* histogram pairs, each array hist_len long */
int  mdb_reader_histogram(const mdb_reader *r, uint64_t *values, uint64_t *occurrences);
/* All k-mers of file ff (0..63) in stored order as full k-mers (prefix<<suffixSize|suffix).
 * Arrays are malloc'd; release with mdb_free.  keys_hi is NULL-filled (zeros) for k <= 32. */
int  mdb_reader_read_file(mdb_reader *r, uint32_t ff, uint64_t **keys_lo, uint64_t **keys_hi,
                          uint32_t **counts, uint64_t *n_kmers);
/* ... and their labels (zeros when the database stores none); labels may be NULL. */
int  mdb_reader_read_file_ex(mdb_reader *r, uint32_t ff, uint64_t **keys_lo, uint64_t **keys_hi,
                             uint32_t **counts, uint64_t **labels, uint64_t *n_kmers);
/* The per-file index (1 << num_blocks_bits entries) and the header of the block at a data-file position: what
 * `meryl dumpFile` prints (src/meryl/meryl.C:41-45; shape documentation/source/usage.rst:24-45). */
typedef struct mdb_block_header {
  uint64_t prefix, n_kmers;
  uint32_t k_code, unary_bits, binary_bits, c_code;
  uint64_t k1, c1, c2;
} mdb_block_header;
int  mdb_reader_file_index(mdb_reader *r, uint32_t ff, mdb_index_entry *entries);
int  mdb_reader_block_header(mdb_reader *r, uint32_t ff, uint64_t position, mdb_block_header *h);
/* One block decoded the way dumpFile lists it: for k-mer i the unary prefix delta, the accumulated top part, the two
 * halves of the binary remainder and the value; arrays of n_kmers entries, malloc'd (mdb_free). */
int  mdb_reader_read_block_raw(mdb_reader *r, uint32_t ff, uint64_t position, mdb_block_header *h,
                               uint64_t **prefix_delta, uint64_t **top, uint64_t **rem_hi, uint64_t **rem_lo,
                               uint32_t **values);
/* For a decoder that does not run on the host (meryl_amd/csrc/mgc_decode.hip): the raw bytes of data file ff (malloc'd, *size
 * bytes followed by 16 zero bytes) and, for every block of its index that holds k-mers, where its stuffedBits object lies
 * and how long it is -- every object's framing validated against the file size, out_offset = k-mers of the blocks before it.
 * MGC_EUNSUPPORTED when an object is framed in a way only the host decoder follows (mdb_reader_read_file_ex reads it). */
typedef struct mdb_raw_block { uint64_t object_offset, n_bits, n_sub_blocks, n_kmers, prefix, out_offset; } mdb_raw_block;
int  mdb_reader_raw_file(mdb_reader *r, uint32_t ff, unsigned char **bytes, uint64_t *size, mdb_raw_block **blocks,
                         uint64_t *n_blocks, uint64_t *n_kmers);
void mdb_reader_close(mdb_reader *r);
void mdb_free(void *p);

/* ------------------------------------------------------------------------
 * Device-resident (k-mer, count) streams -> database files, encoded ON THE DEVICE
 * (meryl_amd/csrc/mgc_encode.hip) -- the MI355X replacement of the 64 host threads that run
 * countKmers + dumpCountedKmers -> merylBlockWriter::addBlock per file
 * (src/meryl/merylOp-countThreads.C:452-459, src/meryl/merylCountArray.C:472-475): the host only copies finished
 * file bytes out of HBM through pinned buffers and writes them.
 *
 * A stream is one writer (or one part of a sharded database, see mdb_writer_open_ex) fed with ascending,
 * non-overlapping prefix ranges.  mgc_db_stream_write queues one range: d_keys/d_counts are DEVICE pointers to the
 * n distinct k-mers (uint64, or {lo,hi} pairs for k > 32; ascending) whose prefixes lie in [prefix_begin, prefix_end)
 * and their counts; every prefix of the range gets its block, empty ones included.  The call returns once the range
 * is planned; encoding, the device-to-host copies and the file writes run on the stream's own threads, so the
 * caller can go on counting (a sharded count writes the buckets of wave i while wave i+1 is exchanged and counted).
 * The buffers must stay valid and unchanged until mgc_db_stream_sync or mgc_db_stream_close returns.
 * Errors are sticky: the first one is returned by every later call; mgc_db_stream_error gives the text. */
typedef struct mgc_db_stream mgc_db_stream;
typedef struct mgc_db_write_profile {
  double   plan_ms;        /* block offsets + sizes + histogram kernels, planning */
  double   encode_ms;      /* device encode kernels (HIP events) */
  double   copy_write_s;   /* wall clock of device-to-host copies + file writes (overlapped with each other) */
  double   total_s;        /* open -> close */
  uint64_t data_bytes;     /* bytes of .merylData written */
  uint64_t n_kmers, n_blocks;
} mgc_db_write_profile;
mgc_db_stream *mgc_db_stream_open(const char *path, uint32_t k, uint32_t w_prefix, uint32_t label_size, uint64_t label,
                                  uint32_t part, uint32_t n_parts, int host_threads, int device);
int  mgc_db_stream_write(mgc_db_stream *d, const void *d_keys, const uint32_t *d_counts, uint64_t n,
                         uint64_t prefix_begin, uint64_t prefix_end);
int  mgc_db_stream_sync(mgc_db_stream *d);
/* Waits for everything queued, closes the writer (a part: its side file; the plain writer: indexes + master index)
 * and frees the stream.  prof may be NULL. */
int  mgc_db_stream_close(mgc_db_stream *d, mgc_db_write_profile *prof);
const char *mgc_db_stream_error(const mgc_db_stream *d);      /* d may be NULL: last open error of this thread */

/* mgc_db_stream_queued: ranges queued so far (the number of the last one).  mgc_db_stream_wait_buffers: ranges 1 .. upto
 * have been encoded and copied out of their device buffers, which may be reused; the file writes of the copied pieces may
 * still be running. */
uint64_t mgc_db_stream_queued(mgc_db_stream *d);
uint64_t mgc_db_stream_done(mgc_db_stream *d);      /* ranges 1 .. this have left their device buffers */
int  mgc_db_stream_wait_buffers(mgc_db_stream *d, uint64_t upto);

/* ------------------------------------------------------------------------
 * Sorted runs of partial results -- the spill of an out-of-core count.
 *
 * What it replaces: merylOperation::countThreads writes every bucket of a full memory as an ITERATION of the output
 * files (writeBatch, src/meryl/merylOp-countThreads.C:285-380, finishBatch :362) and merylBlockWriter::finish() merges
 * the iterations of every file at the end (:461-464; the merge itself is in the absent meryl-utility): the size of a count
 * is bounded by disk, not by memory.  Here a batch's (k-mer, count) result is a RUN: ascending distinct k-mers with their
 * counts.  A run stays in HBM while the store's device budget lasts and is parked in pinned host DRAM otherwise (one
 * asynchronous device-to-host copy behind the next batch's count).  At the end the runs are merged ONCE: the k-mer space is
 * walked in slices (top min(w_prefix, 14) bits), consecutive slices are gathered into chunks that fit the device, every
 * run's piece of the chunk is uploaded (host runs) or used in place (device runs), the pieces are merged pairwise on the
 * device (counts of equal k-mers add, uint32 wrap like the reference's value arithmetic) and the merged chunk goes to
 * the database stream -- uploads of chunk i+1 overlap the encode + write of chunk i.  Nothing is rewritten per batch.
 * The sharded / node counts park one run per (batch, wave) on the owner and merge the owner's prefix range the same way.
 *
 * device_budget_bytes: bytes of runs that may stay in HBM (0: every run goes to the host; ~0: no limit).
 * chunk_bytes: device memory the final merge may use for its buffers (0: a quarter of the free HBM, at most 24 GiB).
 * Environment (tests): MGC_OOC_BUDGET, MGC_OOC_CHUNK override both.  Failure text: mgc_runs_error(). */
typedef struct mgc_runs mgc_runs;
typedef struct mgc_runs_profile {
  uint32_t n_runs, n_host_runs;
  uint64_t n_entries;          /* (k-mer, count) pairs over all runs */
  uint64_t device_bytes, host_bytes;
  uint64_t n_merged;           /* distinct k-mers delivered by mgc_runs_write so far */
  uint32_t n_chunks;
  double   spill_s;            /* wall clock of the device-to-host copies (incl. pinning the host memory) */
  double   upload_s, merge_ms, deliver_s;
  uint64_t peak_hbm_bytes;     /* device memory in use (all of the process) at its highest sampled point */
} mgc_runs_profile;
mgc_runs *mgc_runs_open(uint32_t k, uint32_t w_prefix, int device, uint64_t device_budget_bytes, uint64_t chunk_bytes);
/* copies the n ascending distinct k-mers at d_keys (uint64, or {lo,hi} for k > 32) and their counts into a new run;
 * `stream`: where they were produced.  Returns when the source may be overwritten. */
int  mgc_runs_add(mgc_runs *r, const void *d_keys, const uint32_t *d_counts, uint64_t n, void *stream);
/* merges all runs over prefixes [prefix_begin, prefix_end) into d (ranges ascending, see mgc_db_stream_write) */
int  mgc_runs_write(mgc_runs *r, mgc_db_stream *d, uint64_t prefix_begin, uint64_t prefix_end);
int  mgc_runs_get_profile(const mgc_runs *r, mgc_runs_profile *p);
const char *mgc_runs_error(const mgc_runs *r);                 /* r may be NULL: last open error of this thread */
void mgc_runs_close(mgc_runs *r);
/* the run store of a session that counted in batches (zeros when it counted in one pass) */
int  mgc_get_runs_profile(const struct mgc_session *s, mgc_runs_profile *p);

/* Count result of a session -> database directory, through a device stream (results that live in HBM, or -- out of
 * core -- in the session's runs; host_threads file-writer threads).  prof may be NULL. */
int mgc_write_database(struct mgc_session *s, const char *path, int host_threads);
int mgc_write_database_profiled(struct mgc_session *s, const char *path, int host_threads, mgc_db_write_profile *prof);

/* `union-sum` and friends over whole databases (op = MGC_MERGE_* of include/meryl_gpu_count.h): the reference streams the
 * 64 file slices of its inputs through merylOperation::nextMer (src/meryl/merylOp-nextMer.C:418-683, slices
 * src/meryl/meryl.C:250-263); here every slice is decoded by host threads, merged two inputs at a time on the device,
 * encoded on the device and written to `output` (geometry of the first input).  All inputs must hold the same k.
 * Text of a failure: mgc_db_stream_error(NULL). */
int mgc_db_merge(const char *const *inputs, uint32_t n_inputs, int op, const char *output, int device, int host_threads);
/* The single-input operations over a whole database (value_op = MGC_VALUE_*: less-than ... not-equal-to against `constant`,
 * increase ... modulo by it; src/meryl/merylOp-nextMer.C:490-557): k-mers whose new value is 0 are not written. */
int mgc_db_filter(const char *input, int value_op, uint64_t constant, const char *output, int device, int host_threads);

/* ONE count spread over the GPUs of a node, from one process (meryl_amd/csrc/mgc_node.cpp): rank r's reads are the
 * n_bases[r] bytes at d_bases[r] on device devices[r] (the base stream mgc_push_bases takes; with cfg->homopoly_compress
 * every rank's stream must hold whole sequences).  Every rank extracts the k-mers of its reads, the k-mers travel over
 * xGMI (peer copies, in waves that overlap with counting) to the rank that owns their range of the k-mer space, the
 * owners count and write disjoint prefix ranges, and the parts are stitched: the directory at db_path is
 * byte-identical to what one device gives for the concatenated input.  The reference has no counterpart -- its
 * node-scale recipe is "count pieces, then union-sum" (src/meryl/merylOp-count.C:251-268); this is what replaces it.
 * devices == NULL: rank r uses device r modulo the visible devices; several ranks MAY share a device (that is how a
 * one-GPU box tests the whole plan).  cfg must have been through mgc_configure_counting for the TOTAL input;
 * count-suffix is not supported here.  Failure text: mgc_last_error(NULL).
 * (The one-process-per-GPU form over RCCL is meryl_amd/count.py:count_sharded.) */
typedef struct mgc_node_profile {
  uint32_t n_ranks, bucket_bits;
  uint64_t n_bases, n_instances, n_distinct, data_bytes;
  double   partition_s;        /* slowest rank: histogram + partition of its reads */
  double   exchange_count_s;   /* slowest rank: pulls + grouping + finish + handing blocks to the writer */
  double   close_s;            /* slowest rank: waiting for its part's files */
  double   merge_parts_s, total_s;
  uint32_t n_batches, n_host_runs; /* batches every rank's reads were counted in; parked waves that went to host DRAM */
  uint64_t host_run_bytes;
  double   merge_runs_s;           /* slowest rank: merging its parked waves into its part (batches > 1) */
  uint64_t peak_hbm_bytes;         /* highest sampled device memory in use on any rank's device (batches > 1) */
} mgc_node_profile;
int mgc_count_node(const mgc_count_config *cfg, uint32_t n_ranks, const int *devices,
                   const uint8_t *const *d_bases, const uint64_t *n_bases,
                   const char *db_path, int host_threads, mgc_node_profile *prof);
/* The same with BATCHES (the node form of writeBatch's spill, merylOp-countThreads.C:323-379): no rank ever holds the
 * k-mers of more than batch_bases of its bases at once.  The routing plan comes from one histogram of ALL reads; then,
 * batch by batch, every rank partitions the next slice of its base stream (slices overlap by k-1 bases), the owners pull
 * and count their waves and park every counted wave as a run (include: mgc_runs_*; HBM first, pinned host DRAM beyond
 * the budget); after the last batch every owner merges its runs once into its part of the database.  batch_bases = 0:
 * derived from the free HBM (one batch when everything fits -- then this IS mgc_count_node, waves streamed to the writer
 * as they are counted). */
int mgc_count_node_batched(const mgc_count_config *cfg, uint32_t n_ranks, const int *devices,
                           const uint8_t *const *d_bases, const uint64_t *n_bases, uint64_t batch_bases,
                           const char *db_path, int host_threads, mgc_node_profile *prof);

/* The routing plan of mgc_count_node on its own (host arithmetic only, no device needed): *bucket_bits = the top bits of the k-mer
 * that route it for n_ranks ranks whose largest input is max_rank_bases (6 + ceil(log2 n_ranks), more for very large inputs, at
 * most 10 and at most w_prefix); with bucket_totals[2^*bucket_bits] (k-mers per bucket over all ranks) also cuts[n_ranks + 1]: rank r
 * owns buckets [cuts[r], cuts[r+1]) -- contiguous, every rank at least one, balanced by k-mers.  MGC_EINVAL when the ranks
 * outnumber the routable ranges. */
int mgc_node_plan(uint32_t n_ranks, uint32_t k, uint64_t max_rank_bases, uint32_t w_prefix, uint32_t *bucket_bits,
                  const uint64_t *bucket_totals, uint32_t *cuts);

/* The same for input that went through ONE session (pushed bases, parsed text, files -- the CLI's `gpus=N`): the staged
 * stream (mgc_staged_bases) is cut into n_ranks slices overlapping by k-1 bases -- no k-mer lost or doubled wherever the
 * cut falls; `compress` is applied before cutting -- the slices move to their devices and mgc_count_node runs.  The
 * session itself counts nothing and can only be closed afterwards.  Failure text: mgc_last_error(s). */
int mgc_count_node_staged(struct mgc_session *s, uint32_t n_ranks, const int *devices, const char *db_path,
                          int host_threads, mgc_node_profile *prof);

#ifdef __cplusplus
}
#endif
#endif /* MERYL_DB_H */
