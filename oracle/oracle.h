/*
 * oracle.h -- CPU restatement of marbl/meryl's `count` path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke() may load it; the product
 * (meryl_amd/, include/) never links, imports or calls anything in here.
 *
 * PARITY STATUS
 *   semantic level (sorted canonical kmer -> uint32 count stream, database
 *   geometry wPrefix/wData):  PINNED against the reference's only in-tree
 *   known-answer vector (documentation/source/reference.rst:545-568, the
 *   GGAGCT 3-mer table) and the ordering rule of
 *   src/tests/test-operations.pl:114-118 (tests/golden/).
 *   byte level (.merylData/.merylIndex encoding):  PARITY UNPINNED -- the
 *   encoder lives in the absent submodule marbl/meryl-utility (pinned commit
 *   unknown; nearest marker `snapshot 1.4.2`, src/main.mk:2) and no reference
 *   database exists in /root/reference.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#ifndef MERYL_ORACLE_H
#define MERYL_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned __int128 orc_kmdata;   /* src/tests/merylCountArrayTest.C:27-31 (kmdata is 128-bit) */
typedef uint32_t          orc_kmvalu;   /* documentation/source/reference.rst:47-51 */

enum { ORC_CANONICAL = 0, ORC_FORWARD = 1, ORC_REVERSE = 2 };  /* opCount / opCountForward / opCountReverse,
                                                                  src/meryl/merylOp-countThreads.C:241-258 */

/* 2-bit code of a base, A=0 C=1 T=2 G=3 (reference.rst:525,540-568); -1 for anything that is not ACGTacgt. */
int      orc_base_code(char c);

/* Enumerate every k-mer instance of a base stream in input order, as
 * kmerIterator does at merylOp-countThreads.C:240-258: a non-ACGT byte (incl.
 * the '.' breakers of :196,:214-215) resets the rolling k-mer.  Writes at most
 * cap values (hi/lo 64-bit halves), returns the number of instances. */
uint64_t orc_enumerate_kmers(const char *bases, uint64_t n, uint32_t k, int mode,
                             uint64_t *out_hi, uint64_t *out_lo, uint64_t cap);

/* Brute-force count in the style of src/meryl-simple/meryl-simple.C:131-187:
 * all instances -> one array -> sort ascending -> run-length.  Counts wrap
 * mod 2^32 like `_counts[_nKmers]++` on a uint32 (merylCountArray.C:357).
 * Output arrays are malloc'd; free with orc_free(). */
int      orc_count_brute(const char *bases, uint64_t n, uint32_t k, int mode,
                         uint64_t **keys_hi, uint64_t **keys_lo, uint32_t **counts,
                         uint64_t *n_distinct, uint64_t *n_instances);
int      orc_count_brute_suffix(const char *bases, uint64_t n, uint32_t k, int mode, const char *count_suffix,
                                uint64_t **keys_hi, uint64_t **keys_lo, uint32_t **counts,
                                uint64_t *n_distinct, uint64_t *n_instances);
void     orc_free(void *p);

/* Print form of `meryl print` (merylOp-nextMer.C:665-677): "%s\t%u\n". kmer -> ACTG string. */
void     orc_kmer_to_string(uint64_t hi, uint64_t lo, uint32_t k, char *out /* k+1 bytes */);

/* Homopolymer compression hook of merylInput::loadBases (merylInput.C:261-268).
 * `last_byte` is the _lastByte carried from the previous chunk of the same
 * sequence (0 at a sequence start).  Returns compressed length; out may alias in. */
uint64_t orc_homopoly_compress(const char *in, uint64_t n, char *out, char last_byte);

/* configureCounting restated exactly (merylOp-count.C:118-403). */
typedef struct {
  /* inputs */
  uint32_t k;
  uint64_t n_kmers_estimate;        /* _expNumKmers: n= or the file-size guess (:449) */
  uint64_t memory_allowed;          /* bytes */
  uint32_t count_suffix_length;     /* 0 unless count-suffix= */
  uint32_t page_size;               /* getPageSize(); 4096 here */
  uint32_t sizeof_count_array;      /* sizeof(merylCountArray) == 3232 on LP64 with ADD_INSTRUMENT (merylCountArray.H:44,71-74) */
  /* outputs */
  int      use_simple;
  uint32_t w_prefix;
  uint64_t n_prefix;
  uint32_t w_data;
  uint32_t n_batches;
  uint64_t memory_used;
  uint64_t memory_simple;
  uint64_t memory_complex;
} orc_config;
int      orc_configure_counting(orc_config *c);

/* Reference-algorithm restatement (the timed "port" baseline): 2 MiB chunks
 * with k-1 carry (merylOp-countThreads.C:138-231), prefix/suffix split and
 * per-bucket append (:235-280), per-bucket std::sort + RLE
 * (merylCountArray.C:323-365), 64-file parallel dump (:452-459).
 * Implemented in oracle_port.cpp.  Block callback mirrors
 * merylBlockWriter::addBlock(prefix, nKmers, suffixes, counts). */
typedef void (*orc_block_cb)(void *ctx, uint64_t prefix, uint64_t n_kmers,
                             const orc_kmdata *suffixes, const orc_kmvalu *counts);
int      orc_count_threaded(const char *bases, uint64_t n, uint32_t k, int mode,
                            uint32_t w_prefix, int threads,
                            orc_block_cb cb, void *ctx,
                            uint64_t *n_distinct, uint64_t *n_instances);

/* Convenience for timing/tests: runs orc_count_threaded and folds the blocks
 * into full keys (prefix<<wData | suffix), returning malloc'd arrays in
 * ascending key order. */
int      orc_count_threaded_collect(const char *bases, uint64_t n, uint32_t k, int mode,
                                    uint32_t w_prefix, int threads,
                                    uint64_t **keys_hi, uint64_t **keys_lo, uint32_t **counts,
                                    uint64_t *n_distinct, uint64_t *n_instances);

/* Per-file digests of the port's result (64 x 4 uint64, all sums mod 2^64): n_distinct, sum of counts,
 * sum (x*C1)*count, sum ((x^C2)*(x|1))*count, with x = lo ^ hi*C3 -- the full-size
 * parity test takes the same sums of the GPU result without ever materialising the port's 14 GB of output. */
int      orc_count_threaded_digest(const char *bases, uint64_t n, uint32_t k, int mode, uint32_t w_prefix, int threads,
                                   uint64_t *out, uint64_t *n_distinct, uint64_t *n_instances);

/* Deterministic synthetic reads (splitmix64 counter PRNG): a
 * random genome of `genome_len` bases from `seed`, `n_reads` reads of
 * `read_len` sampled uniformly from both strands with `sub_rate_ppm`
 * substitutions and `n_rate_ppm` N's, written as a base stream with one '.'
 * after every read.  Returns bytes written (n_reads*(read_len+1)).  The GPU
 * generator in meryl_amd/csrc reproduces these bytes exactly. */
uint64_t orc_synth_reads(uint64_t seed, uint64_t genome_len, uint64_t first_read, uint64_t n_reads,
                         uint32_t read_len, uint32_t sub_rate_ppm, uint32_t n_rate_ppm, char *out);
/* the same with repeat families in the genome (repeat_ppm of the `repeat_unit`-base blocks show one of
 * `repeat_families` template sequences, cubic skew); byte-identical to mgc_dev_synth_reads_ex */
uint64_t orc_synth_reads_ex(uint64_t seed, uint64_t genome_len, uint64_t first_read, uint64_t n_reads,
                            uint32_t read_len, uint32_t sub_rate_ppm, uint32_t n_rate_ppm,
                            uint32_t repeat_ppm, uint32_t repeat_unit, uint32_t repeat_families, char *out);

#ifdef __cplusplus
}
#endif
#endif
