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

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mdb_writer mdb_writer;
typedef struct mdb_reader mdb_reader;
struct mgc_session;

/* Creates directory `path` (merylFileWriter ctor, merylOp.C:262) and fixes the
 * geometry (merylFileWriter::initialize(wPrefix), countThreads.C:404). */
mdb_writer *mdb_writer_open(const char *path, uint32_t k, uint32_t w_prefix);

/* merylBlockWriter::addBlock.  May be called concurrently from several threads
 * as long as each FILE (prefix >> (w_prefix-6)) is fed by one thread with
 * ascending prefixes -- the reference's convention (countThreads.C:452-459).
 * suffix_hi may be NULL when 2k - w_prefix <= 64.  Arrays are not retained. */
int mdb_writer_add_block(mdb_writer *w, uint64_t prefix, uint64_t n_kmers,
                         const uint64_t *suffix_lo, const uint64_t *suffix_hi, const uint32_t *counts);

/* merylBlockWriter::finish() + ~merylFileWriter(): per-file indexes, master
 * index with the value histogram.  Frees the writer.  0 on success. */
int mdb_writer_close(mdb_writer *w);

const char *mdb_last_error(void);

typedef struct mdb_info {
  uint32_t k;
  uint32_t prefix_size, suffix_size, num_files_bits, num_blocks_bits, flags;
  uint64_t num_unique, num_distinct, num_total;   /* histogram header */
  uint64_t hist_len;                              /* number of (value, occurrences) pairs */
} mdb_info;

mdb_reader *mdb_reader_open(const char *path);
int  mdb_reader_info(const mdb_reader *r, mdb_info *info);
/* histogram pairs, each array hist_len long */
int  mdb_reader_histogram(const mdb_reader *r, uint64_t *values, uint64_t *occurrences);
/* All k-mers of file ff (0..63) in stored order as full k-mers (prefix<<suffixSize|suffix).
 * Arrays are malloc'd; release with mdb_free.  keys_hi is NULL-filled (zeros) for k <= 32. */
int  mdb_reader_read_file(mdb_reader *r, uint32_t ff, uint64_t **keys_lo, uint64_t **keys_hi,
                          uint32_t **counts, uint64_t *n_kmers);
void mdb_reader_close(mdb_reader *r);
void mdb_free(void *p);

/* Count result -> database directory: mgc_finish() feeding mdb_writer_add_block
 * from `host_threads` threads, then mdb_writer_close. */
int mgc_write_database(struct mgc_session *s, const char *path, int host_threads);

#ifdef __cplusplus
}
#endif
#endif /* MERYL_DB_H */
