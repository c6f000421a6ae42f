/*
 * meryl_lookup.h -- C ABI of the exact k-mer lookup table: how meryl-lookup and Merqury
 * consume a counted database.
 *
 * Replaces, for loading and querying, the reference's
 *   merylExactLookup::estimateMemoryUsage / load(db, maxMemory, ..., minValue, maxValue)
 *   merylExactLookup::value(kmer) / exists(kmer) / nKmers()
 * [class in the absent meryl-utility; call sites src/meryl-lookup/meryl-lookup.C:36-100 (load),
 *  src/meryl-lookup/existence.C:63-82 (value(fmer) > 0 || value(rmer) > 0 per k-mer of a sequence, nKmers())].
 *
 * MI355X form: the table IS the database's own order -- the distinct k-mers ascending with their values, resident in
 * HBM, plus a direct index over their top bits (first k-mer of every 2^P-th part of the key space).  A lookup is
 * one index read and a binary search over the handful of k-mers that share the top bits: exact, no hashing, no extra
 * copy of the keys, 12 (20) B per k-mer + 8 B per index entry.  Queries come in batches that are already on the device:
 * explicit k-mers (mgc_lookup_values) or a base stream whose every window is looked up (mgc_lookup_stream,
 * mgc_lookup_existence -- the reference's -existence report).
 */
#ifndef MERYL_LOOKUP_H
#define MERYL_LOOKUP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mgc_lookup mgc_lookup;

typedef struct mgc_lookup_info {
  uint32_t k, key_words, index_bits, reserved;
  uint64_t n_kmers;            /* merylExactLookup::nKmers(): k-mers kept after the value filter */
  uint64_t n_kmers_in_db;      /* before the filter */
  uint64_t device_bytes;       /* keys + values + index */
} mgc_lookup_info;

/* merylExactLookup::load (src/meryl-lookup/meryl-lookup.C:91): the k-mers of the database at `db_path` whose value v has
 * min_value <= v <= max_value (-min / -max of meryl-lookup; 0 and UINT64_MAX keep everything), decoded by `host_threads`
 * threads, uploaded and indexed on `device` (< 0: current).  NULL on failure; text via mgc_lookup_error(). */
mgc_lookup *mgc_lookup_load(const char *db_path, uint64_t min_value, uint64_t max_value, int device, int host_threads);

/* merylExactLookup::estimateMemoryUsage (src/meryl-lookup/meryl-lookup.C:62-73): what mgc_lookup_load of this database with this
 * value filter will hold on the device, WITHOUT loading it and without a device -- the number of k-mers kept comes from the
 * database's own value histogram (exact), the bytes from the table's layout (8 or 16 B per k-mer + 4 B per value + the top-bits
 * index).  0 on success; text of a failure: mgc_lookup_error(). */
int mgc_lookup_estimate(const char *db_path, uint64_t min_value, uint64_t max_value, mgc_lookup_info *info);

/* The same from a (k-mer, value) stream that is already in HBM (distinct, ascending: a count session's result, a merge):
 * no file round trip.  The arrays are copied. */
mgc_lookup *mgc_lookup_from_device(const void *d_keys, const uint32_t *d_values, uint64_t n, uint32_t k,
                                   uint64_t min_value, uint64_t max_value, int device);

void        mgc_lookup_free(mgc_lookup *t);
int         mgc_lookup_get_info(const mgc_lookup *t, mgc_lookup_info *info);
const char *mgc_lookup_error(void);

/* merylExactLookup::value() for n k-mers on the device (uint64, or {lo, hi} pairs for k > 32; looked up as given -- no
 * canonicalisation): d_values_out[i] = value, 0 if absent. */
int mgc_lookup_values(const mgc_lookup *t, const void *d_kmers, uint64_t n, uint32_t *d_values_out, void *stream);

/* Every k-mer window of a base stream (ASCII, '.' or any non-ACGT byte breaks a k-mer -- the stream mgc_push_bases
 * takes): d_values_out[i] = value of the k-mer STARTING at base i, looked up as the reference's lookups do --
 * value(fmer), or value(rmer) when the forward k-mer is absent (existence.C:73-76) -- and 0 where the k-mer is absent,
 * broken, or would run past the end.  d_values_out holds n_bases entries. */
int mgc_lookup_stream(const mgc_lookup *t, const uint8_t *d_bases, uint64_t n_bases, uint32_t *d_values_out, void *stream);

/* meryl-lookup -existence (src/meryl-lookup/existence.C:48-82): for each of the n_seq sequences -- sequence s is
 * bases[d_seq_start[s], d_seq_start[s+1]) of the stream -- the number of k-mers it holds (d_total[s]) and how many of
 * them are in the table (d_found[s]).  d_seq_start has n_seq + 1 device entries. */
int mgc_lookup_existence(const mgc_lookup *t, const uint8_t *d_bases, uint64_t n_bases, const uint64_t *d_seq_start,
                         uint64_t n_seq, uint64_t *d_total, uint64_t *d_found, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MERYL_LOOKUP_H */
