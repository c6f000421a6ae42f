/*
 * oracle_count.c -- semantic oracle for meryl `count` (plain C).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Restates, function by function,
 * what the reference computes on the count path; every function cites the
 * reference file:line it follows.  Where the arithmetic lives in the absent
 * submodule marbl/meryl-utility (kmerTiny/kmerIterator, homopolyCompress),
 * the behaviour is anchored on the reference's call sites and docs and the
 * remaining assumptions are marked [NOT IN TREE].
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------
 * Base encoding.  A=0 C=1 T=2 G=3: documentation/source/reference.rst:525,
 * 540-568; src/tests/test-operations.pl:114-118 (tr/GT/TG/ makes
 * lexicographic order match database order).  [NOT IN TREE] the encoder
 * itself (kmerTiny::addR, utility/src/kmers-v1/kmers.H) -- the published
 * trick is (ascii >> 1) & 3, which maps upper and lower case identically;
 * everything that is not ACGTacgt breaks the k-mer (call site
 * merylOp-countThreads.C:196,214-215 relies on '.' doing so).
 * ---------------------------------------------------------------------- */
int orc_base_code(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'T': case 't': return 2;
    case 'G': case 'g': return 3;
    default:            return -1;
  }
}

static orc_kmdata kmer_mask(uint32_t k) {
  /* merylOp-count.C:282-286 builds masks the same way: all ones shifted down */
  orc_kmdata m = 0;
  m = ~m;
  m >>= (128 - 2 * k);
  return m;
}

/* ------------------------------------------------------------------------
 * kmerIterator walk, merylOp-countThreads.C:240-258 (and the identical loop
 * of src/meryl-simple/meryl-simple.C:131-138):
 *   fmer rolls left  (new base enters at the low end),
 *   rmer rolls right (complement of the new base enters at the high end);
 *   complement under A0 C1 T2 G3 is code ^ 2;
 *   canonical = the numerically smaller (`fmer() < rmer()`, :245-246);
 *   count-forward keeps fmer, count-reverse keeps rmer (:241,248-258).
 * ---------------------------------------------------------------------- */
/* count-suffix= (merylOp-countSimple.C:50-58, 88-93): with a non-zero mask, the k-mer that would be counted is skipped
 * unless (kmer & suffix_mask) == suffix_test. */
static uint64_t enumerate_filtered(const char *bases, uint64_t n, uint32_t k, int mode,
                                   orc_kmdata suffix_mask, orc_kmdata suffix_test,
                                   uint64_t *out_hi, uint64_t *out_lo, uint64_t cap) {
  if (k == 0 || k > 64) return 0;

  const orc_kmdata mask = kmer_mask(k);
  const uint32_t   lsh  = 2 * k - 2;
  orc_kmdata f = 0, r = 0;
  uint32_t   load = 0;
  uint64_t   cnt  = 0;

  for (uint64_t i = 0; i < n; i++) {
    int c = orc_base_code(bases[i]);
    if (c < 0) {            /* breaker: restart the k-mer */
      load = 0; f = 0; r = 0;
      continue;
    }
    f = ((f << 2) | (orc_kmdata)c) & mask;
    r = (r >> 2) | ((orc_kmdata)(c ^ 2) << lsh);
    if (load < k) load++;
    if (load < k) continue;

    orc_kmdata m;
    if      (mode == ORC_FORWARD) m = f;
    else if (mode == ORC_REVERSE) m = r;
    else                          m = (f < r) ? f : r;

    if ((m & suffix_mask) != suffix_test) continue;   /* merylOp-countSimple.C:88-90 (mask 0: keeps everything) */

    if (cnt < cap) {
      if (out_hi) out_hi[cnt] = (uint64_t)(m >> 64);
      if (out_lo) out_lo[cnt] = (uint64_t)m;
    }
    cnt++;
  }
  return cnt;
}

uint64_t orc_enumerate_kmers(const char *bases, uint64_t n, uint32_t k, int mode,
                             uint64_t *out_hi, uint64_t *out_lo, uint64_t cap) {
  return enumerate_filtered(bases, n, k, mode, 0, 0, out_hi, out_lo, cap);
}

static int cmp_kmdata(const void *a, const void *b) {
  orc_kmdata x = *(const orc_kmdata *)a, y = *(const orc_kmdata *)b;
  return (x < y) ? -1 : (x > y);
}

/* ------------------------------------------------------------------------
 * Brute force: src/meryl-simple/meryl-simple.C:131-187 (collect, std::sort,
 * scan runs).  Count arithmetic follows merylCountArray.C:345-360: a uint32
 * incremented once per instance, so it wraps mod 2^32.
 * ---------------------------------------------------------------------- */
static int count_brute_filtered(const char *bases, uint64_t n, uint32_t k, int mode,
                                orc_kmdata suffix_mask, orc_kmdata suffix_test,
                                uint64_t **keys_hi, uint64_t **keys_lo, uint32_t **counts,
                                uint64_t *n_distinct, uint64_t *n_instances) {
  *keys_hi = NULL; *keys_lo = NULL; *counts = NULL; *n_distinct = 0; *n_instances = 0;
  if (k == 0 || k > 64) return -1;

  uint64_t ni = enumerate_filtered(bases, n, k, mode, suffix_mask, suffix_test, NULL, NULL, 0);
  *n_instances = ni;
  if (ni == 0) return 0;

  uint64_t *hi = (uint64_t *)malloc(sizeof(uint64_t) * ni);
  uint64_t *lo = (uint64_t *)malloc(sizeof(uint64_t) * ni);
  orc_kmdata *all = (orc_kmdata *)malloc(sizeof(orc_kmdata) * ni);
  if (!hi || !lo || !all) { free(hi); free(lo); free(all); return -2; }

  enumerate_filtered(bases, n, k, mode, suffix_mask, suffix_test, hi, lo, ni);
  for (uint64_t i = 0; i < ni; i++)
    all[i] = ((orc_kmdata)hi[i] << 64) | lo[i];

  qsort(all, ni, sizeof(orc_kmdata), cmp_kmdata);

  uint64_t nd = 1;
  for (uint64_t i = 1; i < ni; i++)
    if (all[i] != all[i - 1]) nd++;

  uint32_t *cn = (uint32_t *)malloc(sizeof(uint32_t) * nd);
  if (!cn) { free(hi); free(lo); free(all); return -2; }

  uint64_t d = 0;
  cn[0] = 1; hi[0] = (uint64_t)(all[0] >> 64); lo[0] = (uint64_t)all[0];
  for (uint64_t i = 1; i < ni; i++) {
    if (all[i] != all[i - 1]) {
      d++;
      cn[d] = 0;
      hi[d] = (uint64_t)(all[i] >> 64);
      lo[d] = (uint64_t)all[i];
    }
    cn[d]++;                    /* uint32 wrap, merylCountArray.C:357 */
  }
  free(all);

  *keys_hi = hi; *keys_lo = lo; *counts = cn; *n_distinct = nd;
  return 0;
}

int orc_count_brute(const char *bases, uint64_t n, uint32_t k, int mode,
                    uint64_t **keys_hi, uint64_t **keys_lo, uint32_t **counts,
                    uint64_t *n_distinct, uint64_t *n_instances) {
  return count_brute_filtered(bases, n, k, mode, 0, 0, keys_hi, keys_lo, counts, n_distinct, n_instances);
}

/* `count-suffix=<bases>`: merylOp.H:139-147 packs the string with kmerTiny::addR (2-bit codes, last base in the lowest
 * bits), merylOp-countSimple.C:50-58 builds a mask of as many base pairs, :88-93 keeps a k-mer only if its low bits equal
 * the packed string.  (The reference then strips those bits for its direct-index table and re-appends them on output,
 * :92,231-233 -- the database holds the full k-mer either way.) */
int orc_count_brute_suffix(const char *bases, uint64_t n, uint32_t k, int mode, const char *count_suffix,
                           uint64_t **keys_hi, uint64_t **keys_lo, uint32_t **counts,
                           uint64_t *n_distinct, uint64_t *n_instances) {
  orc_kmdata mask = 0, test = 0;
  for (const char *p = count_suffix; p && *p; p++) {
    const int c = orc_base_code(*p);
    if (c < 0) return -1;
    test = (test << 2) | (orc_kmdata)c;
    mask = (mask << 2) | 3;
  }
  return count_brute_filtered(bases, n, k, mode, mask, test, keys_hi, keys_lo, counts, n_distinct, n_instances);
}

void orc_free(void *p) { free(p); }

/* `meryl print` text form, merylOp-nextMer.C:665-677: kmer string (first base
 * = most significant 2 bits), a tab, the value. */
void orc_kmer_to_string(uint64_t hi, uint64_t lo, uint32_t k, char *out) {
  static const char acgt[4] = { 'A', 'C', 'T', 'G' };
  orc_kmdata m = ((orc_kmdata)hi << 64) | lo;
  for (uint32_t i = 0; i < k; i++)
    out[i] = acgt[(unsigned)(m >> (2 * (k - 1 - i))) & 3];
  out[k] = 0;
}

/* ------------------------------------------------------------------------
 * Homopolymer compression hook, merylInput.C:261-268: applied in place to
 * every chunk loadBases() returns; _lastByte (the last byte of the previous
 * chunk when the sequence continues, else 0) suppresses a run that spans the
 * chunk boundary.  [NOT IN TREE] homopolyCompress() itself
 * (utility/src/sequence/sequence.C): runs are detected case-insensitively
 * (`|0x20`), the first byte of each run is kept as is, and leading bytes equal
 * to `skip` are dropped.
 * ---------------------------------------------------------------------- */
uint64_t orc_homopoly_compress(const char *in, uint64_t n, char *out, char last_byte) {
  uint64_t o = 0;
  char prev = last_byte;
  for (uint64_t i = 0; i < n; i++) {
    char c = in[i];
    if (prev != 0 && ((c | 0x20) == (prev | 0x20)))
      continue;
    out[o++] = c;
    prev = c;
  }
  return o;
}

/* ------------------------------------------------------------------------
 * configureCounting, merylOp-count.C:118-403.
 * ---------------------------------------------------------------------- */

/* countNumberOfBits64 [NOT IN TREE, utility/src/bits]: number of bits needed
 * to represent the value, with 0 -> 0... the historical implementation
 * returns 1 for 0; both callers below are insensitive to that (argument is
 * always >= 1 except expMaxCount which only matters for >= 2^16). */
static uint64_t count_bits64(uint64_t v) {
  uint64_t b = 0;
  while (v) { b++; v >>= 1; }
  return b ? b : 1;
}

/* merylOp-count.C:118-165 (findExpectedSimpleSize); lowBits_t is uint16
 * (merylOp-countSimple.C:41-48). */
static uint64_t simple_size(uint32_t k, uint64_t n_est, uint32_t csl) {
  const uint32_t low_bits = 16;
  if (2 * k - 2 * csl > 42)                                    /* :142 */
    return UINT64_MAX;
  uint64_t n_entries = (uint64_t)1 << (2 * k - 2 * csl);      /* :124 */
  uint64_t exp_max   = (uint64_t)(0.004 * (double)n_est);      /* :126 */
  uint64_t exp_bits  = count_bits64(exp_max) + 1;              /* :127 */
  uint64_t extra     = (exp_bits < low_bits) ? 0 : (exp_bits - low_bits);   /* :128 */
  uint64_t low_mem   = n_entries * low_bits;                   /* :130 */
  uint64_t high_mem  = n_entries * extra;                      /* :131 */
  return (low_mem + high_mem) / 8;                             /* :132 */
}

/* merylOp-count.C:173-227 (findBestPrefixSize) */
static void best_prefix(const orc_config *c, uint64_t n_est, uint64_t mem_allowed,
                        uint32_t *best_prefix_, uint64_t *mem_used_) {
  const uint32_t k          = c->k;
  const uint32_t seg_bits   = 1 * c->page_size * 8;            /* pagesPerSegment()==1, merylCountArray.H:106-107 */
  const uint32_t seg_bytes  = 1 * c->page_size;

  *best_prefix_ = 0;
  *mem_used_    = UINT64_MAX;

  for (uint32_t wp = 1; wp < 2 * k - 1; wp++) {                /* :197 */
    uint64_t n_prefix    = (uint64_t)1 << wp;
    uint64_t kpp         = n_est / n_prefix + 1;               /* :199 */
    uint64_t kps         = seg_bits / (2 * k - wp);            /* :200 */
    uint64_t spp         = kpp / kps + 1;                      /* :201 */

    if (wp + count_bits64(spp) + count_bits64(seg_bytes) >= 64)   /* :203 */
      break;

    uint64_t struct_mem  = (uint64_t)c->sizeof_count_array * n_prefix + 8 * n_prefix * spp;   /* :206-207 */
    uint64_t data_min    = n_prefix * seg_bytes;               /* :208 */
    uint64_t data_mem    = n_prefix * spp * seg_bytes;         /* :209 */
    uint64_t total       = struct_mem + data_mem;              /* :210 */

    if (struct_mem + data_min > mem_allowed)                   /* :216 */
      break;

    /* :219 -- note the comparison wraps when *mem_used_ == UINT64_MAX? no: the
     * left side is small; plain unsigned compare. */
    if ((wp > 9) && (total + (uint64_t)16 * wp * 1024 * 1024 < *mem_used_)) {
      *mem_used_    = total;
      *best_prefix_ = wp;
    }

    /* :224 -- 16 * memoryUsed_ overflows (wraps) while memoryUsed_ is still
     * UINT64_MAX, exactly as in the reference's uint64 arithmetic. */
    if (total > (uint64_t)16 * *mem_used_)
      break;
  }
}

int orc_configure_counting(orc_config *c) {
  if (c->k == 0 || c->k > 64) return -1;
  if (c->page_size == 0)          c->page_size = 4096;
  if (c->sizeof_count_array == 0) c->sizeof_count_array = 3232;

  c->use_simple = 0; c->w_prefix = 0; c->n_prefix = 0; c->w_data = 0;
  c->n_batches = 1;  c->memory_used = 0;

  uint64_t mem_simple  = simple_size(c->k, c->n_kmers_estimate, c->count_suffix_length);   /* :340 */
  uint64_t mem_complex = UINT64_MAX;
  uint32_t best        = 0;
  uint32_t n_batches   = 1;

  if (c->k > 5) {                                              /* :353 */
    /* :354-355 -- nBatches is incremented once more after the fitting call */
    for (n_batches = 1; mem_complex > c->memory_allowed; n_batches++) {
      best_prefix(c, c->n_kmers_estimate / n_batches, c->memory_allowed, &best, &mem_complex);
      if (n_batches > (1u << 20)) return -2;                   /* reference would loop forever */
    }
    /* findBestValues (:232-295) only reads bestPrefix */
    c->w_prefix = best;
    c->n_prefix = (uint64_t)1 << best;
    c->w_data   = 2 * c->k - best;
  }

  if ((mem_simple < mem_complex) && (mem_simple < c->memory_allowed)) {   /* :368-372 */
    c->use_simple  = 1;
    c->memory_used = mem_simple;
  } else {
    c->use_simple  = 0;
    c->memory_used = mem_complex;
  }
  if (c->count_suffix_length > 0) {                            /* :379-382 */
    c->use_simple  = 1;
    c->memory_used = mem_simple;
  }

  c->n_batches      = n_batches;
  c->memory_simple  = mem_simple;
  c->memory_complex = mem_complex;
  return 0;
}

/* ------------------------------------------------------------------------
 * Synthetic reads (our own generator, counter-based so that any
 * box -- and the GPU generator kernel -- regenerates identical bytes).
 * ---------------------------------------------------------------------- */
static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

/* Repeat families ("10 % of bases in repeat families to create a count tail"): the genome is
 * cut into blocks of `unit` bases; a block is a repeat with probability repeat_ppm/1e6 and then shows one of `families`
 * template sequences, chosen with a cubic skew (family 0 is by far the most frequent, like an Alu). */
static uint64_t synth_genome_pos(uint64_t gpos, uint64_t s_rep, uint64_t rep_thresh, uint32_t unit, uint32_t families) {
  if (rep_thresh == 0) return gpos;
  const uint64_t blk = gpos / unit;
  const uint64_t h   = splitmix64(s_rep ^ blk);
  if ((uint64_t)(uint32_t)h >= rep_thresh) return gpos;
  const uint64_t u = h >> 32;                       /* 32 uniform bits -> u^3 in 32-bit fixed point */
  const uint64_t a = (u * u) >> 32;
  const uint64_t b = (a * u) >> 32;
  const uint64_t fam = (b * (uint64_t)families) >> 32;
  return (1ull << 62) + fam * (uint64_t)unit + (gpos - blk * unit);
}

uint64_t orc_synth_reads(uint64_t seed, uint64_t genome_len, uint64_t first_read, uint64_t n_reads,
                         uint32_t read_len, uint32_t sub_rate_ppm, uint32_t n_rate_ppm, char *out) {
  return orc_synth_reads_ex(seed, genome_len, first_read, n_reads, read_len, sub_rate_ppm, n_rate_ppm, 0, 1, 1, out);
}

uint64_t orc_synth_reads_ex(uint64_t seed, uint64_t genome_len, uint64_t first_read, uint64_t n_reads,
                            uint32_t read_len, uint32_t sub_rate_ppm, uint32_t n_rate_ppm,
                            uint32_t repeat_ppm, uint32_t repeat_unit, uint32_t repeat_families, char *out) {
  static const char acgt[4] = { 'A', 'C', 'T', 'G' };
  const uint64_t rep_thresh = (uint64_t)repeat_ppm * 4294967296ull / 1000000ull;
  const uint64_t s_rep      = splitmix64(seed + 3 * 0x632be59bd9b4e019ull);
  if (repeat_unit == 0) repeat_unit = 1;
  if (repeat_families == 0) repeat_families = 1;
  const uint64_t span       = genome_len - read_len + 1;
  const uint64_t sub_thresh = (uint64_t)sub_rate_ppm * 4294967296ull / 1000000ull;   /* compare against 32 random bits */
  const uint64_t n_thresh   = (uint64_t)n_rate_ppm   * 4294967296ull / 1000000ull;
  const uint64_t s_genome   = splitmix64(seed + 0 * 0x632be59bd9b4e019ull);
  const uint64_t s_read     = splitmix64(seed + 1 * 0x632be59bd9b4e019ull);
  const uint64_t s_error    = splitmix64(seed + 2 * 0x632be59bd9b4e019ull);
  uint64_t o = 0;
  for (uint64_t rr = 0; rr < n_reads; rr++) {
    const uint64_t r     = first_read + rr;
    const uint64_t hr    = splitmix64(s_read ^ r);
    const uint64_t start = (uint64_t)(((unsigned __int128)hr * span) >> 64);   /* multiply-high range reduction */
    const int      rev   = (int)(hr & 1);
    for (uint32_t j = 0; j < read_len; j++) {
      uint64_t gpos = rev ? (start + read_len - 1 - j) : (start + j);
      uint32_t code = (uint32_t)(splitmix64(s_genome ^ synth_genome_pos(gpos, s_rep, rep_thresh, repeat_unit, repeat_families)) & 3);
      if (rev) code ^= 2;
      const uint64_t he = splitmix64(s_error ^ (r * read_len + j));
      const uint32_t e1 = (uint32_t)he;
      const uint32_t e2 = (uint32_t)(he >> 32);
      if ((uint64_t)e1 < sub_thresh)
        code = (code + 1 + (e1 % 3)) & 3;
      out[o++] = ((uint64_t)e2 < n_thresh) ? 'N' : acgt[code];
    }
    out[o++] = '.';
  }
  return o;
}
