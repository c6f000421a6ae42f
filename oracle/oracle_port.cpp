/*
 * oracle_port.cpp -- restatement of the reference's THREADED count algorithm,
 * used (a) as a second, structurally different oracle and (b) as the timed
 * CPU baseline (`cpu_baseline.kind == "port"` in bench.py).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  C++ because the reference's inner
 * sort is std::sort on 128-bit integers (merylCountArray.C:330); compiled with
 * the reference's optimisation flags (src/Makefile:106-108,127-131).
 *
 * Follows:
 *   merylOp-countThreads.C:138-231  loadBases: 2 MiB buffers, k-1 carry, '.' breakers
 *   merylOp-countThreads.C:235-280  insertKmers: canonical pick, prefix/suffix split,
 *                                   per-bucket spin lock, append
 *   merylCountArray.C:101-126,490-728   bit-packed suffix store in page-sized segments
 *   merylCountArray.C:276-289,323-365   unpack -> std::sort -> two-pass run-length
 *   merylOp-countThreads.C:452-459  final dump: 64 files in parallel, prefixes ascending,
 *                                   addBlock called for every prefix (empty ones too)
 */
#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <omp.h>

namespace {

typedef orc_kmdata kmdata;
typedef orc_kmvalu kmvalu;

/* One prefix bucket: `width`-bit suffixes appended MSB-first into uint64 words
 * held in segments of (pageBits - 512) bits (merylCountArray.C:117), grown a
 * segment at a time (:254-270). */
struct CountArray {
  uint32_t               width    = 0;
  uint64_t               seg_bits = 0;     /* bits per segment */
  uint64_t               n_bits   = 0;
  std::vector<uint64_t*> segs;

  void init(uint32_t w, uint32_t page_size) {
    width    = w;
    seg_bits = (uint64_t)page_size * 8 - 512;
  }
  ~CountArray() { for (auto s : segs) free(s); }

  inline uint64_t *word(uint64_t bit) {
    uint64_t seg = bit / seg_bits;
    while (seg >= segs.size())
      segs.push_back((uint64_t *)calloc(seg_bits / 64, sizeof(uint64_t)));
    return segs[seg] + (bit % seg_bits) / 64;
  }

  inline uint64_t *seg_ptr(uint64_t seg) {            /* addSegment, merylCountArray.C:254-270 (zeroed here) */
    while (seg >= segs.size())
      segs.push_back((uint64_t *)calloc(seg_bits / 64, sizeof(uint64_t)));
    return segs[seg];
  }

  /* append the low `width` bits of v.  Structure of merylCountArray::add (merylCountArray.C:490-728): one
   * division pair locates segment and bit, then the value lands in one, two or three words of this segment
   * (:553-623), or is split between the last word(s) of this segment and the first of the next (:628-725).
   * The caller masks v to `width` bits (insertKmers' wDataMask, merylOp-countThreads.C:250,255). */
  inline void add(kmdata v) {
    const uint64_t nb      = n_bits;
    const uint64_t seg     = nb / seg_bits;                         /* :497 */
    const uint64_t seg_pos = nb % seg_bits;                         /* :498 */
    n_bits += width;                                                /* :500 */
    const uint32_t word     = (uint32_t)(seg_pos / 64);             /* :505 */
    const uint32_t word_bgn = (uint32_t)(seg_pos % 64);             /* :506 */
    const uint32_t word_end = word_bgn + width;                     /* :507 */
    uint64_t *S = seg_ptr(seg);
    if (seg_pos + width <= seg_bits) {                              /* sameSeg, :541 */
      if (word_end <= 64) {                                         /* oneWord, :572-576 */
        S[word] |= (uint64_t)(v << (64 - word_end));
      } else if (word_end <= 128) {                                 /* twoWord, :578-597 */
        const uint32_t end_bits = width - (64 - word_bgn);
        S[word]     |= (uint64_t)(v >> end_bits);
        S[word + 1]  = (uint64_t)(v << (64 - end_bits));
      } else {                                                      /* thrWord, :600-621 */
        const uint32_t end_bits = width - 64 - (64 - word_bgn);
        S[word]     |= (uint64_t)(v >> (64 + end_bits));
        S[word + 1]  = (uint64_t)(v >> end_bits);
        S[word + 2]  = (uint64_t)(v << (64 - end_bits));
      }
    } else {                                                        /* the value continues in the next segment, :628-725 */
      const uint32_t this_bits = (uint32_t)(seg_bits - seg_pos);
      const uint32_t next_bits = width - this_bits;
      uint64_t *N = seg_ptr(seg + 1);
      S = segs[seg];                                                /* seg_ptr may have grown the vector */
      if (this_bits <= 64) {                                        /* oneThis, :682-689 */
        S[word] |= (uint64_t)(v >> (width - this_bits));
      } else {                                                      /* twoThis, :691-701 */
        S[word]     |= (uint64_t)(v >> (next_bits + 64));
        S[word + 1]  = (uint64_t)(v >> next_bits);
      }
      if (next_bits <= 64) {                                        /* oneNext, :705-709 */
        N[0] = (uint64_t)(v << (64 - next_bits));
      } else {                                                      /* twoNext, :711-717 */
        N[0] = (uint64_t)(v >> (next_bits - 64));
        N[1] = (uint64_t)(v << (128 - next_bits));
      }
    }
  }

  /* merylCountArray.C:750-847 get(kk) */
  inline kmdata get(uint64_t kk) {
    uint64_t bit  = kk * width;
    uint32_t left = width;
    kmdata   v    = 0;
    while (left > 0) {
      uint64_t *w    = word(bit);
      uint32_t  off  = (uint32_t)((bit % seg_bits) % 64);
      uint32_t  room = 64 - off;
      uint64_t  in_seg_left = seg_bits - (bit % seg_bits);
      if (room > in_seg_left) room = (uint32_t)in_seg_left;
      uint32_t  take = (left < room) ? left : room;
      uint64_t  piece = (*w >> (64 - off - take)) & ((take == 64) ? ~0ull : ((1ull << take) - 1));
      v = (v << take) | piece;
      bit  += take;
      left -= take;
    }
    return v;
  }

  void clear() { for (auto s : segs) free(s); segs.clear(); n_bits = 0; }
};

inline int base_code(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'T': case 't': return 2;
    case 'G': case 'g': return 3;
    default:            return -1;
  }
}

struct Chunk { uint64_t bgn, end; };   /* [bgn,end) of the stream, bgn already includes the k-1 carry */

/* merylOp-countThreads.C:138-231.  The stream already carries the '.' that
 * the loader appends at every end-of-sequence (:214-215) / end-of-file (:196),
 * so restating the loader reduces to choosing the buffer cut points: a buffer
 * holds at most 2 MiB, starts with the previous buffer's last k-1 bytes unless
 * that buffer ended in a breaker (:149-155,221-222), and loading stops once
 * fewer than 512 bytes are free (:173). */
std::vector<Chunk> make_chunks(const char *bases, uint64_t n, uint32_t k) {
  const uint64_t buf_max = 2ull * 1024 * 1024;       /* :413 */
  const uint64_t kl      = k - 1;
  std::vector<Chunk> out;
  uint64_t pos = 0;
  bool     carry = false;
  while (pos < n) {
    uint64_t have = carry ? kl : 0;
    uint64_t room = buf_max - have - 512;             /* stop when < 512 free */
    uint64_t end  = (pos + room < n) ? (pos + room) : n;
    Chunk c;
    c.bgn = carry ? (pos - kl) : pos;
    c.end = end;
    out.push_back(c);
    carry = (end >= kl) && (bases[end - 1] != '.');   /* :221-222 */
    pos = end;
  }
  return out;
}

}  // namespace

extern "C"
int orc_count_threaded(const char *bases, uint64_t n, uint32_t k, int mode,
                       uint32_t w_prefix, int threads,
                       orc_block_cb cb, void *ctx,
                       uint64_t *n_distinct, uint64_t *n_instances) {
  if (k == 0 || k > 64 || w_prefix < 6 || w_prefix >= 2 * k) return -1;
  if (threads <= 0) threads = omp_get_max_threads();

  const uint32_t w_data   = 2 * k - w_prefix;
  const uint64_t n_prefix = (uint64_t)1 << w_prefix;
  kmdata full_mask = 0; full_mask = ~full_mask; full_mask >>= (128 - 2 * k);
  kmdata data_mask = 0; data_mask = ~data_mask; data_mask >>= (128 - w_data);   /* merylOp-count.C:282-286 */

  std::vector<CountArray>        data(n_prefix);                 /* merylOp-countThreads.C:56-60 */
  std::vector<std::atomic_flag>  lock(n_prefix);                 /* :45 */
  for (uint64_t pp = 0; pp < n_prefix; pp++) { data[pp].init(w_data, 4096); lock[pp].clear(); }

  std::vector<Chunk> chunks = make_chunks(bases, n, k);
  std::atomic<uint64_t> added(0);

  /* insertKmers, :235-280 */
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
  for (int64_t ci = 0; ci < (int64_t)chunks.size(); ci++) {
    const Chunk c = chunks[ci];
    kmdata   f = 0, r = 0;
    uint32_t load = 0;
    uint64_t mine = 0;
    for (uint64_t i = c.bgn; i < c.end; i++) {
      int code = base_code(bases[i]);
      if (code < 0) { load = 0; f = 0; r = 0; continue; }
      f = ((f << 2) | (kmdata)code) & full_mask;
      r = (r >> 2) | ((kmdata)(code ^ 2) << (2 * k - 2));
      if (load < k) load++;
      if (load < k) continue;

      bool use_f = (mode == ORC_FORWARD);
      if (mode == ORC_CANONICAL) use_f = (f < r);                 /* :245-246 */
      kmdata m  = use_f ? f : r;
      uint64_t pp = (uint64_t)(m >> w_data);                      /* :249,254 */
      kmdata   mm = m & data_mask;                                /* :250,255 */

      while (lock[pp].test_and_set(std::memory_order_acquire))   /* :271-272 */
        ;
      data[pp].add(mm);
      lock[pp].clear(std::memory_order_release);                  /* :278 */
      mine++;
    }
    added += mine;
  }

  /* final dump, :452-459 */
  const uint32_t n_files = 64;
  const uint64_t per_file = n_prefix / n_files;
  std::atomic<uint64_t> distinct(0);

#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
  for (int32_t ff = 0; ff < (int32_t)n_files; ff++) {
    for (uint64_t pp = ff * per_file; pp < (ff + 1) * per_file; pp++) {
      CountArray &a = data[pp];
      uint64_t n_suf = (a.width == 0) ? 0 : a.n_bits / a.width;   /* merylCountArray.C:324 */
      if (n_suf == 0) {                                           /* :454-457: empty bucket still dumped */
        if (cb) cb(ctx, pp, 0, nullptr, nullptr);
        continue;
      }
      std::vector<kmdata> suf(n_suf);
      for (uint64_t kk = 0; kk < n_suf; kk++) suf[kk] = a.get(kk);   /* :283-284 */
      a.clear();

      std::sort(suf.begin(), suf.end());                          /* :330 */

      uint64_t nk = 1;                                            /* :334-338 */
      for (uint64_t kk = 1; kk < n_suf; kk++)
        if (suf[kk - 1] != suf[kk]) nk++;

      std::vector<kmdata> s(nk);
      std::vector<kmvalu> cnt(nk);
      uint64_t o = 0;                                             /* :345-360 */
      cnt[0] = 1; s[0] = suf[0];
      for (uint64_t kk = 1; kk < n_suf; kk++) {
        if (suf[kk - 1] != suf[kk]) { o++; cnt[o] = 0; s[o] = suf[kk]; }
        cnt[o]++;
      }
      distinct += nk;
      if (cb) cb(ctx, pp, nk, s.data(), cnt.data());              /* :472-475 */
    }
  }

  if (n_distinct)  *n_distinct  = distinct.load();
  if (n_instances) *n_instances = added.load();
  return 0;
}

namespace {
/* per-file digests of the (k-mer, count) stream: what the full-size parity test compares with the same sums taken on
 * the GPU result (tests/test_gpu_parity.py); all arithmetic wraps mod 2^64 */
struct Digest { uint32_t w_data, file_shift; uint64_t v[64][4]; };
const uint64_t DG_C1 = 0x9E3779B97F4A7C15ull, DG_C2 = 0xC2B2AE3D27D4EB4Full, DG_C3 = 0x165667B19E3779F9ull;
void digest_cb(void *ctx, uint64_t prefix, uint64_t nk, const kmdata *s, const kmvalu *c) {
  Digest *D = (Digest *)ctx;               /* one thread per file (merylOp-countThreads.C:452-459): no locking needed */
  uint64_t *d = D->v[prefix >> D->file_shift];            /* file = top six bits of the prefix */
  for (uint64_t i = 0; i < nk; i++) {
    const kmdata key = ((kmdata)prefix << D->w_data) | s[i];
    const uint64_t lo = (uint64_t)key, hi = (uint64_t)(key >> 64);
    const uint64_t x = lo ^ (hi * DG_C3), cnt = (uint64_t)c[i];
    d[0] += 1;
    d[1] += cnt;
    d[2] += (x * DG_C1) * cnt;
    d[3] += ((x ^ DG_C2) * (x | 1ull)) * cnt;
  }
}
}  // namespace

/* out[64][4]: per file n_distinct, sum of counts, sum (x*C1)*count, sum ((x^C2)*(x|1))*count, x = lo ^ hi*C3 */
extern "C"
int orc_count_threaded_digest(const char *bases, uint64_t n, uint32_t k, int mode, uint32_t w_prefix, int threads,
                              uint64_t *out, uint64_t *n_distinct, uint64_t *n_instances) {
  Digest *D = new Digest();
  memset(D->v, 0, sizeof(D->v));
  D->w_data = 2 * k - w_prefix;
  D->file_shift = w_prefix - 6;
  int rc = orc_count_threaded(bases, n, k, mode, w_prefix, threads, digest_cb, D, n_distinct, n_instances);
  memcpy(out, D->v, sizeof(D->v));
  delete D;
  return rc;
}

namespace {
/* digests of every file AND the k-mers themselves of the files in `mask` (the full-size parity test compares four whole
 * files element by element beside the 64 digests) */
struct DigestSome {
  Digest D; uint64_t mask; uint32_t w_prefix;
  std::vector<std::vector<kmdata>> keys;     /* per prefix, filled for masked files only */
  std::vector<std::vector<kmvalu>> counts;
};
void digest_some_cb(void *ctx, uint64_t prefix, uint64_t nk, const kmdata *s, const kmvalu *c) {
  DigestSome *S = (DigestSome *)ctx;
  digest_cb(&S->D, prefix, nk, s, c);
  if (!((S->mask >> (prefix >> S->D.file_shift)) & 1ull)) return;
  S->keys[prefix].resize(nk);
  S->counts[prefix].assign(c, c + nk);
  for (uint64_t i = 0; i < nk; i++) S->keys[prefix][i] = ((kmdata)prefix << S->D.w_data) | s[i];
}
}  // namespace

/* as orc_count_threaded_digest; plus, for the files whose bit is set in file_mask, their (k-mer, count) stream in ascending
 * order: file_start[65] (entries of the masked files before file f; unmasked files are empty), malloc'ed arrays the caller frees */
extern "C"
int orc_count_threaded_digest_collect(const char *bases, uint64_t n, uint32_t k, int mode, uint32_t w_prefix, int threads,
                                      uint64_t *out, uint64_t *n_distinct, uint64_t *n_instances, uint64_t file_mask,
                                      uint64_t *file_start, uint64_t **keys_hi, uint64_t **keys_lo, uint32_t **counts) {
  *keys_hi = *keys_lo = nullptr; *counts = nullptr;
  DigestSome *S = new DigestSome();
  memset(S->D.v, 0, sizeof(S->D.v));
  S->D.w_data = 2 * k - w_prefix;
  S->D.file_shift = w_prefix - 6;
  S->mask = file_mask; S->w_prefix = w_prefix;
  S->keys.resize((uint64_t)1 << w_prefix);
  S->counts.resize((uint64_t)1 << w_prefix);
  int rc = orc_count_threaded(bases, n, k, mode, w_prefix, threads, digest_some_cb, S, n_distinct, n_instances);
  memcpy(out, S->D.v, sizeof(S->D.v));
  if (rc == 0) {
    uint64_t total = 0;
    const uint64_t per_file = (uint64_t)1 << (w_prefix - 6);
    for (uint64_t f = 0; f < 64; f++) {
      file_start[f] = total;
      for (uint64_t pp = f * per_file; pp < (f + 1) * per_file; pp++) total += S->keys[pp].size();
    }
    file_start[64] = total;
    uint64_t *hi = (uint64_t *)malloc(8 * (total ? total : 1)), *lo = (uint64_t *)malloc(8 * (total ? total : 1));
    uint32_t *cn = (uint32_t *)malloc(4 * (total ? total : 1));
    uint64_t o = 0;
    for (size_t pp = 0; pp < S->keys.size(); pp++)
      for (size_t i = 0; i < S->keys[pp].size(); i++, o++) {
        hi[o] = (uint64_t)(S->keys[pp][i] >> 64);
        lo[o] = (uint64_t)S->keys[pp][i];
        cn[o] = S->counts[pp][i];
      }
    *keys_hi = hi; *keys_lo = lo; *counts = cn;
  }
  delete S;
  return rc;
}

namespace {
struct Collect {
  uint32_t w_data;
  std::vector<std::vector<kmdata>> keys;     /* per prefix */
  std::vector<std::vector<kmvalu>> counts;
};
void collect_cb(void *ctx, uint64_t prefix, uint64_t nk, const kmdata *s, const kmvalu *c) {
  Collect *C = (Collect *)ctx;               /* called concurrently, distinct prefix per call */
  C->keys[prefix].resize(nk);
  C->counts[prefix].assign(c, c + nk);
  for (uint64_t i = 0; i < nk; i++)
    C->keys[prefix][i] = ((kmdata)prefix << C->w_data) | s[i];
}
}  // namespace

extern "C"
int orc_count_threaded_collect(const char *bases, uint64_t n, uint32_t k, int mode,
                               uint32_t w_prefix, int threads,
                               uint64_t **keys_hi, uint64_t **keys_lo, uint32_t **counts,
                               uint64_t *n_distinct, uint64_t *n_instances) {
  *keys_hi = *keys_lo = nullptr; *counts = nullptr;
  Collect C;
  C.w_data = 2 * k - w_prefix;
  C.keys.resize((uint64_t)1 << w_prefix);
  C.counts.resize((uint64_t)1 << w_prefix);
  int rc = orc_count_threaded(bases, n, k, mode, w_prefix, threads, collect_cb, &C, n_distinct, n_instances);
  if (rc) return rc;
  uint64_t nd = *n_distinct;
  if (nd == 0) return 0;
  uint64_t *hi = (uint64_t *)malloc(8 * nd), *lo = (uint64_t *)malloc(8 * nd);
  uint32_t *cn = (uint32_t *)malloc(4 * nd);
  uint64_t o = 0;
  for (size_t pp = 0; pp < C.keys.size(); pp++)
    for (size_t i = 0; i < C.keys[pp].size(); i++, o++) {
      hi[o] = (uint64_t)(C.keys[pp][i] >> 64);
      lo[o] = (uint64_t)C.keys[pp][i];
      cn[o] = C.counts[pp][i];
    }
  *keys_hi = hi; *keys_lo = lo; *counts = cn;
  return 0;
}
