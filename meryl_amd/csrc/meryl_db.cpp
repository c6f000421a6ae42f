// meryl_db.cpp -- meryl database writer / reader (include/meryl_db.h).
//
// Mirrors the reference's merylFileWriter / merylBlockWriter as used by the
// count path (src/meryl/merylOp.C:262; src/meryl/merylOp-countThreads.C:48,404,
// 453-464; src/meryl/merylCountArray.C:472-475).  The classes themselves live in
// the ABSENT submodule marbl/meryl-utility (utility/src/kmers-v1/kmers-writer.C,
// kmers-writer-block.C, kmers-files.C, bits/stuffedBits-v1*.C), so the byte layout
// below restates that library's v1 database format; what the reference tree pins is
// its SHAPE (documentation/source/usage.rst:13-45, reference.rst:73-77).  PARITY
// UNPINNED at the byte level -- see DESIGN.md "database encoding" for the list of
// layout assumptions (A1..A9 below).
//
//   A1  files: <db>/merylIndex, <db>/0xBBBBBB.merylData, <db>/0xBBBBBB.merylIndex,
//       BBBBBB = file number as six binary digits.
//   A2  every stuffedBits object is stored as: u64 blockLenMaxBits, u32 nBlocks,
//       u32 nBlocksMax, u64 blockBgn[nBlocks], u64 blockLen[nBlocks], then per
//       block: u64 nWords, u64 nWordsAllocated, u64 words[nWords]; words are
//       native-endian, fields packed MSB-first.
//   A3  unary code of v = v zero bits then a one bit.
//   A4  data block = magic "merylDat" "aFile00\n" (two u64), prefix(64),
//       nKmers(64), kCode(8)=1, unaryBits(32), binaryBits(32), k1(64)=0,
//       cCode(8)=1, c1(64)=0, c2(64)=0; unaryBits = ceil(log2(nKmers)) (smallest
//       u with 2^u >= nKmers), binaryBits = suffixSize - unaryBits.
//   A5  k-mer i: unary(hi_i - hi_{i-1}) where hi = suffix >> binaryBits, then the
//       low binaryBits in binary (two fields (binaryBits-64)+64 when > 64).
//   A6  values: 32-bit binary each, after all k-mers of the block.
//   A7  per-file index = raw array of numBlocks {u64 blockPrefix, u64
//       blockPosition (byte offset in the data file), u64 numKmers}.
//   A8  master index stuffedBits: magic "merylInd" "ex__v.02", prefixSize(32),
//       suffixSize(32), numFilesBits(32), numBlocksBits(32), flags(32)
//       (bit0 = multiset), histogram.
//   A9  histogram: numUnique(64) numDistinct(64) numTotal(64) nPairs(64), then
//       (value(64), occurrences(64)) ascending by value.
#include "../../include/meryl_db.h"
#include "../../include/meryl_gpu_count.h"

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <sys/stat.h>
#include <vector>

namespace {

thread_local std::string g_db_error;
void db_err(const char *fmt, const char *a = "", const char *b = "") {
  char buf[1024];
  snprintf(buf, sizeof(buf), fmt, a, b);
  g_db_error = buf;
}

constexpr uint64_t MAGIC_DAT1 = 0x7461446c7972656dull;   // "merylDat"
constexpr uint64_t MAGIC_DAT2 = 0x0a3030656c694661ull;   // "aFile00\n"
constexpr uint64_t MAGIC_IDX1 = 0x646e496c7972656dull;   // "merylInd"  (usage.rst:15)
constexpr uint64_t MAGIC_IDX2 = 0x32302e765f5f7865ull;   // "ex__v.02"
constexpr uint64_t STUFFED_BLOCK_BITS = 16ull * 1024 * 1024 * 8;   // default stuffedBits block

// ---- MSB-first bit packing into uint64 words (stuffedBits semantics) --------
struct BitWriter {
  std::vector<uint64_t> w;
  uint64_t pos = 0;                      // bits written

  inline void reserve_bits(uint64_t bits) { w.reserve((pos + bits + 63) / 64 + 1); }
  inline void put(uint32_t width, uint64_t value) {       // setBinary(width, value), width <= 64
    if (width == 0) return;
    if (width < 64) value &= (1ull << width) - 1;
    const uint64_t word = pos >> 6;
    const uint32_t off  = (uint32_t)(pos & 63);
    if (w.size() < word + 2) w.resize(word + 2, 0);
    const uint32_t room = 64 - off;
    if (width <= room) {
      w[word] |= value << (room - width);
    } else {
      w[word]     |= value >> (width - room);
      w[word + 1] |= value << (64 - (width - room));
    }
    pos += width;
  }
  inline void put_unary(uint64_t v) {                     // A3: v zeros, then a one
    pos += v;
    put(1, 1);
  }
  uint64_t words() const { return (pos + 63) / 64; }
};

struct BitReader {
  const uint64_t *w = nullptr;
  uint64_t nbits = 0, pos = 0;
  inline uint64_t get(uint32_t width) {
    if (width == 0) return 0;
    const uint64_t word = pos >> 6;
    const uint32_t off  = (uint32_t)(pos & 63);
    const uint32_t room = 64 - off;
    uint64_t v;
    if (width <= room) {
      v = (w[word] >> (room - width));
    } else {
      v = (w[word] << (width - room)) | (w[word + 1] >> (64 - (width - room)));
    }
    if (width < 64) v &= (1ull << width) - 1;
    pos += width;
    return v;
  }
  inline uint64_t get_unary() {
    uint64_t v = 0;
    for (;;) {
      const uint64_t word = pos >> 6;
      const uint32_t off  = (uint32_t)(pos & 63);
      const uint64_t rest = w[word] << off;
      if (rest == 0) { v += 64 - off; pos += 64 - off; continue; }
      const uint32_t lz = (uint32_t)__builtin_clzll(rest);
      v += lz;
      pos += lz + 1;
      return v;
    }
  }
};

// A2: dump one stuffedBits object.  Fields never straddle our single logical
// stream, so it is cut into blocks of STUFFED_BLOCK_BITS only when it is larger.
bool dump_stuffed(FILE *f, const BitWriter &bw) {
  const uint64_t total = bw.pos;
  uint32_t nblocks = (uint32_t)((total + STUFFED_BLOCK_BITS - 1) / STUFFED_BLOCK_BITS);
  if (nblocks == 0) nblocks = 1;
  const uint32_t nmax = std::max<uint32_t>(64, nblocks);
  std::vector<uint64_t> bgn(nblocks), len(nblocks);
  for (uint32_t i = 0; i < nblocks; i++) {
    bgn[i] = (uint64_t)i * STUFFED_BLOCK_BITS;
    len[i] = std::min<uint64_t>(STUFFED_BLOCK_BITS, total - bgn[i]);
  }
  const uint64_t lenmax = STUFFED_BLOCK_BITS;
  if (fwrite(&lenmax, 8, 1, f) != 1 || fwrite(&nblocks, 4, 1, f) != 1 || fwrite(&nmax, 4, 1, f) != 1) return false;
  if (fwrite(bgn.data(), 8, nblocks, f) != nblocks || fwrite(len.data(), 8, nblocks, f) != nblocks) return false;
  for (uint32_t i = 0; i < nblocks; i++) {
    const uint64_t nw = (len[i] + 63) / 64, nalloc = STUFFED_BLOCK_BITS / 64;
    if (fwrite(&nw, 8, 1, f) != 1 || fwrite(&nalloc, 8, 1, f) != 1) return false;
    if (nw && fwrite(bw.w.data() + bgn[i] / 64, 8, nw, f) != nw) return false;
  }
  return true;
}

// reads one stuffedBits object into a contiguous word vector
bool load_stuffed(FILE *f, std::vector<uint64_t> &words, uint64_t &nbits) {
  uint64_t lenmax; uint32_t nblocks, nmax;
  if (fread(&lenmax, 8, 1, f) != 1 || fread(&nblocks, 4, 1, f) != 1 || fread(&nmax, 4, 1, f) != 1) return false;
  if (nblocks == 0 || nblocks > (1u << 20)) return false;
  std::vector<uint64_t> bgn(nblocks), len(nblocks);
  if (fread(bgn.data(), 8, nblocks, f) != nblocks || fread(len.data(), 8, nblocks, f) != nblocks) return false;
  words.clear();
  nbits = 0;
  for (uint32_t i = 0; i < nblocks; i++) {
    uint64_t nw, nalloc;
    if (fread(&nw, 8, 1, f) != 1 || fread(&nalloc, 8, 1, f) != 1) return false;
    if (bgn[i] % 64 != 0 || bgn[i] != nbits) return false;        // only what dump_stuffed writes
    const size_t at = words.size();
    words.resize(at + nw + 2, 0);
    if (nw && fread(words.data() + at, 8, nw, f) != nw) return false;
    words.resize(at + nw);
    nbits += len[i];
    if (i + 1 < nblocks && len[i] % 64 != 0) return false;
  }
  words.push_back(0); words.push_back(0);                          // slack for two-word reads
  return true;
}

std::string block_name(const std::string &dir, uint32_t ff, bool index) {        // A1
  char bits[8];
  for (int i = 0; i < 6; i++) bits[i] = ((ff >> (5 - i)) & 1) ? '1' : '0';
  bits[6] = 0;
  return dir + "/0x" + bits + (index ? ".merylIndex" : ".merylData");
}

struct FileIndexEntry { uint64_t prefix, position, n_kmers; };                   // A7

}  // namespace

struct mdb_writer {
  std::string dir;
  uint32_t k = 0, prefix_size = 0, suffix_size = 0, num_blocks_bits = 0;
  uint64_t blocks_per_file = 0;
  FILE *dat[MGC_NUM_FILES];
  std::vector<FileIndexEntry> index[MGC_NUM_FILES];
  uint64_t bytes[MGC_NUM_FILES];
  // histogram: small values dense, big values sparse (merylHistogram keeps the same split)
  std::vector<uint64_t> hist_small[MGC_NUM_FILES];
  std::map<uint64_t, uint64_t> hist_big[MGC_NUM_FILES];
  uint64_t n_distinct[MGC_NUM_FILES], n_total[MGC_NUM_FILES];
  bool failed = false;
};

extern "C" const char *mdb_last_error(void) { return g_db_error.c_str(); }

extern "C" mdb_writer *mdb_writer_open(const char *path, uint32_t k, uint32_t w_prefix) {
  if (!path || k == 0 || k > 64 || w_prefix < MGC_NUM_FILES_BITS || w_prefix > 2 * k) {
    db_err("mdb_writer_open: bad arguments");
    return nullptr;
  }
  if (mkdir(path, 0777) != 0 && errno != EEXIST) { db_err("mdb_writer_open: cannot create '%s': %s", path, strerror(errno)); return nullptr; }
  mdb_writer *w = new mdb_writer();
  w->dir = path;
  w->k = k;
  w->prefix_size = w_prefix;
  w->suffix_size = 2 * k - w_prefix;
  w->num_blocks_bits = w_prefix - MGC_NUM_FILES_BITS;
  w->blocks_per_file = 1ull << w->num_blocks_bits;
  for (int ff = 0; ff < MGC_NUM_FILES; ff++) {
    w->dat[ff] = nullptr; w->bytes[ff] = 0; w->n_distinct[ff] = 0; w->n_total[ff] = 0;
    w->hist_small[ff].assign(1024, 0);
  }
  return w;
}

extern "C" int mdb_writer_add_block(mdb_writer *w, uint64_t prefix, uint64_t n, const uint64_t *slo,
                                    const uint64_t *shi, const uint32_t *counts) {
  if (!w || (n && (!slo || !counts))) return MGC_EINVAL;
  if (prefix >> w->prefix_size) { db_err("add_block: prefix out of range"); return MGC_EINVAL; }
  const uint32_t ff = (uint32_t)(prefix >> w->num_blocks_bits);          // file = top 6 bits of the prefix
  if (!w->dat[ff]) {
    w->dat[ff] = fopen(block_name(w->dir, ff, false).c_str(), "wb");
    if (!w->dat[ff]) { db_err("add_block: cannot open data file in '%s': %s", w->dir.c_str(), strerror(errno)); w->failed = true; return MGC_EINVAL; }
    w->index[ff].reserve(w->blocks_per_file);
  }
  if (!w->index[ff].empty() && w->index[ff].back().prefix >= prefix) { db_err("add_block: prefixes of a file must ascend"); return MGC_ESTATE; }

  // A4
  uint32_t unary_bits = 0;
  for (uint64_t sum = 1; sum < n; sum <<= 1) unary_bits++;
  if (unary_bits > w->suffix_size) unary_bits = w->suffix_size;
  const uint32_t binary_bits = w->suffix_size - unary_bits;
  const bool wide = w->suffix_size > 64;

  BitWriter bw;
  bw.reserve_bits(64 * 8 + n * (binary_bits + 2 + 32));
  bw.put(64, MAGIC_DAT1); bw.put(64, MAGIC_DAT2);
  bw.put(64, prefix); bw.put(64, n);
  bw.put(8, 1); bw.put(32, unary_bits); bw.put(32, binary_bits); bw.put(64, 0);
  bw.put(8, 1); bw.put(64, 0); bw.put(64, 0);

  // A5
  uint64_t last_hi = 0;
  for (uint64_t i = 0; i < n; i++) {
    const uint64_t lo = slo[i], hi = (wide && shi) ? shi[i] : 0;
    uint64_t top;                                  // suffix >> binary_bits (fits 64 bits: unary_bits <= 64)
    if (binary_bits >= 64) top = (binary_bits == 64) ? hi : (hi >> (binary_bits - 64));
    else                   top = (binary_bits == 0) ? lo : ((lo >> binary_bits) | (wide ? (hi << (64 - binary_bits)) : 0));
    bw.put_unary(top - last_hi);
    last_hi = top;
    if (binary_bits <= 64) bw.put(binary_bits, lo);
    else { bw.put(binary_bits - 64, hi); bw.put(64, lo); }
  }
  // A6
  for (uint64_t i = 0; i < n; i++) bw.put(32, counts[i]);

  FileIndexEntry e;
  e.prefix = prefix; e.position = w->bytes[ff]; e.n_kmers = n;
  w->index[ff].push_back(e);
  const long before = ftell(w->dat[ff]);
  if (!dump_stuffed(w->dat[ff], bw)) { db_err("add_block: write failed in '%s'", w->dir.c_str()); w->failed = true; return MGC_EINVAL; }
  w->bytes[ff] += (uint64_t)(ftell(w->dat[ff]) - before);

  // value histogram, per file (merged at close) -- merylBlockWriter adds every value
  std::vector<uint64_t> &hs = w->hist_small[ff];
  for (uint64_t i = 0; i < n; i++) {
    const uint64_t v = counts[i];
    if (v < hs.size()) hs[v]++; else w->hist_big[ff][v]++;
    w->n_total[ff] += v;
  }
  w->n_distinct[ff] += n;
  return MGC_OK;
}

extern "C" int mdb_writer_close(mdb_writer *w) {
  if (!w) return MGC_EINVAL;
  bool ok = !w->failed;
  // per-file indexes (A7).  A file that received no block at all still gets its (empty)
  // data file and an index of empty blocks, so that the directory always has 64+64+1 files.
  for (uint32_t ff = 0; ff < MGC_NUM_FILES; ff++) {
    if (!w->dat[ff]) {
      for (uint64_t bb = 0; bb < w->blocks_per_file && ok; bb++)
        ok = (mdb_writer_add_block(w, ((uint64_t)ff << w->num_blocks_bits) | bb, 0, nullptr, nullptr, nullptr) == MGC_OK);
    }
    if (w->dat[ff]) { if (fclose(w->dat[ff]) != 0) ok = false; w->dat[ff] = nullptr; }
    FILE *f = fopen(block_name(w->dir, ff, true).c_str(), "wb");
    if (!f) { ok = false; continue; }
    // index slot = block number inside the file; blocks that were never added stay {prefix,0,0}
    std::vector<FileIndexEntry> full(w->blocks_per_file);
    for (uint64_t bb = 0; bb < w->blocks_per_file; bb++) { full[bb].prefix = ((uint64_t)ff << w->num_blocks_bits) | bb; full[bb].position = 0; full[bb].n_kmers = 0; }
    for (const FileIndexEntry &e : w->index[ff]) full[e.prefix & (w->blocks_per_file - 1)] = e;
    if (fwrite(full.data(), sizeof(FileIndexEntry), full.size(), f) != full.size()) ok = false;
    if (fclose(f) != 0) ok = false;
  }
  // master index (A8) + histogram (A9)
  std::map<uint64_t, uint64_t> hist;
  uint64_t n_distinct = 0, n_total = 0;
  for (uint32_t ff = 0; ff < MGC_NUM_FILES; ff++) {
    for (size_t v = 0; v < w->hist_small[ff].size(); v++) if (w->hist_small[ff][v]) hist[v] += w->hist_small[ff][v];
    for (auto &kv : w->hist_big[ff]) hist[kv.first] += kv.second;
    n_distinct += w->n_distinct[ff];
    n_total += w->n_total[ff];
  }
  BitWriter bw;
  bw.put(64, MAGIC_IDX1); bw.put(64, MAGIC_IDX2);
  bw.put(32, w->prefix_size); bw.put(32, w->suffix_size); bw.put(32, MGC_NUM_FILES_BITS); bw.put(32, w->num_blocks_bits);
  bw.put(32, 0);
  bw.put(64, hist.count(1) ? hist[1] : 0); bw.put(64, n_distinct); bw.put(64, n_total); bw.put(64, hist.size());
  for (auto &kv : hist) { bw.put(64, kv.first); bw.put(64, kv.second); }
  FILE *f = fopen((w->dir + "/merylIndex").c_str(), "wb");
  if (!f || !dump_stuffed(f, bw)) ok = false;
  if (f && fclose(f) != 0) ok = false;
  if (!ok && g_db_error.empty()) db_err("mdb_writer_close: I/O error in '%s'", w->dir.c_str());
  delete w;
  return ok ? MGC_OK : MGC_EINVAL;
}

// ---------------------------------------------------------------------------
// reader (what `meryl print` / dumpIndex need; used by the tests to round-trip)
// ---------------------------------------------------------------------------
struct mdb_reader {
  std::string dir;
  mdb_info info;
  std::vector<uint64_t> hist_v, hist_n;
};

extern "C" mdb_reader *mdb_reader_open(const char *path) {
  if (!path) return nullptr;
  FILE *f = fopen((std::string(path) + "/merylIndex").c_str(), "rb");
  if (!f) { db_err("mdb_reader_open: '%s/merylIndex': %s", path, strerror(errno)); return nullptr; }
  std::vector<uint64_t> words; uint64_t nbits = 0;
  const bool ok = load_stuffed(f, words, nbits);
  fclose(f);
  if (!ok) { db_err("mdb_reader_open: '%s/merylIndex' is not a meryl index", path); return nullptr; }
  BitReader br; br.w = words.data(); br.nbits = nbits;
  const uint64_t m1 = br.get(64), m2 = br.get(64);
  if (m1 != MAGIC_IDX1 || (m2 & 0x0000ffffffffffffull) != (MAGIC_IDX2 & 0x0000ffffffffffffull)) {
    db_err("mdb_reader_open: bad magic in '%s/merylIndex'", path);
    return nullptr;
  }
  mdb_reader *r = new mdb_reader();
  r->dir = path;
  r->info.prefix_size = (uint32_t)br.get(32);
  r->info.suffix_size = (uint32_t)br.get(32);
  r->info.num_files_bits = (uint32_t)br.get(32);
  r->info.num_blocks_bits = (uint32_t)br.get(32);
  r->info.flags = (uint32_t)br.get(32);
  r->info.k = (r->info.prefix_size + r->info.suffix_size) / 2;
  r->info.num_unique = br.get(64);
  r->info.num_distinct = br.get(64);
  r->info.num_total = br.get(64);
  r->info.hist_len = br.get(64);
  for (uint64_t i = 0; i < r->info.hist_len; i++) { r->hist_v.push_back(br.get(64)); r->hist_n.push_back(br.get(64)); }
  return r;
}

extern "C" int mdb_reader_info(const mdb_reader *r, mdb_info *info) {
  if (!r || !info) return MGC_EINVAL;
  *info = r->info;
  return MGC_OK;
}

extern "C" int mdb_reader_histogram(const mdb_reader *r, uint64_t *values, uint64_t *occ) {
  if (!r || !values || !occ) return MGC_EINVAL;
  for (size_t i = 0; i < r->hist_v.size(); i++) { values[i] = r->hist_v[i]; occ[i] = r->hist_n[i]; }
  return MGC_OK;
}

extern "C" int mdb_reader_read_file(mdb_reader *r, uint32_t ff, uint64_t **klo, uint64_t **khi, uint32_t **cnt,
                                    uint64_t *n_out) {
  if (!r || ff >= MGC_NUM_FILES || !klo || !cnt || !n_out) return MGC_EINVAL;
  *klo = nullptr; if (khi) *khi = nullptr; *cnt = nullptr; *n_out = 0;
  const uint64_t nblocks = 1ull << r->info.num_blocks_bits;
  std::vector<FileIndexEntry> idx(nblocks);
  FILE *fi = fopen(block_name(r->dir, ff, true).c_str(), "rb");
  if (!fi || fread(idx.data(), sizeof(FileIndexEntry), nblocks, fi) != nblocks) { if (fi) fclose(fi); db_err("read_file: bad index in '%s'", r->dir.c_str()); return MGC_EINVAL; }
  fclose(fi);
  uint64_t total = 0;
  for (auto &e : idx) total += e.n_kmers;
  uint64_t *lo = (uint64_t *)malloc(8 * (total ? total : 1)), *hi = (uint64_t *)calloc(total ? total : 1, 8);
  uint32_t *cn = (uint32_t *)malloc(4 * (total ? total : 1));
  FILE *fd = fopen(block_name(r->dir, ff, false).c_str(), "rb");
  if (!fd || !lo || !hi || !cn) { if (fd) fclose(fd); free(lo); free(hi); free(cn); db_err("read_file: cannot open data file in '%s'", r->dir.c_str()); return MGC_EINVAL; }
  const uint32_t ss = r->info.suffix_size;
  uint64_t o = 0;
  std::vector<uint64_t> words;
  bool ok = true;
  for (uint64_t bb = 0; bb < nblocks && ok; bb++) {
    const FileIndexEntry &e = idx[bb];
    if (fseek(fd, (long)e.position, SEEK_SET) != 0) { ok = false; break; }
    uint64_t nbits = 0;
    if (!load_stuffed(fd, words, nbits)) { ok = false; break; }
    BitReader br; br.w = words.data(); br.nbits = nbits;
    if (br.get(64) != MAGIC_DAT1 || br.get(64) != MAGIC_DAT2) { ok = false; break; }
    const uint64_t prefix = br.get(64), n = br.get(64);
    (void)br.get(8);
    const uint32_t ub = (uint32_t)br.get(32), bb_bits = (uint32_t)br.get(32);
    (void)br.get(64); (void)br.get(8); (void)br.get(64); (void)br.get(64);
    if (n != e.n_kmers || prefix != e.prefix || ub + bb_bits != ss) { ok = false; break; }
    uint64_t top = 0;
    for (uint64_t i = 0; i < n; i++) {
      top += br.get_unary();
      // suffix = top << bb_bits | binary ; full k-mer = prefix << ss | suffix  (128-bit)
      unsigned __int128 suf;
      if (bb_bits <= 64) suf = ((unsigned __int128)top << bb_bits) | (bb_bits ? br.get(bb_bits) : 0);
      else { const uint64_t h = br.get(bb_bits - 64), l = br.get(64); suf = ((unsigned __int128)top << bb_bits) | ((unsigned __int128)h << 64) | l; }
      const unsigned __int128 full = ((unsigned __int128)prefix << ss) | suf;
      lo[o + i] = (uint64_t)full;
      hi[o + i] = (uint64_t)(full >> 64);
    }
    for (uint64_t i = 0; i < n; i++) cn[o + i] = (uint32_t)br.get(32);
    o += n;
  }
  fclose(fd);
  if (!ok || o != total) { free(lo); free(hi); free(cn); db_err("read_file: corrupt block in '%s'", r->dir.c_str()); return MGC_EINVAL; }
  *klo = lo; if (khi) *khi = hi; else free(hi); *cnt = cn; *n_out = total;
  return MGC_OK;
}

extern "C" void mdb_reader_close(mdb_reader *r) { delete r; }
extern "C" void mdb_free(void *p) { free(p); }

// ---------------------------------------------------------------------------
// count result -> database
// ---------------------------------------------------------------------------
namespace {
int write_cb(void *ctx, uint64_t prefix, uint64_t n, const uint64_t *slo, const uint64_t *shi, const uint32_t *cnt) {
  return mdb_writer_add_block((mdb_writer *)ctx, prefix, n, slo, shi, cnt);
}
}  // namespace

extern "C" int mgc_write_database(struct mgc_session *s, const char *path, int host_threads) {
  if (!s || !path) return MGC_EINVAL;
  mgc_result_info info;
  int rc = mgc_get_result_info(s, &info);
  if (rc != MGC_OK) return rc;
  mdb_writer *w = mdb_writer_open(path, (info.w_prefix + info.w_data) / 2, info.w_prefix);
  if (!w) return MGC_EINVAL;
  rc = mgc_finish(s, write_cb, w, host_threads);
  const int rc2 = mdb_writer_close(w);
  return rc != MGC_OK ? rc : rc2;
}
