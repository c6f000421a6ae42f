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
//   A10 labels (meryl2's constant count label, src/meryl2/merylCountArray.C:469-471): flags
//       bits 8..15 hold labelSize; when it is non-zero every block carries, after its
//       values, one labelSize-bit binary field per k-mer.  labelSize 0 writes exactly the
//       bytes of A1..A9.
#include "../../include/meryl_db.h"
#include "../../include/meryl_gpu_count.h"
#include "mdb_layout.h"

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace {

using namespace mdb;

thread_local std::string g_db_error;
void db_err(const char *fmt, const char *a = "", const char *b = "") {
  char buf[1024];
  snprintf(buf, sizeof(buf), fmt, a, b);
  g_db_error = buf;
}

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

// Every read is bounds-checked against nbits: a truncated, corrupt or foreign file sets `bad` and yields
// zeros instead of walking off the buffer.
struct BitReader {
  const uint64_t *w = nullptr;
  uint64_t nbits = 0, pos = 0;
  bool bad = false;
  inline uint64_t get(uint32_t width) {
    if (width == 0) return 0;
    if (bad || pos + width > nbits) { bad = true; return 0; }
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
      if (bad || pos >= nbits) { bad = true; return 0; }
      const uint64_t word = pos >> 6;
      const uint32_t off  = (uint32_t)(pos & 63);
      const uint64_t rest = w[word] << off;
      if (rest == 0) { v += 64 - off; pos += 64 - off; continue; }
      const uint32_t lz = (uint32_t)__builtin_clzll(rest);
      v += lz;
      pos += lz + 1;
      if (pos > nbits) { bad = true; return 0; }
      return v;
    }
  }
};

// A2: dump one stuffedBits object.  Fields never straddle our single logical
// stream, so it is cut into blocks of STUFFED_BLOCK_BITS only when it is larger.
void dump_stuffed(std::vector<unsigned char> &out, const BitWriter &bw) {
  const uint64_t total = bw.pos;
  const uint32_t nblocks = (uint32_t)stuffed_sub_blocks(total);
  const uint32_t nmax = std::max<uint32_t>(64, nblocks);
  out.resize(stuffed_bytes(total));
  unsigned char *o = out.data();
  auto put = [&](const void *p, size_t n) { memcpy(o, p, n); o += n; };
  const uint64_t lenmax = STUFFED_BLOCK_BITS;
  put(&lenmax, 8); put(&nblocks, 4); put(&nmax, 4);
  for (uint32_t i = 0; i < nblocks; i++) { const uint64_t bgn = (uint64_t)i * STUFFED_BLOCK_BITS; put(&bgn, 8); }
  for (uint32_t i = 0; i < nblocks; i++) {
    const uint64_t len = std::min<uint64_t>(STUFFED_BLOCK_BITS, total - (uint64_t)i * STUFFED_BLOCK_BITS);
    put(&len, 8);
  }
  for (uint32_t i = 0; i < nblocks; i++) {
    const uint64_t bgn = (uint64_t)i * STUFFED_BLOCK_BITS, len = std::min<uint64_t>(STUFFED_BLOCK_BITS, total - bgn);
    const uint64_t nw = (len + 63) / 64, nalloc = STUFFED_BLOCK_WORDS;
    put(&nw, 8); put(&nalloc, 8);
    if (nw) put(bw.w.data() + bgn / 64, 8 * nw);
  }
}
bool dump_stuffed(FILE *f, const BitWriter &bw) {
  std::vector<unsigned char> buf;
  dump_stuffed(buf, bw);
  return fwrite(buf.data(), 1, buf.size(), f) == buf.size();
}
bool pwrite_all(int fd, const void *p, uint64_t n, uint64_t off) {
  const unsigned char *c = reinterpret_cast<const unsigned char *>(p);
  while (n) {
    const ssize_t r = pwrite(fd, c, n, (off_t)off);
    if (r < 0) { if (errno == EINTR) continue; return false; }
    c += r; n -= (uint64_t)r; off += (uint64_t)r;
  }
  return true;
}

// reads one stuffedBits object into a contiguous word vector
bool load_stuffed(FILE *f, std::vector<uint64_t> &words, uint64_t &nbits) {
  uint64_t lenmax; uint32_t nblocks, nmax;
  if (fread(&lenmax, 8, 1, f) != 1 || fread(&nblocks, 4, 1, f) != 1 || fread(&nmax, 4, 1, f) != 1) return false;
  if (nblocks == 0 || nblocks > (1u << 20) || lenmax == 0 || lenmax > (1ull << 40)) return false;
  std::vector<uint64_t> bgn(nblocks), len(nblocks);
  if (fread(bgn.data(), 8, nblocks, f) != nblocks || fread(len.data(), 8, nblocks, f) != nblocks) return false;
  words.clear();
  nbits = 0;
  for (uint32_t i = 0; i < nblocks; i++) {
    uint64_t nw, nalloc;
    if (fread(&nw, 8, 1, f) != 1 || fread(&nalloc, 8, 1, f) != 1) return false;
    if (bgn[i] % 64 != 0 || bgn[i] != nbits) return false;        // only what dump_stuffed writes
    if (len[i] > lenmax || nw != (len[i] + 63) / 64) return false;
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
std::string part_data_name(const std::string &dir, uint32_t ff, uint32_t part) {
  return block_name(dir, ff, false) + ".part" + std::to_string(part);
}
std::string part_meta_name(const std::string &dir, uint32_t part) { return dir + "/merylParts." + std::to_string(part); }

typedef mdb_index_entry FileIndexEntry;                                           // A7

constexpr uint64_t PART_MAGIC = 0x7472615064626d31ull;

}  // namespace

struct mdb_writer {
  std::string dir;
  uint32_t k = 0, prefix_size = 0, suffix_size = 0, num_blocks_bits = 0, label_size = 0;
  uint32_t part = 0, n_parts = 1;
  uint64_t blocks_per_file = 0;
  int dat[MGC_NUM_FILES];                                  // data files: every write is a pwrite at an explicit offset
  std::vector<FileIndexEntry> index[MGC_NUM_FILES];
  uint64_t bytes[MGC_NUM_FILES];
  // histogram: small values dense, big values sparse (merylHistogram keeps the same split)
  std::vector<uint64_t> hist_small[MGC_NUM_FILES];
  std::map<uint64_t, uint64_t> hist_big[MGC_NUM_FILES];
  std::map<uint64_t, uint64_t> hist_extra;                 // mdb_writer_add_histogram
  // add_block runs on several threads (one per file): the first failure is kept here, under the lock, and
  // surfaces through mdb_last_error() of whoever closes the writer
  std::atomic<bool> failed{false};
  std::mutex err_lock;
  std::string first_error;
  void fail(const char *fmt, const char *a = "", const char *b = "") {
    db_err(fmt, a, b);
    std::lock_guard<std::mutex> g(err_lock);
    if (!failed.exchange(true)) first_error = g_db_error;
  }
  std::string data_name(uint32_t ff) const { return n_parts > 1 ? part_data_name(dir, ff, part) : block_name(dir, ff, false); }
};

extern "C" const char *mdb_last_error(void) { return g_db_error.c_str(); }

extern "C" mdb_writer *mdb_writer_open_ex(const char *path, uint32_t k, uint32_t w_prefix, uint32_t label_size,
                                          uint32_t part, uint32_t n_parts) {
  if (!path || k == 0 || k > 64 || w_prefix < MGC_NUM_FILES_BITS || w_prefix > 2 * k || label_size > 64 ||
      n_parts == 0 || part >= n_parts) {
    db_err("mdb_writer_open: bad arguments");
    return nullptr;
  }
  if (mkdir(path, 0777) != 0 && errno != EEXIST) { db_err("mdb_writer_open: cannot create '%s': %s", path, strerror(errno)); return nullptr; }
  mdb_writer *w = new mdb_writer();
  w->dir = path;
  w->k = k;
  w->prefix_size = w_prefix;
  w->suffix_size = 2 * k - w_prefix;
  w->label_size = label_size;
  w->part = part;
  w->n_parts = n_parts;
  w->num_blocks_bits = w_prefix - MGC_NUM_FILES_BITS;
  w->blocks_per_file = 1ull << w->num_blocks_bits;
  for (int ff = 0; ff < MGC_NUM_FILES; ff++) {
    w->dat[ff] = -1; w->bytes[ff] = 0;
    w->hist_small[ff].assign(1024, 0);
  }
  return w;
}

extern "C" mdb_writer *mdb_writer_open(const char *path, uint32_t k, uint32_t w_prefix) {
  return mdb_writer_open_ex(path, k, w_prefix, 0, 0, 1);
}

namespace {
// opens file ff's data file on first use and checks the ascending-prefix rule
int writer_file_ready(mdb_writer *w, uint32_t ff, uint64_t prefix) {
  if (w->dat[ff] < 0) {
    w->dat[ff] = open(w->data_name(ff).c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (w->dat[ff] < 0) { w->fail("add_block: cannot open data file in '%s': %s", w->dir.c_str(), strerror(errno)); return MGC_EINVAL; }
    w->index[ff].reserve(std::min<uint64_t>(w->blocks_per_file, 1u << 16));
  }
  if (!w->index[ff].empty() && w->index[ff].back().prefix >= prefix) { db_err("add_block: prefixes of a file must ascend"); return MGC_ESTATE; }
  return MGC_OK;
}
}  // namespace

extern "C" int mdb_writer_add_block_labelled(mdb_writer *w, uint64_t prefix, uint64_t n, const uint64_t *slo,
                                             const uint64_t *shi, const uint32_t *counts, const uint64_t *labels,
                                             uint64_t label) {
  if (!w || (n && (!slo || !counts))) return MGC_EINVAL;
  if (w->prefix_size < 64 && (prefix >> w->prefix_size)) { db_err("add_block: prefix out of range"); return MGC_EINVAL; }
  const uint32_t ff = (uint32_t)(prefix >> w->num_blocks_bits);          // file = top 6 bits of the prefix
  int rc = writer_file_ready(w, ff, prefix);
  if (rc != MGC_OK) return rc;

  // A4
  const uint32_t unary_bits = unary_bits_for(n, w->suffix_size);
  const uint32_t binary_bits = w->suffix_size - unary_bits;
  const bool wide = w->suffix_size > 64;

  BitWriter bw;
  bw.reserve_bits(64 * 9 + n * (binary_bits + 2 + VALUE_BITS + w->label_size));
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
    if (top < last_hi) { db_err("add_block: suffixes of a block must ascend"); return MGC_EINVAL; }
    bw.put_unary(top - last_hi);
    last_hi = top;
    if (binary_bits <= 64) bw.put(binary_bits, lo);
    else { bw.put(binary_bits - 64, hi); bw.put(64, lo); }
  }
  // A6
  for (uint64_t i = 0; i < n; i++) bw.put(VALUE_BITS, counts[i]);
  // A10
  if (w->label_size)
    for (uint64_t i = 0; i < n; i++) bw.put(w->label_size, labels ? labels[i] : label);

  FileIndexEntry e;
  e.prefix = prefix; e.position = w->bytes[ff]; e.n_kmers = n;
  w->index[ff].push_back(e);
  std::vector<unsigned char> dump;
  dump_stuffed(dump, bw);
  if (!pwrite_all(w->dat[ff], dump.data(), dump.size(), w->bytes[ff])) { w->fail("add_block: write failed in '%s': %s", w->dir.c_str(), strerror(errno)); return MGC_EINVAL; }
  w->bytes[ff] += dump.size();

  // value histogram, per file (merged at close) -- merylBlockWriter adds every value
  std::vector<uint64_t> &hs = w->hist_small[ff];
  for (uint64_t i = 0; i < n; i++) {
    const uint64_t v = counts[i];
    if (v < hs.size()) hs[v]++; else w->hist_big[ff][v]++;
  }
  return MGC_OK;
}

extern "C" int mdb_writer_add_block(mdb_writer *w, uint64_t prefix, uint64_t n, const uint64_t *slo,
                                    const uint64_t *shi, const uint32_t *counts) {
  return mdb_writer_add_block_labelled(w, prefix, n, slo, shi, counts, nullptr, 0);
}

extern "C" int mdb_writer_reserve_encoded(mdb_writer *w, uint32_t ff, uint64_t nbytes, const mdb_index_entry *entries,
                                          uint64_t n_entries, uint64_t *file_offset) {
  if (!w || ff >= MGC_NUM_FILES || !file_offset || (n_entries && !entries)) return MGC_EINVAL;
  if (nbytes == 0) return n_entries ? MGC_EINVAL : MGC_OK;
  for (uint64_t i = 0; i < n_entries; i++) {
    if ((entries[i].prefix >> w->num_blocks_bits) != ff || entries[i].position >= nbytes ||
        (i && entries[i].prefix <= entries[i - 1].prefix)) { db_err("add_encoded: bad index entries"); return MGC_EINVAL; }
  }
  if (n_entries) {
    int rc = writer_file_ready(w, ff, entries[0].prefix);
    if (rc != MGC_OK) return rc;
  } else if (w->dat[ff] < 0) {                              // no block starts here: the bytes continue the file's last block
    db_err("add_encoded: continuation bytes for a file that holds no block yet");
    return MGC_ESTATE;
  }
  for (uint64_t i = 0; i < n_entries; i++) {
    FileIndexEntry e = entries[i];
    e.position += w->bytes[ff];
    w->index[ff].push_back(e);
  }
  *file_offset = w->bytes[ff];
  w->bytes[ff] += nbytes;
  return MGC_OK;
}

extern "C" int mdb_writer_write_at(mdb_writer *w, uint32_t ff, uint64_t file_offset, const void *bytes, uint64_t nbytes) {
  if (!w || ff >= MGC_NUM_FILES || (nbytes && !bytes) || w->dat[ff] < 0) return MGC_EINVAL;
  if (!pwrite_all(w->dat[ff], bytes, nbytes, file_offset)) { w->fail("add_encoded: write failed in '%s': %s", w->dir.c_str(), strerror(errno)); return MGC_EINVAL; }
  return MGC_OK;
}

extern "C" int mdb_writer_add_encoded(mdb_writer *w, uint32_t ff, const void *bytes, uint64_t nbytes,
                                      const mdb_index_entry *entries, uint64_t n_entries) {
  if (nbytes && !bytes) return MGC_EINVAL;
  uint64_t off = 0;
  int rc = mdb_writer_reserve_encoded(w, ff, nbytes, entries, n_entries, &off);
  if (rc != MGC_OK || nbytes == 0) return rc;
  return mdb_writer_write_at(w, ff, off, bytes, nbytes);
}

extern "C" int mdb_writer_add_histogram(mdb_writer *w, const uint64_t *values, const uint64_t *occ, uint64_t n_pairs) {
  if (!w || (n_pairs && (!values || !occ))) return MGC_EINVAL;
  std::lock_guard<std::mutex> g(w->err_lock);
  for (uint64_t i = 0; i < n_pairs; i++) if (occ[i]) w->hist_extra[values[i]] += occ[i];
  return MGC_OK;
}

namespace {

void merged_histogram(mdb_writer *w, std::map<uint64_t, uint64_t> &hist) {
  hist = w->hist_extra;
  for (uint32_t ff = 0; ff < MGC_NUM_FILES; ff++) {
    for (size_t v = 0; v < w->hist_small[ff].size(); v++) if (w->hist_small[ff][v]) hist[v] += w->hist_small[ff][v];
    for (auto &kv : w->hist_big[ff]) hist[kv.first] += kv.second;
  }
}

// per-file indexes (A7) + master index (A8) + histogram (A9) from the state of a writer whose data files are complete
bool write_indexes(mdb_writer *w) {
  bool ok = true;
  for (uint32_t ff = 0; ff < MGC_NUM_FILES; ff++) {
    FILE *f = fopen(block_name(w->dir, ff, true).c_str(), "wb");
    if (!f) { ok = false; continue; }
    // index slot = block number inside the file; blocks that were never added stay {prefix,0,0}
    std::vector<FileIndexEntry> full(w->blocks_per_file);
    for (uint64_t bb = 0; bb < w->blocks_per_file; bb++) { full[bb].prefix = ((uint64_t)ff << w->num_blocks_bits) | bb; full[bb].position = 0; full[bb].n_kmers = 0; }
    for (const FileIndexEntry &e : w->index[ff]) full[e.prefix & (w->blocks_per_file - 1)] = e;
    if (fwrite(full.data(), sizeof(FileIndexEntry), full.size(), f) != full.size()) ok = false;
    if (fclose(f) != 0) ok = false;
  }
  std::map<uint64_t, uint64_t> hist;
  merged_histogram(w, hist);
  uint64_t n_distinct = 0, n_total = 0;
  for (auto &kv : hist) { n_distinct += kv.second; n_total += kv.first * kv.second; }
  BitWriter bw;
  bw.put(64, MAGIC_IDX1); bw.put(64, MAGIC_IDX2);
  bw.put(32, w->prefix_size); bw.put(32, w->suffix_size); bw.put(32, MGC_NUM_FILES_BITS); bw.put(32, w->num_blocks_bits);
  bw.put(32, (uint64_t)w->label_size << 8);                                       // flags: bit0 multiset, bits 8..15 labelSize (A10)
  bw.put(64, hist.count(1) ? hist[1] : 0); bw.put(64, n_distinct); bw.put(64, n_total); bw.put(64, hist.size());
  for (auto &kv : hist) { bw.put(64, kv.first); bw.put(64, kv.second); }
  FILE *f = fopen((w->dir + "/merylIndex").c_str(), "wb");
  if (!f || !dump_stuffed(f, bw)) ok = false;
  if (f && fclose(f) != 0) ok = false;
  return ok;
}

bool put_u64(FILE *f, uint64_t v) { return fwrite(&v, 8, 1, f) == 1; }
bool get_u64(FILE *f, uint64_t &v) { return fread(&v, 8, 1, f) == 1; }

}  // namespace

// Lets go of a writer WITHOUT writing indexes or a master index (a caller whose setup failed after the writer was opened:
// closing would leave a complete-looking empty database at the output path although the call fails).
extern "C" void mdb_writer_discard(mdb_writer *w) {
  if (!w) return;
  for (uint32_t ff = 0; ff < MGC_NUM_FILES; ff++) if (w->dat[ff] >= 0) { (void)close(w->dat[ff]); w->dat[ff] = -1; }
  delete w;
}

extern "C" int mdb_writer_close(mdb_writer *w) {
  if (!w) return MGC_EINVAL;
  bool ok = !w->failed.load();
  if (w->n_parts > 1) {
    // one part of a sharded database: data files stay under their part names; the side file carries what
    // mdb_merge_parts needs (index entries, sizes, histogram)
    for (uint32_t ff = 0; ff < MGC_NUM_FILES; ff++)
      if (w->dat[ff] >= 0) { if (close(w->dat[ff]) != 0) ok = false; w->dat[ff] = -1; }
    std::map<uint64_t, uint64_t> hist;
    merged_histogram(w, hist);
    const std::string tmp = part_meta_name(w->dir, w->part) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    bool wok = f != nullptr;
    if (f) {
      wok = put_u64(f, PART_MAGIC) && put_u64(f, w->k) && put_u64(f, w->prefix_size) && put_u64(f, w->label_size) &&
            put_u64(f, w->part) && put_u64(f, w->n_parts);
      for (uint32_t ff = 0; ff < MGC_NUM_FILES && wok; ff++) {
        wok = put_u64(f, w->bytes[ff]) && put_u64(f, w->index[ff].size());
        if (wok && !w->index[ff].empty())
          wok = fwrite(w->index[ff].data(), sizeof(FileIndexEntry), w->index[ff].size(), f) == w->index[ff].size();
      }
      wok = wok && put_u64(f, hist.size());
      for (auto &kv : hist) wok = wok && put_u64(f, kv.first) && put_u64(f, kv.second);
      if (fclose(f) != 0) wok = false;
      if (wok && rename(tmp.c_str(), part_meta_name(w->dir, w->part).c_str()) != 0) wok = false;
    }
    if (!wok) { ok = false; w->fail("mdb_writer_close: cannot write the part file in '%s': %s", w->dir.c_str(), strerror(errno)); }
  } else {
    // A file that received no block at all still gets its (empty) data file and an index of empty blocks, so that
    // the directory always has 64+64+1 files.
    for (uint32_t ff = 0; ff < MGC_NUM_FILES; ff++) {
      if (w->dat[ff] < 0) {
        for (uint64_t bb = 0; bb < w->blocks_per_file && ok; bb++)
          ok = (mdb_writer_add_block(w, ((uint64_t)ff << w->num_blocks_bits) | bb, 0, nullptr, nullptr, nullptr) == MGC_OK);
      }
      if (w->dat[ff] >= 0) { if (close(w->dat[ff]) != 0) ok = false; w->dat[ff] = -1; }
    }
    if (!write_indexes(w)) ok = false;
  }
  if (w->failed.load()) { std::lock_guard<std::mutex> g(w->err_lock); g_db_error = w->first_error; }
  else if (!ok) db_err("mdb_writer_close: I/O error in '%s': %s", w->dir.c_str(), strerror(errno));
  delete w;
  return ok ? MGC_OK : MGC_EINVAL;
}

// Stitches the parts of a sharded database (mdb_writer_open_ex) into the final 64+64+1 files.
extern "C" int mdb_merge_parts(const char *path, uint32_t n_parts) {
  if (!path || n_parts == 0) { db_err("mdb_merge_parts: bad arguments"); return MGC_EINVAL; }
  const std::string dir = path;
  mdb_writer *w = nullptr;
  struct PartFile { uint64_t bytes = 0; std::vector<FileIndexEntry> idx; };
  std::vector<std::vector<PartFile>> parts(n_parts, std::vector<PartFile>(MGC_NUM_FILES));
  for (uint32_t p = 0; p < n_parts; p++) {
    FILE *f = fopen(part_meta_name(dir, p).c_str(), "rb");
    if (!f) { db_err("mdb_merge_parts: part file missing in '%s': %s", path, strerror(errno)); delete w; return MGC_EINVAL; }
    uint64_t magic = 0, k = 0, wp = 0, ls = 0, pp = 0, np = 0;
    bool ok = get_u64(f, magic) && get_u64(f, k) && get_u64(f, wp) && get_u64(f, ls) && get_u64(f, pp) && get_u64(f, np) &&
              magic == PART_MAGIC && pp == p && np == n_parts;
    if (ok && !w) { w = mdb_writer_open_ex(path, (uint32_t)k, (uint32_t)wp, (uint32_t)ls, 0, 1); ok = w != nullptr; }
    if (ok && (w->k != k || w->prefix_size != wp || w->label_size != ls)) ok = false;
    for (uint32_t ff = 0; ff < MGC_NUM_FILES && ok; ff++) {
      uint64_t ne = 0;
      ok = get_u64(f, parts[p][ff].bytes) && get_u64(f, ne) && ne <= w->blocks_per_file;
      if (ok && ne) { parts[p][ff].idx.resize(ne); ok = fread(parts[p][ff].idx.data(), sizeof(FileIndexEntry), ne, f) == ne; }
    }
    uint64_t nh = 0;
    ok = ok && get_u64(f, nh);
    for (uint64_t i = 0; i < nh && ok; i++) { uint64_t v = 0, o = 0; ok = get_u64(f, v) && get_u64(f, o); if (ok) w->hist_extra[v] += o; }
    fclose(f);
    if (!ok) { db_err("mdb_merge_parts: bad part file in '%s'", path); delete w; return MGC_EINVAL; }
  }
  if (!w) { db_err("mdb_merge_parts: no parts in '%s'", path); return MGC_EINVAL; }
  // Validate EVERYTHING first (ranges ascend over the parts, every contributing data file exists with the size its side
  // file names): nothing is renamed or appended until the whole stitch is known to be possible, so a failure here leaves
  // the parts as they were and the call can be repeated.
  for (uint32_t ff = 0; ff < MGC_NUM_FILES; ff++) {
    uint64_t last = 0; bool any = false;
    for (uint32_t p = 0; p < n_parts; p++) {
      const PartFile &pf = parts[p][ff];
      if (pf.idx.empty()) continue;
      if (any && last >= pf.idx.front().prefix) { db_err("mdb_merge_parts: prefix ranges of the parts overlap in '%s'", path); delete w; return MGC_EINVAL; }
      for (size_t i = 1; i < pf.idx.size(); i++)
        if (pf.idx[i - 1].prefix >= pf.idx[i].prefix) { db_err("mdb_merge_parts: a part's index is not ascending in '%s'", path); delete w; return MGC_EINVAL; }
      last = pf.idx.back().prefix; any = true;
      struct stat st;
      if (stat(part_data_name(dir, ff, p).c_str(), &st) != 0 || (uint64_t)st.st_size != pf.bytes) {
        db_err("mdb_merge_parts: '%s' is missing or not of the size its part's side file names", part_data_name(dir, ff, p).c_str());
        delete w;
        return MGC_EINVAL;
      }
    }
  }
  bool ok = true;
  std::vector<char> buf(8u << 20);
  for (uint32_t ff = 0; ff < MGC_NUM_FILES && ok; ff++) {
    const std::string final_name = block_name(dir, ff, false);
    bool have_base = false;
    FILE *out = nullptr;
    for (uint32_t p = 0; p < n_parts && ok; p++) {
      PartFile &pf = parts[p][ff];
      if (pf.idx.empty()) continue;
      if (!w->index[ff].empty() && w->index[ff].back().prefix >= pf.idx.front().prefix) { db_err("mdb_merge_parts: prefix ranges of the parts overlap in '%s'", path); ok = false; break; }
      const std::string pname = part_data_name(dir, ff, p);
      if (!have_base) {                                   // first contributor: its file becomes the data file
        if (rename(pname.c_str(), final_name.c_str()) != 0) { db_err("mdb_merge_parts: rename in '%s': %s", path, strerror(errno)); ok = false; break; }
        have_base = true;
      } else {                                            // a file that straddles ranks: append
        if (!out) out = fopen(final_name.c_str(), "ab");
        FILE *in = fopen(pname.c_str(), "rb");
        if (!out || !in) { if (in) fclose(in); db_err("mdb_merge_parts: open in '%s': %s", path, strerror(errno)); ok = false; break; }
        uint64_t left = pf.bytes;
        while (left && ok) {
          const size_t want = (size_t)std::min<uint64_t>(left, buf.size());
          if (fread(buf.data(), 1, want, in) != want || fwrite(buf.data(), 1, want, out) != want) { db_err("mdb_merge_parts: copy in '%s': %s", path, strerror(errno)); ok = false; }
          left -= want;
        }
        fclose(in);
        unlink(pname.c_str());
      }
      for (FileIndexEntry e : pf.idx) { e.position += w->bytes[ff]; w->index[ff].push_back(e); }
      w->bytes[ff] += pf.bytes;
    }
    if (out && fclose(out) != 0) ok = false;
  }
  if (!ok) { delete w; return MGC_EINVAL; }
  for (uint32_t p = 0; p < n_parts; p++) unlink(part_meta_name(dir, p).c_str());
  // files no part contributed to get their empty blocks, then the indexes: the plain writer's close
  for (uint32_t ff = 0; ff < MGC_NUM_FILES; ff++)
    if (!w->index[ff].empty()) {                          // reopen for the close path's bookkeeping (nothing more is written)
      w->dat[ff] = open(block_name(dir, ff, false).c_str(), O_WRONLY);
      if (w->dat[ff] < 0) { db_err("mdb_merge_parts: reopen in '%s': %s", path, strerror(errno)); delete w; return MGC_EINVAL; }
    }
  return mdb_writer_close(w);
}

// ---------------------------------------------------------------------------
// reader (what `meryl print` / dumpIndex / dumpFile need; used by the tests to round-trip)
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
  if (br.bad || m1 != MAGIC_IDX1 || (m2 & 0x0000ffffffffffffull) != (MAGIC_IDX2 & 0x0000ffffffffffffull)) {
    db_err("mdb_reader_open: bad magic in '%s/merylIndex'", path);
    return nullptr;
  }
  mdb_info info;
  memset(&info, 0, sizeof(info));
  info.prefix_size = (uint32_t)br.get(32);
  info.suffix_size = (uint32_t)br.get(32);
  info.num_files_bits = (uint32_t)br.get(32);
  info.num_blocks_bits = (uint32_t)br.get(32);
  info.flags = (uint32_t)br.get(32);
  info.label_size = (info.flags >> 8) & 0xffu;
  info.k = (info.prefix_size + info.suffix_size) / 2;
  info.num_unique = br.get(64);
  info.num_distinct = br.get(64);
  info.num_total = br.get(64);
  info.hist_len = br.get(64);
  // the sizes come from the file: check them before anything is allocated from them
  if (br.bad || info.num_files_bits != MGC_NUM_FILES_BITS || info.prefix_size < MGC_NUM_FILES_BITS ||
      info.num_blocks_bits != info.prefix_size - MGC_NUM_FILES_BITS || info.num_blocks_bits > 40 ||
      ((info.prefix_size + info.suffix_size) & 1u) || info.k == 0 || info.k > 64 || info.label_size > 64 ||
      info.hist_len > (br.nbits - br.pos) / 128) {
    db_err("mdb_reader_open: '%s/merylIndex' holds impossible parameters (not a database this reader understands)", path);
    return nullptr;
  }
  mdb_reader *r = new mdb_reader();
  r->dir = path;
  r->info = info;
  r->hist_v.reserve(info.hist_len); r->hist_n.reserve(info.hist_len);
  for (uint64_t i = 0; i < info.hist_len; i++) { r->hist_v.push_back(br.get(64)); r->hist_n.push_back(br.get(64)); }
  if (br.bad) { db_err("mdb_reader_open: truncated histogram in '%s/merylIndex'", path); delete r; return nullptr; }
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

namespace {
bool load_file_index(mdb_reader *r, uint32_t ff, std::vector<FileIndexEntry> &idx) {
  const uint64_t nblocks = 1ull << r->info.num_blocks_bits;
  // the header's block count is not trusted before the index file has been seen to hold that many entries (a hostile
  // master index could otherwise ask for 2^40 x 24 bytes: std::bad_alloc across the extern "C" boundary)
  struct stat ist;
  if (stat(block_name(r->dir, ff, true).c_str(), &ist) != 0 || (uint64_t)ist.st_size < nblocks * sizeof(FileIndexEntry)) {
    db_err("read_file: '%s' is missing or shorter than the blocks the master index names", block_name(r->dir, ff, true).c_str());
    return false;
  }
  idx.assign(nblocks, FileIndexEntry());
  FILE *fi = fopen(block_name(r->dir, ff, true).c_str(), "rb");
  const bool ok = fi && fread(idx.data(), sizeof(FileIndexEntry), nblocks, fi) == nblocks;
  if (fi) fclose(fi);
  if (!ok) db_err("read_file: bad index in '%s'", r->dir.c_str());
  return ok;
}

struct DecodedHeader { mdb_block_header h; bool ok; };
DecodedHeader decode_header(BitReader &br, uint32_t suffix_size) {
  DecodedHeader d;
  memset(&d.h, 0, sizeof(d.h));
  const uint64_t m1 = br.get(64), m2 = br.get(64);
  d.h.prefix = br.get(64); d.h.n_kmers = br.get(64);
  d.h.k_code = (uint32_t)br.get(8);
  d.h.unary_bits = (uint32_t)br.get(32); d.h.binary_bits = (uint32_t)br.get(32);
  d.h.k1 = br.get(64); d.h.c_code = (uint32_t)br.get(8); d.h.c1 = br.get(64); d.h.c2 = br.get(64);
  d.ok = !br.bad && m1 == MAGIC_DAT1 && m2 == MAGIC_DAT2 && (uint64_t)d.h.unary_bits + d.h.binary_bits == suffix_size &&
         d.h.unary_bits <= 64;
  return d;
}
}  // namespace

extern "C" int mdb_reader_file_index(mdb_reader *r, uint32_t ff, mdb_index_entry *entries) {
  if (!r || ff >= MGC_NUM_FILES || !entries) return MGC_EINVAL;
  std::vector<FileIndexEntry> idx;
  if (!load_file_index(r, ff, idx)) return MGC_EINVAL;
  memcpy(entries, idx.data(), sizeof(FileIndexEntry) * idx.size());
  return MGC_OK;
}

extern "C" int mdb_reader_read_block_raw(mdb_reader *r, uint32_t ff, uint64_t position, mdb_block_header *h,
                                         uint64_t **prefix_delta, uint64_t **top, uint64_t **rem_hi, uint64_t **rem_lo,
                                         uint32_t **values) {
  if (!r || ff >= MGC_NUM_FILES || !h) return MGC_EINVAL;
  if (prefix_delta) *prefix_delta = nullptr;
  if (top) *top = nullptr;
  if (rem_hi) *rem_hi = nullptr;
  if (rem_lo) *rem_lo = nullptr;
  if (values) *values = nullptr;
  FILE *fd = fopen(block_name(r->dir, ff, false).c_str(), "rb");
  if (!fd) { db_err("read_block: cannot open data file in '%s'", r->dir.c_str()); return MGC_EINVAL; }
  std::vector<uint64_t> words; uint64_t nbits = 0;
  const bool loaded = fseek(fd, (long)position, SEEK_SET) == 0 && load_stuffed(fd, words, nbits);
  fclose(fd);
  if (!loaded) { db_err("read_block: no block at that position in '%s'", r->dir.c_str()); return MGC_EINVAL; }
  BitReader br; br.w = words.data(); br.nbits = nbits;
  DecodedHeader d = decode_header(br, r->info.suffix_size);
  // every k-mer takes at least one bit: n_kmers beyond the block's bits is corruption, not an allocation size
  if (!d.ok || d.h.n_kmers > nbits) { db_err("read_block: corrupt block header in '%s'", r->dir.c_str()); return MGC_EINVAL; }
  *h = d.h;
  const bool want_kmers = prefix_delta || top || rem_hi || rem_lo || values;
  if (!want_kmers) return MGC_OK;
  const uint64_t n = d.h.n_kmers, na = n ? n : 1;
  uint64_t *pd = (uint64_t *)malloc(8 * na), *tp = (uint64_t *)malloc(8 * na), *rh = (uint64_t *)malloc(8 * na), *rl = (uint64_t *)malloc(8 * na);
  uint32_t *vv = (uint32_t *)malloc(4 * na);
  bool ok = pd && tp && rh && rl && vv;
  const uint32_t bb = d.h.binary_bits;
  uint64_t acc = 0;
  for (uint64_t i = 0; i < n && ok; i++) {
    const uint64_t dl = br.get_unary();
    acc += dl;
    pd[i] = dl; tp[i] = acc;
    if (bb <= 64) { rh[i] = 0; rl[i] = br.get(bb); } else { rh[i] = br.get(bb - 64); rl[i] = br.get(64); }
  }
  for (uint64_t i = 0; i < n && ok; i++) vv[i] = (uint32_t)br.get(VALUE_BITS);
  if (br.bad) ok = false;
  if (!ok) { free(pd); free(tp); free(rh); free(rl); free(vv); db_err("read_block: corrupt block in '%s'", r->dir.c_str()); return MGC_EINVAL; }
  if (prefix_delta) *prefix_delta = pd; else free(pd);
  if (top) *top = tp; else free(tp);
  if (rem_hi) *rem_hi = rh; else free(rh);
  if (rem_lo) *rem_lo = rl; else free(rl);
  if (values) *values = vv; else free(vv);
  return MGC_OK;
}

extern "C" int mdb_reader_block_header(mdb_reader *r, uint32_t ff, uint64_t position, mdb_block_header *h) {
  return mdb_reader_read_block_raw(r, ff, position, h, nullptr, nullptr, nullptr, nullptr, nullptr);
}

namespace {
// One stuffedBits object that lies in memory (the whole data file is read at once): its logical bit stream as a word
// pointer.  The common case -- one sub-block, every data block below 16 MiB -- is decoded IN PLACE (the words are 8-byte aligned
// because every field of the layout is a multiple of 8 bytes); a longer object is gathered into `gather`.  `avail` = bytes from
// `at` to the end of the buffer (which carries 16 bytes of slack for two-word reads).
bool stuffed_in_memory(const unsigned char *at, uint64_t avail, std::vector<uint64_t> &gather, const uint64_t **words, uint64_t *nbits) {
  if (avail < 16) return false;
  uint64_t lenmax; uint32_t nblocks;
  memcpy(&lenmax, at, 8); memcpy(&nblocks, at + 8, 4);
  if (nblocks == 0 || nblocks > (1u << 20) || lenmax == 0 || lenmax > (1ull << 40)) return false;
  const uint64_t head = 16 + 16ull * nblocks;
  if (avail < head + 16) return false;
  const unsigned char *bgn = at + 16, *len = at + 16 + 8ull * nblocks, *q = at + head;
  uint64_t total = 0;
  if (nblocks > 1) gather.clear();
  for (uint32_t i = 0; i < nblocks; i++) {
    uint64_t b, l, nw;
    memcpy(&b, bgn + 8ull * i, 8); memcpy(&l, len + 8ull * i, 8);
    if ((uint64_t)(q - at) + 16 > avail) return false;
    memcpy(&nw, q, 8);
    q += 16;
    if (b % 64 != 0 || b != total || l > lenmax || nw != (l + 63) / 64) return false;    // only what dump_stuffed writes
    if ((uint64_t)(q - at) + 8 * nw > avail) return false;
    if (nblocks == 1) *words = reinterpret_cast<const uint64_t *>(q);
    else gather.insert(gather.end(), reinterpret_cast<const uint64_t *>(q), reinterpret_cast<const uint64_t *>(q) + nw);
    q += 8 * nw;
    total += l;
    if (i + 1 < nblocks && l % 64 != 0) return false;
  }
  if (nblocks > 1) { gather.push_back(0); gather.push_back(0); *words = gather.data(); }
  *nbits = total;
  return true;
}

// MSB-first bit cursor without per-call bookkeeping: the caller checks `pos` against the stream length once per k-mer
struct FastBits {
  const uint64_t *w; uint64_t pos;
  inline uint64_t peek() const {
    const uint64_t i = pos >> 6; const uint32_t off = (uint32_t)(pos & 63);
    return off ? ((w[i] << off) | (w[i + 1] >> (64 - off))) : w[i];
  }
  inline uint64_t get(uint32_t width) { const uint64_t v = peek() >> (64 - width); pos += width; return v; }   // width 1..64
  inline uint64_t unary(uint64_t limit) {            // zeros before the next one bit; pos = limit + 1 when the stream runs out first
    uint64_t v = 0;
    for (;;) {
      const uint64_t x = peek();
      if (x) { const uint32_t lz = (uint32_t)__builtin_clzll(x); pos += lz + 1; return v + lz; }
      v += 64; pos += 64;
      if (pos > limit) { pos = limit + 1; return v; }
    }
  }
};
}  // namespace

extern "C" int mdb_reader_read_file_ex(mdb_reader *r, uint32_t ff, uint64_t **klo, uint64_t **khi, uint32_t **cnt,
                                       uint64_t **labels, uint64_t *n_out) {
  if (!r || ff >= MGC_NUM_FILES || !klo || !cnt || !n_out) return MGC_EINVAL;
  *klo = nullptr; if (khi) *khi = nullptr; *cnt = nullptr; if (labels) *labels = nullptr; *n_out = 0;
  const uint64_t nblocks = 1ull << r->info.num_blocks_bits;
  std::vector<FileIndexEntry> idx;
  if (!load_file_index(r, ff, idx)) return MGC_EINVAL;
  // the whole data file in one read (a file of the 10 Gbp database is ~100 MB); blocks are decoded where they lie
  const int fd = open(block_name(r->dir, ff, false).c_str(), O_RDONLY);
  if (fd < 0) { db_err("read_file: cannot open data file in '%s'", r->dir.c_str()); return MGC_EINVAL; }
  struct stat st;
  bool ok = fstat(fd, &st) == 0;
  const uint64_t fsize = ok ? (uint64_t)st.st_size : 0;
  // the index is input too: its k-mer total cannot exceed what the data file could hold at one bit per k-mer
  uint64_t total = 0;
  for (auto &e : idx) { if (e.n_kmers > fsize * 8) ok = false; total += e.n_kmers; }
  if (!ok || total > fsize * 8) { close(fd); db_err("read_file: index of '%s' does not fit its data file", r->dir.c_str()); return MGC_EINVAL; }
  unsigned char *file = (unsigned char *)malloc(fsize + 16);
  if (!file) { close(fd); db_err("read_file: out of memory for '%s'", r->dir.c_str()); return MGC_ENOMEM; }
  for (uint64_t got = 0; got < fsize && ok;) {
    const ssize_t n = pread(fd, file + got, fsize - got, (off_t)got);
    if (n < 0 && errno == EINTR) continue;
    if (n <= 0) ok = false; else got += (uint64_t)n;
  }
  close(fd);
  memset(file + fsize, 0, 16);
  const uint32_t ss = r->info.suffix_size, ls = r->info.label_size;
  const bool wide = 2 * r->info.k > 64;
  uint64_t *lo = (uint64_t *)malloc(8 * (total ? total : 1)), *hi = (uint64_t *)calloc(total ? total : 1, 8);
  uint64_t *lb = (uint64_t *)calloc(total ? total : 1, 8);           // untouched (lazily zero) pages unless the database holds labels / k > 32
  uint32_t *cn = (uint32_t *)malloc(4 * (total ? total : 1));
  if (!ok || !lo || !hi || !cn || !lb) {
    free(file); free(lo); free(hi); free(cn); free(lb);
    db_err(ok ? "read_file: out of memory for '%s'" : "read_file: cannot read the data file in '%s'", r->dir.c_str());
    return ok ? MGC_ENOMEM : MGC_EINVAL;
  }
  uint64_t o = 0;
  std::vector<uint64_t> gather;
  for (uint64_t bb = 0; bb < nblocks && ok; bb++) {
    const FileIndexEntry &e = idx[bb];
    // a slot the writer was never handed a block for (mdb_writer_add_block lets callers skip empty prefixes; write_indexes
    // then leaves {prefix, 0, 0}): nothing to decode -- run_dump_file skips it the same way
    if (e.n_kmers == 0) continue;
    const uint64_t *words = nullptr; uint64_t nbits = 0;
    if (e.position >= fsize || (e.position & 7) || !stuffed_in_memory(file + e.position, fsize + 16 - e.position, gather, &words, &nbits)) { ok = false; break; }
    BitReader br; br.w = words; br.nbits = nbits;
    DecodedHeader d = decode_header(br, ss);
    const uint64_t prefix = d.h.prefix, n = d.h.n_kmers;
    const uint32_t bb_bits = d.h.binary_bits;
    if (!d.ok || n != e.n_kmers || prefix != e.prefix || o + n > total) { ok = false; break; }
    FastBits fb{words, br.pos};
    uint64_t top = 0;
    if (!wide && bb_bits >= 1 && bb_bits <= 64) {                        // k <= 32: the k-mer is one word
      const uint64_t pre = ss < 64 ? prefix << ss : 0;
      for (uint64_t i = 0; i < n; i++) {
        top += fb.unary(nbits);
        if (fb.pos + bb_bits > nbits) { ok = false; break; }
        lo[o + i] = pre | (bb_bits < 64 ? top << bb_bits : 0) | fb.get(bb_bits);
      }
    } else {
      for (uint64_t i = 0; i < n; i++) {
        top += fb.unary(nbits);
        if (fb.pos + bb_bits > nbits) { ok = false; break; }
        // suffix = top << bb_bits | binary ; full k-mer = prefix << ss | suffix  (128-bit)
        unsigned __int128 suf = (unsigned __int128)top << bb_bits;
        if (bb_bits > 64) { const uint64_t h = fb.get(bb_bits - 64), l = fb.get(64); suf |= ((unsigned __int128)h << 64) | l; }
        else if (bb_bits) suf |= fb.get(bb_bits);
        const unsigned __int128 full = ((unsigned __int128)prefix << ss) | suf;
        lo[o + i] = (uint64_t)full;
        hi[o + i] = (uint64_t)(full >> 64);
      }
    }
    if (!ok || fb.pos + n * (uint64_t)(VALUE_BITS + ls) > nbits) { ok = false; break; }
    for (uint64_t i = 0; i < n; i++) cn[o + i] = (uint32_t)fb.get(VALUE_BITS);
    if (ls) for (uint64_t i = 0; i < n; i++) lb[o + i] = fb.get(ls);
    o += n;
  }
  free(file);
  if (!ok || o != total) { free(lo); free(hi); free(cn); free(lb); db_err("read_file: corrupt block in '%s'", r->dir.c_str()); return MGC_EINVAL; }
  *klo = lo; if (khi) *khi = hi; else free(hi); *cnt = cn; if (labels) *labels = lb; else free(lb); *n_out = total;
  return MGC_OK;
}

extern "C" int mdb_reader_raw_file(mdb_reader *r, uint32_t ff, unsigned char **bytes, uint64_t *size, mdb_raw_block **blocks,
                                   uint64_t *n_blocks, uint64_t *n_kmers) {
  if (!r || ff >= MGC_NUM_FILES || !bytes || !size || !blocks || !n_blocks || !n_kmers) return MGC_EINVAL;
  *bytes = nullptr; *blocks = nullptr; *size = 0; *n_blocks = 0; *n_kmers = 0;
  std::vector<FileIndexEntry> idx;
  if (!load_file_index(r, ff, idx)) return MGC_EINVAL;
  const std::string name = block_name(r->dir, ff, false);
  struct stat st;
  if (stat(name.c_str(), &st) != 0) { db_err("read_file: cannot stat '%s'", name.c_str()); return MGC_EINVAL; }
  const uint64_t fsize = (uint64_t)st.st_size;
  unsigned char *file = (unsigned char *)malloc(fsize + 16);
  if (!file) { db_err("read_file: out of memory for '%s'", name.c_str()); return MGC_ENOMEM; }
  memset(file + fsize, 0, 16);
  {
    const int fd = open(name.c_str(), O_RDONLY);
    bool ok = fd >= 0;
    uint64_t have = 0;
    while (ok && have < fsize) {
      const ssize_t got = pread(fd, file + have, (size_t)std::min<uint64_t>(fsize - have, 1ull << 30), (off_t)have);
      if (got <= 0) ok = false; else have += (uint64_t)got;
    }
    if (fd >= 0) close(fd);
    if (!ok) { free(file); db_err("read_file: cannot read '%s'", name.c_str()); return MGC_EINVAL; }
  }
  uint64_t nb = 0, total = 0;
  for (const FileIndexEntry &e : idx) if (e.n_kmers) nb++;
  mdb_raw_block *out = (mdb_raw_block *)malloc(sizeof(mdb_raw_block) * (nb ? nb : 1));
  if (!out) { free(file); db_err("read_file: out of memory for '%s'", name.c_str()); return MGC_ENOMEM; }
  int rc = MGC_OK;
  uint64_t j = 0;
  for (const FileIndexEntry &e : idx) {
    if (!e.n_kmers) continue;
    // the framing of the object (A2), checked the way stuffed_in_memory checks it -- but only the canonical shape (every
    // sub-block but the last full) is handed on: then word w of the bit stream sits at stuffed_word_offset(nsb, w)
    bool ok = e.position < fsize && !(e.position & 7) && fsize - e.position >= 48 && e.n_kmers <= fsize * 8;
    uint64_t lenmax = 0, nbits = 0; uint32_t nsb = 0;
    if (ok) {
      memcpy(&lenmax, file + e.position, 8); memcpy(&nsb, file + e.position + 8, 4);
      ok = nsb >= 1 && nsb <= (1u << 20) && e.position + 16 + 32ull * nsb <= fsize;
    }
    if (ok && lenmax != STUFFED_BLOCK_BITS) { rc = MGC_EUNSUPPORTED; break; }
    for (uint32_t i = 0; i < nsb && ok; i++) {
      uint64_t b, l, nw;
      memcpy(&b, file + e.position + 16 + 8ull * i, 8); memcpy(&l, file + e.position + 16 + 8ull * nsb + 8ull * i, 8);
      const uint64_t hdr = e.position + stuffed_word_offset(nsb, (uint64_t)i * STUFFED_BLOCK_WORDS) - 16;
      ok = hdr + 16 <= fsize;
      if (ok) { memcpy(&nw, file + hdr, 8); ok = b == (uint64_t)i * STUFFED_BLOCK_BITS && nw == (l + 63) / 64 && l <= STUFFED_BLOCK_BITS && hdr + 16 + 8 * nw <= fsize; }
      if (ok && i + 1 < nsb && l != STUFFED_BLOCK_BITS) { rc = MGC_EUNSUPPORTED; break; }
      nbits += l;
    }
    if (rc != MGC_OK) break;
    if (!ok || total + e.n_kmers < total) { rc = MGC_EINVAL; db_err("read_file: corrupt block in '%s'", r->dir.c_str()); break; }
    out[j].object_offset = e.position; out[j].n_bits = nbits; out[j].n_sub_blocks = nsb; out[j].n_kmers = e.n_kmers; out[j].prefix = e.prefix;
    out[j].out_offset = total;
    total += e.n_kmers;
    j++;
  }
  if (rc != MGC_OK) { free(file); free(out); return rc; }
  *bytes = file; *size = fsize; *blocks = out; *n_blocks = nb; *n_kmers = total;
  return MGC_OK;
}

extern "C" int mdb_reader_read_file(mdb_reader *r, uint32_t ff, uint64_t **klo, uint64_t **khi, uint32_t **cnt,
                                    uint64_t *n_out) {
  return mdb_reader_read_file_ex(r, ff, klo, khi, cnt, nullptr, n_out);
}

extern "C" void mdb_reader_close(mdb_reader *r) { delete r; }
extern "C" void mdb_free(void *p) { free(p); }

