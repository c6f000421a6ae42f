// meryl_seq.cpp -- FASTA/FASTQ/SAM/BAM (plain, gzip or BGZF) loader, include/meryl_seq.h.
//
// Stands in for dnaSeqFile::loadBases / openSequenceFile of the absent
// meryl-utility submodule (call sites src/meryl/merylOp.C:200,
// src/meryl/merylInput.C:257); contract per src/meryl/merylInput.H:67-70.
// Bytes come from zlib's gzread (transparent for uncompressed files) or, when
// the file is BGZF (bgzip'd FASTA/FASTQ, every BAM), from a block-parallel
// inflater: BGZF blocks are independent deflate streams whose compressed size
// sits in their gzip extra field, so a batch of them is inflated by several
// threads at once -- the single zlib stream is what bounds a .gz input
// (DESIGN.md 8).
// Parsing rules: '>' starts a FASTA record (header to end of
// line, then sequence lines until the next '>' at a line start); '@' starts a
// FASTQ record (header line, sequence lines up to the '+' line, then as many
// quality characters as there were bases).  White space inside sequence
// lines is dropped; every other byte is handed on as is (the k-mer packer
// decides what is a base).
// BAM (SAMv1 4.2: magic, header text, reference list, then records
// block_size | 32 fixed bytes | read_name | cigar | 4-bit seq | qual | aux) and
// SAM (tab-separated, '@' header lines, SEQ is column 10) give one sequence per
// alignment record: SEQ exactly as stored, '=' and the IUPAC codes included
// (they break k-mers like N does), a '*' / zero-length SEQ as an empty sequence.
// No record is filtered by its flags -- the reference reads these files through
// its vendored htslib inside the absent submodule (src/main.mk:92-140), whose
// call sites are not in the tree, so which records it keeps is unpinned here.
#include "../../include/meryl_seq.h"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <sys/stat.h>
#include <zlib.h>

namespace {
thread_local std::string g_seq_error;
void seq_err(const std::string &m) { g_seq_error = m; }
bool ends_with(const std::string &s, const char *suf) {
  const size_t n = strlen(suf);
  return s.size() > n && s.compare(s.size() - n, n, suf) == 0;
}
}  // namespace

// ---------------------------------------------------------------------------
// BGZF: gzip members of at most 64 KiB with the compressed size in a 'BC' extra subfield (SAMv1 4.1)
// ---------------------------------------------------------------------------
namespace {
// total size of the block starting at p (n >= 18 bytes available), 0 if it is not a BGZF block
size_t bgzf_block_size(const unsigned char *p, size_t n) {
  if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
  const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8);
  if (n < 12 + xlen) return 0;
  for (size_t o = 12; o + 4 <= 12 + xlen;) {
    const size_t slen = (size_t)p[o + 2] | ((size_t)p[o + 3] << 8);
    if (p[o] == 'B' && p[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + xlen)
      return ((size_t)p[o + 4] | ((size_t)p[o + 5] << 8)) + 1;
    o += 4 + slen;
  }
  return 0;
}

struct BgzfSource {
  FILE *f = nullptr;
  std::vector<unsigned char> raw;          // compressed bytes not yet decoded
  size_t raw_pos = 0;
  bool   raw_eof = false;
  std::vector<unsigned char> out;          // decoded bytes of the current batch
  size_t out_pos = 0;
  unsigned threads = 1;
  std::string err;

  struct Block { size_t src, csize, hdr, isize, dst; };

  bool fill_raw(size_t want) {             // make at least `want` bytes available at raw_pos (if the file has them)
    if (raw.size() - raw_pos >= want || raw_eof) return raw.size() - raw_pos >= want;
    raw.erase(raw.begin(), raw.begin() + (long)raw_pos);
    raw_pos = 0;
    const size_t chunk = 32u << 20;
    while (raw.size() < want + chunk / 2 && !raw_eof) {
      const size_t old = raw.size();
      raw.resize(old + chunk);
      const size_t got = fread(raw.data() + old, 1, chunk, f);
      raw.resize(old + got);
      if (got < chunk) raw_eof = true;
    }
    return raw.size() - raw_pos >= want;
  }

  // decode the next batch of blocks into `out`; false at end of file or on error (err set)
  bool next_batch(std::vector<unsigned char> &out) {
    out.clear();
    std::vector<Block> blocks;
    size_t total = 0;
    while (blocks.size() < 512) {                     // <= 32 MiB of text per batch: enough for 16 threads, small enough to pipeline
      if (!fill_raw(18)) {
        if (raw.size() != raw_pos) { err = "truncated BGZF block header"; return false; }
        break;
      }
      const size_t bs = bgzf_block_size(raw.data() + raw_pos, raw.size() - raw_pos);
      if (bs == 0) { err = "not a BGZF block (plain gzip data inside a BGZF file?)"; return false; }
      if (!fill_raw(bs)) { err = "truncated BGZF block"; return false; }
      const unsigned char *p = raw.data() + raw_pos;
      const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8), hdr = 12 + xlen;
      if (bs < hdr + 8) { err = "corrupt BGZF block"; return false; }
      const size_t isize = (size_t)p[bs - 4] | ((size_t)p[bs - 3] << 8) | ((size_t)p[bs - 2] << 16) | ((size_t)p[bs - 1] << 24);
      if (isize > 65536) { err = "corrupt BGZF block (ISIZE)"; return false; }
      blocks.push_back({raw_pos, bs, hdr, isize, total});
      total += isize;
      raw_pos += bs;
      // fill_raw may move `raw`: the blocks of one batch must stay inside one buffer generation
      if (raw.size() - raw_pos < 65536 + 18 && !raw_eof) break;
    }
    if (blocks.empty()) return false;
    out.resize(total);
    std::vector<int> bad(threads, 0);
    auto work = [&](unsigned t) {
      z_stream z;
      memset(&z, 0, sizeof(z));
      if (inflateInit2(&z, -15) != Z_OK) { bad[t] = 1; return; }
      for (size_t i = t; i < blocks.size(); i += threads) {
        const Block &b = blocks[i];
        const unsigned char *p = raw.data() + b.src;
        if (b.isize == 0) continue;                  // the EOF marker (and any other empty block): nothing to produce
        inflateReset(&z);
        z.next_in = const_cast<unsigned char *>(p + b.hdr);
        z.avail_in = (unsigned)(b.csize - b.hdr - 8);
        z.next_out = out.data() + b.dst;
        z.avail_out = (unsigned)b.isize;
        const int rc = inflate(&z, Z_FINISH);
        const uint32_t want_crc = (uint32_t)p[b.csize - 8] | ((uint32_t)p[b.csize - 7] << 8) | ((uint32_t)p[b.csize - 6] << 16) |
                                  ((uint32_t)p[b.csize - 5] << 24);
        if (rc != Z_STREAM_END || z.avail_out != 0 || (uint32_t)crc32(0L, out.data() + b.dst, (unsigned)b.isize) != want_crc) bad[t] = 1;
      }
      inflateEnd(&z);
    };
    const unsigned nt = (unsigned)std::min<size_t>(threads, blocks.size());
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; t++) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    for (int b : bad) if (b) { err = "BGZF block failed to inflate (corrupt file)"; return false; }
    return true;
  }

  // The batch after the one being consumed is inflated in the background: the caller's work on a batch (device upload,
  // BAM decode) overlaps the next one's inflate.
  std::vector<unsigned char> ahead;
  std::future<bool> pending;
  bool started = false, finished = false;

  bool advance_batch() {                             // -> out = next batch; false at end of file or on error
    if (finished) return false;
    if (!started) { started = true; pending = std::async(std::launch::async, [this] { return next_batch(ahead); }); }
    const bool ok = pending.get();
    if (!ok) { finished = true; out.clear(); out_pos = 0; return false; }
    out.swap(ahead);
    out_pos = 0;
    pending = std::async(std::launch::async, [this] { return next_batch(ahead); });
    return true;
  }
  ~BgzfSource() { if (pending.valid()) pending.wait(); }

  // up to n bytes; 0 at end of file, -1 on error
  int64_t read(void *dst, size_t n) {
    size_t got = 0;
    while (got < n) {
      if (out_pos == out.size()) {
        if (!advance_batch()) return got ? (int64_t)got : (err.empty() ? 0 : -1);   // what was read is delivered first
        if (out.empty()) continue;                 // a batch of empty blocks (the EOF marker)
      }
      const size_t take = std::min(n - got, out.size() - out_pos);
      memcpy((unsigned char *)dst + got, out.data() + out_pos, take);
      out_pos += take; got += take;
    }
    return (int64_t)got;
  }
};
}  // namespace

struct msr_reader {
  gzFile      gz = nullptr;
  BgzfSource *bgzf = nullptr;
  FILE       *pipe = nullptr;        // bzip2 -dc / xz -dc (the reference's compressed-file reader pipes them too)
  std::string name;
  bool        compressed = false;
  unsigned char *buf = nullptr;
  size_t      cap = 1u << 20, len = 0, pos = 0;
  bool        eof = false;
  int         format = MSR_FORMAT_FASTX;

  // BAM / SAM record state
  bool        bam_header_done = false;
  uint64_t    rec_seq_left = 0;      // bases of the current record not yet handed out
  uint64_t    rec_seq_index = 0;     // BAM: index of the next base in the 4-bit array
  int         rec_byte = 0;          // BAM: the byte holding the current pair of bases
  uint64_t    rec_skip = 0;          // bytes of the record after its sequence
  int         sam_field = 0;         // SAM: column of the next byte (0-based)

  bool        failed = false;        // the byte source reported an error (as opposed to its end)
  std::string failure;
  bool        pipe_done = false;
  bool        pipe_is_cram = false;  // the pipe is `samtools view` (its failures are mostly a missing reference, not a damaged file)

  // up to n bytes; 0 at the end of the input, -1 on a read / decompression error (failure says which)
  int64_t source_read(void *dst, size_t n) {
    if (bgzf) {
      const int64_t got = bgzf->read(dst, n);
      if (got < 0) { failed = true; failure = bgzf->err; }
      return got;
    }
    if (pipe_done) return 0;
    if (pipe) {
      const size_t got = fread(dst, 1, n, pipe);
      if (got) return (int64_t)got;
      const bool rd_err = ferror(pipe) != 0;
      const int status = pclose(pipe);               // the decompressor's verdict on the file
      pipe = nullptr; pipe_done = true;
      if (rd_err || status != 0) {
        failed = true;
        failure = pipe_is_cram ? "`samtools view` failed (reference not found?  set REF_PATH / REF_CACHE, or convert with `samtools view -b -T ref.fa`; corrupt or truncated file?)"
                               : "the decompressor failed (corrupt or truncated file)";
        return -1;
      }
      return 0;
    }
    const int got = gzread(gz, dst, (unsigned)(n > (1u << 30) ? (1u << 30) : n));
    if (got <= 0) {                                  // end of file, or a stream that ended early / is damaged
      int en = Z_OK;
      const char *msg = gzerror(gz, &en);
      if (got < 0 || (en != Z_OK && en != Z_STREAM_END)) {
        failed = true; failure = std::string("zlib: ") + (msg && *msg ? msg : "read error");
        return -1;
      }
    }
    return got;
  }

  enum State { AT_RECORD_START, IN_HEADER, IN_FASTA_SEQ, IN_FASTQ_SEQ, IN_FASTQ_PLUS, IN_FASTQ_QUAL } st = AT_RECORD_START;
  bool        line_start = true;     // next byte is the first of a line
  uint64_t    seq_bases = 0;         // bases of the current FASTQ record (to skip as many qualities)
  uint64_t    qual_left = 0;
  bool        fastq = false;
  bool        in_sequence = false;   // a sequence has been started and not yet reported as ended

  int peek() {
    if (pos == len) {
      if (eof) return -1;
      const int64_t n = source_read(buf, cap);
      if (n <= 0) { eof = true; return -1; }
      len = (size_t)n; pos = 0;
    }
    return buf[pos];
  }
  void advance() { pos++; }
  // n binary bytes (BAM); false at end of input
  bool get_bytes(void *dst, size_t n) {
    unsigned char *d = (unsigned char *)dst;
    while (n) {
      if (peek() < 0) return false;
      const size_t take = std::min(n, len - pos);
      if (d) { memcpy(d, buf + pos, take); d += take; }
      pos += take; n -= take;
    }
    return true;
  }
  bool get_u32(uint32_t *v) {
    unsigned char b[4];
    if (!get_bytes(b, 4)) return false;
    *v = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
    return true;
  }
};

extern "C" const char *msr_last_error(void) { return g_seq_error.c_str(); }

extern "C" msr_reader *msr_open(const char *name) {
  if (!name || !*name) { seq_err("msr_open: empty file name"); return nullptr; }
  const std::string n(name);
  // CRAM (README.md:11 of the reference names it; its htslib call sites are in the absent submodule): decoded by `samtools view`
  // when that binary is on the PATH -- the records arrive as SAM text and take the SAM path (SEQ of every record as stored) --
  // and refused with a message otherwise
  const bool cram = ends_with(n, ".cram");
  auto have_samtools = []() -> bool {                      // probed only when a CRAM file is opened, and again only if PATH changed
    static std::mutex mu;
    static std::string probed_path;
    static bool probed = false, have = false;
    std::lock_guard<std::mutex> lk(mu);
    const char *pe = getenv("PATH");
    const std::string path = pe ? pe : "";
    if (!probed || path != probed_path) { have = system("command -v samtools > /dev/null 2>&1") == 0; probed = true; probed_path = path; }
    return have;
  };
  if (cram && !have_samtools()) {
    seq_err("msr_open: '" + n + "': CRAM input is not supported without `samtools` on the PATH (it is decoded through `samtools view`); convert to BAM");
    return nullptr;
  }
  msr_reader *r = new msr_reader();
  r->name = n;
  r->compressed = ends_with(n, ".gz") || ends_with(n, ".bam") || ends_with(n, ".bz2") || ends_with(n, ".xz") || cram;
  r->buf = (unsigned char *)malloc(r->cap);
  if (ends_with(n, ".bz2") || ends_with(n, ".xz") || cram) {  // through the system's decompressor / decoder
    struct stat st;
    if (stat(name, &st) != 0) { seq_err("msr_open: cannot open '" + n + "': " + strerror(errno)); msr_close(r); return nullptr; }
    std::string quoted = "'";
    for (char c : n) { if (c == '\'') quoted += "'\\''"; else quoted += c; }
    quoted += "'";
    const std::string cmd = cram ? "samtools view -h -- " + quoted
                                 : std::string(ends_with(n, ".bz2") ? "bzip2" : "xz") + " -dc -- " + quoted;
    r->pipe = popen(cmd.c_str(), "r");
    r->pipe_is_cram = cram;
    if (!r->pipe) { seq_err("msr_open: cannot run '" + cmd + "': " + strerror(errno)); msr_close(r); return nullptr; }
  }
  else if (n != "-") {                                   // BGZF (bgzip, BAM)?  then the blocks are inflated in parallel
    FILE *f = fopen(name, "rb");
    if (!f) { seq_err("msr_open: cannot open '" + n + "': " + strerror(errno)); msr_close(r); return nullptr; }
    unsigned char head[64];
    const size_t got = fread(head, 1, sizeof(head), f);
    const char *off = getenv("MERYL_BGZF_THREADS");
    const unsigned want = off ? (unsigned)atoi(off) : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (bgzf_block_size(head, got) != 0 && want >= 1) {
      rewind(f);
      r->bgzf = new BgzfSource();
      r->bgzf->f = f;
      r->bgzf->threads = want;
    } else {
      fclose(f);
    }
  }
  if (!r->bgzf && !r->pipe) {
    r->gz = (n == "-") ? gzdopen(0, "rb") : gzopen(name, "rb");
    if (!r->gz) { seq_err("msr_open: cannot open '" + n + "': " + strerror(errno)); msr_close(r); return nullptr; }
    gzbuffer(r->gz, 1u << 20);
  }
  // what is in it: BAM by its magic, SAM by its name or an @HD line, everything else is FASTA/FASTQ text
  r->peek();
  if (r->failed) { seq_err("msr_open: '" + n + "': " + r->failure); msr_close(r); return nullptr; }
  const size_t have = r->len - r->pos;
  if (have >= 4 && memcmp(r->buf + r->pos, "BAM\1", 4) == 0) r->format = MSR_FORMAT_BAM;
  else if (cram || ends_with(n, ".sam") || ends_with(n, ".sam.gz") || (have >= 4 && memcmp(r->buf + r->pos, "@HD\t", 4) == 0)) r->format = MSR_FORMAT_SAM;
  else if (ends_with(n, ".bam")) { seq_err("msr_open: '" + n + "' is not a BAM file (no BAM magic)"); msr_close(r); return nullptr; }
  return r;
}

extern "C" void msr_close(msr_reader *r) {
  if (!r) return;
  if (r->gz) gzclose(r->gz);
  if (r->pipe) pclose(r->pipe);
  if (r->bgzf) { FILE *f = r->bgzf->f; delete r->bgzf; if (f) fclose(f); }     // the destructor waits for the batch in flight
  free(r->buf);
  delete r;
}

extern "C" int msr_format(const msr_reader *r) { return r ? r->format : -1; }

extern "C" int64_t msr_read_text(msr_reader *r, char *buf, uint64_t max_length) {
  if (!r || !buf) return -1;
  if (r->format != MSR_FORMAT_FASTX) { seq_err("msr_read_text: '" + r->name + "' is SAM/BAM: use msr_load_bases"); return -1; }
  if (r->st != msr_reader::AT_RECORD_START || r->in_sequence) {
    seq_err("msr_read_text: '" + r->name + "': raw and parsed reads cannot be mixed on one reader");
    return -1;
  }
  uint64_t got = 0;
  if (r->pos < r->len) {                             // what msr_open looked at to tell the format
    got = std::min<uint64_t>(max_length, r->len - r->pos);
    memcpy(buf, r->buf + r->pos, got);
    r->pos += got;
  }
  while (got < max_length && !r->eof) {
    const int64_t n = r->source_read(buf + got, max_length - got);
    if (n < 0) { seq_err("msr_read_text: read error in '" + r->name + "': " + r->failure); return -1; }
    if (n == 0) { r->eof = true; break; }
    got += (uint64_t)n;
  }
  return (int64_t)got;
}

extern "C" int msr_is_compressed(const msr_reader *r) { return (r && r->compressed) ? 1 : 0; }

extern "C" uint64_t msr_guess_number_of_kmers(const char *name) {
  if (!name || (name[0] == '-' && name[1] == 0)) return 0;                // merylOp-count.C:416-417
  struct stat st;
  if (stat(name, &st) != 0) return 0;
  const std::string n(name);
  const uint64_t size = (uint64_t)st.st_size;
  if (ends_with(n, ".gz"))  return size * 3;                               // :421-422
  if (ends_with(n, ".bz2")) return (uint64_t)(size * 3.5);                 // :424-425
  if (ends_with(n, ".xz"))  return (uint64_t)(size * 4.0);                 // :427-428
  return size;                                                             // :430-431
}

namespace {
// BAM: one alignment record per call (or a max_length piece of its sequence)
int bam_load_bases(msr_reader *r, char *seq, uint64_t max_length, uint64_t *seq_length, int *end_of_sequence) {
  static const char nt16[] = "=ACMGRSVTWYHKDBN";                        // SAMv1 4.2.3
  auto bad = [&](const char *what) { seq_err("msr_load_bases: '" + r->name + "': " + what); return -2; };
  if (!r->bam_header_done) {
    uint32_t magic, l_text, n_ref;
    if (!r->get_u32(&magic) || !r->get_u32(&l_text) || !r->get_bytes(nullptr, l_text) || !r->get_u32(&n_ref))
      return bad("truncated BAM header");
    for (uint32_t i = 0; i < n_ref; i++) {
      uint32_t l_name, l_ref;
      if (!r->get_u32(&l_name) || !r->get_bytes(nullptr, l_name) || !r->get_u32(&l_ref)) return bad("truncated BAM reference list");
    }
    r->bam_header_done = true;
  }
  if (!r->in_sequence) {                             // start of a record
    if (r->peek() < 0) return r->failed ? bad(r->failure.c_str()) : 0;
    uint32_t block_size;
    unsigned char fix[32];
    if (!r->get_u32(&block_size) || block_size < 32 || !r->get_bytes(fix, 32)) return bad("truncated BAM record");
    const uint64_t l_read_name = fix[8];
    const uint64_t n_cigar = (uint64_t)fix[12] | ((uint64_t)fix[13] << 8);
    const uint64_t l_seq = (uint64_t)fix[16] | ((uint64_t)fix[17] << 8) | ((uint64_t)fix[18] << 16) | ((uint64_t)fix[19] << 24);
    const uint64_t before = 32 + l_read_name + 4 * n_cigar, packed = (l_seq + 1) / 2;
    if (before + packed > block_size) return bad("corrupt BAM record (fields longer than the record)");
    if (!r->get_bytes(nullptr, l_read_name + 4 * n_cigar)) return bad("truncated BAM record");
    r->rec_seq_left = l_seq;
    r->rec_seq_index = 0;
    r->rec_skip = block_size - before - packed;
    r->in_sequence = true;
  }
  uint64_t out = 0;
  static const struct Pairs {                        // byte -> its two bases
    char c[256][2];
    Pairs() { for (int b = 0; b < 256; b++) { c[b][0] = "=ACMGRSVTWYHKDBN"[b >> 4]; c[b][1] = "=ACMGRSVTWYHKDBN"[b & 15]; } }
  } pairs;
  while (r->rec_seq_left && out < max_length) {
    if ((r->rec_seq_index & 1) == 0 && r->pos < r->len) {          // whole bytes straight from the buffer
      const uint64_t nb = std::min<uint64_t>(std::min<uint64_t>(r->rec_seq_left, max_length - out) / 2, r->len - r->pos);
      const unsigned char *src = r->buf + r->pos;
      for (uint64_t i = 0; i < nb; i++) { seq[out + 2 * i] = pairs.c[src[i]][0]; seq[out + 2 * i + 1] = pairs.c[src[i]][1]; }
      out += 2 * nb; r->pos += nb; r->rec_seq_index += 2 * nb; r->rec_seq_left -= 2 * nb;
      if (nb) continue;
    }
    if ((r->rec_seq_index & 1) == 0) {
      const int c = r->peek();
      if (c < 0) return bad("truncated BAM record");
      r->advance();
      r->rec_byte = c;
      seq[out++] = nt16[c >> 4];
    } else {
      seq[out++] = nt16[r->rec_byte & 15];
    }
    r->rec_seq_index++;
    r->rec_seq_left--;
  }
  *seq_length = out;
  if (r->rec_seq_left) return 1;                     // buffer full, the sequence continues
  if (!r->get_bytes(nullptr, r->rec_skip)) return bad("truncated BAM record");
  r->in_sequence = false;
  *end_of_sequence = 1;
  return 1;
}

// SAM: header lines start with '@'; SEQ is the 10th tab-separated column of an alignment line
int sam_load_bases(msr_reader *r, char *seq, uint64_t max_length, uint64_t *seq_length, int *end_of_sequence) {
  uint64_t out = 0;
  for (;;) {
    const int c = r->peek();
    if (c < 0) {
      if (r->failed) { seq_err("msr_load_bases: '" + r->name + "': " + r->failure); return -2; }
      if (r->in_sequence) { r->in_sequence = false; *seq_length = out; *end_of_sequence = 1; return 1; }
      return 0;
    }
    if (r->line_start) {
      if (c == '\n' || c == '\r') { r->advance(); continue; }
      r->line_start = false;
      r->sam_field = (c == '@') ? -1 : 0;           // -1: a header line, skipped whole
    }
    if (c == '\n') {
      r->advance();
      r->line_start = true;
      r->in_sequence = false;
      if (r->sam_field < 0) continue;                // a header line
      if (r->sam_field >= 9) { *seq_length = out; *end_of_sequence = 1; return 1; }
      seq_err("msr_load_bases: '" + r->name + "': SAM alignment line with fewer than 10 columns");
      return -2;
    }
    if (r->sam_field < 0) { r->advance(); continue; }
    if (c == '\t') { r->advance(); r->sam_field++; if (r->sam_field == 9) r->in_sequence = true; continue; }
    if (r->sam_field == 9 && c != '\r' && c != '*') {
      if (out == max_length) { *seq_length = out; return 1; }
      seq[out++] = (char)c;
    }
    r->advance();
  }
}
}  // namespace

namespace {
// BAM records that sit whole in the read buffer are decoded straight from it (no per-field calls): returns the bytes
// written to `out` (bases + one '.' per record); stops at a record that is not complete in the buffer or does not fit.
uint64_t bam_fast_records(msr_reader *r, char *out, uint64_t room) {
  static const struct Pairs16 {
    uint16_t v[256];
    Pairs16() {
      for (int b = 0; b < 256; b++) {
        const unsigned char two[2] = {(unsigned char)"=ACMGRSVTWYHKDBN"[b >> 4], (unsigned char)"=ACMGRSVTWYHKDBN"[b & 15]};
        memcpy(&v[b], two, 2);
      }
    }
  } pairs;
  uint64_t w = 0;
  while (r->len - r->pos >= 36) {
    const unsigned char *p = r->buf + r->pos;
    const uint64_t block_size = (uint64_t)p[0] | ((uint64_t)p[1] << 8) | ((uint64_t)p[2] << 16) | ((uint64_t)p[3] << 24);
    if (block_size < 32 || r->len - r->pos < 4 + block_size) break;
    const unsigned char *fix = p + 4;
    const uint64_t l_read_name = fix[8], n_cigar = (uint64_t)fix[12] | ((uint64_t)fix[13] << 8);
    const uint64_t l_seq = (uint64_t)fix[16] | ((uint64_t)fix[17] << 8) | ((uint64_t)fix[18] << 16) | ((uint64_t)fix[19] << 24);
    const uint64_t before = 32 + l_read_name + 4 * n_cigar, packed = (l_seq + 1) / 2;
    if (before + packed > block_size) break;         // corrupt: let the careful path report it
    if (l_seq + 2 > room - w) break;                 // (one spare byte: the pair table writes two at a time)
    const unsigned char *src = fix + before;
    char *dst = out + w;
    for (uint64_t i = 0; i < packed; i++) memcpy(dst + 2 * i, &pairs.v[src[i]], 2);
    w += l_seq;
    out[w++] = '.';
    r->pos += 4 + block_size;
  }
  return w;
}
}  // namespace

extern "C" int msr_load_stream(msr_reader *r, char *buf, uint64_t max_length, uint64_t *length) {
  if (!r || !buf || !length || max_length < 2) return -1;
  uint64_t out = 0;
  *length = 0;
  while (out + 1 < max_length) {                     // room for at least one base and its breaker
    if (r->format == MSR_FORMAT_BAM && r->bam_header_done && !r->in_sequence) {
      out += bam_fast_records(r, buf + out, max_length - out);
      if (out + 1 >= max_length) break;
    }
    uint64_t n = 0;
    int eos = 0;
    const int rc = msr_load_bases(r, buf + out, max_length - out - 1, &n, &eos);
    if (rc < 0) return rc;
    if (rc == 0) { *length = out; return out ? 1 : 0; }
    out += n;
    if (eos) buf[out++] = '.';
  }
  *length = out;
  return 1;
}

extern "C" int msr_load_bases(msr_reader *r, char *seq, uint64_t max_length, uint64_t *seq_length, int *end_of_sequence) {
  if (!r || !seq || !seq_length || !end_of_sequence) return -1;
  uint64_t out = 0;
  *seq_length = 0;
  *end_of_sequence = 0;
  if (r->format == MSR_FORMAT_BAM) return bam_load_bases(r, seq, max_length, seq_length, end_of_sequence);
  if (r->format == MSR_FORMAT_SAM) return sam_load_bases(r, seq, max_length, seq_length, end_of_sequence);

  for (;;) {
    const int c = r->peek();
    if (c < 0) {                                     // end of input
      if (r->failed) { seq_err("msr_load_bases: '" + r->name + "': " + r->failure); return -2; }
      if (r->in_sequence) { r->in_sequence = false; *seq_length = out; *end_of_sequence = 1; return 1; }
      return 0;
    }
    switch (r->st) {
      case msr_reader::AT_RECORD_START:
        if (c == '>' || c == '@') {
          r->fastq = (c == '@');
          r->st = msr_reader::IN_HEADER;
          r->advance();
        } else if (c == '\n' || c == '\r' || c == ' ' || c == '\t') {
          r->advance();                              // blank lines between records
        } else {
          seq_err("msr_load_bases: '" + r->name + "' is neither FASTA nor FASTQ (record starts with '" + std::string(1, (char)c) + "')");
          return -2;
        }
        break;

      case msr_reader::IN_HEADER:
        r->advance();
        if (c == '\n') {
          r->st = r->fastq ? msr_reader::IN_FASTQ_SEQ : msr_reader::IN_FASTA_SEQ;
          r->line_start = true;
          r->seq_bases = 0;
          r->in_sequence = true;
        }
        break;

      case msr_reader::IN_FASTA_SEQ:
        if (r->line_start && c == '>') {             // next record: this sequence is complete
          r->st = msr_reader::AT_RECORD_START;
          r->in_sequence = false;
          *seq_length = out; *end_of_sequence = 1;
          return 1;
        }
        if (c == '\n') { r->line_start = true; r->advance(); break; }
        r->line_start = false;
        if (c == '\r' || c == ' ' || c == '\t') { r->advance(); break; }
        if (out == max_length) { *seq_length = out; return 1; }          // buffer full, sequence continues
        seq[out++] = (char)c;
        r->advance();
        break;

      case msr_reader::IN_FASTQ_SEQ:
        if (r->line_start && c == '+') {             // separator line: the bases are complete
          r->st = msr_reader::IN_FASTQ_PLUS;
          r->advance();
          break;
        }
        if (c == '\n') { r->line_start = true; r->advance(); break; }
        r->line_start = false;
        if (c == '\r' || c == ' ' || c == '\t') { r->advance(); break; }
        if (out == max_length) { *seq_length = out; return 1; }
        seq[out++] = (char)c;
        r->seq_bases++;
        r->advance();
        break;

      case msr_reader::IN_FASTQ_PLUS:
        r->advance();
        if (c == '\n') { r->st = msr_reader::IN_FASTQ_QUAL; r->qual_left = r->seq_bases; }
        break;

      case msr_reader::IN_FASTQ_QUAL:
        if (r->qual_left == 0) {
          // the record is done once its quality line has ended
          if (c == '\n' || c == '\r') { r->advance(); break; }
          r->st = msr_reader::AT_RECORD_START;
          r->in_sequence = false;
          *seq_length = out; *end_of_sequence = 1;
          return 1;
        }
        r->advance();
        if (c != '\n' && c != '\r') r->qual_left--;
        break;
    }
  }
}
