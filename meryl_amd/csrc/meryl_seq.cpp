// meryl_seq.cpp -- FASTA/FASTQ (plain or gzip) loader, include/meryl_seq.h.
//
// Stands in for dnaSeqFile::loadBases / openSequenceFile of the absent
// meryl-utility submodule (call sites src/meryl/merylOp.C:200,
// src/meryl/merylInput.C:257); contract per src/meryl/merylInput.H:67-70.
// zlib's gzread is transparent for uncompressed files, so one code path
// serves both.  Parsing rules: '>' starts a FASTA record (header to end of
// line, then sequence lines until the next '>' at a line start); '@' starts a
// FASTQ record (header line, sequence lines up to the '+' line, then as many
// quality characters as there were bases).  White space inside sequence
// lines is dropped; every other byte is handed on as is (the k-mer packer
// decides what is a base).
#include "../../include/meryl_seq.h"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

struct msr_reader {
  gzFile      gz = nullptr;
  std::string name;
  bool        compressed = false;
  unsigned char *buf = nullptr;
  size_t      cap = 1u << 20, len = 0, pos = 0;
  bool        eof = false;

  enum State { AT_RECORD_START, IN_HEADER, IN_FASTA_SEQ, IN_FASTQ_SEQ, IN_FASTQ_PLUS, IN_FASTQ_QUAL } st = AT_RECORD_START;
  bool        line_start = true;     // next byte is the first of a line
  uint64_t    seq_bases = 0;         // bases of the current FASTQ record (to skip as many qualities)
  uint64_t    qual_left = 0;
  bool        fastq = false;
  bool        in_sequence = false;   // a sequence has been started and not yet reported as ended

  int peek() {
    if (pos == len) {
      if (eof) return -1;
      const int n = gzread(gz, buf, (unsigned)cap);
      if (n <= 0) { eof = true; return -1; }
      len = (size_t)n; pos = 0;
    }
    return buf[pos];
  }
  void advance() { pos++; }
};

extern "C" const char *msr_last_error(void) { return g_seq_error.c_str(); }

extern "C" msr_reader *msr_open(const char *name) {
  if (!name || !*name) { seq_err("msr_open: empty file name"); return nullptr; }
  const std::string n(name);
  if (ends_with(n, ".bz2") || ends_with(n, ".xz")) { seq_err("msr_open: '" + n + "': bz2/xz input is not supported (use gzip or a pipe)"); return nullptr; }
  if (ends_with(n, ".bam") || ends_with(n, ".cram") || ends_with(n, ".sam")) { seq_err("msr_open: '" + n + "': SAM/BAM/CRAM input is not supported"); return nullptr; }
  msr_reader *r = new msr_reader();
  r->name = n;
  r->compressed = ends_with(n, ".gz");
  r->gz = (n == "-") ? gzdopen(0, "rb") : gzopen(name, "rb");
  if (!r->gz) { seq_err("msr_open: cannot open '" + n + "': " + strerror(errno)); delete r; return nullptr; }
  gzbuffer(r->gz, 1u << 20);
  r->buf = (unsigned char *)malloc(r->cap);
  return r;
}

extern "C" void msr_close(msr_reader *r) {
  if (!r) return;
  if (r->gz) gzclose(r->gz);
  free(r->buf);
  delete r;
}

extern "C" int64_t msr_read_text(msr_reader *r, char *buf, uint64_t max_length) {
  if (!r || !buf) return -1;
  if (r->st != msr_reader::AT_RECORD_START || r->in_sequence || r->pos != r->len) {
    seq_err("msr_read_text: '" + r->name + "': raw and parsed reads cannot be mixed on one reader");
    return -1;
  }
  uint64_t got = 0;
  while (got < max_length && !r->eof) {
    const uint64_t want = max_length - got;
    const int n = gzread(r->gz, buf + got, (unsigned)(want > (1u << 30) ? (1u << 30) : want));
    if (n < 0) { seq_err("msr_read_text: read error in '" + r->name + "'"); return -1; }
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

extern "C" int msr_load_bases(msr_reader *r, char *seq, uint64_t max_length, uint64_t *seq_length, int *end_of_sequence) {
  if (!r || !seq || !seq_length || !end_of_sequence) return -1;
  uint64_t out = 0;
  *seq_length = 0;
  *end_of_sequence = 0;

  for (;;) {
    const int c = r->peek();
    if (c < 0) {                                     // end of input
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
