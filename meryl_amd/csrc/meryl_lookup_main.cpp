// meryl_lookup_main.cpp -- `meryl-lookup -existence`: for every sequence of a FASTA/FASTQ file, the number of k-mers it
// holds and how many of them occur in a meryl database.
//
// Keeps the reference tool's surface for that mode (src/meryl-lookup/meryl-lookup.C:150-200 options, existence.C:48-132):
//   meryl-lookup -existence -sequence <in.fa[.gz]> -mers <db.meryl> [<db2.meryl> ...] [-min v] [-max v] [-output out.tsv]
// output per sequence:  name <TAB> kmersInSequence { <TAB> kmersInDB <TAB> kmersFound } per database   (existence.C:96-113)
// The database is loaded into the device-resident exact lookup table (include/meryl_lookup.h = merylExactLookup); the
// sequences go to the device as one base stream with '.' between them.  The other modes of the reference tool (-dump,
// -include, -exclude, -bed, -wig) are not part of this build.
#include "../../include/meryl_gpu_count.h"
#include "../../include/meryl_lookup.h"
#include "../../include/meryl_seq.h"

#include <hip/hip_runtime.h>

#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {
[[noreturn]] void die(const char *fmt, const char *a = "") {
  fprintf(stderr, fmt, a);
  fprintf(stderr, "\n");
  exit(1);
}

// names and bases of a FASTA/FASTQ text (multi-line FASTA, four-line FASTQ; the reference reads both through
// dnaSeqFile::loadSequence, meryl-utility)
struct Seqs { std::vector<std::string> names; std::vector<uint64_t> start; std::string bases; };

void parse_text(const std::string &text, Seqs &out) {
  size_t p = 0;
  const size_t n = text.size();
  auto line = [&](size_t &a, size_t &b) {                   // next line [a, b) without its terminator; false at the end
    if (p >= n) return false;
    a = p;
    while (p < n && text[p] != '\n') p++;
    b = p;
    if (p < n) p++;
    if (b > a && text[b - 1] == '\r') b--;
    return true;
  };
  size_t a, b;
  bool have = line(a, b);
  while (have) {
    if (b == a) { have = line(a, b); continue; }
    const char c = text[a];
    if (c != '>' && c != '@') die("ERROR: sequence file: a record starts with '%s', neither '>' nor '@'.", std::string(1, c).c_str());
    size_t e = a + 1;
    while (e < b && text[e] != ' ' && text[e] != '\t') e++;
    out.names.push_back(text.substr(a + 1, e - a - 1));       // the identifier: first word of the header (dnaSeq::ident())
    out.start.push_back(out.bases.size());
    if (c == '>') {
      while ((have = line(a, b)) && !(b > a && text[a] == '>')) out.bases.append(text, a, b - a);
    } else {
      if ((have = line(a, b))) out.bases.append(text, a, b - a);
      if ((have = line(a, b))) {                              // '+' line
        if ((have = line(a, b))) have = line(a, b);           // quality line, then the next record
      }
    }
    out.bases.push_back('.');
  }
  out.start.push_back(out.bases.size());
}
}  // namespace

int main(int argc, char **argv) {
  std::string seq_name, out_name;
  std::vector<std::string> dbs;
  uint64_t vmin = 0, vmax = UINT64_MAX;
  bool existence = false, estimate = false;
  double max_memory_gb = 0.0;                                 // -memory: 0 = whatever the device has
  for (int a = 1; a < argc; a++) {
    const std::string w = argv[a];
    if (w == "-existence") existence = true;
    else if (w == "-sequence" && a + 1 < argc) seq_name = argv[++a];
    else if (w == "-output" && a + 1 < argc) out_name = argv[++a];
    else if (w == "-min" && a + 1 < argc) vmin = strtoull(argv[++a], nullptr, 10);
    else if (w == "-max" && a + 1 < argc) vmax = strtoull(argv[++a], nullptr, 10);
    else if (w == "-threads" && a + 1 < argc) ++a;
    else if (w == "-memory" && a + 1 < argc) max_memory_gb = strtod(argv[++a], nullptr);
    else if (w == "-estimate") estimate = true;
    else if (w == "-mers") { while (a + 1 < argc && argv[a + 1][0] != '-') dbs.push_back(argv[++a]); }
    else if (w == "-dump" || w == "-include" || w == "-exclude" || w == "-bed" || w == "-bed-runs" || w == "-wig-count" || w == "-wig-depth")
      die("ERROR: mode '%s' is not part of this build (-existence only).", w.c_str());
    else die("ERROR: unknown option '%s'.", w.c_str());
  }
  if (!existence || (seq_name.empty() && !estimate) || dbs.empty()) {
    fprintf(stderr, "usage: %s -existence -sequence <in.fa|fq[.gz]> -mers <db.meryl> [...] [-min v] [-max v] [-memory GB] [-estimate] [-output out.tsv]\n", argv[0]);
    return 1;
  }

  // meryl-lookup.C:62-87: the memory every table will need, BEFORE anything is loaded (here: device memory, from the databases'
  // own value histograms -- no device is touched); -estimate stops after the report, -memory is the limit it is held against
  double required_gb = 0.0;
  for (const std::string &d : dbs) {
    fprintf(stderr, "\nEstimating memory usage for '%s'.\n", d.c_str());
    mgc_lookup_info est;
    if (mgc_lookup_estimate(d.c_str(), vmin, vmax, &est) != MGC_OK) die("ERROR: %s", mgc_lookup_error());
    fprintf(stderr, "  %" PRIu64 " of %" PRIu64 " %u-mers kept, %.3f GB of device memory.\n", est.n_kmers, est.n_kmers_in_db, est.k,
            (double)est.device_bytes / 1024.0 / 1024.0 / 1024.0);
    required_gb += (double)est.device_bytes / 1024.0 / 1024.0 / 1024.0;
  }
  fprintf(stderr, "\nMemory required:  %.3f GB\n", required_gb);
  if (max_memory_gb > 0.0) {
    fprintf(stderr, "Memory limit:     %.3f GB\n", max_memory_gb);
    if (required_gb > max_memory_gb) { fprintf(stderr, "\nNot enough memory to load databases.  Increase -memory.\n"); return 1; }
  }
  if (estimate) { fprintf(stderr, "\nStopping after memory estimated reported; -estimate option enabled.\n"); return 0; }

  msr_reader *r = msr_open(seq_name.c_str());
  if (!r) die("ERROR: %s", msr_last_error());
  if (msr_format(r) != MSR_FORMAT_FASTX) die("ERROR: '%s' is not FASTA/FASTQ text.", seq_name.c_str());
  std::string text;
  std::vector<char> buf(16u << 20);
  for (;;) {
    const int64_t got = msr_read_text(r, buf.data(), buf.size());
    if (got < 0) die("ERROR: %s", msr_last_error());
    if (got == 0) break;
    text.append(buf.data(), (size_t)got);
  }
  msr_close(r);
  Seqs sq;
  parse_text(text, sq);
  std::string().swap(text);
  const uint64_t n_seq = sq.names.size();

  uint8_t *d_bases = nullptr;
  uint64_t *d_start = nullptr, *d_total = nullptr, *d_found = nullptr;
  auto hip = [](hipError_t e, const char *what) { if (e != hipSuccess) { fprintf(stderr, "ERROR: %s: %s\n", what, hipGetErrorString(e)); exit(1); } };
  hip(hipMalloc(reinterpret_cast<void **>(&d_bases), sq.bases.size() + 1), "hipMalloc");
  hip(hipMalloc(reinterpret_cast<void **>(&d_start), 8 * (n_seq + 1)), "hipMalloc");
  hip(hipMalloc(reinterpret_cast<void **>(&d_total), 8 * (n_seq + 1)), "hipMalloc");
  hip(hipMalloc(reinterpret_cast<void **>(&d_found), 8 * (n_seq + 1)), "hipMalloc");
  hip(hipMemcpy(d_bases, sq.bases.data(), sq.bases.size(), hipMemcpyHostToDevice), "upload");
  hip(hipMemcpy(d_start, sq.start.data(), 8 * (n_seq + 1), hipMemcpyHostToDevice), "upload");

  std::vector<std::vector<uint64_t>> found(dbs.size());
  std::vector<uint64_t> total(n_seq), n_in_db(dbs.size());
  for (size_t d = 0; d < dbs.size(); d++) {
    fprintf(stderr, "\nLoading kmers from '%s' into lookup table.\n", dbs[d].c_str());          // meryl-lookup.C:89
    mgc_lookup *t = mgc_lookup_load(dbs[d].c_str(), vmin, vmax, -1, 0);
    if (!t) die("ERROR: %s", mgc_lookup_error());
    mgc_lookup_info info;
    mgc_lookup_get_info(t, &info);
    n_in_db[d] = info.n_kmers;
    if (mgc_lookup_existence(t, d_bases, sq.bases.size(), d_start, n_seq, d_total, d_found, nullptr) != MGC_OK) die("ERROR: %s", mgc_lookup_error());
    hip(hipDeviceSynchronize(), "lookup");
    found[d].resize(n_seq);
    if (n_seq) {
      hip(hipMemcpy(found[d].data(), d_found, 8 * n_seq, hipMemcpyDeviceToHost), "download");
      hip(hipMemcpy(total.data(), d_total, 8 * n_seq, hipMemcpyDeviceToHost), "download");
    }
    mgc_lookup_free(t);
  }
  FILE *out = out_name.empty() ? stdout : fopen(out_name.c_str(), "w");
  if (!out) die("ERROR: cannot write '%s'.", out_name.c_str());
  for (uint64_t s = 0; s < n_seq; s++) {                                                         // existence.C:96-113
    fprintf(out, "%s\t%" PRIu64, sq.names[s].c_str(), total[s]);
    for (size_t d = 0; d < dbs.size(); d++) fprintf(out, "\t%" PRIu64 "\t%" PRIu64, n_in_db[d], found[d][s]);
    fprintf(out, "\n");
  }
  if (out != stdout) fclose(out);
  (void)hipFree(d_bases); (void)hipFree(d_start); (void)hipFree(d_total); (void)hipFree(d_found);
  return 0;
}
