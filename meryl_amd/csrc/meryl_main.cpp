// meryl_main.cpp -- `meryl` command-line front end for the MI355X count engine.
//
// Keeps the reference's CLI surface for the count path (north_star: `meryl count
// k=K <reads> output <db>`): the positional word grammar of
// src/meryl/meryl.C:30-87 and src/meryl/merylCommandBuilder.C (options :187-326,
// operations :346-385, `output` :440-461, sequence inputs :537-549, `[ ]`
// nesting :133-153), the stderr narrative of src/meryl/merylOp-count.C:324-401
// (including the line Canu parses, :398-401), and the debug verbs `print` and
// `dumpIndex` (src/meryl/meryl.C:36-46, src/meryl/merylOp-nextMer.C:665-677) on the
// databases it writes.  Beyond the count path: `histogram`, `dumpFile`, the set operations over
// databases (union[-min|-max|-sum], intersect[-min|-max|-sum], subtract, difference, symmetric-difference;
// merylOp-nextMer.C:559-613) and the single-input value filters / arithmetic (less-than ... modulo; :490-557), all merged on
// the device.  What stays refused: statistics, compare, ploidy, Canu sequence stores (segment=); CRAM only when no `samtools` is on the PATH.
#include "../../include/meryl_db.h"
#include "../../include/meryl_gpu_count.h"
#include "../../include/meryl_seq.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <chrono>
#include <vector>

namespace {
// wall-clock stamp of static initialisation (just before main()): what lies before the input is read -- the HIP runtime, the
// device, the session's streams -- is reported on its own in the -V TIMING line
const std::chrono::steady_clock::time_point g_t_main = std::chrono::steady_clock::now();

// ---- scaledNumber / scaledUnit / scaledName [meryl-utility, not in tree] ----
uint64_t scaledNumber(uint64_t n, uint32_t div = 1024) { for (int i = 0; i < 8 && n > 9999; i++) n /= div; return n; }
char scaledUnit(uint64_t n, uint32_t div = 1024) {
  const char u[] = { ' ', 'k', 'M', 'G', 'T', 'P', 'E', 'Z', 'Y' };
  int i = 0;
  for (; i < 8 && n > 9999; i++) n /= div;
  return u[i];
}
const char *scaledName(uint64_t n, uint32_t div = 1024) {
  const char *u[] = { "", " thousand", " million", " billion", " trillion", " quadrillion", " quintillion", "", "" };
  int i = 0;
  for (; i < 8 && n > 9999; i++) n /= div;
  return u[i];
}
uint64_t bits64(uint64_t v) { uint64_t b = 0; while (v) { b++; v >>= 1; } return b; }

bool file_exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0 && !S_ISDIR(st.st_mode); }
bool dir_has_index(const std::string &p) { return file_exists(p + "/merylIndex"); }

bool has_compressed_suffix(const std::string &n) {
  for (const char *suf : { ".gz", ".bz2", ".xz", ".bgz", ".bgzf", ".zst" })
    if (n.size() > strlen(suf) && n.compare(n.size() - strlen(suf), strlen(suf), suf) == 0) return true;
  return false;
}

enum OpKind { OP_NONE, OP_COUNT, OP_COUNT_FORWARD, OP_COUNT_REVERSE, OP_PRINT, OP_DUMPINDEX, OP_HISTOGRAM, OP_DUMPFILE, OP_MERGE, OP_VALUE };

struct InputRef { std::string path; int child = -1; };      // a database on the command line, or the output of a child operation

struct Operation {
  OpKind                   kind = OP_NONE;
  std::vector<std::string> seq_inputs;      // sequence files (count ops)
  std::vector<bool>        seq_compress;    // `compress` is sticky per input (merylCommandBuilder.C:237-240,546)
  std::vector<InputRef>    inputs;          // databases and child operations, in command-line order
  std::vector<std::string> db_inputs;       // ... resolved to database paths once the children have run
  std::string              word;            // the operation as typed
  int                      merge_op = -1;   // MGC_MERGE_* (OP_MERGE)
  int                      value_op = -1;   // MGC_VALUE_* (OP_VALUE): less-than ... modulo
  uint64_t                 constant = ~0ull;   // its threshold / constant (merylCommandBuilder.C:216-233,291-294); ~0: not given
  double                   frac_distinct = -1, word_freq = -1;   // distinct=<f> / word-frequency=<f> thresholds (:278-289)
  int                      parent = -1;
  std::string              output;
  uint64_t                 exp_num_kmers = 0;   // n=
  std::string              count_suffix;        // count-suffix=
  uint64_t                 label = 0;           // label=#<n> (meryl2: the constant label of a count, merylCommandBuilder-isAssign.C:124)
};

struct Globals {
  uint32_t k = 0;
  double   memory_gb;
  uint32_t threads;
  int      verbosity = 2;          // sayStandard
  bool     only_config = false;
  bool     compress = false;       // sticky
  uint32_t gpus = 1;               // gpus=<N> (this build only): ranks of ONE count spread over the node's devices
  uint32_t label_size = 0;         // -l <bits> (meryl2: kmerTiny::setLabelSize, merylGlobals.C:75-77)
  bool     fast_exit = false;      // set when ONE count is all the command does (and MERYL_FAST_EXIT is not 0): see run_count
  Globals() {
    const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
    memory_gb = (pages > 0 && psz > 0) ? (double)pages * (double)psz / 1024.0 / 1024.0 / 1024.0 : 16.0;
    threads = std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 1;
  }
};

void usage(const char *prog) {
  fprintf(stderr,
          "usage: %s [k=<K>] [memory=<GB>] [threads=<T>] [gpus=<N>] [n=<kmers>] [compress] [-l <label-bits>] [-C] [-Q] [-V]\n"
          "          count|count-forward|count-reverse [label=#<n>] <reads.fa|fq[.gz]|sam|bam> ... output <database.meryl>\n"
          "       %s print <database.meryl>\n"
          "       %s dumpIndex <database.meryl>\n"
          "       %s dumpFile <database.meryl>/0x######\n"
          "       %s union[-min|-max|-sum]|intersect[-min|-max|-sum]|subtract|difference|symmetric-difference <db | [operation]> ... output <db>\n"
          "       %s less-than|greater-than|at-least|at-most|equal-to|not-equal-to <N | distinct=<f> | word-frequency=<f>> <db | [operation]> output <db>\n"
          "       %s increase|decrease|multiply|divide|divide-round|modulo <N> <db | [operation]> output <db>\n"
          "\n"
          "  MI355X-native implementation of the `count` path of marbl/meryl.  Words are processed left to\n"
          "  right; options apply to the operations that follow.  A leading '[' and trailing ']' group the\n"
          "  words of one operation; an operation inside another one's brackets is its input.  Other meryl operations are\n"
          "  not part of this build.\n",
          prog, prog, prog, prog, prog, prog, prog);
}

[[noreturn]] void die(const char *fmt, const char *a = "") {
  fprintf(stderr, fmt, a);
  fprintf(stderr, "\n");
  exit(1);
}

// ---- the configuration narrative, merylOp-count.C:118-165,231-295,324-401 ----
void print_configuration(const Globals &g, const Operation &op, uint64_t exp_num_kmers, const mgc_count_config &cfg) {
  const uint32_t k = g.k;
  fprintf(stderr, "\n");
  fprintf(stderr, "Counting %" PRIu64 " (estimated)%s %s%s%s %u-mers from %zu input file%s:\n",             // :324-330
          scaledNumber(exp_num_kmers), scaledName(exp_num_kmers),
          (op.kind == OP_COUNT) ? "canonical" : "", (op.kind == OP_COUNT_FORWARD) ? "forward" : "",
          (op.kind == OP_COUNT_REVERSE) ? "reverse" : "", k, op.seq_inputs.size(), (op.seq_inputs.size() == 1) ? "" : "s");
  for (const std::string &n : op.seq_inputs) fprintf(stderr, "  %15s: %s\n", "sequence-file", n.c_str());            // :332-333

  // SIMPLE MODE, :136-162
  fprintf(stderr, "\n\nSIMPLE MODE\n-----------\n\n");
  const uint32_t sl = cfg.count_suffix_length;
  if (2 * k - 2 * sl > 42) {
    fprintf(stderr, "  Not possible.\n");
  } else {
    const uint64_t n_entries = (uint64_t)1 << (2 * k - 2 * sl), low_bits = 16;
    const uint64_t exp_max = (uint64_t)(0.004 * (double)exp_num_kmers), exp_bits = bits64(exp_max) + 1;
    const uint64_t extra = (exp_bits < low_bits) ? 0 : exp_bits - low_bits;
    const uint64_t low_mem = n_entries * low_bits, high_mem = n_entries * extra, tot = (low_mem + high_mem) / 8;
    if (sl == 0) fprintf(stderr, "  %u-mers\n", k);                                                                 // :147-150
    else         fprintf(stderr, "  %u-mers with constant %u-mer suffix '%s'\n", k, sl, cfg.count_suffix);
    fprintf(stderr, "    -> %" PRIu64 " entries for counts up to %u.\n", n_entries, 65535u);
    fprintf(stderr, "    -> %" PRIu64 " %cbits memory used\n", scaledNumber(low_mem), scaledUnit(low_mem));
    fprintf(stderr, "\n  %" PRIu64 " input bases\n", exp_num_kmers);
    fprintf(stderr, "    -> expected max count of %" PRIu64 ", needing %" PRIu64 " extra bits.\n", exp_max, extra);
    if (extra > 0) fprintf(stderr, "    -> %" PRIu64 " %cbits memory used\n", scaledNumber(high_mem), scaledUnit(high_mem));
    else           fprintf(stderr, "    -> no memory used\n");
    fprintf(stderr, "\n  %" PRIu64 " %cB memory needed\n", scaledNumber(tot), scaledUnit(tot));
  }

  // COMPLEX MODE table, :245-293 (evaluated at expNumKmers / nBatches like findBestValues)
  if (k > 5) {
    const uint64_t n_est = exp_num_kmers / (cfg.n_batches ? cfg.n_batches : 1);
    const uint32_t seg_bits = 4096 * 8, seg_bytes = 4096;
    const uint64_t mem_used = cfg.use_simple ? UINT64_MAX : cfg.memory_used;
    fprintf(stderr, "\n\nCOMPLEX MODE\n------------\n\n");
    fprintf(stderr, "prefix     # of   struct   kmers/    segs/      min     data    total\n");
    fprintf(stderr, "  bits   prefix   memory   prefix   prefix   memory   memory   memory\n");
    fprintf(stderr, "------  -------  -------  -------  -------  -------  -------  -------\n");
    for (uint32_t wp = 1; wp < 2 * k - 1; wp++) {
      const uint64_t n_prefix = (uint64_t)1 << wp, kpp = n_est / n_prefix + 1, kps = seg_bits / (2 * k - wp), spp = kpp / kps + 1;
      if (wp + bits64(spp) + bits64(seg_bytes) >= 64) break;
      const uint64_t struct_mem = (uint64_t)3232 * n_prefix + 8 * n_prefix * spp, data_min = n_prefix * seg_bytes;
      const uint64_t data_mem = n_prefix * spp * seg_bytes, total = struct_mem + data_mem;
      fprintf(stderr, "%6u  %4" PRIu64 " %cP  %4" PRIu64 " %cB  %4" PRIu64 " %cM  %4" PRIu64 " %cS  %4" PRIu64 " %cB  %4" PRIu64 " %cB  %4" PRIu64 " %cB%s\n",
              wp, scaledNumber(n_prefix), scaledUnit(n_prefix), scaledNumber(struct_mem), scaledUnit(struct_mem),
              scaledNumber(kpp), scaledUnit(kpp), scaledNumber(spp), scaledUnit(spp),
              scaledNumber(data_min), scaledUnit(data_min), scaledNumber(data_mem), scaledUnit(data_mem),
              scaledNumber(total), scaledUnit(total), (wp == cfg.w_prefix && !cfg.use_simple) ? "  Best Value!" : "");
      if (total > (uint64_t)16 * mem_used) break;
    }
  }

  // FINAL CONFIGURATION, :390-401
  char line[256];
  mgc_format_configured_line(&cfg, line, sizeof(line));
  fprintf(stderr, "\n\nFINAL CONFIGURATION\n-------------------\n\n");
  fprintf(stderr, "Estimated to require %" PRIu64 " %cB memory out of %" PRIu64 " %cB allowed.\n",
          scaledNumber(cfg.memory_used), scaledUnit(cfg.memory_used), scaledNumber(cfg.memory_allowed), scaledUnit(cfg.memory_allowed));
  fprintf(stderr, "Estimated to require %u batch%s.\n", cfg.n_batches, (cfg.n_batches == 1) ? "" : "es");
  fprintf(stderr, "\n%s\n\n", line);
}

// one input of a count, or a byte window of it (a plain-text file read by one rank of a node count)
struct InputPiece { std::string name; uint64_t begin, end; };

// Input: the file's raw text goes to the device and is parsed there (mgc_push_text*); files the device parser refuses
// (multi-line FASTQ ...), SAM/BAM and MERYL_HOST_PARSER=1 take the host state machine of meryl_seq.cpp.  Returns the
// bytes / bases taken in (reported with -V -V).
int g_inflate_threads = 16;                                  // threads= of the command line: the BGZF inflaters (readers of plain text: at most 16)
uint64_t load_inputs(mgc_session *s, const std::vector<InputPiece> &pieces, int reader_threads) {
  const uint64_t buf_max = 2 * 1024 * 1024;                                                            // merylOp-countThreads.C:413
  std::vector<char> buf(buf_max);
  uint64_t total_bases = 0;
  const bool host_parser = getenv("MERYL_HOST_PARSER") && getenv("MERYL_HOST_PARSER")[0] == '1';
  auto load_on_host = [&](const std::string &name) {
    msr_reader *r = msr_open(name.c_str());
    if (!r) die("ERROR: %s", msr_last_error());
    for (;;) {                                                                                         // loader loop, :173-203
      uint64_t len = 0;                                                                                // 2 MiB of sequences, '.' after each
      const int rc = msr_load_stream(r, buf.data(), buf_max, &len);
      if (rc < 0) die("ERROR: %s", msr_last_error());
      if (rc == 0) break;
      if (mgc_push_bases(s, buf.data(), len, 0) != MGC_OK) die("ERROR: %s", mgc_last_error(s));
      total_bases += len;
    }
    mgc_push_bases(s, nullptr, 0, 1);                                                                  // end-of-file breaker, :196
    msr_close(r);
  };
  std::vector<char> text(host_parser ? 0 : (16u << 20));
  for (const InputPiece &pc : pieces) {
    const std::string &name = pc.name;
    const bool whole = pc.begin == 0 && pc.end == ~0ull;
    if (host_parser && whole) { load_on_host(name); continue; }
    msr_reader *r = msr_open(name.c_str());
    if (!r) die("ERROR: %s", msr_last_error());
    if (msr_format(r) != MSR_FORMAT_FASTX) { msr_close(r); load_on_host(name); continue; }   // SAM/BAM records are decoded on the host
    if (name != "-" && !has_compressed_suffix(name)) {
      // plain text: the library reads the file itself, several threads straight into its pinned upload buffers
      msr_close(r);
      const int rc = whole ? mgc_push_text_file(s, name.c_str(), 0, reader_threads)
                           : mgc_push_text_file_range(s, name.c_str(), 0, reader_threads, pc.begin, pc.end);
      if (rc == MGC_EFORMAT && whole) load_on_host(name);
      else if (rc == MGC_EFORMAT) die("ERROR: '%s' is not strict four-line FASTQ / FASTA: it cannot be read in windows by several ranks (gpus=1 reads it)", name.c_str());
      else if (rc != MGC_OK) die("ERROR: %s", mgc_last_error(s));
      struct stat fst;
      if (whole) { if (stat(name.c_str(), &fst) == 0) total_bases += (uint64_t)fst.st_size; }
      else total_bases += pc.end - pc.begin;
      continue;
    }
    if (name != "-" && whole && mgc_is_bgzf_file(name.c_str()) && !getenv("MERYL_BGZF_GENERIC")) {
      // BGZF (bgzip'd FASTA / FASTQ): the library maps the file and inflates its blocks with several threads straight into its pinned
      // upload buffers (MERYL_BGZF_GENERIC=1: through the generic reader below, as every other compressed input)
      msr_close(r);
      const char *bt = getenv("MERYL_BGZF_THREADS");
      const int rc = mgc_push_text_bgzf_file(s, name.c_str(), 0, bt ? atoi(bt) : std::max(reader_threads, g_inflate_threads));
      if (rc == MGC_EFORMAT) load_on_host(name);
      else if (rc != MGC_OK) die("ERROR: %s", mgc_last_error(s));
      struct stat fst;
      if (stat(name.c_str(), &fst) == 0) total_bases += (uint64_t)fst.st_size * 3;             // (text bytes, as the reference guesses them for .gz)
      continue;
    }
    bool begun = false, refused = false;
    for (;;) {
      const int64_t got = msr_read_text(r, text.data(), text.size());
      if (got < 0) die("ERROR: %s", msr_last_error());
      if (got == 0) break;
      if (!begun) {
        int64_t p = 0;
        while (p < got && (text[p] == '\n' || text[p] == '\r' || text[p] == ' ' || text[p] == '\t')) p++;
        const char c = p < got ? text[p] : '>';
        if (c != '>' && c != '@') {
          fprintf(stderr, "ERROR: '%s' is neither FASTA nor FASTQ (record starts with '%c')\n", name.c_str(), c);
          exit(1);
        }
        if (mgc_begin_text(s, c == '@' ? MGC_TEXT_FASTQ : MGC_TEXT_FASTA) != MGC_OK) die("ERROR: %s", mgc_last_error(s));
        begun = true;
      }
      if (mgc_push_text(s, text.data(), (size_t)got) != MGC_OK) die("ERROR: %s", mgc_last_error(s));
      total_bases += (uint64_t)got;                                                                    // text bytes (reported with -V -V)
    }
    msr_close(r);
    if (begun) {
      const int rc = mgc_end_text(s);
      if (rc == MGC_EFORMAT) refused = true;
      else if (rc != MGC_OK) die("ERROR: %s", mgc_last_error(s));
    }
    if (refused) load_on_host(name);
  }
  return total_bases;
}

// gpus=N: ONE count spread over the node's devices (this build only; the reference's node-scale recipe is "count pieces,
// then union-sum", src/meryl/merylOp-count.C:251-268).  Every rank READS ITS OWN share of the input through its own
// device's link -- plain-text files are cut into byte windows at record boundaries (mgc_text_record_start), inputs that
// cannot be cut (compressed, SAM/BAM, stdin) go whole to one rank each -- and stages it on its device; then
// mgc_count_node_batched: routing plan, per-batch partition / pulls / owner count, parked waves, one merge per owner,
// parts stitched into the 64-file directory.
int run_count_node(const Globals &g, const Operation &op, const mgc_count_config &cfg) {
  if (cfg.count_suffix_length) die("ERROR: %s", "count-suffix= and gpus= cannot be combined.");
  const uint32_t N = g.gpus;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) die("ERROR: %s", "no HIP device.");
  const auto t_start = std::chrono::steady_clock::now();
  // the share of every rank: windows of the plain-text files by bytes, the other inputs round-robin
  std::vector<std::vector<InputPiece>> share(N);
  std::vector<std::pair<std::string, uint64_t>> plain;
  uint64_t plain_total = 0;
  uint32_t rr = 0;
  for (const std::string &name : op.seq_inputs) {
    struct stat fst;
    bool cut = false;
    if (name != "-" && !has_compressed_suffix(name) && stat(name.c_str(), &fst) == 0 && S_ISREG(fst.st_mode) &&
        !(getenv("MERYL_HOST_PARSER") && getenv("MERYL_HOST_PARSER")[0] == '1')) {
      msr_reader *r = msr_open(name.c_str());
      if (!r) die("ERROR: %s", msr_last_error());
      cut = msr_format(r) == MSR_FORMAT_FASTX;
      msr_close(r);
    }
    if (cut) { plain.emplace_back(name, (uint64_t)fst.st_size); plain_total += (uint64_t)fst.st_size; }
    else share[rr++ % N].push_back(InputPiece{name, 0, ~0ull});
  }
  {
    uint64_t file_base = 0;                                   // position of the file in the concatenation of the plain files
    for (const auto &pf : plain) {
      const uint64_t size = pf.second;
      uint64_t prev = 0;
      for (uint32_t r = 0; r < N; r++) {
        // the rank's window of the concatenation, clipped to this file, moved up to the next record start
        const uint64_t hi_cat = (r + 1 == N) ? plain_total : (uint64_t)((unsigned __int128)plain_total * (r + 1) / N);
        uint64_t hi = hi_cat <= file_base ? 0 : std::min<uint64_t>(size, hi_cat - file_base);
        if (hi > 0 && hi < size && mgc_text_record_start(pf.first.c_str(), 0, hi, &hi) != MGC_OK) die("ERROR: %s", mgc_last_error(nullptr));
        if (hi < prev) hi = prev;
        if (hi > prev) share[r].push_back(InputPiece{pf.first, prev, hi});
        prev = hi;
      }
      file_base += size;
    }
  }
  std::vector<mgc_session *> sess(N, nullptr);
  std::vector<int> dev(N, 0);
  std::vector<uint64_t> taken(N, 0);
  const int readers = (int)std::max<uint32_t>(2, std::min<uint32_t>(g.threads, 16) / N);
  for (uint32_t r = 0; r < N; r++) {
    dev[r] = (int)(r % (uint32_t)ndev);
    mgc_count_config rc = cfg;
    rc.homopoly_compress = 0;                                 // the node count compresses every rank's stream itself
    sess[r] = mgc_open(&rc, dev[r]);
    if (!sess[r]) die("ERROR: %s", mgc_last_error(nullptr));
    mgc_set_batch_bases(sess[r], ~0ull >> 2);                // a rank's share is staged whole; the node count batches the counting
  }
  {
    std::vector<std::thread> th;
    for (uint32_t r = 0; r < N; r++) th.emplace_back([&, r] { taken[r] = load_inputs(sess[r], share[r], readers); });
    for (auto &t : th) t.join();
  }
  std::vector<const uint8_t *> ptr(N, nullptr);
  std::vector<uint64_t> len(N, 0);
  uint64_t total_bases = 0;
  for (uint32_t r = 0; r < N; r++) {
    if (mgc_staged_bases(sess[r], &ptr[r], &len[r]) != MGC_OK) die("ERROR: %s", mgc_last_error(sess[r]));
    total_bases += taken[r];
  }
  const auto t_loaded = std::chrono::steady_clock::now();
  if (g.verbosity > 0)
    fprintf(stderr, "\nInput complete.  Counting on %u ranks and writing results to '%s', using %u thread%s.\n",
            N, op.output.c_str(), g.threads, (g.threads == 1) ? "" : "s");
  uint64_t batch = 0;
  if (getenv("MERYL_BATCH_BASES") && *getenv("MERYL_BATCH_BASES")) batch = strtoull(getenv("MERYL_BATCH_BASES"), nullptr, 10);
  mgc_node_profile np;
  if (mgc_count_node_batched(&cfg, N, dev.data(), ptr.data(), len.data(), batch, op.output.c_str(), (int)g.threads, &np) != MGC_OK)
    die("ERROR: %s", mgc_last_error(nullptr));
  const auto t_done = std::chrono::steady_clock::now();
  if (g.verbosity > 2)
    fprintf(stderr, "\nTIMING  read+parse+stage=%.3f s (every rank its own share)   node count+write=%.3f s   (ranks=%u, routing bits=%u, "
                    "batches=%u, partition=%.3f s, exchange+count=%.3f s, merge of parked waves=%.3f s, files=%.3f s, stitch=%.3f s, "
                    "database_bytes=%" PRIu64 ")\n",
            std::chrono::duration<double>(t_loaded - t_start).count(), std::chrono::duration<double>(t_done - t_loaded).count(),
            np.n_ranks, np.bucket_bits, np.n_batches, np.partition_s, np.exchange_count_s, np.merge_runs_s, np.close_s, np.merge_parts_s,
            np.data_bytes);
  for (mgc_session *q : sess) mgc_close(q);
  if (g.verbosity > 0) {
    fprintf(stderr, "\nFinished counting.\n");
    if (g.verbosity > 2)
      fprintf(stderr, "  %" PRIu64 " bases, %" PRIu64 " k-mer instances, %" PRIu64 " distinct k-mers, prefix bits %u.\n",
              total_bases, np.n_instances, np.n_distinct, cfg.w_prefix);
  }
  return 0;
}

int run_count(const Globals &g, const Operation &op) {
  if (g.k == 0) die("ERROR: Kmer size not supplied with modifier k=<kmer-size>.");                    // merylOp-count.C:311-312
  if (op.output.empty() && !g.only_config) die("ERROR: No output specified for count operation.");     // :314-315

  uint64_t exp_num_kmers = op.exp_num_kmers;
  if (exp_num_kmers == 0)                                                                              // :317-318
    for (const std::string &n : op.seq_inputs) exp_num_kmers += msr_guess_number_of_kmers(n.c_str());

  mgc_count_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.k = g.k;
  cfg.mode = (op.kind == OP_COUNT) ? MGC_MODE_CANONICAL : (op.kind == OP_COUNT_FORWARD) ? MGC_MODE_FORWARD : MGC_MODE_REVERSE;
  cfg.n_kmers_estimate = exp_num_kmers;
  cfg.memory_allowed = (uint64_t)(g.memory_gb * 1024.0 * 1024.0 * 1024.0);                            // merylCommandBuilder.C:299-302
  cfg.threads = g.threads;
  bool any_compress = false;
  for (bool c : op.seq_compress) any_compress = any_compress || c;
  for (bool c : op.seq_compress)
    if (c != any_compress) die("ERROR: `compress` must apply to all or none of the inputs of one count in this build.");
  cfg.homopoly_compress = any_compress ? 1 : 0;
  if (op.count_suffix.size() > MGC_MAX_COUNT_SUFFIX) die("ERROR: %s", "count-suffix of more than 32 bases.");
  cfg.count_suffix_length = (uint32_t)op.count_suffix.size();                                         // merylOp.H:139-147
  memcpy(cfg.count_suffix, op.count_suffix.c_str(), op.count_suffix.size());
  cfg.label_size = g.label_size;                                                                      // meryl2: every k-mer of a count
  cfg.label_constant = op.label;                                                                      // carries one constant label
  if (mgc_configure_counting(&cfg) != MGC_OK) die("ERROR: %s", mgc_last_error(nullptr));

  if (g.verbosity > 0) print_configuration(g, op, exp_num_kmers, cfg);
  if (g.only_config) return 0;                                                                         // -C, merylOp-countThreads.C:392-393

  if (g.verbosity > 0)
    fprintf(stderr, "Start counting with %s method.\n", cfg.use_simple ? "SIMPLE" : "THREADED");     // merylOp-nextMer.C:205-209

  if (g.gpus > 1) return run_count_node(g, op, cfg);

  mgc_session *s = mgc_open(&cfg, -1);
  if (!s) die("ERROR: %s", mgc_last_error(nullptr));

  if (getenv("MERYL_BATCH_BASES") && *getenv("MERYL_BATCH_BASES"))                                    // tests: force out-of-core batches
    mgc_set_batch_bases(s, strtoull(getenv("MERYL_BATCH_BASES"), nullptr, 10));
  uint64_t total_bases = 0;
  const auto t_start = std::chrono::steady_clock::now();
  uint64_t text_total = 0;
  for (const std::string &name : op.seq_inputs) text_total += msr_guess_number_of_kmers(name.c_str());
  const bool host_parser = getenv("MERYL_HOST_PARSER") && getenv("MERYL_HOST_PARSER")[0] == '1';
  if (!host_parser && text_total) (void)mgc_reserve_text(s, text_total);
  {
    // bases to expect: a FASTQ file is at most half sequence, anything else at most all of it (compressed: the reference's factors)
    uint64_t expect = 0;
    for (const std::string &name : op.seq_inputs) {
      uint64_t e = msr_guess_number_of_kmers(name.c_str());
      if (name != "-" && !has_compressed_suffix(name)) {
        FILE *f = fopen(name.c_str(), "rb");
        if (f) { const int c = fgetc(f); fclose(f); if (c == '@') e /= 2; }
      }
      expect += e;
    }
    if (expect && cfg.count_suffix_length == 0) (void)mgc_prepare(s, expect + 4096);
  }
  std::vector<InputPiece> pieces;
  for (const std::string &name : op.seq_inputs) pieces.push_back(InputPiece{name, 0, ~0ull});
  g_inflate_threads = (int)g.threads;
  total_bases = load_inputs(s, pieces, (int)std::min<uint32_t>(g.threads, 16));

  const auto t_loaded = std::chrono::steady_clock::now();
  if (g.verbosity > 2) (void)mgc_set_profiling(s, 1);           // -V: the device's own clock of the count beside the wall clock
  if (mgc_count(s) != MGC_OK) die("ERROR: %s", mgc_last_error(s));
  const auto t_counted = std::chrono::steady_clock::now();
  mgc_result_info info;
  mgc_get_result_info(s, &info);

  if (g.verbosity > 0) {
    fprintf(stderr, "\nInput complete.  Writing results to '%s', using %u thread%s.\n",                // :447-448
            op.output.c_str(), g.threads, (g.threads == 1) ? "" : "s");
  }
  mgc_db_write_profile wprof;
  if (mgc_write_database_profiled(s, op.output.c_str(), (int)g.threads, &wprof) != MGC_OK) {
    fprintf(stderr, "ERROR: writing '%s' failed: %s %s\n", op.output.c_str(), mdb_last_error(), mgc_last_error(s));
    exit(1);
  }
  const auto t_written = std::chrono::steady_clock::now();
  mgc_get_result_info(s, &info);                       // (an out-of-core result knows its distinct k-mers only once the runs are merged)
  if (g.verbosity > 2) {                                                                               // -V: where the wall clock went
    auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double>(b - a).count();
    };
    fprintf(stderr, "\nTIMING  read+parse+stage=%.3f s   count=%.3f s   encode+write=%.3f s   (device encode=%.4f s, copy+write=%.3f s, "
                    "database_bytes=%" PRIu64 ")   startup=%.3f s (main -> session open: HIP runtime, device, streams)\n",
            sec(t_start, t_loaded), sec(t_loaded, t_counted), sec(t_counted, t_written), wprof.encode_ms / 1e3,
            wprof.copy_write_s, wprof.data_bytes, sec(g_t_main, t_start));
  }
  if (g.verbosity > 2) {
    mgc_profile cp;
    if (mgc_get_profile(s, &cp) == MGC_OK && cp.total_ms > 0)
      fprintf(stderr, "        count on the device: %.1f ms (histogram %.1f, partition %.1f, grouping passes %.1f, sub-bucket count + packing %.1f); "
                      "the rest of count= is the host's: buffers the input did not size, the first launches, synchronisations\n",
              cp.total_ms, cp.stage_ms[MGC_STAGE_HISTOGRAM], cp.stage_ms[MGC_STAGE_PARTITION], cp.stage_ms[MGC_STAGE_SORT], cp.stage_ms[MGC_STAGE_RLE]);
    if (mgc_get_profile(s, &cp) == MGC_OK && cp.n_batches > 1)
      fprintf(stderr, "        counted in %u batches (memory-full spills, merylOp-countThreads.C:323-379), merged on the device in %.1f ms\n",
              cp.n_batches, cp.merge_ms);
  }
  // The database is complete and its files are closed.  When this count is all the command does, the process ends right away:
  // releasing a 120 GB arena, gigabytes of pinned ring and the runtime in an orderly way takes a good part of a second and gives
  // nothing back that process exit does not (MERYL_FAST_EXIT=0: the orderly way).
  if (!g.fast_exit) mgc_close(s);
  if (g.verbosity > 0) {
    fprintf(stderr, "\nFinished counting.\n");                                                         // :473
    if (g.verbosity > 2)
      fprintf(stderr, "  %" PRIu64 " bases, %" PRIu64 " k-mer instances, %" PRIu64 " distinct k-mers, prefix bits %u.\n",
              total_bases, info.n_instances, info.n_distinct, info.w_prefix);
  }
  return 0;
}

int run_print(const Operation &op) {
  static const char acgt[4] = { 'A', 'C', 'T', 'G' };
  for (const std::string &dbn : op.db_inputs) {
    mdb_reader *r = mdb_reader_open(dbn.c_str());
    if (!r) die("ERROR: %s", mdb_last_error());
    mdb_info info;
    mdb_reader_info(r, &info);
    std::vector<char> kstr(info.k + 1, 0);
    for (uint32_t ff = 0; ff < MGC_NUM_FILES; ff++) {            // ascending == `threads=1 print` order (quick-start.rst:74-77)
      uint64_t *lo = nullptr, *hi = nullptr, *lb = nullptr, n = 0;
      uint32_t *cn = nullptr;
      if (mdb_reader_read_file_ex(r, ff, &lo, &hi, &cn, &lb, &n) != MGC_OK) die("ERROR: %s", mdb_last_error());
      char lbits[72];
      for (uint64_t i = 0; i < n; i++) {
        const unsigned __int128 m = ((unsigned __int128)hi[i] << 64) | lo[i];
        for (uint32_t b = 0; b < info.k; b++) kstr[b] = acgt[(unsigned)(m >> (2 * (info.k - 1 - b))) & 3];
        if (info.label_size == 0) {
          fprintf(stdout, "%s\t%u\n", kstr.data(), cn[i]);                                             // merylOp-nextMer.C:673-676
        } else {                                                                                       // meryl2: + the label in binary
          for (uint32_t b = 0; b < info.label_size; b++) lbits[b] = ((lb[i] >> (info.label_size - 1 - b)) & 1) ? '1' : '0';
          lbits[info.label_size] = 0;                                                                  // (src/meryl2/merylOp-nextMer.C:36-43)
          fprintf(stdout, "%s\t%u\t%s\n", kstr.data(), cn[i], lbits);
        }
      }
      mdb_free(lo); mdb_free(hi); mdb_free(cn); mdb_free(lb);
    }
    mdb_reader_close(r);
  }
  return 0;
}

// `meryl histogram <db>`: value <TAB> number of distinct k-mers with that value, ascending -- the histogram the count
// path's writer stored in the master index (src/meryl/merylOp-histogram.C:20-44, quick-start.rst:144)
int run_histogram(const Operation &op) {
  if (op.db_inputs.size() != 1) die("ERROR: told to dump a histogram for more than one input!");
  mdb_reader *r = mdb_reader_open(op.db_inputs[0].c_str());
  if (!r) die("ERROR: %s", mdb_last_error());
  mdb_info i;
  mdb_reader_info(r, &i);
  std::vector<uint64_t> v(i.hist_len), o(i.hist_len);
  if (i.hist_len && mdb_reader_histogram(r, v.data(), o.data()) != 0) die("ERROR: %s", mdb_last_error());
  for (uint64_t ii = 0; ii < i.hist_len; ii++) fprintf(stdout, "%" PRIu64 "\t%" PRIu64 "\n", v[ii], o[ii]);
  mdb_reader_close(r);
  return 0;
}

// `meryl dumpFile <db>/0x######`: the three tables of documentation/source/usage.rst:24-45 (src/meryl/meryl.C:41-45) --
// the file's index, every block's header, every k-mer as stored (unary prefix delta, accumulated prefix, the two halves
// of the binary remainder, value).
int run_dump_file(const Operation &op) {
  for (const std::string &arg : op.db_inputs) {
    const size_t slash = arg.find_last_of('/');
    const std::string dbn = (slash == std::string::npos) ? "." : arg.substr(0, slash);
    std::string base = (slash == std::string::npos) ? arg : arg.substr(slash + 1);
    const size_t dot = base.find('.');
    if (dot != std::string::npos) base.erase(dot);                                                    // 0x000000.merylData is fine too
    if (base.size() != 8 || base.compare(0, 2, "0x") != 0) die("ERROR: dumpFile wants <database>/0x###### (six binary digits), not '%s'.", arg.c_str());
    uint32_t ff = 0;
    for (int i = 2; i < 8; i++) { if (base[i] != '0' && base[i] != '1') die("ERROR: bad file name '%s'.", arg.c_str()); ff = (ff << 1) | (uint32_t)(base[i] - '0'); }
    mdb_reader *r = mdb_reader_open(dbn.c_str());
    if (!r) die("ERROR: %s", mdb_last_error());
    mdb_info info;
    mdb_reader_info(r, &info);
    const uint64_t nblocks = (uint64_t)1 << info.num_blocks_bits;
    std::vector<mdb_index_entry> idx(nblocks);
    if (mdb_reader_file_index(r, ff, idx.data()) != MGC_OK) die("ERROR: %s", mdb_last_error());
    fprintf(stdout, "\n    prefix    blkPos    nKmers\n---------- --------- ---------\n");
    for (const mdb_index_entry &e : idx) fprintf(stdout, "0x%08" PRIx64 " %9" PRIu64 " %9" PRIu64 "\n", e.prefix, e.position, e.n_kmers);
    fprintf(stdout, "\n            prefix   nKmers kCode uBits bBits                 k1 cCode                 c1                 c2\n"
                    "------------------ -------- ----- ----- ----- ------------------ ----- ------------------ ------------------\n");
    for (const mdb_index_entry &e : idx) {
      mdb_block_header h;
      if (mdb_reader_block_header(r, ff, e.position, &h) != MGC_OK) die("ERROR: %s", mdb_last_error());
      if (h.prefix != e.prefix) continue;                                                              // an index slot no block was written for
      fprintf(stdout, "0x%016" PRIx64 " %8" PRIu64 " %5u %5u %5u 0x%016" PRIx64 " %5u 0x%016" PRIx64 " 0x%016" PRIx64 "\n",
              h.prefix, h.n_kmers, h.k_code, h.unary_bits, h.binary_bits, h.k1, h.c_code, h.c1, h.c2);
    }
    fprintf(stdout, "\n kmerIdx prefixDelta      prefix |--- suffix-size and both suffixes ---|    value\n"
                    "-------- ----------- ----------- -- ---------------- -- ---------------- --------\n");
    for (const mdb_index_entry &e : idx) {
      mdb_block_header h;
      uint64_t *pd = nullptr, *tp = nullptr, *rh = nullptr, *rl = nullptr;
      uint32_t *vv = nullptr;
      if (mdb_reader_read_block_raw(r, ff, e.position, &h, &pd, &tp, &rh, &rl, &vv) != MGC_OK) die("ERROR: %s", mdb_last_error());
      if (h.prefix != e.prefix) h.n_kmers = 0;
      const uint32_t hi_bits = h.binary_bits > 64 ? h.binary_bits - 64 : 0, lo_bits = h.binary_bits > 64 ? 64 : h.binary_bits;
      for (uint64_t i = 0; i < h.n_kmers; i++)
        fprintf(stdout, "%8" PRIu64 " %11" PRIu64 " %011" PRIx64 " %2u %016" PRIx64 " %2u %016" PRIx64 " %8u\n",
                i, pd[i], tp[i], hi_bits, rh[i], lo_bits, rl[i], vv[i]);
      mdb_free(pd); mdb_free(tp); mdb_free(rh); mdb_free(rl); mdb_free(vv);
    }
    mdb_reader_close(r);
  }
  return 0;
}

int run_dump_index(const Operation &op) {
  for (const std::string &dbn : op.db_inputs) {
    mdb_reader *r = mdb_reader_open(dbn.c_str());
    if (!r) die("ERROR: %s", mdb_last_error());
    mdb_info i;
    mdb_reader_info(r, &i);
    fprintf(stdout, "Opened '%s'.\n", dbn.c_str());                                                    // shape of usage.rst:13-19
    fprintf(stdout, "  kmerSize       %u\n", i.k);
    fprintf(stdout, "  prefixSize     %u\n", i.prefix_size);
    fprintf(stdout, "  suffixSize     %u\n", i.suffix_size);
    fprintf(stdout, "  numFilesBits   %u (%u files)\n", i.num_files_bits, 1u << i.num_files_bits);
    fprintf(stdout, "  numBlocksBits  %u (%u blocks)\n", i.num_blocks_bits, 1u << i.num_blocks_bits);
    if (i.label_size) fprintf(stdout, "  labelSize      %u\n", i.label_size);
    fprintf(stdout, "  unique         %" PRIu64 "\n  distinct       %" PRIu64 "\n  total          %" PRIu64 "\n",
            i.num_unique, i.num_distinct, i.num_total);
    mdb_reader_close(r);
  }
  return 0;
}

// `union-sum a.meryl b.meryl [count ... output c.meryl] output u.meryl` and the five related operations
// (merylOp-nextMer.C:560-612): the inputs -- databases named on the command line, and the outputs of child operations,
// which ran first (meryl.C:211-227 turns a finished count into a pass-through over its new database) -- are merged on
// the device, file slice by file slice (mgc_db_merge).
int run_merge(const Globals &g, const std::vector<Operation> &ops, const Operation &op) {
  if (op.output.empty()) die("ERROR: operation '%s' needs an 'output <database>' in this build.", op.word.c_str());
  std::vector<std::string> inputs;
  for (const InputRef &in : op.inputs) inputs.push_back(in.child >= 0 ? ops[in.child].output : in.path);
  if (inputs.empty()) die("ERROR: operation '%s' has no inputs.", op.word.c_str());
  for (const std::string &n : inputs) if (!dir_has_index(n)) die("ERROR: input '%s' is not a meryl database.", n.c_str());
  std::vector<const char *> names;
  for (const std::string &n : inputs) names.push_back(n.c_str());
  if (g.verbosity > 0) {
    fprintf(stderr, "\nPROCESSING %s of %zu database%s into '%s'.\n", op.word.c_str(), inputs.size(), inputs.size() == 1 ? "" : "s", op.output.c_str());
    for (const std::string &n : inputs) fprintf(stderr, "  %15s: %s\n", "database", n.c_str());
  }
  if (mgc_db_merge(names.data(), (uint32_t)names.size(), op.merge_op, op.output.c_str(), -1, (int)g.threads) != MGC_OK)
    die("ERROR: %s", mgc_db_stream_error(nullptr));
  return 0;
}

// `less-than 5 a.meryl output b.meryl`, `divide 2 [count ...] output h.meryl`: the single-input operations
// (merylOp-nextMer.C:490-557).  distinct=<f> / word-frequency=<f> turn into a threshold from the input's stored histogram
// (initializeThreshold, :65-118).
int run_value(const Globals &g, const std::vector<Operation> &ops, const Operation &op) {
  if (op.output.empty()) die("ERROR: operation '%s' needs an 'output <database>' in this build.", op.word.c_str());
  if (op.inputs.size() != 1) die("ERROR: operation '%s' takes exactly one input.", op.word.c_str());
  const std::string in = op.inputs[0].child >= 0 ? ops[op.inputs[0].child].output : op.inputs[0].path;
  if (!dir_has_index(in)) die("ERROR: input '%s' is not a meryl database.", in.c_str());
  uint64_t c = op.constant;
  if (op.frac_distinct >= 0 || op.word_freq >= 0) {
    mdb_reader *r = mdb_reader_open(in.c_str());
    if (!r) die("ERROR: %s", mdb_last_error());
    mdb_info info;
    mdb_reader_info(r, &info);
    std::vector<uint64_t> hv(info.hist_len), ho(info.hist_len);
    if (info.hist_len) mdb_reader_histogram(r, hv.data(), ho.data());
    mdb_reader_close(r);
    if (op.frac_distinct >= 0) {                                             // :104-114
      const uint64_t target = (uint64_t)(op.frac_distinct * (double)info.num_distinct);
      uint64_t n = 0;
      for (uint64_t i = 0; i < info.hist_len; i++) { n += ho[i]; if (n >= target) { c = hv[i]; break; } }
    }
    if (op.word_freq >= 0) c = (uint64_t)(op.word_freq * (double)info.num_total);   // :116-118
  }
  if (c == ~0ull) die("ERROR: operation '%s' needs a number (threshold / constant).", op.word.c_str());
  if (g.verbosity > 0) fprintf(stderr, "\nPROCESSING %s %" PRIu64 " of '%s' into '%s'.\n", op.word.c_str(), c, in.c_str(), op.output.c_str());
  if (mgc_db_filter(in.c_str(), op.value_op, c, op.output.c_str(), -1, (int)g.threads) != MGC_OK) die("ERROR: %s", mgc_db_stream_error(nullptr));
  return 0;
}

}  // namespace

int main(int argc, char **argv) {
  Globals g;
  std::vector<Operation> ops;
  std::vector<int> stack;            // merylCommandBuilder's _opStack: words attach to the operation on top
  bool expect_output_name = false;

  if (argc < 2) { usage(argv[0]); return 1; }

  auto top = [&]() -> int { return stack.empty() ? -1 : stack.back(); };
  auto is_counting = [&](int i) { return i >= 0 && ops[i].kind >= OP_COUNT && ops[i].kind <= OP_COUNT_REVERSE; };
  auto ensure_top = [&]() {                                                   // initialize(), :163-174: an empty root operation
    if (stack.empty()) { ops.emplace_back(); stack.push_back((int)ops.size() - 1); }
  };

  for (int a = 1; a < argc; a++) {
    std::string w = argv[a];
    // one leading '[' is ignored, trailing ']'s pop operations off the stack once the word is processed
    // (merylCommandBuilder.C:86-93,129-157)
    if (!w.empty() && w[0] == '[') w.erase(0, 1);
    int closing = 0;
    while (!w.empty() && w.back() == ']') { w.pop_back(); closing++; }

    if (!w.empty()) {
      const size_t eq = w.find('=');
      const std::string key = (eq == std::string::npos) ? w : w.substr(0, eq);
      const std::string val = (eq == std::string::npos) ? "" : w.substr(eq + 1);
      const int merge_code = (w == "union-sum") ? MGC_MERGE_UNION_SUM : (w == "union-min") ? MGC_MERGE_UNION_MIN :
                             (w == "union-max") ? MGC_MERGE_UNION_MAX : (w == "intersect-sum") ? MGC_MERGE_INTERSECT_SUM :
                             (w == "intersect-min") ? MGC_MERGE_INTERSECT_MIN : (w == "intersect-max") ? MGC_MERGE_INTERSECT_MAX :
                             (w == "union") ? MGC_MERGE_UNION : (w == "intersect") ? MGC_MERGE_INTERSECT :
                             (w == "subtract") ? MGC_MERGE_SUBTRACT : (w == "difference") ? MGC_MERGE_DIFFERENCE :
                             (w == "symmetric-difference") ? MGC_MERGE_SYMMETRIC_DIFFERENCE : -1;                 // merylCommandBuilder.C:364-378
      const int value_code = (w == "less-than") ? MGC_VALUE_LESS_THAN : (w == "greater-than") ? MGC_VALUE_GREATER_THAN :
                             (w == "at-least") ? MGC_VALUE_AT_LEAST : (w == "at-most") ? MGC_VALUE_AT_MOST :
                             (w == "equal-to") ? MGC_VALUE_EQUAL_TO : (w == "not-equal-to") ? MGC_VALUE_NOT_EQUAL_TO :
                             (w == "increase") ? MGC_VALUE_INCREASE : (w == "decrease") ? MGC_VALUE_DECREASE :
                             (w == "multiply") ? MGC_VALUE_MULTIPLY : (w == "divide") ? MGC_VALUE_DIVIDE :
                             (w == "divide-round") ? MGC_VALUE_DIVIDE_ROUND : (w == "modulo") ? MGC_VALUE_MODULO : -1;   // :350-362
      const bool is_number = !w.empty() && w.find_first_not_of("0123456789") == std::string::npos;

      if (expect_output_name) {                                             // `output <db>`, :440-461
        if (top() < 0 || ops[top()].kind == OP_NONE) die("ERROR: 'output' without an operation.");
        if (!ops[top()].output.empty()) die("ERROR: operation already has an output ('%s').", ops[top()].output.c_str());   // merylOp.C:256-257
        ops[top()].output = w;
        expect_output_name = false;
      }
      // a bare number is the threshold / constant of the value operation on top ("greater-than 45", "divide 2"; :216-233)
      // (the reference takes an all-digits word as the number whenever the operation wants one, before it is ever looked at
      // as a file name -- a database directory called "5" cannot be an input of `less-than`)
      else if (is_number && top() >= 0 && ops[top()].kind == OP_VALUE && ops[top()].constant == ~0ull) {
        ops[top()].constant = strtoull(w.c_str(), nullptr, 10);
      }
      // threshold= / t= set the THRESHOLD only (setThreshold, :291-294): the arithmetic operations' constant is not touched
      else if ((key == "threshold" || key == "t") && eq != std::string::npos && top() >= 0 && ops[top()].kind == OP_VALUE) {
        if (ops[top()].value_op <= MGC_VALUE_NOT_EQUAL_TO) ops[top()].constant = strtoull(val.c_str(), nullptr, 10);
      }
      else if ((key == "distinct" || key == "d") && eq != std::string::npos && top() >= 0 && ops[top()].kind == OP_VALUE && ops[top()].value_op <= MGC_VALUE_NOT_EQUAL_TO) {
        ops[top()].frac_distinct = strtod(val.c_str(), nullptr);                                                    // :278-282
      }
      else if ((key == "word-frequency" || key == "f") && eq != std::string::npos && top() >= 0 && ops[top()].kind == OP_VALUE && ops[top()].value_op <= MGC_VALUE_NOT_EQUAL_TO) {
        ops[top()].word_freq = strtod(val.c_str(), nullptr);                                                        // :284-288
      }
      // ---- options, merylCommandBuilder.C:187-326 ----
      else if (w.compare(0, 2, "-V") == 0)   { g.verbosity += (int)w.size() - 1; }
      else if (w == "-Q")                    { g.verbosity = 0; }
      else if (w == "-P")                    { /* progress: accepted, unused by count (:206-209) */ }
      else if (w == "-C")                    { g.only_config = true; }
      else if (w == "-l") {                                                  // meryl2 global: label width in bits
        if (a + 1 >= argc) die("ERROR: -l needs the label size in bits.");
        g.label_size = (uint32_t)strtoul(argv[++a], nullptr, 10);
        if (g.label_size > 64) die("ERROR: label size of more than 64 bits.");
      }
      else if (key == "label" && eq != std::string::npos) {                  // label=#<n>: the count's constant label
        if (!is_counting(top())) die("ERROR: option '%s' needs a counting operation before it.", w.c_str());
        if (val.empty() || val[0] != '#') die("ERROR: a count takes a constant label, label=#<integer>, not '%s'.", w.c_str());
        ops[top()].label = strtoull(val.c_str() + 1, nullptr, 0);
      }
      else if (key == "k" && eq != std::string::npos) {
        const uint32_t k = (uint32_t)strtoul(val.c_str(), nullptr, 10);
        if (k == 0 || k > 64) die("ERROR: k=%s is not a valid k-mer size (1..64).", val.c_str());
        if (g.k != 0 && g.k != k) die("ERROR: kmer size already set; cannot change it to '%s'.", val.c_str());   // :254-262
        g.k = k;
      }
      else if (key == "n" && eq != std::string::npos) {                     // :265-268: the operation on top
        ensure_top();
        ops[top()].exp_num_kmers = strtoull(val.c_str(), nullptr, 10);
      }
      else if (key == "memory" && eq != std::string::npos)  { g.memory_gb = strtod(val.c_str(), nullptr); }       // :299-302
      else if (key == "gpus" && eq != std::string::npos) {                                  // this build only (no reference counterpart)
        g.gpus = (uint32_t)strtoul(val.c_str(), nullptr, 10);
        if (g.gpus < 1 || g.gpus > 64) die("ERROR: %s", "gpus= takes 1..64.");
      }
      else if (key == "threads" && eq != std::string::npos) { g.threads = (uint32_t)strtoul(val.c_str(), nullptr, 10); if (!g.threads) g.threads = 1; }   // :306-310
      else if (w == "compress")              { g.compress = true; }                                               // :237-240
      else if (key == "count-suffix") {                                                                            // :271-272
        if (!is_counting(top())) die("ERROR: option '%s' needs a counting operation before it.", w.c_str());
        ops[top()].count_suffix = val;
      }
      else if (key == "segment") { die("ERROR: option '%s' (Canu sequence stores) is not supported in this build.", w.c_str()); }
      else if (top() >= 0 && ops[top()].kind == OP_DUMPFILE && ops[top()].inputs.empty()) {
        InputRef in; in.path = w;
        ops[top()].inputs.push_back(in);                                     // <database>/0x######
      }
      // ---- operations, :346-439 ----
      else if (w == "count" || w == "count-forward" || w == "count-reverse" || w == "print" || w == "dumpIndex" || w == "histogram" ||
               w == "dumpFile" || merge_code >= 0 || value_code >= 0) {
        const OpKind kind = (w == "count") ? OP_COUNT : (w == "count-forward") ? OP_COUNT_FORWARD :
                            (w == "count-reverse") ? OP_COUNT_REVERSE : (w == "print") ? OP_PRINT :
                            (w == "histogram") ? OP_HISTOGRAM : (w == "dumpFile") ? OP_DUMPFILE :
                            (w == "dumpIndex") ? OP_DUMPINDEX : (merge_code >= 0) ? OP_MERGE : OP_VALUE;
        ensure_top();
        if (is_counting(top())) { stack.pop_back(); ensure_top(); }         // :391-407: a counting operation takes no operation as input
        if (ops[top()].kind != OP_NONE) {                                    // :412-421: a new operation, input of the one on top
          const int parent = top();
          ops.emplace_back();
          const int me = (int)ops.size() - 1;
          if (ops[parent].kind == OP_DUMPINDEX || ops[parent].kind == OP_DUMPFILE)
            die("ERROR: operation '%s' cannot be the input of a debug operation.", w.c_str());
          InputRef in; in.child = me;
          ops[parent].inputs.push_back(in);
          ops[me].parent = parent;
          stack.push_back(me);
        }
        ops[top()].kind = kind;                                              // :422-431 (or it replaces the empty operation on top)
        ops[top()].word = w;
        ops[top()].merge_op = merge_code;
        ops[top()].value_op = value_code;
      }
      else if (w == "output")                { expect_output_name = true; }
      else if (w == "statistics" || w == "printACGT" || w == "compare" || w == "noise" || w == "ploidy") {
        die("ERROR: operation '%s' is not part of this build (count, the set operations and the value filters / arithmetic only).", w.c_str());
      }
      // ---- inputs ----
      else if (dir_has_index(w)) {                                           // :159,506-514
        const int t = top();
        if (t < 0 || (ops[t].kind != OP_PRINT && ops[t].kind != OP_DUMPINDEX && ops[t].kind != OP_HISTOGRAM && ops[t].kind != OP_MERGE && ops[t].kind != OP_VALUE))
          die("ERROR: database input '%s' needs a print, histogram, dumpIndex, set or value operation before it.", w.c_str());
        InputRef in; in.path = w;
        ops[t].inputs.push_back(in);
      }
      else if (file_exists(w) || w == "-") {                                 // :537-549: only counting ops take sequence
        if (!is_counting(top())) die("ERROR: sequence file '%s' supplied to a non-counting operation.", w.c_str());
        ops[top()].seq_inputs.push_back(w);
        ops[top()].seq_compress.push_back(g.compress);
      }
      else {                                                                  // meryl.C:84-86
        fprintf(stderr, "\nCan't interpret '%s': not a meryl command, option, or recognized input file.\n\n", argv[a]);
        usage(argv[0]);
        return 1;
      }
    }
    for (; closing > 0 && !stack.empty(); closing--) stack.pop_back();       // terminateOperation(), :86-93
  }
  if (expect_output_name) die("ERROR: 'output' needs a database name.");

  if (g.verbosity > 0) {
    size_t n_trees = 0;
    for (const Operation &op : ops) if (op.kind != OP_NONE && op.parent < 0) n_trees++;
    fprintf(stderr, "\nFound %zu command tree%s.\n", n_trees, (n_trees == 1) ? "" : "s");              // meryl.C:181
  }

  int rc = 0;
  {
    size_t n_ops = 0, n_counts = 0;
    for (const Operation &op : ops) if (op.kind != OP_NONE) { n_ops++; if (op.kind >= OP_COUNT && op.kind <= OP_COUNT_REVERSE) n_counts++; }
    const char *fe = getenv("MERYL_FAST_EXIT");
    g.fast_exit = n_ops == 1 && n_counts == 1 && !g.only_config && g.gpus <= 1 && !(fe && fe[0] == '0');
  }
  for (const Operation &op : ops) {                                           // counting ops first, in list order (meryl.C:211-227)
    if (op.kind >= OP_COUNT && op.kind <= OP_COUNT_REVERSE) {
      if (op.seq_inputs.empty() && !g.only_config) die("ERROR: count operation has no sequence inputs.");
      rc |= run_count(g, op);
    }
  }
  if (g.only_config) {                                                        // -C: configure, report, stop (merylOp-countThreads.C:392-393)
    if (g.verbosity > 0) fprintf(stderr, "\nCleaning up.\n\nBye.\n");
    return rc;
  }
  // the other operations bottom-up: children were appended after their parents, so reverse list order runs a child
  // before the operation that reads its output
  for (size_t i = ops.size(); i-- > 0;)
    if (ops[i].kind == OP_MERGE) rc |= run_merge(g, ops, ops[i]);
    else if (ops[i].kind == OP_VALUE) rc |= run_value(g, ops, ops[i]);
  for (Operation &op : ops) {
    if (op.kind != OP_PRINT && op.kind != OP_DUMPINDEX && op.kind != OP_HISTOGRAM && op.kind != OP_DUMPFILE) continue;
    for (const InputRef &in : op.inputs) {
      if (in.child >= 0 && ops[in.child].output.empty()) die("ERROR: the input operation of '%s' needs an output in this build.", op.word.c_str());
      op.db_inputs.push_back(in.child >= 0 ? ops[in.child].output : in.path);
    }
    if (op.kind == OP_PRINT)     rc |= run_print(op);
    if (op.kind == OP_DUMPINDEX) rc |= run_dump_index(op);
    if (op.kind == OP_HISTOGRAM) rc |= run_histogram(op);
    if (op.kind == OP_DUMPFILE)  rc |= run_dump_file(op);
  }
  if (g.verbosity > 0) fprintf(stderr, "\nCleaning up.\n\nBye.\n");                                    // meryl.C:268,273
  if (g.fast_exit) { fflush(nullptr); _exit(rc); }                            // (every output file is closed; see run_count)
  return rc;
}
