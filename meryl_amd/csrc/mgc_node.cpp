// mgc_node.cpp -- one count spread over the GPUs of a node from ONE process (include/meryl_db.h, mgc_count_node).
//
// What it replaces: the reference has no multi-device form of `count`; a node-scale run is its "split the input, count
// the pieces, union-sum" recipe (src/meryl/merylOp-count.C:251-268, the segment n/m option) or one big-memory countThreads
// (src/meryl/merylOp-countThreads.C:292-476).  Here the k-mer space is what gets split: every rank extracts the k-mers of
// ITS reads, the k-mers travel to the rank that owns their range of the key space, and the owners count and write
// disjoint prefix ranges of ONE database -- no union-sum pass, files byte-identical to a single-device count.
//
// This is the in-process sibling of meryl_amd/count.py:count_sharded (one process per GPU, RCCL): same routing plan
// (bucket = top `bits` bits of the k-mer, balanced contiguous bucket ranges, ~16 waves), but the exchange is the owner
// PULLING its pieces out of the other devices' partition buffers with peer copies over xGMI -- one host thread per
// rank, a thread barrier where the RCCL version has its all-gather.  Several ranks may share a device (tests on a
// one-GPU box run the whole plan that way).
#include "../../include/meryl_db.h"
#include "mgc_session.hpp"
#include "mgc_runs.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

using mgc::set_err;

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Barrier {
  std::mutex m; std::condition_variable cv; uint32_t n = 1, waiting = 0; uint64_t gen = 0;
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    const uint64_t g = gen;
    if (++waiting == n) { waiting = 0; gen++; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g; });
  }
};

struct Rank {
  int device = 0;
  const uint8_t *d_bases = nullptr; uint64_t n_bases = 0;   // this rank's reads (after `compress`: d_own)
  void *d_own = nullptr;                        // the homopolymer-compressed copy, if one was made
  std::vector<uint64_t> total;                  // per bucket: k-mers of ALL of this rank's reads (the routing plan's input)
  void *d_keys = nullptr;                       // the current batch's k-mers, bucket-major (read by the owners' peer copies)
  std::vector<uint64_t> counts, off;            // per bucket, current batch: k-mers, start (in k-mers) inside d_keys
  uint64_t n_distinct = 0, n_instances = 0;
  mgc_db_write_profile wp{};
  mgc_runs_profile rp{};
  double t_partition = 0, t_count = 0, t_close = 0, t_merge = 0;
  // owner side, alive over all batches
  mgc_session *sess = nullptr;
  mgc_db_stream *ds = nullptr;
  mgc_runs *runs = nullptr;                     // batches > 1: every counted wave is parked here, merged when the last batch is done
};

struct Node {
  mgc_count_config cfg{};
  uint32_t n = 1, bits = 6, kw = 1;
  std::string path; int host_threads = 8;
  uint64_t batch_bases = 0;                     // bases of one rank's batch; 0: derive from the free HBM
  uint32_t n_batches = 1;
  std::vector<Rank> ranks;
  std::vector<uint32_t> cuts;                   // n + 1 bucket cut points
  Barrier bar;
  std::atomic<bool> failed{false};
  std::mutex err_m; std::string err;
  void fail(const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    std::lock_guard<std::mutex> lk(err_m);
    if (!failed.exchange(true)) err = buf;
  }
};

// meryl_amd/count.py:shard_bucket_bits
uint32_t bucket_bits_for(uint32_t n_ranks, uint32_t k, uint64_t max_rank_bases, uint32_t w_prefix) {
  uint32_t extra = 0;
  while ((1u << extra) < n_ranks) extra++;
  while (extra < 4 && (((uint64_t)n_ranks * max_rank_bases) >> (6 + extra)) > 180000000ull) extra++;
  if (const char *e = getenv("MGC_SHARD_BITS")) extra = (uint32_t)std::max(6, atoi(e)) - 6;
  uint32_t bits = std::min<uint32_t>({10u, 6 + extra, 2 * k});
  bits = std::max(bits, std::min(6u, 2 * k));
  return std::min(bits, w_prefix);
}

// meryl_amd/count.py:balanced_file_ranges -- contiguous bucket ranges with about the same number of k-mers each
std::vector<uint32_t> balanced_ranges(const std::vector<uint64_t> &total, uint32_t n) {
  const uint32_t nb = (uint32_t)total.size();
  std::vector<double> cum(nb + 1, 0.0);
  for (uint32_t b = 0; b < nb; b++) cum[b + 1] = cum[b] + (double)total[b];
  std::vector<uint32_t> cuts{0};
  for (uint32_t r = 1; r < n; r++) {
    const uint32_t lo = cuts.back() + 1, hi = nb - (n - r);
    const double target = cum[nb] * r / n;
    uint32_t best = lo;
    for (uint32_t c = lo; c <= hi; c++) if (std::fabs(cum[c] - target) < std::fabs(cum[best] - target)) best = c;
    cuts.push_back(best);
  }
  cuts.push_back(nb);
  return cuts;
}

#define ND_HIP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { nd.fail("rank %u: %s -> %s", r, #expr, hipGetErrorString(e__)); return; } } while (0)
#define ND_MGC(expr, sess) do { int rc__ = (expr); if (rc__ != MGC_OK) { nd.fail("rank %u: %s -> %d %s", r, #expr, rc__, mgc_last_error(sess)); return; } } while (0)

struct DevMem {                                 // frees what a phase allocated, whichever way the phase ends
  std::vector<void *> p;
  ~DevMem() { for (void *q : p) if (q) (void)hipFree(q); }
  hipError_t alloc(void **out, size_t bytes) { hipError_t e = hipMalloc(out, std::max<size_t>(bytes, 256)); if (e == hipSuccess) p.push_back(*out); return e; }
};

// once per rank: `compress` (the rank's whole sequences), the bucket histogram of ALL its reads -- every rank needs the same
// routing plan for every batch, so the plan comes from the whole input, not from the first batch
void phase_prepare(Node &nd, uint32_t r, hipStream_t st) {
  Rank &me = nd.ranks[r];
  const uint32_t nbk = 1u << nd.bits;
  DevMem tmp;
  if (nd.cfg.homopoly_compress && me.n_bases) {
    void *ws = nullptr;
    const size_t wsb = mgc_dev_homopoly_workspace_bytes(me.n_bases);
    ND_HIP(hipMalloc(&me.d_own, me.n_bases)); ND_HIP(tmp.alloc(&ws, wsb));
    uint64_t n_out = 0;
    ND_MGC(mgc_dev_homopoly_compress(me.d_bases, me.n_bases, (uint8_t *)me.d_own, &n_out, ws, wsb, st), nullptr);
    me.d_bases = (const uint8_t *)me.d_own; me.n_bases = n_out;
  }
  const size_t wsb = mgc_dev_partition_workspace_bytes(nd.bits);
  void *ws = nullptr, *d_counts = nullptr;
  ND_HIP(tmp.alloc(&ws, wsb)); ND_HIP(tmp.alloc(&d_counts, sizeof(uint64_t) * nbk));
  ND_MGC(mgc_dev_kmer_histogram(me.d_bases, me.n_bases, nd.cfg.k, nd.cfg.mode, nd.bits, (uint64_t *)d_counts, ws, wsb, st), nullptr);
  me.total.assign(nbk, 0);
  ND_HIP(hipMemcpyAsync(me.total.data(), d_counts, sizeof(uint64_t) * nbk, hipMemcpyDeviceToHost, st));
  ND_HIP(hipStreamSynchronize(st));
  for (uint32_t b = 0; b < nbk; b++) me.n_instances += me.total[b];
}

// Batch b of a rank's reads: [cut_b - (k-1), cut_{b+1}) of its base stream -- a window that starts in the last k-1 bases of
// batch b-1 is incomplete there and complete here, so the cut may fall anywhere, also inside a read, and no k-mer is lost or
// counted twice (the rule mgc_count_node_staged cuts its rank slices by).
void batch_range(const Node &nd, const Rank &me, uint32_t b, uint64_t *a, uint64_t *len) {
  const uint64_t n = me.n_bases, k = nd.cfg.k;
  const uint64_t cut = (uint64_t)((unsigned __int128)n * b / nd.n_batches), end = (uint64_t)((unsigned __int128)n * (b + 1) / nd.n_batches);
  *a = (b && cut >= k - 1) ? cut - (k - 1) : 0;     // cut < k-1: the earlier batches are shorter than a k-mer, hold no window
  *len = end - *a;
  if (b && cut < k - 1 && end < k) *len = 0;       // (still no complete window)
}

void phase_partition(Node &nd, uint32_t r, uint32_t batch, hipStream_t st) {
  Rank &me = nd.ranks[r];
  const uint32_t nbk = 1u << nd.bits;
  const double t0 = now_s();
  DevMem tmp;
  uint64_t a = 0, nb = 0;
  batch_range(nd, me, batch, &a, &nb);
  const uint8_t *bases = me.d_bases + a;
  const size_t wsb = mgc_dev_partition_workspace_bytes(nd.bits);
  void *ws = nullptr, *d_counts = nullptr;
  ND_HIP(tmp.alloc(&ws, wsb)); ND_HIP(tmp.alloc(&d_counts, sizeof(uint64_t) * nbk));
  ND_MGC(mgc_dev_kmer_histogram(bases, nb, nd.cfg.k, nd.cfg.mode, nd.bits, (uint64_t *)d_counts, ws, wsb, st), nullptr);
  me.counts.assign(nbk, 0);
  ND_HIP(hipMemcpyAsync(me.counts.data(), d_counts, sizeof(uint64_t) * nbk, hipMemcpyDeviceToHost, st));
  ND_HIP(hipStreamSynchronize(st));
  me.off.assign(nbk + 1, 0);
  for (uint32_t b = 0; b < nbk; b++) me.off[b + 1] = me.off[b] + me.counts[b];
  ND_HIP(hipMemcpyAsync(d_counts, me.off.data(), sizeof(uint64_t) * nbk, hipMemcpyHostToDevice, st));     // now the starts
  ND_HIP(hipMalloc(&me.d_keys, std::max<uint64_t>(me.off[nbk] * nd.kw * 8, 256)));
  ND_MGC(mgc_dev_kmer_partition(bases, nb, nd.cfg.k, nd.cfg.mode, nd.bits, (const uint64_t *)d_counts, me.d_keys, ws, wsb, st), nullptr);
  ND_HIP(hipStreamSynchronize(st));
  me.t_partition += now_s() - t0;
}

// owner side, once: the session that counts the waves, this rank's part of the database, the run store (several batches)
void owner_open(Node &nd, uint32_t r) {
  Rank &me = nd.ranks[r];
  mgc_count_config cfg = nd.cfg;
  cfg.homopoly_compress = 0;                    // the owner side sees k-mers, never bases
  me.sess = mgc_open(&cfg, me.device);
  if (!me.sess) { nd.fail("rank %u: mgc_open: %s", r, mgc_last_error(nullptr)); return; }
  me.ds = mgc_db_stream_open(nd.path.c_str(), cfg.k, cfg.w_prefix, cfg.label_size, cfg.label_constant, r, nd.n, nd.host_threads, me.device);
  if (!me.ds) { nd.fail("rank %u: %s", r, mgc_db_stream_error(nullptr)); return; }
  if (nd.n_batches > 1) {
    // the parked waves share the device with a batch's k-mers, inbox and count arena: a third of what is free now may stay
    // in HBM, the rest goes to pinned host DRAM (MGC_OOC_BUDGET overrides)
    size_t free_b = 0, total_b = 0;
    const uint64_t budget = (hipMemGetInfo(&free_b, &total_b) == hipSuccess) ? (uint64_t)free_b / 3 : 0;
    me.runs = mgc_runs_open(cfg.k, cfg.w_prefix, me.device, budget, 0);
    if (!me.runs) nd.fail("rank %u: %s", r, mgc_runs_error(nullptr));
  }
}

void phase_count(Node &nd, uint32_t r, hipStream_t st_copy) {
  Rank &me = nd.ranks[r];
  const uint32_t nbk = 1u << nd.bits, n = nd.n;
  const uint32_t f0 = nd.cuts[r], f1 = nd.cuts[r + 1];
  const size_t kb = (size_t)nd.kw * 8;
  const double t0 = now_s();
  std::vector<uint64_t> tot(f1 - f0 + 1, 0), file_off(f1 - f0 + 1, 0);
  for (uint32_t f = f0; f < f1; f++) {
    for (uint32_t s = 0; s < n; s++) tot[f - f0] += nd.ranks[s].counts[f];
    file_off[f - f0 + 1] = file_off[f - f0] + tot[f - f0];
  }
  DevMem mem;
  void *inbox = nullptr, *res_keys = nullptr, *res_counts = nullptr;
  ND_HIP(mem.alloc(&inbox, file_off[f1 - f0] * kb));
  // one batch: the counted waves stay here until the writer is closed (a wave has at most as many distinct k-mers as k-mers,
  // so its result fits at its own offset; one allocation for all waves -- hipMalloc beside running kernels is slow).
  // Several batches: a wave's result is parked in the run store at once, so one wave's room is enough.
  const bool direct = nd.n_batches == 1;
  uint32_t most = 0;
  for (uint32_t q = 0; q < n; q++) most = std::max(most, nd.cuts[q + 1] - nd.cuts[q]);
  const uint32_t bpw = std::max(1u, (most + 15) / 16), n_waves = (most + bpw - 1) / bpw;
  uint64_t res_entries = file_off[f1 - f0];
  if (!direct) {
    res_entries = 0;
    for (uint32_t i = 0; i < n_waves; i++) {
      const uint32_t lo = std::min(f1, f0 + i * bpw), hi = std::min(f1, f0 + (i + 1) * bpw);
      res_entries = std::max(res_entries, file_off[hi - f0] - file_off[lo - f0]);
    }
  }
  ND_HIP(mem.alloc(&res_keys, res_entries * kb));
  ND_HIP(mem.alloc(&res_counts, res_entries * sizeof(uint32_t)));
  mgc_session *sess = me.sess;
  mgc_db_stream *ds = me.ds;
  const mgc_count_config &cfg = nd.cfg;
  const uint64_t blocks_per_bucket = 1ull << (cfg.w_prefix - nd.bits);

  std::vector<hipEvent_t> ev(n_waves, nullptr);
  bool ok = true;
  auto hip_ok = [&](hipError_t e, const char *what) { if (e != hipSuccess && ok) { nd.fail("rank %u: %s -> %s", r, what, hipGetErrorString(e)); ok = false; } return e == hipSuccess; };

  auto pull = [&](uint32_t i) {                 // wave i of this rank's range: every source's piece of every bucket
    const uint32_t lo = std::min(f1, f0 + i * bpw), hi = std::min(f1, f0 + (i + 1) * bpw);
    for (uint32_t f = lo; f < hi && ok; f++) {
      uint64_t at = file_off[f - f0];
      for (uint32_t s = 0; s < n && ok; s++) {
        const Rank &src = nd.ranks[s];
        const uint64_t c = src.counts[f];
        if (!c) continue;
        char *dst = (char *)inbox + at * kb;
        const char *from = (const char *)src.d_keys + src.off[f] * kb;
        if (src.device == me.device) hip_ok(hipMemcpyAsync(dst, from, c * kb, hipMemcpyDeviceToDevice, st_copy), "hipMemcpyAsync(D2D)");
        else                         hip_ok(hipMemcpyPeerAsync(dst, me.device, from, src.device, c * kb, st_copy), "hipMemcpyPeerAsync");
        at += c;
      }
    }
    if (ok && hip_ok(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming), "hipEventCreate")) hip_ok(hipEventRecord(ev[i], st_copy), "hipEventRecord");
  };
  auto count_wave = [&](uint32_t i) {
    const uint32_t lo = std::min(f1, f0 + i * bpw), hi = std::min(f1, f0 + (i + 1) * bpw);
    if (lo >= hi) return;
    if (!hip_ok(hipEventSynchronize(ev[i]), "hipEventSynchronize")) return;
    std::vector<uint64_t> bc(nbk, 0);
    uint64_t nk = 0;
    for (uint32_t f = lo; f < hi; f++) { bc[f] = tot[f - f0]; nk += bc[f]; }
    void *ok_keys = nullptr, *ok_counts = nullptr;
    uint64_t ndist = 0;
    if (nk) {
      int rc = mgc_count_buckets(sess, (char *)inbox + file_off[lo - f0] * kb, nd.bits, bc.data());
      mgc_result_info info{};
      if (rc == MGC_OK) rc = mgc_get_result_info(sess, &info);
      if (rc != MGC_OK) { nd.fail("rank %u: mgc_count_buckets: %d %s", r, rc, mgc_last_error(sess)); ok = false; return; }
      ndist = info.n_distinct;
      const uint64_t at = direct ? file_off[lo - f0] : 0;
      ok_keys = (char *)res_keys + at * kb;
      ok_counts = (char *)res_counts + at * sizeof(uint32_t);
      rc = mgc_copy_result_device(sess, ok_keys, (uint32_t *)ok_counts);
      if (rc != MGC_OK) { nd.fail("rank %u: mgc_copy_result_device: %d %s", r, rc, mgc_last_error(sess)); ok = false; return; }
      if (direct) me.n_distinct += ndist;
    }
    if (direct) {
      const int rc = mgc_db_stream_write(ds, ok_keys, (const uint32_t *)ok_counts, ndist, lo * blocks_per_bucket, hi * blocks_per_bucket);
      if (rc != MGC_OK) { nd.fail("rank %u: mgc_db_stream_write: %s", r, mgc_db_stream_error(ds)); ok = false; }
    } else if (ndist) {
      const int rc = mgc_runs_add(me.runs, ok_keys, (const uint32_t *)ok_counts, ndist, nullptr);
      if (rc != MGC_OK) { nd.fail("rank %u: mgc_runs_add: %s", r, mgc_runs_error(me.runs)); ok = false; }
    }
  };

  for (uint32_t i = 0; i <= n_waves && ok && !nd.failed.load(); i++) {
    if (i < n_waves) pull(i);                   // wave i goes onto the links ...
    if (i >= 1 && ok) count_wave(i - 1);        // ... while wave i-1 is grouped, counted and handed to the writer / parked
  }
  (void)hipStreamSynchronize(st_copy);
  if (direct) {                                 // the wave buffers must outlive the writer's reads
    const double t1 = now_s();
    const int crc = mgc_db_stream_close(ds, &me.wp);        // waits for this rank's files
    me.ds = nullptr;
    if (crc != MGC_OK && ok) nd.fail("rank %u: mgc_db_stream_close: %s", r, mgc_db_stream_error(nullptr));
    me.t_close = now_s() - t1;
  }
  me.t_count += now_s() - t0 - me.t_close;
  for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
}

// several batches: every wave of every batch sits in the run store -- merge this rank's prefix range once, into its part
void owner_finish(Node &nd, uint32_t r) {
  Rank &me = nd.ranks[r];
  if (nd.n_batches > 1 && me.runs && me.ds && !nd.failed.load()) {
    const double t0 = now_s();
    const uint64_t bpb = 1ull << (nd.cfg.w_prefix - nd.bits);
    const int rc = mgc_runs_write(me.runs, me.ds, (uint64_t)nd.cuts[r] * bpb, (uint64_t)nd.cuts[r + 1] * bpb);
    if (rc != MGC_OK) nd.fail("rank %u: mgc_runs_write: %s", r, mgc_runs_error(me.runs));
    (void)mgc_runs_get_profile(me.runs, &me.rp);
    me.n_distinct = me.rp.n_merged;
    me.t_merge = now_s() - t0;
    const double t1 = now_s();
    const int crc = mgc_db_stream_close(me.ds, &me.wp);
    me.ds = nullptr;
    if (crc != MGC_OK) nd.fail("rank %u: mgc_db_stream_close: %s", r, mgc_db_stream_error(nullptr));
    me.t_close = now_s() - t1;
  }
  if (me.ds) { (void)mgc_db_stream_close(me.ds, nullptr); me.ds = nullptr; }     // a failed run: release the writer
  if (me.runs) { mgc_runs_close(me.runs); me.runs = nullptr; }
  if (me.sess) { mgc_close(me.sess); me.sess = nullptr; }
}

void rank_main(Node &nd, uint32_t r) {
  Rank &me = nd.ranks[r];
  hipStream_t st = nullptr, st_copy = nullptr;
  bool up = hipSetDevice(me.device) == hipSuccess;
  if (up) {
    for (uint32_t s = 0; s < nd.n; s++) {       // direct xGMI copies where the platform allows them (a refusal only means
      const int peer = nd.ranks[s].device;      // hipMemcpyPeerAsync stages through the host)
      int can = 0;
      if (peer != me.device && hipDeviceCanAccessPeer(&can, me.device, peer) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(peer, 0);
    }
    (void)hipGetLastError();
    up = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&st_copy, hipStreamNonBlocking) == hipSuccess;
  }
  if (!up) nd.fail("rank %u: device %d / stream setup failed", r, me.device);

  if (!nd.failed.load()) phase_prepare(nd, r, st);
  nd.bar.wait();                                // every rank's histogram of ALL its reads is published
  if (!nd.failed.load() && r == 0) {
    const uint32_t nbk = 1u << nd.bits;
    std::vector<uint64_t> total(nbk, 0);
    uint64_t max_nb = 0;
    for (const Rank &q : nd.ranks) { for (uint32_t b = 0; b < nbk; b++) total[b] += q.total[b]; max_nb = std::max(max_nb, q.n_bases); }
    nd.cuts = balanced_ranges(total, nd.n);
    // Batches: a rank holds, per base of a batch, its k-mers (8/16 B), as owner an inbox of about as many, a result copy and the
    // count arena (4 B of counts + the ping-pong file) -- 2 + 28 (52) B with slack; 60 % of the free HBM may go there.
    uint64_t batch = nd.batch_bases;
    if (batch == 0) {
      size_t free_b = 0, total_b = 0;
      batch = ~0ull;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b) batch = (uint64_t)((double)free_b * 0.6 / (double)(2 + 26ull * nd.kw));
    }
    if (batch < 1024) batch = 1024;
    nd.n_batches = (uint32_t)std::max<uint64_t>(1, (max_nb + batch - 1) / batch);
  }
  nd.bar.wait();
  if (!nd.failed.load()) owner_open(nd, r);
  for (uint32_t b = 0; b < nd.n_batches; b++) {
    if (!nd.failed.load()) phase_partition(nd, r, b, st);
    nd.bar.wait();                              // every partition of this batch is complete and its histogram published
    if (!nd.failed.load()) phase_count(nd, r, st_copy);
    nd.bar.wait();                              // nobody reads this rank's k-mers any more
    if (me.d_keys) { (void)hipFree(me.d_keys); me.d_keys = nullptr; }
  }
  owner_finish(nd, r);
  if (me.d_own) { (void)hipFree(me.d_own); me.d_own = nullptr; }
  if (st) (void)hipStreamDestroy(st);
  if (st_copy) (void)hipStreamDestroy(st_copy);
}

}  // namespace

extern "C" int mgc_count_node(const mgc_count_config *cfg, uint32_t n_ranks, const int *devices,
                              const uint8_t *const *d_bases, const uint64_t *n_bases,
                              const char *db_path, int host_threads, mgc_node_profile *prof) {
  return mgc_count_node_batched(cfg, n_ranks, devices, d_bases, n_bases, 0, db_path, host_threads, prof);
}

extern "C" int mgc_count_node_batched(const mgc_count_config *cfg, uint32_t n_ranks, const int *devices,
                                      const uint8_t *const *d_bases, const uint64_t *n_bases, uint64_t batch_bases,
                                      const char *db_path, int host_threads, mgc_node_profile *prof) {
  if (!cfg || !n_ranks || !d_bases || !n_bases || !db_path) { set_err(nullptr, "mgc_count_node: bad arguments"); return MGC_EINVAL; }
  if (cfg->count_suffix_length) { set_err(nullptr, "mgc_count_node: count-suffix is a single-device option"); return MGC_EINVAL; }
  mgc_count_config eff = *cfg;
  if (!mgc::effective_geometry(&eff)) return MGC_EINVAL;                 // simple mode: countSimple's block geometry, like mgc_open
  if (eff.k < 1 || eff.k > 64 || eff.w_prefix < MGC_NUM_FILES_BITS || eff.w_prefix + eff.w_data != 2 * eff.k) {
    set_err(nullptr, "mgc_count_node: the configuration has not been through mgc_configure_counting");
    return MGC_EINVAL;
  }
  cfg = &eff;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_err(nullptr, "mgc_count_node: no HIP device"); return MGC_EHIP; }
  int prev = 0; (void)hipGetDevice(&prev);

  Node nd;
  nd.cfg = *cfg; nd.n = n_ranks; nd.kw = cfg->k > 32 ? 2 : 1; nd.path = db_path; nd.host_threads = std::max(1, host_threads);
  nd.batch_bases = batch_bases;
  nd.ranks.resize(n_ranks);
  uint64_t max_nb = 0, total_bases = 0;
  for (uint32_t r = 0; r < n_ranks; r++) {
    const int dev = devices ? devices[r] : (int)(r % (uint32_t)ndev);
    if (dev < 0 || dev >= ndev) { set_err(nullptr, "mgc_count_node: rank %u names device %d of %d", r, dev, ndev); return MGC_EINVAL; }
    if (n_bases[r] && !d_bases[r]) { set_err(nullptr, "mgc_count_node: rank %u has no bases", r); return MGC_EINVAL; }
    nd.ranks[r].device = dev; nd.ranks[r].d_bases = d_bases[r]; nd.ranks[r].n_bases = n_bases[r];
    max_nb = std::max(max_nb, n_bases[r]); total_bases += n_bases[r];
  }
  nd.bits = bucket_bits_for(n_ranks, cfg->k, max_nb, cfg->w_prefix);
  if ((1u << nd.bits) < n_ranks) {
    set_err(nullptr, "mgc_count_node: %u ranks but only %u ranges of the k-mer space can be routed (k=%u, prefix of %u bits)", n_ranks, 1u << nd.bits, cfg->k, cfg->w_prefix);
    return MGC_EINVAL;
  }
  nd.bar.n = n_ranks;

  const double t0 = now_s();
  std::vector<std::thread> th;
  for (uint32_t r = 0; r < n_ranks; r++) th.emplace_back(rank_main, std::ref(nd), r);
  for (auto &t : th) t.join();
  const double t1 = now_s();
  (void)hipSetDevice(prev);
  if (nd.failed.load()) { set_err(nullptr, "mgc_count_node: %s", nd.err.c_str()); return MGC_EHIP; }
  if (n_ranks > 1 && mdb_merge_parts(db_path, n_ranks) != 0) { set_err(nullptr, "mgc_count_node: %s", mdb_last_error()); return MGC_EHIP; }
  const double t2 = now_s();
  if (prof) {
    *prof = mgc_node_profile{};
    prof->n_ranks = n_ranks; prof->bucket_bits = nd.bits; prof->n_bases = total_bases;
    for (const Rank &q : nd.ranks) {
      prof->n_distinct += q.n_distinct; prof->n_instances += q.n_instances; prof->data_bytes += q.wp.data_bytes;
      prof->partition_s = std::max(prof->partition_s, q.t_partition);
      prof->exchange_count_s = std::max(prof->exchange_count_s, q.t_count);
      prof->close_s = std::max(prof->close_s, q.t_close);
    }
    prof->merge_parts_s = t2 - t1; prof->total_s = t2 - t0;
    prof->n_batches = nd.n_batches;
    for (const Rank &q : nd.ranks) {
      prof->n_host_runs += q.rp.n_host_runs; prof->host_run_bytes += q.rp.host_bytes;
      prof->merge_runs_s = std::max(prof->merge_runs_s, q.t_merge);
      prof->peak_hbm_bytes = std::max<uint64_t>(prof->peak_hbm_bytes, q.rp.peak_hbm_bytes);
    }
  }
  return MGC_OK;
}

// The routing plan mgc_count_node derives, on its own (pure host arithmetic: which bucket granularity, which contiguous bucket
// range every rank owns) -- for callers that want to know where a k-mer will be counted, and for the CPU tests that hold it
// against the RCCL launcher's plan (meryl_amd/count.py: shard_bucket_bits, balanced_file_ranges).
extern "C" int mgc_node_plan(uint32_t n_ranks, uint32_t k, uint64_t max_rank_bases, uint32_t w_prefix, uint32_t *bucket_bits,
                             const uint64_t *bucket_totals, uint32_t *cuts) {
  if (!n_ranks || !bucket_bits || k < 1 || k > 64) return MGC_EINVAL;
  *bucket_bits = bucket_bits_for(n_ranks, k, max_rank_bases, w_prefix);
  if ((1u << *bucket_bits) < n_ranks) return MGC_EINVAL;
  if (bucket_totals && cuts) {
    const std::vector<uint64_t> total(bucket_totals, bucket_totals + (1u << *bucket_bits));
    const std::vector<uint32_t> c = balanced_ranges(total, n_ranks);
    for (uint32_t r = 0; r <= n_ranks; r++) cuts[r] = c[r];
  }
  return MGC_OK;
}

// The CLI's gpus=N: everything was read and parsed through ONE session (device parser, pinned ring); its staged base
// stream is cut into n_ranks slices that overlap by k-1 bases -- a window starting in the last k-1 bases of slice r-1
// is incomplete there and complete in slice r, so no k-mer is lost or counted twice wherever the cut falls --
// `compress` is applied before cutting (it must see whole sequences), slices move to their devices, mgc_count_node runs.
extern "C" int mgc_count_node_staged(mgc_session *s, uint32_t n_ranks, const int *devices, const char *db_path,
                                     int host_threads, mgc_node_profile *prof) {
  if (!s || !n_ranks || !db_path) return MGC_EINVAL;
  const uint8_t *d = nullptr; uint64_t n = 0;
  int rc = mgc_staged_bases(s, &d, &n);
  if (rc != MGC_OK) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_err(&s->err, "mgc_count_node_staged: no HIP device"); return MGC_EHIP; }
  mgc_count_config cfg = s->cfg;
  const uint32_t k = cfg.k;
  std::vector<std::pair<int, void *>> owned;                       // (device, pointer) of what this call allocated
  auto release = [&] { for (auto &o : owned) { (void)hipSetDevice(o.first); (void)hipFree(o.second); } (void)hipSetDevice(s->device); };
  auto hip_fail = [&](hipError_t e, const char *what) { set_err(&s->err, "mgc_count_node_staged: %s -> %s", what, hipGetErrorString(e)); release(); return MGC_EHIP; };
  hipError_t e = hipSetDevice(s->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  if (cfg.homopoly_compress && n) {
    void *out = nullptr, *ws = nullptr;
    const size_t wsb = mgc_dev_homopoly_workspace_bytes(n);
    if ((e = hipMalloc(&out, n)) != hipSuccess) return hip_fail(e, "hipMalloc(compressed stream)");
    owned.emplace_back(s->device, out);
    if ((e = hipMalloc(&ws, std::max<size_t>(wsb, 256))) != hipSuccess) return hip_fail(e, "hipMalloc(compress workspace)");
    owned.emplace_back(s->device, ws);
    uint64_t n_out = 0;
    rc = mgc_dev_homopoly_compress(d, n, (uint8_t *)out, &n_out, ws, wsb, nullptr);
    if (rc == MGC_OK && (e = hipDeviceSynchronize()) != hipSuccess) return hip_fail(e, "homopolymer compression");
    if (rc != MGC_OK) { set_err(&s->err, "mgc_count_node_staged: homopolymer compression failed (%d)", rc); release(); return rc; }
    d = (const uint8_t *)out; n = n_out;
  }
  cfg.homopoly_compress = 0;
  std::vector<const uint8_t *> ptr(n_ranks, nullptr);
  std::vector<uint64_t> len(n_ranks, 0);
  std::vector<int> dev(n_ranks, 0);
  for (uint32_t r = 0; r < n_ranks; r++) {
    const uint64_t cut = (uint64_t)((unsigned __int128)n * r / n_ranks), end = (uint64_t)((unsigned __int128)n * (r + 1) / n_ranks);
    const uint64_t a = (r && cut >= k - 1) ? cut - (k - 1) : 0;     // cut < k-1: the earlier slices are shorter than a k-mer, hold no window
    dev[r] = devices ? devices[r] : (int)(r % (uint32_t)ndev);
    if (dev[r] < 0 || dev[r] >= ndev) { set_err(&s->err, "mgc_count_node_staged: rank %u names device %d of %d", r, dev[r], ndev); release(); return MGC_EINVAL; }
    len[r] = end - a;
    if (!len[r]) continue;
    if (dev[r] == s->device) { ptr[r] = d + a; continue; }
    void *p = nullptr;
    if ((e = hipSetDevice(dev[r])) != hipSuccess) return hip_fail(e, "hipSetDevice(rank device)");
    if ((e = hipMalloc(&p, len[r])) != hipSuccess) return hip_fail(e, "hipMalloc(rank slice)");
    owned.emplace_back(dev[r], p);
    if ((e = hipMemcpyPeer(p, dev[r], d + a, s->device, len[r])) != hipSuccess) return hip_fail(e, "hipMemcpyPeer(rank slice)");
    ptr[r] = (const uint8_t *)p;
  }
  (void)hipSetDevice(s->device);
  rc = mgc_count_node(&cfg, n_ranks, dev.data(), ptr.data(), len.data(), db_path, host_threads, prof);
  if (rc != MGC_OK) set_err(&s->err, "%s", mgc_last_error(nullptr));
  release();
  return rc;
}
