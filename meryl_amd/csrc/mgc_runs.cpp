// mgc_runs.cpp -- sorted runs of partial count results: parked in HBM or in pinned host DRAM, merged once at the end
// (include/meryl_db.h, mgc_runs_*).
//
// Reference side: the spill of countThreads.  writeBatch dumps every bucket of a full memory as an ITERATION of the
// output files (src/meryl/merylOp-countThreads.C:285-380; finishBatch :362) and merylBlockWriter::finish() merges the
// iterations of each file when the count ends (:461-464) -- the result may be larger than memory.  Round 2's form of it
// (merge every batch into a running result that stays in HBM) bounded the result by HBM and rewrote the running result
// once per batch; this is the replacement: a batch result is parked as a run and nothing is merged until the end.
#include "mgc_runs.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>

using mgc::set_err;

namespace {
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
uint64_t env_u64(const char *name, uint64_t dflt) {
  const char *e = getenv(name);
  return (e && *e) ? strtoull(e, nullptr, 10) : dflt;
}
}  // namespace

#define RN_TRY(expr)                                                                             \
  do {                                                                                           \
    hipError_t e__ = (expr);                                                                     \
    if (e__ != hipSuccess) {                                                                     \
      set_err(&err, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__));       \
      return (e__ == hipErrorOutOfMemory) ? MGC_ENOMEM : MGC_EHIP;                               \
    }                                                                                            \
  } while (0)

mgc_runs::mgc_runs(uint32_t k_, uint32_t w_prefix_, int device_, uint64_t budget, uint64_t chunk)
    : k(k_), kw(k_ > 32 ? 2u : 1u), w_prefix(w_prefix_), device(device_), dev_budget(budget), chunk_bytes(chunk) {
  slice_bits = std::min<uint32_t>(w_prefix, 14u);
  if (slice_bits > 2 * k) slice_bits = 2 * k;
  n_slices = 1ull << slice_bits;
  esz = sizeof(uint64_t) * kw + sizeof(uint32_t);
  dev_budget = env_u64("MGC_OOC_BUDGET", dev_budget);
  chunk_bytes = env_u64("MGC_OOC_CHUNK", chunk_bytes);
  memset(&prof, 0, sizeof(prof));
}

void mgc_runs::start_prealloc(size_t kb, size_t cb) {
  pre_thread = std::thread([this, kb, cb] {
    (void)hipSetDevice(device);
    pre_k = pre_c = nullptr; pre_kb = pre_cb = 0;
    if (hipHostMalloc(&pre_k, kb, hipHostMallocDefault) != hipSuccess) { pre_k = nullptr; (void)hipGetLastError(); return; }
    if (hipHostMalloc(&pre_c, cb, hipHostMallocDefault) != hipSuccess) { (void)hipHostFree(pre_k); pre_k = pre_c = nullptr; (void)hipGetLastError(); return; }
    pre_kb = kb; pre_cb = cb;
  });
}

void mgc_runs::drop_prealloc() {
  if (pre_thread.joinable()) pre_thread.join();
  if (pre_k) (void)hipHostFree(pre_k);
  if (pre_c) (void)hipHostFree(pre_c);
  pre_k = pre_c = nullptr; pre_kb = pre_cb = 0;
}

mgc_runs::~mgc_runs() {
  (void)hipSetDevice(device);
  drop_prealloc();
  for (Run &r : runs) free_run(r);
  for (auto &b : buf) b.release();
  d_slices.release();
  if (st_copy) (void)hipStreamDestroy(st_copy);
  if (st_up) (void)hipStreamDestroy(st_up);
  if (st_mg) (void)hipStreamDestroy(st_mg);
  for (hipEvent_t e : ev_up) if (e) (void)hipEventDestroy(e);
  if (ev_src) (void)hipEventDestroy(ev_src);
}

void mgc_runs::free_run(Run &r) {
  if (r.on_host) { if (r.keys) (void)hipHostFree(r.keys); if (r.counts) (void)hipHostFree(r.counts); }
  else           { if (r.keys) (void)hipFree(r.keys);     if (r.counts) (void)hipFree(r.counts); }
  r.keys = nullptr; r.counts = nullptr;
}

int mgc_runs::setup() {
  if (st_copy) return MGC_OK;
  RN_TRY(hipSetDevice(device));
  RN_TRY(hipStreamCreateWithFlags(&st_copy, hipStreamNonBlocking));
  RN_TRY(hipStreamCreateWithFlags(&st_up, hipStreamNonBlocking));
  RN_TRY(hipStreamCreateWithFlags(&st_mg, hipStreamNonBlocking));
  RN_TRY(hipEventCreateWithFlags(&ev_up[0], hipEventDisableTiming));
  RN_TRY(hipEventCreateWithFlags(&ev_up[1], hipEventDisableTiming));
  RN_TRY(hipEventCreateWithFlags(&ev_src, hipEventDisableTiming));
  return MGC_OK;
}

void mgc_runs::sample_hbm() {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b >= free_b)
    prof.peak_hbm_bytes = std::max<uint64_t>(prof.peak_hbm_bytes, total_b - free_b);
}

// One run more: a copy of n ascending distinct k-mers + counts produced on `st`.  Device-to-device while the device
// budget lasts; otherwise device-to-host into freshly pinned memory on the store's copy stream.  Returns when the source
// may be reused (the copy itself is done: the caller's next batch overwrites the source at its very end, and a batch
// takes longer to count than its result takes to copy, so nothing is gained by returning earlier).
int mgc_runs::add(const void *d_keys, const uint32_t *d_counts, uint64_t n, hipStream_t st) {
  if (n == 0) return MGC_OK;
  int rc = setup();
  if (rc != MGC_OK) return rc;
  RN_TRY(hipSetDevice(device));
  (void)hipGetLastError();                                 // (a stale error of this thread is not this call's)
  Run r;
  r.n = n;
  // where every slice of the k-mer space begins in this run
  RN_TRY(d_slices.ensure(sizeof(uint64_t) * (n_slices + 1)));
  RN_TRY(mgc::launch_block_offsets_range(d_keys, n, kw, 2 * k - slice_bits, 0, n_slices, n_slices, d_slices.as<uint64_t>(), st));
  r.slice.resize(n_slices + 1);
  RN_TRY(hipMemcpyAsync(r.slice.data(), d_slices.p, sizeof(uint64_t) * (n_slices + 1), hipMemcpyDeviceToHost, st));
  const size_t kb = sizeof(uint64_t) * kw * n, cb = sizeof(uint32_t) * n;
  const bool to_device = dev_budget == ~0ull || prof.device_bytes + kb + cb <= dev_budget;
#define RN_TRY_RUN(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { free_run(r);                                          \
    set_err(&err, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__));                                           \
    return (e__ == hipErrorOutOfMemory) ? MGC_ENOMEM : MGC_EHIP; } } while (0)
  if (to_device) {
    hipError_t e = hipMalloc(&r.keys, kb);
    if (e == hipSuccess) { void *c = nullptr; e = hipMalloc(&c, cb); r.counts = reinterpret_cast<uint32_t *>(c); }
    if (e != hipSuccess) {                                 // HBM is fuller than the budget assumed: this run goes to the host
      (void)hipGetLastError();
      free_run(r);
      r.keys = nullptr; r.counts = nullptr;
    } else {
      RN_TRY_RUN(hipMemcpyAsync(r.keys, d_keys, kb, hipMemcpyDeviceToDevice, st));
      RN_TRY_RUN(hipMemcpyAsync(r.counts, d_counts, cb, hipMemcpyDeviceToDevice, st));
      RN_TRY_RUN(hipStreamSynchronize(st));
      prof.device_bytes += kb + cb;
    }
  }
  if (!r.keys) {
    const double t0 = now_s();
    r.on_host = true;
    hipError_t e = hipSuccess;
    if (pre_thread.joinable()) pre_thread.join();
    if (pre_k && pre_kb >= kb && pre_cb >= cb) {            // the buffers the helper pinned while this batch was counted
      r.keys = pre_k; r.counts = reinterpret_cast<uint32_t *>(pre_c);
      pre_k = pre_c = nullptr; pre_kb = pre_cb = 0;
    } else {
      drop_prealloc();
      e = hipHostMalloc(&r.keys, kb, hipHostMallocDefault);
      if (e == hipSuccess) { void *c = nullptr; e = hipHostMalloc(&c, cb, hipHostMallocDefault); r.counts = reinterpret_cast<uint32_t *>(c); }
    }
    if (e != hipSuccess) {
      free_run(r);
      set_err(&err, "mgc_runs: %.1f GB of pinned host memory for a spilled run: %s", (kb + cb) / 1e9, hipGetErrorString(e));
      return MGC_ENOMEM;
    }
    RN_TRY_RUN(hipEventRecord(ev_src, st));
    RN_TRY_RUN(hipStreamWaitEvent(st_copy, ev_src, 0));
    RN_TRY_RUN(hipMemcpyAsync(r.keys, d_keys, kb, hipMemcpyDeviceToHost, st_copy));
    RN_TRY_RUN(hipMemcpyAsync(r.counts, d_counts, cb, hipMemcpyDeviceToHost, st_copy));
    RN_TRY_RUN(hipStreamSynchronize(st_copy));
    RN_TRY_RUN(hipStreamSynchronize(st));
    prof.host_bytes += kb + cb;
    prof.n_host_runs++;
    prof.spill_s += now_s() - t0;
    start_prealloc(kb + kb / 16, cb + cb / 16);             // batches are of one size: the next run will be about this large
  }
#undef RN_TRY_RUN
  if (r.slice[0] != 0 || r.slice[n_slices] != n) { free_run(r); set_err(&err, "mgc_runs_add: keys not ascending / beyond 2k bits"); return MGC_EINVAL; }
  prof.n_runs++;
  prof.n_entries += n;
  runs.push_back(std::move(r));
  sample_hbm();
  return MGC_OK;
}

// pairwise merge of device-resident pieces into `dst` (keys) / `dstc` at element offset `at`; *n_out elements
int mgc_runs::merge_pair(const Piece &a, const Piece &b, void *dst, uint32_t *dstc, uint64_t *n_out) {
  RN_TRY(buf[B_WS].ensure(mgc::merge_workspace_bytes(a.n, b.n)));
  RN_TRY(mgc::launch_merge_count(a.k, a.n, b.k, b.n, kw, 0, buf[B_WS].p, st_mg));
  RN_TRY(mgc::merge_read_total(buf[B_WS].p, n_out, st_mg));
  RN_TRY(mgc::launch_merge_emit(a.k, a.c, a.n, b.k, b.c, b.n, kw, 0, buf[B_WS].p, dst, dstc, st_mg));
  return MGC_OK;
}

// All runs, slices [s0, s1) -> ascending merged chunks -> sink.
int mgc_runs::deliver(uint64_t s0, uint64_t s1, mgc::RunSink &sink) {
  int rc = setup();
  if (rc != MGC_OK) return rc;
  RN_TRY(hipSetDevice(device));
  if (s0 > s1 || s1 > n_slices) { set_err(&err, "mgc_runs: bad slice range"); return MGC_EINVAL; }
  drop_prealloc();                                          // no more runs are coming
  const double t_begin = now_s();
  const size_t kbytes = sizeof(uint64_t) * kw;
  // chunk capacity in entries: two input sets + two ping-pong pairs = six buffers of C entries each
  uint64_t budget = chunk_bytes;
  if (budget == 0) {
    size_t free_b = 0, total_b = 0;
    budget = 8ull << 30;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b) budget = std::min<uint64_t>(free_b / 4, 24ull << 30);
  }
  uint64_t C = std::max<uint64_t>(budget / (6 * esz), 4096);
  // a single slice may hold more than that (over all runs): the buffers then grow to the largest slice
  for (uint64_t s = s0; s < s1; s++) {
    uint64_t tot = 0;
    for (const Run &r : runs) tot += r.slice[s + 1] - r.slice[s];
    C = std::max(C, tot);
  }
  // input sets In[2] (uploads of the host pieces), ping-pong P[2], Q[2]; keys and counts apart
  auto bk = [&](int set, int which) -> DBuf & { return buf[B_IN0K + 2 * (3 * set + which)]; };      // which: 0 In, 1 P, 2 Q
  auto bc = [&](int set, int which) -> DBuf & { return buf[B_IN0K + 2 * (3 * set + which) + 1]; };

  struct Chunk { uint64_t a, b, total; };
  std::vector<Chunk> chunks;
  for (uint64_t s = s0; s < s1;) {
    Chunk c{s, s, 0};
    while (c.b < s1) {
      uint64_t tot = 0;
      for (const Run &r : runs) tot += r.slice[c.b + 1] - r.slice[c.b];
      if (c.b > c.a && c.total + tot > C) break;
      c.total += tot; c.b++;
    }
    chunks.push_back(c);
    s = c.b;
  }
  if (chunks.empty()) return MGC_OK;

  // gather + upload of a chunk's pieces into input set `set`
  std::vector<Piece> pieces[2];
  bool any_host = false;
  for (const Run &r : runs) any_host = any_host || r.on_host;
  auto prefetch = [&](size_t ci) -> int {
    const int set = (int)(ci & 1);
    const Chunk &c = chunks[ci];
    pieces[set].clear();
    uint64_t host_total = 0;
    for (const Run &r : runs) if (r.on_host) host_total += r.slice[c.b] - r.slice[c.a];
    if (host_total) { RN_TRY(bk(set, 0).ensure(kbytes * host_total)); RN_TRY(bc(set, 0).ensure(sizeof(uint32_t) * host_total)); }
    uint64_t at = 0;
    const double t0 = now_s();
    for (const Run &r : runs) {
      const uint64_t lo = r.slice[c.a], n = r.slice[c.b] - lo;
      if (!n) continue;
      Piece p; p.n = n;
      if (r.on_host) {
        unsigned char *dk = bk(set, 0).as<unsigned char>() + kbytes * at;
        uint32_t *dc = bc(set, 0).as<uint32_t>() + at;
        RN_TRY(hipMemcpyAsync(dk, reinterpret_cast<const unsigned char *>(r.keys) + kbytes * lo, kbytes * n, hipMemcpyHostToDevice, st_up));
        RN_TRY(hipMemcpyAsync(dc, r.counts + lo, sizeof(uint32_t) * n, hipMemcpyHostToDevice, st_up));
        p.k = dk; p.c = dc;
        at += n;
      } else {
        p.k = reinterpret_cast<const unsigned char *>(r.keys) + kbytes * lo;
        p.c = r.counts + lo;
      }
      pieces[set].push_back(p);
    }
    RN_TRY(hipEventRecord(ev_up[set], st_up));
    prof.upload_s += now_s() - t0;                          // issue time only; the waits are accounted below
    return MGC_OK;
  };

  uint64_t jobs[2] = {0, 0};
  bool job_open[2] = {false, false};
  rc = prefetch(0);
  if (rc != MGC_OK) return rc;
  for (size_t ci = 0; ci < chunks.size(); ci++) {
    const int set = (int)(ci & 1);
    const Chunk &c = chunks[ci];
    { const double t0 = now_s(); RN_TRY(hipEventSynchronize(ev_up[set])); prof.upload_s += now_s() - t0; }
    // ---- merge tree: every level writes all of its pieces into the other ping-pong buffer ----
    std::vector<Piece> cur = pieces[set];
    const double tm0 = now_s();
    int level = 0;
    while (cur.size() > 1) {
      DBuf &ok = bk(set, 1 + (level & 1)), &oc = bc(set, 1 + (level & 1));
      uint64_t bound = 0;
      for (const Piece &p : cur) bound += p.n;
      RN_TRY(ok.ensure(kbytes * bound));
      RN_TRY(oc.ensure(sizeof(uint32_t) * bound));
      std::vector<Piece> next;
      uint64_t at = 0;
      for (size_t i = 0; i < cur.size(); i += 2) {
        Piece o;
        o.k = ok.as<unsigned char>() + kbytes * at;
        uint32_t *ocp = oc.as<uint32_t>() + at;
        o.c = ocp;
        if (i + 1 < cur.size()) {
          uint64_t n_out = 0;
          rc = merge_pair(cur[i], cur[i + 1], const_cast<void *>(o.k), ocp, &n_out);
          if (rc != MGC_OK) return rc;
          o.n = n_out;
          at += cur[i].n + cur[i + 1].n;                    // upper bound: the next piece starts past it
        } else {                                            // the odd one out moves along (its old buffer is the next level's target)
          RN_TRY(hipMemcpyAsync(const_cast<void *>(o.k), cur[i].k, kbytes * cur[i].n, hipMemcpyDeviceToDevice, st_mg));
          RN_TRY(hipMemcpyAsync(ocp, cur[i].c, sizeof(uint32_t) * cur[i].n, hipMemcpyDeviceToDevice, st_mg));
          o.n = cur[i].n;
          at += cur[i].n;
        }
        next.push_back(o);
      }
      cur.swap(next);
      level++;
    }
    RN_TRY(hipStreamSynchronize(st_mg));
    prof.merge_ms += (now_s() - tm0) * 1e3;
    sample_hbm();
    const void *out_k = cur.empty() ? nullptr : cur[0].k;
    const uint32_t *out_c = cur.empty() ? nullptr : cur[0].c;
    const uint64_t out_n = cur.empty() ? 0 : cur[0].n;
    prof.n_merged += out_n;
    prof.n_chunks++;
    rc = sink.put(out_k, out_c, out_n, c.a, c.b, slice_bits, &jobs[set]);
    if (rc != MGC_OK) { set_err(&err, "%s", sink.error()); return rc; }
    job_open[set] = true;
    // the other set's buffers are free once its chunk has left them: then the next chunk's upload may overwrite them
    if (ci + 1 < chunks.size()) {
      if (job_open[set ^ 1]) { rc = sink.wait(jobs[set ^ 1]); if (rc != MGC_OK) { set_err(&err, "%s", sink.error()); return rc; } job_open[set ^ 1] = false; }
      rc = prefetch(ci + 1);
      if (rc != MGC_OK) return rc;
    }
  }
  for (int s = 0; s < 2; s++)
    if (job_open[s]) { rc = sink.wait(jobs[s]); if (rc != MGC_OK) { set_err(&err, "%s", sink.error()); return rc; } }
  prof.deliver_s += now_s() - t_begin;
  return MGC_OK;
}

// Every run is in HBM and small enough: one device-resident result (pairwise tree, inputs freed as they are consumed).
// On return the store holds at most one run; *keys / *counts point into it (owned by the store).
int mgc_runs::collapse(const void **keys, const uint32_t **counts, uint64_t *n) {
  int rc = setup();
  if (rc != MGC_OK) return rc;
  RN_TRY(hipSetDevice(device));
  for (const Run &r : runs) if (r.on_host) { set_err(&err, "mgc_runs: collapse with runs on the host"); return MGC_ESTATE; }
  const size_t kbytes = sizeof(uint64_t) * kw;
  const double t0 = now_s();
  uint32_t pair_merges = 0;
  const uint32_t fail_merge_at = (uint32_t)env_u64("MGC_RUNS_FAIL_MERGE", 0);   // tests (read once per collapse): the n-th pair merge "runs out of memory"
  while (runs.size() > 1) {
    std::vector<Run> next;
    // any failure below leaves the store CONSISTENT: the merged outputs so far, the odd run moved over, and the runs
    // not yet merged (the inputs already merged are freed and dropped) -- every k-mer is still in exactly one run, so
    // the caller can deliver out of core instead (finalize_from_runs) or retry
    size_t i = 0;
    // returns false when a merged output could not get a valid slice table: the store is then NOT deliverable (a run with an
    // all-zero table would look empty in every slice and its k-mers would be dropped silently) -- the caller must fail
    auto keep_survivors = [&]() -> bool {
      for (size_t j = i; j < runs.size(); j++) if (runs[j].keys) next.push_back(std::move(runs[j]));
      runs.swap(next);
      bool ok = true;
      // a merged output has no slice table yet (delivery needs one per run)
      for (Run &r : runs) {
        if (r.slice.size() == (size_t)n_slices + 1 || r.on_host || !r.keys) continue;
        r.slice.assign(n_slices + 1, 0);
        bool got = d_slices.ensure(sizeof(uint64_t) * (n_slices + 1)) == hipSuccess &&
                   mgc::launch_block_offsets_range(r.keys, r.n, kw, 2 * k - slice_bits, 0, n_slices, n_slices, d_slices.as<uint64_t>(), st_mg) == hipSuccess &&
                   hipMemcpyAsync(r.slice.data(), d_slices.p, sizeof(uint64_t) * (n_slices + 1), hipMemcpyDeviceToHost, st_mg) == hipSuccess &&
                   hipStreamSynchronize(st_mg) == hipSuccess;
        if (got && !(r.slice[0] == 0 && r.slice[n_slices] == r.n)) got = false;          // (the check add() makes)
        if (!got) { (void)hipGetLastError(); ok = false; }
      }
      return ok;
    };
#define RN_TRY_KEEP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { free_run(o); const bool ks__ = keep_survivors();  \
      set_err(&err, "%s:%d: %s -> %s%s", __FILE__, __LINE__, #expr, hipGetErrorString(e__), ks__ ? "" : " (and a merged run has no slice table)"); \
      return (e__ == hipErrorOutOfMemory && ks__) ? MGC_ENOMEM : MGC_EHIP; } } while (0)
    for (; i < runs.size(); i += 2) {
      if (i + 1 >= runs.size()) { next.push_back(std::move(runs[i])); continue; }
      Run &a = runs[i], &b = runs[i + 1];
      Run o;
      RN_TRY_KEEP(buf[B_WS].ensure(mgc::merge_workspace_bytes(a.n, b.n)));
      RN_TRY_KEEP(mgc::launch_merge_count(a.keys, a.n, b.keys, b.n, kw, 0, buf[B_WS].p, st_mg));
      uint64_t n_out = 0;
      RN_TRY_KEEP(mgc::merge_read_total(buf[B_WS].p, &n_out, st_mg));
      hipError_t e = hipMalloc(&o.keys, std::max<size_t>(kbytes * n_out, 256));
      if (e == hipSuccess) { void *c = nullptr; e = hipMalloc(&c, std::max<size_t>(sizeof(uint32_t) * n_out, 256)); o.counts = reinterpret_cast<uint32_t *>(c); }
      // tests: the n-th pair merge of this call "runs out of memory" (the fragmentation case hipMemGetInfo cannot foresee)
      if (fail_merge_at && e == hipSuccess && ++pair_merges == fail_merge_at) e = hipErrorOutOfMemory;
      if (e != hipSuccess) {
        (void)hipGetLastError();
        free_run(o);
        if (!keep_survivors()) {                             // not a consistent store: no out-of-core delivery from it
          set_err(&err, "mgc_runs: merging the runs in HBM: %s, and a merged run could not get its slice table", hipGetErrorString(e));
          return MGC_EHIP;
        }
        set_err(&err, "mgc_runs: merging the runs in HBM: %s", hipGetErrorString(e));
        return MGC_ENOMEM;
      }
      RN_TRY_KEEP(mgc::launch_merge_emit(a.keys, a.counts, a.n, b.keys, b.counts, b.n, kw, 0, buf[B_WS].p, o.keys, o.counts, st_mg));
      RN_TRY_KEEP(hipStreamSynchronize(st_mg));
      sample_hbm();
      o.n = n_out;
      free_run(a); free_run(b);
      next.push_back(std::move(o));
    }
#undef RN_TRY_KEEP
    runs.swap(next);
  }
  prof.merge_ms += (now_s() - t0) * 1e3;
  prof.device_bytes = runs.empty() ? 0 : (kbytes + sizeof(uint32_t)) * runs[0].n;
  *keys = runs.empty() ? nullptr : runs[0].keys;
  *counts = runs.empty() ? nullptr : runs[0].counts;
  *n = runs.empty() ? 0 : runs[0].n;
  prof.n_merged = *n;
  return MGC_OK;
}

// ---- sink: a database stream ----------------------------------------------------------------------------------------
namespace {
struct DbSink : mgc::RunSink {
  mgc_db_stream *d; uint32_t w_prefix; std::string msg;
  DbSink(mgc_db_stream *d_, uint32_t w) : d(d_), w_prefix(w) {}
  int put(const void *k, const uint32_t *c, uint64_t n, uint64_t sa, uint64_t sb, uint32_t slice_bits, uint64_t *job) override {
    const uint32_t sh = w_prefix - slice_bits;
    const int rc = mgc_db_stream_write(d, k, c, n, sa << sh, sb << sh);
    if (rc != MGC_OK) msg = mgc_db_stream_error(d);
    *job = mgc_db_stream_queued(d);
    return rc;
  }
  int wait(uint64_t job) override {
    const int rc = mgc_db_stream_wait_buffers(d, job);
    if (rc != MGC_OK) msg = mgc_db_stream_error(d);
    return rc;
  }
  const char *error() const override { return msg.c_str(); }
};
}  // namespace

int mgc_runs::write(mgc_db_stream *d, uint64_t prefix_begin, uint64_t prefix_end) {
  const uint32_t sh = w_prefix - slice_bits;
  if (!d || prefix_begin > prefix_end || prefix_end > (1ull << w_prefix) || (prefix_begin & ((1ull << sh) - 1)) || (prefix_end & ((1ull << sh) - 1))) {
    set_err(&err, "mgc_runs_write: the prefix range must be cut at multiples of 2^%u blocks", sh);
    return MGC_EINVAL;
  }
  DbSink sink(d, w_prefix);
  return deliver(prefix_begin >> sh, prefix_end >> sh, sink);
}

// ---- C ABI ----------------------------------------------------------------------------------------------------------
extern "C" mgc_runs *mgc_runs_open(uint32_t k, uint32_t w_prefix, int device, uint64_t device_budget_bytes, uint64_t chunk_bytes) {
  if (k == 0 || k > 64 || w_prefix < MGC_NUM_FILES_BITS || w_prefix > 2 * k) { set_err(nullptr, "mgc_runs_open: bad arguments"); return nullptr; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_err(nullptr, "mgc_runs_open: no HIP device"); return nullptr; }
  if (device < 0) (void)hipGetDevice(&device);
  return new mgc_runs(k, w_prefix, device, device_budget_bytes, chunk_bytes);
}
extern "C" int mgc_runs_add(mgc_runs *r, const void *d_keys, const uint32_t *d_counts, uint64_t n, void *stream) {
  if (!r || (n && (!d_keys || !d_counts))) return MGC_EINVAL;
  return r->add(d_keys, d_counts, n, (hipStream_t)stream);
}
extern "C" int mgc_runs_write(mgc_runs *r, mgc_db_stream *d, uint64_t prefix_begin, uint64_t prefix_end) {
  if (!r) return MGC_EINVAL;
  return r->write(d, prefix_begin, prefix_end);
}
extern "C" int mgc_runs_get_profile(const mgc_runs *r, mgc_runs_profile *p) {
  if (!r || !p) return MGC_EINVAL;
  *p = r->prof;
  return MGC_OK;
}
extern "C" const char *mgc_runs_error(const mgc_runs *r) { return r ? r->err.c_str() : mgc::thread_last_error().c_str(); }
extern "C" void mgc_runs_close(mgc_runs *r) { delete r; }
