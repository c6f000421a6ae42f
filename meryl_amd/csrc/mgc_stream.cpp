// mgc_stream.cpp -- delivery of a count result: the addBlock-convention callbacks (mgc_finish*) streamed out of HBM
// through pinned buffers, and the database stream (include/meryl_db.h, mgc_db_stream_*) whose blocks are encoded on the
// device (mgc_encode.hip) so that the host only moves finished file bytes.
//
// Reference side of this boundary: the final dump of countThreads, 64 OpenMP threads running
// countKmers + dumpCountedKmers -> merylBlockWriter::addBlock, one file per thread, prefixes ascending, empty blocks
// included (src/meryl/merylOp-countThreads.C:452-459, src/meryl/merylCountArray.C:472-475), then
// merylBlockWriter::finish() / ~merylFileWriter (src/meryl/merylOp-countThreads.C:464, merylOp-nextMer.C:227).
#include "../../include/meryl_db.h"
#include "mdb_layout.h"
#include "mgc_session.hpp"
#include "mgc_runs.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>

using mgc::set_err;

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct DBuf {                                             // grow-only device buffer
  void *p = nullptr; size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes < 256) bytes = 256;
    if (cap >= bytes) return hipSuccess;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

constexpr size_t   SLOT_BYTES = 32u << 20;                // pinned copy buffers: NSLOT x SLOT_BYTES
constexpr int      NSLOT_MAX  = 64;
// device image of one encode chunk (two in flight; ~10 files at 10 Gbp) and pinned copy slots.  MEASURED (profiles/r03o_e2e_db.txt,
// 6.75 GB database): 1 GiB / 16 slots 0.51 s, 2 GiB / 32 slots 0.67 s, 4 GiB / 48 slots 0.86 s -- pinning more slots costs more
// than the extra files in flight give (profiles/r03o_e2e_db.txt, r03p_e2e_db2.txt: other image and slot sizes)
inline uint64_t img_cap() { return 1024ull << 20; }
inline int      n_slot()  { return std::min(NSLOT_MAX, 16); }
#define NSLOT n_slot()
#define IMG_CAP img_cap()

}  // namespace

// ================================================================================================
//  database stream
// ================================================================================================
struct mgc_db_stream {
  // geometry
  uint32_t k = 0, w_prefix = 0, w_data = 0, label_size = 0, kw = 1, num_blocks_bits = 0;
  uint64_t label = 0;
  int      device = 0, n_threads = 1;
  mdb_writer *w = nullptr;

  hipStream_t st_enc = nullptr, st_copy = nullptr;
  hipEvent_t  ev_enc[2] = {nullptr, nullptr}, ev_a = nullptr, ev_b = nullptr;
  DBuf d_bs, d_bytes, d_vbase, d_bb, d_pos, d_hist, d_big, d_img[2];
  char *pinned[NSLOT_MAX] = {nullptr};

  struct Range { const void *keys; const uint32_t *counts; uint64_t n, pb, pe; };
  struct Piece { uint32_t ff; int slot; uint64_t nbytes, file_offset; };

  std::mutex mu;
  std::condition_variable cv;
  bool slot_busy[NSLOT_MAX] = {false}, slot_ready[NSLOT_MAX] = {false};
  int  next_slot = 0;
  std::deque<Range> jobs;
  uint64_t jobs_queued = 0, jobs_done = 0;
  bool closing = false, copy_done = false;
  std::deque<Piece> queue;                                // any pool thread takes any piece: the writes are pwrites
  uint64_t pieces_open = 0;
  std::thread copy_thread;
  std::vector<std::thread> pool;
  int status = MGC_OK;
  std::string err;
  uint64_t next_prefix = 0;                               // ranges must ascend

  std::map<uint64_t, uint64_t> hist_acc;
  mgc_db_write_profile prof;
  double t_open = 0, t_first_copy = 0;

  void fail_locked(int rc, const std::string &msg) { if (status == MGC_OK) { status = rc; err = msg; } }
  void fail(int rc, const std::string &msg) { std::lock_guard<std::mutex> g(mu); fail_locked(rc, msg); cv.notify_all(); }
  int  fail_hip(hipError_t e, const char *what) {
    fail(e == hipErrorOutOfMemory ? MGC_ENOMEM : MGC_EHIP, std::string(what) + ": " + hipGetErrorString(e));
    return status;
  }
  int process(const Range &r);
  void copy_main();
  void pool_main(int t);
};

#define DS_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return fail_hip(e__, #expr); } while (0)

void mgc_db_stream::pool_main(int t) {
  // the pinned copy buffers are allocated here, in parallel, while the first range is being planned and encoded
  // (pinning half a gigabyte from one thread costs as much as writing the database)
  (void)hipSetDevice(device);
  for (int i = t; i < NSLOT; i += n_threads) {
    void *p = nullptr;
    const hipError_t e = hipHostMalloc(&p, SLOT_BYTES, hipHostMallocDefault);
    std::lock_guard<std::mutex> g(mu);
    if (e != hipSuccess) fail_locked(MGC_ENOMEM, std::string("pinned copy buffer: ") + hipGetErrorString(e));
    else { pinned[i] = reinterpret_cast<char *>(p); slot_ready[i] = true; }
    cv.notify_all();
  }
  for (;;) {
    Piece pc;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return !queue.empty() || copy_done; });
      if (queue.empty()) return;
      pc = queue.front();
      queue.pop_front();
    }
    int rc = MGC_OK;
    std::string msg;
    bool skip;
    { std::lock_guard<std::mutex> g(mu); skip = status != MGC_OK; }
    if (!skip) {
      rc = mdb_writer_write_at(w, pc.ff, pc.file_offset, pinned[pc.slot], pc.nbytes);
      if (rc != MGC_OK) msg = std::string("writing the database: ") + mdb_last_error();
    }
    {
      std::lock_guard<std::mutex> g(mu);
      if (rc != MGC_OK) fail_locked(rc, msg);
      slot_busy[pc.slot] = false;
      pieces_open--;
      prof.data_bytes += pc.nbytes;
    }
    cv.notify_all();
  }
}

void mgc_db_stream::copy_main() {
  (void)hipSetDevice(device);
  for (;;) {
    Range r;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return !jobs.empty() || closing; });
      if (jobs.empty()) break;
      r = jobs.front();
    }
    bool run;
    { std::lock_guard<std::mutex> g(mu); run = status == MGC_OK; }
    if (run) (void)process(r);
    {
      std::lock_guard<std::mutex> g(mu);
      jobs.pop_front();
      jobs_done++;
    }
    cv.notify_all();
  }
  { std::lock_guard<std::mutex> g(mu); copy_done = true; }
  cv.notify_all();
}

int mgc_db_stream::process(const Range &r) {
  const double t0 = now_s();
  const uint64_t nblk = r.pe - r.pb;
  const uint32_t ss = w_data;
  DS_TRY(d_bs.ensure(8 * (nblk + 1)));
  DS_TRY(d_bytes.ensure(8 * nblk));
  DS_TRY(d_vbase.ensure(8 * nblk));
  DS_TRY(d_bb.ensure(4 * nblk));
  DS_TRY(d_pos.ensure(8 * nblk));
  const uint32_t nsmall = mgc::value_hist_small_bins();
  DS_TRY(d_hist.ensure(8 * (nsmall + 1)));
  if (d_big.cap == 0) DS_TRY(d_big.ensure(4u << 20));
  DS_TRY(mgc::launch_block_offsets_range(r.keys, r.n, kw, w_data, r.pb, nblk, 1ull << w_prefix, d_bs.as<uint64_t>(), st_enc));
  DS_TRY(mgc::launch_encode_sizes(r.keys, kw, d_bs.as<uint64_t>(), nblk, ss, label_size, d_bytes.as<uint64_t>(),
                                  d_vbase.as<uint64_t>(), d_bb.as<uint32_t>(), st_enc));
  std::vector<uint64_t> h_bs(nblk + 1), h_bytes(nblk), h_pos(nblk), h_hist(nsmall + 1);
  DS_TRY(hipMemcpyAsync(h_bs.data(), d_bs.p, 8 * (nblk + 1), hipMemcpyDeviceToHost, st_enc));
  if (nblk) DS_TRY(hipMemcpyAsync(h_bytes.data(), d_bytes.p, 8 * nblk, hipMemcpyDeviceToHost, st_enc));
  // value histogram (A9): small values in LDS bins, the rare large ones as a list
  for (int attempt = 0; attempt < 2; attempt++) {
    DS_TRY(hipMemsetAsync(d_hist.p, 0, 8 * (nsmall + 1), st_enc));
    DS_TRY(mgc::launch_value_hist(r.counts, r.n, d_hist.as<uint64_t>(), d_big.as<uint32_t>(), d_big.cap / 4,
                                  d_hist.as<uint64_t>() + nsmall, st_enc));
    DS_TRY(hipMemcpyAsync(h_hist.data(), d_hist.p, 8 * (nsmall + 1), hipMemcpyDeviceToHost, st_enc));
    DS_TRY(hipStreamSynchronize(st_enc));
    if (h_hist[nsmall] <= d_big.cap / 4) break;
    DS_TRY(d_big.ensure(4 * h_hist[nsmall]));            // the list overflowed: once more with room for all of it
  }
  if (h_bs[nblk] != r.n || h_bs[0] != 0) { fail(MGC_EINVAL, "mgc_db_stream_write: keys outside the prefix range (or not ascending)"); return status; }
  {
    std::vector<uint32_t> big(h_hist[nsmall]);
    if (!big.empty()) DS_TRY(hipMemcpy(big.data(), d_big.p, 4 * big.size(), hipMemcpyDeviceToHost));
    std::lock_guard<std::mutex> g(mu);
    for (uint32_t v = 0; v < nsmall; v++) if (h_hist[v]) hist_acc[v] += h_hist[v];
    for (uint32_t v : big) hist_acc[v]++;
  }

  // chunks: runs of blocks whose dumped bytes fit one device image (a larger single block gets an image of its own)
  struct Chunk { uint64_t b0, b1, bytes; };
  std::vector<Chunk> chunks;
  for (uint64_t b = 0; b < nblk;) {
    Chunk c; c.b0 = b; c.bytes = 0;
    while (b < nblk && (c.bytes == 0 || c.bytes + h_bytes[b] <= IMG_CAP)) { h_pos[b] = c.bytes; c.bytes += h_bytes[b]; b++; }
    c.b1 = b;
    chunks.push_back(c);
  }
  if (nblk) DS_TRY(hipMemcpyAsync(d_pos.p, h_pos.data(), 8 * nblk, hipMemcpyHostToDevice, st_enc));
  const double t1 = now_s();

  auto encode = [&](size_t ci) -> int {
    const Chunk &c = chunks[ci];
    DBuf &img = d_img[ci & 1];
    DS_TRY(img.ensure(c.bytes));
    DS_TRY(hipMemsetAsync(img.p, 0, c.bytes, st_enc));
    DS_TRY(hipEventRecord(ev_a, st_enc));
    DS_TRY(mgc::launch_encode_chunk(r.keys, r.counts, kw, d_bs.as<uint64_t>(), d_pos.as<uint64_t>(), d_vbase.as<uint64_t>(),
                                    d_bb.as<uint32_t>(), c.b0, c.b1, h_bs[c.b1] - h_bs[c.b0], r.pb, ss, label_size, label,
                                    img.p, st_enc));
    DS_TRY(hipEventRecord(ev_enc[ci & 1], st_enc));
    return MGC_OK;
  };

  const uint64_t blocks_per_file = 1ull << num_blocks_bits;
  double enc_ms = 0, t_slot = 0, t_copy = 0, t_encwait = 0;
  uint64_t n_pieces = 0;
  if (!chunks.empty() && encode(0) != MGC_OK) return status;
  for (size_t ci = 0; ci < chunks.size(); ci++) {
    const Chunk &c = chunks[ci];
    { const double tw = now_s(); DS_TRY(hipEventSynchronize(ev_enc[ci & 1])); t_encwait += now_s() - tw; }
    { float ms = 0; if (hipEventElapsedTime(&ms, ev_a, ev_enc[ci & 1]) == hipSuccess) enc_ms += ms; }
    if (ci + 1 < chunks.size() && encode(ci + 1) != MGC_OK) return status;     // runs while this chunk is copied out
    const unsigned char *img = d_img[ci & 1].as<unsigned char>();
    auto pos_at = [&](uint64_t j) { return j < c.b1 ? h_pos[j] : c.bytes; };
    // 1. cut the chunk into pieces (whole blocks where possible, never across files) and fix every piece's place in its
    //    file -- index entries and offsets are reserved here, in prefix order
    struct PieceDesc { uint64_t a; Piece pc; };
    std::vector<std::vector<PieceDesc>> by_file;          // pieces of the chunk's files, file-major
    uint64_t a = 0, bi = c.b0;                            // bi = first block that starts at or after byte a
    uint32_t last_ff = ~0u;
    while (a < c.bytes) {
      const uint64_t cur_blk = (bi < c.b1 && h_pos[bi] == a) ? bi : bi - 1;
      const uint32_t ff = (uint32_t)((r.pb + cur_blk) >> num_blocks_bits);
      const uint64_t file_end_abs = ((uint64_t)ff + 1) * blocks_per_file;                  // first prefix of the next file
      const uint64_t file_end_blk = std::min<uint64_t>(c.b1, file_end_abs > r.pb ? file_end_abs - r.pb : 0);
      const uint64_t lim = std::min<uint64_t>(a + SLOT_BYTES, pos_at(file_end_blk));
      uint64_t j = bi;
      while (j < file_end_blk && pos_at(j + 1) <= lim) j++;                                // blocks bi..j-1 end inside the piece
      uint64_t e;
      if (a < pos_at(bi)) {                               // the piece starts inside block bi-1
        if (pos_at(bi) <= lim) e = pos_at(j); else { e = lim; j = bi; }
      } else if (j > bi) {
        e = pos_at(j);
      } else {                                            // block bi alone is larger than a copy buffer
        e = lim; j = bi + 1;
      }
      PieceDesc pd;
      pd.a = a;
      pd.pc.ff = ff; pd.pc.nbytes = e - a; pd.pc.slot = -1; pd.pc.file_offset = 0;
      std::vector<mdb_index_entry> entries;
      for (uint64_t jj = bi; jj < j; jj++) {
        mdb_index_entry en;
        en.prefix = r.pb + jj; en.position = h_pos[jj] - a; en.n_kmers = h_bs[jj + 1] - h_bs[jj];
        entries.push_back(en);
      }
      if (mdb_writer_reserve_encoded(w, ff, pd.pc.nbytes, entries.data(), entries.size(), &pd.pc.file_offset) != MGC_OK) {
        fail(MGC_EINVAL, std::string("writing the database: ") + mdb_last_error());
        return status;
      }
      if (ff != last_ff) { by_file.emplace_back(); last_ff = ff; }
      by_file.back().push_back(pd);
      a = e; bi = j;
    }
    // 2. copy them out round-robin over the chunk's files: writes to ONE file serialise in the kernel (inode lock:
    //    6-8 GB/s on tmpfs whatever the thread count), writes to different files do not
    size_t left = 0;
    for (auto &v : by_file) left += v.size();
    std::vector<size_t> next(by_file.size(), 0);
    while (left) {
      for (size_t fi = 0; fi < by_file.size(); fi++) {
        if (next[fi] >= by_file[fi].size()) continue;
        PieceDesc &pd = by_file[fi][next[fi]++];
        left--;
        Piece pc = pd.pc;
        const double ts0 = now_s();
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return (slot_ready[next_slot] && !slot_busy[next_slot]) || status != MGC_OK; });
          if (status != MGC_OK) return status;
          pc.slot = next_slot;
          slot_busy[next_slot] = true;
          next_slot = (next_slot + 1) % NSLOT;
        }
        const double ts1 = now_s();
        hipError_t he = hipMemcpyAsync(pinned[pc.slot], img + pd.a, pc.nbytes, hipMemcpyDeviceToHost, st_copy);
        if (he == hipSuccess) he = hipStreamSynchronize(st_copy);
        t_slot += ts1 - ts0; t_copy += now_s() - ts1; n_pieces++;
        if (he != hipSuccess) {
          { std::lock_guard<std::mutex> g(mu); slot_busy[pc.slot] = false; }
          return fail_hip(he, "copying encoded blocks to the host");
        }
        {
          std::lock_guard<std::mutex> g(mu);
          pieces_open++;
          queue.push_back(pc);
        }
        cv.notify_all();
      }
    }
  }
  const double t2 = now_s();
  if (getenv("MGC_IO_TRACE"))
    fprintf(stderr, "[io] db range: %llu pieces in %zu chunks: waiting for a free pinned slot %.3f s, device-to-host copies %.3f s, "
                    "waiting for the encoder %.3f s, plan %.3f s\n", (unsigned long long)n_pieces, chunks.size(), t_slot, t_copy, t_encwait, t1 - t0);
  std::lock_guard<std::mutex> g(mu);
  prof.plan_ms += (t1 - t0) * 1e3;
  prof.encode_ms += enc_ms;
  prof.copy_write_s += t2 - t1;
  prof.n_kmers += r.n;
  prof.n_blocks += nblk;
  return MGC_OK;
}

extern "C" const char *mgc_db_stream_error(const mgc_db_stream *d) {
  if (!d) return mgc::thread_last_error().c_str();
  // the stream's threads may set the (sticky, written once) error while this is read: copy it under the lock
  std::lock_guard<std::mutex> g(const_cast<mgc_db_stream *>(d)->mu);
  mgc::thread_last_error() = d->err;
  return mgc::thread_last_error().c_str();
}

extern "C" mgc_db_stream *mgc_db_stream_open(const char *path, uint32_t k, uint32_t w_prefix, uint32_t label_size, uint64_t label,
                                             uint32_t part, uint32_t n_parts, int host_threads, int device) {
  if (!path || k == 0 || k > 64 || w_prefix < MGC_NUM_FILES_BITS || w_prefix > 2 * k || w_prefix > 40 + MGC_NUM_FILES_BITS) {
    set_err(nullptr, "mgc_db_stream_open: bad arguments");
    return nullptr;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_err(nullptr, "mgc_db_stream_open: no HIP device"); return nullptr; }
  mgc_db_stream *d = new mgc_db_stream();
  d->k = k; d->w_prefix = w_prefix; d->w_data = 2 * k - w_prefix; d->label_size = label_size; d->label = label;
  d->kw = k > 32 ? 2u : 1u;
  d->num_blocks_bits = w_prefix - MGC_NUM_FILES_BITS;
  memset(&d->prof, 0, sizeof(d->prof));
  d->t_open = now_s();
  if (device < 0) (void)hipGetDevice(&device);
  d->device = device;
  if (host_threads <= 0) host_threads = (int)std::thread::hardware_concurrency();
  d->n_threads = std::max(1, std::min(host_threads, MGC_NUM_FILES));
  d->w = mdb_writer_open_ex(path, k, w_prefix, label_size, part, n_parts);
  if (!d->w) { set_err(nullptr, "mgc_db_stream_open: %s", mdb_last_error()); delete d; return nullptr; }
  bool ok = hipSetDevice(device) == hipSuccess &&
            hipStreamCreateWithFlags(&d->st_enc, hipStreamNonBlocking) == hipSuccess &&
            hipStreamCreateWithFlags(&d->st_copy, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreate(&d->ev_enc[0]) == hipSuccess && hipEventCreate(&d->ev_enc[1]) == hipSuccess &&
            hipEventCreate(&d->ev_a) == hipSuccess && hipEventCreate(&d->ev_b) == hipSuccess;
  if (!ok) {
    set_err(nullptr, "mgc_db_stream_open: HIP stream / pinned buffer setup failed");
    d->closing = true; d->copy_done = true;
    mdb_writer_discard(d->w);                               // no empty database at the output path when the open fails
    for (int i = 0; i < NSLOT; i++) if (d->pinned[i]) (void)hipHostFree(d->pinned[i]);
    delete d;
    return nullptr;
  }
  d->copy_thread = std::thread([d] { d->copy_main(); });
  for (int t = 0; t < d->n_threads; t++) d->pool.emplace_back([d, t] { d->pool_main(t); });
  return d;
}

extern "C" int mgc_db_stream_write(mgc_db_stream *d, const void *d_keys, const uint32_t *d_counts, uint64_t n,
                                   uint64_t prefix_begin, uint64_t prefix_end) {
  if (!d) return MGC_EINVAL;
  std::lock_guard<std::mutex> g(d->mu);
  if (d->status != MGC_OK) return d->status;
  if (prefix_begin > prefix_end || prefix_end > (1ull << d->w_prefix) || prefix_begin < d->next_prefix || (n && (!d_keys || !d_counts))) {
    d->fail_locked(MGC_EINVAL, "mgc_db_stream_write: prefix ranges must ascend inside [0, 2^wPrefix)");
    return d->status;
  }
  if (d->closing) { d->fail_locked(MGC_ESTATE, "mgc_db_stream_write after close"); return d->status; }
  d->next_prefix = prefix_end;
  if (prefix_begin == prefix_end) return n ? MGC_EINVAL : MGC_OK;
  mgc_db_stream::Range r;
  r.keys = d_keys; r.counts = d_counts; r.n = n; r.pb = prefix_begin; r.pe = prefix_end;
  d->jobs.push_back(r);
  d->jobs_queued++;
  d->cv.notify_all();
  return MGC_OK;
}

extern "C" uint64_t mgc_db_stream_queued(mgc_db_stream *d) {
  if (!d) return 0;
  std::lock_guard<std::mutex> g(d->mu);
  return d->jobs_queued;
}

extern "C" uint64_t mgc_db_stream_done(mgc_db_stream *d) {
  if (!d) return 0;
  std::lock_guard<std::mutex> g(d->mu);
  return d->jobs_done;
}

extern "C" int mgc_db_stream_wait_buffers(mgc_db_stream *d, uint64_t upto) {
  if (!d) return MGC_EINVAL;
  std::unique_lock<std::mutex> lk(d->mu);
  d->cv.wait(lk, [&] { return d->jobs_done >= upto || d->jobs_done == d->jobs_queued; });
  return d->status;
}

extern "C" int mgc_db_stream_sync(mgc_db_stream *d) {
  if (!d) return MGC_EINVAL;
  std::unique_lock<std::mutex> lk(d->mu);
  d->cv.wait(lk, [&] { return d->jobs_done == d->jobs_queued && d->pieces_open == 0; });
  return d->status;
}

extern "C" int mgc_db_stream_close(mgc_db_stream *d, mgc_db_write_profile *prof) {
  if (!d) return MGC_EINVAL;
  { std::lock_guard<std::mutex> g(d->mu); d->closing = true; }
  d->cv.notify_all();
  d->copy_thread.join();
  for (auto &t : d->pool) t.join();
  int rc = d->status;
  if (rc == MGC_OK) {
    std::vector<uint64_t> hv, ho;
    for (auto &kv : d->hist_acc) { hv.push_back(kv.first); ho.push_back(kv.second); }
    rc = mdb_writer_add_histogram(d->w, hv.data(), ho.data(), hv.size());
  }
  const int rc2 = mdb_writer_close(d->w);
  if (rc == MGC_OK && rc2 != MGC_OK) { rc = rc2; d->err = std::string("closing the database: ") + mdb_last_error(); }
  if (rc != MGC_OK) set_err(nullptr, "%s", d->err.c_str());
  (void)hipSetDevice(d->device);
  d->d_bs.release(); d->d_bytes.release(); d->d_vbase.release(); d->d_bb.release(); d->d_pos.release();
  d->d_hist.release(); d->d_big.release(); d->d_img[0].release(); d->d_img[1].release();
  for (int i = 0; i < NSLOT; i++) if (d->pinned[i]) (void)hipHostFree(d->pinned[i]);
  for (int i = 0; i < 2; i++) if (d->ev_enc[i]) (void)hipEventDestroy(d->ev_enc[i]);
  if (d->ev_a) (void)hipEventDestroy(d->ev_a);
  if (d->ev_b) (void)hipEventDestroy(d->ev_b);
  if (d->st_enc) (void)hipStreamDestroy(d->st_enc);
  if (d->st_copy) (void)hipStreamDestroy(d->st_copy);
  d->prof.total_s = now_s() - d->t_open;
  if (prof) *prof = d->prof;
  delete d;
  return rc;
}

// ================================================================================================
//  session result -> database
// ================================================================================================
extern "C" int mgc_write_database_profiled(mgc_session *s, const char *path, int host_threads, mgc_db_write_profile *prof) {
  if (!s || !path) return MGC_EINVAL;
  if (!s->counted) { set_err(&s->err, "mgc_write_database before mgc_count"); return MGC_ESTATE; }
  if (prof) memset(prof, 0, sizeof(*prof));
  const mgc_count_config &c = s->cfg;
  mgc_db_stream *d = mgc_db_stream_open(path, c.k, c.w_prefix, c.label_size, c.label_constant, 0, 1, host_threads, s->device);
  if (!d) { set_err(&s->err, "%s", mgc_db_stream_error(nullptr)); return MGC_EINVAL; }
  int rc;
  std::string msg;
  if (s->ooc) {
    // out of core: the runs are merged chunk by chunk straight into the stream (merylBlockWriter::finish() merging the
    // iterations, merylOp-countThreads.C:461-464)
    const uint64_t before = s->runs->prof.n_merged;
    rc = s->runs->write(d, 0, c.n_prefix);
    if (rc != MGC_OK) msg = s->runs->err;
    else s->n_distinct = s->runs->prof.n_merged - before;
  } else {
    rc = mgc_db_stream_write(d, s->d_unique, s->d_counts, s->n_distinct, 0, c.n_prefix);
    if (rc != MGC_OK) msg = mgc_db_stream_error(d);
  }
  const int rc2 = mgc_db_stream_close(d, prof);
  if (rc == MGC_OK && rc2 != MGC_OK) { rc = rc2; msg = mgc_db_stream_error(nullptr); }
  if (rc != MGC_OK) set_err(&s->err, "%s", msg.c_str());
  return rc;
}

extern "C" int mgc_write_database(mgc_session *s, const char *path, int host_threads) {
  return mgc_write_database_profiled(s, path, host_threads, nullptr);
}

// ================================================================================================
//  mgc_finish: the addBlock convention, streamed
// ================================================================================================
namespace {

struct PinnedBuf {
  void *p = nullptr; size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (cap >= bytes) return hipSuccess;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 256, hipHostMallocDefault);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  ~PinnedBuf() { if (p) (void)hipHostFree(p); }
};

// the blocks of prefixes [pb, pe): n_keys ascending distinct k-mers + counts in device memory, bstart[i] = first k-mer of
// prefix pb + i (relative to the view, pe - pb + 1 entries, host)
struct ResultView { const void *d_keys; const uint32_t *d_counts; uint64_t pb, pe; const uint64_t *bstart; };

int deliver_view(mgc_session *s, const ResultView &view, mgc_block_cb cb1, mgc_block_cb2 cb2, void *ctx, int host_threads) {
  const uint64_t np = s->cfg.n_prefix;
  const bool wide = s->key_words == 2;
  const uint32_t kw = s->key_words;
  const uint64_t label = s->cfg.label_constant;

  // suffix = low w_data bits of the k-mer (wDataMask, merylOp-count.C:282-286)
  const uint32_t w_data = s->cfg.w_data;
  const uint64_t mask_lo = (w_data >= 64) ? ~0ull : ((1ull << w_data) - 1ull);
  const uint64_t mask_hi = (w_data <= 64) ? 0ull : ((w_data >= 128) ? ~0ull : ((1ull << (w_data - 64)) - 1ull));
  const uint64_t per_file = np / MGC_NUM_FILES;             // firstPrefixInFile/lastPrefixInFile
  if (host_threads <= 0) host_threads = (int)(s->cfg.threads ? s->cfg.threads : std::thread::hardware_concurrency());
  host_threads = std::max(1, std::min(host_threads, MGC_NUM_FILES));
  HIP_TRY(s, hipSetDevice(s->device));
  const uint64_t *bstart = view.bstart - view.pb;           // indexed by absolute prefix

  auto deliver = [&](uint64_t pp, uint64_t n, const uint64_t *slo, const uint64_t *shi, const uint32_t *cn) -> int {
    return cb2 ? cb2(ctx, pp, n, slo, shi, cn, nullptr, label) : cb1(ctx, pp, n, slo, shi, cn);
  };

  // Every worker owns one file at a time (the reference's `omp parallel for schedule(dynamic,1)` over files) and streams
  // its k-mers out of HBM in pieces of whole blocks through two pinned buffers on its own HIP stream: the copy of piece
  // i+1 runs while the callbacks of piece i do -- no host copy of the whole result is ever made.
  const uint64_t piece_kmers = host_threads > 16 ? (512u << 10) : (1u << 20);
  std::atomic<uint32_t> next_file(0);
  std::atomic<int> status(MGC_OK);
  std::mutex err_mu;
  auto worker = [&]() {
    std::vector<uint64_t> slo, shi;
    hipStream_t st = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    PinnedBuf pk[2], pc[2];
    auto hip_fail = [&](hipError_t e, const char *what) {
      std::lock_guard<std::mutex> g(err_mu);
      set_err(&s->err, "mgc_finish: %s: %s", what, hipGetErrorString(e));
      status.store(e == hipErrorOutOfMemory ? MGC_ENOMEM : MGC_EHIP);
    };
    {
      hipError_t e = hipSetDevice(s->device);
      if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[0], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
      if (e != hipSuccess) { hip_fail(e, "stream setup"); return; }
    }
    struct Span { uint64_t p0, p1; };
    auto plan = [&](uint64_t from, uint64_t file_end) {     // whole blocks, about piece_kmers k-mers, at least one block
      Span sp; sp.p0 = from; sp.p1 = from;
      uint64_t n = 0;
      while (sp.p1 < file_end) {
        const uint64_t nb = bstart[sp.p1 + 1] - bstart[sp.p1];
        if (sp.p1 > from && n + nb > piece_kmers) break;
        n += nb; sp.p1++;
      }
      return sp;
    };
    auto issue = [&](const Span &sp, int b) -> bool {       // device -> pinned buffer b
      const uint64_t k0 = bstart[sp.p0], n = bstart[sp.p1] - k0;
      if (n == 0) return true;
      hipError_t e = pk[b].ensure(sizeof(uint64_t) * kw * n);
      if (e == hipSuccess) e = pc[b].ensure(sizeof(uint32_t) * n);
      if (e == hipSuccess) e = hipMemcpyAsync(pk[b].p, reinterpret_cast<const unsigned char *>(view.d_keys) + sizeof(uint64_t) * kw * k0,
                                              sizeof(uint64_t) * kw * n, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipMemcpyAsync(pc[b].p, view.d_counts + k0, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipEventRecord(ev[b], st);
      if (e != hipSuccess) { hip_fail(e, "device-to-host copy"); return false; }
      return true;
    };
    for (;;) {
      const uint32_t ff = next_file.fetch_add(1);            // dynamic,1 like the reference's omp schedule
      if (ff >= MGC_NUM_FILES || status.load() != MGC_OK) break;
      const uint64_t f0 = std::max<uint64_t>(ff * per_file, view.pb), f1 = std::min<uint64_t>((ff + 1) * per_file, view.pe);
      if (f0 >= f1) continue;                                // the view holds nothing of this file
      int b = 0;
      Span cur = plan(f0, f1);
      if (!issue(cur, b)) break;
      bool stop = false;
      while (cur.p0 < f1 && !stop) {
        const Span nxt = plan(cur.p1, f1);
        if (nxt.p0 < f1 && !issue(nxt, b ^ 1)) { stop = true; break; }
        const uint64_t k0 = bstart[cur.p0], n = bstart[cur.p1] - k0;
        const uint64_t *keys = nullptr;
        const uint32_t *cnts = nullptr;
        if (n) {
          hipError_t e = hipEventSynchronize(ev[b]);
          if (e != hipSuccess) { hip_fail(e, "device-to-host copy"); stop = true; break; }
          keys = reinterpret_cast<const uint64_t *>(pk[b].p);
          cnts = reinterpret_cast<const uint32_t *>(pc[b].p);
        }
        for (uint64_t pp = cur.p0; pp < cur.p1; pp++) {
          const uint64_t o = bstart[pp] - k0, m = bstart[pp + 1] - bstart[pp];
          slo.resize(m);
          if (wide) shi.resize(m);
          if (wide) {
            for (uint64_t i = 0; i < m; i++) { slo[i] = keys[2 * (o + i)] & mask_lo; shi[i] = keys[2 * (o + i) + 1] & mask_hi; }
          } else {
            for (uint64_t i = 0; i < m; i++) slo[i] = keys[o + i] & mask_lo;
          }
          const int r = deliver(pp, m, slo.data(), wide ? shi.data() : nullptr, m ? cnts + o : nullptr);   // empty blocks too
          if (r != 0) { status.store(r); stop = true; break; }
        }
        cur = nxt;
        b ^= 1;
      }
      if (stop) break;
    }
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (int i = 0; i < 2; i++) if (ev[i]) (void)hipEventDestroy(ev[i]);
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < host_threads; t++) pool.emplace_back(worker);
  worker();
  for (auto &t : pool) t.join();
  return status.load();
}

// out-of-core results: every merged chunk of the runs is delivered like a (partial) device result -- synchronously, so a
// chunk's buffers are free again when put() returns.  Chunks ascend, so every file still sees its prefixes in order.
struct CallbackSink : mgc::RunSink {
  mgc_session *s; mgc_block_cb cb1; mgc_block_cb2 cb2; void *ctx; int host_threads;
  void *d_bs = nullptr; size_t bs_cap = 0;
  std::string msg;
  uint64_t n_delivered = 0;
  CallbackSink(mgc_session *s_, mgc_block_cb a, mgc_block_cb2 b, void *c, int t) : s(s_), cb1(a), cb2(b), ctx(c), host_threads(t) {}
  ~CallbackSink() override { if (d_bs) (void)hipFree(d_bs); }
  int put(const void *k, const uint32_t *c, uint64_t n, uint64_t sa, uint64_t sb, uint32_t slice_bits, uint64_t *job) override {
    *job = 0;
    const uint32_t sh = s->cfg.w_prefix - slice_bits;
    const uint64_t pb = sa << sh, pe = sb << sh, nblk = pe - pb;
    const size_t need = sizeof(uint64_t) * (nblk + 1);
    if (bs_cap < need) {
      if (d_bs) (void)hipFree(d_bs);
      d_bs = nullptr; bs_cap = 0;
      if (hipMalloc(&d_bs, need) != hipSuccess) { msg = "mgc_finish: out of device memory for the block offsets of a chunk"; return MGC_ENOMEM; }
      bs_cap = need;
    }
    std::vector<uint64_t> h_bs(nblk + 1);
    hipError_t e = mgc::launch_block_offsets_range(k, n, s->key_words, s->cfg.w_data, pb, nblk, s->cfg.n_prefix, reinterpret_cast<uint64_t *>(d_bs), s->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h_bs.data(), d_bs, need, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess) { msg = std::string("mgc_finish: block offsets of a chunk: ") + hipGetErrorString(e); return MGC_EHIP; }
    if (h_bs[0] != 0 || h_bs[nblk] != n) { msg = "mgc_finish: a merged chunk holds k-mers outside its prefix range"; return MGC_EINVAL; }
    ResultView v{k, c, pb, pe, h_bs.data()};
    const int rc = deliver_view(s, v, cb1, cb2, ctx, host_threads);
    if (rc != MGC_OK) msg = s->err.empty() ? "mgc_finish: the callback stopped the delivery" : s->err;
    n_delivered += n;
    return rc;
  }
  int wait(uint64_t) override { return MGC_OK; }
  const char *error() const override { return msg.c_str(); }
};

int finish_impl(mgc_session *s, mgc_block_cb cb1, mgc_block_cb2 cb2, void *ctx, int host_threads) {
  if (!s || (!cb1 && !cb2)) return MGC_EINVAL;
  if (!s->counted) { set_err(&s->err, "mgc_finish before mgc_count"); return MGC_ESTATE; }
  const uint64_t np = s->cfg.n_prefix;
  if (s->ooc) {
    CallbackSink sink(s, cb1, cb2, ctx, host_threads);
    const int rc = s->runs->deliver(0, s->runs->n_slices, sink);
    if (rc != MGC_OK && s->err.empty()) s->err = s->runs->err;
    if (rc == MGC_OK) s->n_distinct = sink.n_delivered;
    return rc;
  }
  // block boundaries first (small); the k-mers follow file by file
  std::vector<uint64_t> bstart_own(np + 1);
  HIP_TRY(s, hipSetDevice(s->device));
  HIP_TRY(s, hipMemcpy(bstart_own.data(), s->d_block_start, sizeof(uint64_t) * (np + 1), hipMemcpyDeviceToHost));
  ResultView v{s->d_unique, s->d_counts, 0, np, bstart_own.data()};
  return deliver_view(s, v, cb1, cb2, ctx, host_threads);
}

}  // namespace

extern "C" int mgc_finish(mgc_session *s, mgc_block_cb cb, void *ctx, int host_threads) {
  return finish_impl(s, cb, nullptr, ctx, host_threads);
}

extern "C" int mgc_finish_labelled(mgc_session *s, mgc_block_cb2 cb, void *ctx, int host_threads) {
  return finish_impl(s, nullptr, cb, ctx, host_threads);
}

// ================================================================================================
//  union-sum & friends over whole databases
// ================================================================================================
// The reference streams the 64 file slices of its inputs through merylOperation::nextMer -- smallest k-mer over the
// inputs, values of the inputs that hold it combined (src/meryl/merylOp-nextMer.C:418-683; one slice per thread,
// src/meryl/meryl.C:250-263).  Here a slice is decoded by host threads (one per input), merged two inputs at a time on the
// device (mgc_merge.hip), encoded on the device and written by the database stream -- the same merge that folds the
// batches of an out-of-core count.
namespace {
// The slice (file ff) of every input in HBM: in_k[i] / in_c[i] / hn[i].  The data file's bytes are read by one host thread per
// input, uploaded as they are and DECODED ON THE DEVICE (mgc_decode.hip: one thread per block) -- the host only does I/O and
// checks the framing.  MGC_DECODE_HOST=1, or a file framed in a way only the host decoder follows: decoded by the host
// threads (65-70 M k-mers/s each) and uploaded as arrays, as before.
// MGC_DECODE_HOST=1 (tests): read ONCE per merge / filter operation, by its entry point
bool decode_on_host() { const char *e = getenv("MGC_DECODE_HOST"); return e && e[0] == '1'; }
int load_slices(std::vector<mdb_reader *> &rd, uint32_t ff, uint32_t kw, std::vector<DBuf> &in_k, std::vector<DBuf> &in_c,
                std::vector<uint64_t> &hn, hipStream_t st, std::string *msg, bool host_decode) {
  const uint32_t n_inputs = (uint32_t)rd.size();
  struct Raw { unsigned char *bytes = nullptr; uint64_t size = 0; mdb_raw_block *blocks = nullptr; uint64_t nb = 0; bool on_device = false; };
  std::vector<Raw> raw(n_inputs);
  std::vector<std::vector<uint64_t>> hk(n_inputs);
  std::vector<uint32_t *> hc(n_inputs, nullptr);
  std::vector<int> rrc(n_inputs, MGC_OK);
  std::vector<std::string> rmsg(n_inputs);
  std::vector<mdb_info> infos(n_inputs);
  hn.assign(n_inputs, 0);
  {
    std::vector<std::thread> th;
    for (uint32_t i = 0; i < n_inputs; i++)
      th.emplace_back([&, i]() {
        mdb_reader_info(rd[i], &infos[i]);
        if (!host_decode) {
          const int rc = mdb_reader_raw_file(rd[i], ff, &raw[i].bytes, &raw[i].size, &raw[i].blocks, &raw[i].nb, &hn[i]);
          if (rc == MGC_OK) { raw[i].on_device = true; return; }
          if (rc != MGC_EUNSUPPORTED) { rrc[i] = rc; rmsg[i] = mdb_last_error(); return; }
        }
        uint64_t *lo = nullptr, *hi = nullptr;
        rrc[i] = mdb_reader_read_file_ex(rd[i], ff, &lo, &hi, &hc[i], nullptr, &hn[i]);
        if (rrc[i] != MGC_OK) { rmsg[i] = mdb_last_error(); return; }
        hk[i].resize((size_t)kw * hn[i]);
        if (kw == 1) { if (hn[i]) memcpy(hk[i].data(), lo, 8 * hn[i]); }
        else for (uint64_t j = 0; j < hn[i]; j++) { hk[i][2 * j] = lo[j]; hk[i][2 * j + 1] = hi[j]; }
        mdb_free(lo); mdb_free(hi);
      });
    for (auto &t : th) t.join();
  }
  int rc = MGC_OK;
  for (uint32_t i = 0; i < n_inputs; i++)
    if (rrc[i] != MGC_OK && rc == MGC_OK) { rc = rrc[i]; *msg = rmsg[i]; }
  hipError_t e = hipSuccess;
  DBuf d_file, d_blocks, d_err;
  uint32_t h_err = 0;
  if (rc == MGC_OK) e = d_err.ensure(256);
  for (uint32_t i = 0; i < n_inputs && rc == MGC_OK && e == hipSuccess; i++) {
    e = in_k[i].ensure(8 * (size_t)kw * hn[i]);
    if (e == hipSuccess) e = in_c[i].ensure(4 * hn[i]);
    if (e != hipSuccess || !hn[i]) continue;
    if (raw[i].on_device) {
      e = d_file.ensure(raw[i].size + 16);
      if (e == hipSuccess) e = d_blocks.ensure(sizeof(mdb_raw_block) * raw[i].nb);
      if (e == hipSuccess) e = hipMemsetAsync(d_err.p, 0, 4, st);
      if (e == hipSuccess) e = hipMemcpyAsync(d_file.p, raw[i].bytes, raw[i].size + 16, hipMemcpyHostToDevice, st);
      if (e == hipSuccess) e = hipMemcpyAsync(d_blocks.p, raw[i].blocks, sizeof(mdb_raw_block) * raw[i].nb, hipMemcpyHostToDevice, st);
      if (e == hipSuccess) e = mgc::launch_decode_blocks(d_file.p, d_blocks.p, raw[i].nb, infos[i].suffix_size, infos[i].label_size, kw, in_k[i].p,
                                                         in_c[i].as<uint32_t>(), d_err.as<uint32_t>(), st);
      if (e == hipSuccess) e = hipMemcpyAsync(&h_err, d_err.p, 4, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);                     // d_file is reused by the next input
      if (e == hipSuccess && h_err) { rc = MGC_EINVAL; *msg = "corrupt block in a database file (device decoder, code " + std::to_string(h_err) + ")"; }
    } else {
      e = hipMemcpyAsync(in_k[i].p, hk[i].data(), 8 * (size_t)kw * hn[i], hipMemcpyHostToDevice, st);
      if (e == hipSuccess) e = hipMemcpyAsync(in_c[i].p, hc[i], 4 * hn[i], hipMemcpyHostToDevice, st);
    }
  }
  if (e == hipSuccess && rc == MGC_OK) e = hipStreamSynchronize(st);
  for (uint32_t i = 0; i < n_inputs; i++) { mdb_free(hc[i]); mdb_free(raw[i].bytes); mdb_free(raw[i].blocks); }
  d_file.release(); d_blocks.release(); d_err.release();
  if (e != hipSuccess) { rc = (e == hipErrorOutOfMemory) ? MGC_ENOMEM : MGC_EHIP; *msg = std::string("uploading a slice: ") + hipGetErrorString(e); }
  return rc;
}

// inputs of one k, no labels, no multisets -> readers + the first one's info; false: message in the thread error
bool open_merge_inputs(const char *const *inputs, uint32_t n_inputs, std::vector<mdb_reader *> &rd, mdb_info *first, const char *who) {
  rd.assign(n_inputs, nullptr);
  auto close_all = [&]() { for (mdb_reader *r : rd) if (r) mdb_reader_close(r); };
  memset(first, 0, sizeof(*first));
  for (uint32_t i = 0; i < n_inputs; i++) {
    rd[i] = inputs[i] ? mdb_reader_open(inputs[i]) : nullptr;
    if (!rd[i]) { set_err(nullptr, "%s: %s", who, mdb_last_error()); close_all(); return false; }
    mdb_info inf;
    mdb_reader_info(rd[i], &inf);
    // the merge combines VALUES only: labels would be dropped and a multiset's repeated k-mers folded -- refuse rather than
    // write something that silently differs
    if (inf.label_size != 0 || (inf.flags & 1u)) {
      set_err(nullptr, "%s: '%s' %s: not supported here", who, inputs[i], inf.label_size ? "stores labels" : "is a multiset");
      close_all();
      return false;
    }
    if (i == 0) *first = inf;
    else if (inf.k != first->k) {
      set_err(nullptr, "%s: '%s' holds %u-mers, '%s' %u-mers", who, inputs[i], inf.k, inputs[0], first->k);   // merylOp.C: kmer size mismatch
      close_all();
      return false;
    }
  }
  return true;
}
}  // namespace

extern "C" int mgc_db_merge(const char *const *inputs, uint32_t n_inputs, int op, const char *output, int device, int host_threads) {
  if (!inputs || n_inputs == 0 || !output || op < MGC_MERGE_UNION_SUM || op > MGC_MERGE_UNION) {
    set_err(nullptr, "mgc_db_merge: bad arguments");
    return MGC_EINVAL;
  }
  std::vector<mdb_reader *> rd;
  mdb_info first;
  if (!open_merge_inputs(inputs, n_inputs, rd, &first, "mgc_db_merge")) return MGC_EINVAL;
  auto close_all = [&]() { for (mdb_reader *r : rd) if (r) mdb_reader_close(r); };
  const uint32_t k = first.k, w_prefix = first.prefix_size, kw = k > 32 ? 2u : 1u;
  if (device < 0) (void)hipGetDevice(&device);
  mgc_db_stream *d = mgc_db_stream_open(output, k, w_prefix, 0, 0, 0, 1, host_threads, device);
  if (!d) { close_all(); return MGC_EINVAL; }
  int rc = MGC_OK;
  std::string msg;
  auto hip_fail = [&](hipError_t e, const char *what) {
    rc = (e == hipErrorOutOfMemory) ? MGC_ENOMEM : MGC_EHIP;
    msg = std::string("mgc_db_merge: ") + what + ": " + hipGetErrorString(e);
  };
#define MG_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { hip_fail(e__, #expr); goto done; } } while (0)
  {
    hipStream_t st = nullptr;
    std::vector<DBuf> in_k(n_inputs), in_c(n_inputs);
    DBuf acc_k[2], acc_c[2], mem_k[2], mem_c[2], ones, ws;
    std::vector<uint64_t> hn;
    const uint64_t blocks_per_file = 1ull << (w_prefix - MGC_NUM_FILES_BITS);
    // union (value = how many inputs hold the k-mer, :559-561) = union-sum over values of one;
    // symmetric-difference over more than two inputs (in exactly ONE input, :609-612) = the union-sum of the values
    // filtered by "union-sum of ones == 1"; everything else folds from the left with its own two-input step
    const bool by_membership = (op == MGC_MERGE_SYMMETRIC_DIFFERENCE && n_inputs > 2);
    const int fold_op = (op == MGC_MERGE_UNION || by_membership) ? MGC_MERGE_UNION_SUM : op;
    const bool host_decode = decode_on_host();
    MG_TRY(hipSetDevice(device));
    MG_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (uint32_t ff = 0; ff < MGC_NUM_FILES && rc == MGC_OK; ff++) {
      rc = load_slices(rd, ff, kw, in_k, in_c, hn, st, &msg, host_decode);
      if (rc != MGC_OK) { msg = "mgc_db_merge: " + msg; break; }
      // fold: values (and, for the membership forms, ones) through the same sequence of two-input steps
      auto fold = [&](bool use_ones, DBuf (&ak)[2], DBuf (&ac)[2], const void **out_k, const uint32_t **out_c, uint64_t *out_n) -> bool {
        uint64_t most = 0;
        for (uint32_t i = 0; i < n_inputs; i++) most = std::max(most, hn[i]);
        if (use_ones) {
          hipError_t e = ones.ensure(4 * most);
          if (e == hipSuccess) e = mgc::launch_fill_u32(ones.as<uint32_t>(), most, 1u, st);
          if (e != hipSuccess) { hip_fail(e, "ones"); return false; }
        }
        auto cof = [&](uint32_t i) -> const uint32_t * { return use_ones ? ones.as<uint32_t>() : in_c[i].as<uint32_t>(); };
        const void *cur_k = in_k[0].p;
        const uint32_t *cur_c = cof(0);
        uint64_t cur_n = hn[0];
        for (uint32_t i = 1; i < n_inputs; i++) {
          const int t = (int)(i & 1u);
          uint64_t n_new = 0;
          hipError_t e = ws.ensure(mgc::merge_workspace_bytes(cur_n, hn[i]));
          if (e == hipSuccess) e = mgc::launch_merge_count(cur_k, cur_n, in_k[i].p, hn[i], kw, fold_op, ws.p, st, cur_c, cof(i));
          if (e == hipSuccess) e = mgc::merge_read_total(ws.p, &n_new, st);
          if (e == hipSuccess) e = ak[t].ensure(8 * (size_t)kw * n_new);
          if (e == hipSuccess) e = ac[t].ensure(4 * n_new);
          if (e == hipSuccess) e = mgc::launch_merge_emit(cur_k, cur_c, cur_n, in_k[i].p, cof(i), hn[i], kw, fold_op, ws.p, ak[t].p, ac[t].as<uint32_t>(), st);
          if (e == hipSuccess) e = hipStreamSynchronize(st);
          if (e != hipSuccess) { hip_fail(e, "merging a slice"); return false; }
          cur_k = ak[t].p; cur_c = ac[t].as<uint32_t>(); cur_n = n_new;
        }
        *out_k = cur_k; *out_c = cur_c; *out_n = cur_n;
        return true;
      };
      const void *res_k = nullptr; const uint32_t *res_c = nullptr; uint64_t res_n = 0;
      if (!fold(op == MGC_MERGE_UNION, acc_k, acc_c, &res_k, &res_c, &res_n)) break;
      if (by_membership) {
        const void *mk = nullptr; const uint32_t *mc = nullptr; uint64_t mn = 0;
        if (!fold(true, mem_k, mem_c, &mk, &mc, &mn)) break;          // same k-mers as the value fold, values = inputs holding each
        uint64_t n_new = 0;
        DBuf &ok = in_k[0], &oc = in_c[0];                             // the first input's buffers are free by now
        MG_TRY(ws.ensure(mgc::select_workspace_bytes(res_n)));
        MG_TRY(mgc::launch_select_count(res_k, res_c, mc, res_n, kw, 12, 0, ws.p, st));
        MG_TRY(mgc::merge_read_total(ws.p, &n_new, st));
        if (res_k == ok.p) { hip_fail(hipErrorInvalidValue, "symmetric-difference buffers"); break; }
        MG_TRY(ok.ensure(8 * (size_t)kw * n_new));
        MG_TRY(oc.ensure(4 * n_new));
        MG_TRY(mgc::launch_select_emit(res_k, res_c, mc, res_n, kw, 12, 0, ws.p, ok.p, oc.as<uint32_t>(), st));
        MG_TRY(hipStreamSynchronize(st));
        res_k = ok.p; res_c = oc.as<uint32_t>(); res_n = n_new;
      }
      rc = mgc_db_stream_write(d, res_k, res_c, res_n, (uint64_t)ff * blocks_per_file, ((uint64_t)ff + 1) * blocks_per_file);
      if (rc == MGC_OK) rc = mgc_db_stream_sync(d);         // the buffers are reused for the next slice
      if (rc != MGC_OK) msg = std::string("mgc_db_merge: ") + mgc_db_stream_error(d);
    }
  done:
    for (auto &b : in_k) b.release();
    for (auto &b : in_c) b.release();
    for (int t = 0; t < 2; t++) { acc_k[t].release(); acc_c[t].release(); mem_k[t].release(); mem_c[t].release(); }
    ones.release();
    ws.release();
    if (st) (void)hipStreamDestroy(st);
  }
#undef MG_TRY
  close_all();
  const int rc2 = mgc_db_stream_close(d, nullptr);
  if (rc == MGC_OK && rc2 != MGC_OK) { rc = rc2; msg = mgc_db_stream_error(nullptr); }
  if (rc != MGC_OK) set_err(nullptr, "%s", msg.c_str());
  return rc;
}

// The single-input operations over a whole database: less-than ... not-equal-to, increase ... modulo (MGC_VALUE_*).
extern "C" int mgc_db_filter(const char *input, int value_op, uint64_t constant, const char *output, int device, int host_threads) {
  if (!input || !output || value_op < MGC_VALUE_LESS_THAN || value_op > MGC_VALUE_MODULO) { set_err(nullptr, "mgc_db_filter: bad arguments"); return MGC_EINVAL; }
  std::vector<mdb_reader *> rd;
  mdb_info first;
  const char *ins[1] = {input};
  if (!open_merge_inputs(ins, 1, rd, &first, "mgc_db_filter")) return MGC_EINVAL;
  const uint32_t k = first.k, w_prefix = first.prefix_size, kw = k > 32 ? 2u : 1u;
  if (device < 0) (void)hipGetDevice(&device);
  mgc_db_stream *d = mgc_db_stream_open(output, k, w_prefix, 0, 0, 0, 1, host_threads, device);
  if (!d) { mdb_reader_close(rd[0]); return MGC_EINVAL; }
  int rc = MGC_OK;
  std::string msg;
  hipStream_t st = nullptr;
  std::vector<DBuf> in_k(1), in_c(1);
  DBuf out_k, out_c, ws;
  std::vector<uint64_t> hn;
  const uint64_t blocks_per_file = 1ull << (w_prefix - MGC_NUM_FILES_BITS);
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  const bool host_decode = decode_on_host();
  for (uint32_t ff = 0; ff < MGC_NUM_FILES && rc == MGC_OK && e == hipSuccess; ff++) {
    rc = load_slices(rd, ff, kw, in_k, in_c, hn, st, &msg, host_decode);
    if (rc != MGC_OK) { msg = "mgc_db_filter: " + msg; break; }
    uint64_t n_new = 0;
    e = ws.ensure(mgc::select_workspace_bytes(hn[0]));
    if (e == hipSuccess) e = mgc::launch_select_count(in_k[0].p, in_c[0].as<uint32_t>(), nullptr, hn[0], kw, value_op, constant, ws.p, st);
    if (e == hipSuccess) e = mgc::merge_read_total(ws.p, &n_new, st);
    if (e == hipSuccess) e = out_k.ensure(8 * (size_t)kw * n_new);
    if (e == hipSuccess) e = out_c.ensure(4 * n_new);
    if (e == hipSuccess) e = mgc::launch_select_emit(in_k[0].p, in_c[0].as<uint32_t>(), nullptr, hn[0], kw, value_op, constant, ws.p, out_k.p, out_c.as<uint32_t>(), st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) break;
    rc = mgc_db_stream_write(d, out_k.p, out_c.as<uint32_t>(), n_new, (uint64_t)ff * blocks_per_file, ((uint64_t)ff + 1) * blocks_per_file);
    if (rc == MGC_OK) rc = mgc_db_stream_sync(d);
    if (rc != MGC_OK) msg = std::string("mgc_db_filter: ") + mgc_db_stream_error(d);
  }
  if (e != hipSuccess && rc == MGC_OK) { rc = (e == hipErrorOutOfMemory) ? MGC_ENOMEM : MGC_EHIP; msg = std::string("mgc_db_filter: ") + hipGetErrorString(e); }
  in_k[0].release(); in_c[0].release(); out_k.release(); out_c.release(); ws.release();
  if (st) (void)hipStreamDestroy(st);
  mdb_reader_close(rd[0]);
  const int rc2 = mgc_db_stream_close(d, nullptr);
  if (rc == MGC_OK && rc2 != MGC_OK) { rc = rc2; msg = mgc_db_stream_error(nullptr); }
  if (rc != MGC_OK) set_err(nullptr, "%s", msg.c_str());
  return rc;
}
