// mgc_session.hpp -- the session object behind include/meryl_gpu_count.h and the helpers its translation units
// (mgc_api.cpp: input + counting; mgc_stream.cpp: delivery of the result, database streaming) share.  Internal.
#pragma once

#include "../../include/meryl_gpu_count.h"
#include "mgc_device.h"

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace mgc {
// last error of the calling thread (mgc_last_error(NULL)); defined in mgc_api.cpp
std::string &thread_last_error();
// simple mode swaps the configured block geometry for countSimple's (mgc_api.cpp); false + thread error when k is too small
bool effective_geometry(mgc_count_config *c);

inline void set_err(std::string *dst, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (dst) {                                         // a session's error string is written by its worker thread and by the pushing thread
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    *dst = buf;
  }
  thread_last_error() = buf;
}
}  // namespace mgc

#define HIP_TRY(s, expr)                                                                         \
  do {                                                                                           \
    hipError_t e__ = (expr);                                                                     \
    if (e__ != hipSuccess) {                                                                     \
      mgc::set_err((s) ? &(s)->err : nullptr, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,      \
                   hipGetErrorString(e__));                                                      \
      return (e__ == hipErrorOutOfMemory) ? MGC_ENOMEM : MGC_EHIP;                               \
    }                                                                                            \
  } while (0)

struct mgc_session {
  mgc_count_config cfg;
  int              device = -1;
  hipStream_t      stream = nullptr;
  uint64_t         sfx_mask = 0, sfx_test = 0;   // count-suffix= filter (0, 0: none)
  mgc::Switches    sw;                   // the MGC_* switches of the count path, read once by mgc_open (mgc_device.h)
  static constexpr int HUGE_EXTRA = 3;
  hipStream_t      stream_h[HUGE_EXTRA] = {nullptr, nullptr, nullptr};   // more streams for the streaming kernels of oversized sub-buckets (count_device)
  hipStream_t      stream2 = nullptr;    // the streaming hash-count of a file's oversized sub-buckets runs beside its persistent kernel
  hipEvent_t       ev_fork = nullptr, ev_join = nullptr;
  uint64_t        *h_stats = nullptr;    // pinned: per file {largest sub-bucket, oversized sub-buckets, non-empty sub-buckets}
  size_t           h_stats_cap = 0;
  std::string      err;

  // input: what count_device reads (a staging buffer of this session, or the caller's device buffer)
  const uint8_t    *d_bases = nullptr;
  uint64_t          n_bases = 0;
  bool              borrowed = false;

  // result
  bool      counted = false;
  uint64_t  n_instances = 0, n_distinct = 0;
  uint64_t  file_instances[MGC_NUM_FILES];
  // owner side of a sharded count (mgc_count_buckets_into): where the packed result goes when it fits -- the caller's buffers
  void     *ext_out_keys = nullptr; uint32_t *ext_out_counts = nullptr; uint64_t ext_out_cap = 0;
  const uint64_t *ext_fine = nullptr;     // ... and the senders' summed fifteen-bit histogram (the first grouping digit of every bucket)
  void     *d_unique = nullptr;           // uint64[D] (k <= 32) or {lo,hi}[D] (k > 32)
  uint32_t *d_counts = nullptr;
  uint64_t *d_block_start = nullptr;
  uint32_t  key_words = 1;

  // device arena: buffers survive between mgc_count calls (grow-only), so a
  // repeated count does not pay hipMalloc/hipFree of tens of GB every time
  struct Buf { void *p = nullptr; size_t cap = 0; };
  enum { B_PART_WS, B_META, B_X, B_Y, B_SORT_WS, B_RLE_WS, B_UNIQUE, B_COUNTS, B_BLOCKS, B_HPC, B_HPC_WS,
         B_SUBSTART, B_GROUPS, B_GSCAN, B_CNT_TMP, B_LARGE, B_NONEMPTY, B_STAGE0, B_STAGE1, B_TEXT_IN0, B_TEXT_IN1, B_TEXT_WS,
         B_TEXT_STATE, B_RK, B_RC, B_R2K, B_R2C, B_MERGE_WS, B_SORT_HDRS, B_FINE, B_NARROW_WS, B_Y2, B_Y3, B_Y4, B_HWS0, B_HWS1, B_HWS2, B_HWS3, B_NUM };
  Buf buf[B_NUM];
  double tr_alloc = 0;                   // seconds inside hipMalloc / hipFree of the arena (MGC_IO_TRACE)
  uint64_t tr_alloc_bytes = 0;
  hipError_t ensure(int which, size_t bytes) {
    Buf &b = buf[which];
    if (bytes < 256) bytes = 256;
    if (b.cap >= bytes) return hipSuccess;
    const auto t0 = std::chrono::steady_clock::now();
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e == hipSuccess) b.cap = bytes;
    tr_alloc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    tr_alloc_bytes += bytes;
    return e;
  }
  void free_arena() { for (auto &b : buf) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; } }
  // Grows a staging buffer whose first `keep` bytes must survive (device-to-device copy on stream `on`, synchronised).
  // Growth is by 4x up to the batch size, and the old buffer is NOT freed here: hipFree waits for the whole device -- i.e.
  // for the batch the worker thread is counting -- and twenty growth steps of that cost 4.9 of the 5.2 s a 6 Gbp host
  // push took (profiles/r02i).  The old buffers go when the count is done (free_garbage).
  std::vector<void *> garbage;
  hipError_t ensure_preserve(int which, size_t bytes, size_t keep, hipStream_t on) {
    Buf &b = buf[which];
    if (b.cap >= bytes) return hipSuccess;
    struct Tm { double *acc; double t0; static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
                Tm(double *a) : acc(a), t0(now()) {} ~Tm() { *acc += now() - t0; } } tm_(&tr_grow);
    size_t want = b.cap * 4;
    const size_t full = batch_limit ? (size_t)batch_limit + 4 * PIN_CHUNK : 0;       // what a whole batch needs
    if (full && want > full && full >= bytes) want = full;
    if (want < bytes) want = bytes;
    void *np = nullptr;
    hipError_t e = hipMalloc(&np, want);
    if (e != hipSuccess && want > bytes) { want = bytes; e = hipMalloc(&np, want); }
    if (e != hipSuccess) return e;
    if (b.p && keep) e = hipMemcpyAsync(np, b.p, keep, hipMemcpyDeviceToDevice, on);
    if (e == hipSuccess) e = hipStreamSynchronize(on);
    if (b.p) garbage.push_back(b.p);
    b.p = np; b.cap = want;
    return e;
  }
  void free_garbage() { for (void *p : garbage) (void)hipFree(p); garbage.clear(); }

  // ---- input staging -------------------------------------------------------------------------------------------
  // Everything pushed (bases from the host, text parsed on the device) lands in ONE base stream in HBM, stage[fill],
  // '.' after every sequence.  The device-side parse state (B_TEXT_STATE) holds the stream's length: text chunks
  // advance it on the device, plain appends set it; the host's fill_len is exact unless text was parsed since the last
  // read-back (len_inexact) -- then it is an upper bound.  All of it runs on st_in, a stream of its own, so that
  // uploads and parsing overlap the count of the previous batch (which runs on `stream`, from the worker thread).
  static constexpr size_t TEXT_CHUNK = 32u << 20;
  static constexpr size_t PIN_CHUNK  = 32u << 20;
  hipStream_t st_in = nullptr;
  bool        state_ready = false;       // parse state allocated and reset
  int         fill = 0;
  uint64_t    fill_len = 0;
  bool        len_inexact = false;
  bool        input_seen = false;
  // text
  bool        text_open = false, text_cut_in_file = false;
  int         text_format = 0;
  char       *text_pinned[2] = {nullptr, nullptr};
  hipEvent_t  text_ev[2] = {nullptr, nullptr};
  hipStream_t st_up = nullptr;           // text chunks are UPLOADED here while the previous chunk is parsed on st_in
  hipEvent_t  up_ev[2] = {nullptr, nullptr};
  bool        text_ev_used[2] = {false, false};
  uint32_t    text_next = 0;
  static constexpr int TEXT_RING_MAX = 64;
  char       *text_ring[TEXT_RING_MAX] = {nullptr};      // mgc_push_text_file's pinned read-ahead slots (allocated on first use)
  // host-pushed bases: two pinned chunks, the upload of one overlaps the filling of the other
  char       *pin[2] = {nullptr, nullptr};
  size_t      pin_len = 0;
  int         pin_cur = 0;
  hipEvent_t  pin_ev[2] = {nullptr, nullptr};
  bool        pin_used[2] = {false, false};

  // ---- out-of-core batches (the analogue of writeBatch's spill, merylOp-countThreads.C:323-379, and of
  // merylBlockWriter::finish() merging the iterations) ---------------------------------------------------------
  // When the staged bases reach batch_limit, everything up to the last sequence boundary is counted as one batch by
  // the WORKER thread while the caller keeps pushing into the other staging buffer; the batch's (k-mer, count) result
  // is parked as a sorted run -- in HBM while there is room, in pinned host DRAM otherwise -- and the runs are merged once,
  // at the end (mgc_runs.cpp): into one device-resident result when that fits, chunk by chunk into the consumer otherwise.
  uint64_t    batch_limit = 0;            // bases per batch; 0 = derive from free HBM at the first input
  uint64_t    no_cut_below = 0;           // a cut found no sequence boundary: do not scan again before the stream is this long
  bool        have_r = false;             // at least one batch result has been parked
  struct mgc_runs *runs = nullptr;        // the parked batch results (mgc_runs.cpp)
  bool        ooc = false;                // counted, and the result exists only as runs (too large to collapse into HBM)
  uint64_t    result_budget = 0;          // bytes of runs that may stay in HBM; 0 = 60 % of what is free at the first batch's end
  uint64_t    total_bases = 0, total_instances = 0;
  uint64_t    total_file_instances[MGC_NUM_FILES];
  uint32_t    n_batches = 0;
  double      merge_ms = 0;
  std::thread worker;
  double      tr_memcpy = 0, tr_flush = 0, tr_cut = 0, tr_join = 0, tr_grow = 0;   // MGC_IO_TRACE: where the pushing thread's time went
  bool        worker_active = false;      // a batch is being counted (join before touching count state)
  int         worker_rc = MGC_OK;

  // mgc_prepare: the count's largest buffers allocated by a helper thread while the input is still being read
  std::thread prep_thread;
  bool        prep_active = false;

  // profiling
  bool        profiling = false;
  mgc_profile prof;

  void free_result() {            // result views point into the arena
    d_unique = nullptr; d_counts = nullptr; d_block_start = nullptr;
    counted = false;
  }
};

