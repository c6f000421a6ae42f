// mgc_session.hpp -- the session object behind include/meryl_gpu_count.h and the helpers its translation units
// (mgc_api.cpp: input + counting; mgc_stream.cpp: delivery of the result, database streaming) share.  Internal.
#pragma once

#include "../../include/meryl_gpu_count.h"
#include "mgc_device.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace mgc {
// last error of the calling thread (mgc_last_error(NULL)); defined in mgc_api.cpp
std::string &thread_last_error();

inline void set_err(std::string *dst, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (dst) *dst = buf;
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
  hipStream_t      stream2 = nullptr;    // the streaming hash-count of a file's oversized sub-buckets runs beside its persistent kernel
  hipEvent_t       ev_fork = nullptr, ev_join = nullptr;
  std::string      err;

  // input
  std::vector<char> host_bases;          // mgc_push_bases accumulates here (pinned staging is a later round)
  uint8_t          *d_bases_own = nullptr;
  const uint8_t    *d_bases = nullptr;
  uint64_t          n_bases = 0;
  bool              borrowed = false;

  // result
  bool      counted = false;
  uint64_t  n_instances = 0, n_distinct = 0;
  uint64_t  file_instances[MGC_NUM_FILES];
  void     *d_unique = nullptr;           // uint64[D] (k <= 32) or {lo,hi}[D] (k > 32)
  uint32_t *d_counts = nullptr;
  uint64_t *d_block_start = nullptr;
  uint32_t  key_words = 1;

  // device arena: buffers survive between mgc_count calls (grow-only), so a
  // repeated count does not pay hipMalloc/hipFree of tens of GB every time
  struct Buf { void *p = nullptr; size_t cap = 0; };
  enum { B_PART_WS, B_META, B_X, B_Y, B_SORT_WS, B_RLE_WS, B_UNIQUE, B_COUNTS, B_BLOCKS, B_HPC, B_HPC_WS, B_BASES,
         B_SUBSTART, B_GROUPS, B_GSCAN, B_CNT_TMP, B_LARGE, B_NONEMPTY, B_TEXT_OUT, B_TEXT_IN0, B_TEXT_IN1, B_TEXT_WS, B_TEXT_STATE, B_NUM };
  Buf buf[B_NUM];
  hipError_t ensure(int which, size_t bytes) {
    Buf &b = buf[which];
    if (bytes < 256) bytes = 256;
    if (b.cap >= bytes) return hipSuccess;
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e == hipSuccess) b.cap = bytes;
    return e;
  }
  void free_arena() { for (auto &b : buf) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; } }
  // grows a buffer whose first `keep` bytes must survive (device-to-device copy on the session stream)
  hipError_t ensure_preserve(int which, size_t bytes, size_t keep) {
    Buf &b = buf[which];
    if (b.cap >= bytes) return hipSuccess;
    size_t want = b.cap + b.cap / 2;
    if (want < bytes) want = bytes;
    void *np = nullptr;
    hipError_t e = hipMalloc(&np, want);
    if (e != hipSuccess) return e;
    if (b.p && keep) e = hipMemcpyAsync(np, b.p, keep, hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (b.p) (void)hipFree(b.p);
    b.p = np; b.cap = want;
    return e;
  }

  // device-side text parsing (mgc_push_text): pinned staging, two chunks in flight
  static constexpr size_t TEXT_CHUNK = 32u << 20;
  bool        text_mode = false, text_open = false;
  int         text_format = 0;
  uint64_t    text_bound = 0;            // upper bound of the parsed length so far (the device knows the exact one)
  char       *text_pinned[2] = {nullptr, nullptr};
  hipEvent_t  text_ev[2] = {nullptr, nullptr};
  bool        text_ev_used[2] = {false, false};
  uint32_t    text_next = 0;

  // out-of-core batches (the analogue of writeBatch's spill, merylOp-countThreads.C:323-379): when the
  // pushed bases exceed what one pass can hold in HBM, everything up to the last sequence
  // boundary is counted and its (k-mer, count) result parked in host memory; mgc_count merges
  // the parked results per file (summing counts) like merylBlockWriter::finish() merges iterations.
  // parked batch results live in PINNED host memory and are filled by asynchronous copies on the session stream
  // (pageable copies run at a fifth of the PCIe rate); keys stay interleaved {lo[,hi]} exactly as on the device
  template <typename T> struct Pinned {
    T *p = nullptr; size_t n = 0;
    Pinned() = default;
    Pinned(const Pinned &) = delete;
    Pinned &operator=(const Pinned &) = delete;
    Pinned(Pinned &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    ~Pinned() { if (p) (void)hipHostFree(p); }
    hipError_t alloc(size_t count) {
      n = count;
      return hipHostMalloc(reinterpret_cast<void **>(&p), (count ? count : 1) * sizeof(T), hipHostMallocDefault);
    }
  };
  struct BatchResult {
    Pinned<uint64_t> keys, bstart;        // keys: key_words x n_distinct
    Pinned<uint32_t> counts;
    uint64_t n_distinct = 0;
    uint32_t kw = 1;
    uint64_t lo(uint64_t i) const { return keys.p[kw * i]; }
    uint64_t hi(uint64_t i) const { return kw == 2 ? keys.p[2 * i + 1] : 0ull; }
  };
  std::vector<BatchResult> batches;
  uint64_t  batch_limit = 0;              // bases per batch; 0 = derive from free HBM at the first push
  bool      merged = false;               // final result lives in m_* (host) instead of d_* (device)
  std::vector<uint64_t> m_lo, m_hi, m_bstart;
  std::vector<uint32_t> m_counts;
  uint64_t  total_bases = 0, total_instances = 0;
  uint64_t  total_file_instances[MGC_NUM_FILES];

  // profiling
  bool        profiling = false;
  mgc_profile prof;

  void free_result() {            // result views point into the arena
    d_unique = nullptr; d_counts = nullptr; d_block_start = nullptr;
    counted = false;
  }
};

