// mgc_runs.hpp -- the run store behind include/meryl_db.h's mgc_runs_* (mgc_runs.cpp), shared with the session
// (mgc_api.cpp parks batch results in it, mgc_stream.cpp delivers an out-of-core result from it).  Internal.
#pragma once

#include "../../include/meryl_db.h"
#include "mgc_session.hpp"

#include <string>
#include <thread>
#include <vector>

namespace mgc {
// where the merged chunks of a store go: slices [slice_begin, slice_end) of the k-mer space (top slice_bits bits), n ascending
// distinct k-mers + counts in device memory that must stay untouched until wait(*job) returns
struct RunSink {
  virtual int put(const void *d_keys, const uint32_t *d_counts, uint64_t n, uint64_t slice_begin, uint64_t slice_end,
                  uint32_t slice_bits, uint64_t *job) = 0;
  virtual int wait(uint64_t job) = 0;
  virtual const char *error() const = 0;
  virtual ~RunSink() {}
};
}  // namespace mgc

struct mgc_runs {
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
  struct Run {
    uint64_t  n = 0;
    bool      on_host = false;
    void     *keys = nullptr;                               // device memory, or pinned host memory
    uint32_t *counts = nullptr;
    std::vector<uint64_t> slice;                            // [n_slices + 1]: first entry of every slice
  };
  struct Piece { const void *k = nullptr; const uint32_t *c = nullptr; uint64_t n = 0; };

  uint32_t k, kw, w_prefix, slice_bits = 0;
  uint64_t n_slices = 0;
  size_t   esz = 12;
  int      device;
  uint64_t dev_budget, chunk_bytes;
  std::vector<Run> runs;
  hipStream_t st_copy = nullptr, st_up = nullptr, st_mg = nullptr;
  hipEvent_t  ev_up[2] = {nullptr, nullptr}, ev_src = nullptr;
  enum { B_WS, B_IN0K, B_NUM = B_IN0K + 12 };               // + {In, P, Q} x {keys, counts} x two sets
  DBuf buf[B_NUM], d_slices;
  std::string err;
  mgc_runs_profile prof;
  // the NEXT spilled run's pinned memory, allocated by a helper thread while the next batch is counted: pinning costs
  // several times the copy itself (r03h: 106 GB parked in 8.9 s, of which the copies are under 2 s)
  std::thread pre_thread;
  void *pre_k = nullptr, *pre_c = nullptr;
  size_t pre_kb = 0, pre_cb = 0;
  void start_prealloc(size_t kb, size_t cb);
  void drop_prealloc();

  mgc_runs(uint32_t k, uint32_t w_prefix, int device, uint64_t budget, uint64_t chunk);
  ~mgc_runs();
  int  setup();
  void free_run(Run &r);
  void sample_hbm();
  int  add(const void *d_keys, const uint32_t *d_counts, uint64_t n, hipStream_t st);
  int  merge_pair(const Piece &a, const Piece &b, void *dst, uint32_t *dstc, uint64_t *n_out);
  int  deliver(uint64_t s0, uint64_t s1, mgc::RunSink &sink);
  int  collapse(const void **keys, const uint32_t **counts, uint64_t *n);
  int  write(mgc_db_stream *d, uint64_t prefix_begin, uint64_t prefix_end);
  bool all_on_device() const { for (const Run &r : runs) if (r.on_host) return false; return true; }
  uint64_t entries() const { uint64_t t = 0; for (const Run &r : runs) t += r.n; return t; }
};
