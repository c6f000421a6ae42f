// mgc_lookup.hip -- the exact k-mer lookup table (include/meryl_lookup.h), gfx950.
//
// Replaces merylExactLookup [meryl-utility, not in tree] as meryl-lookup uses it: load(db, ..., minValue, maxValue)
// (src/meryl-lookup/meryl-lookup.C:36-100), value(kmer) / nKmers() per k-mer of a sequence
// (src/meryl-lookup/existence.C:63-82).  The table is the sorted (k-mer, value) stream itself plus a direct index over
// the top bits: index read -> binary search among the few k-mers sharing those bits.  Integer work bound by HBM/L2
// latency, not bandwidth: queries are batched so that thousands of searches are in flight per CU.
#include "../../include/meryl_db.h"
#include "../../include/meryl_lookup.h"
#include "mgc_common.hpp"

#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace mgc {

template <typename K> struct LkOps;
template <> struct LkOps<u64> {
  typedef u64 W;                                   // arithmetic type of a rolling k-mer
  static __device__ __forceinline__ u64 bucket(u64 k, u32 shift) { return shift >= 64 ? 0ull : (k >> shift); }
  static __device__ __forceinline__ u64 key(W w) { return w; }
};
template <> struct LkOps<K128> {
  typedef u128 W;
  static __device__ __forceinline__ u64 bucket(K128 k, u32 shift) { return shift >= 128 ? 0ull : (u64)(KeyOps<K128>::v(k) >> shift); }
  static __device__ __forceinline__ K128 key(W w) { return KeyOps<K128>::mk(w); }
};

template <typename K>
__device__ __forceinline__ u32 lk_find(const K *__restrict__ keys, const u32 *__restrict__ vals, const u64 *__restrict__ index,
                                       u32 shift, K q, u64 n_index = ~0ull /* 2^index_bits: the index holds one entry more */) {
  const u64 p = LkOps<K>::bucket(q, shift);
  if (p >= n_index) return 0u;                        // a query with bits above 2k (queries are looked up as given): not a k-mer, not stored
  u64 lo = index[p];
  const u64 end = index[p + 1];
  u64 hi = end;
  while (lo < hi) {
    const u64 mid = lo + ((hi - lo) >> 1);
    if (KeyOps<K>::lt(keys[mid], q)) lo = mid + 1; else hi = mid;
  }
  return (lo < end && !KeyOps<K>::ne(keys[lo], q)) ? vals[lo] : 0u;
}

template <typename K>
__global__ __launch_bounds__(256)
void lookup_values_kernel(const K *__restrict__ keys, const u32 *__restrict__ vals, const u64 *__restrict__ index, u32 shift,
                          const K *__restrict__ q, u64 n, u32 *__restrict__ out, u64 n_index) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  // queries come from the caller as they are: one with bits above 2k must not index past the table
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = lk_find<K>(keys, vals, index, shift, q[i], n_index);
}

// 2-bit code of a base (A0 C1 T2 G3, reference.rst:525), -1 for anything else
__device__ __forceinline__ int lk_code(u32 c) {
  const u32 l = c | 0x20u;
  const bool ok = (l == 'a') | (l == 'c') | (l == 'g') | (l == 't');
  return ok ? (int)((c >> 1) & 3u) : -1;
}

constexpr int LK_RUN = 16;                          // window starts per thread

// Every window start of [i0, i0 + LK_RUN): the rolling forward / reverse-complement pair of the reference's kmerIterator
// (call sites src/meryl-lookup/existence.C:69-77), restarted at i0 -- windows that start at or after i0 depend on no
// earlier base.  visit(start, value_or_0, is_kmer)
template <typename K, typename F>
__device__ __forceinline__ void lk_walk(const K *__restrict__ keys, const u32 *__restrict__ vals, const u64 *__restrict__ index,
                                        u32 shift, const uint8_t *__restrict__ bases, u64 n_bases, u32 k, u64 i0, F visit) {
  typedef typename LkOps<K>::W W;
  const W one = 1;
  const W mask = (2 * k >= sizeof(W) * 8) ? ~(W)0 : ((one << (2 * k)) - 1);
  W f = 0, r = 0;
  u32 load = 0;
  const u64 jend = (i0 + LK_RUN + k - 1 < n_bases) ? (i0 + LK_RUN + k - 1) : n_bases;
  for (u64 j = i0; j < jend; j++) {
    const int code = lk_code(bases[j]);
    if (code < 0) { load = 0; f = 0; r = 0; continue; }
    f = ((f << 2) | (W)code) & mask;
    r = (r >> 2) | ((W)(code ^ 2) << (2 * k - 2));
    if (load < k) load++;
    if (load < k) continue;
    const u64 s = j + 1 - k;                        // >= i0 because load restarted at i0
    if (s >= i0 + LK_RUN) break;
    u32 v = lk_find<K>(keys, vals, index, shift, LkOps<K>::key(f));
    if (v == 0 && f != r) v = lk_find<K>(keys, vals, index, shift, LkOps<K>::key(r));   // value(fmer) > 0 || value(rmer) > 0
    visit(s, v);
  }
}

template <typename K>
__global__ __launch_bounds__(256)
void lookup_stream_kernel(const K *__restrict__ keys, const u32 *__restrict__ vals, const u64 *__restrict__ index, u32 shift,
                          const uint8_t *__restrict__ bases, u64 n_bases, u32 k, u32 *__restrict__ out) {
  const u64 i0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * LK_RUN;
  if (i0 >= n_bases) return;
  for (int q = 0; q < LK_RUN; q++) if (i0 + q < n_bases) out[i0 + q] = 0;
  lk_walk<K>(keys, vals, index, shift, bases, n_bases, k, i0, [&](u64 s, u32 v) { out[s] = v; });
}

template <typename K>
__global__ __launch_bounds__(256)
void lookup_existence_kernel(const K *__restrict__ keys, const u32 *__restrict__ vals, const u64 *__restrict__ index, u32 shift,
                             const uint8_t *__restrict__ bases, u64 n_bases, u32 k, const u64 *__restrict__ seq_start, u64 n_seq,
                             u64 *__restrict__ total, u64 *__restrict__ found) {
  const u64 i0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * LK_RUN;
  if (i0 >= n_bases || n_seq == 0) return;
  u64 cur = ~0ull, tot = 0, fnd = 0;
  auto flush = [&]() {
    if (cur != ~0ull && tot) {
      atomicAdd(reinterpret_cast<unsigned long long *>(total + cur), (unsigned long long)tot);
      if (fnd) atomicAdd(reinterpret_cast<unsigned long long *>(found + cur), (unsigned long long)fnd);
    }
    tot = 0; fnd = 0;
  };
  lk_walk<K>(keys, vals, index, shift, bases, n_bases, k, i0, [&](u64 s, u32 v) {
    if (cur == ~0ull || s >= seq_start[cur + 1]) {   // the window's sequence: last start <= s
      flush();
      u64 lo = 0, hi = n_seq;
      while (hi - lo > 1) { const u64 mid = lo + ((hi - lo) >> 1); if (seq_start[mid] <= s) lo = mid; else hi = mid; }
      cur = lo;
    }
    if (s + k <= seq_start[cur + 1]) { tot++; fnd += (v != 0); }     // a window must lie inside its sequence
  });
  flush();
}

// ---- value filter (minValue <= v <= maxValue), order-preserving: count per tile, scan, emit ----------------------
constexpr int FL_TILE = 256 * 8;
__global__ __launch_bounds__(256)
void filter_count_kernel(const u32 *__restrict__ vals, u64 n, u64 vmin, u64 vmax, u64 *__restrict__ tile_cnt) {
  __shared__ u32 s_tmp[256 / 64 + 1];
  const u64 base = (u64)blockIdx.x * FL_TILE + (u64)threadIdx.x * 8;
  u32 c = 0;
  for (int q = 0; q < 8; q++) { const u64 i = base + q; if (i < n) { const u64 v = vals[i]; c += (v >= vmin && v <= vmax); } }
  u32 tot;
  (void)block_excl_scan<256, u32>(c, s_tmp, &tot);
  if (threadIdx.x == 0) tile_cnt[blockIdx.x] = tot;
}
template <typename K>
__global__ __launch_bounds__(256)
void filter_emit_kernel(const K *__restrict__ keys, const u32 *__restrict__ vals, u64 n, u64 vmin, u64 vmax,
                        const u64 *__restrict__ tile_base, K *__restrict__ out_k, u32 *__restrict__ out_v) {
  __shared__ u32 s_tmp[256 / 64 + 1];
  const u64 base = (u64)blockIdx.x * FL_TILE + (u64)threadIdx.x * 8;
  u32 c = 0;
  for (int q = 0; q < 8; q++) { const u64 i = base + q; if (i < n) { const u64 v = vals[i]; c += (v >= vmin && v <= vmax); } }
  u32 tot;
  u64 o = tile_base[blockIdx.x] + block_excl_scan<256, u32>(c, s_tmp, &tot);
  for (int q = 0; q < 8; q++) {
    const u64 i = base + q;
    if (i < n) { const u64 v = vals[i]; if (v >= vmin && v <= vmax) { out_k[o] = keys[i]; out_v[o] = (u32)v; o++; } }
  }
}

}  // namespace mgc

// ================================================================================================
//  C ABI
// ================================================================================================
namespace {
thread_local std::string g_lk_error;
void lk_err(const std::string &m) { g_lk_error = m; }
}  // namespace

struct mgc_lookup {
  int      device = 0;
  uint32_t k = 0, kw = 1, index_bits = 0, shift = 0;
  uint64_t n = 0, n_db = 0;
  void     *d_keys = nullptr;
  uint32_t *d_vals = nullptr;
  uint64_t *d_index = nullptr;
};

extern "C" const char *mgc_lookup_error(void) { return g_lk_error.c_str(); }

extern "C" void mgc_lookup_free(mgc_lookup *t) {
  if (!t) return;
  (void)hipSetDevice(t->device);
  if (t->d_keys) (void)hipFree(t->d_keys);
  if (t->d_vals) (void)hipFree(t->d_vals);
  if (t->d_index) (void)hipFree(t->d_index);
  delete t;
}

#define LK_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { lk_err(std::string(#expr) + ": " + hipGetErrorString(e__)); mgc_lookup_free(t); return nullptr; } } while (0)

// keys/values already on the device (filtered, owned by the table): build the index
static uint32_t lookup_index_bits(uint32_t k, uint64_t n) {
  uint32_t bits = 0;
  while (bits < 2 * k && bits < 28 && (n >> (bits + 2)) > 0) bits++;             // ~4 k-mers per index entry
  return bits;
}

extern "C" int mgc_lookup_estimate(const char *db_path, uint64_t min_value, uint64_t max_value, mgc_lookup_info *info) {
  if (!db_path || !info) { lk_err("mgc_lookup_estimate: bad arguments"); return MGC_EINVAL; }
  mdb_reader *r = mdb_reader_open(db_path);
  if (!r) { lk_err(std::string("mgc_lookup_estimate: ") + mdb_last_error()); return MGC_EINVAL; }
  mdb_info di;
  mdb_reader_info(r, &di);
  std::vector<uint64_t> values(di.hist_len ? di.hist_len : 1), occ(di.hist_len ? di.hist_len : 1);
  if (di.hist_len && mdb_reader_histogram(r, values.data(), occ.data()) != 0) {
    lk_err(std::string("mgc_lookup_estimate: ") + mdb_last_error());
    mdb_reader_close(r);
    return MGC_EINVAL;
  }
  mdb_reader_close(r);
  uint64_t n_db = 0, n = 0;
  for (uint64_t i = 0; i < di.hist_len; i++) {
    n_db += occ[i];
    if (values[i] >= min_value && values[i] <= max_value) n += occ[i];
  }
  memset(info, 0, sizeof(*info));
  info->k = di.k; info->key_words = di.k > 32 ? 2u : 1u;
  info->index_bits = lookup_index_bits(di.k, n);
  info->n_kmers = n; info->n_kmers_in_db = n_db;
  info->device_bytes = (sizeof(uint64_t) * info->key_words + sizeof(uint32_t)) * n + sizeof(uint64_t) * ((1ull << info->index_bits) + 1);
  return MGC_OK;
}

static mgc_lookup *lookup_finish(mgc_lookup *t) {
  const uint32_t bits = lookup_index_bits(t->k, t->n);
  t->index_bits = bits;
  t->shift = 2 * t->k - bits;
  const uint64_t entries = (1ull << bits) + 1;
  LK_TRY(hipMalloc(reinterpret_cast<void **>(&t->d_index), sizeof(uint64_t) * entries));
  if (bits == 0) {                                          // a handful of k-mers: one bucket
    const uint64_t two[2] = {0, t->n};
    LK_TRY(hipMemcpy(t->d_index, two, sizeof(two), hipMemcpyHostToDevice));
  } else {
    LK_TRY(mgc::launch_block_offsets(t->d_keys, t->n, t->kw, t->shift, 1ull << bits, t->d_index, nullptr));
    LK_TRY(hipStreamSynchronize(nullptr));
  }
  return t;
}

extern "C" mgc_lookup *mgc_lookup_from_device(const void *d_keys, const uint32_t *d_values, uint64_t n, uint32_t k,
                                              uint64_t min_value, uint64_t max_value, int device) {
  if (k == 0 || k > 64 || (n && (!d_keys || !d_values))) { lk_err("mgc_lookup_from_device: bad arguments"); return nullptr; }
  mgc_lookup *t = new mgc_lookup();
  if (device < 0) (void)hipGetDevice(&device);
  t->device = device; t->k = k; t->kw = k > 32 ? 2u : 1u; t->n_db = n;
  LK_TRY(hipSetDevice(device));
  const size_t kbytes = sizeof(uint64_t) * t->kw;
  if (min_value <= 1 && max_value >= 0xFFFFFFFFull) {       // nothing to filter (stored values are >= 1)
    t->n = n;
    LK_TRY(hipMalloc(&t->d_keys, kbytes * (n ? n : 1)));
    LK_TRY(hipMalloc(reinterpret_cast<void **>(&t->d_vals), sizeof(uint32_t) * (n ? n : 1)));
    if (n) {
      LK_TRY(hipMemcpy(t->d_keys, d_keys, kbytes * n, hipMemcpyDeviceToDevice));
      LK_TRY(hipMemcpy(t->d_vals, d_values, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice));
    }
    return lookup_finish(t);
  }
  const uint64_t tiles = (n + mgc::FL_TILE - 1) / mgc::FL_TILE;
  uint64_t *d_tiles = nullptr, kept = 0;
  if (tiles) {
    const size_t elems = tiles + 1 + mgc::scan_scratch_elems(tiles + 1) + 8;
    LK_TRY(hipMalloc(reinterpret_cast<void **>(&d_tiles), sizeof(uint64_t) * elems));
    hipLaunchKernelGGL(mgc::filter_count_kernel, dim3((uint32_t)tiles), dim3(256), 0, nullptr, d_values, (mgc::u64)n, (mgc::u64)min_value,
                       (mgc::u64)max_value, reinterpret_cast<mgc::u64 *>(d_tiles));
    hipError_t e = mgc::scan_u64_exclusive(reinterpret_cast<mgc::u64 *>(d_tiles), tiles, reinterpret_cast<mgc::u64 *>(d_tiles) + tiles + 1 + 8,
                                           reinterpret_cast<mgc::u64 *>(d_tiles) + tiles + 1, nullptr);
    if (e == hipSuccess) e = hipMemcpy(&kept, d_tiles + tiles + 1, sizeof(uint64_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { (void)hipFree(d_tiles); lk_err(std::string("value filter: ") + hipGetErrorString(e)); mgc_lookup_free(t); return nullptr; }
  }
  t->n = kept;
  hipError_t e = hipMalloc(&t->d_keys, kbytes * (kept ? kept : 1));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&t->d_vals), sizeof(uint32_t) * (kept ? kept : 1));
  if (e == hipSuccess && tiles) {
    if (t->kw == 2)
      hipLaunchKernelGGL((mgc::filter_emit_kernel<mgc::K128>), dim3((uint32_t)tiles), dim3(256), 0, nullptr, reinterpret_cast<const mgc::K128 *>(d_keys),
                         d_values, (mgc::u64)n, (mgc::u64)min_value, (mgc::u64)max_value, reinterpret_cast<const mgc::u64 *>(d_tiles),
                         reinterpret_cast<mgc::K128 *>(t->d_keys), t->d_vals);
    else
      hipLaunchKernelGGL((mgc::filter_emit_kernel<mgc::u64>), dim3((uint32_t)tiles), dim3(256), 0, nullptr, reinterpret_cast<const mgc::u64 *>(d_keys),
                         d_values, (mgc::u64)n, (mgc::u64)min_value, (mgc::u64)max_value, reinterpret_cast<const mgc::u64 *>(d_tiles),
                         reinterpret_cast<mgc::u64 *>(t->d_keys), t->d_vals);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  }
  if (d_tiles) (void)hipFree(d_tiles);
  if (e != hipSuccess) { lk_err(std::string("value filter: ") + hipGetErrorString(e)); mgc_lookup_free(t); return nullptr; }
  return lookup_finish(t);
}

extern "C" mgc_lookup *mgc_lookup_load(const char *db_path, uint64_t min_value, uint64_t max_value, int device, int host_threads) {
  if (!db_path) { lk_err("mgc_lookup_load: no database"); return nullptr; }
  mdb_reader *r = mdb_reader_open(db_path);
  if (!r) { lk_err(std::string("mgc_lookup_load: ") + mdb_last_error()); return nullptr; }
  mdb_info info;
  mdb_reader_info(r, &info);
  mdb_reader_close(r);
  const uint32_t kw = info.k > 32 ? 2u : 1u;
  if (host_threads <= 0) host_threads = (int)std::thread::hardware_concurrency();
  host_threads = std::max(1, std::min(host_threads, MGC_NUM_FILES));
  // Device decode (mgc_decode.hip): host threads only READ the 64 data files and check their framing; the bytes go to HBM as
  // they are and every block is decoded there (one thread per block), straight into the table's arrays; the value filter is
  // the device compaction mgc_lookup_from_device uses.  MGC_DECODE_HOST=1, or a file only the host decoder follows: below.
  const bool host_decode = getenv("MGC_DECODE_HOST") && getenv("MGC_DECODE_HOST")[0] == '1';      // once per table load (tests switch it between loads)
  if (!host_decode) {
    if (device < 0) (void)hipGetDevice(&device);
    if (hipSetDevice(device) != hipSuccess) { lk_err("mgc_lookup_load: hipSetDevice failed"); return nullptr; }
    struct Raw { unsigned char *bytes = nullptr; uint64_t size = 0, nb = 0, n = 0; mdb_raw_block *blocks = nullptr; int rc = MGC_OK; };
    std::vector<Raw> raw(MGC_NUM_FILES);
    std::atomic<uint32_t> nextf(0);
    std::mutex mu2;
    std::string msg2;
    auto reader = [&]() {
      mdb_reader *rr = mdb_reader_open(db_path);
      if (!rr) { std::lock_guard<std::mutex> g(mu2); msg2 = mdb_last_error(); for (auto &x : raw) if (x.rc == MGC_OK && !x.bytes) x.rc = MGC_EINVAL; return; }
      for (;;) {
        const uint32_t ff = nextf.fetch_add(1);
        if (ff >= MGC_NUM_FILES) break;
        Raw &x = raw[ff];
        x.rc = mdb_reader_raw_file(rr, ff, &x.bytes, &x.size, &x.blocks, &x.nb, &x.n);
        if (x.rc != MGC_OK && x.rc != MGC_EUNSUPPORTED) { std::lock_guard<std::mutex> g(mu2); msg2 = mdb_last_error(); }
      }
      mdb_reader_close(rr);
    };
    {
      std::vector<std::thread> pool;
      for (int i = 1; i < host_threads; i++) pool.emplace_back(reader);
      reader();
      for (auto &th : pool) th.join();
    }
    bool all_ok = true, hard_fail = false;
    uint64_t total = 0, max_size = 0, max_nb = 0;
    for (const Raw &x : raw) { all_ok = all_ok && x.rc == MGC_OK; hard_fail = hard_fail || (x.rc != MGC_OK && x.rc != MGC_EUNSUPPORTED); total += x.n; max_size = std::max(max_size, x.size); max_nb = std::max(max_nb, x.nb); }
    auto free_raw = [&]() { for (Raw &x : raw) { mdb_free(x.bytes); mdb_free(x.blocks); x.bytes = nullptr; x.blocks = nullptr; } };
    if (hard_fail) { free_raw(); lk_err("mgc_lookup_load: " + msg2); return nullptr; }
    if (all_ok) {
      const size_t kbytes = sizeof(uint64_t) * kw;
      void *dk = nullptr, *dfile = nullptr, *dblocks = nullptr;
      uint32_t *dv = nullptr, *derr = nullptr, h_err = 0;
      hipError_t e = hipMalloc(&dk, kbytes * (total ? total : 1));
      if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dv), sizeof(uint32_t) * (total ? total : 1));
      if (e == hipSuccess) e = hipMalloc(&dfile, max_size + 16);
      if (e == hipSuccess) e = hipMalloc(&dblocks, sizeof(mdb_raw_block) * (max_nb ? max_nb : 1));
      if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&derr), 256);
      if (e == hipSuccess) e = hipMemset(derr, 0, 4);
      uint64_t o = 0;
      for (uint32_t ff = 0; ff < MGC_NUM_FILES && e == hipSuccess; ff++) {
        const Raw &x = raw[ff];
        if (x.n) {
          e = hipMemcpy(dfile, x.bytes, x.size + 16, hipMemcpyHostToDevice);
          if (e == hipSuccess) e = hipMemcpy(dblocks, x.blocks, sizeof(mdb_raw_block) * x.nb, hipMemcpyHostToDevice);
          if (e == hipSuccess) e = mgc::launch_decode_blocks(dfile, dblocks, x.nb, info.suffix_size, info.label_size, kw,
                                                             reinterpret_cast<unsigned char *>(dk) + kbytes * o, dv + o, derr, nullptr);
          if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        }
        o += x.n;
      }
      if (e == hipSuccess) e = hipMemcpy(&h_err, derr, 4, hipMemcpyDeviceToHost);
      free_raw();
      if (dfile) (void)hipFree(dfile);
      if (dblocks) (void)hipFree(dblocks);
      if (derr) (void)hipFree(derr);
      if (e != hipSuccess || h_err) {
        if (dk) (void)hipFree(dk);
        if (dv) (void)hipFree(dv);
        lk_err(e != hipSuccess ? std::string("mgc_lookup_load: ") + hipGetErrorString(e) : std::string("mgc_lookup_load: corrupt block in '") + db_path + "' (device decoder)");
        return nullptr;
      }
      mgc_lookup *t = mgc_lookup_from_device(dk, dv, total, info.k, min_value, max_value, device);
      (void)hipFree(dk); (void)hipFree(dv);
      return t;
    }
    free_raw();                                            // some file needs the host decoder: all of them take it
  }
  // the 64 files decoded by host threads (one reader each: the reader is not shared), value filter applied on the way
  struct Part { std::vector<uint64_t> keys; std::vector<uint32_t> vals; uint64_t n_db = 0; };
  std::vector<Part> parts(MGC_NUM_FILES);
  std::atomic<uint32_t> next(0);
  std::atomic<bool> failed(false);
  std::string fail_msg;
  std::mutex mu;
  auto worker = [&]() {
    mdb_reader *rr = mdb_reader_open(db_path);
    if (!rr) { std::lock_guard<std::mutex> g(mu); failed = true; fail_msg = mdb_last_error(); return; }
    for (;;) {
      const uint32_t ff = next.fetch_add(1);
      if (ff >= MGC_NUM_FILES || failed.load()) break;
      uint64_t *lo = nullptr, *hi = nullptr, n = 0;
      uint32_t *cn = nullptr;
      if (mdb_reader_read_file(rr, ff, &lo, &hi, &cn, &n) != MGC_OK) { std::lock_guard<std::mutex> g(mu); failed = true; fail_msg = mdb_last_error(); break; }
      Part &p = parts[ff];
      p.n_db = n;
      p.keys.reserve((size_t)kw * n); p.vals.reserve(n);
      for (uint64_t i = 0; i < n; i++) {
        if (cn[i] < min_value || cn[i] > max_value) continue;
        p.keys.push_back(lo[i]);
        if (kw == 2) p.keys.push_back(hi[i]);
        p.vals.push_back(cn[i]);
      }
      mdb_free(lo); mdb_free(hi); mdb_free(cn);
    }
    mdb_reader_close(rr);
  };
  std::vector<std::thread> pool;
  for (int i = 1; i < host_threads; i++) pool.emplace_back(worker);
  worker();
  for (auto &th : pool) th.join();
  if (failed.load()) { lk_err("mgc_lookup_load: " + fail_msg); return nullptr; }
  mgc_lookup *t = new mgc_lookup();
  if (device < 0) (void)hipGetDevice(&device);
  t->device = device; t->k = info.k; t->kw = kw;
  for (const Part &p : parts) { t->n += p.vals.size(); t->n_db += p.n_db; }
  LK_TRY(hipSetDevice(device));
  const size_t kbytes = sizeof(uint64_t) * kw;
  LK_TRY(hipMalloc(&t->d_keys, kbytes * (t->n ? t->n : 1)));
  LK_TRY(hipMalloc(reinterpret_cast<void **>(&t->d_vals), sizeof(uint32_t) * (t->n ? t->n : 1)));
  uint64_t o = 0;
  for (const Part &p : parts) {
    const uint64_t m = p.vals.size();
    if (!m) continue;
    LK_TRY(hipMemcpy(reinterpret_cast<unsigned char *>(t->d_keys) + kbytes * o, p.keys.data(), kbytes * m, hipMemcpyHostToDevice));
    LK_TRY(hipMemcpy(t->d_vals + o, p.vals.data(), sizeof(uint32_t) * m, hipMemcpyHostToDevice));
    o += m;
  }
  return lookup_finish(t);
}

extern "C" int mgc_lookup_get_info(const mgc_lookup *t, mgc_lookup_info *info) {
  if (!t || !info) return MGC_EINVAL;
  info->k = t->k; info->key_words = t->kw; info->index_bits = t->index_bits; info->reserved = 0;
  info->n_kmers = t->n; info->n_kmers_in_db = t->n_db;
  info->device_bytes = (sizeof(uint64_t) * t->kw + sizeof(uint32_t)) * t->n + sizeof(uint64_t) * ((1ull << t->index_bits) + 1);
  return MGC_OK;
}

namespace {
int lk_rc(hipError_t e, const char *what) {
  if (e == hipSuccess) return MGC_OK;
  lk_err(std::string(what) + ": " + hipGetErrorString(e));
  return MGC_EHIP;
}
uint32_t lk_grid(uint64_t threads) { const uint64_t g = (threads + 255) / 256; return (uint32_t)(g ? g : 1); }
}  // namespace

extern "C" int mgc_lookup_values(const mgc_lookup *t, const void *d_kmers, uint64_t n, uint32_t *d_out, void *stream) {
  if (!t || (n && (!d_kmers || !d_out))) return MGC_EINVAL;
  if (n == 0) return MGC_OK;
  uint64_t g = (n + 255) / 256;
  if (g > 65536) g = 65536;
  if (t->kw == 2)
    hipLaunchKernelGGL((mgc::lookup_values_kernel<mgc::K128>), dim3((uint32_t)g), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const mgc::K128 *>(t->d_keys),
                       t->d_vals, reinterpret_cast<const mgc::u64 *>(t->d_index), t->shift, reinterpret_cast<const mgc::K128 *>(d_kmers), (mgc::u64)n, d_out, (mgc::u64)1 << t->index_bits);
  else
    hipLaunchKernelGGL((mgc::lookup_values_kernel<mgc::u64>), dim3((uint32_t)g), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const mgc::u64 *>(t->d_keys),
                       t->d_vals, reinterpret_cast<const mgc::u64 *>(t->d_index), t->shift, reinterpret_cast<const mgc::u64 *>(d_kmers), (mgc::u64)n, d_out, (mgc::u64)1 << t->index_bits);
  return lk_rc(hipGetLastError(), "lookup_values");
}

extern "C" int mgc_lookup_stream(const mgc_lookup *t, const uint8_t *d_bases, uint64_t n_bases, uint32_t *d_out, void *stream) {
  if (!t || (n_bases && (!d_bases || !d_out))) return MGC_EINVAL;
  if (n_bases == 0) return MGC_OK;
  const uint32_t g = lk_grid((n_bases + mgc::LK_RUN - 1) / mgc::LK_RUN);
  if (t->kw == 2)
    hipLaunchKernelGGL((mgc::lookup_stream_kernel<mgc::K128>), dim3(g), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const mgc::K128 *>(t->d_keys),
                       t->d_vals, reinterpret_cast<const mgc::u64 *>(t->d_index), t->shift, d_bases, (mgc::u64)n_bases, t->k, d_out);
  else
    hipLaunchKernelGGL((mgc::lookup_stream_kernel<mgc::u64>), dim3(g), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const mgc::u64 *>(t->d_keys),
                       t->d_vals, reinterpret_cast<const mgc::u64 *>(t->d_index), t->shift, d_bases, (mgc::u64)n_bases, t->k, d_out);
  return lk_rc(hipGetLastError(), "lookup_stream");
}

extern "C" int mgc_lookup_existence(const mgc_lookup *t, const uint8_t *d_bases, uint64_t n_bases, const uint64_t *d_seq_start,
                                    uint64_t n_seq, uint64_t *d_total, uint64_t *d_found, void *stream) {
  if (!t || (n_seq && (!d_seq_start || !d_total || !d_found)) || (n_bases && !d_bases)) return MGC_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (n_seq) {
    hipError_t e = hipMemsetAsync(d_total, 0, sizeof(uint64_t) * n_seq, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_found, 0, sizeof(uint64_t) * n_seq, st);
    if (e != hipSuccess) return lk_rc(e, "lookup_existence");
  }
  if (n_bases == 0 || n_seq == 0) return MGC_OK;
  const uint32_t g = lk_grid((n_bases + mgc::LK_RUN - 1) / mgc::LK_RUN);
  if (t->kw == 2)
    hipLaunchKernelGGL((mgc::lookup_existence_kernel<mgc::K128>), dim3(g), dim3(256), 0, st, reinterpret_cast<const mgc::K128 *>(t->d_keys), t->d_vals,
                       reinterpret_cast<const mgc::u64 *>(t->d_index), t->shift, d_bases, (mgc::u64)n_bases, t->k,
                       reinterpret_cast<const mgc::u64 *>(d_seq_start), (mgc::u64)n_seq, reinterpret_cast<mgc::u64 *>(d_total), reinterpret_cast<mgc::u64 *>(d_found));
  else
    hipLaunchKernelGGL((mgc::lookup_existence_kernel<mgc::u64>), dim3(g), dim3(256), 0, st, reinterpret_cast<const mgc::u64 *>(t->d_keys), t->d_vals,
                       reinterpret_cast<const mgc::u64 *>(t->d_index), t->shift, d_bases, (mgc::u64)n_bases, t->k,
                       reinterpret_cast<const mgc::u64 *>(d_seq_start), (mgc::u64)n_seq, reinterpret_cast<mgc::u64 *>(d_total), reinterpret_cast<mgc::u64 *>(d_found));
  return lk_rc(hipGetLastError(), "lookup_existence");
}
