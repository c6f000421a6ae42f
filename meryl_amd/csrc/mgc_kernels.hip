// mgc_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the meryl `count` hot path.
//
// What each kernel replaces in the reference (paths relative to the reference root):
//   kmer_hist_kernel / kmer_partition_kernel
//       kmerIterator + insertKmers            src/meryl/merylOp-countThreads.C:235-280
//       (2-bit pack A0 C1 T2 G3, reverse complement, canonical pick, prefix split;
//        the per-bucket spin-lock + bit-packed append of merylCountArray.C:490-728
//        becomes a histogram + lock-free scatter into per-file regions)
//   radix_* kernels
//       unpack + std::sort of each bucket      src/meryl/merylCountArray.C:276-289,330
//   rle_* kernels
//       the two run-length passes              src/meryl/merylCountArray.C:334-358
//   block_offsets_kernel
//       the per-prefix (prefix, nKmers) split that feeds addBlock
//                                              src/meryl/merylCountArray.C:472-475
//
// All of it is integer / byte work bounded by HBM bandwidth: loads are 16 B (bases) or
// 8 B per lane coalesced, every reorder is staged through LDS so stores leave as
// contiguous runs, ranking uses 64-lane ballots, cross-workgroup prefixes use 8-byte
// {flag,epoch,value} granules with agent-scope relaxed atomics (no fences needed:
// the datum is the flag).  Wave = 64 everywhere.
#include "mgc_device.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace mgc {

typedef unsigned long long u64;
typedef unsigned int       u32;

typedef unsigned __int128 u128;

// 128-bit key (k in 33..64): little-endian halves, so memory order == integer order of lo|hi<<64
struct alignas(16) K128 { u64 lo, hi; };

template <typename K> struct KeyOps;
template <> struct KeyOps<u64> {
  static constexpr int WORDS = 1;
  static __device__ __forceinline__ u32  digit(u64 k, u32 shift, u32 mask) { return (u32)(k >> shift) & mask; }
  static __device__ __forceinline__ u32  bucket(u64 k, u32 shift) { return (u32)(k >> shift); }
  static __device__ __forceinline__ u64  pad() { return ~0ull; }
  static __device__ __forceinline__ u64  zero() { return 0ull; }
  static __device__ __forceinline__ bool ne(u64 a, u64 b) { return a != b; }
  static __device__ __forceinline__ bool lt(u64 a, u64 b) { return a < b; }
  static __device__ __forceinline__ u64  prefix_floor(u64 p, u32 w_data) { return p << w_data; }
};
template <> struct KeyOps<K128> {
  static constexpr int WORDS = 2;
  static __device__ __forceinline__ u128 v(K128 k) { return ((u128)k.hi << 64) | (u128)k.lo; }
  static __device__ __forceinline__ K128 mk(u128 x) { K128 k; k.lo = (u64)x; k.hi = (u64)(x >> 64); return k; }
  static __device__ __forceinline__ u32  digit(K128 k, u32 shift, u32 mask) { return (u32)(v(k) >> shift) & mask; }
  static __device__ __forceinline__ u32  bucket(K128 k, u32 shift) { return (u32)(v(k) >> shift); }
  static __device__ __forceinline__ K128 pad() { K128 k; k.lo = ~0ull; k.hi = ~0ull; return k; }
  static __device__ __forceinline__ K128 zero() { K128 k; k.lo = 0; k.hi = 0; return k; }
  static __device__ __forceinline__ bool ne(K128 a, K128 b) { return (a.lo != b.lo) || (a.hi != b.hi); }
  static __device__ __forceinline__ bool lt(K128 a, K128 b) { return (a.hi < b.hi) || (a.hi == b.hi && a.lo < b.lo); }
  static __device__ __forceinline__ K128 prefix_floor(u64 p, u32 w_data) { return mk((u128)p << w_data); }
};

#define MGC_CHECK(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return e__; } while (0)

// ============================================================================
//  Block-level helpers (wave = 64)
// ============================================================================

__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ u32 wave_id() { return threadIdx.x >> 6; }

// Exclusive prefix sum over one value per thread.  s_tmp: >= BLOCK/64 + 1 entries.
// Every thread of the block must call it.  Leaves the block total in *total.
template <int BLOCK, typename T>
__device__ __forceinline__ T block_excl_scan(T v, T *s_tmp, T *total) {
  constexpr int NW = BLOCK / 64;
  const u32 lane = lane_id(), w = wave_id();
  T x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T y = __shfl_up(x, d);
    if ((int)lane >= d) x += y;
  }
  __syncthreads();                       // s_tmp may still be read from a previous call
  if (lane == 63) s_tmp[w] = x;
  __syncthreads();
  T wave_base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) {
    T t = s_tmp[i];
    if (i < (int)w) wave_base += t;
    tot += t;
  }
  *total = tot;
  return wave_base + x - v;
}

// ============================================================================
//  k-mer extraction (k <= 32, keys are uint64)
// ============================================================================

// 4 ASCII bytes (byte 0 = first base) -> 8 bits of 2-bit codes, first base most
// significant.  code = (ascii >> 1) & 3 gives A0 C1 T2 G3 for both cases.
__device__ __forceinline__ u32 enc4(u32 w) {
  return (((w >> 1) & 0x03030303u) * 0x40100401u) >> 24;
}
// exact per-byte zero detector: 0x80 in every byte of x that is zero
__device__ __forceinline__ u32 zero_bytes(u32 x) {
  return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}
// 4 ASCII bytes -> 4-bit mask, bit 3 = byte 0 is NOT one of ACGTacgt
__device__ __forceinline__ u32 inv4(u32 w) {
  const u32 u = w & 0xDFDFDFDFu;                     // fold case
  const u32 ok = zero_bytes(u ^ 0x41414141u) | zero_bytes(u ^ 0x43434343u) |
                 zero_bytes(u ^ 0x47474747u) | zero_bytes(u ^ 0x54545454u);
  const u32 g = ((~ok) & 0x80808080u) >> 7;
  return ((g * 0x08040201u) >> 24) & 0xFu;
}

__device__ __forceinline__ uint4 load16(const uint8_t *__restrict__ bases, u64 pos, u64 n, bool aligned) {
  if (aligned && pos + 16 <= n)
    return *reinterpret_cast<const uint4 *>(bases + pos);
  u32 w[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    w[i] = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const u64 p = pos + (u64)(i * 4 + b);
      const u32 c = (p < n) ? (u32)bases[p] : (u32)'.';
      w[i] |= c << (8 * b);
    }
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ void encode16(uint4 v, u32 &codes, u32 &inval) {
  codes = (enc4(v.x) << 24) | (enc4(v.y) << 16) | (enc4(v.z) << 8) | enc4(v.w);
  inval = (inv4(v.x) << 12) | (inv4(v.y) << 8) | (inv4(v.z) << 4) | inv4(v.w);
}

// reverse complement of a right-aligned k-mer (complement = xor 2 per base)
__device__ __forceinline__ u64 revcomp64(u64 f, u32 key_shift /* 64-2k */) {
  u64 x = __brevll(f ^ 0xAAAAAAAAAAAAAAAAull);       // reverses bases AND the two bits of each base
  x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
  return x >> key_shift;
}

constexpr int KP_WORDS = KP_TILE / 16 + 4;            // 16 bases per staged word + 64-base halo

// Stage one tile of bases as 2-bit codes + invalid masks in LDS.
__device__ __forceinline__ void kp_stage_tile(const uint8_t *__restrict__ bases, u64 n, u64 tile0, bool aligned,
                                              u32 *s_codes, u32 *s_inval) {
  const u32 t = threadIdx.x;
  {
    u32 c, iv;
    encode16(load16(bases, tile0 + (u64)t * 16, n, aligned), c, iv);
    s_codes[t] = c; s_inval[t] = iv;
  }
  if (t < 4) {
    u32 c, iv;
    encode16(load16(bases, tile0 + (u64)KP_TILE + (u64)t * 16, n, aligned), c, iv);
    s_codes[KP_BLOCK + t] = c; s_inval[KP_BLOCK + t] = iv;
  }
}

// The 16 k-mers starting at tile positions threadIdx.x*16 .. +15.  Returns the
// bit mask of positions that hold a complete k-mer.
__device__ __forceinline__ u32 kp_thread_kmers(const u32 *s_codes, const u32 *s_inval, u32 k, int mode,
                                               u64 (&keys)[KP_ITEMS]) {
  const u32 t = threadIdx.x;
  const u64 A = ((u64)s_codes[t] << 32) | (u64)s_codes[t + 1];      // bases 0..31 of the thread's window
  const u64 B = (u64)s_codes[t + 2] << 32;                          // bases 32..47
  const u64 I = ((u64)s_inval[t] << 48) | ((u64)s_inval[t + 1] << 32) | ((u64)s_inval[t + 2] << 16);
  const u32 key_shift = 64 - 2 * k;
  const u32 top_shift = 2 * k - 2;
  u32 vmask = 0;
  u64 r = 0;
#pragma unroll
  for (int j = 0; j < KP_ITEMS; j++) {
    const u64 top = (j == 0) ? A : ((A << (2 * j)) | (B >> (64 - 2 * j)));
    const u64 f   = top >> key_shift;
    if (j == 0) r = revcomp64(f, key_shift);
    else        r = (r >> 2) | ((((f & 3ull) ^ 2ull)) << top_shift);
    const bool ok = (((I << j) >> (64 - k)) == 0ull);
    u64 key;
    if      (mode == 1) key = f;
    else if (mode == 2) key = r;
    else                key = (f < r) ? f : r;
    keys[j] = key;
    vmask |= (ok ? 1u : 0u) << j;
  }
  return vmask;
}

// Same for k in 33..64: the thread's window is 80 bases (five staged words).
__device__ __forceinline__ u32 kp_thread_kmers(const u32 *s_codes, const u32 *s_inval, u32 k, int mode,
                                               K128 (&keys)[KP_ITEMS]) {
  const u32 t = threadIdx.x;
  const u128 A = ((u128)s_codes[t] << 96) | ((u128)s_codes[t + 1] << 64) | ((u128)s_codes[t + 2] << 32) |
                 (u128)s_codes[t + 3];                               // bases 0..63 of the window
  const u128 B = (u128)s_codes[t + 4] << 96;                         // bases 64..79
  const u128 I = ((u128)s_inval[t] << 112) | ((u128)s_inval[t + 1] << 96) | ((u128)s_inval[t + 2] << 80) |
                 ((u128)s_inval[t + 3] << 64) | ((u128)s_inval[t + 4] << 48);
  const u32 key_shift = 128 - 2 * k;
  const u32 top_shift = 2 * k - 2;
  u32 vmask = 0;
  u128 r = 0;
#pragma unroll
  for (int j = 0; j < KP_ITEMS; j++) {
    const u128 top = (j == 0) ? A : ((A << (2 * j)) | (B >> (128 - 2 * j)));
    const u128 f   = top >> key_shift;
    if (j == 0) {
      // reverse complement: complement, reverse all 128 bits, swap the two bits of every base
      const u128 c = f ^ (((u128)0xAAAAAAAAAAAAAAAAull << 64) | (u128)0xAAAAAAAAAAAAAAAAull);
      u64 lo = __brevll((u64)(c >> 64)), hi = __brevll((u64)c);       // halves swap when reversed
      lo = ((lo >> 1) & 0x5555555555555555ull) | ((lo & 0x5555555555555555ull) << 1);
      hi = ((hi >> 1) & 0x5555555555555555ull) | ((hi & 0x5555555555555555ull) << 1);
      r = (((u128)hi << 64) | (u128)lo) >> key_shift;
    } else {
      r = (r >> 2) | ((u128)(((u64)f & 3ull) ^ 2ull) << top_shift);
    }
    const bool ok = (((I << j) >> (128 - k)) == (u128)0);
    u128 key;
    if      (mode == 1) key = f;
    else if (mode == 2) key = r;
    else                key = (f < r) ? f : r;
    keys[j] = KeyOps<K128>::mk(key);
    vmask |= (ok ? 1u : 0u) << j;
  }
  return vmask;
}

__device__ __forceinline__ void kp_tile_range(u64 num_tiles, u64 &t_begin, u64 &t_end) {
  const u64 per = (num_tiles + gridDim.x - 1) / gridDim.x;
  t_begin = (u64)blockIdx.x * per;
  t_end   = t_begin + per;
  if (t_begin > num_tiles) t_begin = num_tiles;
  if (t_end   > num_tiles) t_end   = num_tiles;
}

// Pass 1: per-workgroup and global per-bucket instance counts.
template <typename K>
__global__ __launch_bounds__(KP_BLOCK)
void kmer_hist_kernel(const uint8_t *__restrict__ bases, u64 n, u32 k, int mode, u32 bucket_bits,
                      u64 num_tiles, u64 *__restrict__ block_hist, u64 *__restrict__ bucket_counts) {
  __shared__ u32 s_codes[KP_WORDS];
  __shared__ u32 s_inval[KP_WORDS];
  __shared__ u32 s_hist[KP_MAX_BUCKETS];

  const u32  nb = 1u << bucket_bits;
  const u32  bucket_shift = 2 * k - bucket_bits;
  const bool aligned = ((reinterpret_cast<uintptr_t>(bases) & 15) == 0);

  for (u32 b = threadIdx.x; b < nb; b += KP_BLOCK) s_hist[b] = 0;
  __syncthreads();

  u64 t_begin, t_end;
  kp_tile_range(num_tiles, t_begin, t_end);
  u32 my_count = 0;

  for (u64 tile = t_begin; tile < t_end; tile++) {
    kp_stage_tile(bases, n, tile * KP_TILE, aligned, s_codes, s_inval);
    __syncthreads();
    K keys[KP_ITEMS];
    const u32 vmask = kp_thread_kmers(s_codes, s_inval, k, mode, keys);
    if (nb == 1) {
      my_count += __popc(vmask);
    } else {
#pragma unroll
      for (int j = 0; j < KP_ITEMS; j++)
        if ((vmask >> j) & 1u) atomicAdd(&s_hist[KeyOps<K>::bucket(keys[j], bucket_shift)], 1u);
    }
    __syncthreads();
  }
  if (nb == 1) atomicAdd(&s_hist[0], my_count);
  __syncthreads();

  for (u32 b = threadIdx.x; b < nb; b += KP_BLOCK) {
    const u64 v = s_hist[b];
    block_hist[(u64)blockIdx.x * nb + b] = v;
    if (v) atomicAdd(&bucket_counts[b], v);
  }
}

// Turns per-workgroup counts into per-workgroup absolute write cursors
// (one workgroup per bucket scans that bucket's column of block_hist).
__global__ __launch_bounds__(256)
void kmer_scan_kernel(u64 *__restrict__ block_hist, u32 grid, u32 nb, const u64 *__restrict__ bucket_starts) {
  __shared__ u64 s_tmp[256 / 64 + 1];
  const u32 b = blockIdx.x;
  u64 carry = bucket_starts[b];
  for (u32 base = 0; base < grid; base += 256 * 8) {
    u64 v[8], sum = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const u32 g = base + threadIdx.x * 8 + q;
      v[q] = (g < grid) ? block_hist[(u64)g * nb + b] : 0ull;
      sum += v[q];
    }
    u64 tot;
    u64 run = carry + block_excl_scan<256, u64>(sum, s_tmp, &tot);
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const u32 g = base + threadIdx.x * 8 + q;
      if (g < grid) block_hist[(u64)g * nb + b] = run;
      run += v[q];
    }
    carry += tot;
  }
}

// Pass 2: pack + scatter.  Each workgroup owns private cursors (from pass 1),
// so there are no global atomics and the result layout is deterministic up to
// the order inside a (workgroup, tile, bucket) run.
// MAXB: bucket capacity of the LDS tables (64 for the 64-file partition of the count path: 36 KiB of LDS per
// workgroup instead of 51, i.e. four workgroups per CU instead of three; 1024 for the general operator)
template <typename K, int MAXB>
__global__ __launch_bounds__(KP_BLOCK, 5)
void kmer_partition_kernel(const uint8_t *__restrict__ bases, u64 n, u32 k, int mode, u32 bucket_bits,
                           u64 num_tiles, const u64 *__restrict__ block_base, K *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kp_dyn_smem[];
  K *s_keys = reinterpret_cast<K *>(kp_dyn_smem);                   // K[KP_TILE]
  __shared__ u64 s_cursor[MAXB];
  __shared__ u32 s_cnt[MAXB];
  __shared__ u32 s_base[MAXB];
  __shared__ u32 s_codes[KP_WORDS];
  __shared__ u32 s_inval[KP_WORDS];
  __shared__ u32 s_tmp[KP_BLOCK / 64 + 1];

  const u32  nb = 1u << bucket_bits;
  const u32  bucket_shift = 2 * k - bucket_bits;
  const bool aligned = ((reinterpret_cast<uintptr_t>(bases) & 15) == 0);
  const u32  tid = threadIdx.x;

  for (u32 b = tid; b < nb; b += KP_BLOCK) s_cursor[b] = block_base[(u64)blockIdx.x * nb + b];

  u64 t_begin, t_end;
  kp_tile_range(num_tiles, t_begin, t_end);

  for (u64 tile = t_begin; tile < t_end; tile++) {
    for (u32 b = tid; b < nb; b += KP_BLOCK) s_cnt[b] = 0;
    kp_stage_tile(bases, n, tile * KP_TILE, aligned, s_codes, s_inval);
    __syncthreads();

    K keys[KP_ITEMS];
    const u32 vmask = kp_thread_kmers(s_codes, s_inval, k, mode, keys);
    u32 total = 0;

    if (nb == 1) {
      // plain compaction: exclusive scan of per-thread counts
      const u32 c = __popc(vmask);
      const u32 base = block_excl_scan<KP_BLOCK, u32>(c, s_tmp, &total);
      u32 o = base;
#pragma unroll
      for (int j = 0; j < KP_ITEMS; j++)
        if ((vmask >> j) & 1u) s_keys[o++] = keys[j];
      if (tid == 0) { s_base[0] = 0; s_cnt[0] = total; }
      __syncthreads();
    } else {
      // rank inside the bucket with LDS atomics (order inside a bucket is free)
      u32 ranks[KP_ITEMS];
#pragma unroll
      for (int j = 0; j < KP_ITEMS; j++) {
        ranks[j] = 0;
        if ((vmask >> j) & 1u) ranks[j] = atomicAdd(&s_cnt[KeyOps<K>::bucket(keys[j], bucket_shift)], 1u);
      }
      __syncthreads();
      // exclusive scan of the bucket counts, 4 consecutive buckets per thread
      u32 v[4], sum = 0;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const u32 b = tid * 4 + q;
        v[q] = (b < nb) ? s_cnt[b] : 0u;
        sum += v[q];
      }
      u32 run = block_excl_scan<KP_BLOCK, u32>(sum, s_tmp, &total);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const u32 b = tid * 4 + q;
        if (b < nb) s_base[b] = run;
        run += v[q];
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < KP_ITEMS; j++)
        if ((vmask >> j) & 1u) s_keys[s_base[KeyOps<K>::bucket(keys[j], bucket_shift)] + ranks[j]] = keys[j];
      __syncthreads();
    }

    // contiguous runs per bucket leave as coalesced stores
    for (u32 i = tid; i < total; i += KP_BLOCK) {
      const K   key = s_keys[i];
      const u32 b   = (nb == 1) ? 0u : KeyOps<K>::bucket(key, bucket_shift);
      out[s_cursor[b] + (u64)(i - s_base[b])] = key;
    }
    __syncthreads();
    for (u32 b = tid; b < nb; b += KP_BLOCK) s_cursor[b] += s_cnt[b];
    __syncthreads();
  }
}

uint32_t kp_grid_size(uint64_t n_bases) {
  const uint64_t num_tiles = (n_bases + KP_TILE - 1) / KP_TILE;
  uint64_t g = num_tiles < 2048 ? num_tiles : 2048;
  return (uint32_t)(g ? g : 1);
}

size_t kp_workspace_bytes(uint32_t bucket_bits) {
  return (size_t)2048 * ((size_t)1 << bucket_bits) * sizeof(uint64_t);
}

hipError_t launch_kmer_histogram(const uint8_t *d_bases, uint64_t n_bases, uint32_t k, int mode,
                                 uint32_t bucket_bits, uint64_t *d_bucket_counts, void *d_ws, hipStream_t st) {
  const uint32_t nb = 1u << bucket_bits;
  MGC_CHECK(hipMemsetAsync(d_bucket_counts, 0, sizeof(uint64_t) * nb, st));
  if (n_bases == 0) return hipSuccess;
  const uint64_t num_tiles = (n_bases + KP_TILE - 1) / KP_TILE;
  const uint32_t grid = kp_grid_size(n_bases);
  if (k <= 32)
    hipLaunchKernelGGL(kmer_hist_kernel<u64>, dim3(grid), dim3(KP_BLOCK), 0, st,
                       d_bases, (u64)n_bases, k, mode, bucket_bits, (u64)num_tiles,
                       reinterpret_cast<u64 *>(d_ws), reinterpret_cast<u64 *>(d_bucket_counts));
  else
    hipLaunchKernelGGL(kmer_hist_kernel<K128>, dim3(grid), dim3(KP_BLOCK), 0, st,
                       d_bases, (u64)n_bases, k, mode, bucket_bits, (u64)num_tiles,
                       reinterpret_cast<u64 *>(d_ws), reinterpret_cast<u64 *>(d_bucket_counts));
  return hipGetLastError();
}

hipError_t launch_kmer_partition(const uint8_t *d_bases, uint64_t n_bases, uint32_t k, int mode,
                                 uint32_t bucket_bits, const uint64_t *d_bucket_starts, void *d_keys,
                                 void *d_ws, hipStream_t st) {
  if (n_bases == 0) return hipSuccess;
  const uint32_t nb = 1u << bucket_bits;
  const uint64_t num_tiles = (n_bases + KP_TILE - 1) / KP_TILE;
  const uint32_t grid = kp_grid_size(n_bases);
  hipLaunchKernelGGL(kmer_scan_kernel, dim3(nb), dim3(256), 0, st,
                     reinterpret_cast<u64 *>(d_ws), grid, nb, reinterpret_cast<const u64 *>(d_bucket_starts));
  MGC_CHECK(hipGetLastError());
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kmer_partition_kernel<K128, 64>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(KP_TILE * sizeof(K128)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kmer_partition_kernel<K128, KP_MAX_BUCKETS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(KP_TILE * sizeof(K128)));
    attr_done = true;
  }
#define MGC_KP_LAUNCH(K_, MAXB_)                                                                                   \
  hipLaunchKernelGGL((kmer_partition_kernel<K_, MAXB_>), dim3(grid), dim3(KP_BLOCK), KP_TILE * sizeof(K_), st,      \
                     d_bases, (u64)n_bases, k, mode, bucket_bits, (u64)num_tiles,                                  \
                     reinterpret_cast<const u64 *>(d_ws), reinterpret_cast<K_ *>(d_keys))
  if (k <= 32) { if (nb <= 64) MGC_KP_LAUNCH(u64, 64); else MGC_KP_LAUNCH(u64, KP_MAX_BUCKETS); }
  else         { if (nb <= 64) MGC_KP_LAUNCH(K128, 64); else MGC_KP_LAUNCH(K128, KP_MAX_BUCKETS); }
#undef MGC_KP_LAUNCH
  return hipGetLastError();
}

// ============================================================================
//  LSB radix sort of uint64 keys
// ============================================================================
//
// One pass = one stable counting sort on a digit of <= RB bits:
//   * keys of a tile (BLOCK*KPT consecutive keys) are loaded wave-striped, so a
//     wave reads 512 contiguous bytes per instruction;
//   * each wave ranks its keys with RB ballots per key (peers holding the same
//     digit) against a wave-private LDS digit counter -- data-independent cost,
//     stable by construction;
//   * digit totals of the tile are prefix-summed; in ONESWEEP mode the tile's
//     global digit bases come from a decoupled look-back over earlier tiles'
//     8-byte status granules, in CLASSIC mode from a precomputed table;
//   * keys are permuted through LDS into digit order and leave as contiguous
//     runs (one run per digit), i.e. coalesced 8 B/lane stores.
// Algorithmic HBM traffic per pass: 8 B read + 8 B write per key (ONESWEEP; the
// digit histograms of all passes are taken in one extra 8 B/key read up front).

constexpr int      RS_MAX_PASSES = 16;
constexpr int      RS_MAX_RADIX  = 512;
constexpr u32      RS_SPIN_LIMIT = 1u << 24;

struct SortHeader {                       // lives at the start of the sort workspace
  u64 ghist[RS_MAX_PASSES][RS_MAX_RADIX]; // digit counts per pass
  u64 gbase[RS_MAX_PASSES][RS_MAX_RADIX]; // exclusive digit bases per pass
  u64 row_total[RS_MAX_RADIX];            // classic mode scratch
  u32 ticket[RS_MAX_PASSES];
  u32 pad[16];
};

struct PassList { u32 n; u32 shift[RS_MAX_PASSES]; u32 mask[RS_MAX_PASSES]; };

// Digit histograms of every pass in one read of the keys.  NP > 0: exactly NP passes (shifts and masks stay in
// registers, the LDS histogram is NP rows: 4 KiB for the two passes of the finish path, so eight 256-thread
// workgroups fit a CU); NP == 0: any number of passes up to RS_MAX_PASSES.
template <typename K, int NP>
__global__ __launch_bounds__(256)
void radix_hist_kernel(const K *__restrict__ in, u64 n, PassList pl, u64 *__restrict__ ghist) {
  constexpr u32 ROWS = NP ? NP : RS_MAX_PASSES;
  __shared__ u32 s_h[ROWS * RS_MAX_RADIX];
  const u32 np = NP ? (u32)NP : pl.n;
  for (u32 i = threadIdx.x; i < np * RS_MAX_RADIX; i += 256) s_h[i] = 0;
  __syncthreads();

  // 4 independent loads in flight per thread (the loop is otherwise latency-bound)
  constexpr u32 UNR = 4;
  const u64 gstride = (u64)gridDim.x * 256 * UNR;
  for (u64 base = (u64)blockIdx.x * 256 * UNR; base < n; base += gstride) {
    K key[UNR];
    bool ok[UNR];
#pragma unroll
    for (u32 j = 0; j < UNR; j++) {
      const u64 i = base + (u64)j * 256 + threadIdx.x;
      ok[j] = i < n;
      if (ok[j]) key[j] = in[i];
    }
#pragma unroll
    for (u32 j = 0; j < UNR; j++) {
      if (!ok[j]) continue;
      if (NP) {
#pragma unroll
        for (u32 p = 0; p < ROWS; p++)
          atomicAdd(&s_h[p * RS_MAX_RADIX + KeyOps<K>::digit(key[j], pl.shift[p], pl.mask[p])], 1u);
      } else {
        for (u32 p = 0; p < np; p++)
          atomicAdd(&s_h[p * RS_MAX_RADIX + KeyOps<K>::digit(key[j], pl.shift[p], pl.mask[p])], 1u);
      }
    }
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < np * RS_MAX_RADIX; i += 256) {
    const u32 v = s_h[i];
    if (v) atomicAdd(&ghist[i], (u64)v);
  }
}

template <typename K>
static void launch_radix_hist(const K *src, u64 n, const PassList &pl, u64 *ghist, hipStream_t st) {
  uint64_t hgrid = (n + 256 * 16 - 1) / (256 * 16);
  if (hgrid > 2048) hgrid = 2048;
  const dim3 g((uint32_t)hgrid), b(256);
  switch (pl.n) {
    case 1:  hipLaunchKernelGGL((radix_hist_kernel<K, 1>), g, b, 0, st, src, n, pl, ghist); break;
    case 2:  hipLaunchKernelGGL((radix_hist_kernel<K, 2>), g, b, 0, st, src, n, pl, ghist); break;
    case 3:  hipLaunchKernelGGL((radix_hist_kernel<K, 3>), g, b, 0, st, src, n, pl, ghist); break;
    case 4:  hipLaunchKernelGGL((radix_hist_kernel<K, 4>), g, b, 0, st, src, n, pl, ghist); break;
    default: hipLaunchKernelGGL((radix_hist_kernel<K, 0>), g, b, 0, st, src, n, pl, ghist); break;
  }
}

// Exclusive scan over the digits of each pass (one workgroup per pass).
__global__ __launch_bounds__(RS_MAX_RADIX)
void radix_digit_scan_kernel(const u64 *__restrict__ ghist, u64 *__restrict__ gbase) {
  __shared__ u64 s_tmp[RS_MAX_RADIX / 64 + 1];
  const u32 p = blockIdx.x;
  const u64 v = ghist[(u64)p * RS_MAX_RADIX + threadIdx.x];
  u64 total;
  const u64 e = block_excl_scan<RS_MAX_RADIX, u64>(v, s_tmp, &total);
  gbase[(u64)p * RS_MAX_RADIX + threadIdx.x] = e;
}

template <int RB>
__device__ __forceinline__ u64 match_digit(u32 d) {
  u64 peers = ~0ull;
#pragma unroll
  for (int b = 0; b < RB; b++) {
    const bool bit = (d >> b) & 1u;
    const u64  m   = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

__device__ __forceinline__ void status_store(u64 *p, u64 v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 status_load(u64 *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Look-back status granule: one 8-byte word carries TWO digits, each as
// flag(2) | value(30); flag 1 = tile aggregate, 2 = inclusive prefix.  Half as many
// fabric transactions as one granule per digit, and a digit pair is published by one
// store, so no fence is needed (the datum is the flag).  Values < 2^30: the look-back
// path handles n < 2^30 keys per sort call, larger calls take the classic path.
__device__ __forceinline__ u64 st_pack(u32 f0, u32 v0, u32 f1, u32 v1) {
  return ((u64)((f1 << 30) | v1) << 32) | (u64)((f0 << 30) | v0);
}

typedef __attribute__((address_space(3))) u64 lds_u64;
typedef __attribute__((address_space(3))) u32 lds_u32;

template <typename K, int RB, int BLOCK, int KPT, int LB = 0>
struct RadixSmem {
  static constexpr int R     = 1 << RB;
  static constexpr int NW    = BLOCK / 64;
  static constexpr int TILE  = BLOCK * KPT;
  // region 0 is shared between the ranking scratch (wave digit counters u32[NW][R] followed by
  // wave match masks u64[NW][R]) and the key exchange buffer
  static constexpr size_t RANK_BYTES = (size_t)NW * R * 12;
  static constexpr size_t REGION0 = ((size_t)TILE * sizeof(K) > RANK_BYTES) ? (size_t)TILE * sizeof(K) : RANK_BYTES;
  static constexpr size_t OFF_GBASE = REGION0;                     // u64[R]
  static constexpr size_t OFF_DBASE = OFF_GBASE + (size_t)R * 8;   // u32[R]
  static constexpr size_t OFF_CNT   = OFF_DBASE + (size_t)R * 4;   // u32[R]
  static constexpr size_t OFF_TMP   = OFF_CNT + (size_t)R * 4;     // u32[64] scan scratch + misc
  static constexpr size_t OFF_WIN   = OFF_TMP + 64 * 4;            // u64[LB_WINDOW][R/2] look-back window (LB == 2)
  static constexpr size_t WIN_BYTES = (LB == 2 || LB == 3) ? ((RB == 9 && BLOCK == 512) ? 4096 : 8192) : 0;
  static constexpr size_t BYTES     = OFF_WIN + WIN_BYTES;
  // workgroups per CU the LDS budget admits (160 KiB per CU), capped at 2
  static constexpr int    WG_PER_CU = (2 * BYTES <= 160 * 1024) ? 2 : 1;
  static constexpr int    MIN_WAVES_PER_SIMD = (WG_PER_CU * BLOCK) / 256;
};

template <typename K, int RB, int BLOCK, int KPT, int LB, int MATCH>
__global__ __launch_bounds__(BLOCK, (RadixSmem<K, RB, BLOCK, KPT, LB>::MIN_WAVES_PER_SIMD))
void radix_scatter_kernel(const K *__restrict__ in, K *__restrict__ out, u64 n, u32 shift, u32 dmask,
                          const u64 *__restrict__ gbase,      // LOOKBACK: exclusive digit bases of this pass
                          u64 *__restrict__ status,           // LOOKBACK: [num_tiles][R/2] granules of this pass (zeroed)
                          u32 *__restrict__ ticket, u32 *__restrict__ error_flag,
                          u32 flags,                          // bit0: XCD-chunked tile order, bit1: non-temporal key loads
                          const u64 *__restrict__ tile_offs,  // !LOOKBACK: [R][num_tiles] absolute offsets
                          u64 num_tiles, u64 *__restrict__ dbg /* optional: 8 cycle stamps per tile */) {
  using SM = RadixSmem<K, RB, BLOCK, KPT, LB>;
#define MGC_STAMP(i) do { if (dbg && threadIdx.x == 0) dbg_t[i] = clock64(); } while (0)
  u64 dbg_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  MGC_STAMP(0);
  using KO = KeyOps<K>;
  constexpr int R = SM::R, NW = SM::NW, TILE = SM::TILE;
  static_assert(BLOCK >= R, "one thread per digit needed");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  K   *s_keys  = reinterpret_cast<K *>(smem);
  u32 *s_whist = reinterpret_cast<u32 *>(smem);                       // aliases s_keys (see barriers)
  u64 *s_gbase = reinterpret_cast<u64 *>(smem + SM::OFF_GBASE);
  u32 *s_dbase = reinterpret_cast<u32 *>(smem + SM::OFF_DBASE);
  u32 *s_cnt   = reinterpret_cast<u32 *>(smem + SM::OFF_CNT);
  u32 *s_tmp   = reinterpret_cast<u32 *>(smem + SM::OFF_TMP);

  const u32 tid = threadIdx.x, lane = lane_id(), w = wave_id();

  constexpr bool LOOKBACK = (LB != 0);
  // Tile ids are tickets: every lower-numbered tile has started, hence is resident and will
  // publish its aggregate -- the look-back cannot deadlock (spins are bounded anyway).
  u64 tile;
  if (LOOKBACK) {
    if (tid == 0) s_tmp[32] = atomicAdd(ticket, 1u);
    __syncthreads();
    tile = s_tmp[32];
  } else {
    tile = blockIdx.x;
  }

  for (u32 i = tid; i < (u32)(NW * R * 3); i += BLOCK) s_whist[i] = 0;   // counters + match masks
  __syncthreads();

  MGC_STAMP(1);
  // ---- load (wave-striped: 512 contiguous bytes per wave instruction) ----
  const u64  tile_base = tile * (u64)TILE;
  const bool full      = (tile_base + TILE <= n);
  const u64  wave_base = tile_base + (u64)w * (64 * KPT) + lane;
  K keys[KPT];
#pragma unroll
  for (int j = 0; j < KPT; j++) {
    const u64 idx = wave_base + (u64)j * 64;
    keys[j] = (full || idx < n) ? in[idx] : KO::pad();  // padding sorts to the end of the last digit
  }
  (void)flags;

  // ---- rank inside the wave (stable: by lane order inside a row, rows in order) ----
  const u64 lt_mask = (1ull << lane) - 1ull;
  u32 ranks[KPT / 2];                                  // two 16-bit ranks per register (rank < TILE <= 2^14)
  if (MATCH == 0) {
    // peers by RB ballots per key: data-independent cost, VALU heavy
    lds_u32 *wh = (lds_u32 *)(smem) + w * R;
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const u32 d     = KO::digit(keys[j], shift, dmask);
      const u64 peers = match_digit<RB>(d);
      const u32 lower = __popcll(peers & lt_mask);
      const u32 base  = __hip_atomic_load(&wh[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      if (lower == 0) __hip_atomic_store(&wh[d], base + (u32)__popcll(peers), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      if (j & 1) ranks[j / 2] |= (base + lower) << 16;
      else       ranks[j / 2]  = (base + lower);
    }
  } else {
    // peers through a wave-private LDS mask per digit: every lane ORs its lane bit into
    // mask[digit], reads the mask back (LDS executes a wave's instructions in order), and the
    // lowest peer bumps the running digit counter and clears the mask for the next row.
    lds_u32 *wh = (lds_u32 *)(smem) + w * R;
    lds_u64 *mk = (lds_u64 *)(smem + (size_t)NW * R * 4) + w * R;
    const u64 lane_bit = 1ull << lane;
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const u32 d = KO::digit(keys[j], shift, dmask);
      __hip_atomic_fetch_or(&mk[d], lane_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      const u64 peers = __hip_atomic_load(&mk[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      const u32 base  = __hip_atomic_load(&wh[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      const u32 lower = __popcll(peers & lt_mask);
      if (lower == 0) {
        __hip_atomic_store(&wh[d], base + (u32)__popcll(peers), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_store(&mk[d], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
      if (j & 1) ranks[j / 2] |= (base + lower) << 16;
      else       ranks[j / 2]  = (base + lower);
    }
  }
  __syncthreads();

  MGC_STAMP(2);
  // ---- digit totals of the tile, wave-exclusive bases ----
  const u32 n_valid = full ? (u32)TILE : (u32)(n - tile_base);
  u32 count = 0;
  if (tid < (u32)R) {
    u32 acc = 0;
#pragma unroll
    for (int ww = 0; ww < NW; ww++) {
      const u32 t = s_whist[ww * R + tid];
      s_whist[ww * R + tid] = acc;
      acc += t;
    }
    count = acc;
    // padding keys all carry the top digit; they are not published to later tiles
    s_cnt[tid] = (tid == dmask) ? count - ((u32)TILE - n_valid) : count;
  }
  u32 tile_total;
  const u32 excl = block_excl_scan<BLOCK, u32>(count, s_tmp, &tile_total);
  if (tid < (u32)R) s_dbase[tid] = excl;

  // granules of this tile: publish the aggregate as early as possible
  // LB 1/2: two digits per granule (30-bit values, n < 2^30); LB 3: one digit per granule (62-bit values)
  constexpr int GW = (LB == 3) ? 1 : 2;                 // digits per granule
  constexpr int G  = R / GW;                            // granules per tile
  constexpr u64 V62 = (1ull << 62) - 1;
  u64 *mine = status + (LOOKBACK ? tile * (u64)G + tid : 0);
  if (LOOKBACK) {
    __syncthreads();                                    // s_cnt / s_dbase visible
    if (tid < (u32)G) {
      const u32 c0 = s_cnt[GW * tid], c1 = (GW == 2) ? s_cnt[GW * tid + 1] : 0u;
      const u32 fl = (tile == 0) ? 2u : 1u;
      status_store(mine, (GW == 2) ? st_pack(fl, c0, fl, c1) : (((u64)fl << 62) | (u64)c0));
      if constexpr (LB == 1) {
        // serial walk per digit pair, four predecessors in flight per round
        u32 p0 = 0, p1 = 0;
        if (tile != 0) {
          bool need0 = true, need1 = true;
          u64  t = tile - 1;
          u32  spins = 0;
          while (need0 || need1) {
            u64 g[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
              g[i] = (t >= (u64)i) ? status_load(status + (t - i) * (u64)G + tid) : st_pack(2, 0, 2, 0);
            u32 used = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
              if (need0 || need1) {
                const u32 lo = (u32)g[i], hi = (u32)(g[i] >> 32);
                const u32 f0 = lo >> 30, f1 = hi >> 30;
                if (f0 == 0 || f1 == 0) break;
                if (need0) { p0 += lo & 0x3FFFFFFFu; if (f0 == 2) need0 = false; }
                if (need1) { p1 += hi & 0x3FFFFFFFu; if (f1 == 2) need1 = false; }
                used++;
              }
            }
            t -= (used <= t) ? used : t;
            if (used == 0) {
              if (++spins > RS_SPIN_LIMIT) { atomicExch(error_flag, 1u); break; }
              __builtin_amdgcn_s_sleep(1);
            }
          }
          status_store(mine, st_pack(2, p0 + c0, 2, p1 + c1));
        }
        s_gbase[2 * tid]     = gbase[2 * tid]     + (u64)p0 - (u64)s_dbase[2 * tid];
        s_gbase[2 * tid + 1] = gbase[2 * tid + 1] + (u64)p1 - (u64)s_dbase[2 * tid + 1];
      }
    }
    if (LB == 1) __syncthreads();
  } else {
    if (tid < (u32)R) s_gbase[tid] = tile_offs[(u64)tid * num_tiles + tile] - (u64)excl;
    __syncthreads();
  }

  MGC_STAMP(3);
  // ---- final position of every key inside the sorted tile ----
#pragma unroll
  for (int j = 0; j < KPT; j++) {           // ranks[] becomes positions in place (still < TILE)
    const u32 d = KO::digit(keys[j], shift, dmask);
    const u32 add = s_dbase[d] + s_whist[w * R + d];
    ranks[j / 2] += (j & 1) ? (add << 16) : add;
  }
  __syncthreads();                          // s_whist is dead; its storage becomes s_keys
#pragma unroll
  for (int j = 0; j < KPT; j++) s_keys[(j & 1) ? (ranks[j / 2] >> 16) : (ranks[j / 2] & 0xFFFFu)] = keys[j];
  __syncthreads();                          // keys now live in LDS only: registers are free for the look-back
  MGC_STAMP(4);

  if constexpr (LB == 2 || LB == 3) {
    // window-parallel look-back after the exchange (keys live in LDS only, registers are free):
    // all waves fetch the granules of the next LB_WINDOW predecessors with coalesced loads into
    // LDS, the R/2 digit-pair threads consume the ready prefix.
    constexpr int TPL = BLOCK / G;                      // predecessor tiles covered by one load per thread
    constexpr int WIN0 = (int)(SM::WIN_BYTES / ((size_t)G * 8));
    constexpr int LB_WINDOW = (WIN0 < TPL) ? TPL : WIN0;
    constexpr int LPT = LB_WINDOW / TPL;
    static_assert(LB_WINDOW % TPL == 0 && LPT >= 1, "window must be a multiple of the tiles one load covers");
    static_assert((size_t)LB_WINDOW * G * 8 <= SM::WIN_BYTES, "look-back window does not fit");
    u64 *s_win = reinterpret_cast<u64 *>(smem + SM::OFF_WIN);
    u32 *s_q   = s_tmp + 40;
    u32 *s_all = s_tmp + 41;
    u32 c0 = 0, c1 = 0;
    u64 p0 = 0, p1 = 0;
    bool need0 = false, need1 = false;
    if (tid < (u32)G) {
      c0 = s_cnt[GW * tid]; c1 = (GW == 2) ? s_cnt[GW * tid + 1] : 0u;
      need0 = (tile != 0);
      need1 = (GW == 2) && (tile != 0);
    }
    if (tile != 0) {                                    // uniform
      const u32 sub = tid / G, g = tid % G;
      u64 t_next = tile - 1;
      u32 spins = 0;
      while (true) {
#pragma unroll
        for (int i = 0; i < LPT; i++) {
          const u32 idx = sub + (u32)(TPL * i);
          s_win[idx * G + g] = (t_next >= (u64)idx) ? status_load(status + (t_next - idx) * (u64)G + g)
                                                    : ((GW == 2) ? st_pack(2, 0, 2, 0) : (2ull << 62));
        }
        if (tid == 0) { *s_q = LB_WINDOW; *s_all = 1u; }
        __syncthreads();
        if (tid < (u32)G) {
          u32 q = 0;
          for (; q < (u32)LB_WINDOW; q++) {
            const u64 v = s_win[q * G + tid];
            if ((GW == 2 && ((u32)v >> 30) == 0u) || ((u32)(v >> 62)) == 0u) break;
          }
          if (q < (u32)LB_WINDOW) atomicMin(s_q, q);
        }
        __syncthreads();
        const u32 q = *s_q;
        if (tid < (u32)G) {
          for (u32 i = 0; i < q && (need0 || need1); i++) {
            const u64 v = s_win[i * G + tid];
            if (GW == 2) {
              const u32 lo = (u32)v, hi = (u32)(v >> 32);
              if (need0) { p0 += lo & 0x3FFFFFFFu; if ((lo >> 30) == 2u) need0 = false; }
              if (need1) { p1 += hi & 0x3FFFFFFFu; if ((hi >> 30) == 2u) need1 = false; }
            } else {
              p0 += v & V62;
              if ((v >> 62) == 2ull) need0 = false;
            }
          }
          if (need0 || need1) *s_all = 0u;
        }
        __syncthreads();
        if (*s_all) break;
        t_next -= (q <= t_next) ? q : t_next;
        if (q == 0) {
          if (++spins > RS_SPIN_LIMIT) { if (tid == 0) atomicExch(error_flag, 1u); break; }
          __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
      }
    }
    if (tid < (u32)G) {
      if (tile != 0)
        status_store(mine, (GW == 2) ? st_pack(2, (u32)p0 + c0, 2, (u32)p1 + c1) : ((2ull << 62) | (p0 + (u64)c0)));
      s_gbase[GW * tid] = gbase[GW * tid] + p0 - (u64)s_dbase[GW * tid];
      if (GW == 2) s_gbase[2 * tid + 1] = gbase[2 * tid + 1] + p1 - (u64)s_dbase[2 * tid + 1];
    }
    __syncthreads();
  }

  MGC_STAMP(5);
  // ---- contiguous runs leave coalesced ----
#pragma unroll
  for (int j = 0; j < KPT; j++) {
    const u32 i = (u32)j * BLOCK + tid;
    if (i < n_valid) {
      const K   key = s_keys[i];
      const u32 d   = KO::digit(key, shift, dmask);
      out[s_gbase[d] + (u64)i] = key;
    }
  }
  MGC_STAMP(6);
  if (dbg && threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 7; i++) dbg[tile * 8 + i] = dbg_t[i];
  }
#undef MGC_STAMP
}

// ---- pipelined look-back pass (plan.lookback == 5) ----------------------------
// Same pass as radix_scatter_kernel<.., LB, MATCH=1>, restructured so HBM never idles behind the
// per-tile work: workgroups are persistent, take the ticket of their NEXT tile at the top of an
// iteration, and once the keys of the current tile live in LDS (after the exchange) the registers are
// refilled with the next tile's keys while the look-back and the write-out run.  s_waitcnt vmcnt is
// per wave and in order, so the waves that poll the status granules (tid < R/2, the first R/128
// waves) fetch their share of the next tile only after their walk; all other waves fetch before it.
// A ticket held one tile ahead keeps the no-deadlock argument: the lowest unfinished tile is always
// some workgroup's current one.  Granules: two digits per 8 bytes, n < 2^30.
template <typename K, int RB, int BLOCK, int KPT, bool DBG>
__global__ __launch_bounds__(BLOCK, (RadixSmem<K, RB, BLOCK, KPT, 4>::MIN_WAVES_PER_SIMD))
void radix_scatter_pipe_kernel(const K *__restrict__ in, K *__restrict__ out, u64 n, u32 shift, u32 dmask,
                               const u64 *__restrict__ gbase, u64 *__restrict__ status, u32 *__restrict__ ticket,
                               u32 *__restrict__ error_flag, u64 num_tiles, u64 *__restrict__ dbg) {
  using SM = RadixSmem<K, RB, BLOCK, KPT, 4>;
  using KO = KeyOps<K>;
  constexpr int R = SM::R, NW = SM::NW, TILE = SM::TILE, G = R / 2;
  constexpr int WALK = 8;                               // predecessor granules in flight per walker thread
  static_assert(BLOCK >= R && G % 64 == 0, "one thread per digit; whole waves walk");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  K   *s_keys  = reinterpret_cast<K *>(smem);
  u32 *s_whist = reinterpret_cast<u32 *>(smem);                       // aliases s_keys (see barriers)
  u64 *s_gbase = reinterpret_cast<u64 *>(smem + SM::OFF_GBASE);
  u32 *s_dbase = reinterpret_cast<u32 *>(smem + SM::OFF_DBASE);
  u32 *s_cnt   = reinterpret_cast<u32 *>(smem + SM::OFF_CNT);
  u32 *s_tmp   = reinterpret_cast<u32 *>(smem + SM::OFF_TMP);
  const u32 tid0 = threadIdx.x;
  u32 tid = tid0, lane = tid0 & 63u, w = tid0 >> 6;

  K keys[KPT];
  auto fetch = [&](u64 t) __attribute__((always_inline)) {
    if (t >= num_tiles) return;
    const u64 tile_base = t * (u64)TILE;
    const u64 wave_off  = tile_base + (u64)w * (64 * KPT) + lane;
    const K  *p         = in + wave_off;
    if (tile_base + TILE <= n) {                        // every tile but the last
#pragma unroll
      for (int j = 0; j < KPT; j++) keys[j] = p[j * 64];
    } else {
#pragma unroll
      for (int j = 0; j < KPT; j++) keys[j] = (wave_off + (u64)j * 64 < n) ? p[j * 64] : KO::pad();
    }
  };

  if (tid == 0) s_tmp[32] = atomicAdd(ticket, 1u);
  __syncthreads();
  u64 tile = s_tmp[32];
  fetch(tile);

  u64 ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
#define PK_STAMP(i) do { if (DBG) { const u64 t = __builtin_readcyclecounter(); ph[i] += t - t0; t0 = t; } } while (0)
  while (tile < num_tiles) {
    if (DBG) t0 = __builtin_readcyclecounter();
    // the thread coordinates are laundered once per iteration: otherwise every LDS address of the
    // unrolled body is hoisted out of the loop and the kernel spills
    tid = tid0;
    asm volatile("" : "+v"(tid));
    lane = tid & 63u; w = tid >> 6;
    const u64 lt_mask = (1ull << lane) - 1ull, lane_bit = 1ull << lane;
    const bool walker = tid < (u32)G;                   // whole waves: G is a multiple of 64
    if (tid == 0) s_tmp[33] = atomicAdd(ticket, 1u);    // published by the barrier below
    for (u32 i = tid; i < (u32)(NW * R * 3); i += BLOCK) s_whist[i] = 0;   // counters + match masks
    __syncthreads();
    const u64 next = s_tmp[33];
    PK_STAMP(0);
    const u64 tile_base = tile * (u64)TILE;
    const u32 n_valid   = (tile_base + TILE <= n) ? (u32)TILE : (u32)(n - tile_base);

    // ---- rank inside the wave through wave-private LDS match masks (see radix_scatter_kernel) ----
    u32 ranks[KPT / 2];
    {
      lds_u32 *wh = (lds_u32 *)(smem) + w * R;
      lds_u64 *mk = (lds_u64 *)(smem + (size_t)NW * R * 4) + w * R;
#pragma unroll
      for (int j = 0; j < KPT; j++) {
        const u32 d = KO::digit(keys[j], shift, dmask);
        __hip_atomic_fetch_or(&mk[d], lane_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const u64 peers = __hip_atomic_load(&mk[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const u32 base  = __hip_atomic_load(&wh[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const u32 lower = __popcll(peers & lt_mask);
        if (lower == 0) {
          __hip_atomic_store(&wh[d], base + (u32)__popcll(peers), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          __hip_atomic_store(&mk[d], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        if (j & 1) ranks[j / 2] |= (base + lower) << 16;
        else       ranks[j / 2]  = (base + lower);
      }
    }
    __syncthreads();
    PK_STAMP(1);

    // ---- digit totals of the tile, wave-exclusive bases ----
    u32 count = 0;
    if (tid < (u32)R) {
      u32 acc = 0;
#pragma unroll
      for (int ww = 0; ww < NW; ww++) {
        const u32 t = s_whist[ww * R + tid];
        s_whist[ww * R + tid] = acc;
        acc += t;
      }
      count = acc;
      s_cnt[tid] = (tid == dmask) ? count - ((u32)TILE - n_valid) : count;   // padding is not published
    }
    u32 tile_total;
    const u32 excl = block_excl_scan<BLOCK, u32>(count, s_tmp, &tile_total);
    if (tid < (u32)R) s_dbase[tid] = excl;
    __syncthreads();

    // ---- publish the aggregate, then move the keys to their place in the sorted tile ----
    u64 *mine = status + tile * (u64)G + tid;
    u32 c0 = 0, c1 = 0;
    if (walker) {
      c0 = s_cnt[2 * tid]; c1 = s_cnt[2 * tid + 1];
      const u32 fl = (tile == 0) ? 2u : 1u;
      status_store(mine, st_pack(fl, c0, fl, c1));
    }
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const u32 d = KO::digit(keys[j], shift, dmask);
      const u32 add = s_dbase[d] + s_whist[w * R + d];
      ranks[j / 2] += (j & 1) ? (add << 16) : add;
    }
    __syncthreads();                          // s_whist is dead; its storage becomes s_keys
#pragma unroll
    for (int j = 0; j < KPT; j++) s_keys[(j & 1) ? (ranks[j / 2] >> 16) : (ranks[j / 2] & 0xFFFFu)] = keys[j];
    __syncthreads();                          // keys live in LDS only: the registers take the next tile
    PK_STAMP(2);

    if (!walker) {
      fetch(next);
    } else {
      // flat walk over the predecessors' granules, WALK in flight per round (a two-level scheme with
      // group sums was tried: it needs a third dependent round trip and measured 25% slower)
      u32 p0 = 0, p1 = 0;
      if (tile != 0) {
        bool done = false;
        u64  t = tile - 1;
        u32  spins = 0;
        while (!done) {
          u64 gv[WALK];
#pragma unroll
          for (int i = 0; i < WALK; i++)
            gv[i] = (t >= (u64)i) ? status_load(status + (t - i) * (u64)G + tid) : st_pack(2, 0, 2, 0);
          u32 used = 0;
          bool open = true;
#pragma unroll
          for (int i = 0; i < WALK; i++) {
            const u32 lo = (u32)gv[i], hi = (u32)(gv[i] >> 32);
            const u32 f = lo >> 30;                     // both halves of a granule carry the same flag
            open = open && !done && (f != 0);
            if (open) {
              p0 += lo & 0x3FFFFFFFu; p1 += hi & 0x3FFFFFFFu;
              if (f == 2) done = true;
              used++;
            }
          }
          t -= (used <= t) ? used : t;
          if (used == 0) {
            if (++spins > RS_SPIN_LIMIT) { atomicExch(error_flag, 1u); break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        status_store(mine, st_pack(2, p0 + c0, 2, p1 + c1));
      }
      s_gbase[2 * tid]     = gbase[2 * tid]     + (u64)p0 - (u64)s_dbase[2 * tid];
      s_gbase[2 * tid + 1] = gbase[2 * tid + 1] + (u64)p1 - (u64)s_dbase[2 * tid + 1];
      fetch(next);
    }
    __syncthreads();
    PK_STAMP(3);

    // ---- contiguous runs leave coalesced ----
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const u32 i = (u32)j * BLOCK + tid;
      if (i < n_valid) {
        const K   key = s_keys[i];
        const u32 d   = KO::digit(key, shift, dmask);
        out[s_gbase[d] + (u64)i] = key;
      }
    }
    PK_STAMP(4);
    __syncthreads();                          // s_keys / s_gbase are rewritten by the next iteration
    PK_STAMP(5);
    tile = next;
    if (DBG) ph[7]++;
  }
  if (DBG && tid0 == 0 && blockIdx.x < 64)
    for (int i = 0; i < 8; i++) dbg[blockIdx.x * 8 + i] = ph[i];
#undef PK_STAMP
}

// ---- grouping pass (plan.mode == 3; the sub-bucket finish path) -----------------------
// The finish path does not need a sorted file, only its k-mers GROUPED by their top bits: sub-bucket
// members may come in any order (the LDS finish counts them anyway).  A grouping pass therefore ranks
// a tile with one returning LDS atomic per key -- no wave-private counters, no match masks, a 2 KiB
// histogram to clear instead of 96 KiB -- and only has to respect the TILE order for the bases.
// With two digits (d_hi:d_lo, LSD): pass 1 groups by d_lo with plain tiles; pass 2 must keep the d_lo
// order inside a d_hi group, which holds if no tile of pass 2 mixes two d_lo values: its tiles are cut
// at the d_lo region boundaries of pass 1's output (region table below), every region's last tile
// being partial.  Persistent workgroups, ticket one tile ahead, next tile's keys fetched behind the
// look-back and the write-out, as in radix_scatter_pipe_kernel.
template <typename K, int RB, int BLOCK, int KPT>
struct GroupSmem {
  static constexpr int R = 1 << RB, NW = BLOCK / 64, TILE = BLOCK * KPT;
  static constexpr size_t OFF_HIST  = (size_t)TILE * sizeof(K);       // u32[R]
  static constexpr size_t OFF_GBASE = OFF_HIST + (size_t)R * 4;       // u64[R]
  static constexpr size_t OFF_DBASE = OFF_GBASE + (size_t)R * 8;      // u32[R]
  static constexpr size_t OFF_CNT   = OFF_DBASE + (size_t)R * 4;      // u32[R]
  static constexpr size_t OFF_TMP   = OFF_CNT + (size_t)R * 4;        // u32[64]
  static constexpr size_t OFF_INFO  = OFF_TMP + 64 * 4;               // u64[4]: key base / valid count of current+next tile
  static constexpr size_t BYTES     = OFF_INFO + 4 * 8;
  static constexpr int    WG_PER_CU = (2 * BYTES <= 160 * 1024) ? 2 : 1;
  static constexpr int    MIN_WAVES_PER_SIMD = (WG_PER_CU * BLOCK) / 256;
};

// region table of a second grouping pass: regions = digit groups of the first pass (gbase_prev = its
// exclusive digit bases, RS_MAX_RADIX entries, unused digits sit at n)
__global__ __launch_bounds__(RS_MAX_RADIX)
void group_regions_kernel(const u64 *__restrict__ gbase_prev, u64 n, u32 tile, u64 *__restrict__ region_start,
                          u32 *__restrict__ region_tiles) {
  __shared__ u32 s_tmp[RS_MAX_RADIX / 64 + 1];
  const u32 r = threadIdx.x;
  const u64 a = gbase_prev[r], b = (r + 1 < RS_MAX_RADIX) ? gbase_prev[r + 1] : n;
  const u32 nt = (u32)((b - a + tile - 1) / tile);
  u32 total;
  const u32 e = block_excl_scan<RS_MAX_RADIX, u32>(nt, s_tmp, &total);
  region_start[r] = a;
  region_tiles[r] = e;
  if (r == 0) { region_start[RS_MAX_RADIX] = n; region_tiles[RS_MAX_RADIX] = total; }
}

template <typename K, int RB, int BLOCK, int KPT, bool DBG>
__global__ __launch_bounds__(BLOCK, (GroupSmem<K, RB, BLOCK, KPT>::MIN_WAVES_PER_SIMD))
void radix_group_kernel(const K *__restrict__ in, K *__restrict__ out, u64 n, u32 shift, u32 dmask,
                        const u64 *__restrict__ gbase, u64 *__restrict__ status, u32 *__restrict__ ticket,
                        u32 *__restrict__ error_flag, u64 num_tiles_plain,
                        const u64 *__restrict__ region_start,   // [RS_MAX_RADIX + 1] or nullptr (plain tiles)
                        const u32 *__restrict__ region_tiles,   // [RS_MAX_RADIX + 1] exclusive; last = total tiles
                        u32 /*flags*/, u64 *__restrict__ dbg) {
  using SM = GroupSmem<K, RB, BLOCK, KPT>;
  using KO = KeyOps<K>;
  constexpr int R = SM::R, TILE = SM::TILE, G = R / 2;
  constexpr int WALK = 16;
  static_assert(BLOCK >= RS_MAX_RADIX && G % 64 == 0 && TILE <= 65536, "one thread per region/digit; 16-bit ranks");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  K   *s_keys  = reinterpret_cast<K *>(smem);
  u32 *s_hist  = reinterpret_cast<u32 *>(smem + SM::OFF_HIST);
  u64 *s_gbase = reinterpret_cast<u64 *>(smem + SM::OFF_GBASE);
  u32 *s_dbase = reinterpret_cast<u32 *>(smem + SM::OFF_DBASE);
  u32 *s_cnt   = reinterpret_cast<u32 *>(smem + SM::OFF_CNT);
  u32 *s_tmp   = reinterpret_cast<u32 *>(smem + SM::OFF_TMP);
  u64 *s_info  = reinterpret_cast<u64 *>(smem + SM::OFF_INFO);
  const u32 tid0 = threadIdx.x;
  u32 tid = tid0, lane = tid0 & 63u, w = tid0 >> 6;

  // this thread's slice of the region table stays in registers
  const bool regions = (region_tiles != nullptr);
  u64 rt_lo = 0, rt_hi = 0, rs = 0, re = 0, total_tiles = num_tiles_plain;
  if (regions) {
    if (tid0 < (u32)RS_MAX_RADIX) {
      rt_lo = region_tiles[tid0]; rt_hi = region_tiles[tid0 + 1];
      rs = region_start[tid0];    re = region_start[tid0 + 1];
    }
    total_tiles = region_tiles[RS_MAX_RADIX];
  }
  auto announce = [&](u64 t, int slot) __attribute__((always_inline)) {   // key range of tile t -> s_info[2*slot..]
    if (t >= total_tiles) return;
    if (regions) {
      if (rt_lo <= t && t < rt_hi) {                      // exactly one thread (empty regions own no tile)
        const u64 kb = rs + (t - rt_lo) * (u64)TILE;
        s_info[2 * slot] = kb;
        s_info[2 * slot + 1] = (re - kb < (u64)TILE) ? re - kb : (u64)TILE;
      }
    } else if (tid0 == 0) {
      const u64 kb = t * (u64)TILE;
      s_info[2 * slot] = kb;
      s_info[2 * slot + 1] = (n - kb < (u64)TILE) ? n - kb : (u64)TILE;
    }
  };

  K keys[KPT];
  auto fetch = [&](u64 kb, u32 nv) __attribute__((always_inline)) {
    const u32 li = w * (64 * KPT) + lane;                 // wave-striped: 512 contiguous bytes per wave instruction
    const K  *p  = in + kb + li;
    if (nv == (u32)TILE) {
#pragma unroll
      for (int j = 0; j < KPT; j++) keys[j] = p[j * 64];
    } else {
#pragma unroll
      for (int j = 0; j < KPT; j++) if (li + (u32)j * 64 < nv) keys[j] = p[j * 64];
    }
  };

  if (tid == 0) s_tmp[32] = atomicAdd(ticket, 1u);
  __syncthreads();
  u64 tile = s_tmp[32];
  announce(tile, 0);
  __syncthreads();
  u64 kb = 0; u32 nv = 0;
  if (tile < total_tiles) { kb = s_info[0]; nv = (u32)s_info[1]; fetch(kb, nv); }

  u64 ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
#define PK_STAMP(i) do { if (DBG) { const u64 t = __builtin_readcyclecounter(); ph[i] += t - t0; t0 = t; } } while (0)
  while (tile < total_tiles) {
    if (DBG) t0 = __builtin_readcyclecounter();
    tid = tid0;
    asm volatile("" : "+v"(tid));                         // keeps the unrolled body's LDS addresses out of the loop preheader
    lane = tid & 63u; w = tid >> 6;
    const bool walker = tid < (u32)G;
    if (tid == 0) s_tmp[33] = atomicAdd(ticket, 1u);
    if (tid < (u32)R) s_hist[tid] = 0;
    __syncthreads();                                      // (A)
    const u64 next = s_tmp[33];
    announce(next, 1);                                    // read after (C)
    PK_STAMP(0);

    // ---- rank: position among the tile's keys of the same digit, in arrival order ----
    const u32 li = w * (64 * KPT) + lane;
    u32 ranks[KPT / 2];
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      u32 r = 0;
      if (li + (u32)j * 64 < nv) r = atomicAdd(&s_hist[KO::digit(keys[j], shift, dmask)], 1u);
      if (j & 1) ranks[j / 2] |= r << 16;
      else       ranks[j / 2]  = r;
    }
    __syncthreads();                                      // (B)
    PK_STAMP(1);

    const u32 count = (tid < (u32)R) ? s_hist[tid] : 0u;
    u32 tile_total;
    const u32 excl = block_excl_scan<BLOCK, u32>(count, s_tmp, &tile_total);
    if (tid < (u32)R) { s_cnt[tid] = count; s_dbase[tid] = excl; }
    __syncthreads();                                      // (C)

    u64 *mine = status + tile * (u64)G + tid;
    u32 c0 = 0, c1 = 0;
    if (walker) {
      c0 = s_cnt[2 * tid]; c1 = s_cnt[2 * tid + 1];
      const u32 fl = (tile == 0) ? 2u : 1u;
      status_store(mine, st_pack(fl, c0, fl, c1));
    }
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      if (li + (u32)j * 64 < nv) {
        const u32 d = KO::digit(keys[j], shift, dmask);
        const u32 r = (j & 1) ? (ranks[j / 2] >> 16) : (ranks[j / 2] & 0xFFFFu);
        s_keys[s_dbase[d] + r] = keys[j];
      }
    }
    u64 nkb = 0; u32 nnv = 0;
    if (next < total_tiles) { nkb = s_info[2]; nnv = (u32)s_info[3]; }
    __syncthreads();                                      // (D) keys live in LDS only
    PK_STAMP(2);

    // Look-back.  Persistent workgroups run in step, so the ~256 tiles in flight reach this point together
    // and the inclusive prefixes can only spread from the oldest tile on: a walker that covers WALK
    // predecessors per round trip sees the frontier move ~2*WALK tiles per round trip.  (Measured: holding
    // the key fetch back, or issuing the status loads ahead of it, does not shorten the walk.)
    u32 p0 = 0, p1 = 0;
    bool done = (tile == 0);
    u64  wt = tile ? tile - 1 : 0;
    u32  spins = 0;
    u64  gv[WALK];
    auto issue = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < WALK; i++)
        gv[i] = (wt >= (u64)i) ? status_load(status + (wt - i) * (u64)G + tid) : st_pack(2, 0, 2, 0);
    };
    auto consume = [&]() __attribute__((always_inline)) {
      u32 used = 0;
      bool open = true;
#pragma unroll
      for (int i = 0; i < WALK; i++) {
        const u32 lo = (u32)gv[i], hi = (u32)(gv[i] >> 32);
        const u32 f = lo >> 30;
        open = open && !done && (f != 0);
        if (open) {
          p0 += lo & 0x3FFFFFFFu; p1 += hi & 0x3FFFFFFFu;
          if (f == 2) done = true;
          used++;
        }
      }
      wt -= (used <= wt) ? used : wt;
      if (used == 0) {
        if (++spins > RS_SPIN_LIMIT) { atomicExch(error_flag, 1u); done = true; }
        else __builtin_amdgcn_s_sleep(1);
      }
    };
    if (!walker) {
      if (next < total_tiles) fetch(nkb, nnv);
    } else {
      while (!done) { issue(); consume(); }
      if (tile != 0) status_store(mine, st_pack(2, p0 + c0, 2, p1 + c1));
      s_gbase[2 * tid]     = gbase[2 * tid]     + (u64)p0 - (u64)s_dbase[2 * tid];
      s_gbase[2 * tid + 1] = gbase[2 * tid + 1] + (u64)p1 - (u64)s_dbase[2 * tid + 1];
      if (next < total_tiles) fetch(nkb, nnv);
    }
    __syncthreads();                                      // (E)
    PK_STAMP(3);

#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const u32 i = (u32)j * BLOCK + tid;
      if (i < nv) {
        const K   key = s_keys[i];
        const u32 d   = KO::digit(key, shift, dmask);
        out[s_gbase[d] + (u64)i] = key;
      }
    }
    PK_STAMP(4);
    __syncthreads();                                      // (F)
    PK_STAMP(5);
    tile = next; kb = nkb; nv = nnv;
    if (DBG) ph[7]++;
  }
  if (DBG && tid0 == 0 && blockIdx.x < 64)
    for (int i = 0; i < 8; i++) dbg[blockIdx.x * 8 + i] = ph[i];
#undef PK_STAMP
  (void)kb;
}

// ---- classic mode: per-tile digit histogram + row scan ----------------------
template <typename K, int RB, int BLOCK, int KPT>
__global__ __launch_bounds__(BLOCK)
void radix_tile_hist_kernel(const K *__restrict__ in, u64 n, u32 shift, u32 dmask, u32 *__restrict__ tile_hist,
                            u64 num_tiles) {
  constexpr int R = 1 << RB, TILE = BLOCK * KPT;
  __shared__ u32 s_h[R];
  const u32 tid = threadIdx.x;
  for (u32 i = tid; i < (u32)R; i += BLOCK) s_h[i] = 0;
  __syncthreads();
  const u64 tile_base = (u64)blockIdx.x * TILE;
#pragma unroll
  for (int j = 0; j < KPT; j++) {
    const u64 idx = tile_base + (u64)j * BLOCK + tid;
    if (idx < n) atomicAdd(&s_h[KeyOps<K>::digit(in[idx], shift, dmask)], 1u);
  }
  __syncthreads();
  for (u32 i = tid; i < (u32)R; i += BLOCK) tile_hist[(u64)i * num_tiles + blockIdx.x] = s_h[i];
}

// One workgroup per digit: exclusive scan of its row of tile counts (-> u64),
// and the row total.
__global__ __launch_bounds__(1024)
void radix_row_scan_kernel(const u32 *__restrict__ tile_hist, u64 *__restrict__ tile_offs, u64 *__restrict__ row_total,
                           u64 num_tiles) {
  __shared__ u64 s_tmp[1024 / 64 + 1];
  const u64 row = (u64)blockIdx.x * num_tiles;
  u64 carry = 0;
  for (u64 c = 0; c < num_tiles; c += 1024) {
    const u64 t = c + threadIdx.x;
    const u64 v = (t < num_tiles) ? (u64)tile_hist[row + t] : 0ull;
    u64 tot;
    const u64 e = block_excl_scan<1024, u64>(v, s_tmp, &tot);
    if (t < num_tiles) tile_offs[row + t] = carry + e;
    carry += tot;
  }
  if (threadIdx.x == 0) row_total[blockIdx.x] = carry;
}

// Adds the exclusive digit base (scan of row totals) to every row.
__global__ __launch_bounds__(256)
void radix_row_add_kernel(u64 *__restrict__ tile_offs, const u64 *__restrict__ row_total, u32 R, u64 num_tiles) {
  __shared__ u64 s_part[4];
  const u32 d = blockIdx.y;
  u64 part = 0;
  for (u32 i = threadIdx.x; i < d; i += 256) part += row_total[i];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) part += __shfl_down(part, o);
  if (lane_id() == 0) s_part[wave_id()] = part;
  __syncthreads();
  const u64 base = s_part[0] + s_part[1] + s_part[2] + s_part[3];
  const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
  if (t < num_tiles) tile_offs[(u64)d * num_tiles + t] += base;
  (void)R;
}

// ---- host side ---------------------------------------------------------------

static int env_int(const char *name, int dflt) {
  const char *s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}

void make_sort_plan(uint32_t begin_bit, uint32_t end_bit, SortPlan *plan) {
  memset(plan, 0, sizeof(*plan));
  // defaults = the fastest measured combination on MI355X (profiles/r01 notes in DESIGN.md):
  // 9-bit digits, one 1024-thread workgroup per CU with 16 keys per thread (16384-key tiles),
  // LDS-mask ranking, window look-back after the exchange
  int rb    = env_int("MGC_RADIX_BITS", 9);
  int mode  = env_int("MGC_SORT_MODE", 0);
  int kpt   = env_int("MGC_SORT_KPT", 16);
  int block = env_int("MGC_SORT_BLOCK", 1024);
  if (rb != 8 && rb != 9) rb = 8;
  if (kpt != 8 && kpt != 16) kpt = 16;
  if (block != 512 && block != 1024) block = 512;
  plan->radix_bits = (uint32_t)rb;
  plan->block      = (uint32_t)block;
  plan->kpt        = (uint32_t)kpt;
  plan->tile       = plan->block * plan->kpt;
  plan->mode       = (mode == 1 || mode == 3) ? (uint32_t)mode : 0u;   // 0 look-back, 1 classic, 3 grouping (finish path only)
  plan->match      = env_int("MGC_SORT_MATCH", 1) ? 1u : 0u;
  { const int lb = env_int("MGC_SORT_LB", 2); plan->lookback = (lb == 2 || lb == 5) ? (uint32_t)lb : 1u; }
  plan->flags      = (uint32_t)env_int("MGC_SORT_FLAGS", 0);
  const uint32_t nbits = (end_bit > begin_bit) ? end_bit - begin_bit : 0;
  uint32_t passes = (nbits + rb - 1) / rb;
  if (passes > RS_MAX_PASSES) passes = RS_MAX_PASSES;
  plan->num_passes = passes;
  uint32_t bit = begin_bit;
  for (uint32_t p = 0; p < passes; p++) {
    uint32_t b = nbits / passes + ((p < nbits % passes) ? 1u : 0u);
    plan->pass_shift[p] = bit;
    plan->pass_bits[p]  = b;
    bit += b;
  }
}

static inline uint64_t max_tiles_for(uint64_t n) { return (n + 4095) / 4096 + 1; }   // tile >= 4096 keys

size_t sort_workspace_bytes(uint64_t n) {
  // header + the larger of {look-back granules (one pass at a time), classic tile_hist + tile_offs}
  const uint64_t t = max_tiles_for(n);
  // + grouping mode: up to RS_MAX_RADIX + 1 extra (partial) tiles of granules and the region table
  return sizeof(SortHeader) + 1024 + (size_t)t * RS_MAX_RADIX * (sizeof(uint32_t) + sizeof(uint64_t)) +
         (size_t)(RS_MAX_RADIX + 2) * (RS_MAX_RADIX / 2) * sizeof(uint64_t) + (size_t)(RS_MAX_RADIX + 2) * 16;
}

static int device_cu_count() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
  }
  return cus;
}

template <typename K, int RB, int BLOCK, int KPT, int MATCH, int LBK>
static hipError_t run_passes(void *d_keys, void *d_alt, uint64_t n, const SortPlan &plan, void *d_ws,
                             uint32_t *d_error, int *result_in_alt, hipStream_t st, hipEvent_t *pass_events) {
  constexpr int LBO = (LBK == 5) ? 2 : LBK;             // look-back flavour of the non-pipelined kernel
  using SM  = RadixSmem<K, RB, BLOCK, KPT, LBO>;
  using SM0 = RadixSmem<K, RB, BLOCK, KPT, 0>;
  constexpr int R = 1 << RB, TILE = BLOCK * KPT;
  SortHeader *hdr = reinterpret_cast<SortHeader *>(d_ws);
  unsigned char *body = reinterpret_cast<unsigned char *>(d_ws) + ((sizeof(SortHeader) + 255) / 256) * 256;
  const uint64_t num_tiles = (n + TILE - 1) / TILE;

  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_scatter_kernel<K, RB, BLOCK, KPT, LBO, MATCH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)SM::BYTES);
    if constexpr (LBK == 5)
    {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_scatter_pipe_kernel<K, RB, BLOCK, KPT, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)RadixSmem<K, RB, BLOCK, KPT, 4>::BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_scatter_pipe_kernel<K, RB, BLOCK, KPT, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)RadixSmem<K, RB, BLOCK, KPT, 4>::BYTES);
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_scatter_kernel<K, RB, BLOCK, KPT, 0, MATCH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)SM0::BYTES);
    attr_done = true;
  }

  K *src = reinterpret_cast<K *>(d_keys), *dst = reinterpret_cast<K *>(d_alt);
  int in_alt = 0;
  const bool lookback = (plan.mode == 0);

  if (lookback) {
    u64 *status = reinterpret_cast<u64 *>(body);
    const size_t status_bytes = (size_t)num_tiles * (LBO == 3 ? R : R / 2) * sizeof(u64);
    MGC_CHECK(hipMemsetAsync(hdr, 0, sizeof(SortHeader), st));
    PassList pl;
    pl.n = plan.num_passes;
    for (uint32_t p = 0; p < plan.num_passes; p++) {
      pl.shift[p] = plan.pass_shift[p];
      pl.mask[p]  = (1u << plan.pass_bits[p]) - 1u;
    }
    launch_radix_hist<K>((const K *)src, (u64)n, pl, &hdr->ghist[0][0], st);
    MGC_CHECK(hipGetLastError());
    hipLaunchKernelGGL(radix_digit_scan_kernel, dim3(plan.num_passes), dim3(RS_MAX_RADIX), 0, st,
                       &hdr->ghist[0][0], &hdr->gbase[0][0]);
    MGC_CHECK(hipGetLastError());
    for (uint32_t p = 0; p < plan.num_passes; p++) {
      MGC_CHECK(hipMemsetAsync(status, 0, status_bytes, st));
      if (pass_events) MGC_CHECK(hipEventRecord(pass_events[2 * p], st));
      if constexpr (LBK == 5) {
        using SMP = RadixSmem<K, RB, BLOCK, KPT, 4>;
        const uint64_t resident = (uint64_t)device_cu_count() * SMP::WG_PER_CU;
        const uint32_t pgrid = (uint32_t)(num_tiles < resident ? num_tiles : resident);
        if (plan.dbg)
          hipLaunchKernelGGL((radix_scatter_pipe_kernel<K, RB, BLOCK, KPT, true>), dim3(pgrid), dim3(BLOCK), SMP::BYTES, st,
                             (const K *)src, dst, (u64)n, plan.pass_shift[p], (1u << plan.pass_bits[p]) - 1u,
                             &hdr->gbase[p][0], status, &hdr->ticket[p], d_error, (u64)num_tiles, reinterpret_cast<u64 *>(plan.dbg));
        else
          hipLaunchKernelGGL((radix_scatter_pipe_kernel<K, RB, BLOCK, KPT, false>), dim3(pgrid), dim3(BLOCK), SMP::BYTES, st,
                             (const K *)src, dst, (u64)n, plan.pass_shift[p], (1u << plan.pass_bits[p]) - 1u,
                             &hdr->gbase[p][0], status, &hdr->ticket[p], d_error, (u64)num_tiles, (u64 *)nullptr);
      } else {
        hipLaunchKernelGGL((radix_scatter_kernel<K, RB, BLOCK, KPT, LBO, MATCH>), dim3((uint32_t)num_tiles),
                           dim3(BLOCK), SM::BYTES, st, (const K *)src, dst, (u64)n, plan.pass_shift[p],
                           (1u << plan.pass_bits[p]) - 1u, &hdr->gbase[p][0], status, &hdr->ticket[p], d_error, plan.flags,
                           (const u64 *)nullptr, (u64)num_tiles, reinterpret_cast<u64 *>(plan.dbg));
      }
      MGC_CHECK(hipGetLastError());
      if (pass_events) MGC_CHECK(hipEventRecord(pass_events[2 * p + 1], st));
      K *t = src; src = dst; dst = t; in_alt ^= 1;
    }
  } else if (plan.mode == 3) {
    // ---- grouping passes (finish path): see radix_group_kernel ----
    using GS = GroupSmem<K, RB, BLOCK, KPT>;
    static bool gattr_done = false;
    if (!gattr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel<K, RB, BLOCK, KPT, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS::BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel<K, RB, BLOCK, KPT, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS::BYTES);
      gattr_done = true;
    }
    const uint64_t max_tiles = num_tiles + RS_MAX_RADIX + 1;          // region-aligned tiles: one partial tile per region
    u64 *status = reinterpret_cast<u64 *>(body);
    const size_t status_bytes = (size_t)max_tiles * (R / 2) * sizeof(u64);
    u64 *region_start = reinterpret_cast<u64 *>(body + ((status_bytes + 255) / 256) * 256);
    u32 *region_tiles = reinterpret_cast<u32 *>(region_start + RS_MAX_RADIX + 1);
    MGC_CHECK(hipMemsetAsync(hdr, 0, sizeof(SortHeader), st));
    PassList pl;
    pl.n = plan.num_passes;
    for (uint32_t p = 0; p < plan.num_passes; p++) {
      pl.shift[p] = plan.pass_shift[p];
      pl.mask[p]  = (1u << plan.pass_bits[p]) - 1u;
    }
    launch_radix_hist<K>((const K *)src, (u64)n, pl, &hdr->ghist[0][0], st);
    MGC_CHECK(hipGetLastError());
    hipLaunchKernelGGL(radix_digit_scan_kernel, dim3(plan.num_passes), dim3(RS_MAX_RADIX), 0, st,
                       &hdr->ghist[0][0], &hdr->gbase[0][0]);
    MGC_CHECK(hipGetLastError());
    const uint64_t resident = (uint64_t)device_cu_count() * GS::WG_PER_CU;
    for (uint32_t p = 0; p < plan.num_passes; p++) {
      if (p == 1) {
        hipLaunchKernelGGL(group_regions_kernel, dim3(1), dim3(RS_MAX_RADIX), 0, st, &hdr->gbase[0][0], (u64)n, (u32)TILE,
                           region_start, region_tiles);
        MGC_CHECK(hipGetLastError());
      }
      MGC_CHECK(hipMemsetAsync(status, 0, status_bytes, st));
      if (pass_events) MGC_CHECK(hipEventRecord(pass_events[2 * p], st));
      const uint64_t tiles_bound = (p == 0) ? num_tiles : max_tiles;
      const uint32_t pgrid = (uint32_t)(tiles_bound < resident ? tiles_bound : resident);
      const u64 *rs = (p == 0) ? nullptr : region_start;
      const u32 *rt = (p == 0) ? nullptr : region_tiles;
      if (plan.dbg)
        hipLaunchKernelGGL((radix_group_kernel<K, RB, BLOCK, KPT, true>), dim3(pgrid), dim3(BLOCK), GS::BYTES, st,
                           (const K *)src, dst, (u64)n, plan.pass_shift[p], (1u << plan.pass_bits[p]) - 1u,
                           &hdr->gbase[p][0], status, &hdr->ticket[p], d_error, (u64)num_tiles, rs, rt, plan.flags,
                           reinterpret_cast<u64 *>(plan.dbg));
      else
        hipLaunchKernelGGL((radix_group_kernel<K, RB, BLOCK, KPT, false>), dim3(pgrid), dim3(BLOCK), GS::BYTES, st,
                           (const K *)src, dst, (u64)n, plan.pass_shift[p], (1u << plan.pass_bits[p]) - 1u,
                           &hdr->gbase[p][0], status, &hdr->ticket[p], d_error, (u64)num_tiles, rs, rt, plan.flags, (u64 *)nullptr);
      MGC_CHECK(hipGetLastError());
      if (pass_events) MGC_CHECK(hipEventRecord(pass_events[2 * p + 1], st));
      K *t = src; src = dst; dst = t; in_alt ^= 1;
    }
  } else {
    // ---- classic: histogram / scan / scatter per pass ----
    u32 *tile_hist = reinterpret_cast<u32 *>(body);
    u64 *tile_offs = reinterpret_cast<u64 *>(body + (((size_t)num_tiles * R * sizeof(u32) + 255) / 256) * 256);
    for (uint32_t p = 0; p < plan.num_passes; p++) {
      const uint32_t shift = plan.pass_shift[p], dmask = (1u << plan.pass_bits[p]) - 1u;
      hipLaunchKernelGGL((radix_tile_hist_kernel<K, RB, BLOCK, KPT>), dim3((uint32_t)num_tiles), dim3(BLOCK), 0, st,
                         (const K *)src, (u64)n, shift, dmask, tile_hist, (u64)num_tiles);
      MGC_CHECK(hipGetLastError());
      hipLaunchKernelGGL(radix_row_scan_kernel, dim3(R), dim3(1024), 0, st, tile_hist, tile_offs,
                         &hdr->row_total[0], (u64)num_tiles);
      MGC_CHECK(hipGetLastError());
      hipLaunchKernelGGL(radix_row_add_kernel, dim3((uint32_t)((num_tiles + 255) / 256), R), dim3(256), 0, st,
                         tile_offs, &hdr->row_total[0], (u32)R, (u64)num_tiles);
      MGC_CHECK(hipGetLastError());
      if (pass_events) MGC_CHECK(hipEventRecord(pass_events[2 * p], st));
      hipLaunchKernelGGL((radix_scatter_kernel<K, RB, BLOCK, KPT, 0, MATCH>), dim3((uint32_t)num_tiles), dim3(BLOCK),
                         SM0::BYTES, st, (const K *)src, dst, (u64)n, shift, dmask, (const u64 *)nullptr, (u64 *)nullptr,
                         (u32 *)nullptr, d_error, plan.flags, tile_offs, (u64)num_tiles, reinterpret_cast<u64 *>(plan.dbg));
      MGC_CHECK(hipGetLastError());
      if (pass_events) MGC_CHECK(hipEventRecord(pass_events[2 * p + 1], st));
      K *t = src; src = dst; dst = t; in_alt ^= 1;
    }
  }
  *result_in_alt = in_alt;
  return hipSuccess;
}

hipError_t launch_radix_sort(void *d_keys, void *d_alt, uint64_t n, uint32_t key_words, const SortPlan &plan,
                             void *d_ws, size_t ws_bytes, uint32_t *d_error, int *result_in_alt,
                             hipStream_t st, hipEvent_t *pass_events) {
  *result_in_alt = 0;
  if (n == 0 || plan.num_passes == 0) return hipSuccess;
  if (ws_bytes < sort_workspace_bytes(n)) return hipErrorInvalidValue;
  if (plan.mode == 3 && (plan.num_passes > 2 || n >= (1ull << 30))) {
    SortPlan stable = plan;                 // grouping is defined for one or two digits and 30-bit granule values
    stable.mode = 0;
    return launch_radix_sort(d_keys, d_alt, n, key_words, stable, d_ws, ws_bytes, d_error, result_in_alt, st, pass_events);
  }
#define MGC_RUN(K_, RB_, BLOCK_, KPT_)                                                                       \
  do {                                                                                                       \
    if (n >= (1ull << 30))   /* packed look-back granules hold 30-bit values: use the wide ones */           \
      return run_passes<K_, RB_, BLOCK_, KPT_, 1, 3>(d_keys, d_alt, n, plan, d_ws, d_error, result_in_alt, st, pass_events); \
    if (plan.match == 0)                                                                                     \
      return run_passes<K_, RB_, BLOCK_, KPT_, 0, 1>(d_keys, d_alt, n, plan, d_ws, d_error, result_in_alt, st, pass_events); \
    if (plan.lookback == 2)                                                                                  \
      return run_passes<K_, RB_, BLOCK_, KPT_, 1, 2>(d_keys, d_alt, n, plan, d_ws, d_error, result_in_alt, st, pass_events); \
    if (plan.lookback == 5)                                                                                  \
      return run_passes<K_, RB_, BLOCK_, KPT_, 1, 5>(d_keys, d_alt, n, plan, d_ws, d_error, result_in_alt, st, pass_events); \
    return run_passes<K_, RB_, BLOCK_, KPT_, 1, 1>(d_keys, d_alt, n, plan, d_ws, d_error, result_in_alt, st, pass_events);   \
  } while (0)
  if (key_words == 2) {
    // 128-bit keys: 8 keys per thread keep the tile at 128 KiB of LDS (one workgroup per CU)
    if (plan.radix_bits == 9) MGC_RUN(K128, 9, 1024, 8);
    MGC_RUN(K128, 8, 1024, 8);
  }
  if (plan.radix_bits == 9) {
    if (plan.block == 1024 && plan.kpt == 8) MGC_RUN(u64, 9, 1024, 8);
    if (plan.block == 1024) MGC_RUN(u64, 9, 1024, 16);
    if (plan.kpt == 8) MGC_RUN(u64, 9, 512, 8);
    MGC_RUN(u64, 9, 512, 16);
  }
  if (plan.block == 1024 && plan.kpt == 8) MGC_RUN(u64, 8, 1024, 8);
  if (plan.block == 1024) MGC_RUN(u64, 8, 1024, 16);
  if (plan.kpt == 8) MGC_RUN(u64, 8, 512, 8);
  MGC_RUN(u64, 8, 512, 16);
#undef MGC_RUN
}

// ============================================================================
//  Multi-block scans over uint64 arrays (in place)
//    forward exclusive sum   : a[i] <- sum_{j<i} a[j]          (total returned in *total)
//    reverse inclusive min   : a[i] <- min(a[i], a[i+1], ..., a[n-1], init)
//  Three phases per level (local scan + block totals, recurse on the totals, add back);
//  4096 entries per workgroup, so two levels cover 16 M entries.
// ============================================================================
constexpr int SC_BLOCK = 1024;
constexpr int SC_ITEMS = 4;
constexpr int SC_CHUNK = SC_BLOCK * SC_ITEMS;

template <bool MIN_REVERSE>
__global__ __launch_bounds__(SC_BLOCK)
void scan_local_kernel(u64 *__restrict__ a, u64 n, u64 *__restrict__ block_tot) {
  __shared__ u64 s_tmp[SC_BLOCK / 64 + 1];
  const u64 chunk0 = (u64)blockIdx.x * SC_CHUNK;
  u64 v[SC_ITEMS];
  if (!MIN_REVERSE) {
    u64 sum = 0;
#pragma unroll
    for (int q = 0; q < SC_ITEMS; q++) {
      const u64 i = chunk0 + (u64)threadIdx.x * SC_ITEMS + q;
      v[q] = (i < n) ? a[i] : 0ull;
      sum += v[q];
    }
    u64 tot;
    u64 run = block_excl_scan<SC_BLOCK, u64>(sum, s_tmp, &tot);
#pragma unroll
    for (int q = 0; q < SC_ITEMS; q++) {
      const u64 i = chunk0 + (u64)threadIdx.x * SC_ITEMS + q;
      if (i < n) a[i] = run;
      run += v[q];
    }
    if (threadIdx.x == 0) block_tot[blockIdx.x] = tot;
  } else {
    // thread t owns entries (reversed inside the chunk) so that a forward min-scan over t is a suffix min
    const u32 rt = SC_BLOCK - 1 - threadIdx.x;
    u64 m = ~0ull;
#pragma unroll
    for (int q = SC_ITEMS - 1; q >= 0; q--) {
      const u64 i = chunk0 + (u64)rt * SC_ITEMS + q;
      v[q] = (i < n) ? a[i] : ~0ull;
      m = (v[q] < m) ? v[q] : m;
      v[q] = m;                                       // suffix min inside the thread's items
    }
    // inclusive min-scan across threads (thread 0 holds the LAST items of the chunk)
    const u32 lane = lane_id(), w = wave_id();
    u64 x = m;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const u64 y = __shfl_up(x, d);
      if ((int)lane >= d) x = (y < x) ? y : x;
    }
    __syncthreads();
    if (lane == 63) s_tmp[w] = x;
    __syncthreads();
    u64 pre = ~0ull;                                  // min over earlier threads (= later entries)
    for (u32 i = 0; i < w; i++) pre = (s_tmp[i] < pre) ? s_tmp[i] : pre;
    const u64 y = __shfl_up(x, 1);
    u64 before = (lane == 0) ? pre : ((y < pre) ? y : pre);
#pragma unroll
    for (int q = 0; q < SC_ITEMS; q++) {
      const u64 i = chunk0 + (u64)rt * SC_ITEMS + q;
      const u64 r = (v[q] < before) ? v[q] : before;
      if (i < n) a[i] = r;
    }
    if (threadIdx.x == SC_BLOCK - 1) {                // owner of the chunk's first entries: chunk minimum
      const u64 tot = (x < pre) ? x : pre;
      block_tot[blockIdx.x] = tot;
    }
  }
}

template <bool MIN_REVERSE>
__global__ __launch_bounds__(SC_BLOCK)
void scan_add_kernel(u64 *__restrict__ a, u64 n, const u64 *__restrict__ block_tot, u64 nblocks, u64 init) {
  // forward sum : add the exclusive prefix of the block totals (block_tot already scanned)
  // reverse min : combine with the suffix min of the LATER blocks' minima (block_tot already scanned) and init
  const u64 b = blockIdx.x;
  u64 carry;
  if (!MIN_REVERSE) carry = block_tot[b];
  else              carry = (b + 1 < nblocks) ? ((block_tot[b + 1] < init) ? block_tot[b + 1] : init) : init;
#pragma unroll
  for (int q = 0; q < SC_ITEMS; q++) {
    const u64 i = b * SC_CHUNK + (u64)threadIdx.x * SC_ITEMS + q;
    if (i < n) {
      if (!MIN_REVERSE) a[i] += carry;
      else { const u64 x = a[i]; a[i] = (x < carry) ? x : carry; }
    }
  }
}

__global__ void scan_store_total_kernel(const u64 *__restrict__ last_tot, u64 *__restrict__ total) { *total = *last_tot; }

// scratch needs (n/4096 + n/4096^2 + 8) uint64
static size_t scan_scratch_elems(uint64_t n) {
  uint64_t e = 8, m = n;
  while (m > 1) { m = (m + SC_CHUNK - 1) / SC_CHUNK; e += m + 1; if (m == 1) break; }
  return (size_t)e + 8;
}

template <bool MIN_REVERSE>
static hipError_t scan_u64_inplace(u64 *a, uint64_t n, u64 *scratch, u64 *d_total /* may be null */, u64 init,
                                   hipStream_t st) {
  if (n == 0) {
    if (d_total) return hipMemsetAsync(d_total, 0, sizeof(u64), st);
    return hipSuccess;
  }
  const uint64_t nblocks = (n + SC_CHUNK - 1) / SC_CHUNK;
  hipLaunchKernelGGL(scan_local_kernel<MIN_REVERSE>, dim3((uint32_t)nblocks), dim3(SC_BLOCK), 0, st, a, (u64)n, scratch);
  MGC_CHECK(hipGetLastError());
  if (nblocks > 1 || MIN_REVERSE) {
    // level 2: scan the block totals (forward: exclusive sum; reverse: suffix min), then fold them back
    if (!MIN_REVERSE) {
      MGC_CHECK(scan_u64_inplace<false>(scratch, nblocks, scratch + nblocks + 1, d_total, 0, st));
    } else if (nblocks > 1) {
      MGC_CHECK(scan_u64_inplace<true>(scratch, nblocks, scratch + nblocks + 1, nullptr, ~0ull, st));
    }
    hipLaunchKernelGGL(scan_add_kernel<MIN_REVERSE>, dim3((uint32_t)nblocks), dim3(SC_BLOCK), 0, st, a, (u64)n,
                       (const u64 *)scratch, (u64)nblocks, init);
    MGC_CHECK(hipGetLastError());
  } else {
    if (!MIN_REVERSE) {
      if (d_total) { hipLaunchKernelGGL(scan_store_total_kernel, dim3(1), dim3(1), 0, st, (const u64 *)scratch, d_total); MGC_CHECK(hipGetLastError()); }
    }
  }
  return hipSuccess;
}

// ============================================================================
//  Run-length count of sorted keys
// ============================================================================

constexpr int RL_BLOCK = 256;
constexpr u64 RL_INF   = ~0ull;
template <typename K> struct RlTile { static constexpr int KPT = 16; };   // 4096 keys per tile
template <> struct RlTile<K128>     { static constexpr int KPT = 8;  };   // 2048 (static LDS stays < 64 KiB)

// workspace: [0] total distinct, [8..): tile_offs u64[T+1], tile_next u64[T+1], scan scratch
struct RleWs {
  u64 *total, *tile_offs, *tile_next, *scratch;
  u64  num_tiles;
};
static inline RleWs rle_ws(void *d_ws, uint64_t n, uint32_t key_words) {
  const uint64_t tile = (uint64_t)RL_BLOCK * (key_words == 2 ? RlTile<K128>::KPT : RlTile<u64>::KPT);
  RleWs w;
  w.num_tiles = (n + tile - 1) / tile;
  w.total     = reinterpret_cast<u64 *>(d_ws);
  w.tile_offs = w.total + 8;
  w.tile_next = w.tile_offs + w.num_tiles + 1;
  w.scratch   = w.tile_next + w.num_tiles + 1;
  return w;
}
size_t rle_workspace_bytes(uint64_t n) {
  const uint64_t t = (n + 2047) / 2048;                // smallest tile in use
  return (size_t)(8 + 2 * (t + 1) + scan_scratch_elems(t + 1)) * sizeof(uint64_t);
}

// per tile: number of run heads and position of the first head
template <typename K>
__global__ __launch_bounds__(RL_BLOCK)
void rle_count_kernel(const K *__restrict__ in, u64 n, u64 *__restrict__ tile_cnt, u64 *__restrict__ tile_first) {
  constexpr int KPT = RlTile<K>::KPT, TILE = RL_BLOCK * KPT;
  __shared__ u32 s_cnt[RL_BLOCK / 64];
  __shared__ u64 s_min[RL_BLOCK / 64];
  const u64 tile_base = (u64)blockIdx.x * TILE;
  u32 c = 0;
  u64 first = RL_INF;
#pragma unroll
  for (int j = 0; j < KPT; j++) {
    const u64 idx = tile_base + (u64)j * RL_BLOCK + threadIdx.x;
    if (idx < n) {
      const K key = in[idx];
      const bool head = (idx == 0) || KeyOps<K>::ne(in[idx - 1], key);
      if (head) { c++; if (idx < first) first = idx; }
    }
  }
  // wave reduce
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    c += __shfl_down(c, d);
    const u64 o = __shfl_down(first, d);
    first = (o < first) ? o : first;
  }
  if (lane_id() == 0) { s_cnt[wave_id()] = c; s_min[wave_id()] = first; }
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 tc = 0; u64 tf = RL_INF;
    for (int i = 0; i < RL_BLOCK / 64; i++) { tc += s_cnt[i]; tf = (s_min[i] < tf) ? s_min[i] : tf; }
    tile_cnt[blockIdx.x]   = tc;
    tile_first[blockIdx.x] = tf;
  }
}

__global__ void rle_set_tail_kernel(u64 *__restrict__ tile_offs, u64 *__restrict__ tile_next, u64 num_tiles, u64 n,
                                    const u64 *__restrict__ total) {
  tile_offs[num_tiles] = *total;
  tile_next[num_tiles] = n;
}

__device__ __forceinline__ u32 rl_pad(u32 i) { return i + (i >> 4); }

template <typename K>
__global__ __launch_bounds__(RL_BLOCK)
void rle_emit_kernel(const K *__restrict__ in, u64 n, const u64 *__restrict__ tile_offs,
                     const u64 *__restrict__ tile_next, const u64 *__restrict__ d_out_base, K *__restrict__ out_keys,
                     u32 *__restrict__ out_counts) {
  const u64 out_base0 = d_out_base ? *d_out_base : 0ull;
  constexpr int RL_KPT = RlTile<K>::KPT, RL_TILE = RL_BLOCK * RL_KPT;
  __shared__ K s_keys[RL_TILE + 1 + (RL_TILE + 1) / 16 + 1];
  __shared__ u32 s_tmp[RL_BLOCK / 64 + 1];
  __shared__ u64 s_wmin[RL_BLOCK / 64];
  const u32 tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const u64 tile_base = (u64)blockIdx.x * RL_TILE;

  // coalesced load; logical slot 0 holds the key preceding the tile
  if (tid == 0) s_keys[rl_pad(0)] = (tile_base > 0) ? in[tile_base - 1] : KeyOps<K>::zero();
#pragma unroll
  for (int j = 0; j < RL_KPT; j++) {
    const u32 i   = (u32)j * RL_BLOCK + tid;
    const u64 idx = tile_base + i;
    s_keys[rl_pad(i + 1)] = (idx < n) ? in[idx] : KeyOps<K>::zero();
  }
  __syncthreads();

  // blocked: thread owns RL_KPT consecutive keys
  K keys[RL_KPT];
  u32 flags = 0;
  K prev = s_keys[rl_pad(tid * RL_KPT)];
#pragma unroll
  for (int j = 0; j < RL_KPT; j++) {
    const u32 i   = tid * RL_KPT + j;
    const u64 idx = tile_base + i;
    keys[j] = s_keys[rl_pad(i + 1)];
    const bool head = (idx < n) && ((idx == 0) || KeyOps<K>::ne(keys[j], prev));
    flags |= (head ? 1u : 0u) << j;
    prev = keys[j];
  }

  const u32 c = __popc(flags);
  u32 tile_heads;
  const u32 slot0 = block_excl_scan<RL_BLOCK, u32>(c, s_tmp, &tile_heads);

  // position of the next head after this thread's keys
  const u64 fh = flags ? (tile_base + (u64)tid * RL_KPT + (u32)(__ffs(flags) - 1)) : RL_INF;
  u64 x = fh;                                   // inclusive suffix-min inside the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u64 y = __shfl_down(x, d);
    if ((int)lane + d < 64) x = (y < x) ? y : x;
  }
  if (lane == 0) s_wmin[w] = x;
  __syncthreads();
  u64 after = tile_next[blockIdx.x + 1];        // first head in any later tile (or n)
  for (int ww = RL_BLOCK / 64 - 1; ww > (int)w; ww--) after = (s_wmin[ww] < after) ? s_wmin[ww] : after;
  u64 e = __shfl_down(x, 1);
  if (lane == 63) e = RL_INF;
  u64 next = (e < after) ? e : after;

  const u64 out_base = out_base0 + tile_offs[blockIdx.x] + slot0;
#pragma unroll
  for (int j = RL_KPT - 1; j >= 0; j--) {
    if ((flags >> j) & 1u) {
      const u64 idx  = tile_base + (u64)tid * RL_KPT + j;
      const u64 slot = out_base + __popc(flags & ((1u << j) - 1u));
      out_keys[slot]   = keys[j];
      out_counts[slot] = (u32)(next - idx);     // wraps mod 2^32 like the reference's uint32 ++
      next = idx;
    }
  }
}

hipError_t launch_rle_count(const void *d_sorted, uint64_t n, uint32_t key_words, void *d_ws, hipStream_t st) {
  RleWs w = rle_ws(d_ws, n, key_words);
  if (n == 0) return hipMemsetAsync(w.total, 0, sizeof(u64), st);
  if (key_words == 2)
    hipLaunchKernelGGL(rle_count_kernel<K128>, dim3((uint32_t)w.num_tiles), dim3(RL_BLOCK), 0, st,
                       reinterpret_cast<const K128 *>(d_sorted), (u64)n, w.tile_offs, w.tile_next);
  else
    hipLaunchKernelGGL(rle_count_kernel<u64>, dim3((uint32_t)w.num_tiles), dim3(RL_BLOCK), 0, st,
                       reinterpret_cast<const u64 *>(d_sorted), (u64)n, w.tile_offs, w.tile_next);
  MGC_CHECK(hipGetLastError());
  // tile_offs: exclusive sum of the per-tile head counts; tile_next: first head at or after each tile (n past the end)
  MGC_CHECK(scan_u64_inplace<false>(w.tile_offs, w.num_tiles, w.scratch, w.total, 0, st));
  MGC_CHECK(scan_u64_inplace<true>(w.tile_next, w.num_tiles, w.scratch, nullptr, (u64)n, st));
  hipLaunchKernelGGL(rle_set_tail_kernel, dim3(1), dim3(1), 0, st, w.tile_offs, w.tile_next, (u64)w.num_tiles, (u64)n,
                     (const u64 *)w.total);
  return hipGetLastError();
}

hipError_t rle_read_total(const void *d_ws, uint64_t *n_distinct, hipStream_t st) {
  MGC_CHECK(hipMemcpyAsync(n_distinct, d_ws, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  return hipStreamSynchronize(st);
}

hipError_t launch_rle_emit(const void *d_sorted, uint64_t n, uint32_t key_words, void *d_ws, void *d_unique,
                           uint32_t *d_counts, hipStream_t st, const uint64_t *d_out_base) {
  if (n == 0) return hipSuccess;
  RleWs w = rle_ws(d_ws, n, key_words);
  if (key_words == 2)
    hipLaunchKernelGGL(rle_emit_kernel<K128>, dim3((uint32_t)w.num_tiles), dim3(RL_BLOCK), 0, st,
                       reinterpret_cast<const K128 *>(d_sorted), (u64)n, w.tile_offs, w.tile_next,
                       reinterpret_cast<const u64 *>(d_out_base), reinterpret_cast<K128 *>(d_unique), d_counts);
  else
    hipLaunchKernelGGL(rle_emit_kernel<u64>, dim3((uint32_t)w.num_tiles), dim3(RL_BLOCK), 0, st,
                       reinterpret_cast<const u64 *>(d_sorted), (u64)n, w.tile_offs, w.tile_next,
                       reinterpret_cast<const u64 *>(d_out_base), reinterpret_cast<u64 *>(d_unique), d_counts);
  return hipGetLastError();
}

// ============================================================================
//  Sub-bucket finish: LDS sort of the low bits + fused run-length count
// ============================================================================
//
// After the global LSB passes have ordered a file by its TOP t bits (below the six
// file bits), every value of those bits is a contiguous sub-bucket of a few thousand
// k-mers.  One workgroup loads one sub-bucket, sorts the remaining low bits with
// stable 8-bit counting passes that never leave LDS, run-length counts it, and
// writes the distinct k-mers in place (front of the sub-bucket's own region) plus
// their counts to a scratch array.  A small compaction then packs all sub-buckets.
// HBM traffic: 8 B read per instance + 12 B per distinct k-mer, instead of two more
// 16 B/key radix passes, an 8 B histogram read and the two run-length passes.

// starts[v] = first index in [0, n) whose top bits ((key >> low) & tmask) are >= v, for v in [0, ng]
template <typename K>
__global__ void subbucket_bounds_kernel(const K *__restrict__ keys, u64 n, u32 low, u32 tmask, u64 ng,
                                        u64 *__restrict__ starts) {
  const u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (v > ng) return;
  if (v == ng) { starts[v] = n; return; }
  u64 lo = 0, hi = n;
  while (lo < hi) {
    const u64 mid = lo + ((hi - lo) >> 1);
    if ((u64)KeyOps<K>::digit(keys[mid], low, tmask) < v) lo = mid + 1; else hi = mid;
  }
  starts[v] = lo;
}

// largest sub-bucket of a file -> *max_out (atomicMax), so the host can pick the kernel capacity
// and the list of the sub-buckets above `threshold` (the ones the large-capacity launch takes)
__global__ void subbucket_max_kernel(const u64 *__restrict__ starts, u64 ng, u64 *__restrict__ max_out, u64 threshold,
                                     u32 *__restrict__ list, u64 *__restrict__ list_count) {
  const u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 sz = (v < ng) ? (starts[v + 1] - starts[v]) : 0ull;
  if (sz > threshold) list[atomicAdd(list_count, 1ull)] = (u32)v;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const u64 o = __shfl_down(sz, d); sz = (o > sz) ? o : sz; }
  if (lane_id() == 0 && sz) atomicMax(max_out, sz);
}

template <typename K, int BLOCK, int KPT>
struct FinishSmem {
  static constexpr int R = 256, NW = BLOCK / 64, CAP = BLOCK * KPT;
  static constexpr size_t RANK_BYTES = (size_t)NW * R * 12;
  static constexpr size_t REGION0 = ((size_t)CAP * sizeof(K) > RANK_BYTES) ? (size_t)CAP * sizeof(K) : RANK_BYTES;
  static constexpr size_t OFF_DBASE = REGION0;                        // u32[R]
  static constexpr size_t OFF_TMP   = OFF_DBASE + (size_t)R * 4;      // u32[64]
  static constexpr size_t OFF_FLAG  = OFF_TMP + 64 * 4;               // u8[CAP] head flags
  static constexpr size_t OFF_HP    = OFF_FLAG + (size_t)CAP;         // u16[CAP + 1] head positions
  static constexpr size_t BYTES     = OFF_HP + (size_t)(CAP + 2) * 2;
};

template <typename K, int BLOCK, int KPT>
__global__ __launch_bounds__(BLOCK)
void lds_sort_count_kernel(K *__restrict__ keys,                      // the file's segment; distinct keys are written in place
                           const u64 *__restrict__ starts,            // [ng+1] sub-bucket offsets inside the segment
                           u32 low_bits, u64 min_size, u64 max_size,   // this launch handles sub-buckets with min < n <= max
                           u32 *__restrict__ cnt_tmp,                 // counts, indexed like `keys`
                           u64 *__restrict__ group_distinct,          // [ng]
                           const u32 *__restrict__ list) {            // optional: the sub-buckets to take (grid = their number)
  using SM = FinishSmem<K, BLOCK, KPT>;
  using KO = KeyOps<K>;
  constexpr int R = SM::R, NW = SM::NW, CAP = SM::CAP;
  static_assert(BLOCK >= R && CAP <= 16384, "16-bit positions, one thread per digit");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  K   *s_keys  = reinterpret_cast<K *>(smem);
  u32 *s_whist = reinterpret_cast<u32 *>(smem);
  u32 *s_dbase = reinterpret_cast<u32 *>(smem + SM::OFF_DBASE);
  u32 *s_tmp   = reinterpret_cast<u32 *>(smem + SM::OFF_TMP);
  unsigned char  *s_flag = smem + SM::OFF_FLAG;
  unsigned short *s_hp   = reinterpret_cast<unsigned short *>(smem + SM::OFF_HP);

  const u32 tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const u64 g = list ? (u64)list[blockIdx.x] : (u64)blockIdx.x;
  const u64 a = starts[g];
  const u64 n64 = starts[g + 1] - a;
  if (n64 <= min_size || n64 > max_size) {             // another launch's (or nobody's) sub-bucket
    if (n64 == 0 && min_size == 0 && tid == 0) group_distinct[g] = 0;
    return;
  }
  const u32 n = (u32)n64;
  K *gk = keys + a;

  // ---- load, wave-striped like the global passes; padding sorts to the very end ----
  K kk[KPT];
#pragma unroll
  for (int j = 0; j < KPT; j++) {
    const u32 idx = w * (64 * KPT) + (u32)j * 64 + lane;
    kk[j] = (idx < n) ? gk[idx] : KO::pad();
  }

  // ---- stable 8-bit counting passes over the low bits, entirely in LDS ----
  const u64 lt_mask = (1ull << lane) - 1ull;
  const u64 lane_bit = 1ull << lane;
  for (u32 shift = 0; shift < low_bits; shift += 8) {
    const u32 bits  = (low_bits - shift < 8u) ? (low_bits - shift) : 8u;
    const u32 dmask = (1u << bits) - 1u;
    for (u32 i = tid; i < (u32)(NW * R * 3); i += BLOCK) s_whist[i] = 0;
    __syncthreads();
    lds_u32 *wh = (lds_u32 *)(smem) + w * R;
    lds_u64 *mk = (lds_u64 *)(smem + (size_t)NW * R * 4) + w * R;
    u32 ranks[KPT / 2];
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const u32 d = KO::digit(kk[j], shift, dmask);
      __hip_atomic_fetch_or(&mk[d], lane_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      const u64 peers = __hip_atomic_load(&mk[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      const u32 base  = __hip_atomic_load(&wh[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      const u32 lower = __popcll(peers & lt_mask);
      if (lower == 0) {
        __hip_atomic_store(&wh[d], base + (u32)__popcll(peers), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_store(&mk[d], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
      if (j & 1) ranks[j / 2] |= (base + lower) << 16;
      else       ranks[j / 2]  = (base + lower);
    }
    __syncthreads();
    u32 count = 0;
    if (tid < (u32)R) {
      u32 acc = 0;
#pragma unroll
      for (int ww = 0; ww < NW; ww++) {
        const u32 t = s_whist[ww * R + tid];
        s_whist[ww * R + tid] = acc;
        acc += t;
      }
      count = acc;
    }
    u32 tot;
    const u32 excl = block_excl_scan<BLOCK, u32>(count, s_tmp, &tot);
    if (tid < (u32)R) s_dbase[tid] = excl;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const u32 d = KO::digit(kk[j], shift, dmask);
      const u32 add = s_dbase[d] + s_whist[w * R + d];
      ranks[j / 2] += (j & 1) ? (add << 16) : add;
    }
    __syncthreads();                                   // scratch is dead; its storage becomes s_keys
#pragma unroll
    for (int j = 0; j < KPT; j++) s_keys[(j & 1) ? (ranks[j / 2] >> 16) : (ranks[j / 2] & 0xFFFFu)] = kk[j];
    __syncthreads();
    if (shift + 8 < low_bits) {                        // next pass ranks in the new order
#pragma unroll
      for (int j = 0; j < KPT; j++) kk[j] = s_keys[w * (64 * KPT) + (u32)j * 64 + lane];
      __syncthreads();
    }
  }
  if (low_bits == 0) {                                 // nothing to sort: all keys of the sub-bucket are equal bits above
#pragma unroll
    for (int j = 0; j < KPT; j++) s_keys[w * (64 * KPT) + (u32)j * 64 + lane] = kk[j];
    __syncthreads();
  }

  // ---- run-length count on the sorted sub-bucket ----
#pragma unroll
  for (int j = 0; j < KPT; j++) {                      // striped: conflict-free neighbour compares
    const u32 i = (u32)j * BLOCK + tid;
    s_flag[i] = (i < n && (i == 0 || KO::ne(s_keys[i], s_keys[i - 1]))) ? 1 : 0;
  }
  __syncthreads();
  u32 myflags = 0;                                      // blocked: KPT consecutive flags per thread
#pragma unroll
  for (int j = 0; j < KPT; j++) myflags |= (u32)s_flag[tid * KPT + j] << j;
  u32 d_total;
  u32 slot = block_excl_scan<BLOCK, u32>(__popc(myflags), s_tmp, &d_total);
#pragma unroll
  for (int j = 0; j < KPT; j++)
    if ((myflags >> j) & 1u) s_hp[slot++] = (unsigned short)(tid * KPT + j);
  if (tid == 0) s_hp[d_total] = (unsigned short)n;      // CAP <= 16384 fits
  __syncthreads();
  for (u32 sidx = tid; sidx < d_total; sidx += BLOCK) {
    const u32 i = s_hp[sidx];
    gk[sidx] = s_keys[i];                               // in place: every key of this region is in LDS by now
    cnt_tmp[a + sidx] = (u32)s_hp[sidx + 1] - i;
  }
  if (tid == 0) group_distinct[g] = d_total;
}

// Hash-count finish for uint64 keys: sub-buckets of real read sets are mostly duplicates (coverage),
// so instead of sorting n keys the workgroup inserts them into an LDS hash table that counts
// (one 64-bit CAS + one add per key, independent per key => the LDS latency overlaps), compacts
// the D distinct entries and ranks them by brute force (D^2 / BLOCK broadcast compares; D ~ n/7).
// EMPTY cannot collide with a key: every key of the file shares its top six bits, ~key0 does not.
template <int BLOCK, int CAP, int SLOTS, bool DBG>
__global__ __launch_bounds__(BLOCK, 5)
void hash_count_kernel(u64 *__restrict__ keys, const u64 *__restrict__ starts, u64 ng, u64 max_size, u32 low_bits,
                       u32 *__restrict__ cnt_tmp, u64 *__restrict__ group_distinct, u64 *__restrict__ dbg) {
  // Inside a sub-bucket the keys differ only in their low `low_bits` (< 32) bits: the table holds
  // 32-bit suffixes (half the LDS, 32-bit CAS and compares); the common prefix is added back on output.
  // Persistent workgroups: the keys of the next sub-bucket are loaded while the current one is counted
  // (a sub-bucket is ~1K keys, so the two dependent HBM round trips would otherwise be a third of its time).
  static_assert((SLOTS & (SLOTS - 1)) == 0 && SLOTS * 3 >= CAP * 4 && SLOTS % BLOCK == 0 && CAP % BLOCK == 0, "table geometry");
  constexpr int KPT = CAP / BLOCK, SPT = SLOTS / BLOCK;
  constexpr u32 EMPTY = 0xFFFFFFFFu;                   // suffixes are < 2^31
  __shared__ __attribute__((aligned(16))) u32 tk[SLOTS];
  __shared__ __attribute__((aligned(16))) u32 tc[SLOTS];
  __shared__ __attribute__((aligned(16))) u32 dk[CAP + 16];
  __shared__ u32 dc[CAP];
  __shared__ u32 s_tmp[BLOCK / 64 + 1];
  const u32 tid = threadIdx.x;
  const u64 G = gridDim.x;
  const u64 low_mask = (1ull << low_bits) - 1ull;

  auto load_bounds = [&](u64 gg, u64 &aa, u64 &nn) {
    aa = 0; nn = 0;
    if (gg < ng) { aa = starts[gg]; nn = starts[gg + 1] - aa; }
  };
  // only the low dword of a key is needed: the rest is the file's prefix and the sub-bucket index
  auto load_keys = [&](u64 aa, u64 nn, u32 (&kr)[KPT]) {
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const u32 idx = (u32)j * BLOCK + tid;
      kr[j] = (nn <= max_size && idx < nn) ? reinterpret_cast<const u32 *>(keys + aa + idx)[0] : 0u;
    }
  };
  const u32 group_shift = low_bits + (u32)__builtin_ctzll(ng);          // ng is a power of two
  const u64 file_base = (keys[0] >> group_shift) << group_shift;

  u64 g = blockIdx.x, a, n64, na, nn;
  u32 kcur[KPT];
  load_bounds(g, a, n64);
  load_keys(a, n64, kcur);
  load_bounds(g + G, na, nn);

  u64 ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
#define HC_STAMP(i) do { if (DBG) { const u64 t = __builtin_readcyclecounter(); ph[i] += t - t0; t0 = t; } } while (0)
  while (g < ng) {
    u32 knext[KPT];
    u64 nna, nnn;
    if (DBG) t0 = __builtin_readcyclecounter();
    load_keys(na, nn, knext);                          // in flight while this sub-bucket is processed
    load_bounds(g + 2 * G, nna, nnn);

    if (n64 == 0) {
      if (tid == 0) group_distinct[g] = 0;
    } else if (n64 <= max_size) {                      // larger ones: a larger-capacity launch takes them
      const u32 n = (u32)n64;
      const u64 prefix = file_base | (g << low_bits);
      u32 kk[KPT], hh[KPT];
      u32 pending = 0;
#pragma unroll
      for (int j = 0; j < KPT; j++) {
        const u32 idx = (u32)j * BLOCK + tid;
        kk[j] = kcur[j] & (u32)low_mask;
        if (idx < n) pending |= 1u << j;
      }
      // table sized to the sub-bucket (load factor <= 0.8 even if every key is distinct): the clear and the
      // compaction below walk the table, so an oversized one costs more than the longer probes of a full one
      u32 slots = 256;
      while (slots < n + n / 4 && slots < (u32)SLOTS) slots <<= 1;
      const u32 smask = slots - 1, sshift = 32 - (u32)__builtin_ctz(slots);
      {
        uint4 *tk4 = reinterpret_cast<uint4 *>(tk), *tc4 = reinterpret_cast<uint4 *>(tc);
        for (u32 i = tid; i < slots / 4; i += BLOCK) {
          tk4[i] = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
          tc4[i] = make_uint4(0u, 0u, 0u, 0u);
        }
      }
#pragma unroll
      for (int j = 0; j < KPT; j++) hh[j] = (kk[j] * 0x9E3779B1u) >> sshift;
      __syncthreads();
      HC_STAMP(0);

      // linear probing; one probe step of every still-pending key per round, so the CASes of a round overlap
      while (pending) {
#pragma unroll
        for (int j = 0; j < KPT; j++) {
          if ((pending >> j) & 1u) {
            const u32 old = atomicCAS(&tk[hh[j]], EMPTY, kk[j]);
            if (old == EMPTY || old == kk[j]) { atomicAdd(&tc[hh[j]], 1u); pending &= ~(1u << j); }
            else hh[j] = (hh[j] + 1) & smask;
          }
        }
      }
      __syncthreads();
      HC_STAMP(1);

      // compact the occupied slots (any order)
      u32 occ = 0;
#pragma unroll
      for (int j = 0; j < SPT; j++) {
        const u32 sl = (u32)j * BLOCK + tid;
        if (sl < slots) occ |= (tc[sl] != 0u ? 1u : 0u) << j;
      }
      u32 D;
      u32 o = block_excl_scan<BLOCK, u32>(__popc(occ), s_tmp, &D);
#pragma unroll
      for (int j = 0; j < SPT; j++)
        if ((occ >> j) & 1u) { dk[o] = tk[(u32)j * BLOCK + tid]; dc[o] = tc[(u32)j * BLOCK + tid]; o++; }
      if (tid < 16) dk[D + tid] = EMPTY;               // padding of the rank loop (D is block-uniform)
      __syncthreads();
      HC_STAMP(2);

      // rank = number of smaller distinct suffixes.  dk is padded with EMPTY (never smaller) to a multiple of 16, so
      // the loop runs on whole 64-byte groups: four independent broadcast 16-byte LDS reads in flight per iteration
      // (a (tried) order-preserving table with cluster-local ranks halves this phase but probes 50 % longer)
      const uint4 *dk4 = reinterpret_cast<const uint4 *>(dk);
      u64 *gk = keys + a;
      const u32 d16 = (D + 15) / 16;
      for (u32 i = tid; i < D; i += BLOCK) {
        const u32 ki = dk[i];
        u32 r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        for (u32 j = 0; j < d16; j++) {
          const uint4 v0 = dk4[4 * j], v1 = dk4[4 * j + 1], v2 = dk4[4 * j + 2], v3 = dk4[4 * j + 3];
          r0 += (v0.x < ki ? 1u : 0u) + (v0.y < ki ? 1u : 0u) + (v0.z < ki ? 1u : 0u) + (v0.w < ki ? 1u : 0u);
          r1 += (v1.x < ki ? 1u : 0u) + (v1.y < ki ? 1u : 0u) + (v1.z < ki ? 1u : 0u) + (v1.w < ki ? 1u : 0u);
          r2 += (v2.x < ki ? 1u : 0u) + (v2.y < ki ? 1u : 0u) + (v2.z < ki ? 1u : 0u) + (v2.w < ki ? 1u : 0u);
          r3 += (v3.x < ki ? 1u : 0u) + (v3.y < ki ? 1u : 0u) + (v3.z < ki ? 1u : 0u) + (v3.w < ki ? 1u : 0u);
        }
        const u32 r = r0 + r1 + r2 + r3;
        gk[r] = prefix | (u64)ki;                      // in place: every key of this region sits in registers
        cnt_tmp[a + r] = dc[i];
      }
      if (tid == 0) group_distinct[g] = D;
      HC_STAMP(3);
      __syncthreads();                                 // dk/dc/s_tmp are reused by the next sub-bucket
      HC_STAMP(4);
    }

#pragma unroll
    for (int j = 0; j < KPT; j++) kcur[j] = knext[j];
    a = na; n64 = nn; na = nna; nn = nnn; g += G;
    if (DBG) { ph[5] += kcur[0] & 1; HC_STAMP(6); ph[7]++; }   // [6]: wait for the prefetched keys
  }
  if (DBG && tid == 0 && blockIdx.x < 64)
    for (int i = 0; i < 8; i++) dbg[blockIdx.x * 8 + i] = ph[i];
#undef HC_STAMP
}

// Same scheme with 64-bit suffixes, for sub-buckets whose keys differ in 32..62 low bits (k from about 28 at the
// 10 Gbp scale): 64-bit CAS, whole keys loaded, 42 KiB of LDS (3 workgroups per CU).
template <int BLOCK, int CAP, int SLOTS>
__global__ __launch_bounds__(BLOCK, 3)
void hash_count64_kernel(u64 *__restrict__ keys, const u64 *__restrict__ starts, u64 ng, u64 max_size, u32 low_bits,
                       u32 *__restrict__ cnt_tmp, u64 *__restrict__ group_distinct) {
  constexpr bool DBG = false;
  u64 *dbg = nullptr;
  // Inside a sub-bucket the keys differ only in their low `low_bits` (< 32) bits: the table holds
  // 32-bit suffixes (half the LDS, 32-bit CAS and compares); the common prefix is added back on output.
  // Persistent workgroups: the keys of the next sub-bucket are loaded while the current one is counted
  // (a sub-bucket is ~1K keys, so the two dependent HBM round trips would otherwise be a third of its time).
  static_assert((SLOTS & (SLOTS - 1)) == 0 && SLOTS * 3 >= CAP * 4 && SLOTS % BLOCK == 0 && CAP % BLOCK == 0, "table geometry");
  constexpr int KPT = CAP / BLOCK, SPT = SLOTS / BLOCK;
  constexpr u64 EMPTY = ~0ull;                         // suffixes are < 2^63
  __shared__ __attribute__((aligned(16))) u64 tk[SLOTS];
  __shared__ __attribute__((aligned(16))) u32 tc[SLOTS];
  __shared__ __attribute__((aligned(16))) u64 dk[CAP + 16];
  __shared__ u32 dc[CAP];
  __shared__ u32 s_tmp[BLOCK / 64 + 1];
  const u32 tid = threadIdx.x;
  const u64 G = gridDim.x;
  const u64 low_mask = (1ull << low_bits) - 1ull;

  auto load_bounds = [&](u64 gg, u64 &aa, u64 &nn) {
    aa = 0; nn = 0;
    if (gg < ng) { aa = starts[gg]; nn = starts[gg + 1] - aa; }
  };
  // only the low dword of a key is needed: the rest is the file's prefix and the sub-bucket index
  auto load_keys = [&](u64 aa, u64 nn, u64 (&kr)[KPT]) {
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const u32 idx = (u32)j * BLOCK + tid;
      kr[j] = (nn <= max_size && idx < nn) ? keys[aa + idx] : 0ull;
    }
  };
  const u32 group_shift = low_bits + (u32)__builtin_ctzll(ng);          // ng is a power of two
  const u64 file_base = (keys[0] >> group_shift) << group_shift;

  u64 g = blockIdx.x, a, n64, na, nn;
  u64 kcur[KPT];
  load_bounds(g, a, n64);
  load_keys(a, n64, kcur);
  load_bounds(g + G, na, nn);

  u64 ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
#define HC_STAMP(i) do { if (DBG) { const u64 t = __builtin_readcyclecounter(); ph[i] += t - t0; t0 = t; } } while (0)
  while (g < ng) {
    u64 knext[KPT];
    u64 nna, nnn;
    if (DBG) t0 = __builtin_readcyclecounter();
    load_keys(na, nn, knext);                          // in flight while this sub-bucket is processed
    load_bounds(g + 2 * G, nna, nnn);

    if (n64 == 0) {
      if (tid == 0) group_distinct[g] = 0;
    } else if (n64 <= max_size) {                      // larger ones: a larger-capacity launch takes them
      const u32 n = (u32)n64;
      const u64 prefix = file_base | (g << low_bits);
      u64 kk[KPT];
      u32 hh[KPT];
      u32 pending = 0;
#pragma unroll
      for (int j = 0; j < KPT; j++) {
        const u32 idx = (u32)j * BLOCK + tid;
        kk[j] = kcur[j] & low_mask;
        if (idx < n) pending |= 1u << j;
      }
      // table sized to the sub-bucket (load factor <= 0.8 even if every key is distinct): the clear and the
      // compaction below walk the table, so an oversized one costs more than the longer probes of a full one
      u32 slots = 256;
      while (slots < n + n / 4 && slots < (u32)SLOTS) slots <<= 1;
      const u32 smask = slots - 1, sshift = 32 - (u32)__builtin_ctz(slots);
      {
        for (u32 i = tid; i < slots; i += BLOCK) { tk[i] = EMPTY; tc[i] = 0u; }
      }
#pragma unroll
      for (int j = 0; j < KPT; j++) hh[j] = (u32)((kk[j] * 0x9E3779B97F4A7C15ull) >> 32) >> sshift;
      __syncthreads();
      HC_STAMP(0);

      // linear probing; one probe step of every still-pending key per round, so the CASes of a round overlap
      while (pending) {
#pragma unroll
        for (int j = 0; j < KPT; j++) {
          if ((pending >> j) & 1u) {
            const u64 old = atomicCAS(&tk[hh[j]], EMPTY, kk[j]);
            if (old == EMPTY || old == kk[j]) { atomicAdd(&tc[hh[j]], 1u); pending &= ~(1u << j); }
            else hh[j] = (hh[j] + 1) & smask;
          }
        }
      }
      __syncthreads();
      HC_STAMP(1);

      // compact the occupied slots (any order)
      u32 occ = 0;
#pragma unroll
      for (int j = 0; j < SPT; j++) {
        const u32 sl = (u32)j * BLOCK + tid;
        if (sl < slots) occ |= (tc[sl] != 0u ? 1u : 0u) << j;
      }
      u32 D;
      u32 o = block_excl_scan<BLOCK, u32>(__popc(occ), s_tmp, &D);
#pragma unroll
      for (int j = 0; j < SPT; j++)
        if ((occ >> j) & 1u) { dk[o] = tk[(u32)j * BLOCK + tid]; dc[o] = tc[(u32)j * BLOCK + tid]; o++; }
      if (tid < 16) dk[D + tid] = EMPTY;               // padding of the rank loop (D is block-uniform)
      __syncthreads();
      HC_STAMP(2);

      // rank = number of smaller distinct suffixes (all pairs, two 16-byte broadcast reads per iteration)
      u64 *gk = keys + a;
      const u32 d4 = (D + 3) / 4;
      const ulonglong2 *dk2 = reinterpret_cast<const ulonglong2 *>(dk);
      for (u32 i = tid; i < D; i += BLOCK) {
        const u64 ki = dk[i];
        u32 r0 = 0, r1 = 0;
        for (u32 j = 0; j < d4; j++) {
          const ulonglong2 v0 = dk2[2 * j], v1 = dk2[2 * j + 1];
          r0 += (v0.x < ki ? 1u : 0u) + (v0.y < ki ? 1u : 0u);
          r1 += (v1.x < ki ? 1u : 0u) + (v1.y < ki ? 1u : 0u);
        }
        const u32 r = r0 + r1;
        gk[r] = prefix | ki;                           // in place: every key of this region sits in registers
        cnt_tmp[a + r] = dc[i];
      }
      if (tid == 0) group_distinct[g] = D;
      HC_STAMP(3);
      __syncthreads();                                 // dk/dc/s_tmp are reused by the next sub-bucket
      HC_STAMP(4);
    }

#pragma unroll
    for (int j = 0; j < KPT; j++) kcur[j] = knext[j];
    a = na; n64 = nn; na = nna; nn = nnn; g += G;
    if (DBG) { ph[5] += kcur[0] & 1; HC_STAMP(6); ph[7]++; }   // [6]: wait for the prefetched keys
  }
  if (DBG && tid == 0 && blockIdx.x < 64)
    for (int i = 0; i < 8; i++) dbg[blockIdx.x * 8 + i] = ph[i];
#undef HC_STAMP
}

// Hash-count finish for 16-byte keys (k = 33..64).  The suffix of a key inside its sub-bucket may be wider than any
// LDS compare-and-swap, so a slot is claimed through its COUNT word instead: 0 = empty, LOCK = being written,
// n >= 1 = valid with n instances.  A thread that finds LOCK simply stays pending for the next round (rounds, not
// spinning: the lane holding the lock may sit in the same wave).  WIDE = the suffix needs the high word too
// (low_bits > 64); otherwise the hi arrays are not even allocated.
template <int BLOCK, int CAP, int SLOTS, bool WIDE>
__global__ __launch_bounds__(BLOCK, WIDE ? 2 : 3)
void hash_count128_kernel(K128 *__restrict__ keys, const u64 *__restrict__ starts, u64 ng, u64 max_size, u32 low_bits,
                          u32 *__restrict__ cnt_tmp, u64 *__restrict__ group_distinct) {
  static_assert((SLOTS & (SLOTS - 1)) == 0 && SLOTS * 3 >= CAP * 4 && SLOTS % BLOCK == 0 && CAP % BLOCK == 0, "table geometry");
  constexpr int KPT = CAP / BLOCK, SPT = SLOTS / BLOCK;
  constexpr u32 LOCK = 0xFFFFFFFFu;
  __shared__ u64 tlo[SLOTS];
  __shared__ u64 thi[WIDE ? SLOTS : 1];
  __shared__ u32 tc[SLOTS];
  __shared__ __attribute__((aligned(16))) u64 dlo[CAP + 4];
  __shared__ __attribute__((aligned(16))) u64 dhi[WIDE ? CAP + 4 : 2];
  __shared__ u32 dc[CAP];
  __shared__ u32 s_tmp[BLOCK / 64 + 1];
  using KO = KeyOps<K128>;
  const u32 tid = threadIdx.x;
  const u64 G = gridDim.x;
  const u128 low_mask = (low_bits >= 128) ? ~(u128)0 : (((u128)1 << low_bits) - 1);
  const u32 group_shift = low_bits + (u32)__builtin_ctzll(ng);          // ng is a power of two
  const u128 file_base = (group_shift >= 128) ? (u128)0 : ((KO::v(keys[0]) >> group_shift) << group_shift);

  auto load_bounds = [&](u64 gg, u64 &aa, u64 &nn) {
    aa = 0; nn = 0;
    if (gg < ng) { aa = starts[gg]; nn = starts[gg + 1] - aa; }
  };
  auto load_keys = [&](u64 aa, u64 nn, K128 (&kr)[KPT]) {
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const u32 idx = (u32)j * BLOCK + tid;
      if (nn <= max_size && idx < nn) kr[j] = keys[aa + idx]; else kr[j] = KO::zero();
    }
  };

  u64 g = blockIdx.x, a, n64, na, nn;
  K128 kcur[KPT];
  load_bounds(g, a, n64);
  load_keys(a, n64, kcur);
  load_bounds(g + G, na, nn);

  while (g < ng) {
    K128 knext[KPT];
    u64 nna, nnn;
    load_keys(na, nn, knext);                          // in flight while this sub-bucket is processed
    load_bounds(g + 2 * G, nna, nnn);

    if (n64 == 0) {
      if (tid == 0) group_distinct[g] = 0;
    } else if (n64 <= max_size) {
      const u32 n = (u32)n64;
      const u128 prefix = file_base | ((u128)g << low_bits);
      u64 klo[KPT], khi[KPT];
      u32 hh[KPT];
      u32 pending = 0;
      u32 slots = 256;
      while (slots < n + n / 4 && slots < (u32)SLOTS) slots <<= 1;
      const u32 smask = slots - 1, sshift = 32 - (u32)__builtin_ctz(slots);
#pragma unroll
      for (int j = 0; j < KPT; j++) {
        const u32 idx = (u32)j * BLOCK + tid;
        const u128 sfx = KO::v(kcur[j]) & low_mask;
        klo[j] = (u64)sfx; khi[j] = (u64)(sfx >> 64);
        const u64 mix = (klo[j] ^ (khi[j] * 0xD6E8FEB86659FD93ull)) * 0x9E3779B97F4A7C15ull;
        hh[j] = (u32)(mix >> 32) >> sshift;
        if (idx < n) pending |= 1u << j;
      }
      for (u32 i = tid; i < slots; i += BLOCK) tc[i] = 0u;
      __syncthreads();

      while (pending) {
#pragma unroll
        for (int j = 0; j < KPT; j++) {
          if ((pending >> j) & 1u) {
            const u32 h = hh[j];
            const u32 old = atomicCAS(&tc[h], 0u, LOCK);
            if (old == 0u) {                           // the slot is ours: fill it, then publish it with count 1
              tlo[h] = klo[j];
              if (WIDE) thi[h] = khi[j];
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
              __hip_atomic_store(&tc[h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              pending &= ~(1u << j);
            } else if (old != LOCK) {                  // valid: same suffix -> count it, another one -> probe on
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
              const bool same = (tlo[h] == klo[j]) && (!WIDE || thi[h] == khi[j]);
              if (same) { atomicAdd(&tc[h], 1u); pending &= ~(1u << j); }
              else hh[j] = (h + 1) & smask;
            }                                          // LOCK: somebody is writing this slot; look again next round
          }
        }
      }
      __syncthreads();

      // compact the occupied slots (any order)
      u32 occ = 0;
#pragma unroll
      for (int j = 0; j < SPT; j++) {
        const u32 sl = (u32)j * BLOCK + tid;
        if (sl < slots) occ |= (tc[sl] != 0u ? 1u : 0u) << j;
      }
      u32 D;
      u32 o = block_excl_scan<BLOCK, u32>(__popc(occ), s_tmp, &D);
#pragma unroll
      for (int j = 0; j < SPT; j++) {
        if ((occ >> j) & 1u) {
          const u32 sl = (u32)j * BLOCK + tid;
          dlo[o] = tlo[sl];
          if (WIDE) dhi[o] = thi[sl];
          dc[o] = tc[sl];
          o++;
        }
      }
      if (tid < 4) { dlo[D + tid] = ~0ull; if (WIDE) dhi[D + tid] = ~0ull; }
      __syncthreads();

      // rank = number of smaller distinct suffixes
      K128 *gk = keys + a;
      const u32 d4 = (D + 3) / 4;                      // the arrays are padded with all-ones (never smaller) to 4
      const ulonglong2 *dlo2 = reinterpret_cast<const ulonglong2 *>(dlo);
      const ulonglong2 *dhi2 = reinterpret_cast<const ulonglong2 *>(dhi);
      for (u32 i = tid; i < D; i += BLOCK) {
        const u64 li = dlo[i], hi = WIDE ? dhi[i] : 0ull;
        u32 r0 = 0, r1 = 0;
        for (u32 j = 0; j < d4; j++) {
          const ulonglong2 a0 = dlo2[2 * j], a1 = dlo2[2 * j + 1];
          if (WIDE) {
            const ulonglong2 b0 = dhi2[2 * j], b1 = dhi2[2 * j + 1];
            r0 += (((b0.x < hi) || (b0.x == hi && a0.x < li)) ? 1u : 0u) + (((b0.y < hi) || (b0.y == hi && a0.y < li)) ? 1u : 0u);
            r1 += (((b1.x < hi) || (b1.x == hi && a1.x < li)) ? 1u : 0u) + (((b1.y < hi) || (b1.y == hi && a1.y < li)) ? 1u : 0u);
          } else {
            r0 += (a0.x < li ? 1u : 0u) + (a0.y < li ? 1u : 0u);
            r1 += (a1.x < li ? 1u : 0u) + (a1.y < li ? 1u : 0u);
          }
        }
        const u32 r = r0 + r1;
        gk[r] = KO::mk(prefix | ((u128)hi << 64) | (u128)li);   // in place: every key of this region sits in registers
        cnt_tmp[a + r] = dc[i];
      }
      if (tid == 0) group_distinct[g] = D;
      __syncthreads();                                 // the tables are reused by the next sub-bucket
    }

#pragma unroll
    for (int j = 0; j < KPT; j++) kcur[j] = knext[j];
    a = na; n64 = nn; na = nna; nn = nnn; g += G;
  }
}

// offs = exclusive scan of group_distinct (offs[ng] = total).  One wave per sub-bucket.
template <typename K>
__global__ __launch_bounds__(256)
void compact_groups_kernel(const K *__restrict__ keys, const u32 *__restrict__ cnt_tmp, const u64 *__restrict__ starts,
                           const u64 *__restrict__ offs, u64 ng, K *__restrict__ out_keys, u32 *__restrict__ out_counts) {
  const u64 g = (u64)blockIdx.x * 4 + wave_id();
  if (g >= ng) return;
  const u64 dst = offs[g], d = offs[g + 1] - dst, src = starts[g];
  for (u64 i = lane_id(); i < d; i += 64) {
    out_keys[dst + i]   = keys[src + i];
    out_counts[dst + i] = cnt_tmp[src + i];
  }
}

__global__ void store_u64_kernel(u64 *__restrict__ dst, const u64 *__restrict__ src) { *dst = *src; }

constexpr u64 FIN_CAP_SMALL = 256 * 16, FIN_CAP_LARGE = 1024 * 8;   // LDS: 46 KiB and 91 KiB per workgroup
constexpr u64 FIN_CAP_HASH  = 1536;                               // hash-count kernel: 2048 slots, 28 KiB of LDS, 5 workgroups per CU

static bool finish_uses_hash(uint32_t key_words, uint32_t low_bits) {
  static const bool use_hash = !(getenv("MGC_FINISH_HASH") && getenv("MGC_FINISH_HASH")[0] == '0');
  static const bool use_hash64 = !(getenv("MGC_FINISH_HASH64") && getenv("MGC_FINISH_HASH64")[0] == '0');
  static const bool use_hash128 = !(getenv("MGC_FINISH_HASH128") && getenv("MGC_FINISH_HASH128")[0] == '0');
  if (key_words == 2) return use_hash && use_hash128 && low_bits <= 122;
  return key_words == 1 && use_hash && (low_bits < 32 || (use_hash64 && low_bits <= 58));
}
// capacity of the first (small) launch of launch_finish_file; larger sub-buckets go on the list
static uint64_t finish_small_capacity(uint32_t key_words, uint32_t low_bits);

hipError_t launch_subbucket_bounds(const void *d_keys, uint64_t n, uint32_t key_words, uint32_t low, uint32_t top_bits,
                                   uint64_t *d_starts, uint64_t *d_max, uint32_t *d_list, uint64_t *d_list_count, hipStream_t st) {
  const uint64_t ng = (uint64_t)1 << top_bits;
  const uint32_t tmask = (uint32_t)(ng - 1);
  const dim3 grid((uint32_t)((ng + 1 + 255) / 256));
  if (key_words == 2)
    hipLaunchKernelGGL(subbucket_bounds_kernel<K128>, grid, dim3(256), 0, st, reinterpret_cast<const K128 *>(d_keys),
                       (u64)n, low, tmask, (u64)ng, reinterpret_cast<u64 *>(d_starts));
  else
    hipLaunchKernelGGL(subbucket_bounds_kernel<u64>, grid, dim3(256), 0, st, reinterpret_cast<const u64 *>(d_keys),
                       (u64)n, low, tmask, (u64)ng, reinterpret_cast<u64 *>(d_starts));
  MGC_CHECK(hipGetLastError());
  hipLaunchKernelGGL(subbucket_max_kernel, dim3((uint32_t)((ng + 255) / 256)), dim3(256), 0, st,
                     reinterpret_cast<const u64 *>(d_starts), (u64)ng, reinterpret_cast<u64 *>(d_max),
                     (u64)finish_small_capacity(key_words, low), d_list, reinterpret_cast<u64 *>(d_list_count));
  return hipGetLastError();
}

template <typename K, int BLOCK, int KPT>
static hipError_t finish_launch(void *d_keys, const uint64_t *d_starts, uint64_t ng, uint32_t low_bits, uint64_t min_size,
                                uint64_t max_size, uint32_t *d_cnt_tmp, uint64_t *d_group_distinct, hipStream_t st,
                                const uint32_t *d_list = nullptr) {
  if (ng == 0) return hipSuccess;
  using SM = FinishSmem<K, BLOCK, KPT>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lds_sort_count_kernel<K, BLOCK, KPT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)SM::BYTES);
    attr_done = true;
  }
  hipLaunchKernelGGL((lds_sort_count_kernel<K, BLOCK, KPT>), dim3((uint32_t)ng), dim3(BLOCK), SM::BYTES, st,
                     reinterpret_cast<K *>(d_keys), reinterpret_cast<const u64 *>(d_starts), low_bits, (u64)min_size,
                     (u64)max_size, d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct), d_list);
  return hipGetLastError();
}

// Sorts + counts every sub-bucket of one file segment; sub-buckets larger than FIN_CAP_SMALL use the
// large-capacity instantiation (launched only if the file has any: max_sub tells).
// MGC_HASH_DBG=1: per-phase cycle sums of the first 64 workgroups of the hash-count kernel, printed for a few launches
static u64 *hash_dbg_buffer() {
  static u64 *buf = nullptr;
  static const bool on = getenv("MGC_HASH_DBG") != nullptr;
  if (on && !buf) { if (hipMalloc(&buf, 64 * 8 * sizeof(u64)) != hipSuccess) buf = nullptr; }
  return buf;
}
static void hash_dbg_report(hipStream_t st, uint64_t ng) {
  u64 *buf = hash_dbg_buffer();
  static int reports = 0;
  if (!buf || reports >= 4) return;
  u64 h[64 * 8];
  if (hipStreamSynchronize(st) != hipSuccess) return;
  if (hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return;
  double sum[8] = {0};
  for (int b = 0; b < 64; b++) for (int i = 0; i < 8; i++) sum[i] += (double)h[b * 8 + i];
  const double it = sum[7] > 0 ? sum[7] : 1;
  fprintf(stderr, "[hash dbg] ng=%llu iters/block=%.1f cycles/iter: init=%.0f probe=%.0f compact=%.0f rank+store=%.0f sync=%.0f wait_next=%.0f\n",
          (unsigned long long)ng, it / 64, sum[0] / it, sum[1] / it, sum[2] / it, sum[3] / it, sum[4] / it, sum[6] / it);
  reports++;
}

hipError_t launch_finish_file(void *d_keys, uint32_t key_words, const uint64_t *d_starts, uint64_t ng, uint32_t low_bits,
                              uint64_t n_large, const uint32_t *d_large_list, uint32_t *d_cnt_tmp, uint64_t *d_group_distinct,
                              hipStream_t st) {
  if (key_words == 2 && finish_uses_hash(key_words, low_bits)) {
    static const uint32_t wgrid_max = getenv("MGC_HASH_GRID") ? (uint32_t)atoi(getenv("MGC_HASH_GRID")) : 256u * 6u;
    const uint32_t wgrid = ng < wgrid_max ? (uint32_t)ng : wgrid_max;
    if (low_bits > 64)
      hipLaunchKernelGGL((hash_count128_kernel<256, (int)FIN_CAP_HASH, 2048, true>), dim3(wgrid), dim3(256), 0, st,
                         reinterpret_cast<K128 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), (u64)ng, (u64)FIN_CAP_HASH, low_bits,
                         d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct));
    else
      hipLaunchKernelGGL((hash_count128_kernel<256, (int)FIN_CAP_HASH, 2048, false>), dim3(wgrid), dim3(256), 0, st,
                         reinterpret_cast<K128 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), (u64)ng, (u64)FIN_CAP_HASH, low_bits,
                         d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct));
    MGC_CHECK(hipGetLastError());
    MGC_CHECK((finish_launch<K128, 1024, 8>(d_keys, d_starts, n_large, low_bits, FIN_CAP_HASH, 8192, d_cnt_tmp, d_group_distinct, st, d_large_list)));
    return hipSuccess;
  }
  if (key_words == 2) {
    // 16-byte keys: 256x8 (2048) and 1024x8 (8192) keep LDS at 32 / 128 KiB
    MGC_CHECK((finish_launch<K128, 256, 8>(d_keys, d_starts, ng, low_bits, 0, 2048, d_cnt_tmp, d_group_distinct, st)));
    MGC_CHECK((finish_launch<K128, 1024, 8>(d_keys, d_starts, n_large, low_bits, 2048, 8192, d_cnt_tmp, d_group_distinct, st, d_large_list)));
    return hipSuccess;
  }
  if (finish_uses_hash(key_words, low_bits)) {
    // <= FIN_CAP_HASH keys: hash-count; larger sub-buckets: LDS radix passes in the 8192-key instantiation
    static const uint32_t hgrid_max = getenv("MGC_HASH_GRID") ? (uint32_t)atoi(getenv("MGC_HASH_GRID")) : 256u * 10u;
    const uint32_t hgrid = ng < hgrid_max ? (uint32_t)ng : hgrid_max;
    if (low_bits >= 32)
      hipLaunchKernelGGL((hash_count64_kernel<256, (int)FIN_CAP_HASH, 2048>), dim3(hgrid), dim3(256), 0, st,
                         reinterpret_cast<u64 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), (u64)ng, (u64)FIN_CAP_HASH, low_bits,
                         d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct));
    else if (hash_dbg_buffer())
      hipLaunchKernelGGL((hash_count_kernel<256, (int)FIN_CAP_HASH, 2048, true>), dim3(hgrid), dim3(256), 0, st,
                         reinterpret_cast<u64 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), (u64)ng, (u64)FIN_CAP_HASH, low_bits,
                         d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct), hash_dbg_buffer());
    else
      hipLaunchKernelGGL((hash_count_kernel<256, (int)FIN_CAP_HASH, 2048, false>), dim3(hgrid), dim3(256), 0, st,
                         reinterpret_cast<u64 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), (u64)ng, (u64)FIN_CAP_HASH, low_bits,
                         d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct), nullptr);
    MGC_CHECK(hipGetLastError());
    hash_dbg_report(st, ng);
    MGC_CHECK((finish_launch<u64, 1024, 8>(d_keys, d_starts, n_large, low_bits, FIN_CAP_HASH, FIN_CAP_LARGE, d_cnt_tmp,
                                           d_group_distinct, st, d_large_list)));
    return hipSuccess;
  }
  MGC_CHECK((finish_launch<u64, 256, 16>(d_keys, d_starts, ng, low_bits, 0, FIN_CAP_SMALL, d_cnt_tmp, d_group_distinct, st)));
  MGC_CHECK((finish_launch<u64, 1024, 8>(d_keys, d_starts, n_large, low_bits, FIN_CAP_SMALL, FIN_CAP_LARGE, d_cnt_tmp,
                                         d_group_distinct, st, d_large_list)));
  return hipSuccess;
}

static uint64_t finish_small_capacity(uint32_t key_words, uint32_t low_bits) {
  if (key_words == 2) return finish_uses_hash(key_words, low_bits) ? FIN_CAP_HASH : 2048;
  return finish_uses_hash(key_words, low_bits) ? FIN_CAP_HASH : FIN_CAP_SMALL;
}

uint64_t finish_capacity_for(uint32_t key_words) { return key_words == 2 ? 8192 : FIN_CAP_LARGE; }
uint64_t finish_target_for(uint32_t key_words) {
  if (const char *t = getenv("MGC_FINISH_TARGET")) return strtoull(t, nullptr, 10);
  const char *h = getenv("MGC_FINISH_HASH");
  if (key_words == 2) return (h && h[0] == '0') ? 1024 : FIN_CAP_HASH / 2;     // measured at k=51: 768 beats 512 and 1152
  return (h && h[0] == '0') ? FIN_CAP_SMALL / 2 : (FIN_CAP_HASH * 3) / 4;
}

// group_distinct[0..ng_total) -> exclusive offsets in place, total at [ng_total]
size_t finish_scan_scratch_bytes(uint64_t ng_total) { return scan_scratch_elems(ng_total + 1) * sizeof(uint64_t); }
hipError_t launch_finish_scan(uint64_t *d_group, uint64_t ng_total, void *d_scratch, hipStream_t st) {
  return scan_u64_inplace<false>(reinterpret_cast<u64 *>(d_group), ng_total, reinterpret_cast<u64 *>(d_scratch),
                                 reinterpret_cast<u64 *>(d_group) + ng_total, 0, st);
}

hipError_t launch_compact_groups(const void *d_keys, uint32_t key_words, const uint32_t *d_cnt_tmp, const uint64_t *d_starts,
                                 const uint64_t *d_offs, uint64_t ng, void *d_out_keys, uint32_t *d_out_counts, hipStream_t st) {
  const dim3 grid((uint32_t)((ng + 3) / 4));
  if (key_words == 2)
    hipLaunchKernelGGL(compact_groups_kernel<K128>, grid, dim3(256), 0, st, reinterpret_cast<const K128 *>(d_keys), d_cnt_tmp,
                       reinterpret_cast<const u64 *>(d_starts), reinterpret_cast<const u64 *>(d_offs), (u64)ng,
                       reinterpret_cast<K128 *>(d_out_keys), d_out_counts);
  else
    hipLaunchKernelGGL(compact_groups_kernel<u64>, grid, dim3(256), 0, st, reinterpret_cast<const u64 *>(d_keys), d_cnt_tmp,
                       reinterpret_cast<const u64 *>(d_starts), reinterpret_cast<const u64 *>(d_offs), (u64)ng,
                       reinterpret_cast<u64 *>(d_out_keys), d_out_counts);
  return hipGetLastError();
}

hipError_t launch_store_u64(uint64_t *d_dst, const uint64_t *d_src, hipStream_t st) {
  hipLaunchKernelGGL(store_u64_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<u64 *>(d_dst), reinterpret_cast<const u64 *>(d_src));
  return hipGetLastError();
}

// ============================================================================
//  Block offsets: first distinct key of every prefix
// ============================================================================
template <typename K>
__global__ void block_offsets_kernel(const K *__restrict__ keys, u64 nd, u32 w_data, u64 n_prefix,
                                     u64 *__restrict__ block_start) {
  const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (p > n_prefix) return;
  if (p == n_prefix) { block_start[p] = nd; return; }
  const K target = KeyOps<K>::prefix_floor(p, w_data);
  u64 lo = 0, hi = nd;
  while (lo < hi) {
    const u64 mid = lo + ((hi - lo) >> 1);
    if (KeyOps<K>::lt(keys[mid], target)) lo = mid + 1; else hi = mid;
  }
  block_start[p] = lo;
}

hipError_t launch_block_offsets(const void *d_unique, uint64_t n_distinct, uint32_t key_words, uint32_t w_data,
                                uint64_t n_prefix, uint64_t *d_block_start, hipStream_t st) {
  const uint64_t threads = n_prefix + 1;
  const dim3 grid((uint32_t)((threads + 255) / 256));
  if (key_words == 2)
    hipLaunchKernelGGL(block_offsets_kernel<K128>, grid, dim3(256), 0, st, reinterpret_cast<const K128 *>(d_unique),
                       (u64)n_distinct, w_data, (u64)n_prefix, reinterpret_cast<u64 *>(d_block_start));
  else
    hipLaunchKernelGGL(block_offsets_kernel<u64>, grid, dim3(256), 0, st, reinterpret_cast<const u64 *>(d_unique),
                       (u64)n_distinct, w_data, (u64)n_prefix, reinterpret_cast<u64 *>(d_block_start));
  return hipGetLastError();
}

// ============================================================================
//  Homopolymer compression (`compress`): merylInput.C:261-268 applies
//  homopolyCompress() to every chunk of a sequence, carrying the last byte across
//  chunks; on the whole base stream that is: drop every byte that equals
//  (case-insensitively) the byte before it.  '.' breakers never equal a base, so runs
//  do not merge across sequences.  Stream compaction: count, scan, emit.
// ============================================================================
constexpr int HP_BLOCK = 256;
constexpr int HP_TILE  = HP_BLOCK * 16;

__device__ __forceinline__ u32 hp_keep_mask(const uint8_t *__restrict__ in, u64 n, u64 pos, bool aligned, uint4 &v) {
  v = load16(in, pos, n, aligned);
  u32 prev = (pos == 0) ? 0x100u : ((u32)in[pos - 1] | 0x20u);
  const u32 w[4] = { v.x, v.y, v.z, v.w };
  u32 keep = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const u32 c = ((w[i >> 2] >> (8 * (i & 3))) & 0xFFu) | 0x20u;
    if (c != prev && pos + i < n) keep |= 1u << i;
    prev = c;
  }
  return keep;
}

__global__ __launch_bounds__(HP_BLOCK)
void hpc_count_kernel(const uint8_t *__restrict__ in, u64 n, u64 *__restrict__ tile_cnt) {
  __shared__ u32 s_tmp[HP_BLOCK / 64 + 1];
  const bool aligned = ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
  uint4 v;
  const u32 keep = hp_keep_mask(in, n, (u64)blockIdx.x * HP_TILE + (u64)threadIdx.x * 16, aligned, v);
  u32 tot;
  (void)block_excl_scan<HP_BLOCK, u32>(__popc(keep), s_tmp, &tot);
  if (threadIdx.x == 0) tile_cnt[blockIdx.x] = tot;
}

__global__ __launch_bounds__(HP_BLOCK)
void hpc_emit_kernel(const uint8_t *__restrict__ in, u64 n, const u64 *__restrict__ tile_offs, uint8_t *__restrict__ out) {
  __shared__ u32 s_tmp[HP_BLOCK / 64 + 1];
  const bool aligned = ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
  uint4 v;
  const u32 keep = hp_keep_mask(in, n, (u64)blockIdx.x * HP_TILE + (u64)threadIdx.x * 16, aligned, v);
  u32 tot;
  const u32 base = block_excl_scan<HP_BLOCK, u32>(__popc(keep), s_tmp, &tot);
  u64 o = tile_offs[blockIdx.x] + base;
  const u32 w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
  for (int i = 0; i < 16; i++)
    if ((keep >> i) & 1u) out[o++] = (uint8_t)((w[i >> 2] >> (8 * (i & 3))) & 0xFFu);
}

size_t hpc_workspace_bytes(uint64_t n) {
  const uint64_t t = (n + HP_TILE - 1) / HP_TILE;
  return (size_t)(8 + t + 1 + scan_scratch_elems(t + 1)) * sizeof(uint64_t);
}

// d_ws[0] receives the compressed length (device); the caller reads it back.
hipError_t launch_homopoly_compress(const uint8_t *d_in, uint64_t n, uint8_t *d_out, void *d_ws, hipStream_t st) {
  u64 *total = reinterpret_cast<u64 *>(d_ws);
  if (n == 0) return hipMemsetAsync(total, 0, sizeof(u64), st);
  const uint64_t t = (n + HP_TILE - 1) / HP_TILE;
  u64 *tile_offs = total + 8, *scratch = tile_offs + t + 1;
  hipLaunchKernelGGL(hpc_count_kernel, dim3((uint32_t)t), dim3(HP_BLOCK), 0, st, d_in, (u64)n, tile_offs);
  MGC_CHECK(hipGetLastError());
  MGC_CHECK(scan_u64_inplace<false>(tile_offs, t, scratch, total, 0, st));
  hipLaunchKernelGGL(hpc_emit_kernel, dim3((uint32_t)t), dim3(HP_BLOCK), 0, st, d_in, (u64)n, (const u64 *)tile_offs, d_out);
  return hipGetLastError();
}

// ============================================================================
//  Synthetic reads (byte-identical to oracle/oracle_count.c orc_synth_reads)
// ============================================================================
__host__ __device__ __forceinline__ u64 splitmix64(u64 x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

struct SynthParams {
  u64 s_genome, s_read, s_error, span, first_read, total_bytes, sub_thresh, n_thresh;
  u32 read_len;
};

__device__ __forceinline__ u32 synth_byte(const SynthParams &P, u64 o) {
  const u64 stride = (u64)P.read_len + 1;
  const u64 rr = o / stride;
  const u32 j  = (u32)(o - rr * stride);
  if (j == P.read_len) return (u32)'.';
  const u64 r     = P.first_read + rr;
  const u64 hr    = splitmix64(P.s_read ^ r);
  const u64 start = __umul64hi(hr, P.span);
  const bool rev  = (hr & 1ull) != 0;
  const u64 gpos  = rev ? (start + P.read_len - 1 - j) : (start + j);
  u32 code = (u32)(splitmix64(P.s_genome ^ gpos) & 3ull);
  if (rev) code ^= 2u;
  const u64 he = splitmix64(P.s_error ^ (r * (u64)P.read_len + j));
  const u32 e1 = (u32)he, e2 = (u32)(he >> 32);
  if ((u64)e1 < P.sub_thresh) code = (code + 1u + (e1 % 3u)) & 3u;
  const u32 acgt = 0x47544341u;                 // 'A','C','T','G' little-endian
  return ((u64)e2 < P.n_thresh) ? (u32)'N' : ((acgt >> (8 * code)) & 0xFFu);
}

__global__ __launch_bounds__(256)
void synth_reads_kernel(SynthParams P, uint8_t *__restrict__ out) {
  const u64 o4 = ((u64)blockIdx.x * 256 + threadIdx.x) * 4;
  if (o4 >= P.total_bytes) return;
  if (o4 + 4 <= P.total_bytes) {
    u32 w = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) w |= synth_byte(P, o4 + b) << (8 * b);
    *reinterpret_cast<u32 *>(out + o4) = w;
  } else {
    for (u64 o = o4; o < P.total_bytes; o++) out[o] = (uint8_t)synth_byte(P, o);
  }
}

hipError_t launch_synth_reads(uint64_t seed, uint64_t genome_len, uint64_t first_read, uint64_t n_reads,
                              uint32_t read_len, uint32_t sub_rate_ppm, uint32_t n_rate_ppm,
                              uint8_t *d_out, hipStream_t st) {
  if (n_reads == 0) return hipSuccess;
  SynthParams P;
  P.s_genome    = splitmix64(seed + 0ull * 0x632be59bd9b4e019ull);
  P.s_read      = splitmix64(seed + 1ull * 0x632be59bd9b4e019ull);
  P.s_error     = splitmix64(seed + 2ull * 0x632be59bd9b4e019ull);
  P.span        = genome_len - read_len + 1;
  P.first_read  = first_read;
  P.read_len    = read_len;
  P.total_bytes = n_reads * ((uint64_t)read_len + 1);
  P.sub_thresh  = (uint64_t)sub_rate_ppm * 4294967296ull / 1000000ull;
  P.n_thresh    = (uint64_t)n_rate_ppm * 4294967296ull / 1000000ull;
  const uint64_t threads = (P.total_bytes + 3) / 4;
  hipLaunchKernelGGL(synth_reads_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, st, P, d_out);
  return hipGetLastError();
}

}  // namespace mgc
