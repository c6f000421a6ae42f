// mgc_common.hpp -- shared by the gfx950 kernel translation units (mgc_kmer.hip, mgc_sort.hip,
// mgc_finish.hip, mgc_scan.hip, mgc_misc.hip, mgc_parse.hip is self-contained).  Not installed.
#pragma once
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

// Dense rank of five bases that never repeat the base before them (homopolymer-compressed sequence): x = 12 key bits, the
// top two = the base before the five.  Order preserving: d_i = c_i - (c_i > previous base) in {0, 1, 2}, rank = sum d_i 3^(4-i).
constexpr u32 HPC_DIGIT_MASK = 0xFFFFFFFFu;          // passed as a pass's "mask": the digit is hpc_digit() of the 12 bits at `shift`
__device__ __forceinline__ u32 hpc_digit(u32 x) {
  u32 prev = (x >> 10) & 3u, d = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const u32 c = (x >> (8 - 2 * i)) & 3u;
    d = d * 3u + (c - (c > prev ? 1u : 0u));
    prev = c;
  }
  return d;                                            // <= 242 for valid input, <= 363 for any input (bins exist up to 511)
}

template <typename K> struct KeyOps;
template <> struct KeyOps<u64> {
  static constexpr int WORDS = 1;
  static __device__ __forceinline__ u32  digit(u64 k, u32 shift, u32 mask) {
    if (mask == HPC_DIGIT_MASK) return hpc_digit((u32)(k >> shift) & 0xFFFu);
    return (u32)(k >> shift) & mask;
  }
  static __device__ __forceinline__ u32  bucket(u64 k, u32 shift) { return (u32)(k >> shift); }
  static __device__ __forceinline__ u64  pad() { return ~0ull; }
  static __device__ __forceinline__ u64  zero() { return 0ull; }
  static __device__ __forceinline__ bool ne(u64 a, u64 b) { return a != b; }
  static __device__ __forceinline__ bool lt(u64 a, u64 b) { return a < b; }
  static __device__ __forceinline__ u64  prefix_floor(u64 p, u32 w_data) { return p << w_data; }
  static __device__ __forceinline__ u64  low64(u64 k) { return k; }
};
// narrowed keys of the two-digit grouping path (mgc_sort.hip, launch_group_narrow): what is left of a k-mer once its
// bucket and its first grouping digit are known from where it lies -- at most 32 bits
template <> struct KeyOps<u32> {
  static constexpr int WORDS = 1;
  static __device__ __forceinline__ u32  digit(u32 k, u32 shift, u32 mask) { return (k >> shift) & mask; }
  static __device__ __forceinline__ u32  pad() { return ~0u; }
  static __device__ __forceinline__ bool ne(u32 a, u32 b) { return a != b; }
};
template <> struct KeyOps<K128> {
  static constexpr int WORDS = 2;
  static __device__ __forceinline__ u128 v(K128 k) { return ((u128)k.hi << 64) | (u128)k.lo; }
  static __device__ __forceinline__ K128 mk(u128 x) { K128 k; k.lo = (u64)x; k.hi = (u64)(x >> 64); return k; }
  static __device__ __forceinline__ u32  digit(K128 k, u32 shift, u32 mask) {
    if (mask == HPC_DIGIT_MASK) return hpc_digit((u32)(v(k) >> shift) & 0xFFFu);
    return (u32)(v(k) >> shift) & mask;
  }
  static __device__ __forceinline__ u32  bucket(K128 k, u32 shift) { return (u32)(v(k) >> shift); }
  static __device__ __forceinline__ K128 pad() { K128 k; k.lo = ~0ull; k.hi = ~0ull; return k; }
  static __device__ __forceinline__ K128 zero() { K128 k; k.lo = 0; k.hi = 0; return k; }
  static __device__ __forceinline__ bool ne(K128 a, K128 b) { return (a.lo != b.lo) || (a.hi != b.hi); }
  static __device__ __forceinline__ bool lt(K128 a, K128 b) { return (a.hi < b.hi) || (a.hi == b.hi && a.lo < b.lo); }
  static __device__ __forceinline__ K128 prefix_floor(u64 p, u32 w_data) { return mk((u128)p << w_data); }
  static __device__ __forceinline__ u64  low64(K128 k) { return k.lo; }
};

// 96-bit key (round 5): what a k-mer of k = 33..51 holds BELOW ITS FILE (2k - 6 <= 96 bits) -- the file is where the k-mer lies, so
// the partition, both whole-key grouping passes and the count kernel move 12 bytes per k-mer instead of 16 (the reference never
// stores more than the suffix bits either: merylCountArray.C:490-728, _sWidth-bit append).  w[0] least significant; 4-byte aligned.
struct alignas(4) K96 { u32 w[3]; };
template <> struct KeyOps<K96> {
  static constexpr int WORDS = 2;
  static __device__ __forceinline__ u128 v(K96 k) { return ((u128)k.w[2] << 64) | ((u128)k.w[1] << 32) | (u128)k.w[0]; }
  static __device__ __forceinline__ K96  mk(u128 x) { K96 k; k.w[0] = (u32)x; k.w[1] = (u32)(x >> 32); k.w[2] = (u32)(x >> 64); return k; }
  static __device__ __forceinline__ u32  digit(K96 k, u32 shift, u32 mask) {
    // (a digit is at most 12 bits: the two words it can touch)
    const u32 wi = shift >> 5, sh = shift & 31u;            // (uniform: selects, no indexed registers)
    const u32 lo_w = wi == 0 ? k.w[0] : (wi == 1 ? k.w[1] : k.w[2]);
    const u32 hi_w = wi == 0 ? k.w[1] : (wi == 1 ? k.w[2] : 0u);
    const u32 x = (u32)((((u64)hi_w << 32) | (u64)lo_w) >> sh);
    if (mask == HPC_DIGIT_MASK) return hpc_digit(x & 0xFFFu);
    return x & mask;
  }
  static __device__ __forceinline__ K96  pad() { K96 k; k.w[0] = k.w[1] = k.w[2] = ~0u; return k; }
  static __device__ __forceinline__ K96  zero() { K96 k; k.w[0] = k.w[1] = k.w[2] = 0u; return k; }
  static __device__ __forceinline__ bool ne(K96 a, K96 b) { return a.w[0] != b.w[0] || a.w[1] != b.w[1] || a.w[2] != b.w[2]; }
  static __device__ __forceinline__ u64  low64(K96 k) { return ((u64)k.w[1] << 32) | (u64)k.w[0]; }
};

// Sub-bucket index of a narrowed file in PHYSICAL order -> the top bits its k-mers hold (launch_group_narrow: with the high
// digit first the file ends up ordered by (low digit : high digit)); a = 0: the same thing.
__device__ __forceinline__ u64 tr_index(u64 p, u32 a, u32 b) { return a ? (((p & ((1ull << a) - 1ull)) << b) | (p >> a)) : p; }

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

typedef __attribute__((address_space(3))) u64 lds_u64;
typedef __attribute__((address_space(3))) u32 lds_u32;

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

// ---- multi-block scans over uint64 arrays, in place (mgc_scan.hip) ----
//   exclusive sum        : a[i] <- sum_{j<i} a[j]; the total goes to *d_total (device, may be null)
//   reverse inclusive min: a[i] <- min(a[i], ..., a[n-1], init)
size_t     scan_scratch_elems(uint64_t n);                 // scratch uint64 entries either scan needs
hipError_t scan_u64_exclusive(u64 *a, uint64_t n, u64 *scratch, u64 *d_total, hipStream_t st);
hipError_t scan_u64_min_reverse(u64 *a, uint64_t n, u64 *scratch, u64 init, hipStream_t st);

}  // namespace mgc
