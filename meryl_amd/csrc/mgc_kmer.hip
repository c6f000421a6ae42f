// mgc_kmer.hip -- k-mer extraction, file histogram and partition (gfx950).
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
#include "mgc_common.hpp"

namespace mgc {

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

// Which of the thread's 16 window starts hold a complete k-mer (k <= 32): bit j = no invalid base among bases j .. j+k-1.
// I has bit 63-p set for an invalid base p of the 48-base window; the k-wide OR is built by doubling (5 shifted ORs for
// any k) instead of one 64-bit shift pair per start.
__device__ __forceinline__ u32 kp_valid_mask(u64 I, u32 k) {
  u64 X = I;
  u32 c = 1;
  while (2 * c <= k) { X |= X << c; c *= 2; }
  if (c < k) X |= X << (k - c);
  return __brev((u32)(~X >> 48) & 0xFFFFu) >> 16;                    // start j -> bit j
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
  u64 r = 0;
#pragma unroll
  for (int j = 0; j < KP_ITEMS; j++) {
    const u64 top = (j == 0) ? A : ((A << (2 * j)) | (B >> (64 - 2 * j)));
    const u64 f   = top >> key_shift;
    if (j == 0) r = revcomp64(f, key_shift);
    else        r = (r >> 2) | ((((f & 3ull) ^ 2ull)) << top_shift);
    u64 key;
    if      (mode == 1) key = f;
    else if (mode == 2) key = r;
    else                key = (f < r) ? f : r;
    keys[j] = key;
  }
  return kp_valid_mask(I, k);
}

// Only the FILE (bucket) of each of those k-mers, for the histogram pass: the top bits of min(f, r) are the smaller of
// f's and r's top bits (equal tops give the same bucket whichever is smaller), so neither the k-mer nor its reverse
// complement is ever assembled -- f's top comes from the first bases of the window start, r's from the reverse
// complement of the k-mer's LAST bases, all sixteen of which sit in one 32-base chunk that is reversed once.
// bucket_bits <= 18, k <= 32, 2k >= bucket_bits.
__device__ __forceinline__ u32 kp_thread_buckets(const u32 *s_codes, const u32 *s_inval, u32 k, int mode, u32 bucket_bits,
                                                 u32 (&bk)[KP_ITEMS], const u32 t = threadIdx.x) {
  const u64 A = ((u64)s_codes[t] << 32) | (u64)s_codes[t + 1];      // bases 0..31 of the thread's window
  const u64 B = (u64)s_codes[t + 2] << 32;                          // bases 32..47
  const u64 I = ((u64)s_inval[t] << 48) | ((u64)s_inval[t + 1] << 32) | ((u64)s_inval[t + 2] << 16);
  const u32 m  = (bucket_bits + 1) / 2;                              // bases that decide the bucket
  const u32 z0 = k - m;                                              // the last m bases of start j are bases z0+j .. z0+j+m-1
  const u64 Z  = (z0 == 0) ? A : ((A << (2 * z0)) | (B >> (64 - 2 * z0)));
  const u64 RC = revcomp64(Z, 0);                                    // base i of Z -> base 31-i, complemented
  const u64 mmask = (1ull << (2 * m)) - 1ull;
  const u32 odd = 2 * m - bucket_bits;
#pragma unroll
  for (int j = 0; j < KP_ITEMS; j++) {
    const u32 ft = (u32)(((j == 0) ? A : (A << (2 * j))) >> (64 - bucket_bits));
    const u32 rt = (u32)((RC >> (2 * j)) & mmask) >> odd;
    bk[j] = (mode == 1) ? ft : (mode == 2) ? rt : (ft < rt ? ft : rt);
  }
  return kp_valid_mask(I, k);
}

// kp_thread_buckets for k in 33..64 (80-base windows, five staged words): the forward tops still come from the window's first
// bases; the last m bases of the sixteen starts (bases z0 .. z0+m+14, z0 = k - m >= 25) sit in one 32-base chunk cut from
// words z0/16 .. -- reversed once, as above.  The valid mask is the k-wide OR of the invalid-base bits, by doubling, on
// 128 bits.  bucket_bits <= 18.
__device__ __forceinline__ u32 kp_thread_buckets_wide(const u32 *s_codes, const u32 *s_inval, u32 k, int mode, u32 bucket_bits,
                                                      u32 (&bk)[KP_ITEMS], const u32 t) {
  const u64 A  = ((u64)s_codes[t] << 32) | (u64)s_codes[t + 1];      // bases 0..31 of the thread's window
  const u32 m  = (bucket_bits + 1) / 2, z0 = k - m;
  const u32 q  = z0 >> 4, r = z0 & 15u;                              // q <= 3: words q, q+1 always exist
  const u64 H  = ((u64)s_codes[t + q] << 32) | (u64)s_codes[t + q + 1];
  const u64 L  = (q + 2 <= 4) ? ((u64)s_codes[t + q + 2] << 32) : 0ull;   // (q = 3: r <= 8, the 23 bases needed all come from H)
  const u64 Z  = r ? ((H << (2 * r)) | (L >> (64 - 2 * r))) : H;      // bases z0 .. z0+31
  const u64 RC = revcomp64(Z, 0);
  const u64 mmask = (1ull << (2 * m)) - 1ull;
  const u32 odd = 2 * m - bucket_bits;
#pragma unroll
  for (int j = 0; j < KP_ITEMS; j++) {
    const u32 ft = (u32)(((j == 0) ? A : (A << (2 * j))) >> (64 - bucket_bits));
    const u32 rt = (u32)((RC >> (2 * j)) & mmask) >> odd;
    bk[j] = (mode == 1) ? ft : (mode == 2) ? rt : (ft < rt ? ft : rt);
  }
  u128 X = ((u128)s_inval[t] << 112) | ((u128)s_inval[t + 1] << 96) | ((u128)s_inval[t + 2] << 80) |
           ((u128)s_inval[t + 3] << 64) | ((u128)s_inval[t + 4] << 48);
  u32 c = 1;
  while (2 * c <= k) { X |= X << c; c *= 2; }
  if (c < k) X |= X << (k - c);
  return __brev((u32)(~X >> 112) & 0xFFFFu) >> 16;                   // start j -> bit j
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

// count-suffix= (merylOp-countSimple.C:88-93): a k-mer is counted only if the one that is counted ends in the given bases
template <typename K>
__device__ __forceinline__ u32 kp_suffix_filter(const K (&keys)[KP_ITEMS], u32 vmask, u64 sfx_mask, u64 sfx_test) {
  if (sfx_mask == 0) return vmask;                   // uniform
#pragma unroll
  for (int j = 0; j < KP_ITEMS; j++)
    if ((KeyOps<K>::low64(keys[j]) & sfx_mask) != sfx_test) vmask &= ~(1u << j);
  return vmask;
}

__device__ __forceinline__ void kp_tile_range(u64 num_tiles, u64 &t_begin, u64 &t_end, u32 vwg = blockIdx.x, u32 vgrid = gridDim.x) {
  const u64 per = (num_tiles + vgrid - 1) / vgrid;
  t_begin = (u64)vwg * per;
  t_end   = t_begin + per;
  if (t_begin > num_tiles) t_begin = num_tiles;
  if (t_end   > num_tiles) t_end   = num_tiles;
}

// Pass 1: per-workgroup and global per-bucket instance counts.
template <typename K>
__global__ __launch_bounds__(KP_BLOCK)
void kmer_hist_kernel(const uint8_t *__restrict__ bases, u64 n, u32 k, int mode, u32 bucket_bits,
                      u64 num_tiles, u64 *__restrict__ block_hist, u64 *__restrict__ bucket_counts, u64 sfx_mask, u64 sfx_test) {
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
    if (sizeof(K) == 8 && nb > 1 && 2 * k >= bucket_bits && sfx_mask == 0) {
      u32 bk[KP_ITEMS];
      const u32 vmask = kp_thread_buckets(s_codes, s_inval, k, mode, bucket_bits, bk);
#pragma unroll
      for (int j = 0; j < KP_ITEMS; j++)
        if ((vmask >> j) & 1u) atomicAdd(&s_hist[bk[j]], 1u);
    } else {
      K keys[KP_ITEMS];
      const u32 vmask = kp_suffix_filter(keys, kp_thread_kmers(s_codes, s_inval, k, mode, keys), sfx_mask, sfx_test);
      if (nb == 1) {
        my_count += __popc(vmask);
      } else {
#pragma unroll
        for (int j = 0; j < KP_ITEMS; j++)
          if ((vmask >> j) & 1u) atomicAdd(&s_hist[KeyOps<K>::bucket(keys[j], bucket_shift)], 1u);
      }
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

// kmer_hist_kernel for the narrowed grouping path (mgc_sort.hip, launch_group_narrow): besides the per-file counts it
// takes the histogram of the file AND the next nine bits of every k-mer -- the digit a file's first grouping pass
// groups by -- so that no pass has to read the file's k-mers just to size its digit regions (the 8 B/k-mer
// radix_hist_kernel read).  2^15 LDS counters = 128 KiB, hence ONE 1024-thread workgroup per CU standing for NV = 4
// workgroups of kmer_hist_kernel (their tile ranges and the block_hist rows the partition kernel expects from workgroups
// blockIdx.x * NV ..), taken one after the other; the next tiles' bases are loaded while the current ones are counted.
// k <= 64 (k > 32: kp_thread_buckets_wide), 2k >= 17, no count-suffix, 64 buckets.
constexpr int KH_NV = 4, KH_FINE_BITS = 15;

// `compress` (HB > 0): the table is indexed by the DENSE RANK of the k-mer's first HB homopolymer-free bases -- the bucket's
// bases (bucket_bits / 2 of them) and the five of the digit a bucket's first grouping pass goes by (hpc_digit, mgc_common.hpp:
// rank = rank of the bucket's bases * 243 + digit).  4 * 3^(HB-1) counters: 8748 (64 buckets) or 26244 (256 buckets).
__host__ __device__ constexpr u32 hpc_table_size(int hb) { u32 t = 4; for (int i = 1; i < hb; i++) t *= 3; return t; }
template <int HB>
__device__ __forceinline__ u32 hpc_dense_rank(u32 x) {               // x: 2 * HB bits, first base most significant
  u32 prev = (x >> (2 * HB - 2)) & 3u, r = prev;
#pragma unroll
  for (int i = 1; i < HB; i++) {
    const u32 c = (x >> (2 * (HB - 1 - i))) & 3u;
    r = r * 3u + (c - (c > prev ? 1u : 0u));
    prev = c;
  }
  return r;
}

template <int HB, int KC = 0>                         // KC: k as a compile-time constant, canonical mode (kmer_partition_kernel)
__global__ __launch_bounds__(KP_BLOCK * KH_NV)
void kmer_hist_fine_kernel(const uint8_t *__restrict__ bases, u64 n, u32 k_arg, int mode_arg, u64 num_tiles, u32 vgrid,
                           u64 *__restrict__ block_hist, u64 *__restrict__ bucket_counts, u64 *__restrict__ fine_hist,
                           u32 fbits = 6 /* HB == 0: the buckets of the per-workgroup rows and of bucket_counts are the top fbits (6..8) bits */,
                           u32 nvp = KH_NV /* virtual workgroups (rows) this workgroup takes, one after the other */) {
  const u32 k = KC ? (u32)KC : k_arg;
  const int mode = KC ? 0 : mode_arg;
  constexpr u32 TABLE = HB ? hpc_table_size(HB) : (1u << KH_FINE_BITS);
  constexpr u32 NBK = HB ? (1u << (2 * (HB - 5))) : 64u;              // buckets: 64 files, or 2^bucket_bits
  constexpr u32 IDX_BITS = HB ? 2u * HB : (u32)KH_FINE_BITS;          // top bits of the k-mer that index the table
  extern __shared__ __attribute__((aligned(16))) u32 kh_fine[];      // [TABLE]
  __shared__ u32 s_codes[2][KH_NV][KP_WORDS];
  __shared__ u32 s_inval[2][KH_NV][KP_WORDS];
  __shared__ u32 s_prev[HB ? NBK : 256u];
  // `compress` (round 6): the dense rank of the k-mer's first HB bases from two tables -- rank of the bucket's bases * 243, and
  // hpc_digit of the five bases below them given the bucket's last base (twelve bits) -- instead of HB dependent compare /
  // multiply-add steps per k-mer (hpc_dense_rank<9>: ~45 VALU instructions, the kernel was VALU-bound: 5.9 against 3.1 ms per 5 Gbp)
  __shared__ unsigned short s_r5[HB ? 4096 : 1];
  __shared__ unsigned short s_rb[HB ? NBK : 1];
  const u32 tid = threadIdx.x, v = tid >> 8, t = tid & 255u;
  if constexpr (HB > 0) {
    for (u32 i = tid; i < 4096u; i += KP_BLOCK * KH_NV) s_r5[i] = (unsigned short)hpc_digit(i);
    if (tid < NBK) s_rb[tid] = (unsigned short)(hpc_dense_rank<HB - 5>(tid) * 243u);
  }
  const bool aligned = ((reinterpret_cast<uintptr_t>(bases) & 15) == 0);
  for (u32 i = tid; i < TABLE; i += KP_BLOCK * KH_NV) kh_fine[i] = 0;
  if (tid < (HB ? NBK : 256u)) s_prev[tid] = 0;

  // The workgroup takes its KH_NV virtual workgroups ONE AFTER THE OTHER, all 1024 threads on one of them (slice v: every
  // KH_NV-th tile of its range): a k-mer costs ONE LDS atomic -- the fifteen-bit counter -- and the per-file counts of a
  // virtual workgroup (the row the partition kernel's cursors come from) are the growth of the 64 file sums of that table
  // while it was counted: 32 reads per thread and virtual workgroup instead of a second, conflict-ridden atomic per k-mer.
  const u64 per = (num_tiles + vgrid - 1) / vgrid;                   // kp_tile_range of a virtual workgroup
  for (u32 vv = 0; vv < nvp; vv++) {
    const u64 vwg = (u64)blockIdx.x * nvp + vv;
    u64 t_begin = vwg * per, t_end = t_begin + per;
    if (t_begin > num_tiles) t_begin = num_tiles;
    if (t_end > num_tiles) t_end = num_tiles;
    if (vwg >= vgrid) t_begin = t_end = num_tiles;

    uint4 cur = make_uint4(0, 0, 0, 0), halo = make_uint4(0, 0, 0, 0);
    auto fetch = [&](u64 tile) {
      if (tile >= t_end) return;
      cur = load16(bases, tile * KP_TILE + (u64)t * 16, n, aligned);
      if (t < 4) halo = load16(bases, tile * KP_TILE + (u64)KP_TILE + (u64)t * 16, n, aligned);
    };
    fetch(t_begin + v);
    __syncthreads();                                                 // (the table is cleared / the sums below are taken)
    const u64 rounds = (t_end - t_begin + KH_NV - 1) / KH_NV;
    for (u64 it = 0; it < rounds; it++) {
      const u64 tile = t_begin + it * KH_NV + v;
      const u32 buf = (u32)it & 1u;
      const bool active = tile < t_end;
      if (active) {
        u32 c, iv;
        encode16(cur, c, iv);
        s_codes[buf][v][t] = c; s_inval[buf][v][t] = iv;
        if (t < 4) { encode16(halo, c, iv); s_codes[buf][v][KP_BLOCK + t] = c; s_inval[buf][v][KP_BLOCK + t] = iv; }
      }
      __syncthreads();
      fetch(tile + KH_NV);                                           // in flight behind the counting below
      if (active) {
        u32 bk[KP_ITEMS];
        const u32 vmask = (k > 32) ? kp_thread_buckets_wide(s_codes[buf][v], s_inval[buf][v], k, mode, IDX_BITS, bk, t)
                                   : kp_thread_buckets(s_codes[buf][v], s_inval[buf][v], k, mode, IDX_BITS, bk, t);
#pragma unroll
        for (int j = 0; j < KP_ITEMS; j++)
          if ((vmask >> j) & 1u) {
            if constexpr (HB > 0) { const u32 r = (u32)s_rb[bk[j] >> 10] + (u32)s_r5[bk[j] & 0xFFFu]; atomicAdd(&kh_fine[r < TABLE ? r : TABLE - 1u], 1u); }
            else atomicAdd(&kh_fine[bk[j]], 1u);
          }
      }
    }
    __syncthreads();
    {                                                                // NBK buckets x TPB threads: the bucket sums so far
      constexpr u32 TPB = (u32)(KP_BLOCK * KH_NV) / NBK;             // 16 (64 buckets) or 4 (256)
      const u32 f = tid / TPB, l = tid % TPB;
      u32 sum = 0;
      (void)f; (void)l; (void)sum;
      if constexpr (HB > 0) {
        constexpr int BB = HB - 5;                                   // bases of a bucket
        bool ok = true;                                              // (a bucket that repeats a base holds no k-mer)
#pragma unroll
        for (int i = 1; i < BB; i++) ok = ok && (((f >> (2 * (BB - 1 - i))) & 3u) != ((f >> (2 * (BB - i))) & 3u));
        if (ok) { const u32 r0 = hpc_dense_rank<BB>(f) * 243u; for (u32 i = l; i < 243u; i += TPB) sum += kh_fine[r0 + i]; }
      }
      if constexpr (HB > 0) {
#pragma unroll
        for (u32 o = TPB / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (l == 0) {
          const u32 c = sum - s_prev[f];
          s_prev[f] = sum;
          if (vwg < vgrid) block_hist[vwg * NBK + f] = c;
          if (c) atomicAdd(&bucket_counts[f], (u64)c);
        }
      } else if (fbits == 6) {                                       // the 64 files: 16 threads each (compile-time shapes: the session's own path)
        for (u32 i = l; i < (1u << (KH_FINE_BITS - 6)); i += TPB) sum += kh_fine[(f << (KH_FINE_BITS - 6)) + i];
#pragma unroll
        for (u32 o = TPB / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (l == 0) {
          const u32 c = sum - s_prev[f];
          s_prev[f] = sum;
          if (vwg < vgrid) block_hist[vwg * NBK + f] = c;
          if (c) atomicAdd(&bucket_counts[f], (u64)c);
        }
      } else {
        // 2^fbits buckets x (1024 >> fbits) threads: the finer buckets of a sharded count's senders (8 or 4 threads each)
        const u32 tpb = (u32)(KP_BLOCK * KH_NV) >> fbits, fr = tid / tpb, lr = tid % tpb, per_bucket = 1u << (KH_FINE_BITS - fbits);
        u32 rs = 0;
        for (u32 i = lr; i < per_bucket; i += tpb) rs += kh_fine[(fr << (KH_FINE_BITS - fbits)) + i];
        for (u32 o = tpb / 2; o > 0; o >>= 1) rs += __shfl_xor(rs, o);
        if (lr == 0) {
          const u32 c = rs - s_prev[fr];
          s_prev[fr] = rs;
          if (vwg < vgrid) block_hist[(vwg << fbits) + fr] = c;
          if (c) atomicAdd(&bucket_counts[fr], (u64)c);
        }
      }
    }
  }
  __syncthreads();
  for (u32 i = tid; i < TABLE; i += KP_BLOCK * KH_NV) {
    const u32 c = kh_fine[i];
    if (c) atomicAdd(&fine_hist[i], (u64)c);
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
// SOA (round 4; 8-byte keys whose bits below the file fit 40: k <= 23): a file's region holds its k-mers as a u32 array (low
// words) followed by a u8 array (bits 32..39) -- 5 bytes per k-mer instead of 8 leave this kernel and enter the file's first
// grouping pass (radix_group_kernel<SOA>), which only ever needed 36 of the 64 bits (the file is where the k-mer lies).
// Measured with the debug forms of round 4 (profiles/r04j_part_dbg.txt): 8-byte stores cost 10.4 of this kernel's 27.6 ms,
// 4-byte stores 3.5.  soa_starts / soa_counts: first k-mer and number of k-mers of every file.
// SOA with 16-byte keys (round 5; k = 33..51: at most 96 bits below the file): the region of a file f with soa_counts[f] != 0 (a
// per-file FLAG here, not a count) holds its k-mers as 12-byte K96 records (mgc_common.hpp) -- 12 of the 16 bytes leave this kernel
// and go through both grouping passes and the count; the other files (too small for two grouping digits) keep 16-byte keys.
// KC (round 5): k as a COMPILE-TIME constant for the k of the BASELINE configs (21, 31, 51), canonical mode, the 64 files, no
// count-suffix filter -- every shift of the sixteen unrolled window extractions (key_shift, top_shift, the reverse complement's)
// is then an immediate and the 64-bit (k > 32: 128-bit) variable shifts on a 32-bit ALU go away; KC = 0: k, mode and bucket_bits
// are the run-time arguments.
// Stores through an address that comes out of LDS as an integer (the per-file output cursors): a plain pointer cast makes it a GENERIC
// pointer and the store a flat_store -- issued to the LDS pipeline as well as to memory, and counted in lgkmcnt, so that the barrier
// behind the write-out also waits for it (round 6, from the ISA: the partition's two stores per k-mer were its only flat instructions).
typedef __attribute__((address_space(1))) u32     kp_gu32;
typedef __attribute__((address_space(1))) uint8_t kp_gu8;
typedef __attribute__((address_space(1))) u64     kp_gu64;
typedef u32 kp_u32x3 __attribute__((ext_vector_type(3)));
typedef __attribute__((address_space(1))) __attribute__((aligned(4))) kp_u32x3 kp_gu32x3;
__device__ __forceinline__ void kp_gstore32(u64 addr, u32 v)     { *reinterpret_cast<kp_gu32 *>(addr) = v; }
__device__ __forceinline__ void kp_gstore8(u64 addr, uint8_t v)  { *reinterpret_cast<kp_gu8 *>(addr) = v; }
__device__ __forceinline__ void kp_gstore64(u64 addr, u64 v)     { *reinterpret_cast<kp_gu64 *>(addr) = v; }
__device__ __forceinline__ void kp_gstore96(u64 addr, u32 a, u32 b, u32 c) { kp_u32x3 v; v.x = a; v.y = b; v.z = c; *reinterpret_cast<kp_gu32x3 *>(addr) = v; }

template <typename K, int MAXB, bool SOA = false, int KC = 0, int BB = 6>   // BB: the bucket bits of a constant-k form (6: the files; 8: `compress` at 5 Gbp and beyond)
__global__ __launch_bounds__(KP_BLOCK, (sizeof(K) == 16) ? 2 : 4)   // 16-byte keys: the 64 KiB exchange tile allows two workgroups; 8-byte: four (five left 96 VGPRs: 20 spilled at constant k)
void kmer_partition_kernel(const uint8_t *__restrict__ bases, u64 n, u32 k_arg, int mode_arg, u32 bucket_bits_arg,
                           u64 num_tiles, const u64 *__restrict__ block_base, K *__restrict__ out, u64 sfx_mask_arg, u64 sfx_test,
                           const u64 *__restrict__ soa_starts = nullptr, const u64 *__restrict__ soa_counts = nullptr,
                           u32 vgrid = 0 /* rows of block_base = virtual workgroups (0: one per workgroup) */) {
  const u32 k = KC ? (u32)KC : k_arg;
  const int mode = KC ? 0 : mode_arg;
  const u32 bucket_bits = KC ? (u32)BB : bucket_bits_arg;
  const u64 sfx_mask = KC ? 0ull : sfx_mask_arg;
  static_assert(KC == 0 || MAXB >= (1 << BB), "the buckets of a constant-k form fit its tables");
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

  __shared__ u64 s_fstart[SOA ? MAXB : 1], s_fhi[SOA ? MAXB : 1];   // SOA: first k-mer of the file; byte offset of its u8 array
  // where tile position 0 WOULD go for every bucket (bucket's cursor minus its first tile position), as byte addresses: a
  // k-mer at tile position i of bucket b goes to s_ob[b] + i * (bytes per k-mer) -- one LDS read and one add per store.  Only
  // the 64-bucket instantiations keep the table (A/B-measured there, commit 400e59a); with 1024 buckets its 8 KiB would cost a
  // workgroup per CU, and the address comes from s_cursor / s_base instead
  constexpr bool OB = MAXB <= 64;
  __shared__ u64 s_ob[OB ? MAXB : 1], s_ob_hi[SOA ? MAXB : 1];
  static_assert(!SOA || OB, "the 5-byte layout is a 64-file layout");
  if constexpr (SOA) {
    for (u32 b = tid; b < nb; b += KP_BLOCK) {
      s_fstart[b] = soa_starts[b];
      if constexpr (sizeof(K) == 8) s_fhi[b] = 8ull * soa_starts[b] + 4ull * soa_counts[b];
      else                          s_fhi[b] = soa_counts[b];                  // K96 flag of the file
    }
  }

  // VIRTUAL workgroups (round 6): the input is cut into vgrid consecutive ranges with a row of cursors each; the workgroups of the
  // launch take them in turn -- blockIdx.x, + gridDim.x, ....  By default vgrid == gridDim.x (one range per workgroup: every workgroup
  // writes at its own place in all 64 files, 2048 x 128 write fronts 264 KB apart over the whole 43 GB output).  With 16384 ranges
  // (MGC_PART_VGRID) the ranges being written at any time are ~2048 CONSECUTIVE ones, ~67 MB per file: measured equal (kp_vgrid_max).
  const u32 nvirt = vgrid ? vgrid : gridDim.x;
  for (u32 vwg = blockIdx.x; vwg < nvirt; vwg += gridDim.x) {
  __syncthreads();                                                   // (the cursors of the range before are no longer read)
  for (u32 b = tid; b < nb; b += KP_BLOCK) s_cursor[b] = block_base[(u64)vwg * nb + b];
  u64 t_begin, t_end;
  kp_tile_range(num_tiles, t_begin, t_end, vwg, nvirt);

  // the next tile's bases are in flight (registers: 16 bytes per thread + the halo of the first four) while this tile is ranked,
  // exchanged and written: a tile's 4 KB used to be asked for at the top of its own iteration (round 6)
  uint4 nx = make_uint4(0, 0, 0, 0), nx_halo = make_uint4(0, 0, 0, 0);
  auto prefetch = [&](u64 tile) __attribute__((always_inline)) {
    if (tile >= t_end) return;
    nx = load16(bases, tile * KP_TILE + (u64)tid * 16, n, aligned);
    if (tid < 4) nx_halo = load16(bases, tile * KP_TILE + (u64)KP_TILE + (u64)tid * 16, n, aligned);
  };
  prefetch(t_begin);
  for (u64 tile = t_begin; tile < t_end; tile++) {
    for (u32 b = tid; b < nb; b += KP_BLOCK) s_cnt[b] = 0;
    {
      u32 c, iv;
      encode16(nx, c, iv);
      s_codes[tid] = c; s_inval[tid] = iv;
      if (tid < 4) { encode16(nx_halo, c, iv); s_codes[KP_BLOCK + tid] = c; s_inval[KP_BLOCK + tid] = iv; }
    }
    prefetch(tile + 1);
    __syncthreads();

    K keys[KP_ITEMS];
    const u32 vmask = kp_suffix_filter(keys, kp_thread_kmers(s_codes, s_inval, k, mode, keys), sfx_mask, sfx_test);
    u32 total = 0;

    if (nb == 1) {
      // plain compaction: exclusive scan of per-thread counts
      const u32 c = __popc(vmask);
      const u32 base = block_excl_scan<KP_BLOCK, u32>(c, s_tmp, &total);
      u32 o = base;
#pragma unroll
      for (int j = 0; j < KP_ITEMS; j++)
        if ((vmask >> j) & 1u) s_keys[o++] = keys[j];
      if (tid == 0) { s_base[0] = 0; s_cnt[0] = total; if constexpr (OB) s_ob[0] = reinterpret_cast<u64>(out) + (u64)sizeof(K) * s_cursor[0]; }
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
        if (b < nb) {
          s_base[b] = run;
          if constexpr (SOA && sizeof(K) == 8) {
            const u64 rel0 = s_cursor[b] - s_fstart[b] - (u64)run;             // (wraps below zero when run > what lies before: added back with i)
            s_ob[b]    = reinterpret_cast<u64>(out) + 8ull * s_fstart[b] + 4ull * rel0;
            s_ob_hi[b] = reinterpret_cast<u64>(out) + s_fhi[b] + rel0;
          } else if constexpr (SOA) {                                          // K96 records at the start of the file's 16-byte-per-k-mer region
            const u64 rel0 = s_cursor[b] - s_fstart[b] - (u64)run;
            // (bit 0 of the address word carries the file's K96 flag: one LDS read per store, as in the other layouts)
            s_ob[b] = (reinterpret_cast<u64>(out) + 16ull * s_fstart[b] + (s_fhi[b] ? 12ull : 16ull) * rel0) | (s_fhi[b] ? 1ull : 0ull);
          } else if constexpr (OB) {
            s_ob[b] = reinterpret_cast<u64>(out) + (u64)sizeof(K) * (s_cursor[b] - (u64)run);
          }
        }
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
      if constexpr (SOA && sizeof(K) == 8) {
        // (the high bytes four at a time -- a quad of consecutive tile positions lies in one file except where two files
        // meet: one unaligned 4-byte store instead of four byte stores -- measured SLOWER: 28.4 ms against 25.2, the
        // hardware splits a byte-aligned dword store anyway and the quad loop reads the tile from LDS once more)
        kp_gstore32(s_ob[b] + 4ull * i, (u32)KeyOps<K>::low64(key));
        kp_gstore8(s_ob_hi[b] + (u64)i, (uint8_t)(KeyOps<K>::low64(key) >> 32));
      }
      else if constexpr (SOA) {
        // the low twelve bytes are the same in both layouts; a 16-byte file's k-mer gets its top word too
        const u64 ob = s_ob[b];
        const bool f96 = (ob & 1ull) != 0;
        const u64 addr = (ob & ~1ull) + (u64)(f96 ? 12u : 16u) * i;
        kp_gstore96(addr, (u32)key.lo, (u32)(key.lo >> 32), (u32)key.hi);
        if (!f96) kp_gstore32(addr + 12ull, (u32)(key.hi >> 32));
      }
      else if constexpr (OB) {
        if constexpr (sizeof(K) == 8) kp_gstore64(s_ob[b] + 8ull * i, (u64)KeyOps<K>::low64(key));
        else *reinterpret_cast<K *>(s_ob[b] + (u64)sizeof(K) * i) = key;
      }
      else out[s_cursor[b] + (u64)(i - s_base[b])] = key;
    }
    __syncthreads();
    for (u32 b = tid; b < nb; b += KP_BLOCK) s_cursor[b] += s_cnt[b];
    __syncthreads();
  }
  }
}

// the constant-k instantiations (KC): canonical counts at the k of the BASELINE configs (on: Switches::const_k)
static int kmer_const_k(uint32_t k, int mode, bool on) {
  if (!on || mode != 0) return 0;
  return (k == 21 || k == 31 || k == 51) ? (int)k : 0;
}

// rows of the per-workgroup histogram = VIRTUAL workgroups = consecutive ranges of the input (kmer_partition_kernel): up to
// KP_VGRID (buckets of more than eight bits: 2048, their rows are 8 KiB each); the kernels launch at most KP_PHYS workgroups
constexpr uint32_t KP_VGRID = 16384, KP_PHYS = 2048;
static uint32_t kp_vgrid_max(uint32_t bucket_bits) {
  // default: one range per workgroup.  MGC_PART_VGRID=16384 (A/B, profiles/r06_ab_runs.txt r06v2 / r06v9): partition 20.2 -> 19.9 ms, histogram
  // 6.0 -> 6.1, the step equal -- the 23.5 ms this was built against turned out to be a property of one BOX (every process on it), not of
  // the memory layout of a process
  static const uint32_t env = [] { const char *e = getenv("MGC_PART_VGRID"); return (e && *e) ? (uint32_t)atoi(e) : 0u; }();
  if (bucket_bits > 8) return KP_PHYS;
  return (env >= 256 && env <= KP_VGRID) ? env : KP_PHYS;
}
uint32_t kp_grid_size(uint64_t n_bases, uint32_t bucket_bits) {
  const uint64_t num_tiles = (n_bases + KP_TILE - 1) / KP_TILE;
  const uint64_t vmax = kp_vgrid_max(bucket_bits);
  uint64_t g = num_tiles < vmax ? num_tiles : vmax;
  return (uint32_t)(g ? g : 1);
}

size_t kp_workspace_bytes(uint32_t bucket_bits) {
  return (size_t)(bucket_bits > 8 ? KP_PHYS : KP_VGRID) * ((size_t)1 << bucket_bits) * sizeof(uint64_t);
}

hipError_t launch_kmer_histogram(const uint8_t *d_bases, uint64_t n_bases, uint32_t k, int mode,
                                 uint32_t bucket_bits, uint64_t *d_bucket_counts, void *d_ws, hipStream_t st,
                                 uint64_t sfx_mask, uint64_t sfx_test) {
  const uint32_t nb = 1u << bucket_bits;
  MGC_CHECK(hipMemsetAsync(d_bucket_counts, 0, sizeof(uint64_t) * nb, st));
  if (n_bases == 0) return hipSuccess;
  const uint64_t num_tiles = (n_bases + KP_TILE - 1) / KP_TILE;
  const uint32_t grid = kp_grid_size(n_bases, bucket_bits);              // one workgroup per row (virtual workgroup)
  if (k <= 32)
    hipLaunchKernelGGL(kmer_hist_kernel<u64>, dim3(grid), dim3(KP_BLOCK), 0, st,
                       d_bases, (u64)n_bases, k, mode, bucket_bits, (u64)num_tiles,
                       reinterpret_cast<u64 *>(d_ws), reinterpret_cast<u64 *>(d_bucket_counts), (u64)sfx_mask, (u64)sfx_test);
  else
    hipLaunchKernelGGL(kmer_hist_kernel<K128>, dim3(grid), dim3(KP_BLOCK), 0, st,
                       d_bases, (u64)n_bases, k, mode, bucket_bits, (u64)num_tiles,
                       reinterpret_cast<u64 *>(d_ws), reinterpret_cast<u64 *>(d_bucket_counts), (u64)sfx_mask, (u64)sfx_test);
  return hipGetLastError();
}

bool kmer_histogram_fine_ok(uint32_t k, uint32_t bucket_bits, uint64_t sfx_mask, const Switches &sw) {
  return sw.fine_hist && k <= 64 && 2 * k >= (uint32_t)KH_FINE_BITS + 2 && bucket_bits == 6 && sfx_mask == 0;
}
// (the bare operator of a sharded count's senders: buckets of 6..8 bits)
bool kmer_histogram_fine_bits_ok(uint32_t k, uint32_t bucket_bits) { return k <= 64 && 2 * k >= (uint32_t)KH_FINE_BITS + 2 && bucket_bits >= 6 && bucket_bits <= 8; }

// launch_kmer_histogram + d_fine_hist[2^15] (zeroed here): k-mers per (file, next nine bits)
hipError_t launch_kmer_histogram_fine(const uint8_t *d_bases, uint64_t n_bases, uint32_t k, int mode, uint64_t *d_bucket_counts,
                                      uint64_t *d_fine_hist, void *d_ws, hipStream_t st, bool const_k, uint32_t bucket_bits) {
  if (bucket_bits < 6 || bucket_bits > 8) return hipErrorInvalidValue;
  MGC_CHECK(hipMemsetAsync(d_bucket_counts, 0, sizeof(uint64_t) << bucket_bits, st));
  MGC_CHECK(hipMemsetAsync(d_fine_hist, 0, sizeof(uint64_t) << KH_FINE_BITS, st));
  if (n_bases == 0) return hipSuccess;
  const uint64_t num_tiles = (n_bases + KP_TILE - 1) / KP_TILE;
  const uint32_t vgrid = kp_grid_size(n_bases, bucket_bits);
  const uint32_t nvp = std::max<uint32_t>((uint32_t)KH_NV, (vgrid + 511u) / 512u);     // <= 512 workgroups (each clears and flushes a 128 KiB table), nvp rows each
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kmer_hist_fine_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(u32) << KH_FINE_BITS));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kmer_hist_fine_kernel<0, 21>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(u32) << KH_FINE_BITS));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kmer_hist_fine_kernel<0, 31>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(u32) << KH_FINE_BITS));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kmer_hist_fine_kernel<0, 51>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(u32) << KH_FINE_BITS));
    attr_done = true;
  }
#define MGC_KH_LAUNCH(KC_)                                                                                                             \
  hipLaunchKernelGGL((kmer_hist_fine_kernel<0, KC_>), dim3((vgrid + nvp - 1) / nvp), dim3(KP_BLOCK * KH_NV), sizeof(u32) << KH_FINE_BITS, st, \
                     d_bases, (u64)n_bases, k, mode, (u64)num_tiles, vgrid, reinterpret_cast<u64 *>(d_ws),                             \
                     reinterpret_cast<u64 *>(d_bucket_counts), reinterpret_cast<u64 *>(d_fine_hist), bucket_bits, nvp)
  const int kc = kmer_const_k(k, mode, const_k);
  if (kc == 21) MGC_KH_LAUNCH(21); else if (kc == 31) MGC_KH_LAUNCH(31); else if (kc == 51) MGC_KH_LAUNCH(51); else MGC_KH_LAUNCH(0);
#undef MGC_KH_LAUNCH
  return hipGetLastError();
}

// `compress`: the same kernel with the table indexed by dense ranks (kmer_hist_fine_kernel<HB>): bucket_bits 6 or 8,
// d_fine_hist[kmer_histogram_hpc_entries(bucket_bits)] = k-mers per (bucket, first grouping digit of the bucket)
bool kmer_histogram_hpc_ok(uint32_t k, uint32_t bucket_bits, uint64_t sfx_mask, const Switches &sw) {
  return sw.hpc_msd && (bucket_bits == 6 || bucket_bits == 8) && k <= 64 && 2 * k >= bucket_bits + 10 + 2 && sfx_mask == 0;
}
uint32_t kmer_histogram_hpc_entries(uint32_t bucket_bits) { return hpc_table_size((int)(bucket_bits / 2 + 5)); }

hipError_t launch_kmer_histogram_hpc(const uint8_t *d_bases, uint64_t n_bases, uint32_t k, int mode, uint32_t bucket_bits,
                                     uint64_t *d_bucket_counts, uint64_t *d_fine_hist, void *d_ws, hipStream_t st, bool const_k) {
  if (bucket_bits != 6 && bucket_bits != 8) return hipErrorInvalidValue;
  const uint32_t entries = kmer_histogram_hpc_entries(bucket_bits);
  MGC_CHECK(hipMemsetAsync(d_bucket_counts, 0, sizeof(uint64_t) << bucket_bits, st));
  MGC_CHECK(hipMemsetAsync(d_fine_hist, 0, sizeof(uint64_t) * entries, st));
  if (n_bases == 0) return hipSuccess;
  const uint64_t num_tiles = (n_bases + KP_TILE - 1) / KP_TILE;
  const uint32_t vgrid = kp_grid_size(n_bases, bucket_bits);
  const uint32_t nvp = std::max<uint32_t>((uint32_t)KH_NV, (vgrid + 511u) / 512u);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kmer_hist_fine_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(u32) * hpc_table_size(8)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kmer_hist_fine_kernel<9>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(u32) * hpc_table_size(9)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kmer_hist_fine_kernel<8, 31>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(u32) * hpc_table_size(8)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kmer_hist_fine_kernel<9, 31>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(u32) * hpc_table_size(9)));
    attr_done = true;
  }
  const bool k31 = kmer_const_k(k, mode, const_k) == 31;
  if (k31 && bucket_bits == 6)
    hipLaunchKernelGGL((kmer_hist_fine_kernel<8, 31>), dim3((vgrid + nvp - 1) / nvp), dim3(KP_BLOCK * KH_NV), sizeof(u32) * entries, st,
                       d_bases, (u64)n_bases, k, mode, (u64)num_tiles, vgrid, reinterpret_cast<u64 *>(d_ws),
                       reinterpret_cast<u64 *>(d_bucket_counts), reinterpret_cast<u64 *>(d_fine_hist), 6u, nvp);
  else if (k31)
    hipLaunchKernelGGL((kmer_hist_fine_kernel<9, 31>), dim3((vgrid + nvp - 1) / nvp), dim3(KP_BLOCK * KH_NV), sizeof(u32) * entries, st,
                       d_bases, (u64)n_bases, k, mode, (u64)num_tiles, vgrid, reinterpret_cast<u64 *>(d_ws),
                       reinterpret_cast<u64 *>(d_bucket_counts), reinterpret_cast<u64 *>(d_fine_hist), 6u, nvp);
  else if (bucket_bits == 6)
    hipLaunchKernelGGL(kmer_hist_fine_kernel<8>, dim3((vgrid + nvp - 1) / nvp), dim3(KP_BLOCK * KH_NV), sizeof(u32) * entries, st,
                       d_bases, (u64)n_bases, k, mode, (u64)num_tiles, vgrid, reinterpret_cast<u64 *>(d_ws),
                       reinterpret_cast<u64 *>(d_bucket_counts), reinterpret_cast<u64 *>(d_fine_hist), 6u, nvp);
  else
    hipLaunchKernelGGL(kmer_hist_fine_kernel<9>, dim3((vgrid + nvp - 1) / nvp), dim3(KP_BLOCK * KH_NV), sizeof(u32) * entries, st,
                       d_bases, (u64)n_bases, k, mode, (u64)num_tiles, vgrid, reinterpret_cast<u64 *>(d_ws),
                       reinterpret_cast<u64 *>(d_bucket_counts), reinterpret_cast<u64 *>(d_fine_hist), 6u, nvp);
  return hipGetLastError();
}

hipError_t launch_kmer_partition(const uint8_t *d_bases, uint64_t n_bases, uint32_t k, int mode,
                                 uint32_t bucket_bits, const uint64_t *d_bucket_starts, void *d_keys,
                                 void *d_ws, hipStream_t st, uint64_t sfx_mask, uint64_t sfx_test, const uint64_t *d_soa_counts, bool const_k) {
  if (n_bases == 0) return hipSuccess;
  const uint32_t nb = 1u << bucket_bits;
  const uint64_t num_tiles = (n_bases + KP_TILE - 1) / KP_TILE;
  const uint32_t rows = kp_grid_size(n_bases, bucket_bits);                // virtual workgroups: rows of cursors
  const uint32_t grid = rows < KP_PHYS ? rows : KP_PHYS;                 // workgroups launched: they take the rows in turn
  hipLaunchKernelGGL(kmer_scan_kernel, dim3(nb), dim3(256), 0, st,
                     reinterpret_cast<u64 *>(d_ws), rows, nb, reinterpret_cast<const u64 *>(d_bucket_starts));
  MGC_CHECK(hipGetLastError());
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kmer_partition_kernel<K128, 64>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(KP_TILE * sizeof(K128)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kmer_partition_kernel<K128, KP_MAX_BUCKETS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(KP_TILE * sizeof(K128)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kmer_partition_kernel<K128, 64, false, 51>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(KP_TILE * sizeof(K128)));
    attr_done = true;
  }
#define MGC_KP_LAUNCH(K_, MAXB_)                                                                                   \
  hipLaunchKernelGGL((kmer_partition_kernel<K_, MAXB_>), dim3(grid), dim3(KP_BLOCK), KP_TILE * sizeof(K_), st,      \
                     d_bases, (u64)n_bases, k, mode, bucket_bits, (u64)num_tiles,                                  \
                     reinterpret_cast<const u64 *>(d_ws), reinterpret_cast<K_ *>(d_keys), (u64)sfx_mask, (u64)sfx_test, (const u64 *)nullptr, (const u64 *)nullptr, rows)
  if (d_soa_counts && k > 32) {                                       // K96 records (k = 33..51)
    if (!(k <= 51 && nb == 64 && sfx_mask == 0)) return hipErrorInvalidValue;
    static bool a96 = false;
    if (!a96) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kmer_partition_kernel<K128, 64, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(KP_TILE * sizeof(K128)));
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kmer_partition_kernel<K128, 64, true, 51>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(KP_TILE * sizeof(K128)));
      a96 = true;
    }
    if (kmer_const_k(k, mode, const_k) == 51)
      hipLaunchKernelGGL((kmer_partition_kernel<K128, 64, true, 51>), dim3(grid), dim3(KP_BLOCK), KP_TILE * sizeof(K128), st, d_bases, (u64)n_bases, k, mode,
                         bucket_bits, (u64)num_tiles, reinterpret_cast<const u64 *>(d_ws), reinterpret_cast<K128 *>(d_keys), (u64)0, (u64)0,
                         reinterpret_cast<const u64 *>(d_bucket_starts), reinterpret_cast<const u64 *>(d_soa_counts), rows);
    else
      hipLaunchKernelGGL((kmer_partition_kernel<K128, 64, true>), dim3(grid), dim3(KP_BLOCK), KP_TILE * sizeof(K128), st, d_bases, (u64)n_bases, k, mode,
                         bucket_bits, (u64)num_tiles, reinterpret_cast<const u64 *>(d_ws), reinterpret_cast<K128 *>(d_keys), (u64)0, (u64)0,
                         reinterpret_cast<const u64 *>(d_bucket_starts), reinterpret_cast<const u64 *>(d_soa_counts), rows);
    return hipGetLastError();
  }
  if (d_soa_counts) {                                                 // 5-byte layout (kmer_partition_soa_ok)
    if (!(k <= 32 && nb == 64 && sfx_mask == 0)) return hipErrorInvalidValue;
    if (kmer_const_k(k, mode, const_k) == 21)
      hipLaunchKernelGGL((kmer_partition_kernel<u64, 64, true, 21>), dim3(grid), dim3(KP_BLOCK), KP_TILE * sizeof(u64), st, d_bases, (u64)n_bases, k, mode,
                         bucket_bits, (u64)num_tiles, reinterpret_cast<const u64 *>(d_ws), reinterpret_cast<u64 *>(d_keys), (u64)0, (u64)0,
                         reinterpret_cast<const u64 *>(d_bucket_starts), reinterpret_cast<const u64 *>(d_soa_counts), rows);
    else
      hipLaunchKernelGGL((kmer_partition_kernel<u64, 64, true>), dim3(grid), dim3(KP_BLOCK), KP_TILE * sizeof(u64), st, d_bases, (u64)n_bases, k, mode,
                         bucket_bits, (u64)num_tiles, reinterpret_cast<const u64 *>(d_ws), reinterpret_cast<u64 *>(d_keys), (u64)0, (u64)0,
                         reinterpret_cast<const u64 *>(d_bucket_starts), reinterpret_cast<const u64 *>(d_soa_counts), rows);
    return hipGetLastError();
  }
#define MGC_KPC_LAUNCH(K_, KC_)                                                                                    \
  hipLaunchKernelGGL((kmer_partition_kernel<K_, 64, false, KC_>), dim3(grid), dim3(KP_BLOCK), KP_TILE * sizeof(K_), st, \
                     d_bases, (u64)n_bases, k, mode, bucket_bits, (u64)num_tiles,                                  \
                     reinterpret_cast<const u64 *>(d_ws), reinterpret_cast<K_ *>(d_keys), (u64)0, (u64)0, (const u64 *)nullptr, (const u64 *)nullptr, rows)
  const int kc = (nb == 64 && sfx_mask == 0) ? kmer_const_k(k, mode, const_k) : 0;
  if (nb == 256 && sfx_mask == 0 && kmer_const_k(k, mode, const_k) == 31)          // (k = 31 `compress` beyond ~4 Gbp: 256 buckets, 4 KiB of tables instead of 16)
    hipLaunchKernelGGL((kmer_partition_kernel<u64, 256, false, 31, 8>), dim3(grid), dim3(KP_BLOCK), KP_TILE * sizeof(u64), st,
                       d_bases, (u64)n_bases, k, mode, bucket_bits, (u64)num_tiles, reinterpret_cast<const u64 *>(d_ws), reinterpret_cast<u64 *>(d_keys), (u64)0, (u64)0, (const u64 *)nullptr, (const u64 *)nullptr, rows);
  else if (kc == 21) MGC_KPC_LAUNCH(u64, 21);
  else if (kc == 31) MGC_KPC_LAUNCH(u64, 31);
  else if (kc == 51) MGC_KPC_LAUNCH(K128, 51);
  else if (k <= 32) { if (nb <= 64) MGC_KP_LAUNCH(u64, 64); else MGC_KP_LAUNCH(u64, KP_MAX_BUCKETS); }
  else              { if (nb <= 64) MGC_KP_LAUNCH(K128, 64); else MGC_KP_LAUNCH(K128, KP_MAX_BUCKETS); }
#undef MGC_KPC_LAUNCH
#undef MGC_KP_LAUNCH
  return hipGetLastError();
}



// (code objects load lazily, at the first launch of any kernel of a translation unit: a fresh process pays that inside its
// first count.  warm_*: touch one kernel per unit -- mgc_prepare's helper thread does it while the input is being read)
hipError_t warm_kmer() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&kmer_scan_kernel)); }

}  // namespace mgc
