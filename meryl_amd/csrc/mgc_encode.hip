// mgc_encode.hip -- database blocks encoded ON THE DEVICE (gfx950).
//
// Replaces, for the count path, the host loop of merylBlockWriter::addBlock [meryl-utility, not in tree; call
// site src/meryl/merylCountArray.C:472-475, 64 host threads in src/meryl/merylOp-countThreads.C:452-459]: the
// sorted (k-mer, count) stream in HBM becomes the bytes of the .merylData files without a host pass over the
// k-mers.  The byte layout is the one meryl_db.cpp writes (assumptions A1..A10 there, constants and closed
// forms in mdb_layout.h); tests/test_db_device.py checks device bytes == host-encoder bytes.
//
// Why this is a flat, scan-free kernel: a block stores k-mer i as  unary(top_i - top_{i-1}) , binary(low bits).
// The unary deltas telescope, so the '1' that ends k-mer i's unary code sits at bit
//     528 + top_i + i * (1 + binaryBits)
// of the block -- a closed form of the k-mer's own value and index.  Values (32 bits each) follow at
//     528 + top_last + n * (1 + binaryBits) + 32 i.
// Every thread takes a run of consecutive k-mers, assembles their bits in a 64-bit accumulator and writes
// whole words; only the first and last word of a run can be shared with a neighbour and go out as atomic ORs
// onto the zeroed image.  HBM-bound integer work: 12 (28) B read + ~(suffix+34)/8 B written per distinct k-mer.
#include "mgc_common.hpp"
#include "mdb_layout.h"

namespace mgc {

using namespace mdb;

template <typename K> struct SufOps;
template <> struct SufOps<u64> {
  // suffix = low `ss` bits of the k-mer (ss <= 58 for 8-byte keys); top = suffix >> bb; low = suffix's low bb bits
  static __device__ __forceinline__ void split(u64 key, u32 ss, u32 bb, u64 &top, u64 &low_hi, u64 &low_lo) {
    const u64 suf = (ss >= 64) ? key : (key & ((1ull << ss) - 1ull));
    top = (bb >= 64) ? 0ull : (suf >> bb);
    low_lo = (bb >= 64) ? suf : (suf & ((1ull << bb) - 1ull));
    low_hi = 0;
  }
};
template <> struct SufOps<K128> {
  static __device__ __forceinline__ void split(K128 key, u32 ss, u32 bb, u64 &top, u64 &low_hi, u64 &low_lo) {
    u128 suf = KeyOps<K128>::v(key);
    if (ss < 128) suf &= (((u128)1 << ss) - 1);
    top = (bb >= 128) ? 0ull : (u64)(suf >> bb);                    // unaryBits <= 64: fits
    const u128 low = (bb >= 128) ? suf : (suf & (((u128)1 << bb) - 1));
    low_lo = (u64)low;
    low_hi = (u64)(low >> 64);
  }
};

// ---- block starts of a prefix range -------------------------------------------------------------
// rel_start[i] = first key >= (prefix_begin + i) << w_data, i in [0, n_blocks].  The keys are supposed to hold only prefixes
// of the range: the caller checks rel_start[0] == 0 and rel_start[n_blocks] == n (the last boundary is searched too unless
// it is the end of the key space, whose floor does not fit the key type when 2k is a multiple of 64)
template <typename K>
__global__ void block_offsets_range_kernel(const K *__restrict__ keys, u64 n, u32 w_data, u64 prefix_begin, u64 n_blocks,
                                           u64 n_prefix_total, u64 *__restrict__ rel_start) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_blocks) return;
  if (i == n_blocks && prefix_begin + i >= n_prefix_total) { rel_start[i] = n; return; }
  const K target = KeyOps<K>::prefix_floor(prefix_begin + i, w_data);
  u64 lo = 0, hi = n;
  while (lo < hi) {
    const u64 mid = lo + ((hi - lo) >> 1);
    if (KeyOps<K>::lt(keys[mid], target)) lo = mid + 1; else hi = mid;
  }
  rel_start[i] = lo;
}

// ---- per-block geometry: binary bits, end of the k-mer section, dumped size ----------------------
template <typename K>
__global__ void encode_sizes_kernel(const K *__restrict__ keys, const u64 *__restrict__ bs, u64 n_blocks, u32 ss, u32 ls,
                                    u64 *__restrict__ blk_bytes, u64 *__restrict__ blk_vbase, u32 *__restrict__ blk_bb) {
  const u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_blocks) return;
  const u64 g0 = bs[b], g1 = bs[b + 1], n = g1 - g0;
  const u32 ub = unary_bits_for(n, ss), bb = ss - ub;
  u64 top_last = 0, lh, ll;
  if (n) SufOps<K>::split(keys[g1 - 1], ss, bb, top_last, lh, ll);
  blk_bb[b] = bb;
  blk_vbase[b] = BLOCK_HEADER_BITS + top_last + n * (u64)(1 + bb);
  blk_bytes[b] = stuffed_bytes(block_bits(n, top_last, bb, ls));
}

// ---- stuffedBits framing + the 528-bit block header ----------------------------------------------
__global__ void encode_headers_kernel(const u64 *__restrict__ bs, const u64 *__restrict__ blk_pos, const u64 *__restrict__ blk_vbase,
                                      const u32 *__restrict__ blk_bb, u64 b0, u64 b1, u64 prefix_of_block0, u32 ss, u32 ls,
                                      unsigned char *__restrict__ img) {
  const u64 b = b0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= b1) return;
  const u64 n = bs[b + 1] - bs[b];
  const u32 bb = blk_bb[b], ub = ss - bb;
  const u64 bits = blk_vbase[b] + n * (u64)(VALUE_BITS + ls);
  const u64 nsb = stuffed_sub_blocks(bits);
  u64 *o = reinterpret_cast<u64 *>(img + blk_pos[b]);
  o[0] = STUFFED_BLOCK_BITS;
  const u64 nmax = nsb > 64 ? nsb : 64;
  o[1] = (nsb & 0xffffffffull) | (nmax << 32);                      // u32 nBlocks, u32 nBlocksMax
  for (u64 i = 0; i < nsb; i++) {
    const u64 bgn = i * STUFFED_BLOCK_BITS;
    const u64 len = (bits - bgn < STUFFED_BLOCK_BITS) ? (bits - bgn) : STUFFED_BLOCK_BITS;
    o[2 + i] = bgn;
    o[2 + nsb + i] = len;
    u64 *sb = reinterpret_cast<u64 *>(img + blk_pos[b] + stuffed_word_offset(nsb, i * STUFFED_BLOCK_WORDS)) - 2;
    sb[0] = (len + 63) / 64;
    sb[1] = STUFFED_BLOCK_WORDS;
  }
  // A4: header words 0..7 belong to the header alone (bits 512..527 are the low bits of c2 = 0 and share word 8 with
  // the first k-mer: nothing to write there)
  u64 *w = reinterpret_cast<u64 *>(img + blk_pos[b] + stuffed_word_offset(nsb, 0));
  w[0] = MAGIC_DAT1; w[1] = MAGIC_DAT2; w[2] = prefix_of_block0 + b; w[3] = n;
  // bits 256..: kCode(8)=1, unaryBits(32), binaryBits(32), k1(64)=0, cCode(8)=1, c1(64)=0, c2(64)=0
  w[4] = (1ull << 56) | ((u64)ub << 24) | ((u64)bb >> 8);
  w[5] = ((u64)bb & 0xffull) << 56;                                  // k1 = 0 fills the rest and 8 bits of word 6
  w[6] = 1ull << 48;                                                 // bits 392..399 = cCode = 1 -> word 6, bits 8..15
  w[7] = 0;
}

// ---- the bit-run writer ---------------------------------------------------------------------------
struct RunWriter {
  unsigned char *blk;       // image address of the current block's dump
  u64 nsb;
  u64 cur;                  // word index in the block's bit stream that `acc` holds; ~0 = none
  u64 acc;
  bool first;               // cur is the first word of this segment (may be shared with the previous run)
  __device__ __forceinline__ void begin(unsigned char *b, u64 nsb_) { blk = b; nsb = nsb_; cur = ~0ull; acc = 0; first = true; }
  __device__ __forceinline__ u64 *addr(u64 w) const { return reinterpret_cast<u64 *>(blk + stuffed_word_offset(nsb, w)); }
  __device__ __forceinline__ void flush_inner() {
    if (cur == ~0ull) return;
    if (first) atomicOr(reinterpret_cast<unsigned long long *>(addr(cur)), (unsigned long long)acc);
    else       *addr(cur) = acc;
    first = false;
  }
  // width in 1..64; positions never decrease within a segment
  __device__ __forceinline__ void put(u64 pos, u32 width, u64 value) {
    const u64 w = pos >> 6;
    const u32 off = (u32)(pos & 63), room = 64 - off;
    if (w != cur) { flush_inner(); cur = w; acc = 0; }
    if (width <= room) {
      acc |= value << (room - width);
    } else {
      acc |= value >> (width - room);
      flush_inner();
      cur = w + 1;
      acc = value << (64 - (width - room));
    }
  }
  // the last word of a segment may be shared with the next run: atomic
  __device__ __forceinline__ void end_segment() {
    if (cur != ~0ull) atomicOr(reinterpret_cast<unsigned long long *>(addr(cur)), (unsigned long long)acc);
    cur = ~0ull; acc = 0; first = true;
  }
};

template <typename K> struct EncRun { static constexpr int R = 16; };
template <> struct EncRun<K128>     { static constexpr int R = 8; };
constexpr int ENC_BLOCK = 256;

// One thread = R consecutive k-mers of the chunk [bs[b0], bs[b1]).
template <typename K>
__global__ __launch_bounds__(ENC_BLOCK)
void encode_kmers_kernel(const K *__restrict__ keys, const u32 *__restrict__ counts, const u64 *__restrict__ bs,
                         const u64 *__restrict__ blk_pos, const u64 *__restrict__ blk_vbase, const u32 *__restrict__ blk_bb,
                         u64 b0, u64 b1, u32 ss, u32 ls, u64 label, unsigned char *__restrict__ img) {
  constexpr int R = EncRun<K>::R;
  const u64 g_begin = bs[b0], g_end = bs[b1];
  const u64 i0 = g_begin + ((u64)blockIdx.x * ENC_BLOCK + threadIdx.x) * R;
  if (i0 >= g_end) return;
  const u64 i1 = (i0 + R < g_end) ? (i0 + R) : g_end;
  K   kreg[R];
  u32 creg[R];
#pragma unroll
  for (int q = 0; q < R; q++) {
    const u64 i = i0 + q;
    kreg[q] = (i < i1) ? keys[i] : KeyOps<K>::zero();
    creg[q] = (i < i1) ? counts[i] : 0u;
  }
  // block of the first k-mer: the last b in [b0, b1) with bs[b] <= i0 (it is not empty: bs[b+1] > i0)
  u64 lo = b0, hi = b1;
  while (hi - lo > 1) {
    const u64 mid = lo + ((hi - lo) >> 1);
    if (bs[mid] <= i0) lo = mid; else hi = mid;
  }
  u64 b = lo;
  RunWriter rw;
  u64 i = i0;
  while (i < i1) {
    while (bs[b + 1] <= i) b++;                                      // skips empty blocks
    const u64 bstart = bs[b], bend = bs[b + 1];
    const u64 jend = (bend < i1) ? bend : i1;                        // this thread's k-mers of block b: [i, jend)
    const u32 bb = blk_bb[b];
    const u64 vbase = blk_vbase[b];
    const u64 n = bend - bstart;
    const u64 bits = vbase + n * (u64)(VALUE_BITS + ls);
    rw.begin(img + blk_pos[b], stuffed_sub_blocks(bits));
    const int q0 = (int)(i - i0), q1 = (int)(jend - i0);
    // segment 1: k-mers (A5)
#pragma unroll
    for (int q = 0; q < R; q++) {
      if (q < q0 || q >= q1) continue;
      u64 top, lh, ll;
      SufOps<K>::split(kreg[q], ss, bb, top, lh, ll);
      const u64 Q = BLOCK_HEADER_BITS + top + (i0 + q - bstart) * (u64)(1 + bb);
      if (bb < 64) {
        rw.put(Q, 1 + bb, (1ull << bb) | ll);
      } else {
        rw.put(Q, 1, 1ull);
        if (bb == 64) rw.put(Q + 1, 64, ll);
        else { rw.put(Q + 1, bb - 64, lh); rw.put(Q + 1 + (bb - 64), 64, ll); }
      }
    }
    rw.end_segment();
    // segment 2: values (A6)
#pragma unroll
    for (int q = 0; q < R; q++) {
      if (q < q0 || q >= q1) continue;
      rw.put(vbase + (i0 + q - bstart) * (u64)VALUE_BITS, VALUE_BITS, (u64)creg[q]);
    }
    rw.end_segment();
    // segment 3: labels (A10)
    if (ls) {
      const u64 lbase = vbase + n * (u64)VALUE_BITS;
      const u64 lv = (ls >= 64) ? label : (label & ((1ull << ls) - 1ull));
      for (int q = q0; q < q1; q++) rw.put(lbase + (i0 + q - bstart) * (u64)ls, ls, lv);
      rw.end_segment();
    }
    i = jend;
  }
}

// ---- value histogram (the master index's, A9) ------------------------------------------------------
constexpr u32 HIST_SMALL = 1024;
__global__ __launch_bounds__(256)
void value_hist_kernel(const u32 *__restrict__ counts, u64 n, u64 *__restrict__ hist /*[HIST_SMALL]*/,
                       u32 *__restrict__ big_list, u64 big_cap, u64 *__restrict__ big_n) {
  __shared__ u32 s_h[HIST_SMALL];
  for (u32 i = threadIdx.x; i < HIST_SMALL; i += blockDim.x) s_h[i] = 0;
  __syncthreads();
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const u32 v = counts[i];
    if (v < HIST_SMALL) atomicAdd(&s_h[v], 1u);
    else {
      const u64 at = atomicAdd(reinterpret_cast<unsigned long long *>(big_n), 1ull);
      if (at < big_cap) big_list[at] = v;
    }
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < HIST_SMALL; i += blockDim.x)
    if (s_h[i]) atomicAdd(reinterpret_cast<unsigned long long *>(hist + i), (unsigned long long)s_h[i]);
}

// ---- launchers -----------------------------------------------------------------------------------
hipError_t launch_block_offsets_range(const void *d_keys, uint64_t n, uint32_t key_words, uint32_t w_data, uint64_t prefix_begin,
                                      uint64_t n_blocks, uint64_t n_prefix_total, uint64_t *d_rel_start, hipStream_t st) {
  const dim3 grid((uint32_t)((n_blocks + 1 + 255) / 256));
  if (key_words == 2)
    hipLaunchKernelGGL(block_offsets_range_kernel<K128>, grid, dim3(256), 0, st, reinterpret_cast<const K128 *>(d_keys), (u64)n,
                       w_data, (u64)prefix_begin, (u64)n_blocks, (u64)n_prefix_total, reinterpret_cast<u64 *>(d_rel_start));
  else
    hipLaunchKernelGGL(block_offsets_range_kernel<u64>, grid, dim3(256), 0, st, reinterpret_cast<const u64 *>(d_keys), (u64)n,
                       w_data, (u64)prefix_begin, (u64)n_blocks, (u64)n_prefix_total, reinterpret_cast<u64 *>(d_rel_start));
  return hipGetLastError();
}

hipError_t launch_encode_sizes(const void *d_keys, uint32_t key_words, const uint64_t *d_bs, uint64_t n_blocks, uint32_t suffix_size,
                               uint32_t label_size, uint64_t *d_blk_bytes, uint64_t *d_blk_vbase, uint32_t *d_blk_bb, hipStream_t st) {
  if (n_blocks == 0) return hipSuccess;
  const dim3 grid((uint32_t)((n_blocks + 255) / 256));
  if (key_words == 2)
    hipLaunchKernelGGL(encode_sizes_kernel<K128>, grid, dim3(256), 0, st, reinterpret_cast<const K128 *>(d_keys),
                       reinterpret_cast<const u64 *>(d_bs), (u64)n_blocks, suffix_size, label_size,
                       reinterpret_cast<u64 *>(d_blk_bytes), reinterpret_cast<u64 *>(d_blk_vbase), d_blk_bb);
  else
    hipLaunchKernelGGL(encode_sizes_kernel<u64>, grid, dim3(256), 0, st, reinterpret_cast<const u64 *>(d_keys),
                       reinterpret_cast<const u64 *>(d_bs), (u64)n_blocks, suffix_size, label_size,
                       reinterpret_cast<u64 *>(d_blk_bytes), reinterpret_cast<u64 *>(d_blk_vbase), d_blk_bb);
  return hipGetLastError();
}

// Encodes blocks [b0, b1) into d_img (zeroed by the caller; d_blk_pos = byte offset of every block inside d_img).
// n_kmers_chunk = bs[b1] - bs[b0] (the host knows it from its copy of bs).
hipError_t launch_encode_chunk(const void *d_keys, const uint32_t *d_counts, uint32_t key_words, const uint64_t *d_bs,
                               const uint64_t *d_blk_pos, const uint64_t *d_blk_vbase, const uint32_t *d_blk_bb,
                               uint64_t b0, uint64_t b1, uint64_t n_kmers_chunk, uint64_t prefix_of_block0,
                               uint32_t suffix_size, uint32_t label_size, uint64_t label, void *d_img, hipStream_t st) {
  if (b1 <= b0) return hipSuccess;
  unsigned char *img = reinterpret_cast<unsigned char *>(d_img);
  hipLaunchKernelGGL(encode_headers_kernel, dim3((uint32_t)((b1 - b0 + 255) / 256)), dim3(256), 0, st,
                     reinterpret_cast<const u64 *>(d_bs), reinterpret_cast<const u64 *>(d_blk_pos),
                     reinterpret_cast<const u64 *>(d_blk_vbase), d_blk_bb, (u64)b0, (u64)b1, (u64)prefix_of_block0,
                     suffix_size, label_size, img);
  MGC_CHECK(hipGetLastError());
  if (n_kmers_chunk == 0) return hipSuccess;
  if (key_words == 2) {
    const uint64_t per_wg = (uint64_t)ENC_BLOCK * EncRun<K128>::R;
    hipLaunchKernelGGL(encode_kmers_kernel<K128>, dim3((uint32_t)((n_kmers_chunk + per_wg - 1) / per_wg)), dim3(ENC_BLOCK), 0, st,
                       reinterpret_cast<const K128 *>(d_keys), d_counts, reinterpret_cast<const u64 *>(d_bs),
                       reinterpret_cast<const u64 *>(d_blk_pos), reinterpret_cast<const u64 *>(d_blk_vbase), d_blk_bb,
                       (u64)b0, (u64)b1, suffix_size, label_size, (u64)label, img);
  } else {
    const uint64_t per_wg = (uint64_t)ENC_BLOCK * EncRun<u64>::R;
    hipLaunchKernelGGL(encode_kmers_kernel<u64>, dim3((uint32_t)((n_kmers_chunk + per_wg - 1) / per_wg)), dim3(ENC_BLOCK), 0, st,
                       reinterpret_cast<const u64 *>(d_keys), d_counts, reinterpret_cast<const u64 *>(d_bs),
                       reinterpret_cast<const u64 *>(d_blk_pos), reinterpret_cast<const u64 *>(d_blk_vbase), d_blk_bb,
                       (u64)b0, (u64)b1, suffix_size, label_size, (u64)label, img);
  }
  return hipGetLastError();
}

uint32_t value_hist_small_bins() { return HIST_SMALL; }

// d_hist[HIST_SMALL] and *d_big_n are accumulated into (zero them first); values >= HIST_SMALL are appended to d_big_list
// (entries beyond big_cap are counted but dropped: the caller re-runs with a larger list)
hipError_t launch_value_hist(const uint32_t *d_counts, uint64_t n, uint64_t *d_hist, uint32_t *d_big_list, uint64_t big_cap,
                             uint64_t *d_big_n, hipStream_t st) {
  if (n == 0) return hipSuccess;
  uint64_t wgs = (n + 256 * 16 - 1) / (256 * 16);
  if (wgs > 2048) wgs = 2048;
  hipLaunchKernelGGL(value_hist_kernel, dim3((uint32_t)wgs), dim3(256), 0, st, d_counts, (u64)n, reinterpret_cast<u64 *>(d_hist),
                     d_big_list, (u64)big_cap, reinterpret_cast<u64 *>(d_big_n));
  return hipGetLastError();
}


hipError_t warm_encode() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&value_hist_kernel)); }

}  // namespace mgc
