// mgc_misc.hip -- homopolymer compression and synthetic reads (gfx950).
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

// Round 5: the kept bytes of a tile are packed in LDS first and leave as aligned dwords (a head and a tail of up to three single
// bytes around them) -- one byte store per kept base and lane, sixteen per thread, each spread over ~700 bytes of the output, was
// 5.0 ms per 5 Gbp (profiles/r05l kernel stats of the `compress` leg); the packed form writes 256 contiguous bytes per wave instruction.
__global__ __launch_bounds__(HP_BLOCK)
void hpc_emit_kernel(const uint8_t *__restrict__ in, u64 n, const u64 *__restrict__ tile_offs, uint8_t *__restrict__ out) {
  __shared__ u32 s_tmp[HP_BLOCK / 64 + 1];
  __shared__ __attribute__((aligned(16))) uint8_t s_out[HP_TILE + 16];
  const bool aligned = ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
  const u32 tid = threadIdx.x;
  uint4 v;
  const u32 keep = hp_keep_mask(in, n, (u64)blockIdx.x * HP_TILE + (u64)tid * 16, aligned, v);
  u32 tot;
  const u32 base = block_excl_scan<HP_BLOCK, u32>(__popc(keep), s_tmp, &tot);
  const u32 w[4] = { v.x, v.y, v.z, v.w };
  u32 o = base;
#pragma unroll
  for (int i = 0; i < 16; i++)
    if ((keep >> i) & 1u) s_out[o++] = (uint8_t)((w[i >> 2] >> (8 * (i & 3))) & 0xFFu);
  // (a funnel shift below may read the word that holds the last kept byte and garbage behind it: those bits are shifted out)
  __syncthreads();
  const u64 g0 = tile_offs[blockIdx.x];                     // where the tile's kept bytes go
  uint8_t *dst = out + g0;
  const u32 mis = (u32)(reinterpret_cast<uintptr_t>(dst) & 3u);
  u32 head = mis ? 4u - mis : 0u;
  if (head > tot) head = tot;
  if (tid < head) dst[tid] = s_out[tid];
  const u32 nd = (tot - head) >> 2;                         // whole, aligned dwords of the output
  const u32 *s32 = reinterpret_cast<const u32 *>(s_out);
  for (u32 j = tid; j < nd; j += HP_BLOCK) {
    const u32 p = head + 4u * j, a = p >> 2, sh = (p & 3u) * 8u;
    const u32 lo = s32[a], hi = s32[a + 1];
    reinterpret_cast<u32 *>(dst + p)[0] = sh ? ((lo >> sh) | (hi << (32u - sh))) : lo;
  }
  const u32 tail0 = head + 4u * nd;
  if (tid < tot - tail0) dst[tail0 + tid] = s_out[tail0 + tid];
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
  MGC_CHECK(scan_u64_exclusive(tile_offs, t, scratch, total, st));
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
  u64 s_rep, rep_thresh;
  u32 read_len, rep_unit, rep_families;
};

// repeat families: oracle_count.c synth_genome_pos, same integer arithmetic
__device__ __forceinline__ u64 synth_genome_pos(const SynthParams &P, u64 gpos) {
  if (P.rep_thresh == 0) return gpos;
  const u64 blk = gpos / P.rep_unit;
  const u64 h   = splitmix64(P.s_rep ^ blk);
  if ((u64)(u32)h >= P.rep_thresh) return gpos;
  const u64 u = h >> 32;
  const u64 a = (u * u) >> 32;
  const u64 b = (a * u) >> 32;
  const u64 fam = (b * (u64)P.rep_families) >> 32;
  return (1ull << 62) + fam * (u64)P.rep_unit + (gpos - blk * P.rep_unit);
}

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
  u32 code = (u32)(splitmix64(P.s_genome ^ synth_genome_pos(P, gpos)) & 3ull);
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
                              uint32_t repeat_ppm, uint32_t repeat_unit, uint32_t repeat_families,
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
  P.s_rep       = splitmix64(seed + 3ull * 0x632be59bd9b4e019ull);
  P.rep_thresh  = (uint64_t)repeat_ppm * 4294967296ull / 1000000ull;
  P.rep_unit    = repeat_unit ? repeat_unit : 1;
  P.rep_families = repeat_families ? repeat_families : 1;
  const uint64_t threads = (P.total_bytes + 3) / 4;
  hipLaunchKernelGGL(synth_reads_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, st, P, d_out);
  return hipGetLastError();
}


}  // namespace mgc
