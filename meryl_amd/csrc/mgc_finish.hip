// mgc_finish.hip -- run-length count, sub-bucket finish (LDS hash-count / sort), block offsets (gfx950).
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
  MGC_CHECK(scan_u64_exclusive(w.tile_offs, w.num_tiles, w.scratch, w.total, st));
  MGC_CHECK(scan_u64_min_reverse(w.tile_next, w.num_tiles, w.scratch, (u64)n, st));
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
// and the list of the non-empty ones (any order; one atomic per wave)
__global__ __launch_bounds__(256)
void subbucket_max_kernel(const u64 *__restrict__ starts, u64 ng, u64 *__restrict__ max_out, u64 threshold,
                          u32 *__restrict__ list, u64 *__restrict__ list_count,
                          u32 *__restrict__ nz, u64 *__restrict__ nz_count) {
  // one atomic per WORKGROUP on the shared counters (a file has 2^18 sub-buckets: one per wave was 4096 atomics on one address)
  __shared__ u32 s_cnt[4];
  __shared__ u64 s_max[4];
  __shared__ u64 s_base;
  const u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 sz = (v < ng) ? (starts[v + 1] - starts[v]) : 0ull;
  if (sz > threshold) list[atomicAdd(list_count, 1ull)] = (u32)v;
  const u64 m = __ballot(sz != 0);
  u64 mx = sz;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const u64 o = __shfl_down(mx, d); mx = (o > mx) ? o : mx; }
  if (lane_id() == 0) { s_cnt[wave_id()] = (u32)__popcll(m); s_max[wave_id()] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const u32 total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    u64 bm = s_max[0];
    for (int w = 1; w < 4; w++) bm = s_max[w] > bm ? s_max[w] : bm;
    s_base = total ? atomicAdd(nz_count, (u64)total) : 0ull;
    if (bm) atomicMax(max_out, bm);
  }
  __syncthreads();
  if (sz) {
    u64 pos = s_base + __popcll(m & ((1ull << lane_id()) - 1ull));
    for (u32 w = 0; w < wave_id(); w++) pos += s_cnt[w];
    nz[pos] = (u32)v;
  }
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
// NARROW: the sub-buckets hold 32-bit narrowed keys (launch_group_narrow) -- u32 loads, and the distinct SUFFIXES go back
// in place as u32 (compact_groups_narrow_kernel puts the prefix back when it packs the result).
// The distinct suffixes are ordered by a counting sort on their top eight bits (BLOCK bins, one per thread) and a brute-force rank
// inside the bin -- O(D) for spread suffixes (the all-pairs rank of round 1, D^2 / BLOCK compares per thread, was removed in round 5).
template <int BLOCK, int CAP, int SLOTS, bool DBG, bool LIST, bool NARROW = false>
__global__ __launch_bounds__(BLOCK, 7)
void hash_count_kernel(u64 *__restrict__ keys, const u64 *__restrict__ starts, u64 ng, u64 max_size, u32 low_bits,
                       u32 *__restrict__ cnt_tmp, u64 *__restrict__ group_distinct,
                       const u32 *__restrict__ nz, const u64 *__restrict__ nz_count, u64 *__restrict__ dbg,
                       u32 tr_a = 0, u32 tr_b = 0 /* tr_index() of the sub-bucket numbers */) {
  // Inside a sub-bucket the keys differ only in their low `low_bits` (< 32) bits: the table holds
  // 32-bit suffixes (half the LDS, 32-bit CAS and compares); the common prefix is added back on output.
  // Persistent workgroups: the keys of the next sub-bucket are loaded while the current one is counted
  // (a sub-bucket is ~1K keys, so the two dependent HBM round trips would otherwise be a third of its time).
  static_assert((SLOTS & (SLOTS - 1)) == 0 && SLOTS * 3 >= CAP * 4 && SLOTS % BLOCK == 0 && CAP % BLOCK == 0, "table geometry");
  constexpr int KPT = CAP / BLOCK;
  constexpr u32 EMPTY = 0xFFFFFFFFu;                   // suffixes are < 2^31
  __shared__ __attribute__((aligned(16))) u32 tk[SLOTS];
  __shared__ __attribute__((aligned(16))) u32 tc[SLOTS / 2];   // counts, two 16-bit halves per word (a count is <= CAP): 26 KiB of LDS in all, six workgroups per CU
  __shared__ __attribute__((aligned(16))) u32 dk[CAP + 16];
  __shared__ unsigned short dc[CAP];                   // slot of a claimed suffix, later its count: both below 2^16
  __shared__ u32 s_nd;                                 // distinct suffixes of the sub-bucket: the claimers of empty slots count themselves
  __shared__ u32 s_bin[BLOCK + 1];
  __shared__ u32 s_scan[BLOCK / 64 + 1];
  static_assert(BLOCK == 256, "one bin per thread, eight bits");
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
      if constexpr (NARROW) kr[j] = (nn <= max_size && idx < nn) ? reinterpret_cast<const u32 *>(keys)[aa + idx] : 0u;
      else                  kr[j] = (nn <= max_size && idx < nn) ? reinterpret_cast<const u32 *>(keys + aa + idx)[0] : 0u;
    }
  };
  const u32 group_shift = low_bits + (u32)__builtin_ctzll(ng);          // ng is a power of two
  const u64 file_base = NARROW ? 0ull : (keys[0] >> group_shift) << group_shift;

  // visit only the NON-EMPTY sub-buckets (list built by subbucket_max_kernel): a sparse key space (homopolymer-
  // compressed k-mers, small k) leaves most of the 2^t grid empty, and an empty visit still costs a memory round trip
  const u64 np = LIST ? *nz_count : ng;
  auto sub_at = [&](u64 pp) -> u64 { return (pp < np) ? (LIST ? (u64)nz[pp] : pp) : ng; };
  u64 p = blockIdx.x;
  u64 g = sub_at(p), g1 = sub_at(p + G), g2 = sub_at(p + 2 * G), a, n64, na, nn;
  u32 kcur[KPT];
  load_bounds(g, a, n64);
  load_keys(a, n64, kcur);
  load_bounds(g1, na, nn);

  u64 ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
#define HC_STAMP(i) do { if (DBG) { const u64 t = __builtin_readcyclecounter(); ph[i] += t - t0; t0 = t; } } while (0)
  while (g < ng) {
    u32 knext[KPT];
    u64 nna, nnn;
    if (DBG) t0 = __builtin_readcyclecounter();
    // (everything the previous iteration loaded is consumed BEFORE the next loads are issued: with the loads in conditional
    // blocks the compiler's wait for the old registers is vmcnt(0) -- issued after the new loads it would wait for THEM)
#pragma unroll
    for (int j = 0; j < KPT; j++) asm volatile("" : "+v"(kcur[j]) :: "memory");
    load_keys(na, nn, knext);                          // in flight while this sub-bucket is processed
    load_bounds(g2, nna, nnn);
    const u64 g3 = sub_at(p + 3 * G);

    if (n64 == 0) {
      if (tid == 0) group_distinct[tr_index(g, tr_a, tr_b)] = 0;
    } else if (n64 <= max_size) {                      // larger ones: other launches take them
      const u32 n = (u32)n64;
      const u64 prefix = file_base | (tr_index(g, tr_a, tr_b) << low_bits);
      u32 kk[KPT], hh[KPT];
      u32 pending = 0;
#pragma unroll
      for (int j = 0; j < KPT; j++) {
        const u32 idx = (u32)j * BLOCK + tid;
        u32 raw = kcur[j];
        kk[j] = raw & (u32)low_mask;
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
          if (i < slots / 8) tc4[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        if (tid == 0) s_nd = 0;
      }
      __syncthreads();
      HC_STAMP(0);

      {
#pragma unroll
        for (int j = 0; j < KPT; j++) hh[j] = (kk[j] * 0x9E3779B1u) >> sshift;
        // linear probing; one probe step of every still-pending key per round, so the CASes of a round overlap
        while (pending) {
#pragma unroll
          for (int j = 0; j < KPT; j++) {
            if ((pending >> j) & 1u) {
              const u32 old = atomicCAS(&tk[hh[j]], EMPTY, kk[j]);
              const bool won = (old == EMPTY);
              // The lanes that claimed an empty slot hold a suffix nobody has seen before: they append it (and its slot)
              // to the compact list right here -- one LDS atomic per wave and round for all of them -- so that no pass
              // over the table and no scan is needed afterwards.
              const u64 wm = __ballot(won);
              if (won) {
                const u32 lane = tid & 63u;
                const int leader = __builtin_ctzll(wm);
                u32 base = 0;
                if ((int)lane == leader) base = atomicAdd(&s_nd, (u32)__popcll(wm));
                base = (u32)__builtin_amdgcn_readlane((int)base, leader);
                const u32 pos = base + (u32)__popcll(wm & ((1ull << lane) - 1ull));
                dk[pos] = kk[j];
                dc[pos] = (unsigned short)hh[j];
              }
              if (won || old == kk[j]) {
                atomicAdd(&tc[hh[j] >> 1], (hh[j] & 1u) ? 0x10000u : 1u);
                pending &= ~(1u << j);
              }
              else hh[j] = (hh[j] + 1) & smask;
            }
          }
        }
      }
      __syncthreads();
      HC_STAMP(1);
      {

      // the compact list is already there (any order): only the counts are still in the table
      const u32 D = s_nd;
      for (u32 i = tid; i < D; i += BLOCK) { const u32 h = dc[i]; dc[i] = (unsigned short)((tc[h >> 1] >> ((h & 1u) * 16u)) & 0xFFFFu); }
      if (tid < 16) dk[D + tid] = EMPTY;               // padding of the rank loop (D is block-uniform)
      __syncthreads();
      HC_STAMP(2);

      u64 *gk = keys + a;
      {
        const u32 bshift = low_bits > 8 ? low_bits - 8 : 0;
        s_bin[tid] = 0;
        __syncthreads();
        u32 li[KPT];
#pragma unroll
        for (int q = 0; q < KPT; q++) {
          const u32 i = (u32)q * BLOCK + tid;
          li[q] = 0;
          if (i < D) li[q] = atomicAdd(&s_bin[dk[i] >> bshift], 1u);
        }
        __syncthreads();
        u32 tot;
        const u32 e = block_excl_scan<BLOCK, u32>(s_bin[tid], s_scan, &tot);
        s_bin[tid] = e;
        if (tid == 0) s_bin[BLOCK] = D;
        __syncthreads();
        // the table is dead (its counts went to dc): it holds the suffixes in bin order now, the count words hold where each came from
        unsigned short *ix = reinterpret_cast<unsigned short *>(tc);
#pragma unroll
        for (int q = 0; q < KPT; q++) {
          const u32 i = (u32)q * BLOCK + tid;
          if (i < D) { const u32 kq = dk[i]; const u32 pos = s_bin[kq >> bshift] + li[q]; tk[pos] = kq; ix[pos] = (unsigned short)i; }
        }
        __syncthreads();
        for (u32 pidx = tid; pidx < D; pidx += BLOCK) {
          const u32 ki = tk[pidx];
          const u32 b = ki >> bshift, lo = s_bin[b], hi = s_bin[b + 1];
          u32 r = lo;
          for (u32 q = lo; q < hi; q++) r += (tk[q] < ki) ? 1u : 0u;
          if constexpr (NARROW) reinterpret_cast<u32 *>(keys)[a + r] = ki;
          else                  gk[r] = prefix | (u64)ki;
          cnt_tmp[a + r] = dc[ix[pidx]];
        }
      }
      if (tid == 0) group_distinct[tr_index(g, tr_a, tr_b)] = D;
      HC_STAMP(3);
      __syncthreads();                                 // dk/dc/s_tmp are reused by the next sub-bucket
      HC_STAMP(4);
      }
    }

#pragma unroll
    for (int j = 0; j < KPT; j++) kcur[j] = knext[j];
    a = na; n64 = nn; na = nna; nn = nnn; g = g1; g1 = g2; g2 = g3; p += G;
    if (DBG) { ph[5] += kcur[0] & 1; HC_STAMP(6); ph[7]++; }   // [6]: wait for the prefetched keys
  }
  if (DBG && tid == 0 && blockIdx.x < 64)
    for (int i = 0; i < 8; i++) dbg[blockIdx.x * 8 + i] = ph[i];
#undef HC_STAMP
}

// Hash-count over R PHYSICALLY CONSECUTIVE sub-buckets of a narrowed file per workgroup iteration (round 4; replaces
// countSingleKmers' sort + run-length passes, merylCountArray.C:323-365, like hash_count_kernel does).  Why: at the judged
// scale a sub-bucket holds ~516 k-mers of which ~70 are distinct -- 2 keys per thread, per-distinct phases on a quarter
// of the lanes, and every barrier, scan and loop header of an iteration paid per 516 keys (r03_pmc_sq.json: 39 of 64
// lanes active, 238 VALU instructions per thread and sub-bucket).  Here
//   * the R sub-buckets [g0, g0 + R) are one contiguous key range; a key's LOCAL sub-bucket number (position against the
//     R - 1 inner boundaries) becomes TAGB tag bits above its suffix: composite = tag << low_bits | suffix, so one table,
//     one compact list and one rank serve all R, and the distinct keys of sub-bucket j are the ranks
//     [Dpre_j, Dpre_{j+1}) -- they go back in place to start_j + (rank - Dpre_j) exactly where the one-at-a-time
//     kernel leaves them (compact_groups_narrow_kernel is unchanged);
//   * a table entry is composite << 11 | count (composite <= 20 bits + 1 so that EMPTY = all ones is no entry, a count
//     <= CAP < 2^11): the claiming CAS deposits count 1, a duplicate adds 1 to the same word -- no second array, no
//     16-bit halves sharing a bank, and entry order == key order, so the bin-rank compares whole words;
//   * the claimers append their SLOT to the compact list (u16); counts are read with the entry afterwards;
//   * the 256-bin counting sort of the distinct entries is scanned by ONE wave (four bins per lane) while the other
//     three clear the table and the bin counters of the NEXT iteration (double-buffered by iteration parity): four
//     workgroup barriers per iteration, none at its top or its end;
//   * PREFETCH THAT PREFETCHES: the next range's keys and the bounds of the one after it are vector loads issued only
//     after everything the previous iteration loaded has been consumed into registers.  (The older kernels issue the
//     next keys first and then touch the current ones: with the loads in conditional blocks the compiler's wait for the
//     OLD registers is s_waitcnt vmcnt(0), i.e. for the loads it has just issued -- MGC_HASH_DBG showed 6-8 K cycles
//     of "clear" per iteration that were this wait -- and their bounds come by scalar loads whose lgkmcnt the first
//     LDS barrier has to drain.)
// A range whose R sub-buckets together exceed CAP is not counted here: its sub-buckets of 1 .. max_size keys go on the
// retry list, which the one-at-a-time kernel (its non-empty-list instantiation) walks right after this launch; the
// > max_size ones belong to the streaming launch as before.
#ifndef HCM_WAVES
#define HCM_WAVES 8
#endif
template <int BLOCK, int CAP, int SLOTS, int R, bool DBG>
__global__ __launch_bounds__(BLOCK, HCM_WAVES)
void hash_count_multi_kernel(u32 *__restrict__ keys, const u64 *__restrict__ starts, u64 ng, u64 max_size, u32 low_bits,
                             u32 *__restrict__ cnt_tmp, u64 *__restrict__ group_distinct, u64 *__restrict__ dbg,
                             u32 tr_a, u32 tr_b, u32 *__restrict__ retry_list, u64 *__restrict__ retry_count) {
  static_assert((SLOTS & (SLOTS - 1)) == 0 && SLOTS * 4 >= CAP * 5 && SLOTS % (4 * BLOCK) == 0 && CAP % BLOCK == 0, "table geometry");
  static_assert(BLOCK == 256 && R >= 1 && R <= 4, "one bin per thread; at most two tag bits");
  constexpr int KPT = CAP / BLOCK;
  constexpr u32 TAGB = R == 1 ? 0u : (R == 2 ? 1u : 2u);
  constexpr u32 CNTB = 11u, CNT_MASK = (1u << CNTB) - 1u;
  static_assert(CAP <= (int)CNT_MASK, "a count must fit its field");
  constexpr u32 EMPTY = 0xFFFFFFFFu;
  __shared__ __attribute__((aligned(16))) u32 tk[SLOTS];            // composite << 11 | count
  __shared__ __attribute__((aligned(16))) u32 srt[CAP];             // the distinct entries in bin order
  __shared__ unsigned short lst[CAP];                               // slots of the claimed entries, in claim order
  __shared__ __attribute__((aligned(16))) u32 s_bin[2][BLOCK + 4];  // bin counts -> starts; [BLOCK] = D
  __shared__ u32 s_nd;
  const u32 tid = threadIdx.x, lane = tid & 63u;
  const u64 G = gridDim.x;
  const u32 low_mask = (u32)((1ull << low_bits) - 1ull);
  const u32 bshift = low_bits + TAGB - 8u;                          // the launcher guarantees 8 <= low_bits + TAGB <= 20
  const u64 nsuper = (ng + R - 1) / R;

  // bounds of super-bucket P as ONE vector load: lane l holds starts[g0 + min(l, R)] (clamped to ng); past the end: zeros
  auto load_bvec = [&](u64 P) -> u64 {
    if (P >= nsuper) return 0ull;
    const u64 gl = P * R + (lane < (u32)R ? lane : (u32)R);
    return starts[gl < ng ? gl : ng];
  };
  auto rdlane64 = [&](u64 v, int l) -> u64 {
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, l), hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), l);
    return ((u64)hi << 32) | lo;
  };
  // first key a0, relative boundaries b[1..R] (b[R] = all its keys), saturated to 32 bits
  auto unpack_bounds = [&](u64 bv, u64 &a0, u32 (&b)[R + 1]) {
    a0 = rdlane64(bv, 0);
    b[0] = 0;
#pragma unroll
    for (int j = 1; j <= R; j++) {
      const u64 d = rdlane64(bv, j) - a0;
      b[j] = d > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)d;
    }
  };
  // (uniform guards, lanes past the end re-read the last key: no per-lane predicate, no 64-bit address per slot)
  auto load_keys = [&](u64 a0, u32 n, u32 (&kr)[KPT]) {
    const bool fits = n <= (u32)CAP;
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      kr[j] = 0u;
      if (fits && (u32)j * BLOCK < n) {
        const u32 *src = keys + a0 + (u32)j * BLOCK;
        const u32 last = n - 1u - (u32)j * BLOCK;
        kr[j] = src[tid < last ? tid : last];
      }
    }
  };
  auto slots_for = [&](u32 n) -> u32 {
    u32 s = 256;
    while (s < n + n / 4 && s < (u32)SLOTS) s <<= 1;
    return s;
  };
  // comp[] of a range from its loaded keys and its inner boundaries
  auto tag_keys = [&](const u32 (&kr)[KPT], const u32 (&b)[R + 1], u32 (&cp)[KPT]) {
    const u32 b1 = (R >= 2) ? b[1] : 0u, b2 = (R >= 3) ? b[2] : 0u, b3 = (R >= 4) ? b[3] : 0u;
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const u32 idx = (u32)j * BLOCK + tid;
      u32 tag = 0;
      if (R >= 2) tag += (idx >= b1) ? 1u : 0u;
      if (R >= 3) tag += (idx >= b2) ? 1u : 0u;
      if (R >= 4) tag += (idx >= b3) ? 1u : 0u;
      cp[j] = (kr[j] & low_mask) | (tag << low_bits);
      asm volatile("" : "+v"(cp[j]) :: "memory");     // consumed HERE: nothing newer is in flight when the wait for it runs
    }
  };

  u64 P = blockIdx.x;
  u64 a0, a0n;
  u32 bc[R + 1], bn[R + 1];
  u32 kcur[KPT], comp[KPT];
  unpack_bounds(load_bvec(P), a0, bc);
  load_keys(a0, bc[R], kcur);
  u64 bvec = load_bvec(P + G);
  tag_keys(kcur, bc, comp);
  unpack_bounds(bvec, a0n, bn);
  load_keys(a0n, bn[R], kcur);                         // the second range's keys and the third's bounds: in flight
  bvec = load_bvec(P + 2 * G);
  u32 par = 0;
  u32 cleared = (u32)SLOTS;                             // tk[0, cleared) is EMPTY whenever an insert phase begins
  {
    uint4 *tk4 = reinterpret_cast<uint4 *>(tk);
    for (u32 i = tid; i < (u32)SLOTS / 4; i += BLOCK) tk4[i] = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
    s_bin[0][tid] = 0; s_bin[1][tid] = 0;
    if (tid == 0) s_nd = 0;
  }
  __syncthreads();

  u64 ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
#define HC_STAMP(i) do { if (DBG) { const u64 t = __builtin_readcyclecounter(); ph[i] += t - t0; t0 = t; } } while (0)
  while (P < nsuper) {
    if (DBG) t0 = __builtin_readcyclecounter();
    const u64 g0 = P * R;
    const u32 nsub = (u32)((ng - g0 < (u64)R) ? ng - g0 : (u64)R);
    const u32 n = bc[R];
    const bool active = n != 0 && n <= (u32)CAP;
    const u32 slots = slots_for(n);
    if (active) {
      u32 hh[KPT];
      const u32 smask = slots - 1, sshift = 32 - (u32)__builtin_ctz(slots);
      if (slots > cleared) {                           // (the range the last clear had been sized for was not counted)
        uint4 *tk4 = reinterpret_cast<uint4 *>(tk);
        for (u32 i = tid; i < (u32)SLOTS / 4; i += BLOCK) tk4[i] = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
        __syncthreads();
      }
      HC_STAMP(0);
      // The kernel is instruction-issue-bound (VALU and SALU both near one per cycle and CU, r04c_pmc): the insert is
      // written for few instructions.  First probe of every key straight-line, slot after slot under uniform guards (only
      // the last slot is partial); a claim is only NOTED (won) -- the claimed slots join the compact list after the loop with
      // one LDS atomic per wave instead of a returning atomic, a wait and a store inside every slot; the few keys whose
      // first slot held another suffix go round the generic loop.
      {
      u32 won = 0, pending = 0;
#pragma unroll
      for (int j = 0; j < KPT; j++) {
        hh[j] = (comp[j] * 0x9E3779B1u) >> sshift;
        if ((u32)j * BLOCK < n) {
          const bool act = (u32)j * BLOCK + tid < n;
          u32 old = 0xFFFFF7FFu;                       // (neither EMPTY nor any suffix)
          if (act) old = atomicCAS(&tk[hh[j]], EMPTY, (comp[j] << CNTB) | 1u);
          const bool w = old == EMPTY, dup = (old >> CNTB) == comp[j];
          if (dup) atomicAdd(&tk[hh[j]], 1u);
          won |= w ? (1u << j) : 0u;
          pending |= (act && !w && !dup) ? (1u << j) : 0u;
        }
      }
      while (pending) {                                // linear probing, one step of every still-pending key per round
#pragma unroll
        for (int j = 0; j < KPT; j++) {
          if ((pending >> j) & 1u) {
            hh[j] = (hh[j] + 1) & smask;
            const u32 old = atomicCAS(&tk[hh[j]], EMPTY, (comp[j] << CNTB) | 1u);
            const bool w = old == EMPTY, dup = (old >> CNTB) == comp[j];
            if (dup) atomicAdd(&tk[hh[j]], 1u);
            if (w) won |= 1u << j;
            if (w || dup) pending &= ~(1u << j);
          }
        }
      }
      {
        u64 wm[KPT];
        u32 tot = 0;
#pragma unroll
        for (int j = 0; j < KPT; j++) { wm[j] = __ballot((won >> j) & 1u); tot += (u32)__popcll(wm[j]); }
        u32 base = 0;
        if (lane == 0) base = atomicAdd(&s_nd, tot);
        base = (u32)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
        for (int j = 0; j < KPT; j++) {
          if (wm[j]) {
            if ((won >> j) & 1u)
              lst[base + __builtin_amdgcn_mbcnt_hi((u32)(wm[j] >> 32), __builtin_amdgcn_mbcnt_lo((u32)wm[j], 0u))] = (unsigned short)hh[j];
            base += (u32)__popcll(wm[j]);
          }
        }
      }
      }
      __syncthreads();
      HC_STAMP(1);
    }

    // Once per iteration, after its insert phase (the registers of the probe loop are dead, the stores of the previous
    // iteration long acknowledged -- they count in vmcnt like the loads): the NEXT range's keys, loaded a whole iteration
    // ago, take the place of this range's in comp[]; only then the keys of the range after it and the bounds of the one after
    // that are issued.
    const u32 b1 = (R >= 2) ? bc[1] : 0u, b2 = (R >= 3) ? bc[2] : 0u, b3 = (R >= 4) ? bc[3] : 0u;
    const u64 a = a0;
    const u32 nslots = bn[R] <= (u32)CAP ? slots_for(bn[R]) : 256u;  // what the next insert phase needs cleared
    tag_keys(kcur, bn, comp);
    a0 = a0n;
#pragma unroll
    for (int j = 0; j <= R; j++) bc[j] = bn[j];
    unpack_bounds(bvec, a0n, bn);
    load_keys(a0n, bn[R], kcur);
    bvec = load_bvec(P + 3 * G);
    HC_STAMP(6);

    if (active) {
      // the distinct entries into 256 bins by their top eight bits (tag first): one returning LDS atomic each
      const u32 D = s_nd;
      const bool big = D > (u32)BLOCK;                 // more distinct suffixes than threads (low-coverage input): staged through LDS
      u32 ent = 0, li = 0;
      if (!big) {
        if (tid < D) { ent = tk[lst[tid]]; li = atomicAdd(&s_bin[par][ent >> (CNTB + bshift)], 1u); }
      } else {
        for (u32 i = tid; i < D; i += BLOCK) {
          const u32 e = tk[lst[i]];
          srt[i] = e;
          lst[i] = (unsigned short)atomicAdd(&s_bin[par][e >> (CNTB + bshift)], 1u);
        }
      }
      __syncthreads();
      HC_STAMP(2);
      if (tid < 64) {                                  // one wave scans the 256 bin counts: four per lane
        uint4 c = reinterpret_cast<uint4 *>(s_bin[par])[tid];
        const u32 s4 = c.x + c.y + c.z + c.w;
        u32 x = s4;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
        const u32 e0 = x - s4;
        reinterpret_cast<uint4 *>(s_bin[par])[tid] = make_uint4(e0, e0 + c.x, e0 + c.x + c.y, e0 + c.x + c.y + c.z);
        if (tid == 63) s_bin[par][BLOCK] = x;
      } else {
        // ... the other three clear what the NEXT insert phase uses (tk, lst and s_nd are dead by now; with more distinct
        // suffixes than threads the table takes the sorted entries first and is cleared at the end)
        uint4 *tk4 = reinterpret_cast<uint4 *>(tk);
        if (!big) for (u32 i = tid - 64; i < nslots / 4; i += BLOCK - 64) tk4[i] = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
        for (u32 i = tid - 64; i < BLOCK; i += BLOCK - 64) s_bin[par ^ 1u][i] = 0;
        if (tid == 64) s_nd = 0;
      }
      cleared = nslots;
      __syncthreads();
      if (!big) {
        if (tid < D) srt[s_bin[par][ent >> (CNTB + bshift)] + li] = ent;
      } else {
        for (u32 i = tid; i < D; i += BLOCK) { const u32 e = srt[i]; tk[s_bin[par][e >> (CNTB + bshift)] + lst[i]] = e; }
      }
      __syncthreads();
      HC_STAMP(3);

      // rank inside the bin, then back in place: sub-bucket j's distinct suffixes ascending from its own start
      const u32 *sb = s_bin[par];
      const u32 *sorted = big ? tk : srt;
      u32 *kout = keys + a;
      u32 *cout = cnt_tmp + a;
      for (u32 p = tid; p < D; p += BLOCK) {
        const u32 e = sorted[p];
        const u32 b = e >> (CNTB + bshift), lo = sb[b], hi = sb[b + 1];
        u32 r = lo;
        for (u32 q = lo; q < hi; q++) r += (sorted[q] < e) ? 1u : 0u;
        const u32 tag = (R >= 2) ? (e >> (CNTB + low_bits)) : 0u;
        u32 dest = r;
        if (R >= 2) {
          const u32 sb_base = (tag == 0) ? 0u : ((tag == 1) ? b1 : ((tag == 2) ? b2 : b3));
          dest = sb_base + (r - sb[tag << (8u - TAGB)]);
        }
        kout[dest] = (e >> CNTB) & low_mask;
        cout[dest] = e & CNT_MASK;
      }
      if (tid < nsub) {
        const u32 dlo = sb[tid << (8u - TAGB)];
        const u32 dhi = (tid + 1 == (1u << TAGB)) ? D : sb[(tid + 1) << (8u - TAGB)];
        group_distinct[tr_index(g0 + tid, tr_a, tr_b)] = dhi - dlo;
      }
      if (big) {                                       // the table held the sorted entries: cleared now, behind two more barriers
        __syncthreads();
        uint4 *tk4 = reinterpret_cast<uint4 *>(tk);
        for (u32 i = tid; i < nslots / 4; i += BLOCK) tk4[i] = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
        __syncthreads();
      }
      par ^= 1u;                                       // (srt and this parity's bin table are next written three barriers on)
      HC_STAMP(4);
    } else if (tid < nsub) {
      // empty range: zeros; a range above the table: its sub-buckets one at a time, by the kernel that walks the retry list
      const u32 lo = (tid == 0) ? 0u : ((tid == 1) ? b1 : ((tid == 2) ? b2 : b3));
      const u32 hi = (tid + 1 == nsub) ? n : ((tid == 0) ? b1 : ((tid == 1) ? b2 : b3));
      u64 nj = hi - lo;
      if (n == 0xFFFFFFFFu) nj = starts[g0 + tid + 1] - starts[g0 + tid];   // (saturated: the exact size)
      if (nj == 0) group_distinct[tr_index(g0 + tid, tr_a, tr_b)] = 0;
      else if (nj <= max_size) retry_list[atomicAdd((unsigned long long *)retry_count, 1ull)] = (u32)(g0 + tid);
    }
    P += G;
    if (DBG) ph[7]++;
  }
  if (DBG && tid == 0 && blockIdx.x < 64)
    for (int i = 0; i < 8; i++) dbg[blockIdx.x * 8 + i] = ph[i];
#undef HC_STAMP
}

// Hash-count with the table sized by DISTINCT suffixes and the keys STREAMED through it in chunks (round 6; replaces
// countSingleKmers' sort + run-length passes, merylCountArray.C:323-365, like the kernels above).  Why: hash_count_multi_kernel
// holds all keys of an iteration in registers and sizes its table for all of them being distinct (CAP = 1536 keys), which pins the
// file plan at sub-buckets of <= 1152 k-mers on average -- 17 or 18 grouping bits, i.e. a nine-bit first digit (128-byte runs) on
// the large files -- although at the judged 30x a 1030-key sub-bucket holds ~140 distinct suffixes (profiles/r06_gate_distinct.json:
// D/N = 0.137; p99.99 of D at 2066-key sub-buckets = 652).  Here
//   * one iteration counts ONE sub-bucket of up to max_size (<= 4094) keys: its keys come in 1..3 chunks of <= BLOCK * KPC keys,
//     evenly cut; chunk c + 1 (or the first chunk of the next sub-bucket) is in flight while chunk c is inserted -- the same
//     consume-before-issue order as hash_count_multi_kernel -- and all chunks claim / add into ONE table, no barrier between them;
//   * a table entry is suffix << 12 | count (suffix <= 20 bits, a count <= 4094 so that EMPTY = all ones is no entry): the claiming
//     CAS deposits count 1, a duplicate adds 1, entry order == suffix order;
//   * the table holds SLOTS entries and the claim list DCAP of them -- capacity in DISTINCT suffixes.  A sub-bucket that claims more
//     (low coverage: D ~ N), or whose probes run long because the table fills, raises s_ovf: nothing of it is written, its number
//     goes on the retry list, and the LIST instantiation (SLOTS = 8192, DCAP >= max_size: cannot overflow) counts it afterwards;
//   * the per-distinct phases are hash_count_multi_kernel's (256 bins by the top eight suffix bits, one-wave scan with the table
//     cleared in its shadow, rank inside the bin) with up to TWO entries per thread in registers (D <= 512), staged through LDS above.
// LIST: visit the sub-buckets visit[0 .. *visit_count) (a sparse grid's non-empty list; the retry list); otherwise all ng.
// retry_list == nullptr: the retry launch itself.
template <typename KT, int BLOCK, int KPC, int SLOTS, int DCAP, bool LIST, bool DBG>
__global__ __launch_bounds__(BLOCK, (sizeof(KT) == 8 ? (SLOTS <= 2048 ? 5 : 1) : (SLOTS <= 2048 ? 8 : 2)))
void hash_count_stream_kernel(KT *__restrict__ keys, const u64 *__restrict__ starts, u64 ng, u64 max_size_, u32 low_bits,
                              u32 *__restrict__ cnt_tmp, u64 *__restrict__ group_distinct, u64 *__restrict__ dbg,
                              u32 tr_a, u32 tr_b, const u32 *__restrict__ visit, const u64 *__restrict__ visit_count,
                              u32 *__restrict__ retry_list, u64 *__restrict__ retry_count) {
  static_assert((SLOTS & (SLOTS - 1)) == 0 && SLOTS % (4 * BLOCK) == 0 && DCAP <= SLOTS && DCAP <= 65536 && BLOCK == 256, "table geometry");
  static_assert(sizeof(KT) == 4 || sizeof(KT) == 8, "32-bit words of a narrowed file, or whole 8-byte k-mers");
  constexpr bool W64 = sizeof(KT) == 8;                               // whole k-mers: 64-bit entries, the bits above the suffix put back on the way out
  constexpr u32 CNTB = 12u, CNT_MASK = (1u << CNTB) - 1u;
  constexpr KT EMPTY = (KT)~(KT)0;
  constexpr u32 CH = (u32)(BLOCK * KPC);
  constexpr u32 PROBE_MAX = 64u;                                     // probe steps of one key beyond which the table counts as full
  __shared__ __attribute__((aligned(16))) KT tk[SLOTS];             // suffix << 12 | count
  __shared__ __attribute__((aligned(16))) KT srt[DCAP];             // the distinct entries in bin order
  __shared__ unsigned short lst[DCAP];                              // slots of the claimed entries, in claim order
  __shared__ __attribute__((aligned(16))) u32 s_bin[2][BLOCK + 4];  // bin counts -> starts; [BLOCK] = D
  __shared__ u32 s_nd, s_ovf;
  const u32 tid = threadIdx.x, lane = tid & 63u;
  const u32 G = gridDim.x;
  const KT  low_mask = (KT)((1ull << low_bits) - 1ull);
  const u32 bshift = low_bits - 8u;                                 // the launcher guarantees 8 <= low_bits <= 20 (whole k-mers: <= 52)
  const u32 np = (u32)(LIST ? *visit_count : ng);                   // (a narrowed file: fewer than 2^30 keys, at most 2^18 sub-buckets)
  const u32 max_size = (u32)max_size_;

  // bounds of sub-bucket g (at position P of the visit order) as ONE vector load of 32-bit values (a narrowed file holds fewer
  // than 2^30 keys): lane 0 its start, every other lane its end; past the end of the visit order: zeros
  auto load_starts = [&](u32 P, u32 g) -> u32 {
    if (P >= np) return 0u;
    return (u32)starts[g + (lane ? 1u : 0u)];
  };
  auto unpack_bounds = [&](u32 bv, u32 &a, u32 &n) {
    a = (u32)__builtin_amdgcn_readlane((int)bv, 0);
    n = (u32)__builtin_amdgcn_readlane((int)bv, 1) - a;
  };
  // LIST: the number of the sub-bucket at position P (one load, the same address in every lane)
  auto load_gnum = [&](u32 P) -> u32 { return (LIST && P < np) ? visit[P] : 0u; };
  // chunks of an n-key sub-bucket: 1..3 of them, evenly cut, whole rows of BLOCK keys (n <= max_size <= 4094 < 3 * CH); 0: not
  // counted here (empty, or above max_size: the streaming launch's)
  auto chunk_size = [&](u32 n) -> u32 {
    if (n == 0u || n > max_size) return 0u;
    u32 per = n;
    if (n > 2u * CH) per = ((n + 2u) * 43691u) >> 17;              // ceil(n / 3)
    else if (n > CH) per = (n + 1u) >> 1;
    return (per + (u32)BLOCK - 1u) & ~((u32)BLOCK - 1u);
  };
  // (uniform guards, lanes past the end re-read the last key: no per-lane predicate, no 64-bit address per slot)
  auto load_chunk = [&](u32 a, u32 cnt, KT (&kr)[KPC]) {
    const KT *src = keys + a;
#pragma unroll
    for (int j = 0; j < KPC; j++) {
      kr[j] = (KT)0;
      if ((u32)j * BLOCK < cnt) {
        const u32 last = cnt - 1u;
        const u32 idx = (u32)j * BLOCK + tid;
        kr[j] = src[idx < last ? idx : last];
      }
    }
  };
  auto slots_for = [&](u32 n) -> u32 {                              // (2n: D is not known, the keys bound it)
    if (2u * n <= 256u) return 256u;
    const u32 s = 1u << (32 - __builtin_clz(2u * n - 1u));
    return s < (u32)SLOTS ? s : (u32)SLOTS;
  };

  u32 P = blockIdx.x;
  u32 a0, a1, g0, g1, n0, n1;                                       // the current sub-bucket, the next one
  KT  kcur[KPC], comp[KPC];
  KT  pre = (KT)0;                                                   // W64: the current sub-bucket's bits above the suffix
  // in flight from one sub-bucket's end to the next: the bounds of the sub-bucket at P + 2G (its number: gb) and, LIST, the number
  // of the one at P + 3G -- consumed at ONE point, right before the loads of that round are issued (a wait behind newer loads
  // would wait for them)
  g0 = LIST ? (u32)__builtin_amdgcn_readfirstlane((int)load_gnum(P)) : P;
  g1 = LIST ? (u32)__builtin_amdgcn_readfirstlane((int)load_gnum(P + G)) : P + G;
  u32 gb = LIST ? (u32)__builtin_amdgcn_readfirstlane((int)load_gnum(P + 2 * G)) : P + 2 * G;
  unpack_bounds(load_starts(P, g0), a0, n0);
  unpack_bounds(load_starts(P + G, g1), a1, n1);
  u32 csz = chunk_size(n0), off = 0;                                // the current chunk: keys [off, off + csz) of the current sub-bucket
  load_chunk(a0, csz < n0 ? csz : n0, kcur);                        // (csz == 0: nothing)
  u32 bvec = load_starts(P + 2 * G, gb);
  u32 gv = load_gnum(P + 3 * G);
  u32 par = 0;
  u32 cleared = (u32)SLOTS;                                         // tk[0, cleared) is EMPTY whenever an insert phase begins
  // (EMPTY is all ones in either width: the table is cleared as 16-byte vectors of ones, `slots` counted in entries)
  constexpr u32 E16 = 16u / (u32)sizeof(KT);                        // entries per 16-byte vector
  constexpr u32 ONES = 0xFFFFFFFFu;
  {
    uint4 *tk4 = reinterpret_cast<uint4 *>(tk);
    for (u32 i = tid; i < (u32)SLOTS / E16; i += BLOCK) tk4[i] = make_uint4(ONES, ONES, ONES, ONES);
    s_bin[0][tid] = 0; s_bin[1][tid] = 0;
    if (tid == 0) { s_nd = 0; s_ovf = 0; }
  }
  __syncthreads();

  u64 ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
#define HC_STAMP(i) do { if (DBG) { const u64 t = __builtin_readcyclecounter(); ph[i] += t - t0; t0 = t; } } while (0)
  // ONE loop over chunks (one load site: the loads of the next chunk land in the registers the consumed one has left, whichever
  // sub-bucket it belongs to -- two load sites merged by copies would wait for the loads they had just issued)
  while (P < np) {
    if (DBG) t0 = __builtin_readcyclecounter();
    const bool active = csz != 0u;
    const u32 rest = n0 - off;
    const u32 cnt = active ? (rest < csz ? rest : csz) : 0u;
    const bool last = !active || rest <= csz;
    const u32 slots = slots_for(n0);
    const u32 smask = slots - 1, sshift = 32 - (u32)__builtin_ctz(slots);
    // the chunk that was loaded a whole chunk ago is consumed into comp[]; only then the next one is issued (a wait for old
    // registers behind new loads would wait for the new loads: hash_count_multi_kernel)
    if constexpr (W64) { if (off == 0u) pre = kcur[0] & ~low_mask; }   // (every lane holds a key of the sub-bucket: lanes past the end re-read the last one)
#pragma unroll
    for (int j = 0; j < KPC; j++) {
      comp[j] = kcur[j] & low_mask;
      asm volatile("" : "+v"(comp[j]) :: "memory");
    }
    const u32 ncs = chunk_size(n1);
    u32 a2 = 0, n2 = 0, g2 = 0, g3 = 0;
    if (last) {                                                     // (uniform) everything the last round loaded is consumed here ...
      unpack_bounds(bvec, a2, n2);
      g2 = gb;
      g3 = LIST ? (u32)__builtin_amdgcn_readfirstlane((int)gv) : P + 3 * G;
    }
    {
      const u32 la = last ? a1 : a0 + off + csz;
      const u32 lrest = last ? n1 : rest - csz, lcs = last ? ncs : csz;
      load_chunk(la, lrest < lcs ? lrest : lcs, kcur);
    }
    if (last) {                                                     // ... and this round's are issued
      bvec = load_starts(P + 3 * G, g3);
      gb = g3;
      if (LIST) gv = load_gnum(P + 4 * G);
    }
    HC_STAMP(6);
    if (active) {
      if (off == 0u && slots > cleared) {                           // (the sub-bucket the last clear had been sized for was not counted)
        uint4 *tk4 = reinterpret_cast<uint4 *>(tk);
        for (u32 i = tid; i < (u32)SLOTS / E16; i += BLOCK) tk4[i] = make_uint4(ONES, ONES, ONES, ONES);
        __syncthreads();
      }
      HC_STAMP(0);
      u32 hh[KPC];
      u32 won = 0, pending = 0;
      // (the first probes issued back to back and their answers looked at afterwards -- six independent atomics in flight per thread
      // -- was measured SLOWER: 0.595 against 0.566 ms per launch, profiles/r06_ab_runs.txt: the kernel is issue-bound, not latency-bound)
#pragma unroll
      for (int j = 0; j < KPC; j++) {
        if constexpr (W64) hh[j] = ((((u32)comp[j]) ^ ((u32)(comp[j] >> 32) * 0x85EBCA6Bu)) * 0x9E3779B1u) >> sshift;   // (suffixes of <= 52 bits: the high word holds <= 20)
        else               hh[j] = (comp[j] * 0x9E3779B1u) >> sshift;
        if ((u32)j * BLOCK < cnt) {
          const bool act = (u32)j * BLOCK + tid < cnt;
          KT old = (KT)0;
          if (act) old = atomicCAS(&tk[hh[j]], EMPTY, (KT)((comp[j] << CNTB) | (KT)1));
          const bool w = act && old == EMPTY, dup = act && !w && (old >> CNTB) == comp[j];   // (EMPTY >> 12 IS the all-ones suffix)
          if (dup) atomicAdd(reinterpret_cast<u32 *>(&tk[hh[j]]), 1u);                        // (the count: the low twelve bits of the entry's low word)
          won |= w ? (1u << j) : 0u;
          pending |= (act && !w && !dup) ? (1u << j) : 0u;
        }
      }
      u32 steps = 0;
      while (pending) {                                // linear probing, one step of every still-pending key per round
#pragma unroll
        for (int j = 0; j < KPC; j++) {
          if ((pending >> j) & 1u) {
            hh[j] = (hh[j] + 1) & smask;
            const KT old = atomicCAS(&tk[hh[j]], EMPTY, (KT)((comp[j] << CNTB) | (KT)1));
            const bool w = old == EMPTY, dup = !w && (old >> CNTB) == comp[j];
            if (dup) atomicAdd(reinterpret_cast<u32 *>(&tk[hh[j]]), 1u);
            if (w) won |= 1u << j;
            if (w || dup) pending &= ~(1u << j);
          }
        }
        if (++steps > PROBE_MAX) { s_ovf = 1u; pending = 0; }       // the table is (nearly) full: more distinct suffixes than it holds
      }
      {
        u64 wm[KPC];
        u32 tot = 0;
#pragma unroll
        for (int j = 0; j < KPC; j++) { wm[j] = __ballot((won >> j) & 1u); tot += (u32)__popcll(wm[j]); }
        u32 base = 0;
        if (lane == 0) base = atomicAdd(&s_nd, tot);
        base = (u32)__builtin_amdgcn_readfirstlane((int)base);
        if (base + tot > (u32)DCAP) { if (lane == 0) s_ovf = 1u; }  // more distinct suffixes than the list holds
        else {
#pragma unroll
          for (int j = 0; j < KPC; j++) {
            if (wm[j]) {
              if ((won >> j) & 1u)
                lst[base + __builtin_amdgcn_mbcnt_hi((u32)(wm[j] >> 32), __builtin_amdgcn_mbcnt_lo((u32)wm[j], 0u))] = (unsigned short)hh[j];
              base += (u32)__popcll(wm[j]);
            }
          }
        }
      }
      HC_STAMP(1);
    }
    if (!last) { off += csz; continue; }               // the sub-bucket's next chunk (no barrier: one table, atomics only)

    // ---- the sub-bucket's last chunk is in: its distinct suffixes, ascending, back in place ----
    const u32 a = a0, g = g0, n = n0;
    const u32 nslots = slots_for(n1);                  // what the next insert phase needs cleared
    if (active) __syncthreads();                       // every insert and every list append of the sub-bucket is done
    a0 = a1; n0 = n1; g0 = g1; off = 0; csz = ncs;
    a1 = a2; n1 = n2; g1 = g2;
    if (active) {
      const bool ovf = s_ovf != 0u;
      const u32 D = s_nd;
      if (!ovf) {
        // the distinct entries into 256 bins by their top eight bits: one returning LDS atomic each
        const bool big = D > 2u * (u32)BLOCK;          // more than two per thread (low-coverage input): staged through LDS
        KT  ent0 = 0, ent1 = 0;
        u32 li0 = 0, li1 = 0;
        if (!big) {
          if (tid < D)              { ent0 = tk[lst[tid]];              li0 = atomicAdd(&s_bin[par][(u32)(ent0 >> (CNTB + bshift))], 1u); }
          if (tid + (u32)BLOCK < D) { ent1 = tk[lst[tid + (u32)BLOCK]]; li1 = atomicAdd(&s_bin[par][(u32)(ent1 >> (CNTB + bshift))], 1u); }
        } else {
          for (u32 i = tid; i < D; i += BLOCK) {
            const KT e = tk[lst[i]];
            srt[i] = e;
            lst[i] = (unsigned short)atomicAdd(&s_bin[par][(u32)(e >> (CNTB + bshift))], 1u);
          }
        }
        __syncthreads();
        HC_STAMP(2);
        if (tid < 64) {                                // one wave scans the 256 bin counts: four per lane
          uint4 c = reinterpret_cast<uint4 *>(s_bin[par])[tid];
          const u32 s4 = c.x + c.y + c.z + c.w;
          u32 x = s4;
#pragma unroll
          for (int d = 1; d < 64; d <<= 1) { const u32 y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
          const u32 e0 = x - s4;
          reinterpret_cast<uint4 *>(s_bin[par])[tid] = make_uint4(e0, e0 + c.x, e0 + c.x + c.y, e0 + c.x + c.y + c.z);
          if (tid == 63) s_bin[par][BLOCK] = x;
        } else {
          // ... the other three clear what the NEXT insert phase uses (tk, lst and s_nd are dead by now; staged: the table takes
          // the sorted entries first and is cleared at the end)
          uint4 *tk4 = reinterpret_cast<uint4 *>(tk);
          if (!big) for (u32 i = tid - 64; i < nslots / E16; i += BLOCK - 64) tk4[i] = make_uint4(ONES, ONES, ONES, ONES);
          for (u32 i = tid - 64; i < BLOCK; i += BLOCK - 64) s_bin[par ^ 1u][i] = 0;
          if (tid == 64) s_nd = 0;
        }
        cleared = nslots;
        __syncthreads();
        if (!big) {
          if (tid < D)              srt[s_bin[par][(u32)(ent0 >> (CNTB + bshift))] + li0] = ent0;
          if (tid + (u32)BLOCK < D) srt[s_bin[par][(u32)(ent1 >> (CNTB + bshift))] + li1] = ent1;
        } else {
          for (u32 i = tid; i < D; i += BLOCK) { const KT e = srt[i]; tk[s_bin[par][(u32)(e >> (CNTB + bshift))] + lst[i]] = e; }
        }
        __syncthreads();
        HC_STAMP(3);

        // rank inside the bin, then back in place: the distinct suffixes ascending from the sub-bucket's own start
        const u32 *sb = s_bin[par];
        const KT *sorted = big ? tk : srt;
        KT *kout = keys + a;
        u32 *cout = cnt_tmp + a;
        for (u32 p = tid; p < D; p += BLOCK) {
          const KT  e = sorted[p];
          const u32 b = (u32)(e >> (CNTB + bshift)), lo = sb[b], hi = sb[b + 1];
          u32 r = lo;
          for (u32 q = lo; q < hi; q++) r += (sorted[q] < e) ? 1u : 0u;
          if constexpr (W64) kout[r] = pre | (e >> CNTB);
          else               kout[r] = e >> CNTB;
          cout[r] = (u32)e & CNT_MASK;
        }
        if (tid == 0) group_distinct[tr_index(g, tr_a, tr_b)] = D;
        if (big) {                                     // the table held the sorted entries: cleared now, behind two more barriers
          __syncthreads();
          uint4 *tk4 = reinterpret_cast<uint4 *>(tk);
          for (u32 i = tid; i < nslots / E16; i += BLOCK) tk4[i] = make_uint4(ONES, ONES, ONES, ONES);
          __syncthreads();
        }
        par ^= 1u;                                     // (srt and this parity's bin table are next written three barriers on)
        HC_STAMP(4);
      } else {
        // more distinct suffixes than the table or the list holds: nothing is written, the retry launch takes the sub-bucket
        __syncthreads();                               // (everybody has read s_ovf and s_nd)
        uint4 *tk4 = reinterpret_cast<uint4 *>(tk);
        for (u32 i = tid; i < (u32)SLOTS / E16; i += BLOCK) tk4[i] = make_uint4(ONES, ONES, ONES, ONES);
        if (tid == 0) {
          s_nd = 0; s_ovf = 0;
          if (retry_list) retry_list[atomicAdd((unsigned long long *)retry_count, 1ull)] = g;
          else            group_distinct[tr_index(g, tr_a, tr_b)] = 0; // (the retry launch: DCAP >= max_size, does not happen)
        }
        cleared = (u32)SLOTS;
        __syncthreads();
      }
    } else if (n == 0u && tid == 0) {
      group_distinct[tr_index(g, tr_a, tr_b)] = 0;     // (larger than max_size: the streaming launch's)
    }
    P += G;
    if (DBG) ph[7]++;
  }
  if (DBG && tid == 0 && blockIdx.x < 64)
    for (int i = 0; i < 8; i++) dbg[blockIdx.x * 8 + i] = ph[i];
#undef HC_STAMP
}

// (Round 5: the kernels hash_countw_kernel replaced -- hash_count64_kernel with 64-bit suffixes in the table, hash_count64i_kernel and
// hash_count128_kernel with the index-claimed table and a pass over it -- and the bitmap-count kernel, measured equal to the hash-count
// in round 3, were removed: DESIGN_HISTORY.md, profiles/r03a_*, r04k_*, r04v_*.)

// The index-claimed table of rounds 2-3 rebuilt the way hash_count_multi_kernel was (round 4; replaces countSingleKmers'
// sort + run-length passes, merylCountArray.C:323-365, for suffixes that do not fit the packed 32-bit table: 8-byte keys with
// 32..58-bit suffixes -- k = 28..32, `compress` -- and 16-byte keys, k = 33..64): the same index-claimed table
// (count << 16 | index of the claiming key's staged suffix) and the same results, but
//   * the claimed slots are NOTED and appended to a compact list after the insert with one LDS atomic per wave: no pass
//     over the whole table, no workgroup scan, no compaction copy;
//   * the bin-rank sorts the ENTRIES (count | index) by their suffix's top eight bits: the bins are scanned by one wave
//     while the other three clear the table for the next sub-bucket; the sorted entries take the place of the list;
//   * six workgroup barriers per sub-bucket instead of fourteen;
//   * the next sub-bucket's keys, loaded a whole iteration earlier, are consumed into registers right after the insert
//     phase, BEFORE the loads of the one after it are issued (a wait for old registers behind new loads waits for the new
//     loads), and the bounds come by one vector load instead of scalar loads the first barrier has to drain.
// KT = u64 | K128; WIDE (K128 only): the suffix needs the high word (low_bits > 64).  LIST: only the non-empty sub-buckets
// are visited (sparse grids: `compress`) -- their numbers are prefetched one more iteration ahead than their bounds.
template <typename KT, int BLOCK, int CAP, int SLOTS, bool WIDE, bool LIST>
__global__ __launch_bounds__(BLOCK, (sizeof(KT) >= 12 ? (CAP <= 768 ? 5 : 4) : (CAP <= 768 ? 6 : 5)))
void hash_countw_kernel(KT *__restrict__ keys, const u64 *__restrict__ starts, u64 ng, u64 max_size, u32 low_bits,
                        u32 *__restrict__ cnt_tmp, u64 *__restrict__ group_distinct,
                        const u32 *__restrict__ nz, const u64 *__restrict__ nz_count, u32 tr_a, u32 tr_b) {
  static_assert((SLOTS & (SLOTS - 1)) == 0 && SLOTS * 3 >= CAP * 4 && SLOTS % (4 * BLOCK) == 0 && CAP % BLOCK == 0 && CAP < 0xFFFF, "table geometry");
  static_assert(BLOCK == 256, "one bin per thread");
  constexpr bool K16 = sizeof(KT) >= 12;               // two-word suffixes: K128, or K96 (12-byte records: the bits below the file)
  constexpr bool IS96 = sizeof(KT) == 12;
  static_assert(K16 || !WIDE, "WIDE is a property of 16- and 12-byte keys");
  constexpr int KPT = CAP / BLOCK;
  constexpr u32 EMPTY = 0x0000FFFFu;
  __shared__ __attribute__((aligned(16))) u64 dlo[CAP];               // staged suffixes of the sub-bucket (low words)
  __shared__ __attribute__((aligned(16))) u64 dhi[WIDE ? CAP : 2];    //   ... high words
  __shared__ __attribute__((aligned(16))) u32 tw[SLOTS];              // instances << 16 | index of the claiming key
  __shared__ __attribute__((aligned(16))) u32 srt[CAP];               // first the claimed slots (u16 list), then the entries in bin order
  // (the bin counters are double-buffered by iteration parity -- the idle waves of the scan phase zero the next iteration's --
  // except where that kilobyte costs a workgroup per CU: 16-byte keys with 1536-key tables, 40 KiB; they are zeroed at the top)
  constexpr bool DB = !(WIDE && CAP > 768);
  __shared__ __attribute__((aligned(16))) u32 s_bin[DB ? 2 : 1][BLOCK + 4];    // bin counts -> starts; [BLOCK] = D
  __shared__ u32 s_nd;
  unsigned short *lst = reinterpret_cast<unsigned short *>(srt);
  const u32 tid = threadIdx.x, lane = tid & 63u;
  const u64 G = gridDim.x;
  const u64 mask_lo = (low_bits >= 64) ? ~0ull : ((1ull << low_bits) - 1ull);
  const u64 mask_hi = (low_bits <= 64) ? 0ull : ((low_bits >= 128) ? ~0ull : ((1ull << (low_bits - 64)) - 1ull));
  const u32 bshift = low_bits > 8 ? low_bits - 8 : 0;
  const u64 np = LIST ? *nz_count : ng;                               // iterations: list entries, or every sub-bucket

  auto bin_of = [&](u64 lo, u64 hi) -> u32 {                          // top eight bits of the suffix
    if (WIDE) return bshift >= 64 ? (u32)(hi >> (bshift - 64)) : (u32)((hi << (64 - bshift)) | (lo >> bshift));   // (WIDE: bshift >= 57)
    return (u32)(lo >> bshift);
  };
  auto rdlane64 = [&](u64 v, int l) -> u64 {
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, l), hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), l);
    return ((u64)hi << 32) | lo;
  };
  // the sub-bucket iteration Q visits (LIST: a vector load, uniform over the wave; ng: none)
  auto load_g = [&](u64 Q) -> u32 {
    if (Q >= np) return 0xFFFFFFFFu;
    return LIST ? nz[Q] : (u32)Q;
  };
  auto load_bvec = [&](u32 g) -> u64 {                                // lane 0: starts[g], the others: starts[g + 1]
    if (g == 0xFFFFFFFFu) return 0ull;
    return starts[(u64)g + (lane ? 1u : 0u)];
  };
  auto unpack_bounds = [&](u64 bv, u64 &a0, u32 &n) {
    a0 = rdlane64(bv, 0);
    const u64 d = rdlane64(bv, 1) - a0;
    n = d > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)d;
  };
  auto load_keys = [&](u64 a0, u32 n, KT (&kr)[KPT]) {
    const bool fits = (u64)n <= max_size;
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      if constexpr (IS96) { kr[j].w[0] = 0u; kr[j].w[1] = 0u; kr[j].w[2] = 0u; } else if constexpr (K16) { kr[j].lo = 0ull; kr[j].hi = 0ull; } else kr[j] = 0ull;
      if (fits && (u32)j * BLOCK < n) {
        const KT *src = keys + a0 + (u32)j * BLOCK;
        const u32 last = n - 1u - (u32)j * BLOCK;
        kr[j] = src[tid < last ? tid : last];
      }
    }
  };
  auto slots_for = [&](u32 n) -> u32 {
    u32 s = 256;
    while (s < n + n / 4 && s < (u32)SLOTS) s <<= 1;
    return s;
  };
  u64 klo[KPT], khi[K16 ? KPT : 1], pre_lo = 0, pre_hi = 0;           // the suffixes being counted; the k-mers' common top bits
  KT kcur[KPT];
  auto consume = [&]() {                                              // kcur -> klo/khi + prefix; nothing newer is in flight when its wait runs
    if constexpr (IS96) { pre_lo = KeyOps<K96>::low64(kcur[0]) & ~mask_lo; pre_hi = (u64)kcur[0].w[2] & ~mask_hi; }
    else if constexpr (K16) { pre_lo = kcur[0].lo & ~mask_lo; pre_hi = kcur[0].hi & ~mask_hi; }
    else pre_lo = kcur[0] & ~mask_lo;
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      if constexpr (IS96) {
        klo[j] = KeyOps<K96>::low64(kcur[j]) & mask_lo; khi[j] = (u64)kcur[j].w[2] & mask_hi;
        asm volatile("" : "+v"(klo[j]) :: "memory"); asm volatile("" : "+v"(khi[j]) :: "memory");
      } else if constexpr (K16) {
        klo[j] = kcur[j].lo & mask_lo; khi[j] = kcur[j].hi & mask_hi;
        asm volatile("" : "+v"(klo[j]) :: "memory"); asm volatile("" : "+v"(khi[j]) :: "memory");
      } else { klo[j] = kcur[j] & mask_lo; asm volatile("" : "+v"(klo[j]) :: "memory"); }
    }
  };

  u64 P = blockIdx.x;
  // pipeline: g of P, P+G, P+2G in scalars; gv = g of P+3G (vector register); bvec = bounds of P+2G; kcur = keys of P+G
  u32 g0 = (u32)__builtin_amdgcn_readfirstlane((int)load_g(P)), g1 = (u32)__builtin_amdgcn_readfirstlane((int)load_g(P + G)),
      g2 = (u32)__builtin_amdgcn_readfirstlane((int)load_g(P + 2 * G));
  u64 a0, a0n;
  u32 nc, nn;
  unpack_bounds(load_bvec(g0), a0, nc);
  load_keys(a0, nc, kcur);
  u64 bvec = load_bvec(g1);
  consume();
  unpack_bounds(bvec, a0n, nn);
  load_keys(a0n, nn, kcur);
  bvec = load_bvec(g2);
  u32 gv = load_g(P + 3 * G);
  u32 par = 0;
  u32 cleared = (u32)SLOTS;                            // tw[0, cleared) is EMPTY whenever an insert phase begins
  {
    uint4 *tw4 = reinterpret_cast<uint4 *>(tw);
    for (u32 i = tid; i < (u32)SLOTS / 4; i += BLOCK) tw4[i] = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
    s_bin[0][tid] = 0; if constexpr (DB) s_bin[1][tid] = 0;
    if (tid == 0) s_nd = 0;
  }
  __syncthreads();

  while (P < np) {
    const u32 n = nc;
    const bool active = n != 0 && (u64)n <= max_size;
    const u32 slots = slots_for(n);
    if (active) {
      u32 hh[KPT];
      const u32 smask = slots - 1, sshift = 32 - (u32)__builtin_ctz(slots);
      if (slots > cleared) {
        uint4 *tw4 = reinterpret_cast<uint4 *>(tw);
        for (u32 i = tid; i < (u32)SLOTS / 4; i += BLOCK) tw4[i] = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
      }
#pragma unroll
      for (int j = 0; j < KPT; j++) {
        const u64 mix = K16 ? (klo[j] ^ (khi[K16 ? j : 0] * 0xD6E8FEB86659FD93ull)) * 0x9E3779B97F4A7C15ull : klo[j] * 0x9E3779B97F4A7C15ull;
        hh[j] = (u32)(mix >> 32) >> sshift;
        if ((u32)j * BLOCK + tid < n) { dlo[(u32)j * BLOCK + tid] = klo[j]; if (WIDE) dhi[(u32)j * BLOCK + tid] = khi[K16 ? j : 0]; }
      }
      if constexpr (!DB) s_bin[0][tid] = 0;
      __syncthreads();                                 // the suffixes are staged: a probe compares with the claimer's
      auto same = [&](u32 rep, int j) -> bool { return dlo[rep] == klo[j] && (!WIDE || dhi[rep] == khi[K16 ? j : 0]); };
      u32 won = 0, pending = 0;
#pragma unroll
      for (int j = 0; j < KPT; j++) {
        if ((u32)j * BLOCK < n) {
          const bool act = (u32)j * BLOCK + tid < n;
          u32 old = 0;
          if (act) old = atomicCAS(&tw[hh[j]], EMPTY, (1u << 16) | ((u32)j * BLOCK + tid));
          const bool w = act && old == EMPTY;
          const bool dup = act && !w && same(old & 0xFFFFu, j);
          if (dup) atomicAdd(&tw[hh[j]], 1u << 16);
          won |= w ? (1u << j) : 0u;
          pending |= (act && !w && !dup) ? (1u << j) : 0u;
        }
      }
      while (pending) {
#pragma unroll
        for (int j = 0; j < KPT; j++) {
          if ((pending >> j) & 1u) {
            hh[j] = (hh[j] + 1) & smask;
            const u32 old = atomicCAS(&tw[hh[j]], EMPTY, (1u << 16) | ((u32)j * BLOCK + tid));
            const bool w = old == EMPTY;
            const bool dup = !w && same(old & 0xFFFFu, j);
            if (dup) atomicAdd(&tw[hh[j]], 1u << 16);
            if (w) won |= 1u << j;
            if (w || dup) pending &= ~(1u << j);
          }
        }
      }
      {
        u64 wm[KPT];
        u32 tot = 0;
#pragma unroll
        for (int j = 0; j < KPT; j++) { wm[j] = __ballot((won >> j) & 1u); tot += (u32)__popcll(wm[j]); }
        u32 base = 0;
        if (lane == 0) base = atomicAdd(&s_nd, tot);
        base = (u32)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
        for (int j = 0; j < KPT; j++) {
          if (wm[j]) {
            if ((won >> j) & 1u)
              lst[base + __builtin_amdgcn_mbcnt_hi((u32)(wm[j] >> 32), __builtin_amdgcn_mbcnt_lo((u32)wm[j], 0u))] = (unsigned short)hh[j];
            base += (u32)__popcll(wm[j]);
          }
        }
      }
      __syncthreads();
    }

    // the next sub-bucket's keys (loaded a whole iteration ago) take the place of this one's in klo/khi; only then the keys of
    // the one after it, the bounds of the one after that and (LIST) the number of the one after that are issued
    const u64 a = a0, pre_lo_cur = pre_lo, pre_hi_cur = pre_hi;
    const u32 g_cur = g0;
    const u32 nslots = (u64)nn <= max_size ? slots_for(nn) : 256u;
    consume();
    a0 = a0n; nc = nn;
    g0 = g1; g1 = g2; g2 = (u32)__builtin_amdgcn_readfirstlane((int)gv);
    unpack_bounds(bvec, a0n, nn);
    load_keys(a0n, nn, kcur);
    bvec = load_bvec(g2);
    gv = load_g(P + 4 * G);

    if (active) {
      const u32 D = s_nd;
      u32 wq[KPT], li[KPT];
#pragma unroll
      for (int q = 0; q < KPT; q++) {
        wq[q] = 0; li[q] = 0;
        if ((u32)q * BLOCK < D) {
          const u32 i = (u32)q * BLOCK + tid;
          if (i < D) {
            wq[q] = tw[lst[i]];
            const u32 rep = wq[q] & 0xFFFFu;
            li[q] = atomicAdd(&s_bin[par][bin_of(dlo[rep], WIDE ? dhi[rep] : 0ull)], 1u);
          }
        }
      }
      __syncthreads();
      if (tid < 64) {                                  // one wave scans the 256 bin counts: four per lane
        uint4 c = reinterpret_cast<uint4 *>(s_bin[par])[tid];
        const u32 s4 = c.x + c.y + c.z + c.w;
        u32 x = s4;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
        const u32 e0 = x - s4;
        reinterpret_cast<uint4 *>(s_bin[par])[tid] = make_uint4(e0, e0 + c.x, e0 + c.x + c.y, e0 + c.x + c.y + c.z);
        if (tid == 63) s_bin[par][BLOCK] = x;
      } else {                                         // ... the other three clear what the next insert phase uses
        uint4 *tw4 = reinterpret_cast<uint4 *>(tw);
        for (u32 i = tid - 64; i < nslots / 4; i += BLOCK - 64) tw4[i] = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
        if constexpr (DB) for (u32 i = tid - 64; i < BLOCK; i += BLOCK - 64) s_bin[par ^ 1u][i] = 0;
        if (tid == 64) s_nd = 0;
      }
      cleared = nslots;
      __syncthreads();
#pragma unroll
      for (int q = 0; q < KPT; q++) {
        if ((u32)q * BLOCK < D) {
          const u32 i = (u32)q * BLOCK + tid;
          if (i < D) { const u32 rep = wq[q] & 0xFFFFu; srt[s_bin[par][bin_of(dlo[rep], WIDE ? dhi[rep] : 0ull)] + li[q]] = wq[q]; }
        }
      }
      __syncthreads();
      // rank inside the bin, then back in place (every key of this region is staged in LDS)
      const u32 *sb = s_bin[par];
      KT *gk = keys + a;
      u32 *cout = cnt_tmp + a;
      for (u32 p = tid; p < D; p += BLOCK) {
        const u32 w = srt[p], rep = w & 0xFFFFu;
        const u64 kl = dlo[rep], kh = WIDE ? dhi[rep] : 0ull;
        const u32 b = bin_of(kl, kh), lo = sb[b], hi = sb[b + 1];
        u32 r = lo;
        for (u32 q = lo; q < hi; q++) {
          const u32 rq = srt[q] & 0xFFFFu;
          const u64 ql = dlo[rq], qh = WIDE ? dhi[rq] : 0ull;
          r += ((WIDE && qh < kh) || ((!WIDE || qh == kh) && ql < kl)) ? 1u : 0u;
        }
        if constexpr (IS96) { KT o; const u64 ol = pre_lo_cur | kl; o.w[0] = (u32)ol; o.w[1] = (u32)(ol >> 32); o.w[2] = (u32)(pre_hi_cur | kh); gk[r] = o; }
        else if constexpr (K16) { KT o; o.lo = pre_lo_cur | kl; o.hi = pre_hi_cur | kh; gk[r] = o; }
        else gk[r] = pre_lo_cur | kl;
        cout[r] = w >> 16;
      }
      if (tid == 0) group_distinct[tr_index(g_cur, tr_a, tr_b)] = D;
      if constexpr (DB) par ^= 1u;
      __syncthreads();                                 // the staged suffixes and the list are rewritten by the next sub-bucket
    } else if (tid == 0 && n == 0 && g_cur != 0xFFFFFFFFu) {
      group_distinct[tr_index(g_cur, tr_a, tr_b)] = 0; // (larger than max_size: the streaming launch owns it)
    }
    P += G;
  }
}

// value of v in lane `lane` (wave-uniform, e.g. taken from a ballot): v_readlane, no LDS round trip
__device__ __forceinline__ u64 read_lane_u64(u64 v, int lane) {
  const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, lane);
  const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), lane);
  return ((u64)hi << 32) | (u64)lo;
}

// In-place bitonic sort of N (a power of two) LDS records by the whole workgroup: gt(i, j) = record i is larger than record j,
// swp(i, j) exchanges them.  The all-pairs rank is D^2 compares -- 190 us for 3000 suffixes on one CU, VALU-bound -- so
// above BLOCK distinct suffixes the compacted list is sorted instead (log^2 N steps of N/2 compare-exchanges: ~25 us).
template <int BLOCK, typename GT, typename SWP>
__device__ __forceinline__ void bitonic_sort_lds(u32 N, GT gt, SWP swp) {
  for (u32 k = 2; k <= N; k <<= 1) {
    for (u32 j = k >> 1; j > 0; j >>= 1) {
      for (u32 t = threadIdx.x; t < N / 2; t += BLOCK) {
        const u32 i0 = ((t & ~(j - 1)) << 1) | (t & (j - 1));       // t with a zero inserted at j's bit
        const u32 i1 = i0 | j;
        const bool up = (i0 & k) == 0;
        if (gt(i0, i1) == up) swp(i0, i1);
      }
      __syncthreads();
    }
  }
}

// r[q] += number of keys among dk[0 .. 16*d16) below ki[q], q < ni (block-uniform).  dk is 16-byte aligned and padded to a
// multiple of 16 keys with all-ones (never smaller).  Whole 64-byte groups, several independent broadcast LDS reads in flight
// and every group compared against all of the thread's keys: a scalar loop of dependent LDS reads is latency-bound (measured:
// 390 us for 3000 keys by 1024 threads against 30 us this way).
template <int NI>
__device__ __forceinline__ void rank_below(const u32 *dk, u32 d16, const u32 (&ki)[NI], u32 (&r)[NI], u32 ni) {
  const uint4 *p = reinterpret_cast<const uint4 *>(dk);
  for (u32 j = 0; j < d16; j++) {
    const uint4 v0 = p[4 * j], v1 = p[4 * j + 1], v2 = p[4 * j + 2], v3 = p[4 * j + 3];
#pragma unroll
    for (int q = 0; q < NI; q++) {
      if ((u32)q < ni) {
        const u32 k = ki[q];
        r[q] += (v0.x < k ? 1u : 0u) + (v0.y < k ? 1u : 0u) + (v0.z < k ? 1u : 0u) + (v0.w < k ? 1u : 0u) +
                (v1.x < k ? 1u : 0u) + (v1.y < k ? 1u : 0u) + (v1.z < k ? 1u : 0u) + (v1.w < k ? 1u : 0u) +
                (v2.x < k ? 1u : 0u) + (v2.y < k ? 1u : 0u) + (v2.z < k ? 1u : 0u) + (v2.w < k ? 1u : 0u) +
                (v3.x < k ? 1u : 0u) + (v3.y < k ? 1u : 0u) + (v3.z < k ? 1u : 0u) + (v3.w < k ? 1u : 0u);
      }
    }
  }
}
template <int NI>
__device__ __forceinline__ void rank_below(const u64 *dk, u32 d16, const u64 (&ki)[NI], u32 (&r)[NI], u32 ni) {
  const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(dk);
  for (u32 j = 0; j < 2 * d16; j++) {
    const ulonglong2 v0 = p[4 * j], v1 = p[4 * j + 1], v2 = p[4 * j + 2], v3 = p[4 * j + 3];
#pragma unroll
    for (int q = 0; q < NI; q++) {
      if ((u32)q < ni) {
        const u64 k = ki[q];
        r[q] += (v0.x < k ? 1u : 0u) + (v0.y < k ? 1u : 0u) + (v1.x < k ? 1u : 0u) + (v1.y < k ? 1u : 0u) +
                (v2.x < k ? 1u : 0u) + (v2.y < k ? 1u : 0u) + (v3.x < k ? 1u : 0u) + (v3.y < k ? 1u : 0u);
      }
    }
  }
}
// 128-bit keys split into dlo/dhi (WIDE) or just dlo
template <int NI, bool WIDE>
__device__ __forceinline__ void rank_below128(const u64 *dlo, const u64 *dhi, u32 d16, const u64 (&kl)[NI], const u64 (&kh)[NI],
                                              u32 (&r)[NI], u32 ni) {
  if (!WIDE) { rank_below<NI>(dlo, d16, kl, r, ni); return; }
  const ulonglong2 *pl = reinterpret_cast<const ulonglong2 *>(dlo), *ph = reinterpret_cast<const ulonglong2 *>(dhi);
  for (u32 j = 0; j < 4 * d16; j++) {
    const ulonglong2 l0 = pl[2 * j], l1 = pl[2 * j + 1], h0 = ph[2 * j], h1 = ph[2 * j + 1];
#pragma unroll
    for (int q = 0; q < NI; q++) {
      if ((u32)q < ni) {
        const u64 a = kl[q], b = kh[q];
        r[q] += (((h0.x < b) || (h0.x == b && l0.x < a)) ? 1u : 0u) + (((h0.y < b) || (h0.y == b && l0.y < a)) ? 1u : 0u) +
                (((h1.x < b) || (h1.x == b && l1.x < a)) ? 1u : 0u) + (((h1.y < b) || (h1.y == b && l1.y < a)) ? 1u : 0u);
      }
    }
  }
}

// Hash-count of the sub-buckets above the persistent kernels' capacity (a k-mer present thousands of times with its error
// variants, a dense corner of the key space): one 1024-thread workgroup per entry of the large-sub-bucket list, entries at
// or below huge_min are somebody else's.  The table stores DISTINCT suffixes only, so the keys are streamed through it in
// rounds of BLOCK*KPT and any number of keys fits as long as at most CAP of them are distinct -- then one pass, the same
// compaction and all-pairs rank as hash_count_kernel, and the output goes in place.
// More than CAP distinct suffixes: the sub-bucket is done in several passes over ascending suffix RANGES [lo, hi], each
// pass streaming all keys and inserting only those of its range; the ranges come out in key order, so their outputs
// concatenate.  Every range first reaches for all that is left; when the table overflows, the range is cut at a quantile
// of the suffixes the table then holds (a fair sample: the keys of a sub-bucket are in input order) and retried.  Later
// passes still need the keys, so multi-pass output goes to alt[] (the sort's second buffer, free at this point) and is
// copied back at the end.
// KT = u32: narrowed keys in (launch_group_narrow), distinct suffixes out (u32, no prefix).
//
// GIGANTIC sub-buckets, in SLICES (round 6).  One workgroup streams ~0.45 keys per ns; a satellite family (one 171-base unit in 3 % of the
// genome: sub-buckets of 1.44 M instances of a few hundred distinct k-mers) kept ONE workgroup busy for 3 ms per sub-bucket and the count
// stage of a 10 Gbp step at 252 ms instead of 25 (profiles/r06_heavy_ab.txt); a human genome's alpha satellites are ten times that.  A
// sub-bucket above HUGE_SLICE_MIN keys is cut into slices of HUGE_SLICE keys (huge_plan_kernel):
//   MODE 1  one workgroup per SLICE: the same streaming passes over the slice's keys; the slice's distinct suffixes and their counts go
//           to alt[] as (suffix, count) pairs, packed behind the other slices' at the sub-bucket's place (a slice whose pairs would
//           not fit half its keys -- little repeats: a DENSE sub-bucket -- marks the sub-bucket; the keys are untouched);
//   MODE 2  one workgroup per sub-bucket: the same passes over the PAIRS, every insert adding the pair's count; the result goes in
//           place (the keys are no longer needed), ascending, as in MODE 0;
//   MODE 3  a marked (dense) sub-bucket -- the junction of a repeat with unique sequence: 1.75 M keys with 10^5 distinct suffixes, 30
//           passes of ONE workgroup over all of them, 120 ms -- by HUGE_RANGES workgroups, each streaming all keys and counting one
//           RANGE of the suffix space; a range's place in the output comes down a chain over the ranges in ascending order (work
//           items are taken by ticket, so a range's predecessor has always started); outputs go to alt[] (the others still read
//           the keys) and huge_copy_back_kernel brings them home.
// MODE 0 skips what the plan has cut (more than huge_max keys in at most HUGE_WMAX slices).
constexpr u32 HUGE_SLICE = 32768, HUGE_SLICE_MIN = 65536, HUGE_WMAX = 8192, HUGE_RANGES = 64;
// a slice with more distinct suffixes than 1/8 of its keys marks the sub-bucket DENSE: the merge of the pairs is ONE workgroup's work,
// it has to stay small (with 1/2: 1.3 ms per launch at k = 31, profiles/r06_heavy_ab.txt)
constexpr u32 HUGE_DENSE_DIV = 8;
struct HugeSliced {                       // device-side plan of one file's gigantic sub-buckets (huge_plan_kernel writes, MODE 1 / 2 / 3 read)
  u32 *counters;                          // [0] sub-buckets cut, [1] slices, [2] MODE 3's ticket, [3] MODE 1's ticket
  u32 *gig_g, *gig_pairs, *gig_fail, *gig_dist;   // per cut sub-bucket: its number, pairs its slices left, 1 = dense (MODE 3), distinct k-mers MODE 3 found
  u32 *slice_g, *slice_j, *slice_q;       // per slice: sub-bucket, index inside it, index of the cut sub-bucket
  u64 *chain;                             // [max_gig][HUGE_RANGES]: bit 63 set = the range is done, below: distinct k-mers up to and including it
  u64 *split;                             // [max_gig][HUGE_RANGES][2]: the largest suffix of a range (low, high word) -- quantiles of a sample (huge_split_kernel)
  u32 *error;                             // a chain wait that timed out (the count then ends with MGC_ETIMEOUT)
  u32 max_gig, max_slices;
};
__device__ __forceinline__ bool huge_is_cut(u64 n, u64 huge_max) { return huge_max != 0 && n > huge_max && (n + HUGE_SLICE - 1) / HUGE_SLICE <= (u64)HUGE_WMAX; }
// a cut sub-bucket of n keys: W slices of equal length (no ragged tail: a slice's pairs have to fit half its keys)
__device__ __forceinline__ u32 huge_slices(u64 n) { return (u32)((n + HUGE_SLICE - 1) / HUGE_SLICE); }
__device__ __forceinline__ u64 huge_slice_len(u64 n) { const u64 W = huge_slices(n); return (n + W - 1) / W; }

__global__ __launch_bounds__(256)
void huge_plan_kernel(const u64 *__restrict__ starts, const u32 *__restrict__ list, u64 n_list, u64 huge_max, HugeSliced hs,
                      u32 mark_dense = 0 /* 16-byte keys: every cut sub-bucket is counted by ranges (MODE 3), no slices */) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_list) return;
  const u64 g = list[i];
  const u64 n = starts[g + 1] - starts[g];
  if (!huge_is_cut(n, huge_max)) return;
  const u32 W = huge_slices(n);
  const u32 q = atomicAdd(&hs.counters[0], 1u);
  const u32 first = atomicAdd(&hs.counters[1], W);
  if (q >= hs.max_gig || first + W > hs.max_slices) return;            // (sized for the file: cannot happen; the single-workgroup form would be skipped too -- see launch)
  hs.gig_g[q] = (u32)g;
  if (mark_dense) { hs.gig_fail[q] = 1u; return; }
  for (u32 j = 0; j < W; j++) { hs.slice_g[first + j] = (u32)g; hs.slice_j[first + j] = j; hs.slice_q[first + j] = q; }
}

// The ranges of MODE 3 are QUANTILES of a sample, not equal parts of the suffix space: the suffixes of a sub-bucket cluster (the k-mers
// at a repeat's edge share their first p bases: at k = 51 all 10^5 distinct suffixes of such a sub-bucket lie in 4^-20 of the space, i.e.
// in ONE of 64 equal parts -- the first form of MODE 3 left one workgroup with all the work: 30 ms per launch, profiles/r06_heavy_ab.txt).
// One workgroup per dense sub-bucket: HUGE_SAMPLE suffixes at equal strides, bitonic sort in LDS, every (HUGE_SAMPLE / HUGE_RANGES)-th
// one is the largest suffix of its range; the last range ends at the mask.  Equal splitters (a heavy k-mer) make empty ranges.
constexpr u32 HUGE_SAMPLE = 4096;
__device__ __forceinline__ u128 huge_suffix128(u32 k, u128 m) { return (u128)k & m; }
__device__ __forceinline__ u128 huge_suffix128(u64 k, u128 m) { return (u128)k & m; }
__device__ __forceinline__ u128 huge_suffix128(const K128 &k, u128 m) { return KeyOps<K128>::v(k) & m; }
__device__ __forceinline__ u128 huge_suffix128(const K96 &k, u128 m) { return KeyOps<K96>::v(k) & m; }
template <typename KT>
__global__ __launch_bounds__(1024)
void huge_split_kernel(const KT *__restrict__ keys, const u64 *__restrict__ starts, u32 low_bits, HugeSliced hs) {
  __shared__ u64 slo[HUGE_SAMPLE], shi[HUGE_SAMPLE];
  const u32 q = blockIdx.x;
  if (q >= hs.counters[0] || q >= hs.max_gig || !hs.gig_fail[q]) return;
  const u64 g = hs.gig_g[q], a = starts[g], n = starts[g + 1] - a;
  const u128 low_mask = (low_bits >= 128) ? ~(u128)0 : (((u128)1 << low_bits) - 1);
  for (u32 i = threadIdx.x; i < HUGE_SAMPLE; i += 1024) {
    const u64 idx = (u64)(((u128)i * n) / HUGE_SAMPLE);
    const u128 sfx = huge_suffix128(keys[a + idx], low_mask);
    slo[i] = (u64)sfx; shi[i] = (u64)(sfx >> 64);
  }
  __syncthreads();
  bitonic_sort_lds<1024>(HUGE_SAMPLE, [&](u32 x, u32 y) { return shi[x] > shi[y] || (shi[x] == shi[y] && slo[x] > slo[y]); },
                         [&](u32 x, u32 y) { const u64 t = slo[x]; slo[x] = slo[y]; slo[y] = t; const u64 h = shi[x]; shi[x] = shi[y]; shi[y] = h; });
  if (threadIdx.x < HUGE_RANGES) {
    const u32 r = threadIdx.x;
    u64 *o = hs.split + ((u64)q * HUGE_RANGES + r) * 2;
    if (r == HUGE_RANGES - 1) { o[0] = (u64)low_mask; o[1] = (u64)(low_mask >> 64); }
    else { const u32 i = (r + 1) * (HUGE_SAMPLE / HUGE_RANGES) - 1; o[0] = slo[i]; o[1] = shi[i]; }
  }
}
// range r of cut sub-bucket q: [lo, hi] inclusive; false: empty
__device__ __forceinline__ bool huge_range(const HugeSliced &hs, u32 q, u32 r, u128 &lo, u128 &hi) {
  const u64 *o = hs.split + ((u64)q * HUGE_RANGES + r) * 2;
  hi = ((u128)o[1] << 64) | (u128)o[0];
  if (r == 0) { lo = 0; return true; }
  const u128 prev = ((u128)o[-1] << 64) | (u128)o[-2];
  if (prev >= hi) return false;
  lo = prev + 1;
  return true;
}

// (the body of the kernel below: `work` = the list entry (MODE 0), the slice item (MODE 1), the cut sub-bucket (MODE 2), the
// (cut sub-bucket, range) item (MODE 3))
template <typename S, int BLOCK, int CAP, int SLOTS, typename KT, int MODE>
__device__ __forceinline__
void hash_count_huge_body(u32 work, KT *__restrict__ keys, const u64 *__restrict__ starts, const u32 *__restrict__ list, u64 ng,
                          u64 huge_min, u32 low_bits, u32 *__restrict__ cnt_tmp, u64 *__restrict__ group_distinct,
                          KT *__restrict__ alt, u32 tr_a, u32 tr_b, u64 huge_max, const HugeSliced &hs) {
  static_assert((SLOTS & (SLOTS - 1)) == 0 && (CAP & (CAP - 1)) == 0 && SLOTS >= CAP * 2 && SLOTS % BLOCK == 0 && CAP % BLOCK == 0,
                "table geometry");
  constexpr int KPT = 4, SPT = SLOTS / BLOCK;
  const S EMPTY = ~(S)0;
  extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
  S   *tk = reinterpret_cast<S *>(hsm);                                   // [SLOTS]
  u32 *tc = reinterpret_cast<u32 *>(hsm + sizeof(S) * SLOTS);             // [SLOTS]
  S   *dk = reinterpret_cast<S *>(hsm + (sizeof(S) + 4) * SLOTS);         // [CAP + 16]
  u32 *dc = reinterpret_cast<u32 *>(hsm + (sizeof(S) + 4) * SLOTS + sizeof(S) * (CAP + 16));   // [CAP]
  constexpr int IPT = CAP / BLOCK;                     // distinct suffixes ranked per thread
  __shared__ u32 s_tmp[BLOCK / 64 + 1];
  __shared__ u32 s_st[3];                              // distinct in this pass, overflow, round of the overflow
  __shared__ u64 s_split;
  __shared__ u64 s_base;
  const u32 tid = threadIdx.x;
  u64 g, a, n64;                                       // the sub-bucket, where its keys (MODE 2: pairs) begin, how many of them this workgroup streams
  u64 a_sub = 0;                                       // MODE 1: where the whole sub-bucket begins
  u32 cq = 0, rr = 0;                                  // MODE 1 / 2 / 3: the cut sub-bucket; MODE 3: the range
  const u64 low_mask = (low_bits >= 64) ? ~0ull : ((1ull << low_bits) - 1ull);
  u64 range_lo = 0, range_hi = low_mask;               // the suffixes this workgroup counts (MODE 3: one of HUGE_RANGES equal parts)
  if constexpr (MODE == 0) {
    g = list[work];
    a = starts[g]; n64 = starts[g + 1] - a;
    if (n64 <= huge_min) return;
    if (huge_is_cut(n64, huge_max)) return;
  } else if constexpr (MODE == 1) {
    const u32 item = work;
    if (item >= hs.counters[1] || item >= hs.max_slices) return;
    g = hs.slice_g[item]; cq = hs.slice_q[item];
    a_sub = starts[g];
    const u64 nall = starts[g + 1] - a_sub, len = huge_slice_len(nall), off = (u64)hs.slice_j[item] * len;
    a = a_sub + off; n64 = nall - off < len ? nall - off : len;
  } else if constexpr (MODE == 2) {
    cq = work;
    if (cq >= hs.counters[0] || cq >= hs.max_gig || hs.gig_fail[cq]) return;
    g = hs.gig_g[cq];
    a = starts[g];
    n64 = hs.gig_pairs[cq];                             // pairs to stream: alt[a + 2 i], alt[a + 2 i + 1]
  } else {
    cq = work / HUGE_RANGES; rr = work % HUGE_RANGES;
    if (cq >= hs.counters[0] || cq >= hs.max_gig || !hs.gig_fail[cq]) return;
    g = hs.gig_g[cq];
    a = starts[g]; n64 = starts[g + 1] - a;
    // the range: between two quantiles of the sub-bucket's sample (huge_split_kernel); an empty one still passes the chain on
    u128 rl, rh;
    if (huge_range(hs, cq, rr, rl, rh)) { range_lo = (u64)rl; range_hi = (u64)rh; }
    else { range_lo = 1; range_hi = 0; n64 = 0; }
  }
  const u64 prefix = (sizeof(KT) == 8 && MODE != 1) ? ((u64)keys[starts[g]] & ~low_mask) : 0ull;   // whole keys: the sub-bucket's first key tells (read before anything is written)
  constexpr u32 smask = SLOTS - 1, sshift = 32 - __builtin_ctz((unsigned)SLOTS);
  KT *gk = keys + a;
  const u64 rounds = (n64 + (u64)BLOCK * KPT - 1) / ((u64)BLOCK * KPT);
  u64 lo = range_lo, hi = range_hi;                    // suffix range of this pass, inclusive
  u64 chain_base = 0;                                  // MODE 3: distinct k-mers of the ranges below this one (known from the first emit on)
  bool chain_known = MODE != 3 || rr == 0, chain_sent = false;
  auto chain_wait = [&]() __attribute__((always_inline)) {
    if constexpr (MODE == 3) {
      if (chain_known) return;
      if (tid == 0) {
        const u64 *c = hs.chain + (u64)cq * HUGE_RANGES + (rr - 1);
        u64 v = 0;
        u32 spins = 0;
        while (!((v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 63)) {
          if (++spins > (1u << 26)) { atomicExch(hs.error, 1u); break; }
          __builtin_amdgcn_s_sleep(8);
        }
        s_base = v & ~(1ull << 63);
      }
      __syncthreads();
      chain_base = s_base;
      chain_known = true;
      __syncthreads();
    }
  };
  u64 out = 0;                                         // distinct k-mers written by the passes before
  bool in_place = false;
  for (;;) {
    for (u32 i = tid; i < (u32)SLOTS; i += BLOCK) { tk[i] = EMPTY; tc[i] = 0u; }
    if (tid < 3) s_st[tid] = 0u;
    __syncthreads();
    const u64 span = hi - lo;
    u64 rd = 0;
    // MODE 0 / 1 / 3: a stream of n64 keys; MODE 2: of (suffix, count) pairs in alt[] at the sub-bucket's place
    {
    const u64 sn = n64;                                                  // items of the stream
    const KT *src = MODE == 2 ? alt + a : gk;                            // MODE 2: pair i = src[2 i], src[2 i + 1]
    // the keys of the next round are in flight while this one goes through the table (one workgroup per CU: nothing else
    // would hide the memory round trip)
    u64 raw[KPT];
    u32 rwt[KPT];
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const u64 idx = (u64)j * BLOCK + tid;
      if constexpr (MODE == 2) { raw[j] = (idx < sn) ? (u64)src[2 * idx] : 0ull; rwt[j] = (idx < sn) ? (u32)src[2 * idx + 1] : 0u; }
      else { raw[j] = (idx < sn) ? (u64)src[idx] : 0ull; rwt[j] = 1u; }
    }
    for (u64 base = 0; base < sn; base += (u64)BLOCK * KPT, rd++) {
      u64 nxt[KPT];
      u32 nwt[KPT];
#pragma unroll
      for (int j = 0; j < KPT; j++) {
        const u64 idx = base + (u64)BLOCK * KPT + (u64)j * BLOCK + tid;
        if constexpr (MODE == 2) { nxt[j] = (idx < sn) ? (u64)src[2 * idx] : 0ull; nwt[j] = (idx < sn) ? (u32)src[2 * idx + 1] : 0u; }
        else { nxt[j] = (idx < sn) ? (u64)src[idx] : 0ull; nwt[j] = 1u; }
      }
      S   kk[KPT];
      u32 hh[KPT], pending = 0;
      u32 w[KPT];
#pragma unroll
      for (int j = 0; j < KPT; j++) {
        const u64 idx = base + (u64)j * BLOCK + tid;
        const u64 sfx = raw[j] & low_mask;
        w[j] = rwt[j];
        raw[j] = nxt[j]; rwt[j] = nwt[j];
        kk[j] = (S)sfx;
        hh[j] = (u32)((sfx * 0x9E3779B97F4A7C15ull) >> 32) >> sshift;
        if (idx < sn && sfx - lo <= span) pending |= 1u << j;
      }
      // a heavy k-mer fills whole waves with one suffix, and LDS atomics on one address serialize: lanes holding the
      // same suffix as the first lane not yet looked at become one insert of their number (a few rounds: the first lane
      // may hold an error variant).  (MODE 2: the pairs of a slice are distinct suffixes: nothing to merge.)
      if constexpr (MODE != 2) {
#pragma unroll
      for (int j = 0; j < KPT; j++) {
        u64 rem = __ballot((pending >> j) & 1u);
        for (int it = 0; it < 2 && rem; it++) {        // wave-uniform
          const int leader = __builtin_ctzll(rem);
          const S k0 = (S)read_lane_u64((u64)kk[j], leader);
          const u64 same = __ballot(((rem >> lane_id()) & 1ull) && kk[j] == k0);
          if ((int)lane_id() == leader) w[j] = (u32)__popcll(same);
          else if ((same >> lane_id()) & 1ull) pending &= ~(1u << j);
          rem &= ~same;
          if (__popcll(same) >= 8) break;              // that was the heavy one
        }
      }
      }
      while (pending && !__hip_atomic_load(&s_st[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
        // all CASes of a trip are issued before the first answer is looked at: their LDS round trips overlap
        S old[KPT];
#pragma unroll
        for (int j = 0; j < KPT; j++) {
          old[j] = EMPTY;
          if ((pending >> j) & 1u) old[j] = atomicCAS(&tk[hh[j]], EMPTY, kk[j]);
        }
#pragma unroll
        for (int j = 0; j < KPT; j++) {
          if ((pending >> j) & 1u) {
            if (old[j] == EMPTY || old[j] == kk[j]) {
              atomicAdd(&tc[hh[j]], w[j]);
              pending &= ~(1u << j);
              if (old[j] == EMPTY && atomicAdd(&s_st[0], 1u) >= (u32)CAP) { s_st[2] = (u32)rd; s_st[1] = 1u; }
            }
            else hh[j] = (hh[j] + 1) & smask;
          }
        }
      }
      if (__hip_atomic_load(&s_st[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
    }
    }
    __syncthreads();
    const u32 overflowed = s_st[1], at_round = s_st[2];
    u32 occ = 0;
#pragma unroll
    for (int j = 0; j < SPT; j++) occ |= (tc[(u32)j * BLOCK + tid] != 0u ? 1u : 0u) << j;
    u32 D;
    u32 o = block_excl_scan<BLOCK, u32>(__popc(occ), s_tmp, &D);
    if (overflowed) {
      // The table filled up after x = (at_round+1)/rounds of the stream.  Its suffixes are a fair sample of the range's
      // distinct ones (slot order is hash order, the keys are in input order), wherever in the range they cluster -- the
      // variants of a heavy k-mer sit within 4^p of it at every scale p, so numeric halving would take one retry per bit.
      // New upper end = the sample's quantile below which the whole stream should bring at most 0.7 CAP distinct
      // suffixes even if everything still to come is new.  Always below the sample's maximum, so the range shrinks.
      const u32 Ds = D < (u32)BLOCK ? D : (u32)BLOCK;
#pragma unroll
      for (int j = 0; j < SPT; j++)
        if ((occ >> j) & 1u) { if (o < (u32)BLOCK) dk[o] = tk[(u32)j * BLOCK + tid]; o++; }
      if (tid < 16) dk[Ds + tid] = EMPTY;
      __syncthreads();
      u64 q = ((u64)Ds * 7u * ((u64)at_round + 1)) / (10u * rounds);
      if (q > (u64)Ds - 2) q = (u64)Ds - 2;
      {
        S ks[1] = {dk[tid < Ds ? tid : 0]};
        u32 rs[1] = {0u};
        rank_below<1>(dk, (Ds + 15) / 16, ks, rs, 1u);
        if (tid < Ds && rs[0] == (u32)q) s_split = (u64)ks[0];
      }
      __syncthreads();
      hi = s_split;
      __syncthreads();                                 // s_st and s_split are rewritten
      continue;
    }
#pragma unroll
    for (int j = 0; j < SPT; j++)
      if ((occ >> j) & 1u) { dk[o] = tk[(u32)j * BLOCK + tid]; dc[o] = tc[(u32)j * BLOCK + tid]; o++; }
    if (tid < 16) dk[D + tid] = EMPTY;
    __syncthreads();
    u64 emit_at = out;                                  // where this pass's results begin
    if constexpr (MODE == 1) {
      // the slice's pairs must fit half its keys (otherwise little repeats -- a dense sub-bucket: MODE 3's): then all slices' pairs fit
      // the sub-bucket's place in alt[], packed in the order the passes of the slices come by
      if (HUGE_DENSE_DIV * (out + D) > n64) { if (tid == 0) hs.gig_fail[cq] = 1u; return; }
      if (tid == 0) s_base = (u64)atomicAdd(&hs.gig_pairs[cq], D);
      __syncthreads();
      emit_at = s_base;
      __syncthreads();
    }
    if constexpr (MODE == 3) {
      chain_wait(); emit_at = chain_base + out;
      // the range's LAST pass: its total is known before anything is sorted or written -- the chain moves on now, the next ranges
      // do not wait for this one's sort and stores (published at the end, the chain was 64 sorts long: 4 ms per dense sub-bucket)
      if (hi == range_hi && !chain_sent) {
        if (tid == 0) __hip_atomic_store(hs.chain + (u64)cq * HUGE_RANGES + rr, (chain_base + out + D) | (1ull << 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        chain_sent = true;
      }
    }
    // one pass over everything: every key of the sub-bucket went through the table, the output can go in place
    // (MODE 2: always -- the input is the pairs in alt[]; MODE 1: never -- pairs, to alt[]; MODE 3: never -- the other ranges read the keys)
    in_place = MODE == 2 || (MODE == 0 && lo == 0 && hi == low_mask);
    KT *dst = MODE == 1 ? alt + a_sub : (in_place ? gk : alt + a);
    auto emit = [&](u64 r, u64 sfx, u32 c) __attribute__((always_inline)) {
      if constexpr (MODE == 1) { dst[2 * r] = (KT)sfx; dst[2 * r + 1] = (KT)c; }
      else { dst[r] = (KT)(prefix | sfx); cnt_tmp[a + r] = c; }
    };
    if (D > (u32)BLOCK) {
      u32 N = 2 * BLOCK;
      while (N < D) N <<= 1;                           // <= CAP, a power of two
      for (u32 i = D + tid; i < N; i += BLOCK) dk[i] = EMPTY;
      __syncthreads();
      bitonic_sort_lds<BLOCK>(N, [&](u32 x, u32 y) { return dk[x] > dk[y]; },
                              [&](u32 x, u32 y) { const S t = dk[x]; dk[x] = dk[y]; dk[y] = t;
                                                  const u32 c = dc[x]; dc[x] = dc[y]; dc[y] = c; });
      for (u32 i = tid; i < D; i += BLOCK) emit(emit_at + i, (u64)dk[i], dc[i]);
    } else {
      S   ks[IPT];
      u32 rs[IPT];
#pragma unroll
      for (int q = 0; q < IPT; q++) { const u32 i = (u32)q * BLOCK + tid; ks[q] = dk[i < D ? i : 0]; rs[q] = 0u; }
      rank_below<IPT>(dk, (D + 15) / 16, ks, rs, (D + BLOCK - 1) / BLOCK);
#pragma unroll
      for (int q = 0; q < IPT; q++) {
        const u32 i = (u32)q * BLOCK + tid;
        if (i < D) emit(emit_at + rs[q], (u64)ks[q], dc[i]);
      }
    }
    out += D;
    if (hi == range_hi) break;
    lo = hi + 1;                                       // next: everything that is left; an overflow will say where to cut
    hi = range_hi;
    __syncthreads();                                   // dk/dc and s_st are reused
  }
  if constexpr (MODE == 1) return;
  if constexpr (MODE == 3) {
    // pass the chain on: the distinct k-mers up to and including this range; the last range knows the sub-bucket's
    chain_wait();
    if (tid == 0) {
      const u64 total = chain_base + out;
      if (!chain_sent) __hip_atomic_store(hs.chain + (u64)cq * HUGE_RANGES + rr, total | (1ull << 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (rr == HUGE_RANGES - 1) { hs.gig_dist[cq] = (u32)total; group_distinct[tr_index(g, tr_a, tr_b)] = total; }
    }
    return;
  }
  if (!in_place) {
    __syncthreads();                                   // all passes have read the keys; alt[] was written by this workgroup
    for (u64 i = tid; i < out; i += BLOCK) gk[i] = alt[a + i];
  }
  if (tid == 0) group_distinct[tr_index(g, tr_a, tr_b)] = out;
}

// MODE 0 / 2: one workgroup per list entry / cut sub-bucket.  MODE 1 / 3: the plan lives on the device -- how many slices, how many
// dense sub-buckets, the host does not know -- so a SMALL grid of workgroups takes work items by ticket until none is left (a launch
// sized for the worst case was thousands of 1024-thread workgroups with 96 KiB of LDS each that looked at a counter and left: 1 ms per
// file, profiles/r06_heavy_ab.txt).  MODE 3's chain stays safe: a ticket is only ever held by a running workgroup.
template <typename S, int BLOCK, int CAP, int SLOTS, typename KT = u64, int MODE = 0>
__global__ __launch_bounds__(BLOCK)
void hash_count_huge_kernel(KT *__restrict__ keys, const u64 *__restrict__ starts, const u32 *__restrict__ list, u64 ng,
                            u64 huge_min, u32 low_bits, u32 *__restrict__ cnt_tmp, u64 *__restrict__ group_distinct,
                            KT *__restrict__ alt, u32 tr_a = 0, u32 tr_b = 0 /* tr_index() of the sub-bucket numbers */,
                            u64 huge_max = 0 /* MODE 0: sub-buckets above it belong to the sliced form */,
                            HugeSliced hs = HugeSliced()) {
  if constexpr (MODE == 0 || MODE == 2) {
    hash_count_huge_body<S, BLOCK, CAP, SLOTS, KT, MODE>(blockIdx.x, keys, starts, list, ng, huge_min, low_bits, cnt_tmp, group_distinct, alt, tr_a, tr_b, huge_max, hs);
  } else {
    __shared__ u32 s_ticket;
    for (;;) {
      __syncthreads();                                 // (everything of the item before has left shared memory)
      if (threadIdx.x == 0) s_ticket = atomicAdd(&hs.counters[MODE == 1 ? 3 : 2], 1u);
      __syncthreads();
      const u32 item = s_ticket;
      const u32 limit = MODE == 1 ? hs.counters[1] : hs.counters[0] * HUGE_RANGES;
      if (item >= limit) return;
      hash_count_huge_body<S, BLOCK, CAP, SLOTS, KT, MODE>(item, keys, starts, list, ng, huge_min, low_bits, cnt_tmp, group_distinct, alt, tr_a, tr_b, huge_max, hs);
    }
  }
}

// The same for 16-byte keys (k >= 33) and 12-byte K96 records: slots are claimed through their count word, lanes of a
// wave that hold the first active lane's suffix are merged into one weighted insert, ranges are 128-bit.
// MODE 1 / 2 / 3 (round 6): a GIGANTIC sub-bucket (above HUGE_SLICE_MIN keys: huge_plan_kernel cuts it, MODE 0 leaves it alone) in slices
// + merge, the dense ones by HUGE_RANGES workgroups over quantile ranges of the suffix space -- hash_count_huge_kernel's three modes
// with 128-bit suffixes; a (suffix, count) pair is two records in alt[].
template <int BLOCK, int CAP, int SLOTS, bool WIDE, typename KT, int MODE>   // KT: K128, or 12-byte K96 records
__device__ __forceinline__
void hash_count128_huge_body(u32 work, KT *__restrict__ keys, const u64 *__restrict__ starts, const u32 *__restrict__ list, u64 ng,
                             u64 huge_min, u32 low_bits, u32 *__restrict__ cnt_tmp, u64 *__restrict__ group_distinct,
                             KT *__restrict__ alt, u32 tr_a, u32 tr_b, u64 huge_max, const HugeSliced &hs) {
  static_assert((SLOTS & (SLOTS - 1)) == 0 && SLOTS >= CAP * 2 && SLOTS % BLOCK == 0 && CAP % BLOCK == 0, "table geometry");
  constexpr int KPT = 2, SPT = SLOTS / BLOCK;
  constexpr u32 LOCK = 0xFFFFFFFFu;
  extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
  u64 *tlo = reinterpret_cast<u64 *>(hsm);                                 // [SLOTS]
  u64 *thi = tlo + SLOTS;                                                  // [WIDE ? SLOTS : 0]
  u64 *dlo = thi + (WIDE ? SLOTS : 0);                                     // [CAP + 16]
  u64 *dhi = dlo + (CAP + 16);                                             // [WIDE ? CAP + 16 : 0]
  u32 *tc  = reinterpret_cast<u32 *>(dhi + (WIDE ? CAP + 16 : 0));         // [SLOTS]
  constexpr int IPT = CAP / BLOCK;
  u32 *dc  = tc + SLOTS;                                                   // [CAP]
  __shared__ u32 s_tmp[BLOCK / 64 + 1];
  __shared__ u32 s_st[3];                              // distinct in this pass, overflow, round of the overflow
  __shared__ u64 s_split[2];
  __shared__ u64 s_base;
  using KO = KeyOps<KT>;
  const u32 tid = threadIdx.x;
  const u128 low_mask = (low_bits >= 128) ? ~(u128)0 : (((u128)1 << low_bits) - 1);
  u64 g, a, n64;
  u64 a_sub = 0;                                       // MODE 1: where the whole sub-bucket begins
  u32 cq = 0, rr = 0;                                  // MODE 1 / 2 / 3: the cut sub-bucket; MODE 3: the range
  u128 range_lo = 0, range_hi = low_mask;
  if constexpr (MODE == 0) {
    g = list[work];
    a = starts[g]; n64 = starts[g + 1] - a;
    if (n64 <= huge_min) return;
    if (huge_is_cut(n64, huge_max)) return;
  } else if constexpr (MODE == 1) {                    // a slice: (suffix, count) pairs -- two records each -- to alt[], packed per sub-bucket
    if (work >= hs.counters[1] || work >= hs.max_slices) return;
    g = hs.slice_g[work]; cq = hs.slice_q[work];
    a_sub = starts[g];
    const u64 nall = starts[g + 1] - a_sub, len = huge_slice_len(nall), off = (u64)hs.slice_j[work] * len;
    a = a_sub + off; n64 = nall - off < len ? nall - off : len;
  } else if constexpr (MODE == 2) {                    // the merge of a sub-bucket's pairs
    cq = work;
    if (cq >= hs.counters[0] || cq >= hs.max_gig || hs.gig_fail[cq]) return;
    g = hs.gig_g[cq];
    a = starts[g];
    n64 = hs.gig_pairs[cq];
  } else {
    cq = work / HUGE_RANGES; rr = work % HUGE_RANGES;
    if (cq >= hs.counters[0] || cq >= hs.max_gig || !hs.gig_fail[cq]) return;
    g = hs.gig_g[cq];
    a = starts[g]; n64 = starts[g + 1] - a;
    // the range: between two quantiles of the sub-bucket's sample (huge_split_kernel); an empty one still passes the chain on
    if (!huge_range(hs, cq, rr, range_lo, range_hi)) { range_lo = 1; range_hi = 0; n64 = 0; }
  }
  const u128 prefix = MODE == 1 ? (u128)0 : (KO::v(keys[starts[g]]) & ~low_mask);
  constexpr u32 smask = SLOTS - 1, sshift = 32 - __builtin_ctz((unsigned)SLOTS);
  KT *gk = keys + a;
  const u64 rounds = (n64 + (u64)BLOCK * KPT - 1) / ((u64)BLOCK * KPT);
  u128 lo = range_lo, hi = range_hi;                   // suffix range of this pass, inclusive
  u64 out = 0;
  bool in_place = false;
  u64 chain_base = 0;                                  // MODE 3: distinct k-mers of the ranges below this one
  bool chain_known = MODE != 3 || rr == 0, chain_sent = false;
  auto chain_wait = [&]() __attribute__((always_inline)) {
    if constexpr (MODE == 3) {
      if (chain_known) return;
      if (tid == 0) {
        const u64 *c = hs.chain + (u64)cq * HUGE_RANGES + (rr - 1);
        u64 v = 0;
        u32 spins = 0;
        while (!((v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 63)) {
          if (++spins > (1u << 26)) { atomicExch(hs.error, 1u); break; }
          __builtin_amdgcn_s_sleep(8);
        }
        s_base = v & ~(1ull << 63);
      }
      __syncthreads();
      chain_base = s_base;
      chain_known = true;
      __syncthreads();
    }
  };
  for (;;) {
    for (u32 i = tid; i < (u32)SLOTS; i += BLOCK) tc[i] = 0u;
    if (tid < 3) s_st[tid] = 0u;
    __syncthreads();
    const u128 span = hi - lo;
    const KT *src = MODE == 2 ? alt + a : gk;         // MODE 2: pair i = src[2 i] (the suffix), src[2 i + 1] (its count)
    KT raw[KPT];                                     // next round's keys in flight while this one goes through the table
    u32 rwt[KPT];
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const u64 idx = (u64)j * BLOCK + tid;
      raw[j] = KO::zero(); rwt[j] = 1u;
      if (idx < n64) { if constexpr (MODE == 2) { raw[j] = src[2 * idx]; rwt[j] = (u32)KO::v(src[2 * idx + 1]); } else raw[j] = src[idx]; }
    }
    for (u64 base = 0, rd = 0; base < n64; base += (u64)BLOCK * KPT, rd++) {
      KT nxt[KPT];
      u32 nwt[KPT];
#pragma unroll
      for (int j = 0; j < KPT; j++) {
        const u64 idx = base + (u64)BLOCK * KPT + (u64)j * BLOCK + tid;
        nxt[j] = KO::zero(); nwt[j] = 1u;
        if (idx < n64) { if constexpr (MODE == 2) { nxt[j] = src[2 * idx]; nwt[j] = (u32)KO::v(src[2 * idx + 1]); } else nxt[j] = src[idx]; }
      }
      u64 klo[KPT], khi[KPT];
      u32 hh[KPT], w[KPT], pending = 0;
#pragma unroll
      for (int j = 0; j < KPT; j++) {
        const u64 idx = base + (u64)j * BLOCK + tid;
        const u128 sfx = KO::v(raw[j]) & low_mask;
        w[j] = rwt[j];
        raw[j] = nxt[j]; rwt[j] = nwt[j];
        klo[j] = (u64)sfx; khi[j] = (u64)(sfx >> 64);
        const u64 mix = (klo[j] ^ (khi[j] * 0xD6E8FEB86659FD93ull)) * 0x9E3779B97F4A7C15ull;
        hh[j] = (u32)(mix >> 32) >> sshift;
        if (idx < n64 && sfx - lo <= span) pending |= 1u << j;
      }
      // a heavy k-mer fills whole waves with one suffix: the lanes holding the first active lane's suffix become one
      // insert of their number (MODE 2: the pairs of a slice are distinct suffixes -- little to merge)
      if constexpr (MODE != 2)
#pragma unroll
      for (int j = 0; j < KPT; j++) {
        u64 rem = __ballot((pending >> j) & 1u);
        for (int it = 0; it < 2 && rem; it++) {        // wave-uniform; a second try: the first lane may hold an error variant
          const int leader = __builtin_ctzll(rem);
          const u64 l0 = read_lane_u64(klo[j], leader), h0 = read_lane_u64(khi[j], leader);
          const u64 same = __ballot(((rem >> lane_id()) & 1ull) && klo[j] == l0 && khi[j] == h0);
          if ((int)lane_id() == leader) w[j] = (u32)__popcll(same);
          else if ((same >> lane_id()) & 1ull) pending &= ~(1u << j);
          rem &= ~same;
          if (__popcll(same) >= 8) break;              // that was the heavy one
        }
      }
      while (pending && !__hip_atomic_load(&s_st[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
#pragma unroll
        for (int j = 0; j < KPT; j++) {
          if ((pending >> j) & 1u) {
            const u32 h = hh[j];
            const u32 old = atomicCAS(&tc[h], 0u, LOCK);
            if (old == 0u) {                           // the slot is ours: fill it, then publish it with its count
              tlo[h] = klo[j];
              if (WIDE) thi[h] = khi[j];
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
              __hip_atomic_store(&tc[h], w[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              pending &= ~(1u << j);
              if (atomicAdd(&s_st[0], 1u) >= (u32)CAP) { s_st[2] = (u32)rd; s_st[1] = 1u; }
            } else if (old != LOCK) {                  // valid: same suffix -> count it, another one -> probe on
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
              const bool same = (tlo[h] == klo[j]) && (!WIDE || thi[h] == khi[j]);
              if (same) { atomicAdd(&tc[h], w[j]); pending &= ~(1u << j); }
              else hh[j] = (h + 1) & smask;
            }                                          // LOCK: somebody is writing this slot; look again next round
          }
        }
      }
      if (__hip_atomic_load(&s_st[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
    }
    __syncthreads();
    const u32 overflowed = s_st[1], at_round = s_st[2];
    u32 occ = 0;
#pragma unroll
    for (int j = 0; j < SPT; j++) occ |= (tc[(u32)j * BLOCK + tid] != 0u ? 1u : 0u) << j;
    u32 D;
    u32 o = block_excl_scan<BLOCK, u32>(__popc(occ), s_tmp, &D);
    if (overflowed) {                                  // new upper end = a quantile of the table's suffixes (see above)
      const u32 Ds = D < (u32)BLOCK ? D : (u32)BLOCK;
#pragma unroll
      for (int j = 0; j < SPT; j++) {
        if ((occ >> j) & 1u) {
          const u32 sl = (u32)j * BLOCK + tid;
          if (o < (u32)BLOCK) { dlo[o] = tlo[sl]; if (WIDE) dhi[o] = thi[sl]; }
          o++;
        }
      }
      if (tid < 16) { dlo[Ds + tid] = ~0ull; if (WIDE) dhi[Ds + tid] = ~0ull; }
      __syncthreads();
      u64 q = ((u64)Ds * 7u * ((u64)at_round + 1)) / (10u * rounds);
      if (q > (u64)Ds - 2) q = (u64)Ds - 2;
      {
        const u32 i = tid < Ds ? tid : 0;
        u64 kl[1] = {dlo[i]}, kh[1] = {WIDE ? dhi[i] : 0ull};
        u32 rs[1] = {0u};
        rank_below128<1, WIDE>(dlo, dhi, (Ds + 15) / 16, kl, kh, rs, 1u);
        if (tid < Ds && rs[0] == (u32)q) { s_split[0] = kl[0]; s_split[1] = kh[0]; }
      }
      __syncthreads();
      hi = ((u128)s_split[1] << 64) | (u128)s_split[0];
      __syncthreads();
      continue;
    }
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
    if (tid < 16) { dlo[D + tid] = ~0ull; if (WIDE) dhi[D + tid] = ~0ull; }
    __syncthreads();
    u64 emit_at = out;
    if constexpr (MODE == 1) {
      if (HUGE_DENSE_DIV * (out + D) > n64) { if (tid == 0) hs.gig_fail[cq] = 1u; return; }
      if (tid == 0) s_base = (u64)atomicAdd(&hs.gig_pairs[cq], D);
      __syncthreads();
      emit_at = s_base;
      __syncthreads();
    }
    if constexpr (MODE == 3) {
      chain_wait(); emit_at = chain_base + out;
      if (hi == range_hi && !chain_sent) {             // (the range's last pass: the chain moves on before the sort and the stores)
        if (tid == 0) __hip_atomic_store(hs.chain + (u64)cq * HUGE_RANGES + rr, (chain_base + out + D) | (1ull << 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        chain_sent = true;
      }
    }
    in_place = MODE == 2 || (MODE == 0 && lo == 0 && hi == low_mask);
    KT *dst = MODE == 1 ? alt + a_sub : (in_place ? gk : alt + a);
    auto emit = [&](u64 r, u128 sfx, u32 c) __attribute__((always_inline)) {
      if constexpr (MODE == 1) { dst[2 * r] = KO::mk(sfx); dst[2 * r + 1] = KO::mk((u128)c); }
      else { dst[r] = KO::mk(prefix | sfx); cnt_tmp[a + r] = c; }
    };
    if (D > (u32)BLOCK) {
      u32 N = 2 * BLOCK;
      while (N < D) N <<= 1;
      for (u32 i = D + tid; i < N; i += BLOCK) { dlo[i] = ~0ull; if (WIDE) dhi[i] = ~0ull; }
      __syncthreads();
      bitonic_sort_lds<BLOCK>(N, [&](u32 x, u32 y) { return WIDE ? (dhi[x] > dhi[y] || (dhi[x] == dhi[y] && dlo[x] > dlo[y]))
                                                                 : (dlo[x] > dlo[y]); },
                              [&](u32 x, u32 y) { const u64 t = dlo[x]; dlo[x] = dlo[y]; dlo[y] = t;
                                                  if (WIDE) { const u64 h = dhi[x]; dhi[x] = dhi[y]; dhi[y] = h; }
                                                  const u32 c = dc[x]; dc[x] = dc[y]; dc[y] = c; });
      for (u32 i = tid; i < D; i += BLOCK) {
        emit(emit_at + i, ((u128)(WIDE ? dhi[i] : 0ull) << 64) | (u128)dlo[i], dc[i]);
      }
    } else {
      u64 kl[IPT], kh[IPT];
      u32 rs[IPT];
#pragma unroll
      for (int q = 0; q < IPT; q++) {
        const u32 i = (u32)q * BLOCK + tid, ii = i < D ? i : 0;
        kl[q] = dlo[ii]; kh[q] = WIDE ? dhi[ii] : 0ull; rs[q] = 0u;
      }
      rank_below128<IPT, WIDE>(dlo, dhi, (D + 15) / 16, kl, kh, rs, (D + BLOCK - 1) / BLOCK);
#pragma unroll
      for (int q = 0; q < IPT; q++) {
        const u32 i = (u32)q * BLOCK + tid;
        if (i < D) {
          emit(emit_at + rs[q], ((u128)kh[q] << 64) | (u128)kl[q], dc[i]);
        }
      }
    }
    out += D;
    if (hi == range_hi) break;
    lo = hi + 1;
    hi = range_hi;
    __syncthreads();
  }
  if constexpr (MODE == 1) return;
  if constexpr (MODE == 3) {
    chain_wait();
    if (tid == 0) {
      const u64 total = chain_base + out;
      if (!chain_sent) __hip_atomic_store(hs.chain + (u64)cq * HUGE_RANGES + rr, total | (1ull << 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (rr == HUGE_RANGES - 1) { hs.gig_dist[cq] = (u32)total; group_distinct[tr_index(g, tr_a, tr_b)] = total; }
    }
    return;
  }
  if (!in_place) {
    __syncthreads();
    for (u64 i = tid; i < out; i += BLOCK) gk[i] = alt[a + i];
  }
  if (tid == 0) group_distinct[tr_index(g, tr_a, tr_b)] = out;
}

template <int BLOCK, int CAP, int SLOTS, bool WIDE, typename KT = K128, int MODE = 0>
__global__ __launch_bounds__(BLOCK)
void hash_count128_huge_kernel(KT *__restrict__ keys, const u64 *__restrict__ starts, const u32 *__restrict__ list, u64 ng,
                               u64 huge_min, u32 low_bits, u32 *__restrict__ cnt_tmp, u64 *__restrict__ group_distinct,
                               KT *__restrict__ alt, u32 tr_a = 0, u32 tr_b = 0, u64 huge_max = 0, HugeSliced hs = HugeSliced()) {
  if constexpr (MODE == 0 || MODE == 2) {
    hash_count128_huge_body<BLOCK, CAP, SLOTS, WIDE, KT, MODE>(blockIdx.x, keys, starts, list, ng, huge_min, low_bits, cnt_tmp, group_distinct, alt, tr_a, tr_b, huge_max, hs);
  } else {
    __shared__ u32 s_ticket;
    for (;;) {                                         // work by ticket from a small grid (hash_count_huge_kernel)
      __syncthreads();
      if (threadIdx.x == 0) s_ticket = atomicAdd(&hs.counters[MODE == 1 ? 3 : 2], 1u);
      __syncthreads();
      const u32 item = s_ticket;
      if (item >= (MODE == 1 ? hs.counters[1] : hs.counters[0] * HUGE_RANGES)) return;
      hash_count128_huge_body<BLOCK, CAP, SLOTS, WIDE, KT, MODE>(item, keys, starts, list, ng, huge_min, low_bits, cnt_tmp, group_distinct, alt, tr_a, tr_b, huge_max, hs);
    }
  }
}

// Would the distinct suffixes of every sub-bucket above huge_min fit the hash-count tables?  One workgroup per entry of the
// large-sub-bucket list streams its keys through a table that only stores suffixes (no counts, nothing written back) and
// raises *file_fail if more than CAP distinct ones turn up.  Run BEFORE the finish kernels touch the file, because the
// alternative for such a file (stable sort of all its bits) needs its k-mers intact.
template <typename S, int BLOCK, int CAP, int SLOTS, typename KT = u64>
__global__ __launch_bounds__(BLOCK)
void hash_probe_kernel(const KT *__restrict__ keys, const u64 *__restrict__ starts, const u32 *__restrict__ list,
                       u64 huge_min, u32 low_bits, u32 *__restrict__ file_fail) {
  constexpr int KPT = CAP / BLOCK;
  const S EMPTY = ~(S)0;
  __shared__ S   tk[SLOTS];
  __shared__ u32 s_st[2];
  const u32 tid = threadIdx.x;
  const u64 g = list[blockIdx.x];
  const u64 a = starts[g], n64 = starts[g + 1] - a;
  if (n64 <= huge_min) return;
  const u64 low_mask = (low_bits >= 64) ? ~0ull : ((1ull << low_bits) - 1ull);
  for (u32 i = tid; i < (u32)SLOTS; i += BLOCK) tk[i] = EMPTY;
  if (tid < 2) s_st[tid] = 0u;
  __syncthreads();
  constexpr u32 smask = SLOTS - 1, sshift = 32 - __builtin_ctz((unsigned)SLOTS);
  for (u64 base = 0; base < n64; base += (u64)CAP) {
    S   kk[KPT];
    u32 hh[KPT], pending = 0;
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const u64 idx = base + (u64)j * BLOCK + tid;
      kk[j] = (idx < n64) ? (S)((u64)keys[a + idx] & low_mask) : (S)0;
      hh[j] = (u32)(((u64)kk[j] * 0x9E3779B97F4A7C15ull) >> 32) >> sshift;
      if (idx < n64) pending |= 1u << j;
    }
    while (pending && !__hip_atomic_load(&s_st[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
#pragma unroll
      for (int j = 0; j < KPT; j++) {
        if ((pending >> j) & 1u) {
          const S old = atomicCAS(&tk[hh[j]], EMPTY, kk[j]);
          if (old == EMPTY || old == kk[j]) {
            pending &= ~(1u << j);
            if (old == EMPTY && atomicAdd(&s_st[0], 1u) >= (u32)CAP) s_st[1] = 1u;
          }
          else hh[j] = (hh[j] + 1) & smask;
        }
      }
    }
    if (__hip_atomic_load(&s_st[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
  }
  __syncthreads();
  if (tid == 0 && s_st[1]) atomicExch(file_fail, 1u);
  if (tid == 0) atomicMax(file_fail + 2, s_st[0]);     // diagnostics (MGC_FINISH_TRACE): most distinct suffixes met in a huge sub-bucket
}

// offs = exclusive scan of group_distinct (offs[ng] = total).  One wave per sub-bucket.
template <typename K>
__global__ __launch_bounds__(256)
void compact_groups_kernel(const K *__restrict__ keys, const u32 *__restrict__ cnt_tmp, const u64 *__restrict__ starts,
                           const u64 *__restrict__ offs, u64 ng, K *__restrict__ out_keys, u32 *__restrict__ out_counts, u32 tr_a, u32 tr_b,
                           const u32 *__restrict__ nz, u64 n_nz /* the non-empty sub-buckets only (a sparse grid: `compress`), or null */) {
  u64 g = (u64)blockIdx.x * 4 + wave_id();
  if (nz) { if (g >= n_nz) return; g = nz[g]; }
  else if (g >= ng) return;
  const u64 gt = tr_index(g, tr_a, tr_b);              // offs[] goes by the real sub-bucket number (whole keys: nothing to put back)
  const u64 dst = offs[gt], d = offs[gt + 1] - dst, src = starts[g];
  for (u64 i = lane_id(); i < d; i += 64) {
    out_keys[dst + i]   = keys[src + i];
    out_counts[dst + i] = cnt_tmp[src + i];
  }
}

// narrowed files: the finish left 32-bit suffixes; the k-mer is  base | sub-bucket << low_bits | suffix
// K96 records (the bits below the file) -> whole 16-byte k-mers: `base` = the file's bits, in their place (bits 2k-6 .. of the k-mer)
__global__ __launch_bounds__(256)
void compact_groups_k96_kernel(const K96 *__restrict__ keys, const u32 *__restrict__ cnt_tmp, const u64 *__restrict__ starts,
                               const u64 *__restrict__ offs, u64 ng, u64 base_lo, u64 base_hi, K128 *__restrict__ out_keys,
                               u32 *__restrict__ out_counts, u32 tr_a, u32 tr_b, const u32 *__restrict__ nz, u64 n_nz) {
  u64 g = (u64)blockIdx.x * 4 + wave_id();
  if (nz) { if (g >= n_nz) return; g = nz[g]; }
  else if (g >= ng) return;
  const u64 gt = tr_index(g, tr_a, tr_b);
  const u64 dst = offs[gt], d = offs[gt + 1] - dst, src = starts[g];
  for (u64 i = lane_id(); i < d; i += 64) {
    const K96 q = keys[src + i];
    K128 o; o.lo = KeyOps<K96>::low64(q) | base_lo; o.hi = (u64)q.w[2] | base_hi;
    out_keys[dst + i]   = o;
    out_counts[dst + i] = cnt_tmp[src + i];
  }
}
// a whole file of K96 records -> 16-byte keys (for the paths that want whole k-mers: streaming count, stable-sort fallback)
__global__ __launch_bounds__(256)
void widen_k96_kernel(const K96 *__restrict__ in, u64 n, u64 base_lo, u64 base_hi, K128 *__restrict__ out) {
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
    const K96 q = in[i];
    K128 o; o.lo = KeyOps<K96>::low64(q) | base_lo; o.hi = (u64)q.w[2] | base_hi;
    out[i] = o;
  }
}

__global__ __launch_bounds__(256)
void compact_groups_narrow_kernel(const u32 *__restrict__ keys32, const u32 *__restrict__ cnt_tmp, const u64 *__restrict__ starts,
                                  const u64 *__restrict__ offs, u64 ng, u64 base, u32 low_bits, u64 *__restrict__ out_keys,
                                  u32 *__restrict__ out_counts, u32 tr_a, u32 tr_b, const u32 *__restrict__ nz, u64 n_nz) {
  u64 g = (u64)blockIdx.x * 4 + wave_id();
  if (nz) { if (g >= n_nz) return; g = nz[g]; }
  else if (g >= ng) return;
  const u64 gt = tr_index(g, tr_a, tr_b);              // offs[] and the k-mers' top bits go by the real sub-bucket number
  const u64 dst = offs[gt], d = offs[gt + 1] - dst, src = starts[g], pre = base | (gt << low_bits);
  for (u64 i = lane_id(); i < d; i += 64) {
    out_keys[dst + i]   = pre | (u64)keys32[src + i];
    out_counts[dst + i] = cnt_tmp[src + i];
  }
}

// back to whole k-mers (a narrowed file that turns out to need a path that wants them: the LDS sort of an oversized
// sub-bucket, the stable-sort fallback)
__global__ __launch_bounds__(256)
void widen_groups_kernel(const u32 *__restrict__ keys32, const u64 *__restrict__ starts, u64 ng, u64 base, u32 low_bits,
                         u64 *__restrict__ out, u32 tr_a, u32 tr_b) {
  const u64 g = (u64)blockIdx.x * 4 + wave_id();
  if (g >= ng) return;
  const u64 a = starts[g], e = starts[g + 1], pre = base | (tr_index(g, tr_a, tr_b) << low_bits);
  const u32 low_mask = (low_bits >= 32) ? ~0u : ((1u << low_bits) - 1u);
  for (u64 i = a + lane_id(); i < e; i += 64) out[i] = pre | (u64)(keys32[i] & low_mask);
}

__global__ void store_u64_kernel(u64 *__restrict__ dst, const u64 *__restrict__ src) { *dst = *src; }

constexpr u64 FIN_CAP_SMALL = 256 * 16, FIN_CAP_LARGE = 1024 * 8;   // LDS: 46 KiB and 91 KiB per workgroup
// streamed sub-buckets: load factor <= 0.5 -- a probe step of one lane is a loop trip of its whole wave, and at 0.75 the
// longest chain among 64 lanes made an insert round 4x as long (scripts/gpu_huge.sh)
constexpr int HUGE_CAP32 = 4096, HUGE_SLOTS32 = 8192;            // 32-bit suffixes: 96 KiB of LDS
constexpr int HUGE_CAP64 = 2048, HUGE_SLOTS64 = 4096;            // 64-bit suffixes: 72 KiB
constexpr u64 FIN_CAP_HASH  = 1536;                               // hash-count kernel: 2048 slots, 26 KiB of LDS, 6 workgroups per CU
// hash_count_stream_kernel: sub-buckets of up to 4094 keys (a 12-bit count, all ones excluded) in chunks of <= 1536; 2048-entry table
// and room for 1280 distinct suffixes (17.6 KiB of LDS, eight workgroups per CU); the retry instantiation: 8192 entries, 58 KiB
constexpr u64 FIN_CAP_STREAM = 4094;
constexpr int FIN_STREAM_KPC = 6, FIN_STREAM_SLOTS = 2048, FIN_STREAM_DCAP = 1280, FIN_STREAM_RETRY_SLOTS = 8192, FIN_STREAM_RETRY_DCAP = 4096;
// ... on whole 8-byte k-mers (64-bit entries): 2048 entries and room for 1024 distinct suffixes: 28.3 KiB, five workgroups per CU
constexpr int FIN_STREAM64_DCAP = 1024;

// 16-byte keys: the same 1536-key tables (sub-buckets of up to 1152 k-mers); a file whose largest sub-bucket holds at most 768 takes
// the 768-key instantiation (launch_finish_file)
constexpr u64 FIN_CAP_HASH128 = 1536;

// which files the hash-count kernels take: 8-byte keys with suffixes of up to 58 bits (the packed 32-bit table below 32, the
// index-claimed table of hash_countw_kernel above), 16-byte keys with suffixes of up to 122 bits
static bool finish_uses_hash(uint32_t key_words, uint32_t low_bits) {
  if (key_words == 2) return low_bits <= 122;
  return key_words == 1 && low_bits <= 58;
}
// capacity of the first (small) launch of launch_finish_file; larger sub-buckets go on the list
static uint64_t finish_small_capacity(uint32_t key_words, uint32_t low_bits);

hipError_t launch_subbucket_bounds(const void *d_keys, uint64_t n, uint32_t key_words, uint32_t low, uint32_t top_bits,
                                   uint64_t *d_starts, uint64_t *d_max, uint32_t *d_list, uint64_t *d_list_count,
                                   uint32_t *d_nz, uint64_t *d_nz_count, hipStream_t st) {
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
  return launch_subbucket_max(d_starts, key_words, low, top_bits, d_max, d_list, d_list_count, d_nz, d_nz_count, st);
}

// the second half of launch_subbucket_bounds on its own: for files whose boundaries came with the grouping passes
hipError_t launch_subbucket_max(const uint64_t *d_starts, uint32_t key_words, uint32_t low, uint32_t top_bits, uint64_t *d_max,
                                uint32_t *d_list, uint64_t *d_list_count, uint32_t *d_nz, uint64_t *d_nz_count, hipStream_t st,
                                uint64_t small_cap) {
  const uint64_t ng = (uint64_t)1 << top_bits;
  hipLaunchKernelGGL(subbucket_max_kernel, dim3((uint32_t)((ng + 255) / 256)), dim3(256), 0, st,
                     reinterpret_cast<const u64 *>(d_starts), (u64)ng, reinterpret_cast<u64 *>(d_max),
                     (u64)(small_cap ? small_cap : finish_small_capacity(key_words, low)), d_list, reinterpret_cast<u64 *>(d_list_count),
                     d_nz, reinterpret_cast<u64 *>(d_nz_count));
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
static u64 *hash_dbg_buffer(bool on) {
  static u64 *buf = nullptr;
  if (!on) return nullptr;
  if (!buf) { if (hipMalloc(&buf, 64 * 8 * sizeof(u64)) != hipSuccess) buf = nullptr; }
  return buf;
}
static void hash_dbg_report(hipStream_t st, uint64_t ng, bool multi = false) {
  u64 *buf = hash_dbg_buffer(true);
  static int reports = 0;
  if (!buf || reports >= 4) return;
  u64 h[64 * 8];
  if (hipStreamSynchronize(st) != hipSuccess) return;
  if (hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return;
  double sum[8] = {0};
  for (int b = 0; b < 64; b++) for (int i = 0; i < 8; i++) sum[i] += (double)h[b * 8 + i];
  const double it = sum[7] > 0 ? sum[7] : 1;
  if (multi)     // hash_count_multi_kernel's stamps: [0] table sizing, [1] insert + list append, [6] next range consumed + loads issued, [2] bin counts, [3] scan + scatter, [4] rank + store
    fprintf(stderr, "[hash dbg multi] ng=%llu iters/block=%.1f cycles/iter: setup=%.0f insert=%.0f prefetch_advance=%.0f bin=%.0f scan+scatter=%.0f rank+store=%.0f\n",
            (unsigned long long)ng, it / 64, sum[0] / it, sum[1] / it, sum[6] / it, sum[2] / it, sum[3] / it, sum[4] / it);
  else
  fprintf(stderr, "[hash dbg] ng=%llu iters/block=%.1f cycles/iter: init=%.0f probe=%.0f compact=%.0f rank+store=%.0f sync=%.0f wait_next=%.0f\n",
          (unsigned long long)ng, it / 64, sum[0] / it, sum[1] / it, sum[2] / it, sum[3] / it, sum[4] / it, sum[6] / it);
  reports++;
}

// ---- the streaming count of a file's oversized sub-buckets: the plain launch, or -- the file holds a gigantic one and the caller
// brought a workspace -- plan + slices + merge, the range-parallel form for the dense ones, the single-workgroup form for the rest ----
size_t finish_huge_workspace_bytes(uint64_t n_keys) {
  const uint64_t max_gig = n_keys / HUGE_SLICE_MIN + 2, max_slices = n_keys / HUGE_SLICE + max_gig + 2;
  return 256 + sizeof(u32) * (4 * max_gig + 3 * max_slices) + 8 + sizeof(u64) * max_gig * HUGE_RANGES * 3;
}
// MODE 3 left a dense sub-bucket's k-mers in alt[] (its other ranges were still reading the keys): home, now that all of them are done
template <typename KT>
__global__ __launch_bounds__(1024)
void huge_copy_back_kernel(KT *__restrict__ keys, const KT *__restrict__ alt, const u64 *__restrict__ starts, HugeSliced hs) {
  const u32 q = blockIdx.x;
  if (q >= hs.counters[0] || q >= hs.max_gig || !hs.gig_fail[q]) return;
  const u64 a = starts[hs.gig_g[q]];
  const u32 d = hs.gig_dist[q];
  for (u32 i = threadIdx.x; i < d; i += 1024) keys[a + i] = alt[a + i];
}
template <typename S, int CAP, int SLOTS, typename KT>
static hipError_t launch_huge(KT *keys, const u64 *starts, const u32 *list, u64 n_large, u64 ng, u64 huge_min, u32 low_bits, u32 *cnt_tmp,
                              u64 *group_distinct, KT *alt, u32 tr_a, u32 tr_b, size_t smem, hipStream_t st, u64 max_sub, u64 n_keys,
                              void *ws, size_t ws_bytes, u64 ws_keys, u32 *d_error) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&hash_count_huge_kernel<S, 1024, CAP, SLOTS, KT, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&hash_count_huge_kernel<S, 1024, CAP, SLOTS, KT, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&hash_count_huge_kernel<S, 1024, CAP, SLOTS, KT, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&hash_count_huge_kernel<S, 1024, CAP, SLOTS, KT, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr = true;
  }
  if (n_large == 0) return hipSuccess;
  // (the workspace is laid out for ws_keys keys, the largest file of the count: the same layout for every file)
  const bool sliced = ws && d_error && max_sub > (u64)HUGE_SLICE_MIN && n_keys && n_keys <= ws_keys && ws_bytes >= finish_huge_workspace_bytes(ws_keys);
  if (!sliced) {
    hipLaunchKernelGGL((hash_count_huge_kernel<S, 1024, CAP, SLOTS, KT, 0>), dim3((uint32_t)n_large), dim3(1024), smem, st, keys, starts, list, ng, huge_min,
                       low_bits, cnt_tmp, group_distinct, alt, tr_a, tr_b, (u64)0, HugeSliced());
    return hipGetLastError();
  }
  HugeSliced hs;
  hs.max_gig = (u32)(ws_keys / HUGE_SLICE_MIN + 2);
  hs.max_slices = (u32)(ws_keys / HUGE_SLICE + hs.max_gig + 2);
  u32 *w = reinterpret_cast<u32 *>(ws);
  hs.counters = w;
  hs.gig_g = w + 64; hs.gig_pairs = hs.gig_g + hs.max_gig; hs.gig_fail = hs.gig_pairs + hs.max_gig; hs.gig_dist = hs.gig_fail + hs.max_gig;
  hs.slice_g = hs.gig_dist + hs.max_gig; hs.slice_j = hs.slice_g + hs.max_slices; hs.slice_q = hs.slice_j + hs.max_slices;
  hs.chain = reinterpret_cast<u64 *>((reinterpret_cast<uintptr_t>(hs.slice_q + hs.max_slices) + 7) & ~(uintptr_t)7);
  hs.split = hs.chain + (size_t)hs.max_gig * HUGE_RANGES;
  hs.error = d_error;
  MGC_CHECK(hipMemsetAsync(ws, 0, finish_huge_workspace_bytes(ws_keys), st));
  hipLaunchKernelGGL(huge_plan_kernel, dim3((uint32_t)((n_large + 255) / 256)), dim3(256), 0, st, starts, list, n_large, (u64)HUGE_SLICE_MIN, hs, 0u);
  MGC_CHECK(hipGetLastError());
  const uint32_t sgrid = (uint32_t)std::min<u64>(512, std::min<u64>((u64)hs.max_slices, n_keys / HUGE_SLICE + n_keys / HUGE_SLICE_MIN + 4));   // (by ticket)
  const uint32_t ggrid = (uint32_t)std::min<u64>(std::min<u64>(n_large, (u64)hs.max_gig), n_keys / HUGE_SLICE_MIN + 1);
  // the slices first: they are the long pole
  hipLaunchKernelGGL((hash_count_huge_kernel<S, 1024, CAP, SLOTS, KT, 1>), dim3(sgrid), dim3(1024), smem, st, keys, starts, list, ng, huge_min,
                     low_bits, cnt_tmp, group_distinct, alt, tr_a, tr_b, (u64)0, hs);
  MGC_CHECK(hipGetLastError());
  hipLaunchKernelGGL((hash_count_huge_kernel<S, 1024, CAP, SLOTS, KT, 0>), dim3((uint32_t)n_large), dim3(1024), smem, st, keys, starts, list, ng, huge_min,
                     low_bits, cnt_tmp, group_distinct, alt, tr_a, tr_b, (u64)HUGE_SLICE_MIN, hs);
  MGC_CHECK(hipGetLastError());
  hipLaunchKernelGGL((hash_count_huge_kernel<S, 1024, CAP, SLOTS, KT, 2>), dim3(ggrid), dim3(1024), smem, st, keys, starts, list, ng, huge_min,
                     low_bits, cnt_tmp, group_distinct, alt, tr_a, tr_b, (u64)0, hs);
  MGC_CHECK(hipGetLastError());
  // ... and the dense ones (their slices' pairs did not fit): quantiles of a sample, HUGE_RANGES workgroups each, by ticket; then their k-mers home from alt[]
  hipLaunchKernelGGL(huge_split_kernel<KT>, dim3(ggrid), dim3(1024), 0, st, (const KT *)keys, starts, low_bits, hs);
  MGC_CHECK(hipGetLastError());
  hipLaunchKernelGGL((hash_count_huge_kernel<S, 1024, CAP, SLOTS, KT, 3>), dim3(256), dim3(1024), smem, st, keys, starts, list, ng, huge_min,
                     low_bits, cnt_tmp, group_distinct, alt, tr_a, tr_b, (u64)0, hs);
  MGC_CHECK(hipGetLastError());
  hipLaunchKernelGGL(huge_copy_back_kernel<KT>, dim3(ggrid), dim3(1024), 0, st, keys, (const KT *)alt, starts, hs);
  return hipGetLastError();
}

// the same for 16-byte keys / K96 records (a pair = two records: the suffix, the count)
template <bool WIDE, typename KT>
static hipError_t launch_huge128(KT *keys, const u64 *starts, const u32 *list, u64 n_large, u64 ng, u64 huge_min, u32 low_bits, u32 *cnt_tmp,
                                 u64 *group_distinct, KT *alt, u32 tr_a, u32 tr_b, hipStream_t st, u64 max_sub, u64 n_keys,
                                 void *ws, size_t ws_bytes, u64 ws_keys, u32 *d_error) {
  constexpr int HS = 4096, HC = 2048;
  constexpr size_t smem = WIDE ? (size_t)(8 + 8 + 4) * HS + (size_t)(8 + 8) * (HC + 16) + (size_t)4 * HC
                               : (size_t)(8 + 4) * HS + (size_t)8 * (HC + 16) + (size_t)4 * HC;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&hash_count128_huge_kernel<1024, HC, HS, WIDE, KT, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&hash_count128_huge_kernel<1024, HC, HS, WIDE, KT, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&hash_count128_huge_kernel<1024, HC, HS, WIDE, KT, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&hash_count128_huge_kernel<1024, HC, HS, WIDE, KT, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr = true;
  }
  if (n_large == 0) return hipSuccess;
  const bool sliced = ws && d_error && max_sub > (u64)HUGE_SLICE_MIN && n_keys && n_keys <= ws_keys && ws_bytes >= finish_huge_workspace_bytes(ws_keys);
  if (!sliced) {
    hipLaunchKernelGGL((hash_count128_huge_kernel<1024, HC, HS, WIDE, KT, 0>), dim3((uint32_t)n_large), dim3(1024), smem, st, keys, starts, list, ng, huge_min,
                       low_bits, cnt_tmp, group_distinct, alt, tr_a, tr_b, (u64)0, HugeSliced());
    return hipGetLastError();
  }
  HugeSliced hs;
  hs.max_gig = (u32)(ws_keys / HUGE_SLICE_MIN + 2);
  hs.max_slices = (u32)(ws_keys / HUGE_SLICE + hs.max_gig + 2);
  u32 *w = reinterpret_cast<u32 *>(ws);
  hs.counters = w;
  hs.gig_g = w + 64; hs.gig_pairs = hs.gig_g + hs.max_gig; hs.gig_fail = hs.gig_pairs + hs.max_gig; hs.gig_dist = hs.gig_fail + hs.max_gig;
  hs.slice_g = hs.gig_dist + hs.max_gig; hs.slice_j = hs.slice_g + hs.max_slices; hs.slice_q = hs.slice_j + hs.max_slices;
  hs.chain = reinterpret_cast<u64 *>((reinterpret_cast<uintptr_t>(hs.slice_q + hs.max_slices) + 7) & ~(uintptr_t)7);
  hs.split = hs.chain + (size_t)hs.max_gig * HUGE_RANGES;
  hs.error = d_error;
  MGC_CHECK(hipMemsetAsync(ws, 0, finish_huge_workspace_bytes(ws_keys), st));
  hipLaunchKernelGGL(huge_plan_kernel, dim3((uint32_t)((n_large + 255) / 256)), dim3(256), 0, st, starts, list, n_large, (u64)HUGE_SLICE_MIN, hs, 0u);
  MGC_CHECK(hipGetLastError());
  const uint32_t sgrid = (uint32_t)std::min<u64>(512, std::min<u64>((u64)hs.max_slices, n_keys / HUGE_SLICE + n_keys / HUGE_SLICE_MIN + 4));   // (by ticket)
  hipLaunchKernelGGL((hash_count128_huge_kernel<1024, HC, HS, WIDE, KT, 1>), dim3(sgrid), dim3(1024), smem, st, keys, starts, list, ng, huge_min,
                     low_bits, cnt_tmp, group_distinct, alt, tr_a, tr_b, (u64)0, hs);
  MGC_CHECK(hipGetLastError());
  const uint32_t ggrid = (uint32_t)std::min<u64>(std::min<u64>(n_large, (u64)hs.max_gig), n_keys / HUGE_SLICE_MIN + 1);
  hipLaunchKernelGGL((hash_count128_huge_kernel<1024, HC, HS, WIDE, KT, 2>), dim3(ggrid), dim3(1024), smem, st, keys, starts, list, ng, huge_min,
                     low_bits, cnt_tmp, group_distinct, alt, tr_a, tr_b, (u64)0, hs);
  MGC_CHECK(hipGetLastError());
  hipLaunchKernelGGL(huge_split_kernel<KT>, dim3(ggrid), dim3(1024), 0, st, (const KT *)keys, starts, low_bits, hs);
  MGC_CHECK(hipGetLastError());
  hipLaunchKernelGGL((hash_count128_huge_kernel<1024, HC, HS, WIDE, KT, 3>), dim3(256), dim3(1024), smem, st, keys, starts, list, ng, huge_min,
                     low_bits, cnt_tmp, group_distinct, alt, tr_a, tr_b, (u64)0, hs);
  MGC_CHECK(hipGetLastError());
  hipLaunchKernelGGL((hash_count128_huge_kernel<1024, HC, HS, WIDE, KT, 0>), dim3((uint32_t)n_large), dim3(1024), smem, st, keys, starts, list, ng, huge_min,
                     low_bits, cnt_tmp, group_distinct, alt, tr_a, tr_b, (u64)HUGE_SLICE_MIN, hs);
  MGC_CHECK(hipGetLastError());
  hipLaunchKernelGGL(huge_copy_back_kernel<KT>, dim3(ggrid), dim3(1024), 0, st, keys, (const KT *)alt, starts, hs);
  return hipGetLastError();
}

bool finish_can_stream(uint32_t key_words, uint32_t low_bits) {
  return finish_uses_hash(key_words, low_bits);
}

hipError_t launch_finish_probe(const void *d_keys, uint32_t key_words, const uint64_t *d_starts, uint32_t low_bits,
                               uint64_t n_large, const uint32_t *d_large_list, uint32_t *d_file_fail, hipStream_t st, uint64_t stream_max, bool narrow) {
  if (n_large == 0 || key_words != 1) return hipSuccess;
  if (narrow) {
    if (low_bits >= 32) return hipErrorInvalidValue;
    hipLaunchKernelGGL((hash_probe_kernel<u32, 1024, HUGE_CAP32, HUGE_SLOTS32, u32>), dim3((uint32_t)n_large), dim3(1024), 0, st,
                       reinterpret_cast<const u32 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), d_large_list,
                       (u64)stream_max, low_bits, d_file_fail);
    return hipGetLastError();
  }
  if (low_bits < 32)
    hipLaunchKernelGGL((hash_probe_kernel<u32, 1024, HUGE_CAP32, HUGE_SLOTS32>), dim3((uint32_t)n_large), dim3(1024), 0, st,
                       reinterpret_cast<const u64 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), d_large_list,
                       (u64)stream_max, low_bits, d_file_fail);
  else
    hipLaunchKernelGGL((hash_probe_kernel<u64, 1024, HUGE_CAP64, HUGE_SLOTS64>), dim3((uint32_t)n_large), dim3(1024), 0, st,
                       reinterpret_cast<const u64 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), d_large_list,
                       (u64)stream_max, low_bits, d_file_fail);
  return hipGetLastError();
}

// stream: the sub-buckets on the large list go through hash_count_huge_kernel (d_alt: room for the file's keys); otherwise
// through the LDS sort, and the file holds none above its capacity (the caller checked the largest sub-bucket)
hipError_t launch_finish_file(void *d_keys, uint32_t key_words, const uint64_t *d_starts, uint64_t ng, uint32_t low_bits,
                              uint64_t n_large, const uint32_t *d_large_list, uint32_t *d_cnt_tmp, uint64_t *d_group_distinct,
                              bool stream, void *d_alt, hipStream_t st_huge, const uint32_t *d_nz, const uint64_t *d_nz_count,
                              hipStream_t st, bool narrow, uint32_t tr_a, uint32_t tr_b, uint64_t max_sub, uint64_t n_keys,
                              uint32_t *d_retry_list, uint64_t *d_retry_count, bool k96, int hash_multi, bool hash_dbg,
                              uint64_t stream_cap, void *d_huge_ws, size_t huge_ws_bytes, uint64_t huge_ws_keys, uint32_t *d_error) {
  const u64 *nzc = reinterpret_cast<const u64 *>(d_nz_count);
  if (stream_cap && !(d_retry_list && d_retry_count && finish_stream_ok(key_words, low_bits, narrow) && !k96 && n_keys < (1ull << 32) &&
                      stream_cap <= FIN_CAP_STREAM && (n_large == 0 || stream)))
    return hipErrorInvalidValue;
  if (k96) {
    // 12-byte K96 records: the persistent hash-count, and the streaming count of oversized sub-buckets (a file whose oversized
    // sub-buckets nothing streams is widened to 16-byte keys by the caller first)
    if (key_words != 2 || narrow || (n_large && !stream) || !finish_uses_hash(key_words, low_bits)) return hipErrorInvalidValue;
    const bool small96 = max_sub && max_sub <= 768 && n_large == 0;
    const u64 msize = small96 ? (u64)768 : FIN_CAP_HASH128;
    const bool lst = d_nz != nullptr, wide = low_bits > 64;
    const uint32_t gmax = 256u * (small96 ? 12u : 8u), g96 = ng < gmax ? (uint32_t)ng : gmax;
#define MGC_W96_LAUNCH(CAP_, SLOTS_, WIDE_, LIST_)                                                                                       \
    hipLaunchKernelGGL((hash_countw_kernel<K96, 256, CAP_, SLOTS_, WIDE_, LIST_>), dim3(g96), dim3(256), 0, st,                          \
                       reinterpret_cast<K96 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), (u64)ng, msize, low_bits,               \
                       d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct), d_nz, nzc, tr_a, tr_b)
    if (small96) {
      if (wide) { if (lst) MGC_W96_LAUNCH(768, 1024, true, true);  else MGC_W96_LAUNCH(768, 1024, true, false); }
      else      { if (lst) MGC_W96_LAUNCH(768, 1024, false, true); else MGC_W96_LAUNCH(768, 1024, false, false); }
    } else {
      if (wide) { if (lst) MGC_W96_LAUNCH(1536, 2048, true, true);  else MGC_W96_LAUNCH(1536, 2048, true, false); }
      else      { if (lst) MGC_W96_LAUNCH(1536, 2048, false, true); else MGC_W96_LAUNCH(1536, 2048, false, false); }
    }
#undef MGC_W96_LAUNCH
    MGC_CHECK(hipGetLastError());
    if (n_large) {
      if (wide)
        MGC_CHECK((launch_huge128<true, K96>(reinterpret_cast<K96 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), d_large_list, (u64)n_large, (u64)ng,
                   FIN_CAP_HASH128, low_bits, d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct), reinterpret_cast<K96 *>(d_alt), tr_a, tr_b, st_huge,
                   (u64)max_sub, (u64)n_keys, d_huge_ws, huge_ws_bytes, (u64)huge_ws_keys, d_error)));
      else
        MGC_CHECK((launch_huge128<false, K96>(reinterpret_cast<K96 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), d_large_list, (u64)n_large, (u64)ng,
                   FIN_CAP_HASH128, low_bits, d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct), reinterpret_cast<K96 *>(d_alt), tr_a, tr_b, st_huge,
                   (u64)max_sub, (u64)n_keys, d_huge_ws, huge_ws_bytes, (u64)huge_ws_keys, d_error)));
    }
    return hipSuccess;
  }
  const bool use_list = d_nz != nullptr;
  if (narrow) {
    // narrowed keys (u32): the 32-bit hash-count kernels and the streaming kernel have u32-storage instantiations; anything
    // else wants whole k-mers -- the caller widens the file first (launch_widen_groups)
    if (key_words != 1 || low_bits >= 32 || (n_large && !stream)) return hipErrorInvalidValue;
    const uint32_t hgrid = ng < 256u * 14u ? (uint32_t)ng : 256u * 14u;
#define MGC_NARROW_LAUNCH(DBG_, LIST_, DBGBUF_)                                                                                          \
    hipLaunchKernelGGL((hash_count_kernel<256, (int)FIN_CAP_HASH, 2048, DBG_, LIST_, true>), dim3(hgrid), dim3(256), 0, st,              \
                       reinterpret_cast<u64 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), (u64)ng, (u64)FIN_CAP_HASH, low_bits, \
                       d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct), d_nz, nzc, DBGBUF_, tr_a, tr_b)
    // R physically consecutive sub-buckets per iteration with packed key|count entries (hash_count_multi_kernel): dense grids
    // whose tagged suffix fits 20 bits.  R from the file's average sub-bucket (MGC_HASH_MULTI=0: the one-at-a-time kernel;
    // 1/2/4: that R whatever the average -- tests).  Measured at 10 Gbp (profiles/r05a: MGC_HASH_MULTI=2 on every file): R = 2
    // on files whose sub-buckets average more than 690 k-mers sends most ranges to the retry list (they exceed the 1536-key
    // table): 68 ms of count stage against 29 -- R = 1 there is the table's size, not the heuristic's choice.
    const int multi_env = hash_multi;
    int multi_r = 0;
    if (multi_env != 0 && !d_nz && ng >= 4 && d_retry_list && d_retry_count) {
      const uint64_t avg = n_keys ? n_keys / ng : FIN_CAP_HASH;
      multi_r = multi_env > 0 ? multi_env : (avg <= 340 ? 4 : (avg <= 690 ? 2 : 1));
      const uint32_t tagb = multi_r == 1 ? 0u : (multi_r == 2 ? 1u : 2u);
      if (multi_env < 0 && multi_r > 1 && low_bits + tagb > 20) multi_r = (multi_r == 4 && low_bits + 1 <= 20) ? 2 : 1;   // fewer tag bits
      const uint32_t tagb2 = multi_r == 1 ? 0u : (multi_r == 2 ? 1u : 2u);
      if (multi_r > 4 || multi_r == 3 || low_bits + tagb2 < 8 || low_bits + tagb2 > 20) multi_r = 0;
    }
    u64 *dbgb = hash_dbg_buffer(hash_dbg);                           // MGC_HASH_DBG=1: the instrumented instantiations (per-phase cycle stamps)
    if (stream_cap) {
      // the distinct-sized count: one sub-bucket of up to stream_cap keys per iteration, streamed through a 2048-entry table
      // in chunks; sub-buckets with more distinct suffixes than it holds land on the retry list (launch_finish_retry)
      const uint64_t gmax = 256ull * 16ull;
      const dim3 sgrid((uint32_t)(ng < gmax ? ng : gmax));
      if (use_list)
        hipLaunchKernelGGL((hash_count_stream_kernel<u32, 256, FIN_STREAM_KPC, FIN_STREAM_SLOTS, FIN_STREAM_DCAP, true, false>), sgrid, dim3(256), 0, st,
                           reinterpret_cast<u32 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), (u64)ng, (u64)stream_cap, low_bits,
                           d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct), (u64 *)nullptr, tr_a, tr_b, d_nz, nzc,
                           d_retry_list, reinterpret_cast<u64 *>(d_retry_count));
      else if (dbgb)
        hipLaunchKernelGGL((hash_count_stream_kernel<u32, 256, FIN_STREAM_KPC, FIN_STREAM_SLOTS, FIN_STREAM_DCAP, false, true>), sgrid, dim3(256), 0, st,
                           reinterpret_cast<u32 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), (u64)ng, (u64)stream_cap, low_bits,
                           d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct), dbgb, tr_a, tr_b, (const u32 *)nullptr, (const u64 *)nullptr,
                           d_retry_list, reinterpret_cast<u64 *>(d_retry_count));
      else
        hipLaunchKernelGGL((hash_count_stream_kernel<u32, 256, FIN_STREAM_KPC, FIN_STREAM_SLOTS, FIN_STREAM_DCAP, false, false>), sgrid, dim3(256), 0, st,
                           reinterpret_cast<u32 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), (u64)ng, (u64)stream_cap, low_bits,
                           d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct), (u64 *)nullptr, tr_a, tr_b, (const u32 *)nullptr,
                           (const u64 *)nullptr, d_retry_list, reinterpret_cast<u64 *>(d_retry_count));
      MGC_CHECK(hipGetLastError());
      if (dbgb && !use_list) hash_dbg_report(st, ng, true);
    }
    else if (multi_r) {
      const uint64_t nsuper = (ng + (uint64_t)multi_r - 1) / (uint64_t)multi_r;
#define MGC_MULTI_LAUNCH(R_, DBG_)                                                                                                       \
      do { const uint64_t gmax = 256ull * 16ull;                                                                                        \
           hipLaunchKernelGGL((hash_count_multi_kernel<256, (int)FIN_CAP_HASH, 2048, R_, DBG_>), dim3((uint32_t)(nsuper < gmax ? nsuper : gmax)), \
                       dim3(256), 0, st, reinterpret_cast<u32 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), (u64)ng,              \
                       (u64)FIN_CAP_HASH, low_bits, d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct), dbgb, tr_a, tr_b,             \
                       d_retry_list, reinterpret_cast<u64 *>(d_retry_count)); } while (0)
      if (dbgb && multi_r == 2) MGC_MULTI_LAUNCH(2, true);
      else if (multi_r == 1)    MGC_MULTI_LAUNCH(1, false);
      else if (multi_r == 2)    MGC_MULTI_LAUNCH(2, false);
      else                      MGC_MULTI_LAUNCH(4, false);
#undef MGC_MULTI_LAUNCH
      MGC_CHECK(hipGetLastError());
      if (dbgb && multi_r == 2) hash_dbg_report(st, ng, true);
      // the sub-buckets of ranges above the table (retry list, usually short): one at a time.  (R = 1: a range is one sub-bucket,
      // the list stays empty -- no launch: queued behind the other stream's persistent kernel an empty one still lasted 170 us)
      if (multi_r > 1) { const uint32_t rgrid = ng < 256u * 7u ? (uint32_t)ng : 256u * 7u;
        hipLaunchKernelGGL((hash_count_kernel<256, (int)FIN_CAP_HASH, 2048, false, true, true>), dim3(rgrid), dim3(256), 0, st,
                           reinterpret_cast<u64 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), (u64)ng, (u64)FIN_CAP_HASH, low_bits,
                           d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct), d_retry_list, reinterpret_cast<const u64 *>(d_retry_count),
                           (u64 *)nullptr, tr_a, tr_b);
        MGC_CHECK(hipGetLastError()); }
    }
    else if (dbgb) {
      if (use_list) MGC_NARROW_LAUNCH(true, true, dbgb); else MGC_NARROW_LAUNCH(true, false, dbgb);
      MGC_CHECK(hipGetLastError());
      hash_dbg_report(st, ng);
    }
    else if (use_list) MGC_NARROW_LAUNCH(false, true, (u64 *)nullptr);
    else               MGC_NARROW_LAUNCH(false, false, (u64 *)nullptr);
#undef MGC_NARROW_LAUNCH
    MGC_CHECK(hipGetLastError());
    if (n_large) {
      constexpr size_t B32 = (size_t)(4 + 4) * HUGE_SLOTS32 + (size_t)(4 + 4) * HUGE_CAP32 + 16 * 4;
      MGC_CHECK((launch_huge<u32, HUGE_CAP32, HUGE_SLOTS32, u32>(reinterpret_cast<u32 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), d_large_list, (u64)n_large,
                 (u64)ng, (u64)(stream_cap ? stream_cap : FIN_CAP_HASH), low_bits, d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct),
                 reinterpret_cast<u32 *>(d_alt), tr_a, tr_b, B32, st_huge, (u64)max_sub, (u64)n_keys, d_huge_ws, huge_ws_bytes, (u64)huge_ws_keys, d_error)));
    }
    return hipSuccess;
  }
  // whole keys in (low digit : high digit) order (launch_group_wide): only the hash-count kernels translate the sub-bucket numbers
  if (tr_a && !(finish_uses_hash(key_words, low_bits) && (n_large == 0 || stream))) return hipErrorInvalidValue;
  // a file whose LARGEST sub-bucket holds at most 768 k-mers (`compress`: 59049 sub-buckets per bucket, a few hundred k-mers each):
  // three keys per thread instead of six -- fewer idle unrolled slots, half the LDS, more workgroups per CU
  const bool small = max_sub && max_sub <= 768 && n_large == 0;
  if (key_words == 2 && finish_uses_hash(key_words, low_bits)) {
    const u64 msize = small ? (u64)768 : FIN_CAP_HASH128;
#define MGC_W128_LAUNCH(CAP_, SLOTS_, WIDE_, LIST_, GRID_)                                                                               \
    hipLaunchKernelGGL((hash_countw_kernel<K128, 256, CAP_, SLOTS_, WIDE_, LIST_>), dim3(GRID_), dim3(256), 0, st,                       \
                       reinterpret_cast<K128 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), (u64)ng, msize, low_bits,              \
                       d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct), d_nz, nzc, tr_a, tr_b)
    const uint32_t g128_max = 256u * (small ? 12u : 8u);
    const uint32_t g128 = ng < g128_max ? (uint32_t)ng : g128_max;
    const bool wide128 = low_bits > 64;
    if (small) {
      if (wide128) { if (use_list) MGC_W128_LAUNCH(768, 1024, true, true, g128);  else MGC_W128_LAUNCH(768, 1024, true, false, g128); }
      else         { if (use_list) MGC_W128_LAUNCH(768, 1024, false, true, g128); else MGC_W128_LAUNCH(768, 1024, false, false, g128); }
    } else {
      if (wide128) { if (use_list) MGC_W128_LAUNCH(1536, 2048, true, true, g128);  else MGC_W128_LAUNCH(1536, 2048, true, false, g128); }
      else         { if (use_list) MGC_W128_LAUNCH(1536, 2048, false, true, g128); else MGC_W128_LAUNCH(1536, 2048, false, false, g128); }
    }
#undef MGC_W128_LAUNCH
    MGC_CHECK(hipGetLastError());
    if (stream && n_large) {
      if (low_bits > 64)
        MGC_CHECK((launch_huge128<true, K128>(reinterpret_cast<K128 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), d_large_list, (u64)n_large, (u64)ng,
                   FIN_CAP_HASH128, low_bits, d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct), reinterpret_cast<K128 *>(d_alt), tr_a, tr_b, st_huge,
                   (u64)max_sub, (u64)n_keys, d_huge_ws, huge_ws_bytes, (u64)huge_ws_keys, d_error)));
      else
        MGC_CHECK((launch_huge128<false, K128>(reinterpret_cast<K128 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), d_large_list, (u64)n_large, (u64)ng,
                   FIN_CAP_HASH128, low_bits, d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct), reinterpret_cast<K128 *>(d_alt), tr_a, tr_b, st_huge,
                   (u64)max_sub, (u64)n_keys, d_huge_ws, huge_ws_bytes, (u64)huge_ws_keys, d_error)));
    } else {
      MGC_CHECK((finish_launch<K128, 1024, 8>(d_keys, d_starts, n_large, low_bits, FIN_CAP_HASH128, 8192, d_cnt_tmp, d_group_distinct, st, d_large_list)));
    }
    return hipSuccess;
  }
  if (key_words == 2) {
    // 16-byte keys: 256x8 (2048) and 1024x8 (8192) keep LDS at 32 / 128 KiB
    MGC_CHECK((finish_launch<K128, 256, 8>(d_keys, d_starts, ng, low_bits, 0, 2048, d_cnt_tmp, d_group_distinct, st)));
    MGC_CHECK((finish_launch<K128, 1024, 8>(d_keys, d_starts, n_large, low_bits, 2048, 8192, d_cnt_tmp, d_group_distinct, st, d_large_list)));
    return hipSuccess;
  }
  if (finish_uses_hash(key_words, low_bits)) {
    // <= FIN_CAP_HASH keys: hash-count; larger sub-buckets: streamed, or LDS radix passes in the 8192-key instantiation
    if (stream_cap) {
      // the distinct-sized count on whole 8-byte k-mers (k = 24..32, `compress`): 64-bit entries suffix << 12 | count, the bits above the
      // suffix put back on the way out; sub-buckets with more distinct suffixes than the table holds land on the retry list
      const uint64_t gmax = 256ull * 10ull;
      const dim3 sgrid((uint32_t)(ng < gmax ? ng : gmax));
      if (use_list)
        hipLaunchKernelGGL((hash_count_stream_kernel<u64, 256, FIN_STREAM_KPC, FIN_STREAM_SLOTS, FIN_STREAM64_DCAP, true, false>), sgrid, dim3(256), 0, st,
                           reinterpret_cast<u64 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), (u64)ng, (u64)stream_cap, low_bits,
                           d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct), (u64 *)nullptr, tr_a, tr_b, d_nz, nzc,
                           d_retry_list, reinterpret_cast<u64 *>(d_retry_count));
      else
        hipLaunchKernelGGL((hash_count_stream_kernel<u64, 256, FIN_STREAM_KPC, FIN_STREAM_SLOTS, FIN_STREAM64_DCAP, false, false>), sgrid, dim3(256), 0, st,
                           reinterpret_cast<u64 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), (u64)ng, (u64)stream_cap, low_bits,
                           d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct), (u64 *)nullptr, tr_a, tr_b, (const u32 *)nullptr,
                           (const u64 *)nullptr, d_retry_list, reinterpret_cast<u64 *>(d_retry_count));
    } else if (low_bits >= 32) {
#define MGC_W64_LAUNCH(CAP_, SLOTS_, LIST_, GRID_, MS_)                                                                                  \
      hipLaunchKernelGGL((hash_countw_kernel<u64, 256, CAP_, SLOTS_, false, LIST_>), dim3(GRID_), dim3(256), 0, st, reinterpret_cast<u64 *>(d_keys), \
                         reinterpret_cast<const u64 *>(d_starts), (u64)ng, (u64)(MS_), low_bits, d_cnt_tmp,                             \
                         reinterpret_cast<u64 *>(d_group_distinct), d_nz, nzc, tr_a, tr_b)
      if (small) {
        const uint32_t sgrid = ng < 256u * 12u ? (uint32_t)ng : 256u * 12u;
        if (use_list) MGC_W64_LAUNCH(768, 1024, true, sgrid, 768); else MGC_W64_LAUNCH(768, 1024, false, sgrid, 768);
      } else {
        const uint32_t mgrid = ng < 256u * 10u ? (uint32_t)ng : 256u * 10u;
        if (use_list) MGC_W64_LAUNCH((int)FIN_CAP_HASH, 2048, true, mgrid, FIN_CAP_HASH); else MGC_W64_LAUNCH((int)FIN_CAP_HASH, 2048, false, mgrid, FIN_CAP_HASH);
      }
#undef MGC_W64_LAUNCH
    } else {
      const uint32_t hgrid = ng < 256u * 14u ? (uint32_t)ng : 256u * 14u;
      u64 *dbgb = hash_dbg_buffer(hash_dbg);
#define MGC_HASH_LAUNCH(DBG_, LIST_)                                                                                                     \
      hipLaunchKernelGGL((hash_count_kernel<256, (int)FIN_CAP_HASH, 2048, DBG_, LIST_, false>), dim3(hgrid), dim3(256), 0, st,           \
                         reinterpret_cast<u64 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), (u64)ng, (u64)FIN_CAP_HASH, low_bits, \
                         d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct), d_nz, nzc, dbgb, tr_a, tr_b)
      if (dbgb)          { if (use_list) MGC_HASH_LAUNCH(true, true); else MGC_HASH_LAUNCH(true, false); }
      else if (use_list) MGC_HASH_LAUNCH(false, true);
      else               MGC_HASH_LAUNCH(false, false);
#undef MGC_HASH_LAUNCH
      MGC_CHECK(hipGetLastError());
      if (dbgb) hash_dbg_report(st, ng);
    }
    MGC_CHECK(hipGetLastError());
    if (stream && n_large) {
      // sub-buckets above the small tables: one 1024-thread workgroup each, keys streamed through a large table
      constexpr size_t B32 = (size_t)(4 + 4) * HUGE_SLOTS32 + (size_t)(4 + 4) * HUGE_CAP32 + 16 * 4;
      constexpr size_t B64 = (size_t)(8 + 4) * HUGE_SLOTS64 + (size_t)(8 + 4) * HUGE_CAP64 + 16 * 8;
      if (low_bits < 32)
        MGC_CHECK((launch_huge<u32, HUGE_CAP32, HUGE_SLOTS32, u64>(reinterpret_cast<u64 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), d_large_list, (u64)n_large,
                   (u64)ng, (u64)(stream_cap ? stream_cap : FIN_CAP_HASH), low_bits, d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct),
                   reinterpret_cast<u64 *>(d_alt), tr_a, tr_b, B32, st_huge, (u64)max_sub, (u64)n_keys, d_huge_ws, huge_ws_bytes, (u64)huge_ws_keys, d_error)));
      else
        MGC_CHECK((launch_huge<u64, HUGE_CAP64, HUGE_SLOTS64, u64>(reinterpret_cast<u64 *>(d_keys), reinterpret_cast<const u64 *>(d_starts), d_large_list, (u64)n_large,
                   (u64)ng, (u64)(stream_cap ? stream_cap : FIN_CAP_HASH), low_bits, d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct),
                   reinterpret_cast<u64 *>(d_alt), tr_a, tr_b, B64, st_huge, (u64)max_sub, (u64)n_keys, d_huge_ws, huge_ws_bytes, (u64)huge_ws_keys, d_error)));
    } else {
      MGC_CHECK((finish_launch<u64, 1024, 8>(d_keys, d_starts, n_large, low_bits, FIN_CAP_HASH, FIN_CAP_LARGE, d_cnt_tmp,
                                             d_group_distinct, st, d_large_list)));
    }
    return hipSuccess;
  }
  MGC_CHECK((finish_launch<u64, 256, 16>(d_keys, d_starts, ng, low_bits, 0, FIN_CAP_SMALL, d_cnt_tmp, d_group_distinct, st)));
  MGC_CHECK((finish_launch<u64, 1024, 8>(d_keys, d_starts, n_large, low_bits, FIN_CAP_SMALL, FIN_CAP_LARGE, d_cnt_tmp,
                                         d_group_distinct, st, d_large_list)));
  return hipSuccess;
}

// the retry list of hash_count_stream_kernel: the same kernel with a table no sub-bucket of up to FIN_CAP_STREAM keys can overflow
hipError_t launch_finish_retry(void *d_keys32, const uint64_t *d_starts, uint64_t ng, uint32_t low_bits, uint32_t *d_cnt_tmp,
                               uint64_t *d_group_distinct, uint32_t tr_a, uint32_t tr_b, const uint32_t *d_retry_list,
                               const uint64_t *d_retry_count, uint64_t n_retry, uint64_t stream_cap, hipStream_t st, bool narrow, void *d_alt) {
  if (n_retry == 0) return hipSuccess;
  if (!finish_stream_ok(1, low_bits, narrow) || stream_cap == 0 || stream_cap > FIN_CAP_STREAM) return hipErrorInvalidValue;
  if (!narrow) {
    // whole 8-byte k-mers: a table for 4094 distinct 64-bit entries does not fit a workgroup's static LDS -- the sub-buckets on the
    // list go through the streaming kernel of the oversized ones, one workgroup each (huge_min = 0: every listed sub-bucket)
    if (!d_alt) return hipErrorInvalidValue;
    constexpr size_t B32 = (size_t)(4 + 4) * HUGE_SLOTS32 + (size_t)(4 + 4) * HUGE_CAP32 + 16 * 4;
    constexpr size_t B64 = (size_t)(8 + 4) * HUGE_SLOTS64 + (size_t)(8 + 4) * HUGE_CAP64 + 16 * 8;
    static bool rattr = false;
    if (!rattr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&hash_count_huge_kernel<u32, 1024, HUGE_CAP32, HUGE_SLOTS32>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)B32);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&hash_count_huge_kernel<u64, 1024, HUGE_CAP64, HUGE_SLOTS64>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)B64);
      rattr = true;
    }
    if (low_bits < 32)
      hipLaunchKernelGGL((hash_count_huge_kernel<u32, 1024, HUGE_CAP32, HUGE_SLOTS32>), dim3((uint32_t)n_retry), dim3(1024), B32, st,
                         reinterpret_cast<u64 *>(d_keys32), reinterpret_cast<const u64 *>(d_starts), d_retry_list, (u64)ng, (u64)0, low_bits,
                         d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct), reinterpret_cast<u64 *>(d_alt), tr_a, tr_b);
    else
      hipLaunchKernelGGL((hash_count_huge_kernel<u64, 1024, HUGE_CAP64, HUGE_SLOTS64>), dim3((uint32_t)n_retry), dim3(1024), B64, st,
                         reinterpret_cast<u64 *>(d_keys32), reinterpret_cast<const u64 *>(d_starts), d_retry_list, (u64)ng, (u64)0, low_bits,
                         d_cnt_tmp, reinterpret_cast<u64 *>(d_group_distinct), reinterpret_cast<u64 *>(d_alt), tr_a, tr_b);
    return hipGetLastError();
  }
  static_assert(FIN_STREAM_RETRY_DCAP >= (int)FIN_CAP_STREAM, "the retry table holds every suffix of a sub-bucket");
  const uint64_t gmax = 256ull * 2ull;
  hipLaunchKernelGGL((hash_count_stream_kernel<u32, 256, FIN_STREAM_KPC, FIN_STREAM_RETRY_SLOTS, FIN_STREAM_RETRY_DCAP, true, false>),
                     dim3((uint32_t)(n_retry < gmax ? n_retry : gmax)), dim3(256), 0, st, reinterpret_cast<u32 *>(d_keys32),
                     reinterpret_cast<const u64 *>(d_starts), (u64)ng, (u64)stream_cap, low_bits, d_cnt_tmp,
                     reinterpret_cast<u64 *>(d_group_distinct), (u64 *)nullptr, tr_a, tr_b, d_retry_list,
                     reinterpret_cast<const u64 *>(d_retry_count), (u32 *)nullptr, (u64 *)nullptr);
  return hipGetLastError();
}

bool     finish_stream_ok(uint32_t key_words, uint32_t low_bits, bool narrow) { return key_words == 1 && low_bits >= 8 && low_bits <= (narrow ? 20u : 52u); }
uint64_t finish_stream_capacity() { return FIN_CAP_STREAM; }
uint64_t finish_stream_distinct() { return FIN_STREAM_DCAP; }
uint64_t finish_stream_target(const Switches &sw) {
  if (sw.finish_target) return sw.finish_target;              // (tests make the sub-buckets tiny)
  return (FIN_CAP_HASH * 3) / 2;                              // 2304: twice the others' average; +7 sigma of a file's sub-buckets stays below the capacity
}

static uint64_t finish_small_capacity(uint32_t key_words, uint32_t low_bits) {
  if (key_words == 2) return finish_uses_hash(key_words, low_bits) ? FIN_CAP_HASH128 : 2048;
  return finish_uses_hash(key_words, low_bits) ? FIN_CAP_HASH : FIN_CAP_SMALL;
}

uint64_t finish_capacity_for(uint32_t key_words) { return key_words == 2 ? 8192 : FIN_CAP_LARGE; }
uint64_t finish_target_for(uint32_t key_words, const Switches &sw) {
  if (sw.finish_target) return sw.finish_target;              // (tests make the sub-buckets tiny)
  return key_words == 2 ? (FIN_CAP_HASH128 * 3) / 4 : (FIN_CAP_HASH * 3) / 4;
}

// group_distinct[0..ng_total) -> exclusive offsets in place, total at [ng_total]
size_t finish_scan_scratch_bytes(uint64_t ng_total) { return scan_scratch_elems(ng_total + 1) * sizeof(uint64_t); }
hipError_t launch_finish_scan(uint64_t *d_group, uint64_t ng_total, void *d_scratch, hipStream_t st) {
  return scan_u64_exclusive(reinterpret_cast<u64 *>(d_group), ng_total, reinterpret_cast<u64 *>(d_scratch),
                            reinterpret_cast<u64 *>(d_group) + ng_total, st);
}

hipError_t launch_compact_groups(const void *d_keys, uint32_t key_words, const uint32_t *d_cnt_tmp, const uint64_t *d_starts,
                                 const uint64_t *d_offs, uint64_t ng, void *d_out_keys, uint32_t *d_out_counts, hipStream_t st,
                                 uint32_t tr_a, uint32_t tr_b, const uint32_t *d_nz, uint64_t n_nz) {
  const dim3 grid((uint32_t)(((d_nz ? n_nz : ng) + 3) / 4));
  if (d_nz && n_nz == 0) return hipSuccess;
  if (key_words == 2)
    hipLaunchKernelGGL(compact_groups_kernel<K128>, grid, dim3(256), 0, st, reinterpret_cast<const K128 *>(d_keys), d_cnt_tmp,
                       reinterpret_cast<const u64 *>(d_starts), reinterpret_cast<const u64 *>(d_offs), (u64)ng,
                       reinterpret_cast<K128 *>(d_out_keys), d_out_counts, tr_a, tr_b, d_nz, (u64)n_nz);
  else
    hipLaunchKernelGGL(compact_groups_kernel<u64>, grid, dim3(256), 0, st, reinterpret_cast<const u64 *>(d_keys), d_cnt_tmp,
                       reinterpret_cast<const u64 *>(d_starts), reinterpret_cast<const u64 *>(d_offs), (u64)ng,
                       reinterpret_cast<u64 *>(d_out_keys), d_out_counts, tr_a, tr_b, d_nz, (u64)n_nz);
  return hipGetLastError();
}

hipError_t launch_compact_groups_narrow(const void *d_keys32, const uint32_t *d_cnt_tmp, const uint64_t *d_starts, const uint64_t *d_offs,
                                        uint64_t ng, uint64_t base, uint32_t low_bits, void *d_out_keys, uint32_t *d_out_counts, hipStream_t st,
                                        uint32_t tr_a, uint32_t tr_b, const uint32_t *d_nz, uint64_t n_nz) {
  if (d_nz && n_nz == 0) return hipSuccess;
  hipLaunchKernelGGL(compact_groups_narrow_kernel, dim3((uint32_t)(((d_nz ? n_nz : ng) + 3) / 4)), dim3(256), 0, st, reinterpret_cast<const u32 *>(d_keys32),
                     d_cnt_tmp, reinterpret_cast<const u64 *>(d_starts), reinterpret_cast<const u64 *>(d_offs), (u64)ng, (u64)base, low_bits,
                     reinterpret_cast<u64 *>(d_out_keys), d_out_counts, tr_a, tr_b, d_nz, (u64)n_nz);
  return hipGetLastError();
}

hipError_t launch_compact_groups_k96(const void *d_keys96, const uint32_t *d_cnt_tmp, const uint64_t *d_starts, const uint64_t *d_offs,
                                     uint64_t ng, uint64_t base_lo, uint64_t base_hi, void *d_out_keys, uint32_t *d_out_counts, hipStream_t st,
                                     uint32_t tr_a, uint32_t tr_b, const uint32_t *d_nz, uint64_t n_nz) {
  if (d_nz && n_nz == 0) return hipSuccess;
  hipLaunchKernelGGL(compact_groups_k96_kernel, dim3((uint32_t)(((d_nz ? n_nz : ng) + 3) / 4)), dim3(256), 0, st, reinterpret_cast<const K96 *>(d_keys96),
                     d_cnt_tmp, reinterpret_cast<const u64 *>(d_starts), reinterpret_cast<const u64 *>(d_offs), (u64)ng, (u64)base_lo, (u64)base_hi,
                     reinterpret_cast<K128 *>(d_out_keys), d_out_counts, tr_a, tr_b, d_nz, (u64)n_nz);
  return hipGetLastError();
}

hipError_t launch_widen_k96(const void *d_keys96, uint64_t n, uint64_t base_lo, uint64_t base_hi, void *d_out128, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const uint64_t g = (n + 255) / 256;
  hipLaunchKernelGGL(widen_k96_kernel, dim3((uint32_t)(g < 256u * 32u ? g : 256u * 32u)), dim3(256), 0, st, reinterpret_cast<const K96 *>(d_keys96), (u64)n,
                     (u64)base_lo, (u64)base_hi, reinterpret_cast<K128 *>(d_out128));
  return hipGetLastError();
}

hipError_t launch_widen_groups(const void *d_keys32, const uint64_t *d_starts, uint64_t ng, uint64_t base, uint32_t low_bits, void *d_out64,
                               hipStream_t st, uint32_t tr_a, uint32_t tr_b) {
  hipLaunchKernelGGL(widen_groups_kernel, dim3((uint32_t)((ng + 3) / 4)), dim3(256), 0, st, reinterpret_cast<const u32 *>(d_keys32),
                     reinterpret_cast<const u64 *>(d_starts), (u64)ng, (u64)base, low_bits, reinterpret_cast<u64 *>(d_out64), tr_a, tr_b);
  return hipGetLastError();
}

__global__ __launch_bounds__(256)
void sum_u64_kernel(const u64 *__restrict__ in, u64 n, u64 *__restrict__ out) {
  u64 v = 0;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) v += in[i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
  if (lane_id() == 0 && v) atomicAdd((unsigned long long *)out, v);
}
hipError_t launch_sum_u64(const uint64_t *d_in, uint64_t n, uint64_t *d_out, hipStream_t st) {
  MGC_CHECK(hipMemsetAsync(d_out, 0, sizeof(uint64_t), st));
  if (n == 0) return hipSuccess;
  const uint64_t blocks = (n + 1023) / 1024;
  hipLaunchKernelGGL(sum_u64_kernel, dim3((uint32_t)(blocks < 256 ? blocks : 256)), dim3(256), 0, st, reinterpret_cast<const u64 *>(d_in),
                     (u64)n, reinterpret_cast<u64 *>(d_out));
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



// Every MGC_* switch of the count path, from the environment (mgc_device.h: once per session / per bare-operator call)
Switches read_switches() {
  Switches sw;
  auto off = [](const char *n) { const char *e = getenv(n); return e && e[0] == '0'; };
  auto on1 = [](const char *n) { const char *e = getenv(n); return e && e[0] == '1'; };
  auto num = [](const char *n, uint64_t d) { const char *e = getenv(n); return (e && *e) ? strtoull(e, nullptr, 10) : d; };
  sw.fine_hist = !off("MGC_FINE_HIST"); sw.hpc_msd = !off("MGC_HPC_MSD"); sw.hpc_digits = !off("MGC_HPC_DIGITS");
  sw.const_k = !off("MGC_KMER_CONST_K"); sw.narrow = !off("MGC_NARROW"); sw.wide_msd = !off("MGC_WIDE_MSD");
  sw.group_pipe = !off("MGC_GROUP_PIPE"); sw.soa5 = !off("MGC_SOA5"); sw.k96 = !off("MGC_K96"); sw.finish = !off("MGC_FINISH");
  sw.nolist = on1("MGC_FINISH_NOLIST");
  sw.finish_trace = getenv("MGC_FINISH_TRACE") != nullptr; sw.group_dbg = getenv("MGC_GROUP_DBG") != nullptr; sw.hash_dbg = getenv("MGC_HASH_DBG") != nullptr;
  { const char *e = getenv("MGC_HASH_MULTI"); sw.hash_multi = (e && *e) ? atoi(e) : -1; }
  { const char *e = getenv("MGC_HASH_STREAM"); sw.hash_stream = (e && *e) ? atoi(e) : -1; }
  sw.min_top = (uint32_t)num("MGC_FINISH_MIN_TOP", 0);
  sw.finish_target = num("MGC_FINISH_TARGET", 0);
  sw.stream_max = num("MGC_STREAM_MAX", (uint64_t)1 << 22);
  sw.bucket_bases = num("MGC_BUCKET_BASES", 0);
  sw.huge_streams = (uint32_t)num("MGC_HUGE_STREAMS", 4);
  sw.huge_slices = !off("MGC_HUGE_SLICES");
  sw.pass_stagger = (uint32_t)num("MGC_PASS_STAGGER", 0);
  return sw;
}

hipError_t warm_finish() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&store_u64_kernel)); }

}  // namespace mgc
