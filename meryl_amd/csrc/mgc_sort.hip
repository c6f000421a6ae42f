// mgc_sort.hip -- radix sort / grouping passes (gfx950).
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

#include <algorithm>

namespace mgc {

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


template <typename K, int RB, int BLOCK, int KPT, int LB>
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
  static constexpr size_t WIN_BYTES = 8192;
  static constexpr size_t BYTES     = OFF_WIN + WIN_BYTES;
  // workgroups per CU the LDS budget admits (160 KiB per CU), capped at 2
  static constexpr int    WG_PER_CU = (2 * BYTES <= 160 * 1024) ? 2 : 1;
  static constexpr int    MIN_WAVES_PER_SIMD = (WG_PER_CU * BLOCK) / 256;
};

// LB 2: two digits per look-back granule (30-bit values, n < 2^30); LB 3: one digit per granule (62-bit values)
template <typename K, int RB, int BLOCK, int KPT, int LB>
__global__ __launch_bounds__(BLOCK, (RadixSmem<K, RB, BLOCK, KPT, LB>::MIN_WAVES_PER_SIMD))
void radix_scatter_kernel(const K *__restrict__ in, K *__restrict__ out, u64 n, u32 shift, u32 dmask,
                          const u64 *__restrict__ gbase,      // exclusive digit bases of this pass
                          u64 *__restrict__ status,           // [num_tiles][R/2 or R] granules of this pass (zeroed)
                          u32 *__restrict__ ticket, u32 *__restrict__ error_flag) {
  static_assert(LB == 2 || LB == 3, "window look-back with packed or wide granules");
  using SM = RadixSmem<K, RB, BLOCK, KPT, LB>;
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

  // Tile ids are tickets: every lower-numbered tile has started, hence is resident and will
  // publish its aggregate -- the look-back cannot deadlock (spins are bounded anyway).
  if (tid == 0) s_tmp[32] = atomicAdd(ticket, 1u);
  __syncthreads();
  const u64 tile = s_tmp[32];

  for (u32 i = tid; i < (u32)(NW * R * 3); i += BLOCK) s_whist[i] = 0;   // counters + match masks
  __syncthreads();

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

  // ---- rank inside the wave (stable: by lane order inside a row, rows in order) ----
  const u64 lt_mask = (1ull << lane) - 1ull;
  u32 ranks[KPT / 2];                                  // two 16-bit ranks per register (rank < TILE <= 2^14)
  {
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
  constexpr int GW = (LB == 3) ? 1 : 2;                 // digits per granule
  constexpr int G  = R / GW;                            // granules per tile
  constexpr u64 V62 = (1ull << 62) - 1;
  u64 *mine = status + tile * (u64)G + tid;
  __syncthreads();                                      // s_cnt / s_dbase visible
  if (tid < (u32)G) {
    const u32 c0 = s_cnt[GW * tid], c1 = (GW == 2) ? s_cnt[GW * tid + 1] : 0u;
    const u32 fl = (tile == 0) ? 2u : 1u;
    status_store(mine, (GW == 2) ? st_pack(fl, c0, fl, c1) : (((u64)fl << 62) | (u64)c0));
  }

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

  {
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

// NARROW (u64 keys only): the keys leave as 32-bit words WITHOUT the digit of this pass -- ((key >> (shift + digit_bits))
// << shift) | low `shift` bits -- because where a key lies now tells its digit (launch_group_narrow): every later pass and
// the finish move half the bytes.
template <typename K, bool NARROW> struct GroupOut { typedef K type; };
template <> struct GroupOut<u64, true> { typedef u32 type; };

// HIST2: the pass also takes the histogram of ANOTHER digit (the next pass's) of the keys it reads -- 512 LDS counters per
// workgroup, flushed with global atomics at the end -- so that nobody has to read the keys for it.
struct GroupExtra { u32 digit_bits /* NARROW */; u32 shift2, mask2; u64 *ghist2 /* HIST2 */;
                    u32 soa_hi_mask = 0 /* SOA: payload bits of the u8 array */;
                    u32 stagger_groups = 0, stagger_cycles = 0 /* STAGGER below: groups, cycles between two groups' starts */; };

// SOA (u64 keys): `in` is the 5-byte layout kmer_partition_kernel<SOA> leaves -- u32 in[n] low words, then u8[n] bits 32..39 --
// and a key is put together as it is fetched: four keys per lane and group from one 16-byte and one 4-byte load.
// PIPE (round 6; the 5-byte first pass): the tile AFTER the next one is fetched -- into registers, in the form it has in memory (5 bytes
// per key: 20 registers for 16 keys) -- a WHOLE tile ahead: its loads are issued behind the write-out of the current tile and are consumed
// behind the look-back of the NEXT one, so they are in flight beside a tile's ranking, scan and exchange as well (round 5's form asks for
// the next tile at the start of the look-back and waits for it at the next ranking: the persistent workgroups, in step because of the
// look-back chain, left the read side of the memory system idle for a third of a tile's time).  Tickets run two tiles ahead.
// Measured (profiles/r06_pipe_ab.txt): first pass 0.545 -> 0.465 ms per launch.  The 32-bit second pass LOSES with the same change
// (0.355 -> 0.385 / 0.45 ms at 20 words per thread: its walkers' wait for the staged words also waits for the granule they have just
// published -- stores count in vmcnt on gfx9 -- and 24 + 24 words per thread do not fit the register file): it keeps round 5's form.
// HPCD (round 6; `compress`): both digits are dense ranks (hpc_digit: ~30 VALU instructions, taken three times per key and pass plus once
// for HIST2 -- the whole-key passes of `compress` were VALU-bound at 0.357 ms per 69 M keys against 0.266 ms for plain bit digits): the
// 4096 ranks of the twelve bits a digit is taken from lie in 8 KiB of LDS behind the tile, a digit is one 16-bit LDS read.
template <typename K, int RB, int BLOCK, int KPT, bool DBG, bool NARROW = false, bool HIST2 = false, bool SOA = false, int PIPE = 0 /* 1: the next fetch before the write-out; 2: after it */,
          int HPCD = 0 /* 1: both digits are dense ranks; 2: only this pass's (the other digit, HIST2's, is a plain bit field) */>
__global__ __launch_bounds__(BLOCK, (GroupSmem<K, RB, BLOCK, KPT>::MIN_WAVES_PER_SIMD))
void radix_group_kernel(const K *__restrict__ in, typename GroupOut<K, NARROW>::type *__restrict__ out, u64 n, u32 shift, u32 dmask,
                        const u64 *__restrict__ gbase, u64 *__restrict__ status, u32 *__restrict__ ticket,
                        u32 *__restrict__ error_flag, u64 num_tiles_plain,
                        const u64 *__restrict__ region_start,   // [RS_MAX_RADIX + 1] or nullptr (plain tiles)
                        const u32 *__restrict__ region_tiles,   // [RS_MAX_RADIX + 1] exclusive; last = total tiles
                        GroupExtra ex, u64 *__restrict__ dbg) {
  const u32 digit_bits = ex.digit_bits;
  __shared__ u32 s_h2[HIST2 ? RS_MAX_RADIX : 1];
  if (HIST2) { for (u32 i = threadIdx.x; i < (u32)RS_MAX_RADIX; i += BLOCK) s_h2[i] = 0; }
  using SM = GroupSmem<K, RB, BLOCK, KPT>;
  using KO = KeyOps<K>;
  constexpr int R = SM::R, TILE = SM::TILE, G = R / 2;
  constexpr int WALK = (sizeof(K) == 4 || PIPE) ? 8 : 16;      // 32 words per thread leave registers for eight granules in flight, not sixteen (no spills)
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
  static_assert(!HPCD || SM::BYTES + 8192 <= 160 * 1024, "the rank table lies behind the tile");
  unsigned short *s_rank = reinterpret_cast<unsigned short *>(smem + (HPCD ? SM::BYTES : 0));
  if constexpr (HPCD) { for (u32 i = tid0; i < 4096u; i += BLOCK) s_rank[i] = (unsigned short)hpc_digit(i); }   // (in place at the first barrier below)
  auto dig  = [&](const K &key) __attribute__((always_inline)) -> u32 {
    if constexpr (HPCD) return (u32)s_rank[KO::digit(key, shift, 0xFFFu)];
    else return KO::digit(key, shift, dmask);
  };
  auto dig2 = [&](const K &key) __attribute__((always_inline)) -> u32 {
    if constexpr (HPCD == 1) return (u32)s_rank[KO::digit(key, ex.shift2, 0xFFFu)];
    else return KO::digit(key, ex.shift2, ex.mask2);
  };

  // this thread's slice of the region table stays in registers
  // (a narrowing pass is a first pass: plain tiles -- known at compile time, so that the region slice below costs it no registers: the
  // 5-byte pass spilled them and reloaded them at the top of every tile behind an s_waitcnt vmcnt(0), i.e. behind its own prefetch)
  const bool regions = !NARROW && (region_tiles != nullptr);
  u64 rt_lo = 0, rt_hi = 0, rs = 0, re = 0, total_tiles = num_tiles_plain;
  if (regions) {
    if (tid0 < (u32)RS_MAX_RADIX) {
      rt_lo = region_tiles[tid0]; rt_hi = region_tiles[tid0 + 1];
      rs = region_start[tid0];    re = region_start[tid0 + 1];
    }
    total_tiles = region_tiles[RS_MAX_RADIX];
  }
  // (values, not branches over the captured variables: the two-branch form made the compiler select between the ADDRESSES of `re` and
  // `n` -- both in scratch for it, 40 bytes per thread -- and the announcing thread's wave reloaded them behind an s_waitcnt vmcnt(0),
  // i.e. behind its write-out stores, at the top of every tile: round 6, found in the ISA)
  auto announce = [&](u64 t, int slot) __attribute__((always_inline)) {   // key range of tile t -> s_info[2*slot..]
    if (t >= total_tiles) return;
    const bool mine = regions ? (rt_lo <= t && t < rt_hi) : (tid0 == 0);   // exactly one thread (empty regions own no tile)
    const u64  kb   = regions ? rs + (t - rt_lo) * (u64)TILE : t * (u64)TILE;
    const u64  endk = regions ? re : n;
    if (mine) {
      s_info[2 * slot] = kb;
      s_info[2 * slot + 1] = (endk - kb < (u64)TILE) ? endk - kb : (u64)TILE;
    }
  };

  // Loads are wave-striped 16-byte vectors (VEC keys per lane and instruction: 1 KiB contiguous per wave instruction).  Fewer,
  // wider loads matter beyond issue slots: the walkers' granule loads queue behind the prefetch of the other twelve waves in
  // the CU's own memory pipeline, by INSTRUCTION -- with 4-byte loads of 32-bit words the look-back took 45 K cycles per tile
  // instead of 17 K (profiles/r02x_groupdbg.log).  Key j of a thread is element idx_of(j) of the tile; order inside a tile is free.
  static_assert(!SOA || (sizeof(K) == 8 && KPT % 4 == 0), "the 5-byte layout holds 8-byte keys");
  // (12-byte K96 records: four per lane and group = three 16-byte loads)
  constexpr int VEC = SOA ? 4 : (sizeof(K) == 12 ? (KPT % 4 == 0 ? 4 : 1) : ((sizeof(K) < 16 && KPT % (16 / sizeof(K)) == 0) ? (int)(16 / sizeof(K)) : 1));
  struct __attribute__((aligned(4))) KVec { K v[VEC]; };     // 4-byte alignment is all a file / region start guarantees
  auto idx_of = [&](int j) __attribute__((always_inline)) -> u32 {
    return w * (u32)(64 * KPT) + ((u32)(j / VEC) * 64u + lane) * (u32)VEC + (u32)(j % VEC);
  };
  K keys[KPT];
  auto fetch = [&](u64 kb, u32 nv) __attribute__((always_inline)) {
    if constexpr (SOA) {
      struct __attribute__((aligned(4))) LVec { u32 v[4]; };
      struct __attribute__((packed, aligned(1))) HWord { u32 v; };
      const u32 *lo32 = reinterpret_cast<const u32 *>(in) + kb;
      const uint8_t *hi8 = reinterpret_cast<const uint8_t *>(in) + 4ull * n + kb;
      const u32 hm = ex.soa_hi_mask;
#pragma unroll
      for (int g = 0; g < KPT / 4; g++) {
        const u32 first = idx_of(g * 4);
        if (nv == (u32)TILE || first + 4u <= nv) {
          const LVec l = *reinterpret_cast<const LVec *>(lo32 + first);
          const u32 h = reinterpret_cast<const HWord *>(hi8 + first)->v;
#pragma unroll
          for (int c = 0; c < 4; c++) keys[g * 4 + c] = (K)((u64)l.v[c] | ((u64)((h >> (8 * c)) & hm) << 32));
        } else {
#pragma unroll
          for (int c = 0; c < 4; c++)
            if (first + (u32)c < nv) keys[g * 4 + c] = (K)((u64)lo32[first + c] | ((u64)((u32)hi8[first + c] & hm) << 32));
        }
      }
      return;
    }
    const K *base = in + kb;
#pragma unroll
    for (int g = 0; g < KPT / VEC; g++) {
      const u32 first = idx_of(g * VEC);
      if (nv == (u32)TILE || first + (u32)VEC <= nv) {
        const KVec q = *reinterpret_cast<const KVec *>(base + first);
#pragma unroll
        for (int c = 0; c < VEC; c++) keys[g * VEC + c] = q.v[c];
      } else {
#pragma unroll
        for (int c = 0; c < VEC; c++) if (first + (u32)c < nv) keys[g * VEC + c] = base[first + c];
      }
    }
  };

  // PIPE: a tile in the form it has in memory -- SOA: per group of four keys four low words + one word of high bytes; u32: the words
  static_assert(!PIPE || SOA || sizeof(K) == 4, "the pipelined fetch holds 5-byte or 4-byte keys");
  constexpr int RAWN = PIPE ? (SOA ? KPT + KPT / 4 : KPT) : 1;
  u32 raw[RAWN];
#pragma unroll
  for (int i = 0; i < RAWN; i++) raw[i] = 0u;
  auto fetch_raw = [&](u64 kb, u32 nv) __attribute__((always_inline)) {
    if constexpr (PIPE && SOA) {
      struct __attribute__((aligned(4))) LVec { u32 v[4]; };
      struct __attribute__((packed, aligned(1))) HWord { u32 v; };
      const u32 *lo32 = reinterpret_cast<const u32 *>(in) + kb;
      const uint8_t *hi8 = reinterpret_cast<const uint8_t *>(in) + 4ull * n + kb;
#pragma unroll
      for (int g = 0; g < KPT / 4; g++) {
        const u32 first = idx_of(g * 4);
        if (nv == (u32)TILE || first + 4u <= nv) {
          const LVec l = *reinterpret_cast<const LVec *>(lo32 + first);
#pragma unroll
          for (int c = 0; c < 4; c++) raw[g * 5 + c] = l.v[c];
          raw[g * 5 + 4] = reinterpret_cast<const HWord *>(hi8 + first)->v;
        } else {
          u32 h = 0;
#pragma unroll
          for (int c = 0; c < 4; c++)
            if (first + (u32)c < nv) { raw[g * 5 + c] = lo32[first + c]; h |= (u32)hi8[first + c] << (8 * c); }
          raw[g * 5 + 4] = h;
        }
      }
    } else if constexpr (PIPE) {
      const u32 *base = reinterpret_cast<const u32 *>(in) + kb;
      struct __attribute__((aligned(4))) WVec { u32 v[4]; };
      static_assert(!PIPE || SOA || KPT % 4 == 0, "four words per load");
#pragma unroll
      for (int g = 0; g < KPT / 4; g++) {
        const u32 first = idx_of(g * 4);
        if (nv == (u32)TILE || first + 4u <= nv) {
          const WVec q = *reinterpret_cast<const WVec *>(base + first);
#pragma unroll
          for (int c = 0; c < 4; c++) raw[g * 4 + c] = q.v[c];
        } else {
#pragma unroll
          for (int c = 0; c < 4; c++) if (first + (u32)c < nv) raw[g * 4 + c] = base[first + c];
        }
      }
    }
  };
  auto assemble = [&]() __attribute__((always_inline)) {
    if constexpr (PIPE && SOA) {
      const u32 hm = ex.soa_hi_mask;
#pragma unroll
      for (int g = 0; g < KPT / 4; g++)
#pragma unroll
        for (int c = 0; c < 4; c++) keys[g * 4 + c] = (K)((u64)raw[g * 5 + c] | ((u64)((raw[g * 5 + 4] >> (8 * c)) & hm) << 32));
    } else if constexpr (PIPE) {
#pragma unroll
      for (int j = 0; j < KPT; j++) keys[j] = (K)raw[j];
    }
  };

  // STAGGER (round 6; the 5-byte first pass): the persistent workgroups start in ex.stagger_groups groups, group g a g/P-th of a
  // tile's time behind group 0, and stay apart -- nothing but the look-back couples them, and it only asks that earlier tiles are not
  // behind.  Started together the ~256 workgroups run in step: a tile's walkers then find nothing but aggregates among their
  // predecessors and the inclusive prefixes spread from the oldest tile on; apart, the tiles of the groups ahead have their prefixes
  // out.  The first tiles are handed out statically in the order in which they will be processed (burst i * P + g = iteration i of
  // group g; a ticket taken while another group sleeps would put a LATER tile in front of its predecessors: measured, +9 %), the
  // tickets continue behind them.  Measured (profiles/r06_ab_runs.txt, r06x_rb8b): first pass 0.453 -> 0.444 ms per launch.
  // OPT-IN (MGC_PASS_STAGGER=8), not the default: with tickets a workgroup that is not resident yet holds no tile, so the walkers only
  // ever wait for workgroups that run; a static hand-out gives tiles to workgroups that may still wait for a CU -- another process or
  // session holding part of the device -- and the running ones would spin for them until the spin limit (MGC_ETIMEOUT).
  const bool stag = ex.stagger_groups > 1u && gridDim.x % (8u * ex.stagger_groups) == 0u;
  u32 st_g = 0, st_r = 0, st_w = 0, tk_off = 0;
  if (stag) {
    st_g = (blockIdx.x >> 3) % ex.stagger_groups;          // (consecutive workgroups go to the eight XCDs in turn: every group on all of them)
    st_r = ((blockIdx.x >> 3) / ex.stagger_groups) * 8u + (blockIdx.x & 7u);
    st_w = gridDim.x / ex.stagger_groups;
    tk_off = (PIPE ? 3u : 2u) * gridDim.x;
    if (st_g) {
      const u64 c0 = __builtin_readcyclecounter(), want = (u64)st_g * ex.stagger_cycles;
      while (__builtin_readcyclecounter() - c0 < want) __builtin_amdgcn_s_sleep(16);
    }
  }
  if (tid == 0) s_tmp[32] = stag ? st_g * st_w + st_r : atomicAdd(ticket, 1u);
  __syncthreads();
  u64 tile = s_tmp[32];
  announce(tile, 0);
  __syncthreads();
  u64 kb = 0; u32 nv = 0;
  u64 tile1 = 0, kb1 = 0; u32 nv1 = 0;                    // PIPE: the tile after the current one (its keys in flight / in raw[])
  if constexpr (!PIPE) {
    if (tile < total_tiles) { kb = s_info[0]; nv = (u32)s_info[1]; fetch(kb, nv); }
  } else {
    if (tile < total_tiles) { kb = s_info[0]; nv = (u32)s_info[1]; fetch_raw(kb, nv); assemble(); }
    if (tid == 0) s_tmp[33] = stag ? (ex.stagger_groups + st_g) * st_w + st_r : atomicAdd(ticket, 1u);
    __syncthreads();
    tile1 = s_tmp[33];
    announce(tile1, 1);
    __syncthreads();
    if (tile1 < total_tiles) { kb1 = s_info[2]; nv1 = (u32)s_info[3]; fetch_raw(kb1, nv1); }
    __syncthreads();                                      // (s_tmp[33] and s_info[2..3] are written again at the top of the loop)
  }
  // the ticket of the tile after those (from now on taken in the shadow of the look-back, below)
  if (tid == 0) s_tmp[34] = stag ? ((PIPE ? 2u : 1u) * ex.stagger_groups + st_g) * st_w + st_r : atomicAdd(ticket, 1u);

  u64 ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
#define PK_STAMP(i) do { if (DBG) { const u64 t = __builtin_readcyclecounter(); ph[i] += t - t0; t0 = t; } } while (0)
  while (tile < total_tiles) {
    if (DBG) t0 = __builtin_readcyclecounter();
    tid = tid0;
    asm volatile("" : "+v"(tid));                         // keeps the unrolled body's LDS addresses out of the loop preheader
    lane = tid & 63u; w = tid >> 6;
    const bool walker = tid < (u32)G;
    if (tid < (u32)R) s_hist[tid] = 0;
    __syncthreads();                                      // (A)
    const u64 next = s_tmp[34];
    announce(next, 1);                                    // read after (C)
    PK_STAMP(0);

    // ---- rank: position among the tile's keys of the same digit, in arrival order ----
    u32 ranks[KPT / 2];
    u32 dgs[HPCD ? KPT / 2 : 1];                          // HPCD: the digits stay in registers for the exchange (one table read less per key)
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      u32 r = 0, d = 0;
      if (idx_of(j) < nv) {
        d = dig(keys[j]);
        r = atomicAdd(&s_hist[d], 1u);
      }
      if (j & 1) ranks[j / 2] |= r << 16;
      else       ranks[j / 2]  = r;
      if constexpr (HPCD) { if (j & 1) dgs[j / 2] |= d << 16; else dgs[j / 2] = d; }
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
      if (idx_of(j) < nv) {
        u32 d;
        if constexpr (HPCD) d = (j & 1) ? (dgs[j / 2] >> 16) : (dgs[j / 2] & 0xFFFFu);
        else d = dig(keys[j]);
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
      // the last thread asks for the ticket of the tile after `next` here and hands it over before (E) (read behind the next (A)): its
      // wave's wait for the answer -- vmcnt is in order: also for what the wave has in flight -- lies in the shadow of the look-back
      // (until round 6 tid 0 asked at the top of the loop and the whole workgroup waited at (A) behind that wave's write-out stores and,
      // PIPE, the prefetch it had just issued: ~2 K cycles per tile)
      u32 tk_pre = 0;
      if (tid == (u32)BLOCK - 1u) tk_pre = atomicAdd(ticket, 1u) + tk_off;
      if constexpr (!PIPE) { if (next < total_tiles) fetch(nkb, nnv); }
      // the waves that do not walk would only wait now: they count the other digit of the tile's keys (in LDS, in digit
      // order since the exchange) -- LDS work in the shadow of the look-back
      if constexpr (HIST2) {
        for (u32 i = tid - (u32)G; i < nv; i += (u32)(BLOCK - G)) atomicAdd(&s_h2[dig2(s_keys[i])], 1u);
      }
      if (tid == (u32)BLOCK - 1u) s_tmp[34] = tk_pre;
    } else {
      while (!done) { issue(); consume(); }
      if (tile != 0) status_store(mine, st_pack(2, p0 + c0, 2, p1 + c1));
      s_gbase[2 * tid]     = gbase[2 * tid]     + (u64)p0 - (u64)s_dbase[2 * tid];
      s_gbase[2 * tid + 1] = gbase[2 * tid + 1] + (u64)p1 - (u64)s_dbase[2 * tid + 1];
      if constexpr (!PIPE) { if (next < total_tiles) fetch(nkb, nnv); }
    }
    __syncthreads();                                      // (E)
    PK_STAMP(3);
    if constexpr (PIPE) {
      // the tile fetched a whole tile ago leaves raw[] for keys[] (nothing newer is in flight when the wait for it runs: the
      // write-out's stores come AFTER this point), and the tile after it is asked for
      if (tile1 < total_tiles) assemble();
      if (PIPE == 1 && next < total_tiles) fetch_raw(nkb, nnv);
    }

#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const u32 i = (u32)j * BLOCK + tid;
      if (i < nv) {
        const K   key = s_keys[i];
        const u32 d   = dig(key);
        if constexpr (NARROW) out[s_gbase[d] + (u64)i] = (u32)(((key >> (shift + digit_bits)) << shift) | (key & ((1ull << shift) - 1ull)));
        else                  out[s_gbase[d] + (u64)i] = key;
      }
    }
    PK_STAMP(4);
    if constexpr (PIPE == 2) { if (next < total_tiles) fetch_raw(nkb, nnv); }
    __syncthreads();                                      // (F)
    PK_STAMP(5);
    if constexpr (PIPE) { tile = tile1; kb = kb1; nv = nv1; tile1 = next; kb1 = nkb; nv1 = nnv; }
    else                { tile = next; kb = nkb; nv = nnv; }
    if (DBG) ph[7]++;
  }
  if (DBG && tid0 == 0 && blockIdx.x < 64)
    for (int i = 0; i < 8; i++) dbg[blockIdx.x * 8 + i] = ph[i];
#undef PK_STAMP
  (void)kb;
  if constexpr (HIST2) {
    __syncthreads();
    for (u32 i = tid0; i < (u32)RS_MAX_RADIX; i += BLOCK) { const u32 c = s_h2[i]; if (c) atomicAdd(&ex.ghist2[i], (u64)c); }
  }
}

struct NarrowPrep { unsigned char bits_a[256]; unsigned char on[256]; };   // per bucket: bits of its first (high) digit, narrowed at all

// ---- host side ---------------------------------------------------------------

void make_sort_plan(uint32_t begin_bit, uint32_t end_bit, SortPlan *plan) {
  memset(plan, 0, sizeof(*plan));
  // the fastest measured combination on MI355X (profiles/r01*, DESIGN_HISTORY.md): nine-bit digits, one 1024-thread workgroup
  // per CU with 16 keys per thread (8 for 16-byte keys), LDS-mask ranking, window look-back after the exchange.  The other
  // shapes (eight-bit digits, 512-thread workgroups, ballot ranking, serial and pipelined look-back, the classic
  // histogram / scan / scatter form) were measured in rounds 1-2 and removed in round 5.
  plan->radix_bits = 9;
  plan->block      = 1024;
  plan->kpt        = 16;
  plan->tile       = plan->block * plan->kpt;
  plan->mode       = 0;                                                 // 0 stable sort; the caller sets 3 for grouping passes
  const uint32_t rb = plan->radix_bits;
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

void make_hpc_group_plan(uint32_t low_bit, uint32_t passes, SortPlan *plan) {
  make_sort_plan(0, 18, plan);                       // tile shape, look-back flavour, env overrides
  plan->radix_bits = 9;                              // 243 of the 512 bins are used
  plan->mode = 3;
  plan->hpc = 1;
  plan->num_passes = passes;
  for (uint32_t p = 0; p < passes; p++) { plan->pass_shift[p] = low_bit + 10 * p; plan->pass_bits[p] = 10; }
}

// `compress`, a bucket of 14..45 M k-mers on the distinct-sized count: the HIGH digit a dense rank of five bases (243 values), the LOW
// one the plain eight bits of the four bases below them (81 of 256 patterns occur): 19683 occupied sub-buckets instead of 59049
void make_hpc_mixed_plan(uint32_t low_bit, SortPlan *plan) {
  make_hpc_group_plan(low_bit, 2, plan);
  plan->hpc = 2;
  plan->pass_shift[0] = low_bit;     plan->pass_bits[0] = 8;
  plan->pass_shift[1] = low_bit + 8; plan->pass_bits[1] = 10;
}

// the `mask` argument of a pass's kernels (hpc == 2: only the high digit of the two is a dense rank)
static inline u32 plan_mask(const SortPlan &plan, uint32_t p) {
  return (plan.hpc == 1 || (plan.hpc == 2 && p == 1)) ? HPC_DIGIT_MASK : (1u << plan.pass_bits[p]) - 1u;
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

// digit histograms + exclusive digit bases of a file's grouping passes into `hdr` (tickets zeroed)
template <typename K>
static hipError_t group_prepare(const K *src, uint64_t n, const SortPlan &plan, SortHeader *hdr, hipStream_t st) {
  MGC_CHECK(hipMemsetAsync(hdr, 0, sizeof(SortHeader), st));
  PassList pl;
  pl.n = plan.num_passes;
  for (uint32_t p = 0; p < plan.num_passes; p++) {
    pl.shift[p] = plan.pass_shift[p];
    pl.mask[p]  = plan_mask(plan, p);
  }
  launch_radix_hist<K>(src, (u64)n, pl, &hdr->ghist[0][0], st);
  MGC_CHECK(hipGetLastError());
  hipLaunchKernelGGL(radix_digit_scan_kernel, dim3(plan.num_passes), dim3(RS_MAX_RADIX), 0, st,
                     &hdr->ghist[0][0], &hdr->gbase[0][0]);
  return hipGetLastError();
}

// LB 2: packed granules (n < 2^30); LB 3: wide granules
template <typename K, int KPT, int LB>
static hipError_t run_passes(void *d_keys, void *d_alt, uint64_t n, const SortPlan &plan, void *d_ws,
                             uint32_t *d_error, int *result_in_alt, hipStream_t st, hipEvent_t *pass_events) {
  constexpr int RB = 9, BLOCK = 1024;
  using SM = RadixSmem<K, RB, BLOCK, KPT, LB>;
  constexpr int R = 1 << RB, TILE = BLOCK * KPT;
  SortHeader *hdr = reinterpret_cast<SortHeader *>(d_ws);
  unsigned char *body = reinterpret_cast<unsigned char *>(d_ws) + ((sizeof(SortHeader) + 255) / 256) * 256;
  const uint64_t num_tiles = (n + TILE - 1) / TILE;

  K *src = reinterpret_cast<K *>(d_keys), *dst = reinterpret_cast<K *>(d_alt);
  int in_alt = 0;

  if (plan.mode != 3) {
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_scatter_kernel<K, RB, BLOCK, KPT, LB>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)SM::BYTES);
      attr_done = true;
    }
    u64 *status = reinterpret_cast<u64 *>(body);
    const size_t status_bytes = (size_t)num_tiles * (LB == 3 ? R : R / 2) * sizeof(u64);
    MGC_CHECK(hipMemsetAsync(hdr, 0, sizeof(SortHeader), st));
    PassList pl;
    pl.n = plan.num_passes;
    for (uint32_t p = 0; p < plan.num_passes; p++) {
      pl.shift[p] = plan.pass_shift[p];
      pl.mask[p]  = plan_mask(plan, p);
    }
    launch_radix_hist<K>((const K *)src, (u64)n, pl, &hdr->ghist[0][0], st);
    MGC_CHECK(hipGetLastError());
    hipLaunchKernelGGL(radix_digit_scan_kernel, dim3(plan.num_passes), dim3(RS_MAX_RADIX), 0, st,
                       &hdr->ghist[0][0], &hdr->gbase[0][0]);
    MGC_CHECK(hipGetLastError());
    for (uint32_t p = 0; p < plan.num_passes; p++) {
      MGC_CHECK(hipMemsetAsync(status, 0, status_bytes, st));
      if (pass_events) MGC_CHECK(hipEventRecord(pass_events[2 * p], st));
      hipLaunchKernelGGL((radix_scatter_kernel<K, RB, BLOCK, KPT, LB>), dim3((uint32_t)num_tiles),
                         dim3(BLOCK), SM::BYTES, st, (const K *)src, dst, (u64)n, plan.pass_shift[p],
                         plan_mask(plan, p), &hdr->gbase[p][0], status, &hdr->ticket[p], d_error);
      MGC_CHECK(hipGetLastError());
      if (pass_events) MGC_CHECK(hipEventRecord(pass_events[2 * p + 1], st));
      K *t = src; src = dst; dst = t; in_alt ^= 1;
    }
  } else {
    // ---- grouping passes (finish path): see radix_group_kernel ----
    using GS = GroupSmem<K, RB, BLOCK, KPT>;
    static bool gattr_done = false;
    if (!gattr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel<K, RB, BLOCK, KPT, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS::BYTES);
      gattr_done = true;
    }
    const uint64_t max_tiles = num_tiles + RS_MAX_RADIX + 1;          // region-aligned tiles: one partial tile per region
    u64 *status = reinterpret_cast<u64 *>(body);
    const size_t status_bytes = (size_t)max_tiles * (R / 2) * sizeof(u64);
    u64 *region_start = reinterpret_cast<u64 *>(body + ((status_bytes + 255) / 256) * 256);
    u32 *region_tiles = reinterpret_cast<u32 *>(region_start + RS_MAX_RADIX + 1);
    MGC_CHECK(group_prepare<K>((const K *)src, n, plan, hdr, st));
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
      hipLaunchKernelGGL((radix_group_kernel<K, RB, BLOCK, KPT, false>), dim3(pgrid), dim3(BLOCK), GS::BYTES, st,
                         (const K *)src, dst, (u64)n, plan.pass_shift[p], plan_mask(plan, p),
                         &hdr->gbase[p][0], status, &hdr->ticket[p], d_error, (u64)num_tiles, rs, rt, GroupExtra{0u, 0u, 0u, nullptr}, (u64 *)nullptr);
      MGC_CHECK(hipGetLastError());
      if (pass_events) MGC_CHECK(hipEventRecord(pass_events[2 * p + 1], st));
      K *t = src; src = dst; dst = t; in_alt ^= 1;
    }
  }
  *result_in_alt = in_alt;
  return hipSuccess;
}

// ---- two grouping passes with narrowed keys (the finish path of k-mers that leave <= 32 bits below their first digit) ----
// Sub-bucket v = (digit of pass 1 : digit of pass 0).  Pass 1's tiles never mix two pass-0 digits (regions), its output is
// ordered by (pass-1 digit, tile), so sub-bucket v starts at  gbase1[d1] + the inclusive prefix of d1 at the last tile
// before region d0  -- which is exactly what the look-back granules of pass 1 hold once the pass is over: the boundaries
// come for free, no key has to be looked at (the keys do not even hold d0 any more).
__global__ void narrow_bounds_kernel(const u64 *__restrict__ status, const u32 *__restrict__ region_tiles,
                                     const u64 *__restrict__ gbase1, u64 n, u32 b0, u64 ng, u64 *__restrict__ starts,
                                     u32 granules /* per tile: half the digits of the pass's instantiation */) {
  const u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (v > ng) return;
  if (v == ng) { starts[v] = n; return; }
  u32 d1 = (u32)(v >> b0), d0 = (u32)v & ((1u << b0) - 1u);
  // dense-rank digits (`compress`) live in ten-bit fields but stay below 364: nothing lies at or beyond a digit >= RS_MAX_RADIX
  if (d1 >= (u32)RS_MAX_RADIX) { starts[v] = n; return; }
  if (d0 > (u32)RS_MAX_RADIX) d0 = RS_MAX_RADIX;
  const u32 tiles_before = region_tiles[d0];
  u64 before = 0;
  if (tiles_before) {
    if ((d1 >> 1) >= granules) { starts[v] = n; return; }   // (a digit the pass's instantiation does not have: nothing lies there)
    const u64 g = status[(u64)(tiles_before - 1) * granules + (d1 >> 1)];
    before = (u64)(((d1 & 1u) ? (u32)(g >> 32) : (u32)g) & 0x3FFFFFFFu);
  }
  starts[v] = gbase1[d1] + before;
}

bool sort_plan_narrows(const SortPlan &plan, uint64_t n, uint32_t key_words, bool on) {
  return on && key_words == 1 && plan.mode == 3 && !plan.hpc && plan.num_passes == 2 && plan.radix_bits == 9 && n > 0 && n < (1ull << 30) &&
         plan.pass_shift[1] == plan.pass_shift[0] + plan.pass_bits[0] &&
         plan.pass_shift[0] + std::max(plan.pass_bits[0], plan.pass_bits[1]) <= 32;     // whichever digit goes first, the rest fits a word
}

// ---- the small per-file steps of the high-digit-first form, batched over all files (a launch per file and step costs more idle
// time than the steps themselves: ~8 us each, ten of them per file) ----

// one workgroup per file: header cleared, histogram of the file's top bits_a[f] bits from the 512 fine counts, its exclusive scan
// fb: fine holds 2^fb counts per bucket (the fifteen-bit histogram of 2^(15 - fb) buckets: 9 for the 64 files, 8 / 7 for the 128 / 256
// buckets of a sharded count); bits_a[f] <= fb
__global__ __launch_bounds__(RS_MAX_RADIX)
void narrow_prepare_kernel(const u64 *__restrict__ fine, NarrowPrep prep, unsigned char *__restrict__ hdrs, u32 hdr_stride, u32 fb) {
  __shared__ u64 s_f[RS_MAX_RADIX];
  __shared__ u64 s_tmp[RS_MAX_RADIX / 64 + 1];
  const u32 f = blockIdx.x, x = threadIdx.x;
  if (!prep.on[f]) return;
  SortHeader *hdr = reinterpret_cast<SortHeader *>(hdrs + (size_t)f * hdr_stride);
  u32 *w = reinterpret_cast<u32 *>(hdr);
  for (u32 i = x; i < sizeof(SortHeader) / 4; i += RS_MAX_RADIX) w[i] = 0;
  s_f[x] = (x < (1u << fb)) ? fine[((size_t)f << fb) + x] : 0ull;
  __syncthreads();
  const u32 bits = prep.bits_a[f], span = 1u << (fb - bits);
  u64 c = 0;
  if (x < (1u << bits)) for (u32 i = 0; i < span; i++) c += s_f[x * span + i];
  u64 total;
  const u64 e = block_excl_scan<RS_MAX_RADIX, u64>(c, s_tmp, &total);
  hdr->ghist[0][x] = c;
  hdr->gbase[0][x] = e;
}

// between the two passes: exclusive scan of the second digit's histogram (taken by the first pass) + the region table
__global__ __launch_bounds__(RS_MAX_RADIX)
void narrow_mid_kernel(SortHeader *__restrict__ hdr, u64 n, u32 tile, u64 *__restrict__ region_start, u32 *__restrict__ region_tiles) {
  __shared__ u64 s_tmp[RS_MAX_RADIX / 64 + 1];
  __shared__ u32 s_tmp32[RS_MAX_RADIX / 64 + 1];
  const u32 r = threadIdx.x;
  u64 total;
  const u64 e = block_excl_scan<RS_MAX_RADIX, u64>(hdr->ghist[1][r], s_tmp, &total);
  hdr->gbase[1][r] = e;
  const u64 a = hdr->gbase[0][r], b = (r + 1 < RS_MAX_RADIX) ? hdr->gbase[0][r + 1] : n;
  const u32 nt = (u32)((b - a + tile - 1) / tile);
  u32 tt;
  const u32 et = block_excl_scan<RS_MAX_RADIX, u32>(nt, s_tmp32, &tt);
  region_start[r] = a;
  region_tiles[r] = et;
  if (r == 0) { region_start[RS_MAX_RADIX] = n; region_tiles[RS_MAX_RADIX] = tt; }
}

// per-file scratch of the batched form: [status of pass A][status of pass B][region table]
constexpr uint64_t NARROW_TILE0 = 16384, NARROW_TILE1 = 24576;   // keys per tile of the first / second pass (launch_group_narrow)
constexpr uint32_t NARROW_TILE0_CYCLES = 31000;                 // a first-pass tile's time (MGC_GROUP_DBG): the staggered groups share it
size_t narrow_scratch_bytes(uint64_t n) {
  const uint64_t tiles0 = (n + NARROW_TILE0 - 1) / NARROW_TILE0, tiles1_max = (n + NARROW_TILE1 - 1) / NARROW_TILE1 + RS_MAX_RADIX + 1;
  return (size_t)(tiles0 + tiles1_max) * (RS_MAX_RADIX / 2) * sizeof(u64) + (size_t)(RS_MAX_RADIX + 2) * 16 + 512;
}

// headers of all files at once (files with on[f] = 0 are skipped); d_hdrs: nb x sort_header_bytes()
hipError_t launch_narrow_prepare(const uint64_t *d_fine, uint32_t nb, const unsigned char *bits_a, const unsigned char *on, void *d_hdrs,
                                 hipStream_t st) {
  if (nb != 64 && nb != 128 && nb != 256) return hipErrorInvalidValue;
  const u32 fb = nb == 64 ? 9u : (nb == 128 ? 8u : 7u);
  NarrowPrep prep;
  memset(&prep, 0, sizeof(prep));
  memcpy(prep.bits_a, bits_a, nb);
  memcpy(prep.on, on, nb);
  for (uint32_t b = 0; b < nb; b++) if (prep.on[b] && prep.bits_a[b] > fb) return hipErrorInvalidValue;
  hipLaunchKernelGGL(narrow_prepare_kernel, dim3(nb), dim3(RS_MAX_RADIX), 0, st, reinterpret_cast<const u64 *>(d_fine), prep,
                     reinterpret_cast<unsigned char *>(d_hdrs), (u32)sort_header_bytes(), fb);
  return hipGetLastError();
}

// d_keys: u64[n] in; u32[n] out over its first half (grouped by the plan's two digits, each key without the digit of the
// FIRST pass, truncated to 32 bits).  d_alt: room for n u32.  d_sub_starts: 2^(b0+b1) + 1, in PHYSICAL order.
// d_prepared == nullptr: the low digit first (LSD; one read of the keys for both digit histograms, scratch = d_ws); physical
// order = key order.
// d_prepared (this file's header from launch_narrow_prepare, i.e. the histogram of its HIGH digit taken from the fifteen-bit
// file histogram) + d_scratch (narrow_scratch_bytes(n), its status part zeroed by the caller): the high digit goes first, the
// low digit's histogram is taken by the first pass itself -- nobody reads the keys for a histogram -- and the physical order
// is (low digit : high digit): sub-bucket p holds the k-mers whose top bits are
// ((p & (2^*tr_a - 1)) << *tr_b) | (p >> *tr_a)  (*tr_a = 0: p itself).
hipError_t launch_group_narrow(void *d_keys, void *d_alt, uint64_t n, const SortPlan &plan, void *d_ws, size_t ws_bytes,
                               uint32_t *d_error, uint64_t *d_sub_starts, hipStream_t st, hipEvent_t *pass_events,
                               void *d_prepared, void *d_scratch, uint32_t *tr_a, uint32_t *tr_b, uint32_t soa_hi_mask, bool group_dbg, bool pipe, uint32_t stagger) {
  if (!sort_plan_narrows(plan, n, 1) || ws_bytes < sort_workspace_bytes(n)) return hipErrorInvalidValue;
  constexpr int RB = 9, BLOCK = 1024, KPT0 = 16, KPT1 = 24, R = 1 << RB;
  using GS0 = GroupSmem<u64, RB, BLOCK, KPT0>;
  using GS1 = GroupSmem<u32, RB, BLOCK, KPT1>;
  constexpr uint64_t TILE0 = (uint64_t)BLOCK * KPT0, TILE1 = (uint64_t)BLOCK * KPT1;
  static_assert(TILE0 == NARROW_TILE0 && TILE1 == NARROW_TILE1, "narrow_scratch_bytes");
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel<u64, RB, BLOCK, KPT0, false, true, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS0::BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel<u64, RB, BLOCK, KPT0, false, true, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS0::BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel<u32, RB, BLOCK, KPT1, false, false, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS1::BYTES);
    attr_done = true;
  }
  const bool msd = d_prepared != nullptr && d_scratch != nullptr;
  const uint64_t tile1 = TILE1;    // (28 words per thread -- 28672-word tiles, the most the register file takes -- measured equal: r06_ab_runs.txt)
  const uint64_t tiles0 = (n + TILE0 - 1) / TILE0, tiles1_max = (n + tile1 - 1) / tile1 + RS_MAX_RADIX + 1;
  SortHeader *hdr;
  u64 *status_a, *status_b, *region_start;
  if (msd) {
    hdr = reinterpret_cast<SortHeader *>(d_prepared);
    status_a = reinterpret_cast<u64 *>(d_scratch);
    status_b = status_a + (size_t)tiles0 * (R / 2);
    region_start = status_b + (size_t)tiles1_max * (R / 2);
  } else {
    hdr = reinterpret_cast<SortHeader *>(d_ws);
    unsigned char *body = reinterpret_cast<unsigned char *>(d_ws) + ((sizeof(SortHeader) + 255) / 256) * 256;
    status_a = status_b = reinterpret_cast<u64 *>(body);
    const size_t status_bytes = (size_t)std::max(tiles0, tiles1_max) * (R / 2) * sizeof(u64);
    region_start = reinterpret_cast<u64 *>(body + ((status_bytes + 255) / 256) * 256);
  }
  u32 *region_tiles = reinterpret_cast<u32 *>(region_start + RS_MAX_RADIX + 1);
  const u32 low = plan.pass_shift[0], b_lo = plan.pass_bits[0], b_hi = plan.pass_bits[1];
  // first pass: digit A at shA (bA bits), dropped from the keys; second pass: digit B -- after the drop it sits at `low`
  const u32 bA = msd ? b_hi : b_lo, bB = msd ? b_lo : b_hi, shA = msd ? low + b_lo : low;
  *tr_a = msd ? bA : 0u; *tr_b = bB;
  const uint64_t cus = (uint64_t)device_cu_count();

  if (!msd) {
    MGC_CHECK(group_prepare<u64>(reinterpret_cast<const u64 *>(d_keys), n, plan, hdr, st));   // rows 0 / 1 = low / high digit = A / B
    MGC_CHECK(hipMemsetAsync(status_a, 0, (size_t)tiles0 * (R / 2) * sizeof(u64), st));
  }
  if (pass_events) MGC_CHECK(hipEventRecord(pass_events[0], st));
  const dim3 grid0((uint32_t)std::min(tiles0, cus * GS0::WG_PER_CU));
  // MGC_GROUP_DBG=1: per-phase cycle sums of the first 64 workgroups of both passes, printed for the first two files (developer
  // instrumentation; the instrumented instantiations run instead of the plain ones for those files)
  static int dbg_reports = 2;
  static u64 *dbg_buf = nullptr;
  const bool dbg = group_dbg && msd && dbg_reports > 0;
  if (dbg && !dbg_buf) {
    static bool dattr = false;
    if (!dattr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel<u64, RB, BLOCK, KPT0, true, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS0::BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel<u32, RB, BLOCK, KPT1, true, false, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS1::BYTES);
      dattr = true;
    }
    if (hipMalloc(&dbg_buf, 2 * 64 * 8 * sizeof(u64)) != hipSuccess) dbg_buf = nullptr;
  }
  if (dbg && dbg_buf) MGC_CHECK(hipMemsetAsync(dbg_buf, 0, 2 * 64 * 8 * sizeof(u64), st));
  if (soa_hi_mask && !msd) return hipErrorInvalidValue;   // the 5-byte layout: high digit first
  GroupExtra ex_first{bA, low, (1u << bB) - 1u, &hdr->ghist[1][0], soa_hi_mask};
  if (stagger > 1u && soa_hi_mask && pipe) { ex_first.stagger_groups = stagger; ex_first.stagger_cycles = NARROW_TILE0_CYCLES / stagger; }
  const GroupExtra ex_second{0u, 0u, 0u, nullptr};
  if (dbg && dbg_buf && soa_hi_mask) {                     // the shipped first pass, instrumented: 5-byte layout, the fetch a whole tile ahead (or not)
    static bool dsattr = false;
    if (!dsattr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel<u64, RB, BLOCK, KPT0, true, true, true, true, 2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS0::BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel<u64, RB, BLOCK, KPT0, true, true, true, true, 0>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS0::BYTES);
      dsattr = true;
    }
    if (pipe)
      hipLaunchKernelGGL((radix_group_kernel<u64, RB, BLOCK, KPT0, true, true, true, true, 2>), grid0, dim3(BLOCK), GS0::BYTES, st,
                         reinterpret_cast<const u64 *>(d_keys), reinterpret_cast<u32 *>(d_alt), (u64)n, shA, (1u << bA) - 1u,
                         &hdr->gbase[0][0], status_a, &hdr->ticket[0], d_error, (u64)tiles0, (const u64 *)nullptr, (const u32 *)nullptr,
                         ex_first, dbg_buf);
    else
      hipLaunchKernelGGL((radix_group_kernel<u64, RB, BLOCK, KPT0, true, true, true, true, 0>), grid0, dim3(BLOCK), GS0::BYTES, st,
                         reinterpret_cast<const u64 *>(d_keys), reinterpret_cast<u32 *>(d_alt), (u64)n, shA, (1u << bA) - 1u,
                         &hdr->gbase[0][0], status_a, &hdr->ticket[0], d_error, (u64)tiles0, (const u64 *)nullptr, (const u32 *)nullptr,
                         ex_first, dbg_buf);
  }
  else if (dbg && dbg_buf)
    hipLaunchKernelGGL((radix_group_kernel<u64, RB, BLOCK, KPT0, true, true, true>), grid0, dim3(BLOCK), GS0::BYTES, st,
                       reinterpret_cast<const u64 *>(d_keys), reinterpret_cast<u32 *>(d_alt), (u64)n, shA, (1u << bA) - 1u,
                       &hdr->gbase[0][0], status_a, &hdr->ticket[0], d_error, (u64)tiles0, (const u64 *)nullptr, (const u32 *)nullptr,
                       GroupExtra{bA, low, (1u << bB) - 1u, &hdr->ghist[1][0]}, dbg_buf);
  else if (msd && soa_hi_mask) {
    static bool sattr = false;
    if (!sattr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel<u64, RB, BLOCK, KPT0, false, true, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS0::BYTES);
      sattr = true;
    }
    static bool spattr = false;
    if (pipe && !spattr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel<u64, RB, BLOCK, KPT0, false, true, true, true, 2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS0::BYTES);
      spattr = true;
    }
    if (pipe && bA <= 8u) {
      // an eight-bit first digit (the plan of the judged files): 256 counters, 128 walkers and look-back rows of 128 granules instead of
      // 512 / 256 / 256 -- half the status traffic, two more waves for the low digit's count
      using GS08 = GroupSmem<u64, 8, BLOCK, KPT0>;
      static bool r8attr = false;
      if (!r8attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel<u64, 8, BLOCK, KPT0, false, true, true, true, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS08::BYTES);
        r8attr = true;
      }
      hipLaunchKernelGGL((radix_group_kernel<u64, 8, BLOCK, KPT0, false, true, true, true, 2>), grid0, dim3(BLOCK), GS08::BYTES, st,
                         reinterpret_cast<const u64 *>(d_keys), reinterpret_cast<u32 *>(d_alt), (u64)n, shA, (1u << bA) - 1u,
                         &hdr->gbase[0][0], status_a, &hdr->ticket[0], d_error, (u64)tiles0, (const u64 *)nullptr, (const u32 *)nullptr,
                         ex_first, (u64 *)nullptr);
    }
    else if (pipe)
      hipLaunchKernelGGL((radix_group_kernel<u64, RB, BLOCK, KPT0, false, true, true, true, 2>), grid0, dim3(BLOCK), GS0::BYTES, st,
                         reinterpret_cast<const u64 *>(d_keys), reinterpret_cast<u32 *>(d_alt), (u64)n, shA, (1u << bA) - 1u,
                         &hdr->gbase[0][0], status_a, &hdr->ticket[0], d_error, (u64)tiles0, (const u64 *)nullptr, (const u32 *)nullptr,
                         ex_first, (u64 *)nullptr);
    else
    hipLaunchKernelGGL((radix_group_kernel<u64, RB, BLOCK, KPT0, false, true, true, true>), grid0, dim3(BLOCK), GS0::BYTES, st,
                       reinterpret_cast<const u64 *>(d_keys), reinterpret_cast<u32 *>(d_alt), (u64)n, shA, (1u << bA) - 1u,
                       &hdr->gbase[0][0], status_a, &hdr->ticket[0], d_error, (u64)tiles0, (const u64 *)nullptr, (const u32 *)nullptr,
                       ex_first, (u64 *)nullptr);
  }
  else if (msd)
    hipLaunchKernelGGL((radix_group_kernel<u64, RB, BLOCK, KPT0, false, true, true>), grid0, dim3(BLOCK), GS0::BYTES, st,
                       reinterpret_cast<const u64 *>(d_keys), reinterpret_cast<u32 *>(d_alt), (u64)n, shA, (1u << bA) - 1u,
                       &hdr->gbase[0][0], status_a, &hdr->ticket[0], d_error, (u64)tiles0, (const u64 *)nullptr, (const u32 *)nullptr,
                       GroupExtra{bA, low, (1u << bB) - 1u, &hdr->ghist[1][0]}, (u64 *)nullptr);
  else
    hipLaunchKernelGGL((radix_group_kernel<u64, RB, BLOCK, KPT0, false, true, false>), grid0, dim3(BLOCK), GS0::BYTES, st,
                       reinterpret_cast<const u64 *>(d_keys), reinterpret_cast<u32 *>(d_alt), (u64)n, shA, (1u << bA) - 1u,
                       &hdr->gbase[0][0], status_a, &hdr->ticket[0], d_error, (u64)tiles0, (const u64 *)nullptr, (const u32 *)nullptr,
                       GroupExtra{bA, 0u, 0u, nullptr}, (u64 *)nullptr);
  MGC_CHECK(hipGetLastError());
  if (pass_events) MGC_CHECK(hipEventRecord(pass_events[1], st));

  if (msd) {
    hipLaunchKernelGGL(narrow_mid_kernel, dim3(1), dim3(RS_MAX_RADIX), 0, st, hdr, (u64)n, (u32)tile1, region_start, region_tiles);
    MGC_CHECK(hipGetLastError());
  } else {
    hipLaunchKernelGGL(group_regions_kernel, dim3(1), dim3(RS_MAX_RADIX), 0, st, &hdr->gbase[0][0], (u64)n, (u32)TILE1, region_start, region_tiles);
    MGC_CHECK(hipGetLastError());
    MGC_CHECK(hipMemsetAsync(status_b, 0, (size_t)tiles1_max * (R / 2) * sizeof(u64), st));
  }
  if (pass_events) MGC_CHECK(hipEventRecord(pass_events[2], st));
  u32 granules1 = (u32)(R / 2);                             // granules per tile of the second pass's status rows (narrow_bounds_kernel)
  if (dbg && dbg_buf)
    hipLaunchKernelGGL((radix_group_kernel<u32, RB, BLOCK, KPT1, true, false, false>), dim3((uint32_t)std::min(tiles1_max, cus * GS1::WG_PER_CU)), dim3(BLOCK),
                       GS1::BYTES, st, reinterpret_cast<const u32 *>(d_alt), reinterpret_cast<u32 *>(d_keys), (u64)n, low, (1u << bB) - 1u,
                       &hdr->gbase[1][0], status_b, &hdr->ticket[1], d_error, (u64)((n + TILE1 - 1) / TILE1), region_start, region_tiles,
                       ex_second, dbg_buf + 64 * 8);
  else if (bB <= 8u) {                                      // an eight-bit digit: half the counters, walkers and granules (as in the first pass)
    using GS18 = GroupSmem<u32, 8, BLOCK, KPT1>;
    static bool r8attr1 = false;
    if (!r8attr1) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel<u32, 8, BLOCK, KPT1, false, false, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS18::BYTES);
      r8attr1 = true;
    }
    granules1 = 128u;
    hipLaunchKernelGGL((radix_group_kernel<u32, 8, BLOCK, KPT1, false, false, false>), dim3((uint32_t)std::min(tiles1_max, cus * GS18::WG_PER_CU)), dim3(BLOCK),
                       GS18::BYTES, st, reinterpret_cast<const u32 *>(d_alt), reinterpret_cast<u32 *>(d_keys), (u64)n, low, (1u << bB) - 1u,
                       &hdr->gbase[1][0], status_b, &hdr->ticket[1], d_error, (u64)((n + TILE1 - 1) / TILE1), region_start, region_tiles,
                       ex_second, (u64 *)nullptr);
  }
  else
  hipLaunchKernelGGL((radix_group_kernel<u32, RB, BLOCK, KPT1, false, false, false>), dim3((uint32_t)std::min(tiles1_max, cus * GS1::WG_PER_CU)), dim3(BLOCK),
                     GS1::BYTES, st, reinterpret_cast<const u32 *>(d_alt), reinterpret_cast<u32 *>(d_keys), (u64)n, low, (1u << bB) - 1u,
                     &hdr->gbase[1][0], status_b, &hdr->ticket[1], d_error, (u64)((n + TILE1 - 1) / TILE1), region_start, region_tiles,
                     ex_second, (u64 *)nullptr);
  MGC_CHECK(hipGetLastError());
  if (pass_events) MGC_CHECK(hipEventRecord(pass_events[3], st));

  const u64 ng = (u64)1 << (bA + bB);
  hipLaunchKernelGGL(narrow_bounds_kernel, dim3((uint32_t)((ng + 1 + 255) / 256)), dim3(256), 0, st, status_b, region_tiles,
                     &hdr->gbase[1][0], (u64)n, bA, ng, reinterpret_cast<u64 *>(d_sub_starts), granules1);
  MGC_CHECK(hipGetLastError());
  if (dbg && dbg_buf) {
    dbg_reports--;
    u64 h[2 * 64 * 8];
    MGC_CHECK(hipStreamSynchronize(st));
    MGC_CHECK(hipMemcpy(h, dbg_buf, sizeof(h), hipMemcpyDeviceToHost));
    for (int pass = 0; pass < 2; pass++) {
      double ps[8] = {0};
      for (int b = 0; b < 64; b++) for (int i = 0; i < 8; i++) ps[i] += (double)h[(pass * 64 + b) * 8 + i];
      const double it = ps[7] > 0 ? ps[7] : 1;
      fprintf(stderr, "[groupdbg] %s pass, %llu keys, %d-key tiles, tiles/wg=%.1f cycles/tile: ticket+zero=%.0f rank=%.0f scan+exchange=%.0f "
                      "lookback(+prefetch%s)=%.0f writeout=%.0f endsync=%.0f total=%.0f\n",
              pass ? "second (u32 -> u32)" : (soa_hi_mask ? (pipe ? "first (5 B -> u32, fetch a tile ahead)" : "first (5 B -> u32)") : "first (u64 -> u32)"), (unsigned long long)n, pass ? (int)TILE1 : (int)TILE0, it / 64,
              ps[0] / it, ps[1] / it, ps[2] / it, pass ? "" : ", low-digit count", ps[3] / it, ps[4] / it, ps[5] / it,
              (ps[0] + ps[1] + ps[2] + ps[3] + ps[4] + ps[5]) / it);
    }
  }
  return hipSuccess;
}

// ---- the same two passes for WHOLE keys (mgc_device.h, launch_group_wide) ----
bool sort_plan_wide_msd(const SortPlan &plan, uint64_t n, bool on) {
  return on && plan.mode == 3 && plan.num_passes == 2 && plan.radix_bits == 9 && n > 0 && n < (1ull << 30) &&
         plan.pass_shift[1] == plan.pass_shift[0] + plan.pass_bits[0];
}

// one workgroup per bucket: header cleared, histogram of the bucket's high dense-rank digit from the dense-rank table, its scan
struct HpcPrep { u32 bucket_bits; u64 on[4]; };
__global__ __launch_bounds__(RS_MAX_RADIX)
void hpc_prepare_kernel(const u64 *__restrict__ fine, HpcPrep prep, unsigned char *__restrict__ hdrs, u32 hdr_stride) {
  __shared__ u64 s_tmp[RS_MAX_RADIX / 64 + 1];
  const u32 b = blockIdx.x, x = threadIdx.x;
  if (!((prep.on[b >> 6] >> (b & 63u)) & 1ull)) return;
  SortHeader *hdr = reinterpret_cast<SortHeader *>(hdrs + (size_t)b * hdr_stride);
  u32 *w = reinterpret_cast<u32 *>(hdr);
  for (u32 i = x; i < sizeof(SortHeader) / 4; i += RS_MAX_RADIX) w[i] = 0;
  const u32 bb = prep.bucket_bits / 2;                       // dense rank of the bucket's bases (a bucket with k-mers repeats none)
  u32 prev = (b >> (2 * bb - 2)) & 3u, r = prev;
  for (u32 i = 1; i < bb; i++) { const u32 c = (b >> (2 * (bb - 1 - i))) & 3u; r = r * 3u + (c - (c > prev ? 1u : 0u)); prev = c; }
  const u64 c = (x < 243u) ? fine[(size_t)r * 243u + x] : 0ull;
  __syncthreads();                                           // (the header is cleared)
  u64 total;
  const u64 e = block_excl_scan<RS_MAX_RADIX, u64>(c, s_tmp, &total);
  hdr->ghist[0][x] = c;
  hdr->gbase[0][x] = e;
}

hipError_t launch_hpc_prepare(const uint64_t *d_fine_hpc, uint32_t bucket_bits, const uint64_t on[4], void *d_hdrs, hipStream_t st) {
  if (bucket_bits != 6 && bucket_bits != 8) return hipErrorInvalidValue;
  HpcPrep prep;
  prep.bucket_bits = bucket_bits;
  for (int i = 0; i < 4; i++) prep.on[i] = on[i];
  hipLaunchKernelGGL(hpc_prepare_kernel, dim3(1u << bucket_bits), dim3(RS_MAX_RADIX), 0, st, reinterpret_cast<const u64 *>(d_fine_hpc), prep,
                     reinterpret_cast<unsigned char *>(d_hdrs), (u32)sort_header_bytes());
  return hipGetLastError();
}

// (the look-back scratch is sized for the SMALLEST tile of any whole-key instantiation of a key width: K128 8192 keys, K96 12288 --
// group_wide asserts that its tile is not smaller)
static inline uint64_t wide_tile(uint32_t key_words) { return key_words == 2 ? 1024u * 8u : 1024u * 16u; }
size_t wide_scratch_bytes(uint64_t n, uint32_t key_words) {
  const uint64_t tile = wide_tile(key_words);
  const uint64_t tiles0 = (n + tile - 1) / tile, tiles1_max = tiles0 + RS_MAX_RADIX + 1;
  return (size_t)(tiles0 + tiles1_max) * (RS_MAX_RADIX / 2) * sizeof(u64) + (size_t)(RS_MAX_RADIX + 2) * 16 + 512;
}

// RB: 9, or 8 where both digits have at most eight bits (plain bit digits only): half the counters, walkers and look-back granules
template <typename K, int KPT, int RB>
static hipError_t group_wide_rb(void *d_keys, void *d_alt, uint64_t n, const SortPlan &plan, uint32_t *d_error, uint64_t *d_sub_starts,
                                hipStream_t st, hipEvent_t *pass_events, void *d_prepared, void *d_scratch, uint32_t *tr_a, uint32_t *tr_b) {
  constexpr int BLOCK = 1024, R = 1 << RB;
  using GS = GroupSmem<K, RB, BLOCK, KPT>;
  constexpr uint64_t TILE = (uint64_t)BLOCK * KPT;
  static_assert(TILE >= (sizeof(K) >= 12 ? 1024u * 8u : 1024u * 16u), "wide_scratch_bytes sizes the granules for tiles of at least wide_tile() keys");
  // `compress` (dense-rank digits): the instantiations with the rank table in LDS -- where 8 KiB are left behind the tile
  constexpr bool HPC_TAB = RB == 9 && GS::BYTES + 8192 <= 160 * 1024;
  if (RB != 9 && (plan.hpc || plan.pass_bits[0] > (u32)RB || plan.pass_bits[1] > (u32)RB)) return hipErrorInvalidValue;
  const bool hpcd = HPC_TAB && plan.hpc && plan_mask(plan, 0) == HPC_DIGIT_MASK && plan_mask(plan, 1) == HPC_DIGIT_MASK;
  // (make_hpc_mixed_plan: the high digit from the rank table, the low one a plain eight-bit field -- its pass is the eight-bit instantiation)
  const bool mixed = HPC_TAB && plan.hpc == 2 && plan.pass_bits[0] <= 8u;
  using GS8 = GroupSmem<K, 8, BLOCK, KPT>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel<K, RB, BLOCK, KPT, false, false, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS::BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel<K, RB, BLOCK, KPT, false, false, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS::BYTES);
    if constexpr (HPC_TAB) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel<K, RB, BLOCK, KPT, false, false, true, false, 0, 1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS::BYTES + 8192);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel<K, RB, BLOCK, KPT, false, false, false, false, 0, 1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS::BYTES + 8192);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel<K, RB, BLOCK, KPT, false, false, true, false, 0, 2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS::BYTES + 8192);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel<K, 8, BLOCK, KPT, false, false, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS8::BYTES);
    }
    attr_done = true;
  }
  const uint64_t tiles0 = (n + TILE - 1) / TILE, tiles1_max = tiles0 + RS_MAX_RADIX + 1;
  SortHeader *hdr = reinterpret_cast<SortHeader *>(d_prepared);
  u64 *status_a = reinterpret_cast<u64 *>(d_scratch);
  u64 *status_b = status_a + (size_t)tiles0 * (R / 2);
  u64 *region_start = status_b + (size_t)tiles1_max * (R / 2);
  u32 *region_tiles = reinterpret_cast<u32 *>(region_start + RS_MAX_RADIX + 1);
  // the keys stay whole: digit A (high) at shA, digit B (low) at `low`, both where the plan put them
  // (`compress`: dense-rank digits in ten-bit fields -- the kernels take the digit with hpc_digit(), the sub-bucket numbers keep the fields)
  const u32 low = plan.pass_shift[0], bB = plan.pass_bits[0], bA = plan.pass_bits[1], shA = low + bB;
  const u32 maskA = plan_mask(plan, 1), maskB = plan_mask(plan, 0);
  *tr_a = bA; *tr_b = bB;
  const uint64_t resident = (uint64_t)device_cu_count() * GS::WG_PER_CU;

  if (pass_events) MGC_CHECK(hipEventRecord(pass_events[0], st));
  if constexpr (HPC_TAB) {
    if (hpcd)
      hipLaunchKernelGGL((radix_group_kernel<K, RB, BLOCK, KPT, false, false, true, false, 0, 1>), dim3((uint32_t)std::min(tiles0, resident)), dim3(BLOCK), GS::BYTES + 8192, st,
                         reinterpret_cast<const K *>(d_keys), reinterpret_cast<K *>(d_alt), (u64)n, shA, maskA,
                         &hdr->gbase[0][0], status_a, &hdr->ticket[0], d_error, (u64)tiles0, (const u64 *)nullptr, (const u32 *)nullptr,
                         GroupExtra{0u, low, maskB, &hdr->ghist[1][0]}, (u64 *)nullptr);
  }
  if constexpr (HPC_TAB) {
    if (mixed)
      hipLaunchKernelGGL((radix_group_kernel<K, RB, BLOCK, KPT, false, false, true, false, 0, 2>), dim3((uint32_t)std::min(tiles0, resident)), dim3(BLOCK), GS::BYTES + 8192, st,
                         reinterpret_cast<const K *>(d_keys), reinterpret_cast<K *>(d_alt), (u64)n, shA, maskA,
                         &hdr->gbase[0][0], status_a, &hdr->ticket[0], d_error, (u64)tiles0, (const u64 *)nullptr, (const u32 *)nullptr,
                         GroupExtra{0u, low, maskB, &hdr->ghist[1][0]}, (u64 *)nullptr);
  }
  if (!hpcd && !mixed)
  hipLaunchKernelGGL((radix_group_kernel<K, RB, BLOCK, KPT, false, false, true>), dim3((uint32_t)std::min(tiles0, resident)), dim3(BLOCK), GS::BYTES, st,
                     reinterpret_cast<const K *>(d_keys), reinterpret_cast<K *>(d_alt), (u64)n, shA, maskA,
                     &hdr->gbase[0][0], status_a, &hdr->ticket[0], d_error, (u64)tiles0, (const u64 *)nullptr, (const u32 *)nullptr,
                     GroupExtra{0u, low, maskB, &hdr->ghist[1][0]}, (u64 *)nullptr);
  MGC_CHECK(hipGetLastError());
  if (pass_events) MGC_CHECK(hipEventRecord(pass_events[1], st));
  hipLaunchKernelGGL(narrow_mid_kernel, dim3(1), dim3(RS_MAX_RADIX), 0, st, hdr, (u64)n, (u32)TILE, region_start, region_tiles);
  MGC_CHECK(hipGetLastError());
  if (pass_events) MGC_CHECK(hipEventRecord(pass_events[2], st));
  if constexpr (HPC_TAB) {
    if (hpcd)
      hipLaunchKernelGGL((radix_group_kernel<K, RB, BLOCK, KPT, false, false, false, false, 0, 1>), dim3((uint32_t)std::min(tiles1_max, resident)), dim3(BLOCK), GS::BYTES + 8192, st,
                         reinterpret_cast<const K *>(d_alt), reinterpret_cast<K *>(d_keys), (u64)n, low, maskB,
                         &hdr->gbase[1][0], status_b, &hdr->ticket[1], d_error, (u64)tiles0, region_start, region_tiles,
                         GroupExtra{0u, 0u, 0u, nullptr}, (u64 *)nullptr);
  }
  u32 granules1 = (u32)(R / 2);
  if constexpr (HPC_TAB) {
    if (mixed) {
      granules1 = 128u;
      hipLaunchKernelGGL((radix_group_kernel<K, 8, BLOCK, KPT, false, false, false>), dim3((uint32_t)std::min(tiles1_max, resident)), dim3(BLOCK), GS8::BYTES, st,
                         reinterpret_cast<const K *>(d_alt), reinterpret_cast<K *>(d_keys), (u64)n, low, maskB,
                         &hdr->gbase[1][0], status_b, &hdr->ticket[1], d_error, (u64)tiles0, region_start, region_tiles,
                         GroupExtra{0u, 0u, 0u, nullptr}, (u64 *)nullptr);
    }
  }
  if (!hpcd && !mixed)
  hipLaunchKernelGGL((radix_group_kernel<K, RB, BLOCK, KPT, false, false, false>), dim3((uint32_t)std::min(tiles1_max, resident)), dim3(BLOCK), GS::BYTES, st,
                     reinterpret_cast<const K *>(d_alt), reinterpret_cast<K *>(d_keys), (u64)n, low, maskB,
                     &hdr->gbase[1][0], status_b, &hdr->ticket[1], d_error, (u64)tiles0, region_start, region_tiles,
                     GroupExtra{0u, 0u, 0u, nullptr}, (u64 *)nullptr);
  MGC_CHECK(hipGetLastError());
  if (pass_events) MGC_CHECK(hipEventRecord(pass_events[3], st));
  const u64 ng = (u64)1 << (bA + bB);
  hipLaunchKernelGGL(narrow_bounds_kernel, dim3((uint32_t)((ng + 1 + 255) / 256)), dim3(256), 0, st, status_b, region_tiles,
                     &hdr->gbase[1][0], (u64)n, bA, ng, reinterpret_cast<u64 *>(d_sub_starts), granules1);
  return hipGetLastError();
}

template <typename K, int KPT>
static hipError_t group_wide(void *d_keys, void *d_alt, uint64_t n, const SortPlan &plan, uint32_t *d_error, uint64_t *d_sub_starts,
                             hipStream_t st, hipEvent_t *pass_events, void *d_prepared, void *d_scratch, uint32_t *tr_a, uint32_t *tr_b) {
  if (!plan.hpc && plan.pass_bits[0] <= 8u && plan.pass_bits[1] <= 8u)
    return group_wide_rb<K, KPT, 8>(d_keys, d_alt, n, plan, d_error, d_sub_starts, st, pass_events, d_prepared, d_scratch, tr_a, tr_b);
  return group_wide_rb<K, KPT, 9>(d_keys, d_alt, n, plan, d_error, d_sub_starts, st, pass_events, d_prepared, d_scratch, tr_a, tr_b);
}

hipError_t launch_group_wide(void *d_keys, void *d_alt, uint64_t n, uint32_t key_words, const SortPlan &plan, uint32_t *d_error,
                             uint64_t *d_sub_starts, hipStream_t st, hipEvent_t *pass_events, void *d_prepared, void *d_scratch,
                             uint32_t *tr_a, uint32_t *tr_b, bool k96) {
  if (!sort_plan_wide_msd(plan, n) || !d_prepared || !d_scratch) return hipErrorInvalidValue;
  // 12-byte K96 records (k = 33..51, the bits below the file): 12288-key tiles (144 KiB of LDS)
  if (k96) return key_words == 2 ? group_wide<K96, 12>(d_keys, d_alt, n, plan, d_error, d_sub_starts, st, pass_events, d_prepared, d_scratch, tr_a, tr_b)
                                 : hipErrorInvalidValue;
  if (key_words == 2) return group_wide<K128, 8>(d_keys, d_alt, n, plan, d_error, d_sub_starts, st, pass_events, d_prepared, d_scratch, tr_a, tr_b);
  return group_wide<u64, 16>(d_keys, d_alt, n, plan, d_error, d_sub_starts, st, pass_events, d_prepared, d_scratch, tr_a, tr_b);
}

size_t sort_header_bytes() { return ((sizeof(SortHeader) + 255) / 256) * 256; }

hipError_t launch_radix_sort(void *d_keys, void *d_alt, uint64_t n, uint32_t key_words, const SortPlan &plan,
                             void *d_ws, size_t ws_bytes, uint32_t *d_error, int *result_in_alt,
                             hipStream_t st, hipEvent_t *pass_events) {
  *result_in_alt = 0;
  if (n == 0 || plan.num_passes == 0) return hipSuccess;
  if (ws_bytes < sort_workspace_bytes(n)) return hipErrorInvalidValue;
  if (plan.hpc && !(plan.mode == 3 && plan.num_passes <= 2 && n < (1ull << 30))) return hipErrorInvalidValue;
  if (plan.mode == 3 && (plan.num_passes > 2 || n >= (1ull << 30))) {
    SortPlan stable = plan;                 // grouping is defined for one or two digits and 30-bit granule values
    stable.mode = 0;
    return launch_radix_sort(d_keys, d_alt, n, key_words, stable, d_ws, ws_bytes, d_error, result_in_alt, st, pass_events);
  }
  // packed look-back granules hold 30-bit values: larger calls use the wide ones; 16-byte keys: 8 per thread keep the tile at 128 KiB
  if (key_words == 2) {
    if (n >= (1ull << 30)) return run_passes<K128, 8, 3>(d_keys, d_alt, n, plan, d_ws, d_error, result_in_alt, st, pass_events);
    return run_passes<K128, 8, 2>(d_keys, d_alt, n, plan, d_ws, d_error, result_in_alt, st, pass_events);
  }
  if (n >= (1ull << 30)) return run_passes<u64, 16, 3>(d_keys, d_alt, n, plan, d_ws, d_error, result_in_alt, st, pass_events);
  return run_passes<u64, 16, 2>(d_keys, d_alt, n, plan, d_ws, d_error, result_in_alt, st, pass_events);
}



hipError_t warm_sort() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&narrow_mid_kernel)); }

}  // namespace mgc
