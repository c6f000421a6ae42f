// What a CU's LDS makes of a grouping pass's PER-TILE work with no global memory in the way (four first-pass
// structures with a second resident workgroup, or with every store a whole line, all ended within 5 % of the look-back kernel --
// the suspect left is the ranking / exchange itself).  Persistent 1024-thread workgroups, 16 keys per thread and tile made up
// from a counter hash (a 9-bit digit + an 18-bit rest), per tile exactly the LDS phases of radix_group5_kernel:
//   clear 512 counters | rank: one returning atomic per key | scan | exchange: s_dbase[digit] read + scattered word write
//   (+ H2: the second digit's atomic) | read-out: every word read once, coalesced.
// Sweeps: which phases run, random against conflict-free digits, one against two workgroups per CU (the dynamic LDS size decides).
// Prints cycles per tile and the time 135 M keys (one file of the judged workload: 32.2 tiles per CU) would take.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/rank scripts/ubench/rank.hip && /tmp/rank
// Written at the end of round 4, after the GPU budget was spent: compiled, NOT RUN yet (at the 64-VGPR bound three of the
// instantiations spill 8-12 dwords: keep an eye on that when reading their numbers).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int BLOCK = 1024, KPT = 16, TILE = BLOCK * KPT, R = 512;

__device__ __forceinline__ u32 mix(u32 x) {            // (cheap and good enough: the digits only have to look random to the banks)
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// PHASES bit 0: rank, 1: scan + exchange, 2: second-digit atomics in the exchange, 3: read-out.  SEQ: digit = position / 32
// (every wave instruction hits 2 counters: no bank conflicts, heavy same-address traffic) instead of a random digit.
template <int PHASES, bool SEQ>
__global__ __launch_bounds__(BLOCK, 8) void rank_kernel   // (64 VGPRs: two workgroups must fit a CU)
(u32 tiles_per_wg, u32 *__restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32 *s_words = reinterpret_cast<u32 *>(smem);
  u32 *s_hist  = s_words + TILE;
  u32 *s_dbase = s_hist + R;
  u32 *s_h2    = s_dbase + R;
  u32 *s_tmp   = s_h2 + R;
  const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
  if (tid < (u32)R) s_h2[tid] = 0;
  u32 acc = 0;
  for (u32 t = 0; t < tiles_per_wg; t++) {
    u32 key[KPT];
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const u32 i = w * (u32)(64 * KPT) + ((u32)(j / 4) * 64u + lane) * 4u + (u32)(j % 4);
      key[j] = SEQ ? (((i >> 5) << 18) | (mix(i + t) & 0x3FFFFu)) : (mix((blockIdx.x * tiles_per_wg + t) * (u32)TILE + i) & 0x7FFFFFFu);
    }
    if (tid < (u32)R) s_hist[tid] = 0;
    __syncthreads();
    u32 ranks[KPT / 2];
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      u32 r = 0;
      if (PHASES & 1) r = atomicAdd(&s_hist[key[j] >> 18], 1u);
      if (j & 1) ranks[j / 2] |= r << 16;
      else       ranks[j / 2]  = r;
    }
    __syncthreads();
    if (PHASES & 2) {
      // (exclusive scan of the 512 counts: wave scans + one cross-wave step, as block_excl_scan does)
      u32 c = (tid < (u32)R) ? s_hist[tid] : 0u, x = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const u32 y = __shfl_up(x, o); if (lane >= (u32)o) x += y; }
      if (lane == 63u) s_tmp[w] = x;
      __syncthreads();
      u32 base = 0;
      for (u32 q = 0; q < w; q++) base += s_tmp[q];
      if (tid < (u32)R) s_dbase[tid] = base + x - c;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < KPT; j++) {
        const u32 r = (j & 1) ? (ranks[j / 2] >> 16) : (ranks[j / 2] & 0xFFFFu);
        const u32 pos = (PHASES & 1) ? s_dbase[key[j] >> 18] + r : (w * (u32)(64 * KPT) + (u32)j * 64u + lane);
        s_words[pos & (u32)(TILE - 1)] = key[j] & 0x3FFFFu;
        if (PHASES & 4) atomicAdd(&s_h2[(key[j] >> 9) & 511u], 1u);
      }
      __syncthreads();
    } else {
#pragma unroll
      for (int j = 0; j < KPT / 2; j++) acc += ranks[j];
    }
    if (PHASES & 8) {
#pragma unroll
      for (int j = 0; j < KPT; j++) acc += s_words[(u32)j * BLOCK + tid];
      __syncthreads();
    }
  }
  if (tid < (u32)R) acc += s_h2[tid];
  if (acc == 0x12345u) sink[0] = acc;
}

template <int PHASES, bool SEQ>
static void run(const char *what, int wg_per_cu, u32 *sink) {
  const size_t need = (size_t)(TILE + 3 * R + 64) * 4;
  const size_t lds = wg_per_cu == 1 ? 140 * 1024 : need;          // 140 KiB: nothing else fits on the CU; 76 KiB: two do
  hipFuncSetAttribute(reinterpret_cast<const void *>(&rank_kernel<PHASES, SEQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const u32 tiles_total = 256u * 64u, grid = 256u * (u32)wg_per_cu, per = tiles_total / grid;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  rank_kernel<PHASES, SEQ><<<grid, BLOCK, lds>>>(per, sink);
  hipEventRecord(a);
  for (int r = 0; r < 5; r++) rank_kernel<PHASES, SEQ><<<grid, BLOCK, lds>>>(per, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  ms /= 5;
  const double per_tile_us = ms * 1e3 / 64.0;                     // 64 tiles per CU in every configuration
  printf("%-64s %d wg/CU: %7.3f us per tile and CU  -> %.3f ms per 135 M keys\n", what, wg_per_cu, per_tile_us, per_tile_us * 32.2 / 1e3);
  hipEventDestroy(a); hipEventDestroy(b);
}

int main() {
  u32 *sink = nullptr;
  hipMalloc(&sink, 256);
  for (int wg = 1; wg <= 2; wg++) {
    run<1, false>("rank only (one returning atomic per key), random digits", wg, sink);
    run<1, true >("rank only, conflict-free digits", wg, sink);
    run<3, false>("rank + scan + exchange, random digits", wg, sink);
    run<3, true >("rank + scan + exchange, conflict-free digits", wg, sink);
    run<7, false>("rank + scan + exchange + second-digit atomics, random digits", wg, sink);
    run<11, false>("rank + scan + exchange + read-out, random digits", wg, sink);
    run<15, false>("all phases, random digits", wg, sink);
    run<10, false>("exchange by position + read-out only (no atomics)", wg, sink);
  }
  hipFree(sink);
  return 0;
}
