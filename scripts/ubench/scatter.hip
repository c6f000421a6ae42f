// What the memory system makes of a radix pass's ACCESS PATTERN, with no ranking or LDS work in the way: 256 persistent
// 1024-thread workgroups read 8-byte keys streaming (16 per thread and tile, like radix_group_kernel) and write 4-byte (or
// 8-byte) words as 512 contiguous runs per tile -- run d of tile t goes to region d at offset t * L -- i.e. exactly the
// scatter of a pass over uniformly distributed digits.  Sweeps the run length L (words per digit per tile) by changing the
// tile size; prints GB/s of (read + written) bytes.  Also: reads only, the same bytes written streaming, and runs that start
// unaligned (the real pass's cursors are arbitrary).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/scatter scripts/ubench/scatter.hip && /tmp/scatter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;

// FLAGS (round 4): 1 = a workgroup takes CONSECUTIVE tiles (the chunk-local pass's order: both halves of a straddled line come
// from the same CU, one tile apart) instead of tiles blockIdx.x, blockIdx.x + grid, ...; 2 = nontemporal loads of the input
// stream (it should not push the open output lines out of the XCD's L2); 4 = nontemporal stores.
// ALIGN (round 5; MODE 1 only): a run starts at a multiple of ALIGN words (4 * ALIGN bytes) instead of anywhere -- which start
// alignment does the memory system need to stop charging for the two partial lines of a run?
template <typename OUT, int KPT, int MODE, int FLAGS = 0, int ALIGN = 1>   // MODE 0 scatter aligned, 1 scatter with odd run starts, 2 streaming writes, 3 reads only
__global__ __launch_bounds__(1024) void pass_kernel(const u64 *__restrict__ in, OUT *__restrict__ out, u64 n, u64 region, u32 skew) {
  constexpr u32 TILE = 1024 * KPT, L = TILE / 512;
  const u64 tiles = n / TILE;
  u64 acc = 0;
  const u64 per = (tiles + gridDim.x - 1) / gridDim.x;
  const u64 t_begin = (FLAGS & 1) ? blockIdx.x * per : blockIdx.x, t_step = (FLAGS & 1) ? 1 : gridDim.x;
  const u64 t_end = (FLAGS & 1) ? (t_begin + per < tiles ? t_begin + per : tiles) : tiles;
  // FLAGS 8 (round 5): XCD-grouped tile order.  Workgroups are dispatched to the 8 XCDs round-robin (XCD = blockIdx.x % 8); here
  // XCD x takes the x-th eighth of the tiles and its 32 workgroups take 32 CONSECUTIVE tiles at a time, so that both halves of
  // a line two adjacent tiles share reach the SAME L2 within the skew between neighbouring CUs -- does that L2 merge them?
  const u32 xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, per_x = gridDim.x >> 3;
  const u64 seg = tiles / 8, rounds = (FLAGS & 8) ? (seg + per_x - 1) / per_x : 0;
  for (u64 it = 0, t = t_begin; (FLAGS & 8) ? it < rounds : t < t_end; it++, t += t_step) {
    if (FLAGS & 8) { const u64 o = it * per_x + slot; if (o >= seg) break; t = xcd * seg + o; }
    u64 k[KPT];
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const u64 *p = in + t * TILE + (u64)j * 1024 + threadIdx.x;
      k[j] = (FLAGS & 2) ? __builtin_nontemporal_load(p) : *p;
    }
    if (MODE == 3) {
#pragma unroll
      for (int j = 0; j < KPT; j++) acc ^= k[j];
      continue;
    }
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const u32 i = (u32)j * 1024 + threadIdx.x;       // position in the (conceptually sorted) tile
      const u32 d = i / L, w = i % L;
      u64 pos;
      if (MODE == 2) pos = t * TILE + i;
      else           pos = (u64)d * region + t * L + w + (MODE == 1 ? (ALIGN == 1 ? (u64)(d * skew) % 29 : (u64)(((d * skew) % (32u / ALIGN)) * ALIGN)) : 0);
      if (FLAGS & 4) __builtin_nontemporal_store((OUT)k[j], out + pos);
      else           out[pos] = (OUT)k[j];
    }
  }
  if (MODE == 3 && acc == 0x1234567) out[0] = (OUT)acc;
}

template <typename OUT, int KPT, int MODE, int FLAGS = 0, int ALIGN = 1>
static void run(const char *what, const u64 *in, void *out, u64 n) {
  constexpr u32 TILE = 1024 * KPT;
  const u64 tiles = n / TILE, region = tiles * (TILE / 512) + 64;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  pass_kernel<OUT, KPT, MODE, FLAGS, ALIGN><<<256, 1024>>>(in, (OUT *)out, n, region, 7);
  hipEventRecord(a);
  for (int r = 0; r < 5; r++) pass_kernel<OUT, KPT, MODE, FLAGS, ALIGN><<<256, 1024>>>(in, (OUT *)out, n, region, 7);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  ms /= 5;
  const double bytes = (double)tiles * TILE * (8.0 + (MODE == 3 ? 0.0 : sizeof(OUT)));
  printf("%-58s tile %6u keys, run %4zu B: %.3f ms  %.2f TB/s\n", what, TILE, (size_t)(TILE / 512) * sizeof(OUT), ms, bytes / ms / 1e9);
}

int main() {
  const u64 n = 135ull << 20;                       // one file of the 10 Gbp workload
  u64 *in; void *out;
  hipMalloc(&in, n * 8); hipMalloc(&out, n * 8 + (1 << 24));
  hipMemset(in, 1, n * 8); hipMemset(out, 0, n * 8);
  run<u32, 16, 3>("reads only (8 B keys)", in, out, n);
  run<u32, 16, 2>("8 B in, 4 B out, streaming writes", in, out, n);
  run<u32, 8, 0>("8 B in, 4 B out, 512 runs per tile", in, out, n);
  run<u32, 16, 0>("8 B in, 4 B out, 512 runs per tile", in, out, n);
  run<u32, 16, 1>("8 B in, 4 B out, 512 runs per tile, odd run starts", in, out, n);
  run<u32, 16, 1, 1>("  odd starts, consecutive tiles per workgroup", in, out, n);
  run<u32, 16, 1, 2>("  odd starts, nt loads", in, out, n);
  run<u32, 16, 1, 3>("  odd starts, consecutive tiles + nt loads", in, out, n);
  run<u32, 16, 1, 4>("  odd starts, nt stores", in, out, n);
  run<u32, 16, 1, 6>("  odd starts, nt loads + nt stores", in, out, n);
  run<u32, 16, 1, 7>("  odd starts, consecutive tiles + nt loads + nt stores", in, out, n);
  run<u32, 16, 0, 2>("  aligned, nt loads", in, out, n);
  run<u32, 16, 0, 6>("  aligned, nt loads + nt stores", in, out, n);
  run<u32, 32, 0>("8 B in, 4 B out, 512 runs per tile", in, out, n);
  run<u32, 32, 1>("8 B in, 4 B out, 512 runs per tile, odd run starts", in, out, n);
  run<u32, 64, 0>("8 B in, 4 B out, 512 runs per tile", in, out, n);
  run<u64, 16, 0>("8 B in, 8 B out, 512 runs per tile", in, out, n);
  run<u64, 16, 1>("8 B in, 8 B out, 512 runs per tile, odd run starts", in, out, n);
  run<u64, 32, 0>("8 B in, 8 B out, 512 runs per tile", in, out, n);
  run<u64, 4, 0>("8 B in, 8 B out, 512 runs per tile (the partition's 64 B)", in, out, n);
  // ---- round 5: which run-START alignment removes the odd-start tax?  (offsets: multiples of ALIGN words, different per digit) ----
  printf("-- run-start alignment sweep, 128-byte runs (16384-key tiles), 135 M keys\n");
  run<u32, 16, 1, 0, 1>("  start aligned to   4 B", in, out, n);
  run<u32, 16, 1, 0, 2>("  start aligned to   8 B", in, out, n);
  run<u32, 16, 1, 0, 4>("  start aligned to  16 B", in, out, n);
  run<u32, 16, 1, 0, 8>("  start aligned to  32 B", in, out, n);
  run<u32, 16, 1, 0, 16>("  start aligned to  64 B", in, out, n);
  run<u32, 16, 0>("  start aligned to 128 B", in, out, n);
  printf("-- run-start alignment sweep, 256-byte runs (32768-key tiles), 135 M keys\n");
  run<u32, 32, 1, 0, 1>("  start aligned to   4 B", in, out, n);
  run<u32, 32, 1, 0, 2>("  start aligned to   8 B", in, out, n);
  run<u32, 32, 1, 0, 4>("  start aligned to  16 B", in, out, n);
  run<u32, 32, 1, 0, 8>("  start aligned to  32 B", in, out, n);
  run<u32, 32, 1, 0, 16>("  start aligned to  64 B", in, out, n);
  run<u32, 32, 0>("  start aligned to 128 B", in, out, n);
  printf("-- XCD-grouped tile order (an XCD's 32 workgroups write 32 consecutive tiles at a time), 135 M keys\n");
  run<u32, 16, 1, 8>("  odd starts, XCD-grouped", in, out, n);
  run<u32, 16, 1, 10>("  odd starts, XCD-grouped + nt loads", in, out, n);
  run<u32, 16, 0, 8>("  aligned starts, XCD-grouped", in, out, n);
  run<u32, 16, 1, 8, 8>("  32-B-aligned starts, XCD-grouped", in, out, n);
  run<u32, 32, 1, 8>("  odd starts, XCD-grouped, 256-byte runs", in, out, n);
  run<u64, 16, 1, 8>("  8 B out, odd starts, XCD-grouped", in, out, n);
  // ---- round 5: does the tax survive when the OUTPUT of a launch fits the 256 MiB Infinity Cache? ----
  for (u64 mk : {16ull, 32ull, 64ull, 135ull}) {
    const u64 nn = mk << 20;
    printf("-- %llu M keys: %llu MB in, %llu MB out\n", mk, (nn * 8) >> 20, (nn * 4) >> 20);
    run<u32, 16, 0>("  aligned starts", in, out, nn);
    run<u32, 16, 1>("  odd starts", in, out, nn);
    run<u32, 16, 1, 0, 8>("  32-B-aligned starts", in, out, nn);
    run<u32, 16, 2>("  streaming writes", in, out, nn);
  }
  return 0;
}
