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
template <typename OUT, int KPT, int MODE, int FLAGS = 0>   // MODE 0 scatter aligned, 1 scatter with odd run starts, 2 streaming writes, 3 reads only
__global__ __launch_bounds__(1024) void pass_kernel(const u64 *__restrict__ in, OUT *__restrict__ out, u64 n, u64 region, u32 skew) {
  constexpr u32 TILE = 1024 * KPT, L = TILE / 512;
  const u64 tiles = n / TILE;
  u64 acc = 0;
  const u64 per = (tiles + gridDim.x - 1) / gridDim.x;
  const u64 t_begin = (FLAGS & 1) ? blockIdx.x * per : blockIdx.x, t_step = (FLAGS & 1) ? 1 : gridDim.x;
  const u64 t_end = (FLAGS & 1) ? (t_begin + per < tiles ? t_begin + per : tiles) : tiles;
  for (u64 t = t_begin; t < t_end; t += t_step) {
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
      else           pos = (u64)d * region + t * L + w + (MODE == 1 ? (u64)(d * skew) % 29 : 0);
      if (FLAGS & 4) __builtin_nontemporal_store((OUT)k[j], out + pos);
      else           out[pos] = (OUT)k[j];
    }
  }
  if (MODE == 3 && acc == 0x1234567) out[0] = (OUT)acc;
}

template <typename OUT, int KPT, int MODE, int FLAGS = 0>
static void run(const char *what, const u64 *in, void *out, u64 n) {
  constexpr u32 TILE = 1024 * KPT;
  const u64 tiles = n / TILE, region = tiles * (TILE / 512) + 64;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  pass_kernel<OUT, KPT, MODE, FLAGS><<<256, 1024>>>(in, (OUT *)out, n, region, 7);
  hipEventRecord(a);
  for (int r = 0; r < 5; r++) pass_kernel<OUT, KPT, MODE, FLAGS><<<256, 1024>>>(in, (OUT *)out, n, region, 7);
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
  return 0;
}
