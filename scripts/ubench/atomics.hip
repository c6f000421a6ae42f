// Micro-benchmarks behind DESIGN.md's look-back discussion (developer tool, not part of the library):
//   1. returning global atomics on R contended cursors from every CU at once
//   2. latency of an agent-scope (sc1) load chain
// hipcc --offload-arch=gfx950 -O3 -o /tmp/ub scripts/ubench/atomics.hip && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;

__global__ void k_atomics(u32 *cur, u32 R, int iters, u64 *sink, int spread) {
  u64 acc = 0;
  for (int i = 0; i < iters; i++) {
    const u32 slot = spread ? ((threadIdx.x + i * 37u) % R) : (threadIdx.x % R);
    acc += atomicAdd(&cur[slot], 1u);
  }
  if (acc == 0x12345) sink[0] = acc;
}
__global__ void k_atomics64(u64 *cur, u32 R, int iters, u64 *sink) {
  u64 acc = 0;
  for (int i = 0; i < iters; i++) acc += atomicAdd(&cur[threadIdx.x % R], 1ull);
  if (acc == 0x12345) sink[0] = acc;
}
__global__ void k_chase(u64 *buf, u64 n, int hops, u64 *out) {
  u64 p = 0;
  const u64 t0 = __builtin_readcyclecounter();
  for (int i = 0; i < hops; i++) p = __hip_atomic_load(&buf[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const u64 t1 = __builtin_readcyclecounter();
  out[0] = t1 - t0; out[1] = p;
}
__global__ void k_chase_plain(const u64 *buf, u64 n, int hops, u64 *out) {
  u64 p = 0;
  const u64 t0 = __builtin_readcyclecounter();
  for (int i = 0; i < hops; i++) p = buf[p];
  const u64 t1 = __builtin_readcyclecounter();
  out[0] = t1 - t0; out[1] = p;
}
int main() {
  u32 *cur; u64 *cur64, *sink, *buf, *out;
  hipMalloc(&cur, 4096 * 4); hipMalloc(&cur64, 4096 * 8); hipMalloc(&sink, 64); hipMalloc(&out, 64);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int grid : {64, 256, 512}) for (u32 R : {512u, 4096u}) {
    hipMemset(cur, 0, 4096 * 4);
    const int iters = 64;
    k_atomics<<<grid, 512>>>(cur, R, iters, sink, 0);
    hipDeviceSynchronize();
    hipEventRecord(a); k_atomics<<<grid, 512>>>(cur, R, iters, sink, 0); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("u32 atomics: grid %d x 512 threads, %u cursors, %d rounds: %.1f us total, %.2f us per round, %.2f G atomics/s\n",
           grid, R, iters, ms * 1e3, ms * 1e3 / iters, (double)grid * 512 * iters / ms / 1e6);
    hipMemset(cur64, 0, 4096 * 8);
    hipEventRecord(a); k_atomics64<<<grid, 512>>>(cur64, R, iters, sink); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("u64 atomics: grid %d x 512 threads, %u cursors, %d rounds: %.1f us total, %.2f us per round\n", grid, R, iters, ms * 1e3, ms * 1e3 / iters);
  }
  const u64 n = 1 << 22;                         // 32 MiB of u64
  hipMalloc(&buf, n * 8);
  std::vector<u64> h(n);
  u64 x = 1;
  for (u64 i = 0; i < n; i++) h[i] = 0;
  u64 p = 0;                                     // random cycle
  std::vector<u64> perm(n); for (u64 i = 0; i < n; i++) perm[i] = i;
  for (u64 i = n - 1; i > 0; i--) { x = x * 6364136223846793005ull + 1442695040888963407ull; u64 j = (x >> 33) % (i + 1); std::swap(perm[i], perm[j]); }
  for (u64 i = 0; i < n; i++) h[perm[i]] = perm[(i + 1) % n];
  (void)p;
  hipMemcpy(buf, h.data(), n * 8, hipMemcpyHostToDevice);
  u64 ho[2];
  k_chase<<<1, 1>>>(buf, n, 2000, out); hipDeviceSynchronize(); hipMemcpy(ho, out, 16, hipMemcpyDeviceToHost);
  printf("agent-scope load chain: %.0f cycles per hop\n", (double)ho[0] / 2000);
  k_chase_plain<<<1, 1>>>(buf, n, 2000, out); hipDeviceSynchronize(); hipMemcpy(ho, out, 16, hipMemcpyDeviceToHost);
  printf("plain load chain:       %.0f cycles per hop\n", (double)ho[0] / 2000);
  return 0;
}
