// mgc_scan.hip -- multi-block scans (gfx950).
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
//  Multi-block scans over uint64 arrays (in place)
//    forward exclusive sum   : a[i] <- sum_{j<i} a[j]          (total returned in *total)
//    reverse inclusive min   : a[i] <- min(a[i], a[i+1], ..., a[n-1], init)
//  Three phases per level (local scan + block totals, recurse on the totals, add back);
//  4096 entries per workgroup, so two levels cover 16 M entries.
// ============================================================================
constexpr int SC_BLOCK = 1024;
constexpr int SC_ITEMS = 4;
constexpr int SC_CHUNK = SC_BLOCK * SC_ITEMS;

template <bool MIN_REVERSE>
__global__ __launch_bounds__(SC_BLOCK)
void scan_local_kernel(u64 *__restrict__ a, u64 n, u64 *__restrict__ block_tot) {
  __shared__ u64 s_tmp[SC_BLOCK / 64 + 1];
  const u64 chunk0 = (u64)blockIdx.x * SC_CHUNK;
  u64 v[SC_ITEMS];
  if (!MIN_REVERSE) {
    u64 sum = 0;
#pragma unroll
    for (int q = 0; q < SC_ITEMS; q++) {
      const u64 i = chunk0 + (u64)threadIdx.x * SC_ITEMS + q;
      v[q] = (i < n) ? a[i] : 0ull;
      sum += v[q];
    }
    u64 tot;
    u64 run = block_excl_scan<SC_BLOCK, u64>(sum, s_tmp, &tot);
#pragma unroll
    for (int q = 0; q < SC_ITEMS; q++) {
      const u64 i = chunk0 + (u64)threadIdx.x * SC_ITEMS + q;
      if (i < n) a[i] = run;
      run += v[q];
    }
    if (threadIdx.x == 0) block_tot[blockIdx.x] = tot;
  } else {
    // thread t owns entries (reversed inside the chunk) so that a forward min-scan over t is a suffix min
    const u32 rt = SC_BLOCK - 1 - threadIdx.x;
    u64 m = ~0ull;
#pragma unroll
    for (int q = SC_ITEMS - 1; q >= 0; q--) {
      const u64 i = chunk0 + (u64)rt * SC_ITEMS + q;
      v[q] = (i < n) ? a[i] : ~0ull;
      m = (v[q] < m) ? v[q] : m;
      v[q] = m;                                       // suffix min inside the thread's items
    }
    // inclusive min-scan across threads (thread 0 holds the LAST items of the chunk)
    const u32 lane = lane_id(), w = wave_id();
    u64 x = m;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const u64 y = __shfl_up(x, d);
      if ((int)lane >= d) x = (y < x) ? y : x;
    }
    __syncthreads();
    if (lane == 63) s_tmp[w] = x;
    __syncthreads();
    u64 pre = ~0ull;                                  // min over earlier threads (= later entries)
    for (u32 i = 0; i < w; i++) pre = (s_tmp[i] < pre) ? s_tmp[i] : pre;
    const u64 y = __shfl_up(x, 1);
    u64 before = (lane == 0) ? pre : ((y < pre) ? y : pre);
#pragma unroll
    for (int q = 0; q < SC_ITEMS; q++) {
      const u64 i = chunk0 + (u64)rt * SC_ITEMS + q;
      const u64 r = (v[q] < before) ? v[q] : before;
      if (i < n) a[i] = r;
    }
    if (threadIdx.x == SC_BLOCK - 1) {                // owner of the chunk's first entries: chunk minimum
      const u64 tot = (x < pre) ? x : pre;
      block_tot[blockIdx.x] = tot;
    }
  }
}

template <bool MIN_REVERSE>
__global__ __launch_bounds__(SC_BLOCK)
void scan_add_kernel(u64 *__restrict__ a, u64 n, const u64 *__restrict__ block_tot, u64 nblocks, u64 init) {
  // forward sum : add the exclusive prefix of the block totals (block_tot already scanned)
  // reverse min : combine with the suffix min of the LATER blocks' minima (block_tot already scanned) and init
  const u64 b = blockIdx.x;
  u64 carry;
  if (!MIN_REVERSE) carry = block_tot[b];
  else              carry = (b + 1 < nblocks) ? ((block_tot[b + 1] < init) ? block_tot[b + 1] : init) : init;
#pragma unroll
  for (int q = 0; q < SC_ITEMS; q++) {
    const u64 i = b * SC_CHUNK + (u64)threadIdx.x * SC_ITEMS + q;
    if (i < n) {
      if (!MIN_REVERSE) a[i] += carry;
      else { const u64 x = a[i]; a[i] = (x < carry) ? x : carry; }
    }
  }
}

__global__ void scan_store_total_kernel(const u64 *__restrict__ last_tot, u64 *__restrict__ total) { *total = *last_tot; }

// scratch needs (n/4096 + n/4096^2 + 8) uint64
size_t scan_scratch_elems(uint64_t n) {
  uint64_t e = 8, m = n;
  while (m > 1) { m = (m + SC_CHUNK - 1) / SC_CHUNK; e += m + 1; if (m == 1) break; }
  return (size_t)e + 8;
}

template <bool MIN_REVERSE>
static hipError_t scan_u64_inplace(u64 *a, uint64_t n, u64 *scratch, u64 *d_total /* may be null */, u64 init,
                                   hipStream_t st) {
  if (n == 0) {
    if (d_total) return hipMemsetAsync(d_total, 0, sizeof(u64), st);
    return hipSuccess;
  }
  const uint64_t nblocks = (n + SC_CHUNK - 1) / SC_CHUNK;
  hipLaunchKernelGGL(scan_local_kernel<MIN_REVERSE>, dim3((uint32_t)nblocks), dim3(SC_BLOCK), 0, st, a, (u64)n, scratch);
  MGC_CHECK(hipGetLastError());
  if (nblocks > 1 || MIN_REVERSE) {
    // level 2: scan the block totals (forward: exclusive sum; reverse: suffix min), then fold them back
    if (!MIN_REVERSE) {
      MGC_CHECK(scan_u64_inplace<false>(scratch, nblocks, scratch + nblocks + 1, d_total, 0, st));
    } else if (nblocks > 1) {
      MGC_CHECK(scan_u64_inplace<true>(scratch, nblocks, scratch + nblocks + 1, nullptr, ~0ull, st));
    }
    hipLaunchKernelGGL(scan_add_kernel<MIN_REVERSE>, dim3((uint32_t)nblocks), dim3(SC_BLOCK), 0, st, a, (u64)n,
                       (const u64 *)scratch, (u64)nblocks, init);
    MGC_CHECK(hipGetLastError());
  } else {
    if (!MIN_REVERSE) {
      if (d_total) { hipLaunchKernelGGL(scan_store_total_kernel, dim3(1), dim3(1), 0, st, (const u64 *)scratch, d_total); MGC_CHECK(hipGetLastError()); }
    }
  }
  return hipSuccess;
}

hipError_t scan_u64_exclusive(u64 *a, uint64_t n, u64 *scratch, u64 *d_total, hipStream_t st) {
  return scan_u64_inplace<false>(a, n, scratch, d_total, 0, st);
}
hipError_t scan_u64_min_reverse(u64 *a, uint64_t n, u64 *scratch, u64 init, hipStream_t st) {
  return scan_u64_inplace<true>(a, n, scratch, nullptr, init, st);
}



hipError_t warm_scan() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&scan_store_total_kernel)); }
}  // namespace mgc
