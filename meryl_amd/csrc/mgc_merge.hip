// mgc_merge.hip -- two sorted (k-mer, value) streams -> one, on the device (gfx950).
//
// What it replaces in the reference (paths relative to the reference root):
//   * merylBlockWriter::finish() merging the iterations writeBatch spilled when memory filled up
//     (src/meryl/merylOp-countThreads.C:285-380; the merge itself is in the absent meryl-utility): here the running
//     result of the batches counted so far stays in HBM and every new batch result is merged into it;
//   * the k-way streaming merge of merylOperation::nextMer for `union-sum` and friends
//     (src/meryl/merylOp-nextMer.C:418-683: find the smallest k-mer over the inputs :478-523, combine the values of the
//     inputs that hold it :560-641), restricted to two inputs per launch; more inputs fold pairwise.
//
// Merge path: tile t owns merged positions [t*TILE, (t+1)*TILE) of the (conceptual) merged-with-duplicates sequence; one
// binary search per tile boundary over the two global arrays finds where the tile's A and B ranges begin, the ranges are
// staged in LDS, every thread finds its own split in LDS and merges ITEMS elements serially.  Both inputs hold distinct
// ascending keys, so a key occurs at most twice and -- ties take A first -- as the adjacent pair (A, B): the A element
// looks at the next B candidate, the B element at the previous A element; no cross-thread exchange is needed.  Two passes
// (count heads per tile, scan, emit): HBM-bound, (8|16)+4 B read per input element per pass + the output.
#include "mgc_common.hpp"

namespace mgc {

constexpr int MG_BLOCK = 256;
constexpr int MG_ITEMS = 8;
constexpr int MG_TILE  = MG_BLOCK * MG_ITEMS;

// ops: what the output holds and how values combine (meryl's union-* / intersect-* over two inputs)
//   0 union-sum   1 union-min   2 union-max   3 intersect-sum   4 intersect-min   5 intersect-max
__device__ __forceinline__ bool mg_is_intersect(int op) { return op >= 3; }
__device__ __forceinline__ u32 mg_combine(int op, u32 a, u32 b) {
  const int f = op % 3;
  return f == 0 ? a + b : (f == 1 ? (a < b ? a : b) : (a > b ? a : b));       // the sum wraps mod 2^32 like kmvalu arithmetic
}

// number of A elements among the first d merged elements (ties: A first)
template <typename K>
__device__ __forceinline__ u64 mg_path(const K *__restrict__ A, u64 nA, const K *__restrict__ B, u64 nB, u64 d) {
  u64 lo = d > nB ? d - nB : 0, hi = d < nA ? d : nA;
  while (lo < hi) {
    const u64 mid = lo + ((hi - lo) >> 1);
    if (!KeyOps<K>::lt(B[d - 1 - mid], A[mid])) lo = mid + 1; else hi = mid;   // A[mid] <= B[d-1-mid]: A[mid] is inside
  }
  return lo;
}

template <typename K, bool EMIT>
__global__ __launch_bounds__(MG_BLOCK)
void merge_kernel(const K *__restrict__ A, const u32 *__restrict__ cA, u64 nA, const K *__restrict__ B,
                  const u32 *__restrict__ cB, u64 nB, int op, u64 *__restrict__ tile_cnt /*EMIT: exclusive bases*/,
                  K *__restrict__ outK, u32 *__restrict__ outC) {
  __shared__ K   s_keys[MG_TILE];
  __shared__ u64 s_split[2];
  __shared__ u32 s_tmp[MG_BLOCK / 64 + 1];
  const u64 total = nA + nB;
  const u64 d0 = (u64)blockIdx.x * MG_TILE, d1 = (d0 + MG_TILE < total) ? d0 + MG_TILE : total;
  if (threadIdx.x < 2) s_split[threadIdx.x] = mg_path<K>(A, nA, B, nB, threadIdx.x ? d1 : d0);
  __syncthreads();
  const u64 a0 = s_split[0], a1 = s_split[1], b0 = d0 - a0, b1 = d1 - a1;
  const u32 na = (u32)(a1 - a0), nb = (u32)(b1 - b0), nt = na + nb;
  for (u32 i = threadIdx.x; i < nt; i += MG_BLOCK) s_keys[i] = (i < na) ? A[a0 + i] : B[b0 + (i - na)];
  __syncthreads();
  const K *sA = s_keys, *sB = s_keys + na;
  // this thread's merged positions [l0, l1) of the tile
  const u32 l0 = (threadIdx.x * MG_ITEMS < nt) ? threadIdx.x * MG_ITEMS : nt;
  const u32 l1 = (l0 + MG_ITEMS < nt) ? l0 + MG_ITEMS : nt;
  u32 la;
  {
    u32 lo = l0 > nb ? l0 - nb : 0, hi = l0 < na ? l0 : na;
    while (lo < hi) {
      const u32 mid = (lo + hi) >> 1;
      if (!KeyOps<K>::lt(sB[l0 - 1 - mid], sA[mid])) lo = mid + 1; else hi = mid;
    }
    la = lo;
  }
  u32 lb = l0 - la;
  const bool inter = mg_is_intersect(op);
  // pass over the thread's elements: which are output heads, and with what value
  u32 heads = 0;
  u32 head_mask = 0;                       // bit q: element q is written
  K   kreg[MG_ITEMS];
  u32 vreg[MG_ITEMS];
#pragma unroll
  for (int q = 0; q < MG_ITEMS; q++) {
    kreg[q] = KeyOps<K>::zero(); vreg[q] = 0;
    if (l0 + q >= l1) continue;
    const bool take_a = (la < na) && (lb >= nb || !KeyOps<K>::lt(sB[lb], sA[la]));
    if (take_a) {
      const K key = sA[la];
      // the equal B element, if any, is the next B candidate: in LDS, or the first B element after the tile
      bool dup = false;
      u64 bidx = b0 + lb;
      if (lb < nb) dup = !KeyOps<K>::ne(sB[lb], key);
      else if (bidx < nB) dup = !KeyOps<K>::ne(B[bidx], key);
      const bool out = inter ? dup : true;
      if (out) {
        head_mask |= 1u << q; heads++;
        if (EMIT) { kreg[q] = key; const u32 va = cA[a0 + la]; vreg[q] = dup ? mg_combine(op, va, cB[bidx]) : va; }
      }
      la++;
    } else {
      const K key = sB[lb];
      // a duplicate iff the A element merged just before it is equal (it then carried the combined value)
      bool dup = false;
      if (la > 0) dup = !KeyOps<K>::ne(sA[la - 1], key);
      else if (a0 > 0) dup = !KeyOps<K>::ne(A[a0 - 1], key);
      const bool out = !inter && !dup;
      if (out) {
        head_mask |= 1u << q; heads++;
        if (EMIT) { kreg[q] = key; vreg[q] = cB[b0 + lb]; }
      }
      lb++;
    }
  }
  u32 tot;
  const u32 base = block_excl_scan<MG_BLOCK, u32>(heads, s_tmp, &tot);
  if (!EMIT) {
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = tot;
    return;
  }
  u64 o = tile_cnt[blockIdx.x] + base;
#pragma unroll
  for (int q = 0; q < MG_ITEMS; q++) {
    if (head_mask & (1u << q)) { outK[o] = kreg[q]; outC[o] = vreg[q]; o++; }
  }
}

// workspace: [0] total (u64), [8..] tile counts (u64 x tiles), scan scratch
static inline uint64_t merge_tiles(uint64_t na, uint64_t nb) { return (na + nb + MG_TILE - 1) / MG_TILE; }

size_t merge_workspace_bytes(uint64_t na, uint64_t nb) {
  const uint64_t t = merge_tiles(na, nb);
  return (size_t)(8 + t + 1 + scan_scratch_elems(t + 1)) * sizeof(u64) + 256;
}

// pass 1: leaves the output length at ws[0] (read it with merge_read_total after the stream is synchronised)
hipError_t launch_merge_count(const void *dA, uint64_t na, const void *dB, uint64_t nb, uint32_t key_words, int op, void *d_ws,
                              hipStream_t st) {
  u64 *ws = reinterpret_cast<u64 *>(d_ws);
  const uint64_t t = merge_tiles(na, nb);
  if (t == 0) return hipMemsetAsync(ws, 0, 8, st);
  u64 *tiles = ws + 8, *scratch = tiles + t + 1;
  if (key_words == 2)
    hipLaunchKernelGGL((merge_kernel<K128, false>), dim3((uint32_t)t), dim3(MG_BLOCK), 0, st, reinterpret_cast<const K128 *>(dA),
                       (const u32 *)nullptr, (u64)na, reinterpret_cast<const K128 *>(dB), (const u32 *)nullptr, (u64)nb, op, tiles,
                       (K128 *)nullptr, (u32 *)nullptr);
  else
    hipLaunchKernelGGL((merge_kernel<u64, false>), dim3((uint32_t)t), dim3(MG_BLOCK), 0, st, reinterpret_cast<const u64 *>(dA),
                       (const u32 *)nullptr, (u64)na, reinterpret_cast<const u64 *>(dB), (const u32 *)nullptr, (u64)nb, op, tiles,
                       (u64 *)nullptr, (u32 *)nullptr);
  MGC_CHECK(hipGetLastError());
  return scan_u64_exclusive(tiles, t, scratch, ws, st);
}

hipError_t merge_read_total(const void *d_ws, uint64_t *n_out, hipStream_t st) {
  hipError_t e = hipMemcpyAsync(n_out, d_ws, sizeof(uint64_t), hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(st);
}

// pass 2 (same inputs, the workspace pass 1 left): writes the merged stream
hipError_t launch_merge_emit(const void *dA, const uint32_t *cA, uint64_t na, const void *dB, const uint32_t *cB, uint64_t nb,
                             uint32_t key_words, int op, void *d_ws, void *d_out_keys, uint32_t *d_out_counts, hipStream_t st) {
  const uint64_t t = merge_tiles(na, nb);
  if (t == 0) return hipSuccess;
  u64 *tiles = reinterpret_cast<u64 *>(d_ws) + 8;
  if (key_words == 2)
    hipLaunchKernelGGL((merge_kernel<K128, true>), dim3((uint32_t)t), dim3(MG_BLOCK), 0, st, reinterpret_cast<const K128 *>(dA), cA,
                       (u64)na, reinterpret_cast<const K128 *>(dB), cB, (u64)nb, op, tiles, reinterpret_cast<K128 *>(d_out_keys),
                       d_out_counts);
  else
    hipLaunchKernelGGL((merge_kernel<u64, true>), dim3((uint32_t)t), dim3(MG_BLOCK), 0, st, reinterpret_cast<const u64 *>(dA), cA,
                       (u64)na, reinterpret_cast<const u64 *>(dB), cB, (u64)nb, op, tiles, reinterpret_cast<u64 *>(d_out_keys),
                       d_out_counts);
  return hipGetLastError();
}

// ---- small helper of the batch cut: position after the last '.' of a staged base stream ------------------------------
__global__ __launch_bounds__(256)
void last_breaker_kernel(const uint8_t *__restrict__ bases, u64 n, u64 *__restrict__ out /*zeroed*/) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  u64 best = 0;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    if (bases[i] == (uint8_t)'.') best = i + 1;
  // wave max, then one atomic per wave
  for (int d = 32; d > 0; d >>= 1) { const u64 o = __shfl_xor(best, d); best = o > best ? o : best; }
  if (lane_id() == 0 && best) atomicMax(reinterpret_cast<unsigned long long *>(out), (unsigned long long)best);
}

// *d_out (device u64) <- 1 + index of the last '.' in bases[0, n), 0 if there is none
hipError_t launch_last_breaker(const uint8_t *d_bases, uint64_t n, uint64_t *d_out, hipStream_t st) {
  MGC_CHECK(hipMemsetAsync(d_out, 0, sizeof(uint64_t), st));
  if (n == 0) return hipSuccess;
  uint64_t wgs = (n + 256 * 64 - 1) / (256 * 64);
  if (wgs > 4096) wgs = 4096;
  hipLaunchKernelGGL(last_breaker_kernel, dim3((uint32_t)wgs), dim3(256), 0, st, d_bases, (u64)n, reinterpret_cast<u64 *>(d_out));
  return hipGetLastError();
}

}  // namespace mgc
