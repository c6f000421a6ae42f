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

// ops: what the output holds and how values combine (meryl's union-* / intersect-* / set operations over two inputs,
// src/meryl/merylOp-nextMer.C:559-613; more inputs fold from the left)
//   0 union-sum   1 union-min   2 union-max   3 intersect-sum   4 intersect-min   5 intersect-max
//   6 intersect (the first input's value)      7 subtract (a - b while a > b, otherwise dropped; :52-62)
//   8 difference (in A only)                   9 symmetric-difference (in exactly one of the two)
constexpr int MG_NUM_OPS = 10;
__device__ __forceinline__ u32 mg_combine(int op, u32 a, u32 b) {
  if (op == 6) return a;
  if (op == 7) return a - b;
  const int f = op % 3;
  return f == 0 ? a + b : (f == 1 ? (a < b ? a : b) : (a > b ? a : b));       // the sum wraps mod 2^32 like kmvalu arithmetic
}
// an A element (dup: B holds the same k-mer) / a B element (dup: A holds it too) is written
__device__ __forceinline__ bool mg_keep_a(int op, bool dup, u32 va, u32 vb) {
  if (op <= 2) return true;
  if (op <= 6) return dup;
  if (op == 7) return !dup || va > vb;
  return !dup;
}
__device__ __forceinline__ bool mg_keep_b(int op, bool dup) { return (op <= 2 || op == 9) && !dup; }
__device__ __forceinline__ bool mg_needs_values(int op) { return op == 7; }   // whether an element is written depends on the values

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
  const bool need_v = EMIT || mg_needs_values(op);
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
      u32 va = 0, vb = 0;
      if (need_v) { va = cA[a0 + la]; if (dup) vb = cB[bidx]; }
      if (mg_keep_a(op, dup, va, vb)) {
        head_mask |= 1u << q; heads++;
        if (EMIT) { kreg[q] = key; vreg[q] = dup ? mg_combine(op, va, vb) : va; }
      }
      la++;
    } else {
      const K key = sB[lb];
      // a duplicate iff the A element merged just before it is equal (it then carried the combined value)
      bool dup = false;
      if (la > 0) dup = !KeyOps<K>::ne(sA[la - 1], key);
      else if (a0 > 0) dup = !KeyOps<K>::ne(A[a0 - 1], key);
      if (mg_keep_b(op, dup)) {
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
                              hipStream_t st, const uint32_t *cA, const uint32_t *cB) {
  if (op < 0 || op >= MG_NUM_OPS || (op == 7 && ((na && !cA) || (nb && !cB)))) return hipErrorInvalidValue;
  u64 *ws = reinterpret_cast<u64 *>(d_ws);
  const uint64_t t = merge_tiles(na, nb);
  if (t == 0) return hipMemsetAsync(ws, 0, 8, st);
  u64 *tiles = ws + 8, *scratch = tiles + t + 1;
  if (key_words == 2)
    hipLaunchKernelGGL((merge_kernel<K128, false>), dim3((uint32_t)t), dim3(MG_BLOCK), 0, st, reinterpret_cast<const K128 *>(dA),
                       cA, (u64)na, reinterpret_cast<const K128 *>(dB), cB, (u64)nb, op, tiles,
                       (K128 *)nullptr, (u32 *)nullptr);
  else
    hipLaunchKernelGGL((merge_kernel<u64, false>), dim3((uint32_t)t), dim3(MG_BLOCK), 0, st, reinterpret_cast<const u64 *>(dA),
                       cA, (u64)na, reinterpret_cast<const u64 *>(dB), cB, (u64)nb, op, tiles,
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

// ---- one stream: values transformed, k-mers whose new value is zero dropped (order kept) ------------------------------------
// The single-input operations of src/meryl/merylOp-nextMer.C:490-557: the value filters (less-than ... not-equal-to: the value
// passes or becomes 0) and the arithmetic ones (increase ... modulo, with the reference's overflow / underflow / divide-by-zero
// results); a value of 0 means "do not output" (:470-474).  fop 12: keep the k-mers whose FLAG (a second value array) is 1 --
// the exactly-one-input test of symmetric-difference over more than two inputs.  kmvalu is 32 bits: results are truncated.
__device__ __forceinline__ u32 sel_value(int fop, u32 v, u64 c, u32 flag) {
  switch (fop) {
    case 0:  return (u64)v <  c ? v : 0u;
    case 1:  return (u64)v >  c ? v : 0u;
    case 2:  return (u64)v >= c ? v : 0u;
    case 3:  return (u64)v <= c ? v : 0u;
    case 4:  return (u64)v == c ? v : 0u;
    case 5:  return (u64)v != c ? v : 0u;
    case 6:  return (~0ull - (u64)v < c) ? ~0u : (u32)((u64)v + c);
    case 7:  return ((u64)v < c) ? 0u : (u32)((u64)v - c);
    case 8:  return (v && ~0ull / (u64)v < c) ? ~0u : (u32)((u64)v * c);
    case 9:  return c == 0 ? 0u : (u32)((u64)v / c);
    case 10: return c == 0 ? 0u : ((u64)v < c ? 1u : (u32)(u64)__builtin_round((double)v / (double)c));
    case 11: return c == 0 ? 0u : (u32)((u64)v % c);
    default: return flag == 1u ? v : 0u;
  }
}

constexpr int SL_BLOCK = 256, SL_ITEMS = 8, SL_TILE = SL_BLOCK * SL_ITEMS;
template <typename K, bool EMIT>
__global__ __launch_bounds__(SL_BLOCK)
void select_kernel(const K *__restrict__ keys, const u32 *__restrict__ vals, const u32 *__restrict__ flags, u64 n, int fop, u64 c,
                   u64 *__restrict__ tile_cnt /*EMIT: exclusive bases*/, K *__restrict__ outK, u32 *__restrict__ outC) {
  __shared__ u32 s_tmp[SL_BLOCK / 64 + 1];
  const u64 base = (u64)blockIdx.x * SL_TILE + (u64)threadIdx.x * SL_ITEMS;
  u32 nv[SL_ITEMS], kept = 0;
#pragma unroll
  for (int q = 0; q < SL_ITEMS; q++) {
    nv[q] = 0;
    if (base + q < n) nv[q] = sel_value(fop, vals[base + q], c, flags ? flags[base + q] : 0u);
    kept += nv[q] ? 1u : 0u;
  }
  u32 tot;
  const u32 off = block_excl_scan<SL_BLOCK, u32>(kept, s_tmp, &tot);
  if (!EMIT) { if (threadIdx.x == 0) tile_cnt[blockIdx.x] = tot; return; }
  u64 o = tile_cnt[blockIdx.x] + off;
#pragma unroll
  for (int q = 0; q < SL_ITEMS; q++)
    if (nv[q]) { outK[o] = keys[base + q]; outC[o] = nv[q]; o++; }
}

__global__ void fill_u32_kernel(u32 *__restrict__ p, u64 n, u32 v) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

static inline uint64_t select_tiles(uint64_t n) { return (n + SL_TILE - 1) / SL_TILE; }
size_t select_workspace_bytes(uint64_t n) {
  const uint64_t t = select_tiles(n);
  return (size_t)(8 + t + 1 + scan_scratch_elems(t + 1)) * sizeof(u64) + 256;
}
// pass 1: the number of k-mers kept lands at ws[0] (merge_read_total reads it)
hipError_t launch_select_count(const void *d_keys, const uint32_t *d_vals, const uint32_t *d_flags, uint64_t n, uint32_t key_words, int fop,
                               uint64_t constant, void *d_ws, hipStream_t st) {
  if (fop < 0 || fop > 12 || (fop == 12 && n && !d_flags)) return hipErrorInvalidValue;
  u64 *ws = reinterpret_cast<u64 *>(d_ws);
  const uint64_t t = select_tiles(n);
  if (t == 0) return hipMemsetAsync(ws, 0, 8, st);
  u64 *tiles = ws + 8, *scratch = tiles + t + 1;
  if (key_words == 2)
    hipLaunchKernelGGL((select_kernel<K128, false>), dim3((uint32_t)t), dim3(SL_BLOCK), 0, st, reinterpret_cast<const K128 *>(d_keys), d_vals, d_flags,
                       (u64)n, fop, (u64)constant, tiles, (K128 *)nullptr, (u32 *)nullptr);
  else
    hipLaunchKernelGGL((select_kernel<u64, false>), dim3((uint32_t)t), dim3(SL_BLOCK), 0, st, reinterpret_cast<const u64 *>(d_keys), d_vals, d_flags,
                       (u64)n, fop, (u64)constant, tiles, (u64 *)nullptr, (u32 *)nullptr);
  MGC_CHECK(hipGetLastError());
  return scan_u64_exclusive(tiles, t, scratch, ws, st);
}
hipError_t launch_select_emit(const void *d_keys, const uint32_t *d_vals, const uint32_t *d_flags, uint64_t n, uint32_t key_words, int fop,
                              uint64_t constant, void *d_ws, void *d_out_keys, uint32_t *d_out_vals, hipStream_t st) {
  const uint64_t t = select_tiles(n);
  if (t == 0) return hipSuccess;
  u64 *tiles = reinterpret_cast<u64 *>(d_ws) + 8;
  if (key_words == 2)
    hipLaunchKernelGGL((select_kernel<K128, true>), dim3((uint32_t)t), dim3(SL_BLOCK), 0, st, reinterpret_cast<const K128 *>(d_keys), d_vals, d_flags,
                       (u64)n, fop, (u64)constant, tiles, reinterpret_cast<K128 *>(d_out_keys), d_out_vals);
  else
    hipLaunchKernelGGL((select_kernel<u64, true>), dim3((uint32_t)t), dim3(SL_BLOCK), 0, st, reinterpret_cast<const u64 *>(d_keys), d_vals, d_flags,
                       (u64)n, fop, (u64)constant, tiles, reinterpret_cast<u64 *>(d_out_keys), d_out_vals);
  return hipGetLastError();
}
hipError_t launch_fill_u32(uint32_t *d, uint64_t n, uint32_t v, hipStream_t st) {
  if (n == 0) return hipSuccess;
  uint64_t wgs = (n + 255) / 256;
  if (wgs > 8192) wgs = 8192;
  hipLaunchKernelGGL(fill_u32_kernel, dim3((uint32_t)wgs), dim3(256), 0, st, d, (u64)n, v);
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
