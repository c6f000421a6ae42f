// mgc_decode.hip -- database blocks decoded ON THE DEVICE (gfx950): the inverse of mgc_encode.hip.
//
// What it replaces: the consumers of a database -- `union-sum` and friends (merylOperation::nextMer reading its inputs
// through merylFileReader, src/meryl/merylOp-nextMer.C:418-470) and merylExactLookup::load
// (src/meryl-lookup/meryl-lookup.C:89-100) -- decode every block on host threads: 65-70 M k-mers/s per thread here
// (DESIGN.md section 9), i.e. a 9 G k-mers/s counter was read back at ~2 G/s on 32 threads.  Here the data file's bytes go
// to HBM as they are (less than half the size of the decoded arrays) and the blocks are decoded there.
//
// A block (mdb_layout.h / meryl_db.cpp A2-A6, A10) is a bit stream: header (528 bits), then k-mer i as
// unary(top_i - top_{i-1}) followed by its low `binaryBits` bits, then n 32-bit values, then n labels.  The unary codes
// make the k-mer section sequential -- where k-mer i begins depends on every k-mer before it -- so the unit of
// parallelism is the BLOCK: one thread decodes one block (a 10 Gbp database has 2^18 of them, ~4500 k-mers each; a
// database with few, large blocks decodes more slowly but never wrongly).  Each thread streams its block's words
// through a two-word window; the host has validated every block's framing against the file size beforehand
// (mdb_reader_raw_file), and the kernel checks the header fields and every bit position against the block's length:
// a corrupt block sets the error word and stops, it never reads outside its object.
#include "mgc_common.hpp"
#include "mdb_layout.h"
#include "../../include/meryl_db.h"

namespace mgc {

struct BitWin {                                        // MSB-first cursor over the logical words of one stuffedBits object
  const unsigned char *obj; u64 nsb, wi, nw; u64 w0, w1; u32 off;
  // (the window always holds word wi + 1 as well: beyond the object's last logical word it is zero, never a load -- the
  // last object of a file ends where the upload ends, and a bit length that is a whole number of sub-blocks would
  // otherwise step over a sub-block header that does not exist)
  __device__ __forceinline__ u64 word(u64 i) const { return i < nw ? *reinterpret_cast<const u64 *>(obj + mdb::stuffed_word_offset(nsb, i)) : 0ull; }
  __device__ __forceinline__ void seek(u64 pos) { wi = pos >> 6; off = (u32)(pos & 63); w0 = word(wi); w1 = word(wi + 1); }
  __device__ __forceinline__ u64 peek() const { return off ? ((w0 << off) | (w1 >> (64 - off))) : w0; }
  __device__ __forceinline__ void skip(u32 n) {        // n <= 64
    off += n;
    if (off >= 64) { off -= 64; wi++; w0 = w1; w1 = word(wi + 1); }
  }
  __device__ __forceinline__ u64 get(u32 width) { const u64 v = peek() >> (64 - width); skip(width); return v; }   // width 1..64
  __device__ __forceinline__ u64 pos() const { return (wi << 6) + off; }
};

template <typename K>
__global__ __launch_bounds__(128)
void decode_blocks_kernel(const unsigned char *__restrict__ file, const mdb_raw_block *__restrict__ blocks, u64 n_blocks, u32 ss,
                          u32 label_size, K *__restrict__ keys, u32 *__restrict__ counts, u32 *__restrict__ err) {
  const u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_blocks) return;
  const mdb_raw_block d = blocks[b];
  BitWin bw;
  bw.obj = file + d.object_offset; bw.nsb = d.n_sub_blocks; bw.nw = (d.n_bits + 63) >> 6;
  bw.seek(0);
  const u64 nbits = d.n_bits;
  // header (A4)
  bool ok = nbits >= mdb::BLOCK_HEADER_BITS;
  u64 n = 0; u32 ub = 0, bb = 0;
  if (ok) {
    const u64 m1 = bw.get(64), m2 = bw.get(64), prefix = bw.get(64);
    n = bw.get(64);
    (void)bw.get(8);
    ub = (u32)bw.get(32); bb = (u32)bw.get(32);
    (void)bw.get(64); (void)bw.get(8); (void)bw.get(64); (void)bw.get(64);
    ok = m1 == mdb::MAGIC_DAT1 && m2 == mdb::MAGIC_DAT2 && prefix == d.prefix && n == d.n_kmers && (u64)ub + bb == ss && ub <= 64;
    // the block cannot hold fewer bits than its k-mers need at their shortest
    ok = ok && nbits >= mdb::BLOCK_HEADER_BITS + n * (u64)(1 + bb + mdb::VALUE_BITS + label_size);
  }
  if (!ok) { atomicExch(err, 1u); return; }
  K *ko = keys + d.out_offset;
  u32 *co = counts + d.out_offset;
  u64 top = 0;
  const u128 pre = (u128)d.prefix << ss;
  for (u64 i = 0; i < n; i++) {
    // unary: zeros up to the next one bit
    for (;;) {
      const u64 x = bw.peek();
      if (x) { const u32 z = (u32)__builtin_clzll(x); top += z; bw.skip(z + 1); break; }
      top += 64; bw.skip(64);
      if (bw.pos() > nbits) { atomicExch(err, 2u); return; }
    }
    if (bw.pos() + bb > nbits || top >> (ub < 64 ? ub : 63) > (ub < 64 ? 0ull : 1ull)) { atomicExch(err, 2u); return; }
    u128 suf = (bb < 128) ? ((u128)top << bb) : (u128)0;
    if (bb > 64) { const u64 h = bw.get(bb - 64), l = bw.get(64); suf |= ((u128)h << 64) | (u128)l; }
    else if (bb) suf |= (u128)bw.get(bb);
    const u128 full = pre | suf;
    if constexpr (sizeof(K) == 16) ko[i] = KeyOps<K128>::mk(full);
    else                           ko[i] = (u64)full;
  }
  if (bw.pos() + n * (u64)(mdb::VALUE_BITS + label_size) > nbits) { atomicExch(err, 3u); return; }
  for (u64 i = 0; i < n; i++) co[i] = (u32)bw.get(mdb::VALUE_BITS);
}

hipError_t launch_decode_blocks(const void *d_file, const void *d_blocks, uint64_t n_blocks, uint32_t suffix_size, uint32_t label_size,
                                uint32_t key_words, void *d_keys, uint32_t *d_counts, uint32_t *d_err, hipStream_t st) {
  if (n_blocks == 0) return hipSuccess;
  const dim3 grid((uint32_t)((n_blocks + 127) / 128));
  if (key_words == 2)
    hipLaunchKernelGGL(decode_blocks_kernel<K128>, grid, dim3(128), 0, st, reinterpret_cast<const unsigned char *>(d_file),
                       reinterpret_cast<const mdb_raw_block *>(d_blocks), (u64)n_blocks, suffix_size, label_size, reinterpret_cast<K128 *>(d_keys), d_counts, d_err);
  else
    hipLaunchKernelGGL(decode_blocks_kernel<u64>, grid, dim3(128), 0, st, reinterpret_cast<const unsigned char *>(d_file),
                       reinterpret_cast<const mdb_raw_block *>(d_blocks), (u64)n_blocks, suffix_size, label_size, reinterpret_cast<u64 *>(d_keys), d_counts, d_err);
  return hipGetLastError();
}

}  // namespace mgc
