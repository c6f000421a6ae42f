// mdb_layout.h -- the database byte layout (assumptions A1..A10 of meryl_db.cpp) as constants and
// closed-form size functions, shared by the host encoder (meryl_db.cpp) and the device encoder
// (mgc_encode.hip) so that both write the same bytes.  Internal; not installed.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define MDB_HD __host__ __device__ inline
#else
#define MDB_HD inline
#endif

namespace mdb {

constexpr uint64_t MAGIC_DAT1 = 0x7461446c7972656dull;   // "merylDat"
constexpr uint64_t MAGIC_DAT2 = 0x0a3030656c694661ull;   // "aFile00\n"
constexpr uint64_t MAGIC_IDX1 = 0x646e496c7972656dull;   // "merylInd"  (usage.rst:15)
constexpr uint64_t MAGIC_IDX2 = 0x32302e765f5f7865ull;   // "ex__v.02"
constexpr uint64_t STUFFED_BLOCK_BITS  = 16ull * 1024 * 1024 * 8;   // default stuffedBits block (A2)
constexpr uint64_t STUFFED_BLOCK_WORDS = STUFFED_BLOCK_BITS / 64;
constexpr uint32_t BLOCK_HEADER_BITS   = 528;            // A4: 4x64 + 8 + 32 + 32 + 64 + 8 + 64 + 64
constexpr uint32_t VALUE_BITS          = 32;             // A6

// A4: unaryBits = smallest u with 2^u >= nKmers, at most suffixSize
MDB_HD uint32_t unary_bits_for(uint64_t n, uint32_t suffix_size) {
  uint32_t u = 0;
  for (uint64_t sum = 1; sum < n; sum <<= 1) u++;
  return u > suffix_size ? suffix_size : u;
}

// bits of one encoded data block: header, k-mers (A5: the unary deltas of a block telescope to the last
// k-mer's top part), values (A6), labels (A10)
MDB_HD uint64_t block_bits(uint64_t n, uint64_t top_last, uint32_t binary_bits, uint32_t label_size) {
  return BLOCK_HEADER_BITS + (n ? top_last : 0) + n * (uint64_t)(1 + binary_bits + VALUE_BITS + label_size);
}

// A2: bytes of one dumped stuffedBits object holding `bits` bits
MDB_HD uint64_t stuffed_sub_blocks(uint64_t bits) {
  const uint64_t nsb = (bits + STUFFED_BLOCK_BITS - 1) / STUFFED_BLOCK_BITS;
  return nsb ? nsb : 1;
}
MDB_HD uint64_t stuffed_bytes(uint64_t bits) {
  const uint64_t nsb = stuffed_sub_blocks(bits);
  const uint64_t last = bits - (nsb - 1) * STUFFED_BLOCK_BITS;
  return 16 + 16 * nsb + 16 * nsb + 8 * ((nsb - 1) * STUFFED_BLOCK_WORDS + (last + 63) / 64);
}
// byte offset, inside the dumped object, of word w of the object's logical bit stream
MDB_HD uint64_t stuffed_word_offset(uint64_t nsb, uint64_t w) {
  return 16 + 16 * nsb + 16 * (w / STUFFED_BLOCK_WORDS + 1) + 8 * w;
}

}  // namespace mdb
