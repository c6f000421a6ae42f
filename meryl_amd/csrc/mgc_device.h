// mgc_device.h -- internal launch interface between the C-ABI layer
// (mgc_api.cpp) and the gfx950 kernels (mgc_kmer / mgc_sort / mgc_scan / mgc_finish / mgc_misc / mgc_parse .hip).  Not installed.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mgc {

// ---- switches ----------------------------------------------------------------
// Every MGC_* environment switch of the count path (tests and A/B measurements; the defaults are the shipped plan), read ONCE:
// mgc_open reads them into the session (mgc_session::sw), the bare mgc_dev_* operators per call -- nothing below reads the
// environment on its own.
struct Switches {
  bool fine_hist = true;        // MGC_FINE_HIST=0: no fifteen-bit file histogram (every file's digit histogram by a read of its keys)
  bool hpc_msd = true;          // MGC_HPC_MSD=0: `compress`, low digit first everywhere
  bool hpc_digits = true;       // MGC_HPC_DIGITS=0: `compress` with plain bit digits
  bool const_k = true;          // MGC_KMER_CONST_K=0: the front-end kernels with run-time k only
  bool narrow = true;           // MGC_NARROW=0: no 32-bit words after the first grouping pass
  bool wide_msd = true;         // MGC_WIDE_MSD=0: whole keys low digit first off a histogram read
  bool soa5 = true;             // MGC_SOA5=0: no 5-byte layout (k = 20..23)
  bool group_pipe = true;       // MGC_GROUP_PIPE=0: the narrowed grouping passes fetch the next tile inside the look-back phase (round 5's form)
  bool k96 = true;              // MGC_K96=0: no 12-byte records (k = 33..51)
  bool finish = true;           // MGC_FINISH=0: stable sort of all bits + run-length kernels
  bool nolist = false;          // MGC_FINISH_NOLIST=1: the dense-grid instantiations of the count kernels whatever the grid holds
  bool finish_trace = false;    // MGC_FINISH_TRACE: what happens to oversized sub-buckets, on stderr
  bool group_dbg = false;       // MGC_GROUP_DBG: per-phase cycle sums of the grouping passes (instrumented instantiations)
  uint32_t pass_stagger = 0;    // MGC_PASS_STAGGER: groups the 5-byte first pass's workgroups start in (0 / 1: together -- the default: the
                                // staggered start hands the first tiles out statically, which is only safe while ALL workgroups of the launch are
                                // resident, i.e. while nothing else holds CUs of the device; mgc_sort.hip, STAGGER)
  bool hash_dbg = false;        // MGC_HASH_DBG: per-phase cycle sums of the count kernels
  int  hash_multi = -1;         // MGC_HASH_MULTI: sub-buckets per iteration of hash_count_multi_kernel (-1: by the file's average; 0: off)
  int  hash_stream = -1;        // MGC_HASH_STREAM: the distinct-sized count (hash_count_stream_kernel) and its coarser file plan (-1: where the probe
                                // file's distinct / instances ratio allows it; 0: off; 1: on every file whose suffix fits, whatever the ratio)
                                // 2: like -1, but only on narrowed files -- not on whole 8-byte k-mers (k = 24..32: the 64-bit-entry instantiation))
  uint32_t min_top = 0;         // MGC_FINISH_MIN_TOP: at least that many grouping bits per file (tests reach the large-input plans)
  uint64_t finish_target = 0;   // MGC_FINISH_TARGET: k-mers per sub-bucket the plan aims at (0: the kernels' default)
  uint64_t stream_max = (uint64_t)1 << 22;   // MGC_STREAM_MAX: sub-buckets up to this many keys are streamed without asking the probe
  bool     huge_slices = true;  // MGC_HUGE_SLICES=0: a gigantic sub-bucket (above 65536 keys) is streamed by ONE workgroup (round 5)
  uint32_t huge_streams = 4;    // MGC_HUGE_STREAMS: streams the streaming kernels of oversized sub-buckets are spread over (1..4)
  uint64_t bucket_bases = 0;    // MGC_BUCKET_BASES: bases per partition bucket above which the partition gets finer (0: default)
};
Switches read_switches();



// ---- k-mer extraction / partition (mgc_kmer.hip) -----------------------
constexpr int      KP_BLOCK       = 256;                 // threads per workgroup
constexpr int      KP_ITEMS       = 16;                  // window starts per thread
constexpr int      KP_TILE        = KP_BLOCK * KP_ITEMS; // 4096 starts per tile
constexpr int      KP_MAX_BUCKETS = 1024;

// Persistent grid size used by both passes (they must agree).
uint32_t kp_grid_size(uint64_t n_bases, uint32_t bucket_bits);   // rows of the per-workgroup histogram (virtual workgroups)
size_t   kp_workspace_bytes(uint32_t bucket_bits);

// sfx_mask / sfx_test: count-suffix= filter, a k-mer is kept iff (its low word & sfx_mask) == sfx_test (0, 0: keep all)
// the same + the k-mers per (file, next nine bits) into d_fine_hist[2^15] (the first digit of the narrowed grouping passes)
bool       kmer_histogram_fine_ok(uint32_t k, uint32_t bucket_bits, uint64_t sfx_mask, const Switches &sw);
// `compress`: the same histogram over dense ranks -- k-mers per (bucket, dense-rank digit of the five bases below the bucket's),
// d_fine_hist[kmer_histogram_hpc_entries(bucket_bits)], bucket_bits 6 or 8 (mgc_kmer.hip)
bool       kmer_histogram_hpc_ok(uint32_t k, uint32_t bucket_bits, uint64_t sfx_mask, const Switches &sw);
uint32_t   kmer_histogram_hpc_entries(uint32_t bucket_bits);
hipError_t launch_kmer_histogram_hpc(const uint8_t *d_bases, uint64_t n_bases, uint32_t k, int mode, uint32_t bucket_bits,
                                     uint64_t *d_bucket_counts, uint64_t *d_fine_hist, void *d_ws, hipStream_t st, bool const_k = true /*Switches::const_k*/);
hipError_t launch_kmer_histogram_fine(const uint8_t *d_bases, uint64_t n_bases, uint32_t k, int mode, uint64_t *d_bucket_counts,
                                      uint64_t *d_fine_hist, void *d_ws, hipStream_t st, bool const_k = true /*Switches::const_k*/,
                                      uint32_t bucket_bits = 6 /*6..8: d_bucket_counts[2^bucket_bits] and the per-workgroup rows the partition
                                      takes its cursors from go by that many top bits (a sharded count's senders); d_fine_hist stays 2^15*/);
bool       kmer_histogram_fine_bits_ok(uint32_t k, uint32_t bucket_bits);
hipError_t launch_kmer_histogram(const uint8_t *d_bases, uint64_t n_bases, uint32_t k, int mode,
                                 uint32_t bucket_bits, uint64_t *d_bucket_counts, void *d_ws, hipStream_t st,
                                 uint64_t sfx_mask = 0, uint64_t sfx_test = 0);
// keys are uint64 for k <= 32, 16-byte little-endian {lo,hi} for k in 33..64 (key_words 1 / 2)
hipError_t launch_kmer_partition(const uint8_t *d_bases, uint64_t n_bases, uint32_t k, int mode,
                                 uint32_t bucket_bits, const uint64_t *d_bucket_starts, void *d_keys,
                                 void *d_ws, hipStream_t st, uint64_t sfx_mask = 0, uint64_t sfx_test = 0,
                                 const uint64_t *d_soa_counts = nullptr /*64 buckets, 8-byte keys, <= 40 bits below the file: a file's region =
                                 u32[count] low words + u8[count] bits 32..39 (5 bytes per k-mer; launch_group_narrow(soa_hi_mask))*/,
                                 bool const_k = true /*Switches::const_k*/);

// ---- radix sort ------------------------------------------------------------
struct SortPlan {
  uint32_t radix_bits;      // digit width (9)
  uint32_t block;           // threads per workgroup (1024)
  uint32_t kpt;             // keys per thread (16; 16-byte keys: 8)
  uint32_t tile;            // block*kpt
  uint32_t mode;            // 0 = stable sort (decoupled look-back), 3 = grouping passes (not a sort: see radix_group_kernel; at most two digits)
  uint32_t hpc;             // 1: digits are DENSE RANKS of five homopolymer-free bases (make_hpc_group_plan), not bit fields; 2: only the high digit is
  uint32_t num_passes;
  uint32_t pass_shift[16];
  uint32_t pass_bits[16];
};

// Chooses digit widths for bits [begin_bit, end_bit); mode 0 (the caller sets 3 for grouping passes).
void   make_sort_plan(uint32_t begin_bit, uint32_t end_bit, SortPlan *plan);
// Grouping plan for homopolymer-compressed k-mers (`compress`): no base equals the one before it, so five bases after a
// known base take 3^5 = 243 of their 1024 bit patterns.  A digit is the dense, ORDER-PRESERVING rank of five bases given
// the base before them (10 key bits -> 0..242): `passes` (1 or 2) digits cover the 10*passes key bits above `low_bit`
// (which must be a whole number of bases above bit 0, with one more base above the top digit).  Grouping by these
// digits orders the keys exactly as grouping by the bits would, over 3^(5*passes) occupied sub-buckets instead of a
// 4^(5*passes) grid that is 94 % empty -- two digits do what took three stable passes.
void   make_hpc_group_plan(uint32_t low_bit, uint32_t passes, SortPlan *plan);
// ... and the two-digit form whose LOW digit is the plain eight bits of four bases (81 of 256 patterns): 18 key bits above `low_bit`,
// 3^9 = 19683 occupied sub-buckets of a 2^18 grid (SortPlan::hpc == 2)
void   make_hpc_mixed_plan(uint32_t low_bit, SortPlan *plan);
size_t sort_workspace_bytes(uint64_t n);

// Sorts n keys; returns where the result is via *result_in_alt.  d_error is a
// device uint32 the kernels set on a look-back timeout (checked by the caller).
hipError_t launch_radix_sort(void *d_keys, void *d_alt, uint64_t n, uint32_t key_words, const SortPlan &plan,
                             void *d_ws, size_t ws_bytes, uint32_t *d_error, int *result_in_alt,
                             hipStream_t st, hipEvent_t *pass_events /* 2 per pass or null */);
size_t     sort_header_bytes();
// Two grouping passes with NARROWED keys: uint64 keys whose bits above the first digit fit 32 bits leave the first pass as
// uint32 words without that digit (its value is where the key lies), the second pass and the finish move half the bytes,
// and the sub-bucket boundaries fall out of the second pass's look-back granules.  d_keys: uint64[n] in, uint32[n] out
// (over its first half); d_alt: room for n uint32; d_sub_starts: 2^(pass_bits[0] + pass_bits[1]) + 1 entries.
bool       sort_plan_narrows(const SortPlan &plan, uint64_t n, uint32_t key_words, bool on = true /*Switches::narrow*/);
// The high-digit-first form (files of a session with the fifteen-bit file histogram): launch_narrow_prepare fills one
// header per file (nb <= 64, sort_header_bytes() apart) from d_fine; every file then gets its header and a scratch area of
// narrow_scratch_bytes(n) that the caller has zeroed.  See mgc_sort.hip: nobody reads the
// keys for a digit histogram, and the sub-buckets come out in the order tr_index(., *tr_a, *tr_b) describes.
// d_prepared / d_scratch == nullptr: low digit first off one histogram read, scratch in d_ws, sub-buckets in key order.
size_t     narrow_scratch_bytes(uint64_t n);
hipError_t launch_narrow_prepare(const uint64_t *d_fine, uint32_t nb, const unsigned char *bits_a, const unsigned char *on, void *d_hdrs,
                                 hipStream_t st);
hipError_t launch_group_narrow(void *d_keys, void *d_alt, uint64_t n, const SortPlan &plan, void *d_ws, size_t ws_bytes,
                               uint32_t *d_error, uint64_t *d_sub_starts, hipStream_t st, hipEvent_t *pass_events /* 4 or null */,
                               void *d_prepared, void *d_scratch, uint32_t *tr_a, uint32_t *tr_b,
                               uint32_t soa_hi_mask = 0 /*nonzero: d_keys is the 5-byte layout of launch_kmer_partition(d_soa_counts); the mask of
                               the u8 array's payload bits (bits 32.. of the k-mer below the file)*/,
                               bool group_dbg = false /*Switches::group_dbg*/,
                               bool pipe = true /*Switches::group_pipe: the fetch a whole tile ahead (the 5-byte first pass, the second pass)*/,
                               uint32_t stagger = 0 /*Switches::pass_stagger: the first pass's workgroups start in that many groups, a tile's time shared between them*/);

// The same high-digit-first form for WHOLE keys (8-byte keys that leave more than 32 bits below their first digit: k = 27..32
// at the 10 Gbp scale; every 16-byte key): the high digit's histogram comes from the fifteen-bit file histogram
// (launch_narrow_prepare), the first pass counts the low digit as it goes, the boundaries fall out of the second pass's
// look-back granules -- neither the 8/16 B per k-mer digit-histogram read nor the boundary search over the grouped keys
// happens.  d_keys: n keys in, the grouped keys out (same place); d_alt: room for n keys; d_scratch: wide_scratch_bytes(n,
// key_words), zeroed by the caller; sub-buckets in tr_index(., *tr_a, *tr_b) order.
// `compress`: the plan's digits are dense ranks (make_hpc_group_plan, two digits); the headers then come from
// launch_hpc_prepare (the histogram of launch_kmer_histogram_hpc; `on`: one bit per bucket, nb <= 256)
bool       sort_plan_wide_msd(const SortPlan &plan, uint64_t n, bool on = true /*Switches::wide_msd*/);
hipError_t launch_hpc_prepare(const uint64_t *d_fine_hpc, uint32_t bucket_bits, const uint64_t on[4], void *d_hdrs, hipStream_t st);
size_t     wide_scratch_bytes(uint64_t n, uint32_t key_words);
hipError_t launch_group_wide(void *d_keys, void *d_alt, uint64_t n, uint32_t key_words, const SortPlan &plan, uint32_t *d_error,
                             uint64_t *d_sub_starts, hipStream_t st, hipEvent_t *pass_events /* 4 or null */, void *d_prepared,
                             void *d_scratch, uint32_t *tr_a, uint32_t *tr_b,
                             bool k96 = false /* d_keys / d_alt hold 12-byte K96 records (launch_kmer_partition(d_soa_counts) at k = 33..51) */);

// ---- run-length count ------------------------------------------------------
size_t     rle_workspace_bytes(uint64_t n);
hipError_t launch_rle_count(const void *d_sorted, uint64_t n, uint32_t key_words, void *d_ws, hipStream_t st);
// after launch_rle_count + stream sync: number of distinct keys sits at ws[0]
hipError_t rle_read_total(const void *d_ws, uint64_t *n_distinct, hipStream_t st);
// d_out_base (device, optional): offset added to every output slot
hipError_t launch_rle_emit(const void *d_sorted, uint64_t n, uint32_t key_words, void *d_ws, void *d_unique,
                           uint32_t *d_counts, hipStream_t st, const uint64_t *d_out_base = nullptr);

// ---- sub-bucket finish (LDS sort of the low bits + fused run-length count) ---------------
uint64_t   finish_capacity_for(uint32_t key_words);     // largest sub-bucket the LDS kernels accept
uint64_t   finish_target_for(uint32_t key_words, const Switches &sw);       // average sub-bucket size to aim for
// The distinct-sized count (hash_count_stream_kernel, round 6): narrowed files whose suffix fits its packed entry take sub-buckets of
// up to finish_stream_capacity() keys (twice the others' average); its table holds finish_stream_distinct() distinct suffixes, a
// sub-bucket with more goes on the retry list (launch_finish_retry).
// Whole 8-byte k-mers on the high-digit-first passes (k = 24..32, `compress`; narrow = false) take the same kernel with 64-bit entries
// (suffixes of up to 52 bits); their retry list goes through the streaming kernel of the oversized sub-buckets.
bool       finish_stream_ok(uint32_t key_words, uint32_t low_bits, bool narrow = true);
uint64_t   finish_stream_capacity();
uint64_t   finish_stream_target(const Switches &sw);
uint64_t   finish_stream_distinct();
hipError_t launch_subbucket_bounds(const void *d_keys, uint64_t n, uint32_t key_words, uint32_t low, uint32_t top_bits,
                                   uint64_t *d_starts /*[2^top+1]*/, uint64_t *d_max /*atomicMax target*/,
                                   uint32_t *d_large_list /*[2^top]: sub-buckets above the small-kernel capacity*/,
                                   uint64_t *d_large_count /*zeroed by the caller*/,
                                   uint32_t *d_nonempty_list /*[2^top]*/, uint64_t *d_nonempty_count /*zeroed by the caller*/,
                                   hipStream_t st);
hipError_t launch_subbucket_max(const uint64_t *d_starts, uint32_t key_words, uint32_t low, uint32_t top_bits, uint64_t *d_max,
                                uint32_t *d_large_list, uint64_t *d_large_count, uint32_t *d_nonempty_list, uint64_t *d_nonempty_count,
                                hipStream_t st, uint64_t small_cap = 0 /*0: the capacity of the kernels (key_words, low) selects*/);
                                // the list half of launch_subbucket_bounds (boundaries already known)
// sub-buckets above finish_capacity_for(): can they be streamed through the hash-count tables (distinct suffixes fit)?
bool       finish_can_stream(uint32_t key_words, uint32_t low_bits);
hipError_t launch_finish_probe(const void *d_keys, uint32_t key_words, const uint64_t *d_starts, uint32_t low_bits,
                               uint64_t n_large, const uint32_t *d_large_list, uint32_t *d_file_fail /*set to 1: no*/, hipStream_t st,
                               uint64_t stream_max /*Switches::stream_max*/, bool narrow = false /*d_keys: uint32 narrowed keys*/);
hipError_t launch_finish_file(void *d_keys, uint32_t key_words, const uint64_t *d_starts, uint64_t ng, uint32_t low_bits,
                              uint64_t n_large, const uint32_t *d_large_list, uint32_t *d_cnt_tmp, uint64_t *d_group_distinct,
                              bool stream /*large list -> streaming hash-count*/, void *d_alt /*room for the file's keys*/,
                              hipStream_t st_huge /*where that kernel is launched (st, or a stream forked from it)*/,
                              const uint32_t *d_nonempty_list, const uint64_t *d_nonempty_count /*the hash kernels visit only these;
                              d_group_distinct must be zero for the others*/, hipStream_t st,
                              bool narrow = false /*d_keys/d_alt: uint32 narrowed keys; the distinct SUFFIXES go back in place*/,
                              uint32_t tr_a = 0, uint32_t tr_b = 0 /*launch_group_narrow's / launch_group_wide's (whole keys: hash paths only)*/,
                              uint64_t max_sub = 0 /*the file's largest sub-bucket, if known: small files take smaller tables*/,
                              uint64_t n_keys = 0 /*keys of the file, if known: the narrowed hash-count takes several sub-buckets per iteration by their average*/,
                              uint32_t *d_retry_list = nullptr /*[ng] + a zeroed counter: with both, dense narrowed files take hash_count_multi_kernel*/,
                              uint64_t *d_retry_count = nullptr,
                              bool k96 = false /*d_keys: 12-byte K96 records (key_words 2; oversized sub-buckets only with `stream`)*/,
                              int hash_multi = -1 /*Switches::hash_multi*/, bool hash_dbg = false /*Switches::hash_dbg*/,
                              uint64_t stream_cap = 0 /*nonzero (narrowed dense files, finish_stream_ok): hash_count_stream_kernel counts the
                              sub-buckets of up to that many keys; the ones with too many distinct suffixes go on d_retry_list*/,
                              void *d_huge_ws = nullptr, size_t huge_ws_bytes = 0, uint64_t huge_ws_keys = 0, uint32_t *d_error = nullptr /*(d_error: the count's look-back / chain time-out flag) finish_huge_workspace_bytes(huge_ws_keys >= n_keys), one per stream the streaming
                              kernels run on: with it (8-byte and narrowed keys) a GIGANTIC sub-bucket -- above 65536 keys -- is counted in slices by
                              many workgroups instead of one*/);
size_t     finish_huge_workspace_bytes(uint64_t n_keys);
// the sub-buckets hash_count_stream_kernel put on the retry list (their number is on the device: the caller brings it back first)
hipError_t launch_finish_retry(void *d_keys32, const uint64_t *d_starts, uint64_t ng, uint32_t low_bits, uint32_t *d_cnt_tmp,
                               uint64_t *d_group_distinct, uint32_t tr_a, uint32_t tr_b, const uint32_t *d_retry_list,
                               const uint64_t *d_retry_count, uint64_t n_retry, uint64_t stream_cap, hipStream_t st,
                               bool narrow = true /*false: d_keys32 holds whole 8-byte k-mers*/, void *d_alt = nullptr /*whole k-mers: room for the file's keys*/);
size_t     finish_scan_scratch_bytes(uint64_t ng_total);
hipError_t launch_finish_scan(uint64_t *d_group /*[ng_total+1]*/, uint64_t ng_total, void *d_scratch, hipStream_t st);
hipError_t launch_compact_groups(const void *d_keys, uint32_t key_words, const uint32_t *d_cnt_tmp, const uint64_t *d_starts,
                                 const uint64_t *d_offs, uint64_t ng, void *d_out_keys, uint32_t *d_out_counts, hipStream_t st,
                                 uint32_t tr_a = 0, uint32_t tr_b = 0 /*launch_group_wide's: d_offs goes by tr_index(sub-bucket)*/,
                                 const uint32_t *d_nonempty_list = nullptr, uint64_t n_nonempty = 0 /*visit only these (host count)*/);
// narrowed files: k-mer = base | sub-bucket << low_bits | suffix
hipError_t launch_compact_groups_narrow(const void *d_keys32, const uint32_t *d_cnt_tmp, const uint64_t *d_starts, const uint64_t *d_offs,
                                        uint64_t ng, uint64_t base, uint32_t low_bits, void *d_out_keys, uint32_t *d_out_counts, hipStream_t st,
                                        uint32_t tr_a, uint32_t tr_b, const uint32_t *d_nonempty_list = nullptr, uint64_t n_nonempty = 0);
// K96 records (k = 33..51: the bits below the file, 12 bytes) -> whole 16-byte k-mers; base_lo / base_hi: the file's bits in their place
hipError_t launch_compact_groups_k96(const void *d_keys96, const uint32_t *d_cnt_tmp, const uint64_t *d_starts, const uint64_t *d_offs,
                                     uint64_t ng, uint64_t base_lo, uint64_t base_hi, void *d_out_keys, uint32_t *d_out_counts, hipStream_t st,
                                     uint32_t tr_a, uint32_t tr_b, const uint32_t *d_nonempty_list = nullptr, uint64_t n_nonempty = 0);
hipError_t launch_widen_k96(const void *d_keys96, uint64_t n, uint64_t base_lo, uint64_t base_hi, void *d_out128, hipStream_t st);
hipError_t launch_widen_groups(const void *d_keys32, const uint64_t *d_starts, uint64_t ng, uint64_t base, uint32_t low_bits, void *d_out64,
                               hipStream_t st, uint32_t tr_a, uint32_t tr_b);      // tr_a != 0: the result is NOT in key order
hipError_t launch_store_u64(uint64_t *d_dst, const uint64_t *d_src, hipStream_t st);
hipError_t launch_sum_u64(const uint64_t *d_in, uint64_t n, uint64_t *d_out, hipStream_t st);    // *d_out <- sum of d_in[0, n)

hipError_t launch_block_offsets(const void *d_unique, uint64_t n_distinct, uint32_t key_words, uint32_t w_data,
                                uint64_t n_prefix, uint64_t *d_block_start, hipStream_t st);

// ---- database blocks encoded on the device (mgc_encode.hip; layout: mdb_layout.h) -------------------------------
// rel_start[i] = first key >= (prefix_begin + i) << w_data for i in [0, n_blocks] (n_prefix_total = 2^wPrefix)
hipError_t launch_block_offsets_range(const void *d_keys, uint64_t n, uint32_t key_words, uint32_t w_data, uint64_t prefix_begin,
                                      uint64_t n_blocks, uint64_t n_prefix_total, uint64_t *d_rel_start, hipStream_t st);
hipError_t launch_encode_sizes(const void *d_keys, uint32_t key_words, const uint64_t *d_bs, uint64_t n_blocks, uint32_t suffix_size,
                               uint32_t label_size, uint64_t *d_blk_bytes, uint64_t *d_blk_vbase, uint32_t *d_blk_bb, hipStream_t st);
hipError_t launch_encode_chunk(const void *d_keys, const uint32_t *d_counts, uint32_t key_words, const uint64_t *d_bs,
                               const uint64_t *d_blk_pos, const uint64_t *d_blk_vbase, const uint32_t *d_blk_bb,
                               uint64_t b0, uint64_t b1, uint64_t n_kmers_chunk, uint64_t prefix_of_block0,
                               uint32_t suffix_size, uint32_t label_size, uint64_t label, void *d_img, hipStream_t st);
uint32_t   value_hist_small_bins();
hipError_t launch_value_hist(const uint32_t *d_counts, uint64_t n, uint64_t *d_hist, uint32_t *d_big_list, uint64_t big_cap,
                             uint64_t *d_big_n, hipStream_t st);

// ---- database blocks decoded on the device (mgc_decode.hip): d_file = the data file's bytes (+ 16 bytes of slack),
// d_blocks = mdb_raw_block[n_blocks] (include/meryl_db.h, mdb_reader_raw_file); one thread per block; *d_err != 0: a corrupt block
hipError_t launch_decode_blocks(const void *d_file, const void *d_blocks, uint64_t n_blocks, uint32_t suffix_size, uint32_t label_size,
                                uint32_t key_words, void *d_keys, uint32_t *d_counts, uint32_t *d_err, hipStream_t st);

// ---- merge of two sorted distinct (k-mer, value) streams (mgc_merge.hip) ----------------------------------------
// op: 0 union-sum, 1 union-min, 2 union-max, 3 intersect-sum, 4 intersect-min, 5 intersect-max
size_t     merge_workspace_bytes(uint64_t na, uint64_t nb);
//     6 intersect (first input's value), 7 subtract, 8 difference, 9 symmetric-difference (mgc_merge.hip)
// cA / cB: the values -- needed by the count pass only when what is written depends on them (op 7)
hipError_t launch_merge_count(const void *dA, uint64_t na, const void *dB, uint64_t nb, uint32_t key_words, int op, void *d_ws,
                              hipStream_t st, const uint32_t *cA = nullptr, const uint32_t *cB = nullptr);
hipError_t merge_read_total(const void *d_ws, uint64_t *n_out, hipStream_t st);       // synchronises the stream
hipError_t launch_merge_emit(const void *dA, const uint32_t *cA, uint64_t na, const void *dB, const uint32_t *cB, uint64_t nb,
                             uint32_t key_words, int op, void *d_ws, void *d_out_keys, uint32_t *d_out_counts, hipStream_t st);
// one stream through a value transform (fop 0..5 filters against `constant`: less-than, greater-than, at-least, at-most, equal-to,
// not-equal-to; 6..11 arithmetic: increase, decrease, multiply, divide, divide-round, modulo; 12: keep where d_flags[i] == 1),
// k-mers whose new value is 0 dropped; two passes like the merge (count -> merge_read_total -> emit)
size_t     select_workspace_bytes(uint64_t n);
hipError_t launch_select_count(const void *d_keys, const uint32_t *d_vals, const uint32_t *d_flags, uint64_t n, uint32_t key_words, int fop,
                               uint64_t constant, void *d_ws, hipStream_t st);
hipError_t launch_select_emit(const void *d_keys, const uint32_t *d_vals, const uint32_t *d_flags, uint64_t n, uint32_t key_words, int fop,
                              uint64_t constant, void *d_ws, void *d_out_keys, uint32_t *d_out_vals, hipStream_t st);
hipError_t launch_fill_u32(uint32_t *d, uint64_t n, uint32_t v, hipStream_t st);
// *d_out <- 1 + index of the last '.' in bases[0, n), 0 if none
hipError_t launch_last_breaker(const uint8_t *d_bases, uint64_t n, uint64_t *d_out, hipStream_t st);

// ---- homopolymer compression -------------------------------------------------
size_t     hpc_workspace_bytes(uint64_t n);
// compressed length lands in the first uint64 of the workspace
hipError_t launch_homopoly_compress(const uint8_t *d_in, uint64_t n, uint8_t *d_out, void *d_ws, hipStream_t st);

// ---- FASTA / FASTQ text -> base stream (mgc_parse.hip) ----------------------------------------
size_t     text_parse_state_bytes();
size_t     text_parse_workspace_bytes(uint64_t n);            // for one chunk of n text bytes
// appends the bases of one text chunk to d_out at the running length kept in d_state
hipError_t launch_text_parse(const uint8_t *d_text, uint64_t n, int fastq, void *d_state, void *d_ws, uint8_t *d_out, hipStream_t st);
// what: 0 begin file, 1 end file (breaker), 2 roll the current file back, 3 reset
hipError_t launch_text_file_op(void *d_state, uint8_t *d_out, int what, hipStream_t st);
// the stream now holds new_len bytes (bases appended by a plain copy, or the tail kept after a batch was cut off);
// rebase_file: the open file's rollback point moves to 0 (what it wrote before the cut is gone)
hipError_t launch_text_set_len(void *d_state, uint64_t new_len, int rebase_file, hipStream_t st);

hipError_t launch_synth_reads(uint64_t seed, uint64_t genome_len, uint64_t first_read, uint64_t n_reads,
                              uint32_t read_len, uint32_t sub_rate_ppm, uint32_t n_rate_ppm,
                              uint32_t repeat_ppm, uint32_t repeat_unit, uint32_t repeat_families,
                              uint8_t *d_out, hipStream_t st);

// code objects load lazily at the first launch of a kernel of their translation unit: these touch one kernel each
hipError_t warm_kmer(); hipError_t warm_sort(); hipError_t warm_finish(); hipError_t warm_encode(); hipError_t warm_scan();
}  // namespace mgc
