// mgc_api.cpp -- C-ABI layer (include/meryl_gpu_count.h) over the gfx950 kernels.
//
// Host-side mirror of the reference's counting engine:
//   mgc_configure_counting  <-> merylOperation::configureCounting  src/meryl/merylOp-count.C:300-403
//   mgc_open/push/count/finish <-> merylOperation::countThreads     src/meryl/merylOp-countThreads.C:385-474
// There is NO CPU fallback in here: every compute entry point launches HIP
// kernels and fails with MGC_EHIP if the device or the code object is missing.
#include "../../include/meryl_gpu_count.h"
#include "mgc_device.h"
#include "mgc_session.hpp"
#include "mgc_runs.hpp"

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <zlib.h>

namespace mgc {
std::string &thread_last_error() {
  thread_local std::string e;
  return e;
}
}  // namespace mgc
using mgc::set_err;

namespace {

// ---------------------------------------------------------------------------
// configureCounting restated (src/meryl/merylOp-count.C:118-403).  This is the
// product's own copy; oracle/oracle_count.c holds an independent restatement
// used by the tests to cross-check it.
// ---------------------------------------------------------------------------

uint64_t number_of_bits64(uint64_t v) {      // countNumberOfBits64 [meryl-utility]
  uint64_t b = 0;
  while (v) { b++; v >>= 1; }
  return b;
}

// findExpectedSimpleSize, merylOp-count.C:118-165; lowBits_t is uint16 (merylOp-countSimple.C:41-48)
uint64_t expected_simple_size(uint32_t k, uint64_t n_est, uint32_t suffix_len) {
  if (2 * k - 2 * suffix_len > 42) return UINT64_MAX;                            // :142
  const uint64_t low_bits  = 16;
  const uint64_t n_entries = (uint64_t)1 << (2 * k - 2 * suffix_len);            // :124
  const uint64_t exp_max   = (uint64_t)(0.004 * (double)n_est);                  // :126
  const uint64_t exp_bits  = number_of_bits64(exp_max) + 1;                      // :127
  const uint64_t extra     = (exp_bits < low_bits) ? 0 : (exp_bits - low_bits);  // :128
  return (n_entries * low_bits + n_entries * extra) / 8;                         // :130-132
}

// findBestPrefixSize, merylOp-count.C:173-227
void best_prefix_size(const mgc_count_config &c, uint64_t n_est, uint64_t mem_allowed, uint32_t *best,
                      uint64_t *mem_used) {
  const uint32_t k = c.k;
  const uint32_t seg_bits = c.page_size * 8, seg_bytes = c.page_size;            // pagesPerSegment()==1
  *best = 0;
  *mem_used = UINT64_MAX;
  for (uint32_t wp = 1; wp < 2 * k - 1; wp++) {                                  // :197
    const uint64_t n_prefix = (uint64_t)1 << wp;
    const uint64_t kpp = n_est / n_prefix + 1;                                   // :199
    const uint64_t kps = seg_bits / (2 * k - wp);                                // :200
    const uint64_t spp = kpp / kps + 1;                                          // :201
    if (wp + number_of_bits64(spp) + number_of_bits64(seg_bytes) >= 64) break;   // :203
    const uint64_t struct_mem = (uint64_t)c.sizeof_count_array * n_prefix + 8 * n_prefix * spp;   // :206-207
    const uint64_t data_min   = n_prefix * seg_bytes;                            // :208
    const uint64_t total      = struct_mem + n_prefix * spp * seg_bytes;         // :209-210
    if (struct_mem + data_min > mem_allowed) break;                              // :216
    if ((wp > 9) && (total + (uint64_t)16 * wp * 1024 * 1024 < *mem_used)) {     // :219
      *mem_used = total;
      *best = wp;
    }
    if (total > (uint64_t)16 * *mem_used) break;                                 // :224 (uint64 wrap kept)
  }
}

}  // namespace

extern "C" uint32_t mgc_version(void) { return (0u << 16) | 3u; }

extern "C" int mgc_configure_counting(mgc_count_config *c) {
  if (!c) return MGC_EINVAL;
  if (c->k == 0 || c->k > 64) {                                                  // :311-312 (fatal there)
    set_err(nullptr, "ERROR: Kmer size not supplied with modifier k=<kmer-size>.");
    return MGC_EINVAL;
  }
  if (c->mode < 0 || c->mode > 2) return MGC_EINVAL;
  if (c->page_size == 0) c->page_size = 4096;
  if (c->sizeof_count_array == 0) c->sizeof_count_array = 3232;

  c->use_simple = 0; c->w_prefix = 0; c->n_prefix = 0; c->w_data = 0; c->n_batches = 1; c->memory_used = 0;

  const uint64_t mem_simple = expected_simple_size(c->k, c->n_kmers_estimate, c->count_suffix_length);   // :340
  uint64_t mem_complex = UINT64_MAX;
  uint32_t best = 0, n_batches = 1;

  if (c->k > 5) {                                                                // :353
    for (n_batches = 1; mem_complex > c->memory_allowed; n_batches++) {          // :354-355
      best_prefix_size(*c, c->n_kmers_estimate / n_batches, c->memory_allowed, &best, &mem_complex);
      if (n_batches > (1u << 20)) {
        set_err(nullptr, "configureCounting: no prefix size fits in %lu bytes", (unsigned long)c->memory_allowed);
        return MGC_EINVAL;
      }
    }
    c->w_prefix = best;                                                          // findBestValues :273-276
    c->n_prefix = (uint64_t)1 << best;
    c->w_data   = 2 * c->k - best;
  }
  if ((mem_simple < mem_complex) && (mem_simple < c->memory_allowed)) {          // :368-372
    c->use_simple = 1; c->memory_used = mem_simple;
  } else {
    c->use_simple = 0; c->memory_used = mem_complex;
  }
  if (c->count_suffix_length > 0) { c->use_simple = 1; c->memory_used = mem_simple; }   // :379-382
  c->n_batches = n_batches;
  return MGC_OK;
}

extern "C" int mgc_format_configured_line(const mgc_count_config *c, char *buf, size_t buflen) {
  if (!c || !buf) return MGC_EINVAL;
  const uint64_t m = (c->memory_used < c->memory_allowed) ? c->memory_used : c->memory_allowed;
  snprintf(buf, buflen, "Configured %s mode for %.3f GB memory per batch, and up to %u batch%s.",   // :398-401
           c->use_simple ? "simple" : "complex", m / 1024.0 / 1024.0 / 1024.0, c->n_batches,
           (c->n_batches == 1) ? "" : "es");
  return MGC_OK;
}

// ---------------------------------------------------------------------------
// Stateless device operators
// ---------------------------------------------------------------------------
namespace {
int hip_rc(hipError_t e, const char *what) {
  if (e == hipSuccess) return MGC_OK;
  set_err(nullptr, "%s: %s", what, hipGetErrorString(e));
  return (e == hipErrorOutOfMemory) ? MGC_ENOMEM : MGC_EHIP;
}
bool key_args_ok(uint32_t k, int mode, uint32_t bucket_bits) {
  if (k == 0 || k > 64) { set_err(nullptr, "k=%u out of range (1..64)", k); return false; }
  if (mode < 0 || mode > 2) { set_err(nullptr, "bad mode %d", mode); return false; }
  if (bucket_bits > MGC_MAX_BUCKET_BITS || bucket_bits > 2 * k) { set_err(nullptr, "bad bucket_bits %u", bucket_bits); return false; }
  return true;
}
}  // namespace

extern "C" size_t mgc_dev_partition_workspace_bytes(uint32_t bucket_bits) {
  return mgc::kp_workspace_bytes(bucket_bits);
}

extern "C" int mgc_dev_kmer_histogram(const uint8_t *d_bases, uint64_t n_bases, uint32_t k, int mode,
                                      uint32_t bucket_bits, uint64_t *d_bucket_counts, void *d_ws, size_t ws_bytes,
                                      void *stream) {
  if (!key_args_ok(k, mode, bucket_bits)) return MGC_EINVAL;
  if ((!d_bases && n_bases) || !d_bucket_counts || !d_ws || ws_bytes < mgc::kp_workspace_bytes(bucket_bits)) return MGC_EINVAL;
  return hip_rc(mgc::launch_kmer_histogram(d_bases, n_bases, k, mode, bucket_bits, d_bucket_counts, d_ws,
                                           (hipStream_t)stream), "kmer_histogram");
}

extern "C" int mgc_dev_kmer_partition(const uint8_t *d_bases, uint64_t n_bases, uint32_t k, int mode,
                                      uint32_t bucket_bits, const uint64_t *d_bucket_starts, void *d_keys,
                                      void *d_ws, size_t ws_bytes, void *stream) {
  if (!key_args_ok(k, mode, bucket_bits)) return MGC_EINVAL;
  if ((!d_bases && n_bases) || !d_bucket_starts || !d_ws || ws_bytes < mgc::kp_workspace_bytes(bucket_bits)) return MGC_EINVAL;
  return hip_rc(mgc::launch_kmer_partition(d_bases, n_bases, k, mode, bucket_bits, d_bucket_starts, d_keys, d_ws,
                                           (hipStream_t)stream), "kmer_partition");
}

// the histogram of a sharded count's senders: k-mers per bucket (2^bucket_bits, 6..8) for the routing plan + the per-workgroup rows
// mgc_dev_kmer_partition takes its cursors from + k-mers per top FIFTEEN bits (d_fine_hist[2^15]) for the owners' first grouping digit
extern "C" int mgc_dev_kmer_histogram_fine(const uint8_t *d_bases, uint64_t n_bases, uint32_t k, int mode, uint32_t bucket_bits,
                                           uint64_t *d_bucket_counts, uint64_t *d_fine_hist, void *d_ws, size_t ws_bytes, void *stream) {
  if (!key_args_ok(k, mode, bucket_bits) || !mgc::kmer_histogram_fine_bits_ok(k, bucket_bits)) return MGC_EINVAL;
  if ((!d_bases && n_bases) || !d_bucket_counts || !d_fine_hist || !d_ws || ws_bytes < mgc::kp_workspace_bytes(bucket_bits)) return MGC_EINVAL;
  const mgc::Switches sw = mgc::read_switches();
  return hip_rc(mgc::launch_kmer_histogram_fine(d_bases, n_bases, k, mode, d_bucket_counts, d_fine_hist, d_ws, (hipStream_t)stream,
                                                sw.const_k, bucket_bits), "kmer_histogram_fine");
}

extern "C" size_t mgc_dev_sort_workspace_bytes(uint64_t n) { return mgc::sort_workspace_bytes(n) + 256; }

static int dev_radix_passes(void *d_keys, void *d_alt, uint64_t n, uint32_t key_words, uint32_t begin_bit, uint32_t end_bit, void *d_ws,
                           size_t ws_bytes, int *result_in_alt, void *stream, bool group);

extern "C" int mgc_dev_radix_sort(void *d_keys, void *d_alt, uint64_t n, uint32_t key_words, uint32_t begin_bit,
                                  uint32_t end_bit, void *d_ws, size_t ws_bytes, int *result_in_alt, void *stream) {
  return dev_radix_passes(d_keys, d_alt, n, key_words, begin_bit, end_bit, d_ws, ws_bytes, result_in_alt, stream, false);
}
// the grouping passes of the count path as a bare operator (its own entry point: an environment switch inside mgc_dev_radix_sort
// used to change that operator's contract -- ADVICE r5)
extern "C" int mgc_dev_radix_group(void *d_keys, void *d_alt, uint64_t n, uint32_t key_words, uint32_t begin_bit,
                                   uint32_t end_bit, void *d_ws, size_t ws_bytes, int *result_in_alt, void *stream) {
  return dev_radix_passes(d_keys, d_alt, n, key_words, begin_bit, end_bit, d_ws, ws_bytes, result_in_alt, stream, true);
}

static int dev_radix_passes(void *d_keys, void *d_alt, uint64_t n, uint32_t key_words, uint32_t begin_bit, uint32_t end_bit, void *d_ws,
                           size_t ws_bytes, int *result_in_alt, void *stream, bool group) {
  if (!result_in_alt || begin_bit > end_bit || (key_words != 1 && key_words != 2) || end_bit > 64 * key_words) return MGC_EINVAL;
  *result_in_alt = 0;
  if (n == 0 || begin_bit == end_bit) return MGC_OK;
  if (!d_keys || !d_alt || !d_ws || ws_bytes < mgc::sort_workspace_bytes(n) + 256) return MGC_EINVAL;
  mgc::SortPlan plan;
  mgc::make_sort_plan(begin_bit, end_bit, &plan);
  if (group) plan.mode = 3;
  // the last 256 bytes of the workspace hold the look-back error word
  uint32_t *d_err = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(d_ws) + mgc::sort_workspace_bytes(n));
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(d_err, 0, 4, st);
  if (e != hipSuccess) return hip_rc(e, "radix_sort memset");
  e = mgc::launch_radix_sort(d_keys, d_alt, n, key_words, plan, d_ws, ws_bytes - 256, d_err, result_in_alt, st, nullptr);
  if (e != hipSuccess) return hip_rc(e, "radix_sort");
  uint32_t h_err = 0;
  e = hipMemcpyAsync(&h_err, d_err, 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return hip_rc(e, "radix_sort sync");
  if (h_err) { set_err(nullptr, "radix sort look-back timed out"); return MGC_ETIMEOUT; }
  return MGC_OK;
}

extern "C" size_t mgc_dev_rle_workspace_bytes(uint64_t n) { return mgc::rle_workspace_bytes(n); }

extern "C" int mgc_dev_rle_count(const void *d_sorted, uint64_t n, uint32_t key_words, void *d_ws, size_t ws_bytes,
                                 uint64_t *n_distinct, void *stream) {
  if (!n_distinct || !d_ws || ws_bytes < mgc::rle_workspace_bytes(n) || (!d_sorted && n) ||
      (key_words != 1 && key_words != 2)) return MGC_EINVAL;
  hipError_t e = mgc::launch_rle_count(d_sorted, n, key_words, d_ws, (hipStream_t)stream);
  if (e != hipSuccess) return hip_rc(e, "rle_count");
  return hip_rc(mgc::rle_read_total(d_ws, n_distinct, (hipStream_t)stream), "rle_count sync");
}

extern "C" int mgc_dev_rle_emit(const void *d_sorted, uint64_t n, uint32_t key_words, void *d_ws, size_t ws_bytes,
                                void *d_unique, uint32_t *d_counts, void *stream) {
  if (!d_ws || ws_bytes < mgc::rle_workspace_bytes(n) || (n && (!d_sorted || !d_unique || !d_counts)) ||
      (key_words != 1 && key_words != 2)) return MGC_EINVAL;
  return hip_rc(mgc::launch_rle_emit(d_sorted, n, key_words, d_ws, d_unique, d_counts, (hipStream_t)stream), "rle_emit");
}

extern "C" int mgc_dev_block_offsets(const void *d_unique, uint64_t n_distinct, uint32_t key_words, uint32_t w_data,
                                     uint64_t n_prefix, uint64_t *d_block_start, void *stream) {
  if (!d_block_start || (key_words != 1 && key_words != 2) || w_data >= 64 * key_words || (n_distinct && !d_unique))
    return MGC_EINVAL;
  return hip_rc(mgc::launch_block_offsets(d_unique, n_distinct, key_words, w_data, n_prefix, d_block_start,
                                          (hipStream_t)stream), "block_offsets");
}

extern "C" size_t mgc_dev_merge_workspace_bytes(uint64_t na, uint64_t nb) { return mgc::merge_workspace_bytes(na, nb); }

extern "C" int mgc_dev_merge_count(const void *dA, uint64_t na, const void *dB, uint64_t nb, uint32_t key_words, int op, void *d_ws,
                                   size_t ws_bytes, uint64_t *n_out, void *stream) {
  if (!n_out || !d_ws || ws_bytes < mgc::merge_workspace_bytes(na, nb) || (na && !dA) || (nb && !dB) ||
      (key_words != 1 && key_words != 2) || op < 0 || op > MGC_MERGE_SYMMETRIC_DIFFERENCE || op == MGC_MERGE_SUBTRACT) return MGC_EINVAL;
  hipError_t e = mgc::launch_merge_count(dA, na, dB, nb, key_words, op, d_ws, (hipStream_t)stream);
  if (e != hipSuccess) return hip_rc(e, "merge_count");
  return hip_rc(mgc::merge_read_total(d_ws, n_out, (hipStream_t)stream), "merge_count sync");
}

extern "C" int mgc_dev_merge_count_values(const void *dA, const uint32_t *cA, uint64_t na, const void *dB, const uint32_t *cB, uint64_t nb,
                                          uint32_t key_words, int op, void *d_ws, size_t ws_bytes, uint64_t *n_out, void *stream) {
  if (!n_out || !d_ws || ws_bytes < mgc::merge_workspace_bytes(na, nb) || (na && (!dA || !cA)) || (nb && (!dB || !cB)) ||
      (key_words != 1 && key_words != 2) || op < 0 || op > MGC_MERGE_SYMMETRIC_DIFFERENCE) return MGC_EINVAL;
  hipError_t e = mgc::launch_merge_count(dA, na, dB, nb, key_words, op, d_ws, (hipStream_t)stream, cA, cB);
  if (e != hipSuccess) return hip_rc(e, "merge_count");
  return hip_rc(mgc::merge_read_total(d_ws, n_out, (hipStream_t)stream), "merge_count sync");
}

extern "C" size_t mgc_dev_select_workspace_bytes(uint64_t n) { return mgc::select_workspace_bytes(n); }

extern "C" int mgc_dev_select_count(const void *d_keys, const uint32_t *d_values, uint64_t n, uint32_t key_words, int value_op, uint64_t constant,
                                    void *d_ws, size_t ws_bytes, uint64_t *n_out, void *stream) {
  if (!n_out || !d_ws || ws_bytes < mgc::select_workspace_bytes(n) || (n && (!d_keys || !d_values)) || (key_words != 1 && key_words != 2) ||
      value_op < 0 || value_op > MGC_VALUE_MODULO) return MGC_EINVAL;
  hipError_t e = mgc::launch_select_count(d_keys, d_values, nullptr, n, key_words, value_op, constant, d_ws, (hipStream_t)stream);
  if (e != hipSuccess) return hip_rc(e, "select_count");
  return hip_rc(mgc::merge_read_total(d_ws, n_out, (hipStream_t)stream), "select_count sync");
}

extern "C" int mgc_dev_select_emit(const void *d_keys, const uint32_t *d_values, uint64_t n, uint32_t key_words, int value_op, uint64_t constant,
                                   void *d_ws, size_t ws_bytes, void *d_keys_out, uint32_t *d_values_out, void *stream) {
  if (!d_ws || ws_bytes < mgc::select_workspace_bytes(n) || (n && (!d_keys || !d_values)) || (key_words != 1 && key_words != 2) ||
      value_op < 0 || value_op > MGC_VALUE_MODULO) return MGC_EINVAL;
  return hip_rc(mgc::launch_select_emit(d_keys, d_values, nullptr, n, key_words, value_op, constant, d_ws, d_keys_out, d_values_out,
                                        (hipStream_t)stream), "select_emit");
}

extern "C" int mgc_dev_merge_emit(const void *dA, const uint32_t *cA, uint64_t na, const void *dB, const uint32_t *cB, uint64_t nb,
                                  uint32_t key_words, int op, void *d_ws, size_t ws_bytes, void *d_keys_out, uint32_t *d_counts_out,
                                  void *stream) {
  if (!d_ws || ws_bytes < mgc::merge_workspace_bytes(na, nb) || (na && (!dA || !cA)) || (nb && (!dB || !cB)) ||
      (key_words != 1 && key_words != 2) || op < 0 || op > MGC_MERGE_SYMMETRIC_DIFFERENCE) return MGC_EINVAL;
  return hip_rc(mgc::launch_merge_emit(dA, cA, na, dB, cB, nb, key_words, op, d_ws, d_keys_out, d_counts_out, (hipStream_t)stream),
                "merge_emit");
}

extern "C" size_t mgc_dev_homopoly_workspace_bytes(uint64_t n) { return mgc::hpc_workspace_bytes(n); }

extern "C" int mgc_dev_homopoly_compress(const uint8_t *d_in, uint64_t n, uint8_t *d_out, uint64_t *n_out, void *d_ws,
                                         size_t ws_bytes, void *stream) {
  if (!n_out || !d_ws || ws_bytes < mgc::hpc_workspace_bytes(n) || (n && (!d_in || !d_out))) return MGC_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = mgc::launch_homopoly_compress(d_in, n, d_out, d_ws, st);
  if (e == hipSuccess) e = hipMemcpyAsync(n_out, d_ws, sizeof(uint64_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  return hip_rc(e, "homopoly_compress");
}

extern "C" int mgc_dev_synth_reads(uint64_t seed, uint64_t genome_len, uint64_t first_read, uint64_t n_reads,
                                   uint32_t read_len, uint32_t sub_rate_ppm, uint32_t n_rate_ppm, uint8_t *d_out,
                                   void *stream) {
  return mgc_dev_synth_reads_ex(seed, genome_len, first_read, n_reads, read_len, sub_rate_ppm, n_rate_ppm, 0, 1, 1, d_out, stream);
}

extern "C" int mgc_dev_synth_reads_ex(uint64_t seed, uint64_t genome_len, uint64_t first_read, uint64_t n_reads,
                                      uint32_t read_len, uint32_t sub_rate_ppm, uint32_t n_rate_ppm,
                                      uint32_t repeat_ppm, uint32_t repeat_unit, uint32_t repeat_families,
                                      uint8_t *d_out, void *stream) {
  if (!d_out || read_len == 0 || genome_len < read_len) return MGC_EINVAL;
  return hip_rc(mgc::launch_synth_reads(seed, genome_len, first_read, n_reads, read_len, sub_rate_ppm, n_rate_ppm,
                                        repeat_ppm, repeat_unit, repeat_families, d_out, (hipStream_t)stream), "synth_reads");
}

// ---------------------------------------------------------------------------
// Session
// ---------------------------------------------------------------------------
extern "C" const char *mgc_last_error(const mgc_session *s) {
  return s ? s->err.c_str() : mgc::thread_last_error().c_str();
}

// The block geometry a count really uses: the configuration's, or -- in simple mode -- countSimple's.
bool mgc::effective_geometry(mgc_count_config *c) {
  const uint32_t sfx_len = c->count_suffix_length;
  if (c->use_simple) {
    // The reference switches to countSimple (merylOp-count.C:368-372): a direct-index counter
    // whose RESULT is the same sorted (k-mer, count) stream but whose database geometry is
    //   psbits = 2k - 6, wSuffix = min(20, psbits), wPrefix = 6 + psbits - wSuffix
    // (merylOp-countSimple.C:172-175).  The sort-based engine produces that stream for any
    // k, so simple mode only changes the block geometry.
    if (2 * c->k < MGC_NUM_FILES_BITS + 2 * sfx_len) { set_err(nullptr, "mgc_open: k=%u too small for a 64-file database", c->k); return false; }
    // A k-mer is [file][blockPrefix][suffix][count-suffix] (:140): the count-suffix bases take no part in the split, and
    // travel at the end of every block suffix (:231-233)
    const uint32_t psbits = 2 * c->k - 2 * sfx_len - MGC_NUM_FILES_BITS;
    const uint32_t w_suffix = (psbits > 20) ? 20 : psbits;
    c->w_prefix = MGC_NUM_FILES_BITS + psbits - w_suffix;
    c->n_prefix = (uint64_t)1 << c->w_prefix;
    c->w_data   = w_suffix + 2 * sfx_len;
  }
  return true;
}

extern "C" mgc_session *mgc_open(const mgc_count_config *cfg, int device) {
  if (!cfg) { set_err(nullptr, "mgc_open: NULL config"); return nullptr; }
  mgc_count_config eff = *cfg;
  if (cfg->label_size > 64) { set_err(nullptr, "mgc_open: label_size %u (at most 64 bits)", cfg->label_size); return nullptr; }
  uint64_t sfx_mask = 0, sfx_test = 0;
  const uint32_t sfx_len = cfg->count_suffix_length;
  if (sfx_len) {
    // count-suffix=<bases>: merylOp.H:139-147 packs the string like a k-mer (2-bit codes, last base lowest);
    // merylOp-countSimple.C:50-58 builds the mask, :88-93 tests the k-mer that is counted against it
    if (sfx_len > MGC_MAX_COUNT_SUFFIX || strnlen(cfg->count_suffix, sizeof(cfg->count_suffix)) != sfx_len) {
      set_err(nullptr, "mgc_open: count_suffix must hold count_suffix_length (1..%d) bases", MGC_MAX_COUNT_SUFFIX);
      return nullptr;
    }
    if (cfg->k < sfx_len + 3) { set_err(nullptr, "mgc_open: count-suffix of %u bases needs k >= %u", sfx_len, sfx_len + 3); return nullptr; }
    if (2 * cfg->k - 2 * sfx_len > 42) {            // findExpectedSimpleSize, merylOp-count.C:142-145: "Not possible."
      set_err(nullptr, "mgc_open: count-suffix forces simple mode, which takes at most 21 free bases (k - suffix length = %u)", cfg->k - sfx_len);
      return nullptr;
    }
    for (uint32_t i = 0; i < sfx_len; i++) {
      const char ch = cfg->count_suffix[i];
      const bool ok = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T' || ch == 'a' || ch == 'c' || ch == 'g' || ch == 't';
      if (!ok) { set_err(nullptr, "mgc_open: count-suffix '%s' holds something that is not ACGT", cfg->count_suffix); return nullptr; }
      sfx_test = (sfx_test << 2) | (uint64_t)(((unsigned char)ch >> 1) & 3);        // A0 C1 T2 G3
    }
    sfx_mask = (sfx_len == 32) ? ~0ull : (((uint64_t)1 << (2 * sfx_len)) - 1);
    eff.use_simple = 1;                                                               // merylOp-count.C:379-382
  }
  if (!mgc::effective_geometry(&eff)) return nullptr;
  cfg = &eff;
  if (cfg->k == 0 || cfg->k > 64 || cfg->w_prefix < MGC_NUM_FILES_BITS || cfg->w_prefix > 2 * cfg->k ||
      cfg->w_data != 2 * cfg->k - cfg->w_prefix) {
    set_err(nullptr, "mgc_open: config has not been through mgc_configure_counting (k=%u wPrefix=%u wData=%u)",
            cfg->k, cfg->w_prefix, cfg->w_data);
    return nullptr;
  }

  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    set_err(nullptr, "mgc_open: no HIP device (%s)", hipGetErrorString(e));
    return nullptr;
  }
  mgc_session *s = new mgc_session();
  s->cfg = *cfg;
  s->sfx_mask = sfx_mask;
  s->sfx_test = sfx_test;
  s->key_words = (cfg->k > 32) ? 2u : 1u;
  if (device >= 0) {
    e = hipSetDevice(device);
    if (e != hipSuccess) { set_err(nullptr, "hipSetDevice(%d): %s", device, hipGetErrorString(e)); delete s; return nullptr; }
    s->device = device;
  } else {
    (void)hipGetDevice(&s->device);
  }
  e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->st_in, hipStreamNonBlocking);
  if (e != hipSuccess) { set_err(nullptr, "hipStreamCreate: %s", hipGetErrorString(e)); delete s; return nullptr; }
  if (hipStreamCreateWithFlags(&s->stream2, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming) != hipSuccess) {
    set_err(nullptr, "hipStreamCreate / hipEventCreate failed"); mgc_close(s); return nullptr;
  }
  memset(&s->prof, 0, sizeof(s->prof));
  memset(s->file_instances, 0, sizeof(s->file_instances));
  memset(s->total_file_instances, 0, sizeof(s->total_file_instances));
  s->sw = mgc::read_switches();                              // the environment is read here, once per session
  return s;
}

static void join_prepare(mgc_session *s) {
  if (s->prep_active) { s->prep_thread.join(); s->prep_active = false; }
}

extern "C" void mgc_close(mgc_session *s) {
  if (!s) return;
  if (s->worker_active) { s->worker.join(); s->worker_active = false; }
  join_prepare(s);
  (void)hipSetDevice(s->device);
  if (s->st_in) (void)hipStreamSynchronize(s->st_in);
  s->free_result();
  s->free_garbage();
  s->free_arena();
  delete s->runs;
  s->runs = nullptr;
  for (int i = 0; i < 2; i++) {
    if (s->text_pinned[i]) (void)hipHostFree(s->text_pinned[i]);
    if (s->text_ev[i]) (void)hipEventDestroy(s->text_ev[i]);
    if (s->up_ev[i]) (void)hipEventDestroy(s->up_ev[i]);
    if (s->pin[i]) (void)hipHostFree(s->pin[i]);
    if (s->pin_ev[i]) (void)hipEventDestroy(s->pin_ev[i]);
  }
  for (char *&p : s->text_ring) if (p) { (void)hipHostFree(p); p = nullptr; }
  if (s->st_up) { (void)hipStreamSynchronize(s->st_up); (void)hipStreamDestroy(s->st_up); }
  if (s->st_in) (void)hipStreamDestroy(s->st_in);
  if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
  if (s->h_stats) (void)hipHostFree(s->h_stats);
  if (s->ev_join) (void)hipEventDestroy(s->ev_join);
  if (s->stream2) (void)hipStreamDestroy(s->stream2);
  for (int i = 0; i < mgc_session::HUGE_EXTRA; i++) if (s->stream_h[i]) (void)hipStreamDestroy(s->stream_h[i]);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

// ---------------------------------------------------------------------------------------------------------------
//  Input staging (see mgc_session.hpp): one device-resident base stream per batch, double buffered
// ---------------------------------------------------------------------------------------------------------------
struct HostParseState { uint64_t out_len, file_start_len; uint32_t state, prev_nl, error, pad; };

static int count_staged_batch(mgc_session *s, int which, uint64_t n);      // count stage[which][0, n) and merge it into R

// The count's largest buffer (the k-mer instances, 8 / 16 B per base at most) is allocated NOW, by a helper thread, while the
// caller reads and uploads its input: a first large hipMalloc is the slowest single thing a freshly started process does
// (0.3 s for 100 GB on a quiet device, seconds when another process has just released that much), and until the count starts
// nobody needs the memory.  Only when the whole input is expected to fit one pass; a wrong estimate costs a re-allocation,
// never a wrong result.
extern "C" int mgc_prepare(mgc_session *s, uint64_t expected_bases) {
  if (!s) return MGC_EINVAL;
  if (s->borrowed || s->counted || s->prep_active || expected_bases == 0) return MGC_OK;
  HIP_TRY(s, hipSetDevice(s->device));
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return MGC_OK;
  const uint64_t per_base = 2 + 14ull * s->key_words;
  if (s->batch_limit && expected_bases > s->batch_limit) return MGC_OK;
  if ((double)expected_bases * (double)per_base > 0.8 * (double)free_b) return MGC_OK;       // batches: their arena is sized by the first one
  const size_t bytes = sizeof(uint64_t) * s->key_words * expected_bases;
  s->prep_active = true;
  s->prep_thread = std::thread([s, bytes] {
    (void)hipSetDevice(s->device);
    if (s->ensure(mgc_session::B_X, bytes) != hipSuccess) (void)hipGetLastError();           // the count will say so itself
    // ... and the code objects of the count and of the database encoder: loaded lazily at the first launch, i.e. inside a fresh
    // process's first count otherwise (0.36 s of count for a 0.12 s step, profiles/r04o)
    (void)mgc::warm_kmer(); (void)mgc::warm_scan(); (void)mgc::warm_sort(); (void)mgc::warm_finish(); (void)mgc::warm_encode();
    (void)hipGetLastError();
  });
  return MGC_OK;
}

extern "C" int mgc_set_batch_bases(mgc_session *s, uint64_t bases_per_batch) {
  if (!s) return MGC_EINVAL;
  s->batch_limit = bases_per_batch;
  return MGC_OK;
}

static inline int stage_id(int which) { return which ? mgc_session::B_STAGE1 : mgc_session::B_STAGE0; }
static inline uint8_t *stage_ptr(mgc_session *s, int which) { return reinterpret_cast<uint8_t *>(s->buf[stage_id(which)].p); }

// first input of a session: streams, parse state, the batch size
static int input_setup(mgc_session *s) {
  if (s->borrowed || s->counted) { set_err(&s->err, "input after device input / count"); return MGC_ESTATE; }
  HIP_TRY(s, hipSetDevice(s->device));
  if (s->state_ready) return MGC_OK;
  if (s->batch_limit == 0) {
    // One pass holds, per base: the staged base, 0.87 k-mer instances of 8 (16) B in X, 4 B of count scratch each, the
    // distinct k-mers with their counts (about a seventh of the instances at 30x), the ping-pong buffer of the largest
    // file -- measured 13 B per base at k=21 (130 GB for 10 Gbp: DESIGN.md section 2), 16 (30) B with slack.  A fifth of
    // the HBM stays free for the running result of earlier batches and its merge target.
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b) {
      const uint64_t per_base = 2 + 14ull * s->key_words;
      s->batch_limit = (uint64_t)((double)free_b * 0.80 / (double)per_base);
    }
    if (s->batch_limit < (1u << 20)) s->batch_limit = 1u << 20;
  }
  HIP_TRY(s, s->ensure(mgc_session::B_TEXT_STATE, mgc::text_parse_state_bytes() + 64));
  HIP_TRY(s, s->ensure_preserve(stage_id(s->fill), 1u << 20, 0, s->st_in));
  HIP_TRY(s, mgc::launch_text_file_op(s->buf[mgc_session::B_TEXT_STATE].p, stage_ptr(s, s->fill), 3, s->st_in));
  s->state_ready = true;
  return MGC_OK;
}

// the device knows the exact length of the staged stream; bring the host's copy up to date (synchronises st_in)
static int resolve_length(mgc_session *s, HostParseState *out = nullptr) {
  HostParseState h;
  HIP_TRY(s, hipMemcpyAsync(&h, s->buf[mgc_session::B_TEXT_STATE].p, sizeof(h), hipMemcpyDeviceToHost, s->st_in));
  HIP_TRY(s, hipStreamSynchronize(s->st_in));
  s->fill_len = h.out_len;
  s->len_inexact = false;
  if (out) *out = h;
  return MGC_OK;
}

static double io_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int join_worker(mgc_session *s) {
  if (s->worker_active) {
    const double t0 = io_now();
    s->worker.join();
    s->tr_join += io_now() - t0;
    s->worker_active = false;
    if (s->worker_rc != MGC_OK) return s->worker_rc;
  }
  return s->worker_rc;
}

// host-pushed bases collected in the current pinned chunk -> stage[fill] (asynchronous; the other chunk takes over)
static int flush_pinned(mgc_session *s) {
  if (s->pin_len == 0) return MGC_OK;
  if (s->len_inexact) { const int rc = resolve_length(s); if (rc != MGC_OK) return rc; }
  const int c = s->pin_cur;
  HIP_TRY(s, s->ensure_preserve(stage_id(s->fill), s->fill_len + s->pin_len + 4096, s->fill_len, s->st_in));
  HIP_TRY(s, hipMemcpyAsync(stage_ptr(s, s->fill) + s->fill_len, s->pin[c], s->pin_len, hipMemcpyHostToDevice, s->st_in));
  s->fill_len += s->pin_len;
  HIP_TRY(s, mgc::launch_text_set_len(s->buf[mgc_session::B_TEXT_STATE].p, s->fill_len, 0, s->st_in));
  HIP_TRY(s, hipEventRecord(s->pin_ev[c], s->st_in));
  s->pin_used[c] = true;
  s->pin_len = 0;
  s->pin_cur = c ^ 1;
  if (s->pin_used[s->pin_cur]) HIP_TRY(s, hipEventSynchronize(s->pin_ev[s->pin_cur]));     // that chunk's upload has read it
  return MGC_OK;
}

// The staged stream has reached the batch size: everything up to the last sequence boundary becomes a batch, counted by
// the worker thread while the caller goes on filling the other staging buffer (k-mers never span a breaker, so no
// carry is needed, and `compress` stays a per-sequence operation).
static int cut_batch(mgc_session *s) {
  int rc = flush_pinned(s);
  if (rc != MGC_OK) return rc;
  HostParseState h;
  rc = resolve_length(s, &h);
  if (rc != MGC_OK) return rc;
  if (s->text_open && h.error) return MGC_OK;              // the open file is about to be refused and rolled back: not now
  if (s->fill_len < s->batch_limit) return MGC_OK;         // the bound was pessimistic
  if (s->fill_len < s->no_cut_below) return MGC_OK;        // no sequence boundary was found last time: look again once the stream has doubled
  uint64_t *d_last = reinterpret_cast<uint64_t *>(reinterpret_cast<unsigned char *>(s->buf[mgc_session::B_TEXT_STATE].p) +
                                                  ((mgc::text_parse_state_bytes() + 7) / 8) * 8);
  uint64_t cut = 0;
  HIP_TRY(s, mgc::launch_last_breaker(stage_ptr(s, s->fill), s->fill_len, d_last, s->st_in));
  HIP_TRY(s, hipMemcpyAsync(&cut, d_last, sizeof(cut), hipMemcpyDeviceToHost, s->st_in));
  HIP_TRY(s, hipStreamSynchronize(s->st_in));
  uint64_t overlap = 0;
  if (cut == 0) {
    // One sequence longer than a batch (this used to return and re-scan the whole staged stream on every later
    // push -- O(n^2) -- while staging grew without bound).  The reference spills inside a sequence too; here the batch is cut
    // at the end of what is staged and the last k-1 bases are staged AGAIN in front of the next batch: a window that starts
    // in them is incomplete in this batch and complete in the next, so no k-mer is lost or counted twice (the rule the
    // node count cuts its slices by).  `compress` needs whole sequences: such a stream waits until it has doubled.
    if (s->cfg.homopoly_compress || s->text_open || s->fill_len < 4ull * s->cfg.k) {
      s->no_cut_below = s->fill_len * 2;
      return MGC_OK;
    }
    cut = s->fill_len;
    overlap = s->cfg.k - 1;
  }
  rc = join_worker(s);                                      // the previous batch is done: its staging buffer is free, R is stable
  if (rc != MGC_OK) return rc;
  const uint64_t tail = s->fill_len - cut + overlap;
  cut -= overlap;                                           // the tail starts k-1 bases before the cut ...
  const int other = s->fill ^ 1;
  HIP_TRY(s, s->ensure_preserve(stage_id(other), tail + (1u << 20), 0, s->st_in));
  if (tail) HIP_TRY(s, hipMemcpyAsync(stage_ptr(s, other), stage_ptr(s, s->fill) + cut, tail, hipMemcpyDeviceToDevice, s->st_in));
  HIP_TRY(s, mgc::launch_text_set_len(s->buf[mgc_session::B_TEXT_STATE].p, tail, s->text_open ? 1 : 0, s->st_in));
  HIP_TRY(s, hipStreamSynchronize(s->st_in));
  if (s->text_open) s->text_cut_in_file = true;
  const int which = s->fill;
  s->fill = other;
  s->fill_len = tail;
  s->no_cut_below = 0;
  cut += overlap;                                           // ... and the batch still ends at the cut
  s->worker_rc = MGC_OK;
  s->worker_active = true;
  s->worker = std::thread([s, which, cut] { s->worker_rc = count_staged_batch(s, which, cut); });
  return MGC_OK;
}

extern "C" int mgc_push_bases(mgc_session *s, const char *bases, size_t len, int end_of_sequence) {
  if (!s || (!bases && len)) return MGC_EINVAL;
  int rc = input_setup(s);
  if (rc != MGC_OK) return rc;
  if (s->text_open) { set_err(&s->err, "mgc_push_bases while a text file is open (mgc_end_text first)"); return MGC_ESTATE; }
  s->input_seen = true;
  for (int i = 0; i < 2; i++) {
    if (!s->pin[i]) HIP_TRY(s, hipHostMalloc(reinterpret_cast<void **>(&s->pin[i]), mgc_session::PIN_CHUNK, hipHostMallocDefault));
    if (!s->pin_ev[i]) HIP_TRY(s, hipEventCreateWithFlags(&s->pin_ev[i], hipEventDisableTiming));
  }
  const char breaker = '.';                                  // merylOp-countThreads.C:214-215
  for (int part = 0; part < 2; part++) {
    const char *src = part ? &breaker : bases;
    size_t left = part ? (end_of_sequence ? 1 : 0) : len;
    while (left) {
      const size_t take = std::min(left, mgc_session::PIN_CHUNK - s->pin_len);
      const double t0 = io_now();
      memcpy(s->pin[s->pin_cur] + s->pin_len, src, take);
      const double t1 = io_now();
      s->tr_memcpy += t1 - t0;
      s->pin_len += take; src += take; left -= take;
      if (s->pin_len == mgc_session::PIN_CHUNK) { rc = flush_pinned(s); if (rc != MGC_OK) return rc; }
      const double t2 = io_now();
      s->tr_flush += t2 - t1;
      if (s->fill_len + s->pin_len >= s->batch_limit) { rc = cut_batch(s); if (rc != MGC_OK) return rc; s->tr_cut += io_now() - t2; }
    }
  }
  return MGC_OK;
}

// ---- text input, parsed on the device (include/meryl_gpu_count.h) ------------------------------
static int text_setup(mgc_session *s) {
  int rc = input_setup(s);
  if (rc != MGC_OK) return rc;
  if (!s->text_pinned[0]) {
    HIP_TRY(s, s->ensure(mgc_session::B_TEXT_IN0, mgc_session::TEXT_CHUNK));
    HIP_TRY(s, s->ensure(mgc_session::B_TEXT_IN1, mgc_session::TEXT_CHUNK));
    HIP_TRY(s, s->ensure(mgc_session::B_TEXT_WS, mgc::text_parse_workspace_bytes(mgc_session::TEXT_CHUNK)));
    if (!s->st_up) HIP_TRY(s, hipStreamCreateWithFlags(&s->st_up, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) {
      if (!s->text_pinned[i]) HIP_TRY(s, hipHostMalloc(reinterpret_cast<void **>(&s->text_pinned[i]), mgc_session::TEXT_CHUNK, hipHostMallocDefault));
      if (!s->text_ev[i]) HIP_TRY(s, hipEventCreateWithFlags(&s->text_ev[i], hipEventDisableTiming));
      if (!s->up_ev[i]) HIP_TRY(s, hipEventCreateWithFlags(&s->up_ev[i], hipEventDisableTiming));
    }
  }
  return MGC_OK;
}

extern "C" int mgc_reserve_text(mgc_session *s, uint64_t text_bytes) {
  if (!s) return MGC_EINVAL;
  int rc = text_setup(s);
  if (rc != MGC_OK) return rc;
  // no more than a batch (plus what a chunk can add before the cut is noticed) is ever staged at once
  const uint64_t want = std::min<uint64_t>(text_bytes, s->batch_limit + 2 * mgc_session::TEXT_CHUNK) + 4096;
  HIP_TRY(s, s->ensure_preserve(stage_id(s->fill), want, s->fill_len, s->st_in));
  return MGC_OK;
}

extern "C" int mgc_begin_text(mgc_session *s, int format) {
  if (!s || (format != MGC_TEXT_FASTA && format != MGC_TEXT_FASTQ)) return MGC_EINVAL;
  if (s->text_open) { set_err(&s->err, "mgc_begin_text: the previous file was not ended"); return MGC_ESTATE; }
  int rc = text_setup(s);
  if (rc != MGC_OK) return rc;
  rc = flush_pinned(s);                                     // host-pushed bases come first in the stream
  if (rc != MGC_OK) return rc;
  HIP_TRY(s, mgc::launch_text_file_op(s->buf[mgc_session::B_TEXT_STATE].p, stage_ptr(s, s->fill), 0, s->st_in));
  s->text_format = format;
  s->text_open = true;
  s->text_cut_in_file = false;
  s->input_seen = true;
  return MGC_OK;
}

// one piece (<= TEXT_CHUNK bytes, in pinned memory) -> device input buffer b -> parse kernels; returns once they are queued
static int text_submit(mgc_session *s, const char *pinned_src, size_t piece) {
  const uint32_t b = s->text_next & 1u;
  if (s->text_ev_used[b]) HIP_TRY(s, hipEventSynchronize(s->text_ev[b]));       // device buffer b (and its previous source) are free again
  HIP_TRY(s, s->ensure_preserve(stage_id(s->fill), s->fill_len + piece + 4096, s->fill_len, s->st_in));
  uint8_t *d_in = reinterpret_cast<uint8_t *>(s->buf[b ? mgc_session::B_TEXT_IN1 : mgc_session::B_TEXT_IN0].p);
  // the copy runs on its own stream, so that chunk c travels while chunk c-1 is parsed (one stream did them in turn: 0.56 ms
  // of copy + 0.4 ms of parse per 32 MiB, 0.6 s of the 20 GB file -> database run; profiles/r03m_e2e_io.txt)
  HIP_TRY(s, hipMemcpyAsync(d_in, pinned_src, piece, hipMemcpyHostToDevice, s->st_up));
  HIP_TRY(s, hipEventRecord(s->up_ev[b], s->st_up));
  HIP_TRY(s, hipStreamWaitEvent(s->st_in, s->up_ev[b], 0));
  HIP_TRY(s, mgc::launch_text_parse(d_in, piece, s->text_format == MGC_TEXT_FASTQ, s->buf[mgc_session::B_TEXT_STATE].p,
                                    s->buf[mgc_session::B_TEXT_WS].p, stage_ptr(s, s->fill), s->st_in));
  HIP_TRY(s, hipEventRecord(s->text_ev[b], s->st_in));
  s->text_ev_used[b] = true;
  s->text_next++;
  s->fill_len += piece;                                     // an upper bound: headers, qualities and line ends are dropped
  s->len_inexact = true;
  if (s->fill_len >= s->batch_limit) return cut_batch(s);
  return MGC_OK;
}

extern "C" int mgc_push_text(mgc_session *s, const char *text, size_t len) {
  if (!s || (!text && len)) return MGC_EINVAL;
  if (!s->text_open) { set_err(&s->err, "mgc_push_text without mgc_begin_text"); return MGC_ESTATE; }
  HIP_TRY(s, hipSetDevice(s->device));
  while (len) {
    const size_t piece = len < mgc_session::TEXT_CHUNK ? len : mgc_session::TEXT_CHUNK;
    const uint32_t b = s->text_next & 1u;
    if (s->text_ev_used[b]) HIP_TRY(s, hipEventSynchronize(s->text_ev[b]));     // pinned + device buffer b are free again
    memcpy(s->text_pinned[b], text, piece);
    const int rc = text_submit(s, s->text_pinned[b], piece);
    if (rc != MGC_OK) return rc;
    text += piece;
    len -= piece;
  }
  return MGC_OK;
}

// A whole uncompressed FASTA/FASTQ file: `reader_threads` threads pread() 32 MiB chunks straight into a ring of pinned
// buffers (no intermediate copy), the calling thread uploads and parses them in file order.  A 20 GB FASTQ on tmpfs is
// otherwise bound by ONE thread's read()+memcpy (measured 11 GB/s, 1.8 s of a 3 s file -> database run).
// format of a FASTA / FASTQ file from its first byte that is not white space; 0: neither
static int sniff_text_format(int fd, char *first) {
  char head[4096];
  const ssize_t got = pread(fd, head, sizeof(head), 0);
  char c = '>';
  for (ssize_t i = 0; i < got; i++) if (head[i] != '\n' && head[i] != '\r' && head[i] != ' ' && head[i] != '\t') { c = head[i]; break; }
  if (first) *first = c;
  return c == '@' ? MGC_TEXT_FASTQ : (c == '>' ? MGC_TEXT_FASTA : 0);
}

// First record start at or after `offset` of a FASTA / FASTQ file -- where a reader that takes the file from the middle
// may begin (the ranks of a node count read disjoint byte windows of the input, each through its own device's link).
// FASTA: a line that starts with '>'.  FASTQ (four-line records): a line that starts with '@' whose next-but-one line starts
// with '+' -- a quality line may start with '@', but then the line two below it is a sequence line, never '+'.
extern "C" int mgc_text_record_start(const char *path, int format, uint64_t offset, uint64_t *start) {
  if (!path || !start) return MGC_EINVAL;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) { set_err(nullptr, "mgc_text_record_start: cannot open '%s': %s", path, strerror(errno)); return MGC_EINVAL; }
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); set_err(nullptr, "mgc_text_record_start: '%s' is not a regular file", path); return MGC_EINVAL; }
  const uint64_t size = (uint64_t)st.st_size;
  if (format == 0) format = sniff_text_format(fd, nullptr);
  if (format != MGC_TEXT_FASTA && format != MGC_TEXT_FASTQ) { close(fd); set_err(nullptr, "'%s' is neither FASTA nor FASTQ", path); return MGC_EFORMAT; }
  if (offset == 0 || offset >= size) { close(fd); *start = offset >= size ? size : 0; return MGC_OK; }
  // line starts from offset - 1 on: the byte before a line start is '\n'
  std::vector<char> buf(1u << 20);
  uint64_t pos = offset - 1;                               // file position of buf[0]
  std::vector<uint64_t> ls;                                // line starts found so far (file offsets), with their first byte
  std::vector<char> lc;
  uint64_t answer = size;
  bool found = false;
  while (!found && pos < size) {
    const ssize_t got = pread(fd, buf.data(), buf.size(), (off_t)pos);
    if (got <= 0) break;
    for (ssize_t i = 0; i < got && !found; i++) {
      if (buf[i] != '\n') continue;
      const uint64_t line = pos + (uint64_t)i + 1;
      if (line >= size) break;
      char c;
      if (i + 1 < got) c = buf[i + 1];
      else if (pread(fd, &c, 1, (off_t)line) != 1) break;
      if (format == MGC_TEXT_FASTA) { if (c == '>') { answer = line; found = true; } continue; }
      ls.push_back(line); lc.push_back(c);
      const size_t m = ls.size();
      if (m >= 3 && lc[m - 3] == '@' && lc[m - 1] == '+') { answer = ls[m - 3]; found = true; }
    }
    pos += (uint64_t)got;
  }
  close(fd);
  *start = found ? answer : size;
  return MGC_OK;
}

static int push_text_file_range(mgc_session *s, const char *path, int format, int reader_threads, uint64_t range_begin, uint64_t range_end);

extern "C" int mgc_push_text_file(mgc_session *s, const char *path, int format, int reader_threads) {
  return push_text_file_range(s, path, format, reader_threads, 0, ~0ull);
}

extern "C" int mgc_push_text_file_range(mgc_session *s, const char *path, int format, int reader_threads, uint64_t begin, uint64_t end) {
  return push_text_file_range(s, path, format, reader_threads, begin, end);
}

// bytes [range_begin, range_end) of the file (range_begin at a record start; range_end = the next reader's start, or past the end)
static int push_text_file_range(mgc_session *s, const char *path, int format, int reader_threads, uint64_t range_begin, uint64_t range_end) {
  if (!s || !path) return MGC_EINVAL;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) { set_err(&s->err, "mgc_push_text_file: cannot open '%s': %s", path, strerror(errno)); return MGC_EINVAL; }
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); set_err(&s->err, "mgc_push_text_file: '%s' is not a regular file", path); return MGC_EINVAL; }
  if (format == 0) {                                        // sniff: first byte that is not white space
    char c = 0;
    format = sniff_text_format(fd, &c);
    if (!format) { close(fd); set_err(&s->err, "'%s' is neither FASTA nor FASTQ (record starts with '%c')", path, c); return MGC_EFORMAT; }
  }
  const uint64_t file_size = (uint64_t)st.st_size;
  if (range_end > file_size) range_end = file_size;
  if (range_begin > range_end) range_begin = range_end;
  const uint64_t base = range_begin;                        // every file offset below is relative to the window
  const uint64_t size = range_end - range_begin;
  int rc = mgc_begin_text(s, format);
  if (rc != MGC_OK) { close(fd); return rc; }

  // Readers and ring: measured on the 2 x 64-core box (scripts/e2e_cli.py, 20.5 GB FASTQ on tmpfs): 6 readers / 8 slots keep
  // the uploader waiting 1.7 s, 16 readers / 24 slots 0.01 s (the loop then runs at the 0.5 s of upload + parse).  The pinned
  // slots are allocated by the readers themselves, in parallel, on first use, and stay with the session for the next file.
  constexpr int RMAX = mgc_session::TEXT_RING_MAX;
  if (reader_threads <= 0) reader_threads = 16;
  reader_threads = std::max(1, std::min(reader_threads, RMAX - 8));
  int R = reader_threads + 8;                               // up to R-2 chunks of read-ahead
  const size_t CH = mgc_session::TEXT_CHUNK;
  const uint64_t nchunks = (size + CH - 1) / CH;
  reader_threads = (int)std::min<uint64_t>((uint64_t)std::max(1, std::min(reader_threads, R - 2)), nchunks ? nchunks : 1);
  char **ring = s->text_ring;
  std::atomic<bool> alloc_failed(false);
  std::mutex mu;
  std::condition_variable cv;
  uint64_t free_gen[RMAX], ready_chunk[RMAX];               // slot i may be filled with chunk c iff free_gen[i] == c / R
  size_t   ready_len[RMAX];
  std::vector<double> t_pread(64, 0.0), t_slotwait(64, 0.0);
  std::atomic<int> reader_ids(0);
  for (int i = 0; i < R; i++) { free_gen[i] = 0; ready_chunk[i] = ~0ull; ready_len[i] = 0; }
  std::atomic<uint64_t> next_chunk(0);
  bool abort_all = false, read_failed = false;
  auto reader = [&]() {
    const int me = reader_ids.fetch_add(1) & 63;
    auto rnow = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (;;) {
      const uint64_t c = next_chunk.fetch_add(1);
      if (c >= nchunks) return;
      const int slot = (int)(c % R);
      const double w0 = rnow();
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return abort_all || free_gen[slot] == c / R; });
        if (abort_all) return;
      }
      if (!ring[slot]) {                                    // first use of this slot (exactly one reader gets here per slot)
        (void)hipSetDevice(s->device);
        if (hipHostMalloc(reinterpret_cast<void **>(&ring[slot]), CH, hipHostMallocDefault) != hipSuccess) {
          ring[slot] = nullptr;
          alloc_failed.store(true);
          std::lock_guard<std::mutex> g(mu);
          read_failed = true; abort_all = true;
          cv.notify_all();
          return;
        }
      }
      const double w1 = rnow();
      t_slotwait[me] += w1 - w0;
      const uint64_t off = c * CH;
      const size_t want = (size_t)std::min<uint64_t>(CH, size - off);
      size_t have = 0;
      bool ok = true;
      while (have < want) {
        const ssize_t r = pread(fd, ring[slot] + have, want - have, (off_t)(base + off + have));
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) { ok = false; break; }                  // an error, or the file shrank under us
        have += (size_t)r;
      }
      t_pread[me] += rnow() - w1;
      std::lock_guard<std::mutex> g(mu);
      if (!ok) { read_failed = true; abort_all = true; }
      ready_chunk[slot] = c; ready_len[slot] = have;
      cv.notify_all();
    }
  };
  std::vector<std::thread> readers;
  for (int t = 0; t < reader_threads; t++) readers.emplace_back(reader);
  const bool trace = getenv("MGC_IO_TRACE") != nullptr;
  double t_wait = 0, t_submit = 0;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (uint64_t c = 0; c < nchunks && rc == MGC_OK; c++) {
    const int slot = (int)(c % R);
    size_t len = 0;
    const double t0 = now();
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return abort_all || ready_chunk[slot] == c; });
      if (abort_all) break;
      len = ready_len[slot];
    }
    const double t1 = now();
    t_wait += t1 - t0;
    // text_submit first waits for the parse of chunk c-2 (same device buffer): after that the pinned slot of chunk c-2
    // has been read by its upload and goes back to the readers
    rc = text_submit(s, ring[slot], len);
    t_submit += now() - t1;
    if (c >= 2) {
      std::lock_guard<std::mutex> g(mu);
      free_gen[(c - 2) % R]++;
      cv.notify_all();
    }
  }
  { std::lock_guard<std::mutex> g(mu); if (rc != MGC_OK) abort_all = true; cv.notify_all(); }
  // the last uploads still read from the ring
  for (int b = 0; b < 2; b++) if (s->text_ev_used[b]) (void)hipEventSynchronize(s->text_ev[b]);
  { std::lock_guard<std::mutex> g(mu); abort_all = abort_all || true; cv.notify_all(); }
  for (auto &t : readers) t.join();
  close(fd);
  if (alloc_failed.load()) { set_err(&s->err, "mgc_push_text_file: pinned buffers: out of memory"); (void)mgc_end_text(s); return MGC_ENOMEM; }
  if (trace) {
    double sp = 0, sw = 0;
    for (int i = 0; i < 64; i++) { sp += t_pread[i]; sw += t_slotwait[i]; }
    fprintf(stderr, "[io] text file %.2f GB in %llu chunks, %d readers, ring %d: waiting for readers %.3f s, upload+parse submit (incl. waits "
                    "for the device) %.3f s; readers: %.3f s in pread (%.1f GB/s each), %.3f s waiting for a free slot\n",
            size / 1e9, (unsigned long long)nchunks, reader_threads, R, t_wait, t_submit, sp, sp > 0 ? size / 1e9 / sp : 0.0, sw);
  }
  if (read_failed) { set_err(&s->err, "mgc_push_text_file: reading '%s' failed: %s", path, strerror(errno)); rc = MGC_EINVAL; }
  const int rc_end = mgc_end_text(s);                       // closes the file in every case (rolls it back on MGC_EFORMAT)
  return rc != MGC_OK ? rc : rc_end;
}

// ---- a BGZF file (bgzip'd FASTA / FASTQ: independent gzip members of <= 64 KiB of text, their compressed size in a 'BC' extra field,
// SAMv1 4.1) inflated by `threads` threads STRAIGHT into the pinned upload ring (round 6) ----
// Through the generic reader (meryl_seq.cpp: BgzfSource -> msr_read_text -> mgc_push_text) the text of a batch of blocks was copied twice
// by the calling thread (out of the inflater's batch, into the pinned buffer) behind batches of 32 MiB whose threads were spawned per
// batch: 4 GB/s of text with 32 threads (profiles/r06w: bench.py e2e_compressed).  Here the file is mapped, its blocks are indexed in one
// walk over the headers, chunks of <= TEXT_CHUNK of text are handed to persistent worker threads that inflate block after block into the
// chunk's ring slot, and the calling thread uploads and parses the chunks in order (push_text_file_range's ring).
namespace {
struct BgzfBlock { uint64_t off; uint32_t csize, hdr, isize; };
// size of the BGZF block at p (n >= 18 bytes there), its header length; 0: not a BGZF block
static size_t bgzf_block_at(const unsigned char *p, size_t n, uint32_t *hdr) {
  if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
  const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8);
  if (12 + xlen > n) return 0;
  size_t o = 12;
  while (o + 4 <= 12 + xlen) {
    const size_t slen = (size_t)p[o + 2] | ((size_t)p[o + 3] << 8);
    if (p[o] == 'B' && p[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + xlen) { *hdr = (uint32_t)(12 + xlen); return ((size_t)p[o + 4] | ((size_t)p[o + 5] << 8)) + 1; }
    o += 4 + slen;
  }
  return 0;
}
}  // namespace

extern "C" int mgc_is_bgzf_file(const char *path) {
  if (!path) return 0;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return 0;
  unsigned char head[64];
  const ssize_t got = pread(fd, head, sizeof(head), 0);
  close(fd);
  uint32_t hdr = 0;
  return (got >= 18 && bgzf_block_at(head, (size_t)got, &hdr) != 0) ? 1 : 0;
}

extern "C" int mgc_push_text_bgzf_file(mgc_session *s, const char *path, int format, int threads) {
  if (!s || !path) return MGC_EINVAL;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) { set_err(&s->err, "mgc_push_text_bgzf_file: cannot open '%s': %s", path, strerror(errno)); return MGC_EINVAL; }
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 28) { close(fd); set_err(&s->err, "mgc_push_text_bgzf_file: '%s' is not a regular BGZF file", path); return MGC_EINVAL; }
  const size_t fsize = (size_t)st.st_size;
  const unsigned char *map = reinterpret_cast<const unsigned char *>(mmap(nullptr, fsize, PROT_READ, MAP_SHARED, fd, 0));
  if (map == MAP_FAILED) { close(fd); set_err(&s->err, "mgc_push_text_bgzf_file: mmap of '%s' failed: %s", path, strerror(errno)); return MGC_EINVAL; }
  auto unmap = [&]() { munmap(const_cast<unsigned char *>(map), fsize); close(fd); };
  // ---- index: one walk over the block headers; chunks of whole blocks, <= TEXT_CHUNK of text each ----
  const size_t CH = mgc_session::TEXT_CHUNK;
  std::vector<BgzfBlock> blocks;
  struct Chunk { size_t first, last; size_t text; };
  std::vector<Chunk> chunks;
  {
    size_t off = 0, text = 0, first = 0;
    while (off < fsize) {
      uint32_t hdr = 0;
      const size_t bs = bgzf_block_at(map + off, fsize - off, &hdr);
      if (bs == 0 || off + bs > fsize || bs < (size_t)hdr + 8) { unmap(); set_err(&s->err, "'%s': not a BGZF block at offset %zu (plain gzip data, or a truncated file)", path, off); return MGC_EFORMAT; }
      const unsigned char *e = map + off + bs;
      const uint32_t isize = (uint32_t)e[-4] | ((uint32_t)e[-3] << 8) | ((uint32_t)e[-2] << 16) | ((uint32_t)e[-1] << 24);
      if (isize > 65536) { unmap(); set_err(&s->err, "'%s': corrupt BGZF block at offset %zu (ISIZE %u)", path, off, isize); return MGC_EFORMAT; }
      if (text + isize > CH) { chunks.push_back({first, blocks.size(), text}); first = blocks.size(); text = 0; }
      blocks.push_back({(uint64_t)off, (uint32_t)bs, hdr, isize});
      text += isize;
      off += bs;
    }
    if (blocks.size() > first) chunks.push_back({first, blocks.size(), text});
  }
  auto inflate_block = [&](z_stream &z, const BgzfBlock &b, unsigned char *dst) -> bool {
    if (b.isize == 0) return true;                             // the end-of-file marker (and any other empty block)
    const unsigned char *p = map + b.off;
    if (inflateReset(&z) != Z_OK) return false;
    z.next_in = const_cast<unsigned char *>(p + b.hdr);
    z.avail_in = b.csize - b.hdr - 8;
    z.next_out = dst;
    z.avail_out = b.isize;
    const int zr = inflate(&z, Z_FINISH);
    const uint32_t want = (uint32_t)p[b.csize - 8] | ((uint32_t)p[b.csize - 7] << 8) | ((uint32_t)p[b.csize - 6] << 16) | ((uint32_t)p[b.csize - 5] << 24);
    return zr == Z_STREAM_END && z.avail_out == 0 && (uint32_t)crc32(0L, dst, b.isize) == want;
  };
  if (format == 0) {                                            // sniff: the first byte of text that is not white space
    char c = 0;
    z_stream z; memset(&z, 0, sizeof(z));
    std::vector<unsigned char> tmp(65536);
    if (inflateInit2(&z, -15) != Z_OK) { unmap(); set_err(&s->err, "zlib: inflateInit2 failed"); return MGC_ENOMEM; }
    for (size_t i = 0; i < blocks.size() && !c; i++) {
      if (!inflate_block(z, blocks[i], tmp.data())) { inflateEnd(&z); unmap(); set_err(&s->err, "'%s': BGZF block %zu failed to inflate (corrupt file)", path, i); return MGC_EFORMAT; }
      for (uint32_t j = 0; j < blocks[i].isize; j++) if (tmp[j] != '\n' && tmp[j] != '\r' && tmp[j] != ' ' && tmp[j] != '\t') { c = (char)tmp[j]; break; }
    }
    inflateEnd(&z);
    format = c == '@' ? MGC_TEXT_FASTQ : (c == '>' ? MGC_TEXT_FASTA : 0);
    if (!format) { unmap(); set_err(&s->err, "'%s' is neither FASTA nor FASTQ (record starts with '%c')", path, c ? c : '?'); return MGC_EFORMAT; }
  }
  int rc = mgc_begin_text(s, format);
  if (rc != MGC_OK) { unmap(); return rc; }

  constexpr int RMAX = mgc_session::TEXT_RING_MAX;
  if (threads <= 0) threads = 16;
  threads = std::max(1, std::min(threads, RMAX - 8));
  const int R = threads + 8;
  const uint64_t nchunks = chunks.size();
  threads = (int)std::min<uint64_t>((uint64_t)std::max(1, std::min(threads, R - 2)), nchunks ? nchunks : 1);
  char **ring = s->text_ring;
  std::mutex mu;
  std::condition_variable cv;
  uint64_t free_gen[RMAX], ready_chunk[RMAX];                 // slot i may be filled with chunk c iff free_gen[i] == c / R
  for (int i = 0; i < R; i++) { free_gen[i] = 0; ready_chunk[i] = ~0ull; }
  std::atomic<uint64_t> next_chunk(0);
  bool abort_all = false, failed = false, alloc_failed = false;
  std::atomic<uint64_t> bad_block(~0ull);
  auto worker = [&]() {
    z_stream z; memset(&z, 0, sizeof(z));
    if (inflateInit2(&z, -15) != Z_OK) { std::lock_guard<std::mutex> g(mu); failed = abort_all = true; cv.notify_all(); return; }
    for (;;) {
      const uint64_t c = next_chunk.fetch_add(1);
      if (c >= nchunks) break;
      const int slot = (int)(c % R);
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return abort_all || free_gen[slot] == c / R; });
        if (abort_all) break;
      }
      if (!ring[slot]) {                                      // first use of this slot (exactly one worker gets here per slot)
        (void)hipSetDevice(s->device);
        if (hipHostMalloc(reinterpret_cast<void **>(&ring[slot]), CH, hipHostMallocDefault) != hipSuccess) {
          ring[slot] = nullptr;
          std::lock_guard<std::mutex> g(mu);
          alloc_failed = failed = abort_all = true;
          cv.notify_all();
          break;
        }
      }
      bool ok = true;
      size_t at = 0;
      for (size_t i = chunks[c].first; i < chunks[c].last && ok; i++) {
        ok = inflate_block(z, blocks[i], reinterpret_cast<unsigned char *>(ring[slot]) + at);
        if (!ok) bad_block.store(i);
        at += blocks[i].isize;
      }
      std::lock_guard<std::mutex> g(mu);
      if (!ok) { failed = abort_all = true; }
      ready_chunk[slot] = c;
      cv.notify_all();
    }
    inflateEnd(&z);
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; t++) pool.emplace_back(worker);
  const bool trace = getenv("MGC_IO_TRACE") != nullptr;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_wait = 0, t_submit = 0;
  uint64_t text_total = 0;
  for (uint64_t c = 0; c < nchunks && rc == MGC_OK; c++) {
    const int slot = (int)(c % R);
    const double t0 = now();
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return abort_all || ready_chunk[slot] == c; });
      if (abort_all) break;
    }
    const double t1 = now();
    t_wait += t1 - t0;
    if (chunks[c].text) rc = text_submit(s, ring[slot], chunks[c].text);       // (a chunk of empty blocks: the end-of-file marker)
    text_total += chunks[c].text;
    t_submit += now() - t1;
    if (c >= 2) { std::lock_guard<std::mutex> g(mu); free_gen[(c - 2) % R]++; cv.notify_all(); }
  }
  { std::lock_guard<std::mutex> g(mu); if (rc != MGC_OK) abort_all = true; cv.notify_all(); }
  for (int b = 0; b < 2; b++) if (s->text_ev_used[b]) (void)hipEventSynchronize(s->text_ev[b]);   // the last uploads still read from the ring
  { std::lock_guard<std::mutex> g(mu); abort_all = true; cv.notify_all(); }
  for (auto &t : pool) t.join();
  unmap();
  if (trace)
    fprintf(stderr, "[io] BGZF file %.2f GB -> %.2f GB of text in %llu chunks (%zu blocks), %d inflaters, ring %d: waiting for the inflaters %.3f s, "
                    "upload+parse submit (incl. waits for the device) %.3f s\n", fsize / 1e9, text_total / 1e9, (unsigned long long)nchunks, blocks.size(),
            threads, R, t_wait, t_submit);
  if (alloc_failed) { set_err(&s->err, "mgc_push_text_bgzf_file: pinned buffers: out of memory"); (void)mgc_end_text(s); return MGC_ENOMEM; }
  if (failed) {
    // what the file has put into the stream is taken back (as for a file that stops being strict FASTQ), unless part of it already went
    // into a counted batch
    const unsigned long long bb = (unsigned long long)bad_block.load();
    s->text_open = false;
    if (s->text_cut_in_file) { set_err(&s->err, "'%s': BGZF block %llu failed to inflate after part of the file was counted (input larger than one batch)", path, bb); return MGC_EINVAL; }
    HIP_TRY(s, mgc::launch_text_file_op(s->buf[mgc_session::B_TEXT_STATE].p, stage_ptr(s, s->fill), 2, s->st_in));
    const int rrc = resolve_length(s);
    if (rrc != MGC_OK) return rrc;
    set_err(&s->err, "'%s': BGZF block %llu failed to inflate (corrupt file)", path, bb);
    return MGC_EFORMAT;
  }
  const int rc_end = mgc_end_text(s);                        // closes the file in every case (rolls it back on MGC_EFORMAT)
  return rc != MGC_OK ? rc : rc_end;
}

extern "C" int mgc_end_text(mgc_session *s) {
  if (!s) return MGC_EINVAL;
  if (!s->text_open) { set_err(&s->err, "mgc_end_text without mgc_begin_text"); return MGC_ESTATE; }
  HIP_TRY(s, hipSetDevice(s->device));
  s->text_open = false;
  HIP_TRY(s, s->ensure_preserve(stage_id(s->fill), s->fill_len + 4096, s->fill_len, s->st_in));
  HIP_TRY(s, mgc::launch_text_file_op(s->buf[mgc_session::B_TEXT_STATE].p, stage_ptr(s, s->fill), 1, s->st_in));
  HostParseState h;
  int rc = resolve_length(s, &h);
  if (rc != MGC_OK) return rc;
  if (h.error) {
    if (s->text_cut_in_file) {
      // part of this file went into a batch that is already counted: it cannot be taken back
      set_err(&s->err, "the file stops being strict four-line FASTQ after part of it was counted (input larger than one batch): "
                       "convert it, or feed it through mgc_push_bases from the start");
      return MGC_EINVAL;
    }
    HIP_TRY(s, mgc::launch_text_file_op(s->buf[mgc_session::B_TEXT_STATE].p, stage_ptr(s, s->fill), 2, s->st_in));
    rc = resolve_length(s);
    if (rc != MGC_OK) return rc;
    set_err(&s->err, "the file is not strict four-line FASTQ: feed it through mgc_push_bases (meryl_seq.h reader)");
    return MGC_EFORMAT;
  }
  if (s->fill_len >= s->batch_limit) return cut_batch(s);
  return MGC_OK;
}

extern "C" int mgc_push_bases_device(mgc_session *s, const uint8_t *d_bases, uint64_t n_bases) {
  if (!s || (!d_bases && n_bases)) return MGC_EINVAL;
  if (s->borrowed || s->input_seen || s->counted) { set_err(&s->err, "device input must be the only input"); return MGC_ESTATE; }
  s->d_bases = d_bases;
  s->n_bases = n_bases;
  s->borrowed = true;
  return MGC_OK;
}

extern "C" int mgc_set_profiling(mgc_session *s, int enable) {
  if (!s) return MGC_EINVAL;
  s->profiling = enable != 0;
  return MGC_OK;
}

extern "C" int mgc_get_profile(const mgc_session *s, mgc_profile *p) {
  if (!s || !p) return MGC_EINVAL;
  *p = s->prof;
  return MGC_OK;
}

namespace {
struct DevBuf {                                   // frees on scope exit
  void *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 256); }
  template <typename T> T *as() { return reinterpret_cast<T *>(p); }
  void *release() { void *q = p; p = nullptr; return q; }
};

struct StageTimer {
  bool on;
  hipStream_t st;
  hipEvent_t ev[MGC_NUM_STAGES][2];
  bool used[MGC_NUM_STAGES];
  StageTimer(bool enable, hipStream_t s) : on(enable), st(s) {
    for (int i = 0; i < MGC_NUM_STAGES; i++) {
      used[i] = false;
      if (on) { (void)hipEventCreate(&ev[i][0]); (void)hipEventCreate(&ev[i][1]); }
    }
  }
  ~StageTimer() {
    if (on) for (int i = 0; i < MGC_NUM_STAGES; i++) { (void)hipEventDestroy(ev[i][0]); (void)hipEventDestroy(ev[i][1]); }
  }
  void begin(int i) { if (on) { (void)hipEventRecord(ev[i][0], st); used[i] = true; } }
  void end(int i)   { if (on) (void)hipEventRecord(ev[i][1], st); }
};
}  // namespace

// One pass over bases that are resident in HBM (s->d_bases / s->n_bases): results stay in HBM.
// ext_keys/ext_counts: k-mers already extracted and grouped by file by the caller (the owner side of a sharded
// count): extraction and partition are skipped, the caller's buffer is processed in place.
static int count_device(mgc_session *s, void *ext_keys = nullptr, const uint64_t *ext_counts = nullptr,
                        uint32_t ext_bucket_bits = MGC_NUM_FILES_BITS) {
  join_prepare(s);
  s->free_result();
  HIP_TRY(s, hipSetDevice(s->device));
  hipStream_t st = s->stream;
  const mgc_count_config &c = s->cfg;
  const mgc::Switches &sw = s->sw;
  const uint32_t k = c.k;
  // buckets = the 64 files, or (sharded owner side) finer top-bit ranges of the k-mer: 2^bucket_bits of them
  // The session's own partition uses the files while a file stays within what two grouping digits cover
  // (1152 << 18 = 302 M k-mers); larger inputs are partitioned one or more bits finer -- the 64 files are ranges
  // of buckets either way.
  uint32_t bucket_bits = ext_keys ? ext_bucket_bits : (uint32_t)MGC_NUM_FILES_BITS;
  if (!ext_keys) {
    // `compress`: two dense-rank digits cover 3^10 sub-buckets (below), i.e. buckets of up to 68 M k-mers, and the digits
    // are whole bases, so the buckets get finer two bits at a time
    // (`compress` buckets are uneven -- a canonical k-mer starts with A or C twice as often as with G or T, and 36 of the 64 / 108 of the
    // 256 bucket prefixes repeat no base -- so the largest bucket of a 10 Gbp input at 256 buckets holds ~100 M k-mers: above the 68 M of the
    // index-claimed tables.  Round 6: such a bucket keeps its two dense-rank digits and is counted by the distinct-sized kernel,
    // sub-buckets of up to 2304 k-mers on average, instead of falling back to the stable sort: hpc_stream[] below.)
    auto per_bucket = [&](uint32_t) -> uint64_t {
      if (sw.bucket_bases) return sw.bucket_bases;            // tests force finer buckets on small inputs
      return c.homopoly_compress ? 60000000ull : 180000000ull;
    };
    const uint32_t step = c.homopoly_compress ? 2u : 1u;
    while (bucket_bits + step <= MGC_MAX_BUCKET_BITS && bucket_bits + step <= 2 * c.k && (s->n_bases >> bucket_bits) > per_bucket(bucket_bits)) bucket_bits += step;
  }
  const uint32_t nb = 1u << bucket_bits;
  const uint32_t kw = s->key_words;
  const size_t   kbytes = sizeof(uint64_t) * kw;
  memset(&s->prof, 0, sizeof(s->prof));

  // ---- `compress`: homopolymer-compress the base stream on the device (merylInput.C:261-268) ----
  const uint8_t *d_bases = s->d_bases;
  uint64_t n_bases = s->n_bases;
  if (c.homopoly_compress && n_bases && !ext_keys) {
    HIP_TRY(s, s->ensure(mgc_session::B_HPC, n_bases));
    HIP_TRY(s, s->ensure(mgc_session::B_HPC_WS, mgc::hpc_workspace_bytes(n_bases)));
    uint8_t *d_hpc = reinterpret_cast<uint8_t *>(s->buf[mgc_session::B_HPC].p);
    void *hws = s->buf[mgc_session::B_HPC_WS].p;
    HIP_TRY(s, mgc::launch_homopoly_compress(d_bases, n_bases, d_hpc, hws, st));
    uint64_t n_out = 0;
    HIP_TRY(s, hipMemcpyAsync(&n_out, hws, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(s, hipStreamSynchronize(st));
    d_bases = d_hpc;
    n_bases = n_out;
  }

  hipEvent_t ev_all[2];
  if (s->profiling) { (void)hipEventCreate(&ev_all[0]); (void)hipEventCreate(&ev_all[1]); (void)hipEventRecord(ev_all[0], st); }
  StageTimer tm(s->profiling, st);

  // ---- pass 1: per-file histogram ----
  HIP_TRY(s, s->ensure(mgc_session::B_PART_WS, mgc::kp_workspace_bytes(bucket_bits)));
  HIP_TRY(s, s->ensure(mgc_session::B_META, sizeof(uint64_t) * nb * 3));        // counts, starts, per-file K96 flags
  void *part_ws = s->buf[mgc_session::B_PART_WS].p;
  uint64_t *d_counts64 = reinterpret_cast<uint64_t *>(s->buf[mgc_session::B_META].p), *d_starts = d_counts64 + nb;
  std::vector<uint64_t> h_counts_v(nb), h_starts_v(nb + 1);
  uint64_t *h_counts = h_counts_v.data(), *h_starts = h_starts_v.data();
  // The narrowed grouping passes (k <= ~25, below) group a file by its TOP digit first when that digit's histogram is at
  // hand: the file histogram then counts fifteen top bits instead of six (one kernel, same read of the bases) and the
  // 8 B/k-mer digit-histogram read of every file goes away.
  const uint64_t *d_fine = nullptr;
  const uint64_t *d_fine_hpc = nullptr;                              // `compress`: the dense-rank form of that histogram
  if (!ext_keys) {
    tm.begin(MGC_STAGE_HISTOGRAM);
    // (two digits cover at most 18 bits: beyond 2k - 6 = 41 nothing narrows -- the files' WHOLE keys then take the same
    // high-digit-first passes, mgc::launch_group_wide, 16-byte keys included; MGC_WIDE_MSD=0: low digit first off a histogram
    // read of the keys, as `compress` always does: its digits are dense ranks)
    const bool wide_msd_on = sw.wide_msd && !c.homopoly_compress;
    if (n_bases >= (1u << 22) && ((kw == 1 && 2 * k - bucket_bits <= 41) || wide_msd_on) && mgc::kmer_histogram_fine_ok(k, bucket_bits, s->sfx_mask, sw)) {
      HIP_TRY(s, s->ensure(mgc_session::B_FINE, sizeof(uint64_t) << 15));
      uint64_t *fine = reinterpret_cast<uint64_t *>(s->buf[mgc_session::B_FINE].p);
      HIP_TRY(s, mgc::launch_kmer_histogram_fine(d_bases, n_bases, k, c.mode, d_counts64, fine, part_ws, st, sw.const_k));
      d_fine = fine;
    } else if (c.homopoly_compress && n_bases >= (1u << 22) && (2 * k - bucket_bits) % 2 == 0 && 2 * k - bucket_bits >= 20 &&
               sw.hpc_digits && mgc::kmer_histogram_hpc_ok(k, bucket_bits, s->sfx_mask, sw)) {
      // `compress`: k-mers per (bucket, dense-rank digit below it) -- the buckets' high digit goes first as well (MGC_HPC_MSD=0: off)
      HIP_TRY(s, s->ensure(mgc_session::B_FINE, sizeof(uint64_t) * std::max<size_t>((size_t)1 << 15, mgc::kmer_histogram_hpc_entries(bucket_bits))));
      uint64_t *fine = reinterpret_cast<uint64_t *>(s->buf[mgc_session::B_FINE].p);
      HIP_TRY(s, mgc::launch_kmer_histogram_hpc(d_bases, n_bases, k, c.mode, bucket_bits, d_counts64, fine, part_ws, st, sw.const_k));
      d_fine_hpc = fine;
    } else
    HIP_TRY(s, mgc::launch_kmer_histogram(d_bases, n_bases, k, c.mode, bucket_bits, d_counts64, part_ws, st, s->sfx_mask, s->sfx_test));
    tm.end(MGC_STAGE_HISTOGRAM);
    s->prof.stage_launches[MGC_STAGE_HISTOGRAM] = 1;
    HIP_TRY(s, hipMemcpyAsync(h_counts, d_counts64, sizeof(uint64_t) * nb, hipMemcpyDeviceToHost, st));
    HIP_TRY(s, hipStreamSynchronize(st));
  } else {
    memcpy(h_counts, ext_counts, sizeof(uint64_t) * nb);
    // the owner side of a sharded count: the senders' fifteen-bit histograms, summed over the ranks (mgc_count_buckets_into), give
    // every bucket's first grouping digit -- 15 - bucket_bits bits of it -- so that nobody reads the keys for a histogram here either
    uint64_t n_ext = 0;
    for (uint32_t b = 0; b < nb; b++) n_ext += h_counts[b];
    const bool wide_msd_on = sw.wide_msd && !c.homopoly_compress;
    if (s->ext_fine && sw.fine_hist && bucket_bits <= 8 && n_ext >= (1u << 22) && s->sfx_mask == 0 && !c.homopoly_compress &&
        2 * k >= 15 + 2 && ((kw == 1 && 2 * k - bucket_bits <= 41) || wide_msd_on))
      d_fine = s->ext_fine;
  }
  const uint32_t fine_bits = 15u - bucket_bits;              // bits of a bucket's first digit the fifteen-bit histogram knows (9 for the 64 files)
  uint64_t N = 0, max_bucket = 0;
  memset(s->file_instances, 0, sizeof(s->file_instances));
  for (uint32_t b = 0; b < nb; b++) {
    h_starts[b] = N;
    N += h_counts[b];
    max_bucket = std::max(max_bucket, h_counts[b]);
    s->file_instances[b >> (bucket_bits - MGC_NUM_FILES_BITS)] += h_counts[b];
  }
  h_starts[nb] = N;
  s->n_instances = N;

  // ---- pass 2: pack + scatter into per-file regions ----
  // Two ways from file-grouped k-mers to the (k-mer, count) stream:
  //   finish (default): LSB-sort only the top t bits of every file globally, then sort the low bits of
  //                     every sub-bucket in LDS with the run-length count fused in (mgc_finish.hip);
  //   full   (MGC_FINISH=0, and the fallback for files with an oversized sub-bucket): LSB-sort all 2k-6
  //                     bits globally, then the separate run-length kernels.
  const bool use_finish = sw.finish;
  mgc::SortPlan plan;
  mgc::make_sort_plan(0, 2 * k - bucket_bits, &plan);
  const bool odd = !use_finish && (plan.num_passes & 1u) != 0;
  if (!ext_keys) HIP_TRY(s, s->ensure(mgc_session::B_X, kbytes * N));
  HIP_TRY(s, s->ensure(mgc_session::B_Y, kbytes * (odd ? N : max_bucket)));
  unsigned char *X = ext_keys ? reinterpret_cast<unsigned char *>(ext_keys) : reinterpret_cast<unsigned char *>(s->buf[mgc_session::B_X].p);
  unsigned char *Y = reinterpret_cast<unsigned char *>(s->buf[mgc_session::B_Y].p);
  // The partition is launched once the plan of the files is known (below): when every file takes the narrowed passes its
  // k-mers leave as 5 bytes (u32 + u8 per file) instead of 8 -- the file's first grouping pass puts them together again.
  bool partition_done = false;
  std::vector<uint64_t> h_k96flags(nb, 0);                   // (lives until the partition's copy has been issued and the stream synchronised)
  auto run_partition = [&](bool soa, bool k96 = false) -> int {
    if (ext_keys || partition_done) return MGC_OK;
    partition_done = true;
    HIP_TRY(s, hipMemcpyAsync(d_starts, h_starts, sizeof(uint64_t) * nb, hipMemcpyHostToDevice, st));
    uint64_t *d_k96flags = d_starts + nb;
    if (k96) HIP_TRY(s, hipMemcpyAsync(d_k96flags, h_k96flags.data(), sizeof(uint64_t) * nb, hipMemcpyHostToDevice, st));
    tm.begin(MGC_STAGE_PARTITION);
    HIP_TRY(s, mgc::launch_kmer_partition(d_bases, n_bases, k, c.mode, bucket_bits, d_starts, (void *)X, part_ws, st,
                                          s->sfx_mask, s->sfx_test, k96 ? d_k96flags : (soa ? d_counts64 : nullptr), sw.const_k));
    tm.end(MGC_STAGE_PARTITION);
    s->prof.hist_bytes = n_bases;
    s->prof.partition_bytes = n_bases;
    for (uint32_t b = 0; b < nb; b++) s->prof.partition_bytes += h_counts[b] * (soa ? 5u : ((k96 && h_k96flags[b]) ? 12u : (uint64_t)kbytes));
    s->prof.stage_launches[MGC_STAGE_PARTITION] = 2;
    return MGC_OK;
  };
  uint32_t soa_hi_mask = 0;                                  // nonzero: the files lie in the 5-byte layout
  std::vector<char> file_k96(nb, 0), k96_passes(nb, 0);      // files that lie as 12-byte K96 records (k = 33..51; below); ... and whose passes moved them

  // ---- per-file LSB radix sort of the low 2k-6 bits ----
  const size_t sort_ws_bytes = mgc::sort_workspace_bytes(max_bucket) + 256;
  HIP_TRY(s, s->ensure(mgc_session::B_SORT_WS, sort_ws_bytes));
  void *sort_ws = s->buf[mgc_session::B_SORT_WS].p;
  // device flags: [0] look-back timeout, [1] scratch answer of the hash probe, [2] overflow of a streamed sub-bucket
  uint32_t *d_err = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(sort_ws) + sort_ws_bytes - 256);
  HIP_TRY(s, hipMemsetAsync(d_err, 0, 32, st));

  const uint32_t rem_bits = 2 * k - bucket_bits;
  const uint32_t ev_per_file = 2 * 16;                     // room for 16 passes per file
  std::vector<hipEvent_t> pass_ev;
  std::vector<uint32_t> file_passes(nb, 0);
  std::vector<char> narrowed(nb, 0);                       // files whose grouping passes ran on 32-bit words
  if (s->profiling) {
    pass_ev.resize((size_t)nb * ev_per_file);
    for (auto &e : pass_ev) (void)hipEventCreate(&e);
  }
  uint32_t sort_launch_groups = 0;
  uint64_t nd = 0;

  if (!use_finish) {
    { const int prc = run_partition(false); if (prc != MGC_OK) return prc; }
    tm.begin(MGC_STAGE_SORT);
    for (uint32_t b = 0; b < nb; b++) {
      if (h_counts[b] == 0) continue;
      void *src = X + kbytes * h_starts[b];
      void *alt = odd ? (void *)(Y + kbytes * h_starts[b]) : (void *)Y;
      int in_alt = 0;
      hipEvent_t *pe = s->profiling ? &pass_ev[(size_t)b * ev_per_file] : nullptr;
      HIP_TRY(s, mgc::launch_radix_sort(src, alt, h_counts[b], kw, plan, sort_ws, sort_ws_bytes - 256, d_err, &in_alt, st, pe));
      file_passes[b] = plan.num_passes;
      sort_launch_groups++;
      (void)in_alt;   // odd pass count: every file ends in Y at the same offsets; even: back in X
    }
    tm.end(MGC_STAGE_SORT);
    void *d_sorted = odd ? (void *)Y : (void *)X;

    // ---- run-length count ----
    HIP_TRY(s, s->ensure(mgc_session::B_RLE_WS, mgc::rle_workspace_bytes(N)));
    void *rle_ws = s->buf[mgc_session::B_RLE_WS].p;
    tm.begin(MGC_STAGE_RLE);
    HIP_TRY(s, mgc::launch_rle_count(d_sorted, N, kw, rle_ws, st));
    HIP_TRY(s, mgc::rle_read_total(rle_ws, &nd, st));
    s->n_distinct = nd;
    HIP_TRY(s, s->ensure(mgc_session::B_UNIQUE, kbytes * nd));
    HIP_TRY(s, s->ensure(mgc_session::B_COUNTS, sizeof(uint32_t) * nd));
    s->d_unique = s->buf[mgc_session::B_UNIQUE].p;
    s->d_counts = reinterpret_cast<uint32_t *>(s->buf[mgc_session::B_COUNTS].p);
    HIP_TRY(s, mgc::launch_rle_emit(d_sorted, N, kw, rle_ws, s->d_unique, s->d_counts, st));
    tm.end(MGC_STAGE_RLE);
    s->prof.stage_launches[MGC_STAGE_RLE] = 3;
  } else {
    // ---- plan: per file, t top bits so that a sub-bucket holds ~target k-mers ----
    const uint64_t target = mgc::finish_target_for(kw, sw), cap = mgc::finish_capacity_for(kw);
    std::vector<uint32_t> top_bits(nb);
    std::vector<char> hpc_digits(nb, 0);
    std::vector<uint64_t> gbase(nb + 1), sbase(nb + 1);
    gbase[0] = sbase[0] = 0;
    // fstream[b] (round 6): the file takes the DISTINCT-sized count (hash_count_stream_kernel: up to 4094 keys per sub-bucket streamed
    // through a table that holds ~1280 distinct suffixes) and with it one grouping bit fewer -- sub-buckets of 1152..2304 k-mers on
    // average instead of 576..1152, so that a file of up to 302 M k-mers groups by an eight-bit first digit (256-byte runs out of
    // the 16384-key tiles instead of 128-byte ones).  Narrowed files whose suffix fits the packed entry (8..20 bits).
    std::vector<char> fstream(nb, 0);
    std::vector<uint32_t> top_str(nb, 0);                              // the candidate: grouping bits under that plan (0: none)
    const bool stream_on = sw.hash_stream != 0 && kw == 1 && !c.homopoly_compress;
    const uint64_t starget = mgc::finish_stream_target(sw);
    // `compress`: the grouping digits are dense ranks of five homopolymer-free bases (make_hpc_group_plan): 10 key bits
    // hold 243 patterns, 20 bits 59049.  Needs the remaining bits to be whole bases (the 64 files, or an even number of
    // bucket bits) and the bucket to fit 59049 sub-buckets; otherwise the generic bit digits below (MGC_HPC_DIGITS=0: always).
    const bool hpc_ok = sw.hpc_digits && c.homopoly_compress && (rem_bits % 2 == 0) && bucket_bits >= 2;
    // hpc_stream[b] (round 6): a two-digit `compress` bucket whose sub-buckets average more than the index-claimed tables take counts
    // its whole 8-byte k-mers with the distinct-sized kernel (64-bit entries): everything the high-digit-first passes of such a
    // bucket need is known here (the dense-rank histogram is at hand, the suffix has 32..52 bits), so the plan is final at once
    const bool hpc_stream_ok = hpc_ok && sw.hash_stream != 0 && sw.hash_stream != 2 && kw == 1 && d_fine_hpc && nb <= 256 && sw.wide_msd &&
                               rem_bits >= 20 + 32 && mgc::finish_stream_ok(kw, rem_bits - 20, false);
    std::vector<char> hpc_stream(nb, 0), hpc_cand(nb, 0);
    // hpc_mixed[b]: a candidate bucket of a size at which 3^9 sub-buckets (dense-rank high digit + the plain eight bits of four bases:
    // make_hpc_mixed_plan) average what the distinct-sized count likes (0.3 .. 1 of its target: 700 .. 2304 k-mers) -- 3^10 of them hold a few hundred k-mers
    // each at 5 Gbp and the count kernel's per-sub-bucket steps dominate.  Taken where the probe says coverage is high.
    std::vector<char> hpc_mixed(nb, 0);
    const bool hpc_mixed_ok = hpc_stream_ok && rem_bits >= 18 + 32 && mgc::finish_stream_ok(kw, rem_bits - 18, false) && mgc::finish_can_stream(kw, rem_bits - 18);
    for (uint32_t b = 0; b < nb; b++) {
      uint32_t t = 0;
      if (hpc_ok && h_counts[b] > target) {                            // sub-buckets average `target` k-mers or fewer
        if (h_counts[b] <= 243ull * target && rem_bits >= 10) t = 10;
        else if (h_counts[b] <= 59049ull * target && rem_bits >= 20) t = 20;
        else if (hpc_stream_ok && h_counts[b] <= 59049ull * starget && h_counts[b] < (1ull << 30)) { t = 20; hpc_stream[b] = 1; }
        if (t == 20 && sw.hash_stream == 1 && hpc_stream_ok && h_counts[b] < (1ull << 30)) hpc_stream[b] = 1;   // (tests, A/B: every two-digit bucket)
        if (t) hpc_digits[b] = 1;                                      // else (tiny k, gigantic bucket): generic path
      }
      if (!hpc_digits[b]) {
        while (t < rem_bits && t < 26 && (h_counts[b] >> t) > target) t++;
        // tests reach the large-input plans (two nine-bit digits, 18-bit suffixes at k = 21) on small inputs
        if (sw.min_top) { const uint32_t m = sw.min_top; if (h_counts[b] && t < m) t = m < rem_bits ? m : rem_bits; }
        if (stream_on && h_counts[b]) {
          uint32_t ts = 0;
          while (ts < rem_bits && ts < 26 && (h_counts[b] >> ts) > starget) ts++;
          if (sw.min_top) { const uint32_t m = sw.min_top; if (ts < m) ts = m < rem_bits ? m : rem_bits; }
          top_str[b] = ts;                                             // (clamped and validated once the file's kind of passes is known: below)
        }
      }
      if (c.homopoly_compress && t && !hpc_digits[b]) {
        // homopolymer-compressed sequence never repeats a base: every 2-bit group after the first takes 3 of its 4
        // values, so only (3/4)^(t/2) of the 2^t top-bit patterns occur and the occupied sub-buckets are that much
        // larger than planned: log2(4/3)/2 = 0.2075 of every key bit carries no information
        const double scale = 1.0 / (1.0 - 0.2075);
        const uint32_t tc = (uint32_t)((double)t * scale + 0.5);
        t = tc < rem_bits ? (tc < 26 ? tc : 26) : rem_bits;
      }
      top_bits[b] = t;
    }

    // ---- A. global LSB passes on the top bits only ----
    // the finish only needs the file grouped by its top bits
    std::vector<mgc::SortPlan> fplan(nb);
    // narrow[b]: the file's k-mers travel as 32-bit words from the first grouping pass on (mgc::launch_group_narrow)
    std::vector<char> narrow(nb, 0);
    // wide_msd[b]: the file's whole keys take the high-digit-first passes (mgc::launch_group_wide)
    std::vector<char> wide_msd(nb, 0);
    std::vector<uint32_t> tr_a(nb, 0), tr_b(nb, 0);        // ... and in which order its sub-buckets lie (mgc::tr_index)
    const size_t hdr_bytes = mgc::sort_header_bytes();
    // msd_ok[b]: the histogram of the bucket's HIGH digit is at hand (the fifteen-bit histogram holds fine_bits bits below the bucket:
    // a plan whose high digit is wider gives bits to the low one; a low digit that would pass nine bits keeps the low digit first)
    std::vector<char> msd_ok(nb, 1);
    auto fit_split = [&](uint32_t b) {
      mgc::SortPlan &fp = fplan[b];
      if (!d_fine || fp.num_passes != 2 || fp.hpc || fp.pass_bits[1] <= fine_bits) return;
      const uint32_t t = fp.pass_bits[0] + fp.pass_bits[1];
      if (t - fine_bits > 9) { msd_ok[b] = 0; return; }
      fp.pass_bits[1] = fine_bits; fp.pass_bits[0] = t - fine_bits;
      fp.pass_shift[1] = fp.pass_shift[0] + fp.pass_bits[0];
    };
    for (uint32_t b = 0; b < nb; b++) {
      if (h_counts[b] == 0 || top_bits[b] == 0) continue;
      if (hpc_digits[b]) {
        mgc::make_hpc_group_plan(rem_bits - top_bits[b], top_bits[b] / 10, &fplan[b]);
        // (sub-bucket numbers made of dense ranks are no key bits: the kernels that put a k-mer's top bits back from its sub-bucket
        // number -- 32-bit suffixes -- stay with the low digit first)
        wide_msd[b] = d_fine_hpc && nb <= 256 && top_bits[b] == 20 && (kw == 2 || rem_bits - top_bits[b] >= 32) &&
                      mgc::finish_can_stream(kw, rem_bits - top_bits[b]) && mgc::sort_plan_wide_msd(fplan[b], h_counts[b], sw.wide_msd);
        // (above the index-claimed tables' reach: the distinct-sized count whatever the coverage; below: a candidate the probe file decides on)
        if (hpc_stream[b] && wide_msd[b]) { fstream[b] = 1; s->prof.stream_files++; }
        else if (hpc_stream_ok && wide_msd[b] && top_bits[b] == 20 && h_counts[b] < (1ull << 30)) hpc_cand[b] = 1;
        if ((hpc_cand[b] || (fstream[b] && sw.hash_stream == 1)) && hpc_mixed_ok && h_counts[b] >= 19683ull * (starget * 3 / 10) &&
            h_counts[b] <= 19683ull * starget) {
          mgc::SortPlan mp;
          mgc::make_hpc_mixed_plan(rem_bits - 18, &mp);
          hpc_mixed[b] = mgc::sort_plan_wide_msd(mp, h_counts[b], sw.wide_msd) ? 1 : 0;
          if (hpc_mixed[b] && sw.hash_stream == 1) {                   // (tests, A/B: no probe)
            top_bits[b] = 18; fplan[b] = mp; s->prof.hpc_mixed_files++;
            if (!fstream[b]) { fstream[b] = 1; s->prof.stream_files++; }
            hpc_mixed[b] = 0; hpc_cand[b] = 0;
          }
        }
        continue;
      }
      mgc::make_sort_plan(rem_bits - top_bits[b], rem_bits, &fplan[b]);
      if (fplan[b].mode == 0) fplan[b].mode = 3;
      fit_split(b);
      const uint32_t low = rem_bits - top_bits[b];
      narrow[b] = low < 32 && mgc::finish_can_stream(kw, low) && mgc::sort_plan_narrows(fplan[b], h_counts[b], kw, sw.narrow);
      // (only the hash-count kernels translate the sub-bucket numbers of whole keys)
      wide_msd[b] = !narrow[b] && d_fine && nb <= 256 && msd_ok[b] && mgc::finish_can_stream(kw, low) &&
                    mgc::sort_plan_wide_msd(fplan[b], h_counts[b], sw.wide_msd);
      if (top_str[b]) {
        // the coarser plan stays a candidate if the file narrows under BOTH plans, its suffix then has to fit the packed 32-bit entry
        // (20 bits) -- or if its whole 8-byte k-mers take the high-digit-first passes under both (k = 24..32: 64-bit entries, 52 bits)
        uint32_t ts = top_str[b];
        const uint32_t t = top_bits[b], max_low = narrow[b] ? 20u : 52u;
        if (rem_bits - ts > max_low) ts = rem_bits - max_low;
        // (a plan that does not coarsen the file keeps the kernels it has -- unless MGC_HASH_STREAM=1 asks for the new one)
        bool ok = ts >= 1 && ts <= t && ts <= 18 && (ts < t || sw.hash_stream == 1) && (narrow[b] || (wide_msd[b] && kw == 1 && sw.hash_stream != 2)) &&
                  h_counts[b] < (1ull << 32) && mgc::finish_stream_ok(kw, rem_bits - ts, narrow[b] != 0) && mgc::finish_can_stream(kw, rem_bits - ts);
        if (ok) {
          mgc::SortPlan sp;
          mgc::make_sort_plan(rem_bits - ts, rem_bits, &sp);
          if (sp.mode == 0) sp.mode = 3;
          if (narrow[b]) ok = mgc::sort_plan_narrows(sp, h_counts[b], kw, sw.narrow);
          else {
            // (whole keys need the high digit's histogram at hand under the coarser plan as well: fit_split's test)
            const bool split_ok = !(d_fine && sp.num_passes == 2 && sp.pass_bits[1] > fine_bits && ts - fine_bits > 9);
            ok = split_ok && mgc::sort_plan_wide_msd(sp, h_counts[b], sw.wide_msd);
          }
        }
        top_str[b] = ok ? ts : 0;
      }
    }
    // a file's sub-bucket tables are laid out for the FINER of its two plans (2^top_bits slots); ngf(b) of them are in use
    for (uint32_t b = 0; b < nb; b++) {
      const uint64_t ng = h_counts[b] ? ((uint64_t)1 << top_bits[b]) : 0;
      gbase[b + 1] = gbase[b] + ng;
      sbase[b + 1] = sbase[b] + (ng ? ng + 1 : 0);
    }
    auto ngf = [&](uint32_t b) -> uint64_t { return h_counts[b] ? ((uint64_t)1 << top_bits[b]) : 0; };
    auto take_mixed_plan = [&](uint32_t b) {
      top_bits[b] = 18;
      mgc::make_hpc_mixed_plan(rem_bits - 18, &fplan[b]);
      s->prof.hpc_mixed_files++;
    };
    auto take_stream_plan = [&](uint32_t b) {                // the candidate becomes the file's plan (narrow[] / wide_msd[] stay as they are)
      fstream[b] = 1; top_bits[b] = top_str[b];
      mgc::make_sort_plan(rem_bits - top_bits[b], rem_bits, &fplan[b]);
      if (fplan[b].mode == 0) fplan[b].mode = 3;
      msd_ok[b] = 1;
      fit_split(b);
      s->prof.stream_files++;
    };
    // Which plan?  The distinct-sized count pays off when a sub-bucket's distinct suffixes are few against its keys (measured at
    // 10 Gbp, profiles/r06_coverage_ab.txt: D / N = 0.14 -> -3.6 ms, 0.24 -> -1.5, 0.45 -> +3, 0.72 -> +50: above its table the retry
    // launch counts the sub-bucket a second time), and D / N is not known before something has been counted: ONE file -- the PROBE
    // file, the smallest one that is still a fair sample -- goes through its passes and its count first, on the finer plan; its
    // distinct / instances ratio (one 8-byte copy) decides for the others.  MGC_HASH_STREAM=1: every candidate, no probe; 0: none.
    int probe = -1;
    {
      bool any_cand = false;
      for (uint32_t b = 0; b < nb; b++) any_cand = any_cand || top_str[b] != 0 || hpc_cand[b] != 0;
      if (any_cand && sw.hash_stream == 1) { for (uint32_t b = 0; b < nb; b++) if (top_str[b]) take_stream_plan(b); }
      else if (any_cand) {
        uint64_t best = ~0ull;
        for (uint32_t b = 0; b < nb; b++)
          if (h_counts[b] >= max_bucket / 16 && h_counts[b] >= 4096 && h_counts[b] < best) { best = h_counts[b]; probe = (int)b; }
      }
    }
    const uint64_t ng_total = gbase[nb];
    HIP_TRY(s, s->ensure(mgc_session::B_SUBSTART, sizeof(uint64_t) * (sbase[nb] + 1)));
    HIP_TRY(s, s->ensure(mgc_session::B_GROUPS, sizeof(uint64_t) * (ng_total + 2 + 3 * (uint64_t)nb)));
    HIP_TRY(s, s->ensure(mgc_session::B_LARGE, sizeof(uint32_t) * (ng_total + 1)));
    HIP_TRY(s, s->ensure(mgc_session::B_NONEMPTY, sizeof(uint32_t) * (ng_total + 1) + sizeof(uint64_t) * 2 * ((uint64_t)nb + 1)));
    HIP_TRY(s, s->ensure(mgc_session::B_GSCAN, mgc::finish_scan_scratch_bytes(ng_total + 1)));
    HIP_TRY(s, s->ensure(mgc_session::B_RLE_WS, mgc::rle_workspace_bytes(max_bucket)));
    uint64_t *d_substart = reinterpret_cast<uint64_t *>(s->buf[mgc_session::B_SUBSTART].p);
    uint64_t *d_group    = reinterpret_cast<uint64_t *>(s->buf[mgc_session::B_GROUPS].p);   // [ng_total+1], then the files' statistics
    // per file, three words side by side (one small copy brings a file's back): [0] its largest sub-bucket, [1] how many are
    // above the persistent kernels' capacity, [2] how many are not empty
    uint64_t *d_stats    = d_group + ng_total + 1;
    auto d_maxsub  = [&](uint32_t b) { return d_stats + 3 * (size_t)b; };
    auto d_nlarge  = [&](uint32_t b) { return d_stats + 3 * (size_t)b + 1; };
    auto d_nzcount = [&](uint32_t b) { return d_stats + 3 * (size_t)b + 2; };
    uint32_t *d_large    = reinterpret_cast<uint32_t *>(s->buf[mgc_session::B_LARGE].p);
    uint64_t *d_retrycnt = reinterpret_cast<uint64_t *>(s->buf[mgc_session::B_NONEMPTY].p) + nb + 1;  // [nb] retry lists of the count kernels
    uint32_t *d_nz       = reinterpret_cast<uint32_t *>(d_retrycnt + nb + 1);                         // [ng_total] (a dense file's part: its retry list)
    HIP_TRY(s, hipMemsetAsync(s->buf[mgc_session::B_NONEMPTY].p, 0, sizeof(uint64_t) * 2 * ((size_t)nb + 1), st));
    HIP_TRY(s, hipMemsetAsync(d_group, 0, sizeof(uint64_t) * (ng_total + 1 + 3 * (size_t)nb), st));   // empty sub-buckets stay 0
    void     *rle_ws     = s->buf[mgc_session::B_RLE_WS].p;
    {
      // 5-byte layout: 8-byte keys with 33..40 bits below the file (k = 20..23), every non-empty file on the narrowed passes with
      // the high digit first off the fifteen-bit histogram (the instrumented instantiation reads whole keys).  MGC_SOA5=0: whole keys.
      bool soa = sw.soa5 && !ext_keys && kw == 1 && nb == 64 && d_fine && rem_bits > 32 && rem_bits <= 40 && s->sfx_mask == 0;
      for (uint32_t b = 0; b < nb && soa; b++) if (h_counts[b] && !(narrow[b] && top_bits[b])) soa = false;
      if (soa) soa_hi_mask = (1u << (rem_bits - 32)) - 1u;
      // K96 records (round 5): 16-byte keys with at most 96 bits below the file (k = 33..51), every non-empty file on the whole-key
      // high-digit-first passes: 12 of the 16 bytes leave the partition, go through both passes and into the count kernel
      // (mgc_common.hpp K96; the region of a file stays 16 bytes per k-mer, so a file can be widened back in place).  MGC_K96=0: whole keys.
      bool k96 = sw.k96 && !ext_keys && kw == 2 && nb == 64 && d_fine && rem_bits <= 96 && s->sfx_mask == 0 && !c.homopoly_compress;
      bool any96 = false;                                             // per file: the ones on the two-digit whole-key passes (a small file keeps 16-byte keys)
      for (uint32_t b = 0; b < nb && k96; b++) if (h_counts[b] && wide_msd[b] && top_bits[b]) { file_k96[b] = 1; h_k96flags[b] = 1; any96 = true; s->prof.k96_files++; }
      const int prc = run_partition(soa, k96 && any96);
      if (prc != MGC_OK) return prc;
    }
    // Where the counts of a file's distinct k-mers wait for the packing step (one uint32 per k-mer instance position).  A NARROWED
    // file keeps 4-byte words in the front half of its 8-byte region from the first grouping pass on: the back half is free and
    // takes the counts -- no buffer of its own (35 GB of the 123 GB arena at 10 Gbp; a large first hipMalloc is the slowest thing
    // a freshly started process does, profiles/r03m_e2e_io.txt).  The other files share B_CNT_TMP; a narrowed file that has to be
    // widened back later (a sub-bucket nothing can stream) gets a buffer of its own then.
    std::vector<uint32_t *> cnt_ptr(nb, nullptr);
    std::deque<DevBuf> cnt_extra;                           // (a deque: DevBuf owns its pointer and must not be relocated)
    {
      uint64_t wide_total = 0;
      for (uint32_t b = 0; b < nb; b++) if (!narrow[b]) wide_total += h_counts[b];
      HIP_TRY(s, s->ensure(mgc_session::B_CNT_TMP, sizeof(uint32_t) * wide_total));
      uint32_t *wide = reinterpret_cast<uint32_t *>(s->buf[mgc_session::B_CNT_TMP].p);
      uint64_t at = 0;
      for (uint32_t b = 0; b < nb; b++) {
        if (narrow[b]) cnt_ptr[b] = reinterpret_cast<uint32_t *>(X + kbytes * h_starts[b]) + h_counts[b];
        else { cnt_ptr[b] = wide + at; at += h_counts[b]; }
      }
    }
    tm.begin(MGC_STAGE_SORT);
    // high digit first: the headers of all narrowed files in one launch, their look-back granules zeroed in one memset
    std::vector<size_t> nws_off(nb + 1, 0);
    unsigned char *d_nws = nullptr, *d_nhdrs = nullptr;
    // (with a probe file: its header first, the others' once their plan is known)
    auto prepare_headers = [&](int only, int skip) -> int {
      unsigned char bits_a[256] = {0}, on[256] = {0};
      bool any = false;
      for (uint32_t b = 0; b < nb; b++) {
        if ((!narrow[b] && !wide_msd[b]) || !msd_ok[b] || (only >= 0 && (int)b != only) || (int)b == skip) continue;
        on[b] = 1; bits_a[b] = (unsigned char)fplan[b].pass_bits[1]; any = true;
      }
      if (any) HIP_TRY(s, mgc::launch_narrow_prepare(d_fine, nb, bits_a, on, d_nhdrs, st));
      return MGC_OK;
    };
    if (d_fine && nb <= 256) {
      bool any = false;
      for (uint32_t b = 0; b < nb; b++) {
        nws_off[b + 1] = nws_off[b];
        if ((!narrow[b] && !wide_msd[b]) || !msd_ok[b]) continue;
        any = true;
        nws_off[b + 1] += ((narrow[b] ? mgc::narrow_scratch_bytes(h_counts[b]) : mgc::wide_scratch_bytes(h_counts[b], kw)) + 255) / 256 * 256;
      }
      if (any) {
        HIP_TRY(s, s->ensure(mgc_session::B_SORT_HDRS, hdr_bytes * nb));
        HIP_TRY(s, s->ensure(mgc_session::B_NARROW_WS, nws_off[nb]));
        d_nhdrs = reinterpret_cast<unsigned char *>(s->buf[mgc_session::B_SORT_HDRS].p);
        d_nws = reinterpret_cast<unsigned char *>(s->buf[mgc_session::B_NARROW_WS].p);
        { const int prc = prepare_headers(probe >= 0 ? probe : -1, -1); if (prc != MGC_OK) return prc; }
        HIP_TRY(s, hipMemsetAsync(d_nws, 0, nws_off[nb], st));
      }
    }
    if (d_fine_hpc && nb <= 256) {
      uint64_t on[4] = {0, 0, 0, 0};
      bool any = false;
      for (uint32_t b = 0; b < nb; b++) {
        nws_off[b + 1] = nws_off[b];
        if (!wide_msd[b]) continue;
        on[b >> 6] |= 1ull << (b & 63u); any = true;
        nws_off[b + 1] += (mgc::wide_scratch_bytes(h_counts[b], kw) + 255) / 256 * 256;
      }
      if (any) {
        HIP_TRY(s, s->ensure(mgc_session::B_SORT_HDRS, hdr_bytes * nb));
        HIP_TRY(s, s->ensure(mgc_session::B_NARROW_WS, nws_off[nb]));
        d_nhdrs = reinterpret_cast<unsigned char *>(s->buf[mgc_session::B_SORT_HDRS].p);
        d_nws = reinterpret_cast<unsigned char *>(s->buf[mgc_session::B_NARROW_WS].p);
        HIP_TRY(s, mgc::launch_hpc_prepare(d_fine_hpc, bucket_bits, on, d_nhdrs, st));
        HIP_TRY(s, hipMemsetAsync(d_nws, 0, nws_off[nb], st));
      }
    }
    // Stage after stage: the passes of all files go to the session stream back to back, one synchronisation brings the files'
    // statistics back, then the count kernels run.  (A pipelined form -- a file's count kernel on another stream as soon as its
    // statistics are back, beside the passes of the files after it -- was built in round 4, measured slower and removed in round 5:
    // profiles/r04y_pipe_ab.txt, DESIGN_HISTORY.md.)
    if (s->h_stats_cap < 3 * (size_t)nb) {
      if (s->h_stats) { (void)hipHostFree(s->h_stats); s->h_stats = nullptr; s->h_stats_cap = 0; }
      HIP_TRY(s, hipHostMalloc(reinterpret_cast<void **>(&s->h_stats), sizeof(uint64_t) * 3 * (size_t)nb, hipHostMallocDefault));
      s->h_stats_cap = 3 * (size_t)nb;
    }
    unsigned char *huge_alt = Y;

    // ---- A + B/C of one file: its grouping passes, then its sub-bucket boundaries and its largest sub-bucket ----
    auto group_file = [&](uint32_t b) -> int {
      if (h_counts[b] == 0) return MGC_OK;
      const mgc::SortPlan &fp = fplan[b];
      void *src = X + kbytes * h_starts[b];
      int in_alt = 0;
      hipEvent_t *pe = s->profiling ? &pass_ev[(size_t)b * ev_per_file] : nullptr;
      if (top_bits[b] == 0) {
        // (a file of one sub-bucket: nothing to group)
      } else if (narrow[b]) {                                // X (8 B) -> Y (4 B) -> front of X (4 B); boundaries included
        const bool msd = d_nhdrs && d_nws && msd_ok[b];
        HIP_TRY(s, mgc::launch_group_narrow(src, (void *)Y, h_counts[b], fp, sort_ws, sort_ws_bytes - 256, d_err, d_substart + sbase[b], st, pe,
                                            msd ? (void *)(d_nhdrs + hdr_bytes * b) : nullptr, msd ? (void *)(d_nws + nws_off[b]) : nullptr,
                                            &tr_a[b], &tr_b[b], soa_hi_mask, sw.group_dbg, sw.group_pipe, sw.pass_stagger));
        file_passes[b] = 2;
        narrowed[b] = 1;
        sort_launch_groups++;
      } else if (wide_msd[b] && d_nhdrs) {                   // X -> Y -> X, whole keys; boundaries included
        HIP_TRY(s, mgc::launch_group_wide(src, (void *)Y, h_counts[b], kw, fp, d_err, d_substart + sbase[b], st, pe,
                                          (void *)(d_nhdrs + hdr_bytes * b), (void *)(d_nws + nws_off[b]), &tr_a[b], &tr_b[b], file_k96[b] != 0));
        file_passes[b] = 2;
        k96_passes[b] = file_k96[b];
        sort_launch_groups++;
        s->prof.wide_msd_files++;
      } else {
        wide_msd[b] = 0;
        HIP_TRY(s, mgc::launch_radix_sort(src, (void *)Y, h_counts[b], kw, fp, sort_ws, sort_ws_bytes - 256, d_err, &in_alt, st, pe));
        if (in_alt) HIP_TRY(s, hipMemcpyAsync(src, Y, kbytes * h_counts[b], hipMemcpyDeviceToDevice, st));
        file_passes[b] = fp.num_passes;
        sort_launch_groups++;
      }
      return MGC_OK;
    };
    auto stats_file = [&](uint32_t b) -> int {
      if (h_counts[b] == 0) return MGC_OK;
      if (narrow[b] || wide_msd[b])
        HIP_TRY(s, mgc::launch_subbucket_max(d_substart + sbase[b], kw, rem_bits - top_bits[b], top_bits[b], d_maxsub(b),
                                             d_large + gbase[b], d_nlarge(b), d_nz + gbase[b], d_nzcount(b), st,
                                             fstream[b] ? mgc::finish_stream_capacity() : 0));
      else
        HIP_TRY(s, mgc::launch_subbucket_bounds(X + kbytes * h_starts[b], h_counts[b], kw, rem_bits - top_bits[b], top_bits[b],
                                                d_substart + sbase[b], d_maxsub(b), d_large + gbase[b], d_nlarge(b),
                                                d_nz + gbase[b], d_nzcount(b), st));
      return MGC_OK;
    };
    std::vector<uint64_t> h_maxsub(nb, 0), h_nlarge(nb, 0), h_nzcount(nb, 0);
    uint32_t grouped = 0;                                    // files [0, grouped) have their passes on the session stream
    auto group_upto = [&](uint32_t end) -> int {
      for (; grouped < end && grouped < nb; grouped++) {
        if ((int)grouped != probe) { const int rc = group_file(grouped); if (rc != MGC_OK) return rc; }   // (the probe file went first)
        if (grouped + 1 == nb) {
          tm.end(MGC_STAGE_SORT);
          tm.begin(MGC_STAGE_RLE);
          // the small statistics kernels of all files back to back: they run beside each other
          for (uint32_t b = 0; b < nb; b++) { if ((int)b == probe) continue; const int rc2 = stats_file(b); if (rc2 != MGC_OK) return rc2; }
          HIP_TRY(s, hipMemcpyAsync(s->h_stats, d_stats, sizeof(uint64_t) * 3 * (size_t)nb, hipMemcpyDeviceToHost, st));
        }
      }
      return MGC_OK;
    };

    // ---- D. finish every file: LDS sort + count, or the full-sort fallback ----
    // The streaming kernel of a file's oversized sub-buckets goes to a second stream: it touches other sub-buckets than
    // the persistent kernel, and one gigantic sub-bucket occupies ONE workgroup for hundreds of microseconds -- beside
    // the persistent kernels of this and the next files that tail costs nothing.
    const bool fork_huge = s->stream2 != nullptr;
    hipStream_t st_huge = fork_huge ? s->stream2 : st;
    // The persistent kernels of odd files go to the second stream too, so that the tail of one file's launch overlaps the
    // head of the next: finish stage 58.5 -> 54.2 ms per 10 Gbp.  All streaming kernels stay on stream2: they share one
    // second buffer.
    const bool alt_files = fork_huge;
    bool forked = false, need_join = false;          // forked: stream2 is ordered after everything st holds that it must see
    // Round 6: the streaming kernels of different files on up to FOUR streams, each with a second buffer of its own.  One gigantic
    // sub-bucket (a repeat family's k-mers: 266 K keys at 30x of a 10 % repeat genome) occupies ONE workgroup for ~600 us; with every
    // file's streaming launch queued on one stream those tails added up to 39 ms of a 57 ms count stage (profiles/r06y: BASELINE config 3's
    // read shape at 10 Gbp) while the device had room for all of them at once.  MGC_HUGE_STREAMS=1: one stream (round 5).
    constexpr int NH = 1 + mgc_session::HUGE_EXTRA;
    hipStream_t hstream[NH];
    unsigned char *halt[NH];
    void *hws[NH] = {nullptr, nullptr, nullptr, nullptr};     // the sliced count of gigantic sub-buckets: one plan workspace per stream (launch_finish_file)
    const size_t hws_bytes = mgc::finish_huge_workspace_bytes(max_bucket);
    int n_huge_streams = 1, huge_next = 0;
    hstream[0] = st_huge; halt[0] = Y;
    auto huge_sync_all = [&]() -> int {               // (the host waits for every streaming kernel: Y and its siblings are free)
      for (int i = 0; i < n_huge_streams; i++) HIP_TRY(s, hipStreamSynchronize(hstream[i]));
      return MGC_OK;
    };
    auto huge_join_all = [&]() -> int {               // (st is ordered behind every streaming kernel)
      for (int i = 0; i < n_huge_streams; i++) {
        HIP_TRY(s, hipEventRecord(s->ev_join, hstream[i]));
        HIP_TRY(s, hipStreamWaitEvent(st, s->ev_join, 0));
      }
      return MGC_OK;
    };
    // (The streams are created on first need, behind all the others: HIP maps streams onto a few hardware queues in creation order, and
    // three more of them created at mgc_open put the two count streams on ONE queue -- their kernels no longer ran side by side, the
    // judged count stage went 25.4 -> 28.8 ms, profiles/r06_ab_runs.txt r06z.  Needed only where several files hold a GIGANTIC sub-bucket:
    // the many slightly oversized ones of an ordinary file -- 615 of up to 1946 keys in a dense file of the judged workload -- are short.)
    auto huge_setup = [&]() -> int {                  // once the files' statistics are back
      uint32_t files_gigantic = 0;
      for (uint32_t b = 0; b < nb; b++) if (h_counts[b] && s->h_stats[3 * (size_t)b + 1] != 0 && s->h_stats[3 * (size_t)b] > 16384) files_gigantic++;
      if (files_gigantic && sw.huge_slices) {
        HIP_TRY(s, s->ensure(mgc_session::B_HWS0, hws_bytes));
        hws[0] = s->buf[mgc_session::B_HWS0].p;
      }
      const int want = (int)std::min<uint64_t>((uint64_t)sw.huge_streams, (uint64_t)NH);
      if (!fork_huge || want <= 1 || files_gigantic < 2) return MGC_OK;
      for (int i = 1; i < want; i++) {
        if (!s->stream_h[i - 1] && hipStreamCreateWithFlags(&s->stream_h[i - 1], hipStreamNonBlocking) != hipSuccess) { s->stream_h[i - 1] = nullptr; (void)hipGetLastError(); }
        if (!s->stream_h[i - 1]) break;
        const int id = mgc_session::B_Y2 + (i - 1);
        HIP_TRY(s, s->ensure(id, kbytes * max_bucket));
        hstream[i] = s->stream_h[i - 1];
        halt[i] = reinterpret_cast<unsigned char *>(s->buf[id].p);
        if (sw.huge_slices) { HIP_TRY(s, s->ensure(mgc_session::B_HWS0 + i, hws_bytes)); hws[i] = s->buf[mgc_session::B_HWS0 + i].p; }
        n_huge_streams = i + 1;
      }
      return MGC_OK;
    };
    // tests run the dense-grid instantiations of the count kernels on small inputs (whose 2^t grids are mostly empty)
    const bool finish_nolist = sw.nolist;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> fin_ev;       // profiling: around every file's count-kernel launch
    uint64_t fin_keys = 0, fin_in_bytes = 0;
    bool fin_narrow = false;
    std::vector<uint64_t> h_fallback_distinct(nb);
    std::vector<char> fallback(nb);
    auto finish_file = [&](uint32_t b) -> int {
      fallback[b] = false;
      if (h_counts[b] == 0) return MGC_OK;
      const uint32_t low = rem_bits - top_bits[b];
      void *seg = X + kbytes * h_starts[b];
      // sub-buckets above the persistent kernels' capacity (a k-mer repeated thousands of times, a dense corner of the key
      // space) are streamed through a large hash table, in several suffix ranges if their distinct k-mers do not fit at once.
      // Only a gigantic one is asked about first (one pass must do), before anything touches the file
      bool stream = mgc::finish_can_stream(kw, low) && h_nlarge[b] > 0;
      if (stream && h_maxsub[b] > sw.stream_max && kw == 2) {
        stream = false;                                 // no probe for 16-byte keys: a sub-bucket that large takes the sort
      } else if (stream && h_maxsub[b] > sw.stream_max) {
        uint32_t h_fail[3] = {0, 0, 0};                 // [0] answer, [2] most distinct suffixes met (diagnostics)
        HIP_TRY(s, hipMemsetAsync(d_err + 4, 0, 12, st));
        HIP_TRY(s, mgc::launch_finish_probe(seg, kw, d_substart + sbase[b], low, h_nlarge[b], d_large + gbase[b], d_err + 4, st, sw.stream_max, narrow[b] != 0));
        HIP_TRY(s, hipMemcpyAsync(h_fail, d_err + 4, 12, hipMemcpyDeviceToHost, st));
        HIP_TRY(s, hipStreamSynchronize(st));
        stream = (h_fail[0] == 0);
        if (sw.finish_trace)
          fprintf(stderr, "[finish] bucket %u: largest sub-bucket %llu > %llu, up to %u distinct in one: %s\n", b,
                  (unsigned long long)h_maxsub[b], (unsigned long long)sw.stream_max, h_fail[2],
                  stream ? "streamed through the hash tables" : "too many distinct: stable-sort fallback");
      } else if (sw.finish_trace && h_maxsub[b] > cap) {
        fprintf(stderr, "[finish] bucket %u: largest sub-bucket %llu > %llu: %s\n", b, (unsigned long long)h_maxsub[b],
                (unsigned long long)cap, stream ? "streamed through the hash tables" : "stable-sort fallback");
      }
      if (file_k96[b] && h_nlarge[b] > 0 && !stream) {
        // K96 records with an oversized sub-bucket that nothing streams: the LDS sort / the stable-sort fallback want whole 16-byte
        // k-mers -- the file is widened in its own (16 bytes per k-mer) region, through Y, and goes on as a launch_group_wide file
        if (need_join) { const int jr = huge_sync_all(); if (jr != MGC_OK) return jr; }   // Y is the streaming kernels' second buffer
        forked = false;
        const unsigned __int128 fb = (unsigned __int128)b << rem_bits;
        HIP_TRY(s, mgc::launch_widen_k96(seg, h_counts[b], (uint64_t)fb, (uint64_t)(fb >> 64), (void *)Y, st));
        HIP_TRY(s, hipMemcpyAsync(seg, Y, kbytes * h_counts[b], hipMemcpyDeviceToDevice, st));
        file_k96[b] = 0;
        s->prof.k96_widened_files++;
      }
      bool unordered = false;
      if (narrow[b] && h_nlarge[b] > 0 && !stream) {
        // an oversized sub-bucket that cannot be streamed: the LDS sort / the stable-sort fallback want whole k-mers back
        if (need_join) { const int jr = huge_sync_all(); if (jr != MGC_OK) return jr; }   // Y is the streaming kernels' second buffer
        forked = false;
        HIP_TRY(s, mgc::launch_widen_groups(seg, d_substart + sbase[b], ngf(b), (uint64_t)b << rem_bits, low, (void *)Y, st,
                                            tr_a[b], tr_b[b]));
        HIP_TRY(s, hipMemcpyAsync(seg, Y, kbytes * h_counts[b], hipMemcpyDeviceToDevice, st));
        narrow[b] = 0;
        unordered = tr_a[b] != 0;      // grouped, but not in key order: only the stable sort of all bits can take it from here
        // (the coarser plan's oversized list was cut at ITS capacity: the whole-key kernels would miss the sub-buckets in between)
        if (fstream[b]) { fstream[b] = 0; unordered = true; }
        cnt_extra.emplace_back();      // the back half of its region holds k-mers again: counts of its own
        HIP_TRY(s, cnt_extra.back().alloc(sizeof(uint32_t) * h_counts[b]));
        cnt_ptr[b] = cnt_extra.back().as<uint32_t>();
      }
      // whole keys in (low digit : high digit) order whose oversized sub-buckets nothing streams: the stable sort of all bits
      if (wide_msd[b] && h_nlarge[b] > 0 && !stream) unordered = true;
      if ((h_maxsub[b] <= cap || stream) && !unordered) {
        hipStream_t fst;
        {
          const bool on_second = alt_files && (b & 1u);
          if ((stream || on_second) && fork_huge && !forked) {   // everything the forked kernels read is complete at this point of st
            HIP_TRY(s, hipEventRecord(s->ev_fork, st));
            for (int i = 0; i < n_huge_streams; i++) HIP_TRY(s, hipStreamWaitEvent(hstream[i], s->ev_fork, 0));
            forked = need_join = true;
          }
          fst = on_second ? s->stream2 : st;
        }
        const int hsel = (stream && h_nlarge[b] > 0 && n_huge_streams > 1) ? (huge_next++ % n_huge_streams) : 0;
        if (s->profiling) {
          fin_ev.emplace_back(); (void)hipEventCreate(&fin_ev.back().first); (void)hipEventCreate(&fin_ev.back().second);
          (void)hipEventRecord(fin_ev.back().first, fst);
          fin_keys += h_counts[b];
          fin_in_bytes += h_counts[b] * (narrow[b] ? 4u : (file_k96[b] ? 12u : (uint64_t)kbytes));
          fin_narrow = fin_narrow || narrow[b];
        }
        HIP_TRY(s, mgc::launch_finish_file(seg, kw, d_substart + sbase[b], ngf(b), low, h_nlarge[b],
                                           d_large + gbase[b], cnt_ptr[b], d_group + gbase[b], stream, (void *)halt[hsel], hstream[hsel],
                                           // the list pays off only when a good part of the 2^t grid is empty
                                           (4 * h_nzcount[b] < 3 * ngf(b) && !finish_nolist) ? d_nz + gbase[b] : nullptr,
                                           d_nzcount(b), fst, narrow[b] != 0, tr_a[b], tr_b[b], h_maxsub[b], h_counts[b],
                                           // (the distinct-sized count's retry list: behind the file's oversized list -- a sub-bucket is on one of them at most)
                                           fstream[b] ? d_large + gbase[b] + h_nlarge[b] : d_nz + gbase[b], d_retrycnt + b, file_k96[b] != 0,
                                           sw.hash_multi, sw.hash_dbg, fstream[b] ? mgc::finish_stream_capacity() : 0, hws[hsel], hws_bytes, max_bucket, d_err));
        if (s->profiling) (void)hipEventRecord(fin_ev.back().second, fst);
      } else {
        // a sub-bucket does not fit in LDS (heavily repeated k-mers): finish this file the long way
        fallback[b] = true;
        if (need_join) { const int jr = huge_sync_all(); if (jr != MGC_OK) return jr; }   // the sort below uses Y, the streaming kernels' second buffer
        forked = false;                                            // ... and the next streaming kernel must wait for that sort
        if (low || unordered) {
          // LSD order: the low bits cannot be sorted after the top bits, so the whole key is redone
          mgc::SortPlan lp;
          mgc::make_sort_plan(0, rem_bits, &lp);
          int in_alt = 0;
          HIP_TRY(s, mgc::launch_radix_sort(seg, (void *)Y, h_counts[b], kw, lp, sort_ws, sort_ws_bytes - 256, d_err, &in_alt, st, nullptr));
          if (in_alt) HIP_TRY(s, hipMemcpyAsync(seg, Y, kbytes * h_counts[b], hipMemcpyDeviceToDevice, st));
        }
        HIP_TRY(s, mgc::launch_rle_count(seg, h_counts[b], kw, rle_ws, st));
        HIP_TRY(s, mgc::rle_read_total(rle_ws, &h_fallback_distinct[b], st));
        HIP_TRY(s, hipMemsetAsync(d_group + gbase[b], 0, sizeof(uint64_t) * (gbase[b + 1] - gbase[b]), st));
        HIP_TRY(s, hipMemcpyAsync(d_group + gbase[b], &h_fallback_distinct[b], sizeof(uint64_t), hipMemcpyHostToDevice, st));
        HIP_TRY(s, hipStreamSynchronize(st));
      }
      return MGC_OK;
    };

    if (probe >= 0) {
      // the probe file: passes, statistics, count -- then its distinct / instances ratio chooses the others' plan
      const uint32_t pb = (uint32_t)probe;
      { const int rc = group_file(pb); if (rc != MGC_OK) return rc; }
      { const int rc = stats_file(pb); if (rc != MGC_OK) return rc; }
      HIP_TRY(s, hipMemcpyAsync(s->h_stats + 3 * (size_t)pb, d_stats + 3 * (size_t)pb, sizeof(uint64_t) * 3, hipMemcpyDeviceToHost, st));
      HIP_TRY(s, hipStreamSynchronize(st));
      h_maxsub[pb] = s->h_stats[3 * (size_t)pb]; h_nlarge[pb] = s->h_stats[3 * (size_t)pb + 1]; h_nzcount[pb] = s->h_stats[3 * (size_t)pb + 2];
      { const int rc = finish_file(pb); if (rc != MGC_OK) return rc; }
      if (need_join) { const int jr = huge_join_all(); if (jr != MGC_OK) return jr; }
      forked = false;                                        // (the second stream has to be ordered behind the other files' passes again)
      uint64_t h_pd = 0;
      HIP_TRY(s, mgc::launch_sum_u64(d_group + gbase[pb], gbase[pb + 1] - gbase[pb], d_group + ng_total, st));
      HIP_TRY(s, hipMemcpyAsync(&h_pd, d_group + ng_total, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
      HIP_TRY(s, hipStreamSynchronize(st));
      const double ratio = (double)h_pd / (double)h_counts[pb];
      s->prof.probe_ratio = ratio;
      if (sw.finish_trace) fprintf(stderr, "[finish] probe file %u: %llu distinct of %llu k-mers (%.3f): the other files take the %s plan\n", pb,
                                   (unsigned long long)h_pd, (unsigned long long)h_counts[pb], ratio, ratio <= 0.30 ? "distinct-sized" : "finer");
      if (ratio <= 0.30) for (uint32_t b = 0; b < nb; b++) {
        if (b != pb && top_str[b]) take_stream_plan(b);
        else if (b != pb && hpc_cand[b] && !fstream[b]) {      // (`compress`: the other count kernel, on 3^9 sub-buckets where the bucket's size asks for them)
          fstream[b] = 1; s->prof.stream_files++;
          if (hpc_mixed[b]) take_mixed_plan(b);
        }
      }
      if (d_fine && nb <= 256 && d_nhdrs) { const int prc = prepare_headers(-1, probe); if (prc != MGC_OK) return prc; }
    }
    bool stats_back = false;
    for (uint32_t b = 0; b < nb; b++) {
      { const int rc = group_upto(nb); if (rc != MGC_OK) return rc; }
      if (!stats_back) {
        HIP_TRY(s, hipStreamSynchronize(st)); stats_back = true;
        const int hrc = huge_setup(); if (hrc != MGC_OK) return hrc;
      }
      if ((int)b == probe) continue;
      h_maxsub[b] = s->h_stats[3 * (size_t)b]; h_nlarge[b] = s->h_stats[3 * (size_t)b + 1]; h_nzcount[b] = s->h_stats[3 * (size_t)b + 2];
      const int rc = finish_file(b);
      if (rc != MGC_OK) return rc;
    }

    if (need_join) { const int jr = huge_join_all(); if (jr != MGC_OK) return jr; }
    if (sw.finish_trace && hws[0]) {                          // what the LAST sliced file on every streaming stream did (diagnostics)
      HIP_TRY(s, hipStreamSynchronize(st));
      for (int i = 0; i < n_huge_streams; i++) {
        if (!hws[i]) continue;
        std::vector<uint32_t> w(hws_bytes / 4);
        HIP_TRY(s, hipMemcpy(w.data(), hws[i], hws_bytes / 4 * 4, hipMemcpyDeviceToHost));
        const uint32_t max_gig = (uint32_t)(max_bucket / 65536 + 2);
        uint32_t dense = 0;
        for (uint32_t q = 0; q < w[0] && q < max_gig; q++) dense += w[64 + 2 * (size_t)max_gig + q] ? 1u : 0u;
        fprintf(stderr, "[finish] sliced count, stream %d: the last file had %u sub-buckets cut into %u slices; %u of them dense (counted by ranges)\n", i, w[0], w[1], dense);
      }
    }

    // the sub-buckets hash_count_stream_kernel could not hold (more distinct suffixes than its table: low coverage, D ~ N): their
    // numbers are on the device -- one small copy brings the counts back, the files that have any get the retry launch
    {
      bool any_stream = false;
      for (uint32_t b = 0; b < nb; b++) any_stream = any_stream || (fstream[b] && !fallback[b] && h_counts[b]);
      if (any_stream) {
        std::vector<uint64_t> h_retry(nb, 0);
        HIP_TRY(s, hipMemcpyAsync(h_retry.data(), d_retrycnt, sizeof(uint64_t) * nb, hipMemcpyDeviceToHost, st));
        HIP_TRY(s, hipStreamSynchronize(st));
        for (uint32_t b = 0; b < nb; b++) {
          if (!fstream[b] || fallback[b] || h_retry[b] == 0) continue;
          s->prof.stream_retries += h_retry[b];
          HIP_TRY(s, mgc::launch_finish_retry(X + kbytes * h_starts[b], d_substart + sbase[b], ngf(b), rem_bits - top_bits[b],
                                              cnt_ptr[b], d_group + gbase[b], tr_a[b], tr_b[b], d_large + gbase[b] + h_nlarge[b], d_retrycnt + b,
                                              h_retry[b], mgc::finish_stream_capacity(), st, narrow[b] != 0, (void *)huge_alt));
        }
      }
    }

    // ---- E/F. offsets of every sub-bucket in the packed result ----
    hipEvent_t ev_pack[2] = {nullptr, nullptr};
    if (s->profiling) { (void)hipEventCreate(&ev_pack[0]); (void)hipEventCreate(&ev_pack[1]); (void)hipEventRecord(ev_pack[0], st); }
    HIP_TRY(s, mgc::launch_finish_scan(d_group, ng_total, s->buf[mgc_session::B_GSCAN].p, st));
    HIP_TRY(s, hipMemcpyAsync(&nd, d_group + ng_total, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(s, hipStreamSynchronize(st));
    if (ng_total == 0) nd = 0;
    s->n_distinct = nd;
    if (s->ext_out_keys && s->ext_out_counts && nd <= s->ext_out_cap) {
      // (mgc_count_buckets_into: the packing kernels write the caller's pre-sized result -- no copy out of the arena afterwards)
      s->d_unique = s->ext_out_keys;
      s->d_counts = s->ext_out_counts;
    } else {
      HIP_TRY(s, s->ensure(mgc_session::B_UNIQUE, kbytes * nd));
      HIP_TRY(s, s->ensure(mgc_session::B_COUNTS, sizeof(uint32_t) * nd));
      s->d_unique = s->buf[mgc_session::B_UNIQUE].p;
      s->d_counts = reinterpret_cast<uint32_t *>(s->buf[mgc_session::B_COUNTS].p);
    }

    // ---- G/H. pack ----
    for (uint32_t b = 0; b < nb; b++) {
      if (h_counts[b] == 0) continue;
      void *seg = X + kbytes * h_starts[b];
      // a sparse sub-bucket grid (`compress`: 59049 of 2^20): only the non-empty ones are visited
      const uint32_t *nzl = (4 * h_nzcount[b] < 3 * ngf(b) && !finish_nolist) ? d_nz + gbase[b] : nullptr;
      if (narrow[b]) {
        HIP_TRY(s, mgc::launch_compact_groups_narrow(seg, cnt_ptr[b], d_substart + sbase[b], d_group + gbase[b],
                                                     ngf(b), (uint64_t)b << rem_bits, rem_bits - top_bits[b],
                                                     s->d_unique, s->d_counts, st, tr_a[b], tr_b[b], nzl, h_nzcount[b]));
      } else if (!fallback[b] && file_k96[b]) {
        const unsigned __int128 fb = (unsigned __int128)b << rem_bits;
        HIP_TRY(s, mgc::launch_compact_groups_k96(seg, cnt_ptr[b], d_substart + sbase[b], d_group + gbase[b], ngf(b),
                                                  (uint64_t)fb, (uint64_t)(fb >> 64), s->d_unique, s->d_counts, st, tr_a[b], tr_b[b], nzl, h_nzcount[b]));
      } else if (!fallback[b]) {
        HIP_TRY(s, mgc::launch_compact_groups(seg, kw, cnt_ptr[b], d_substart + sbase[b], d_group + gbase[b],
                                              ngf(b), s->d_unique, s->d_counts, st, tr_a[b], tr_b[b], nzl, h_nzcount[b]));
      } else {
        HIP_TRY(s, mgc::launch_rle_count(seg, h_counts[b], kw, rle_ws, st));
        HIP_TRY(s, mgc::launch_rle_emit(seg, h_counts[b], kw, rle_ws, s->d_unique, s->d_counts, st, d_group + gbase[b]));
      }
    }
    if (s->profiling) (void)hipEventRecord(ev_pack[1], st);
    tm.end(MGC_STAGE_RLE);
    s->prof.stage_launches[MGC_STAGE_RLE] = 4 * nb;
    if (s->profiling) {
      HIP_TRY(s, hipStreamSynchronize(st));
      { float pms = 0; if (hipEventElapsedTime(&pms, ev_pack[0], ev_pack[1]) == hipSuccess) s->prof.pack_ms = pms; }
      (void)hipEventDestroy(ev_pack[0]); (void)hipEventDestroy(ev_pack[1]);
      for (auto &pe : fin_ev) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pe.first, pe.second) == hipSuccess) { s->prof.finish_ms += ms; s->prof.finish_launches++; }
        (void)hipEventDestroy(pe.first); (void)hipEventDestroy(pe.second);
      }
      s->prof.finish_keys = fin_keys;
      s->prof.finish_bytes = fin_in_bytes + nd * (fin_narrow ? 8u : (uint64_t)kbytes + 4u);
    }
  }

  // ---- block offsets ----
  HIP_TRY(s, s->ensure(mgc_session::B_BLOCKS, sizeof(uint64_t) * (c.n_prefix + 1)));
  s->d_block_start = reinterpret_cast<uint64_t *>(s->buf[mgc_session::B_BLOCKS].p);
  tm.begin(MGC_STAGE_BLOCKS);
  HIP_TRY(s, mgc::launch_block_offsets(s->d_unique, nd, kw, c.w_data, c.n_prefix, s->d_block_start, st));
  tm.end(MGC_STAGE_BLOCKS);
  s->prof.stage_launches[MGC_STAGE_BLOCKS] = 1;

  uint32_t h_err[3] = {0, 0, 0};
  HIP_TRY(s, hipMemcpyAsync(h_err, d_err, sizeof(h_err), hipMemcpyDeviceToHost, st));
  if (s->profiling) (void)hipEventRecord(ev_all[1], st);
  HIP_TRY(s, hipStreamSynchronize(st));
  if (h_err[0]) { set_err(&s->err, "radix sort look-back timed out"); return MGC_ETIMEOUT; }

  if (s->profiling) {
    float ms = 0;
    for (int i = 0; i < MGC_NUM_STAGES; i++)
      if (tm.used[i] && hipEventElapsedTime(&ms, tm.ev[i][0], tm.ev[i][1]) == hipSuccess) s->prof.stage_ms[i] = ms;
    if (hipEventElapsedTime(&ms, ev_all[0], ev_all[1]) == hipSuccess) s->prof.total_ms = ms;
    s->prof.stage_launches[MGC_STAGE_SORT] = sort_launch_groups * (plan.num_passes + 2);
    for (uint32_t b = 0; b < nb; b++) {
      if (h_counts[b] == 0) continue;
      for (uint32_t p = 0; p < file_passes[b]; p++) {
        hipEvent_t *pe = &pass_ev[(size_t)b * ev_per_file + 2 * p];
        if (hipEventElapsedTime(&ms, pe[0], pe[1]) == hipSuccess) {
          s->prof.sort_pass_ms_total += ms;
          s->prof.sort_pass_launches++;
          s->prof.sort_pass_keys += h_counts[b];
          const int pi = p ? 1 : 0;
          s->prof.pass_ms[pi] += ms;
          s->prof.pass_launches[pi]++;
          s->prof.pass_keys[pi] += h_counts[b];
          s->prof.pass_bytes[pi] += h_counts[b] * ((size_t)b < narrowed.size() && narrowed[b] ? (p ? 8u : (soa_hi_mask ? 9u : 12u)) : (k96_passes[b] ? 24u : 2u * kbytes));
        }
      }
    }
    for (auto &e : pass_ev) (void)hipEventDestroy(e);
    (void)hipEventDestroy(ev_all[0]); (void)hipEventDestroy(ev_all[1]);
    // (an elapsed-time query on an event pair a small file never recorded fails, harmlessly -- but the runtime keeps the error for
    // the thread's next hipGetLastError(): the CLI's -V on a batched count failed in the run store's first launch that way)
    (void)hipGetLastError();
  }
  return MGC_OK;                                             // the caller marks the session counted (a batch is not the result yet)
}

static int copy_device_result(mgc_session *s, uint64_t *keys_lo, uint64_t *keys_hi, uint32_t *counts, uint64_t *block_start) {
  const uint64_t nd = s->n_distinct;
  if (nd && (keys_lo || keys_hi)) {
    if (s->key_words == 1) {
      if (keys_lo) HIP_TRY(s, hipMemcpy(keys_lo, s->d_unique, sizeof(uint64_t) * nd, hipMemcpyDeviceToHost));
      if (keys_hi) memset(keys_hi, 0, sizeof(uint64_t) * nd);
    } else {
      std::vector<uint64_t> tmp(2 * nd);                   // {lo,hi} pairs
      HIP_TRY(s, hipMemcpy(tmp.data(), s->d_unique, sizeof(uint64_t) * 2 * nd, hipMemcpyDeviceToHost));
      for (uint64_t i = 0; i < nd; i++) {
        if (keys_lo) keys_lo[i] = tmp[2 * i];
        if (keys_hi) keys_hi[i] = tmp[2 * i + 1];
      }
    }
  }
  if (counts && nd) HIP_TRY(s, hipMemcpy(counts, s->d_counts, sizeof(uint32_t) * nd, hipMemcpyDeviceToHost));
  if (block_start)  HIP_TRY(s, hipMemcpy(block_start, s->d_block_start, sizeof(uint64_t) * (s->cfg.n_prefix + 1), hipMemcpyDeviceToHost));
  return MGC_OK;
}

// The (k-mer, count) result count_device just left in the arena is parked as a RUN (mgc_runs.cpp): in HBM while the
// store's device budget lasts, in pinned host DRAM otherwise.  Nothing is merged here -- the runs are merged once, when
// the count ends (the analogue of writeBatch's iterations and merylBlockWriter::finish() merging them,
// merylOp-countThreads.C:323-379,461-464).
static int park_batch_result(mgc_session *s) {
  if (!s->runs) {
    // Runs may take what the count itself leaves free: the arena (grow-only, sized by the batches) and the staging buffers are
    // allocated by now, and the final merge frees the arena before it allocates anything.
    uint64_t budget = s->result_budget;
    if (budget == 0) {
      size_t free_b = 0, total_b = 0;
      budget = (hipMemGetInfo(&free_b, &total_b) == hipSuccess) ? (uint64_t)((double)free_b * 0.6) : 0;
    }
    s->runs = new mgc_runs(s->cfg.k, s->cfg.w_prefix, s->device, budget, 0);
  }
  const int rc = s->runs->add(s->d_unique, s->d_counts, s->n_distinct, s->stream);
  if (rc != MGC_OK) s->err = s->runs->err;
  s->have_r = true;
  s->free_result();
  return rc;
}

// one batch: stage[which][0, n) -> count -> park.  Runs on the worker thread while the caller stages the next batch.
static int count_staged_batch(mgc_session *s, int which, uint64_t n) {
  s->d_bases = stage_ptr(s, which);
  s->n_bases = n;
  int rc = count_device(s);
  if (rc != MGC_OK) return rc;
  s->total_bases += n;
  s->total_instances += s->n_instances;
  for (int f = 0; f < MGC_NUM_FILES; f++) s->total_file_instances[f] += s->file_instances[f];
  s->n_batches++;
  return park_batch_result(s);
}

// All batches are counted.  If every run is still in HBM and their merge fits beside them, the runs collapse into ONE
// device-resident result -- then the session looks like a single pass's (every result call works).  Otherwise the result
// is OUT OF CORE: it exists only as the runs, and is delivered by streaming (mgc_write_database, mgc_finish*), which merge
// chunk by chunk; the calls that hand out the whole result refuse (MGC_ESTATE).
static int finalize_from_runs(mgc_session *s) {
  hipStream_t st = s->stream;
  s->n_instances = s->total_instances;
  s->n_bases = s->total_bases;
  memcpy(s->file_instances, s->total_file_instances, sizeof(s->file_instances));
  HIP_TRY(s, hipStreamSynchronize(s->st_in));
  s->free_arena();                                            // the count's buffers: the merge gets their room
  s->state_ready = false;
  mgc_runs *r = s->runs;
  const size_t esz = sizeof(uint64_t) * s->key_words + sizeof(uint32_t);
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  const bool fits = r->all_on_device() && (uint64_t)free_b > r->entries() * esz + (64ull << 20);
  s->prof.n_batches = s->n_batches;
  if (fits) {
    const void *k = nullptr; const uint32_t *c = nullptr; uint64_t n = 0;
    int rc = r->collapse(&k, &c, &n);
    if (rc == MGC_ENOMEM) {                                   // (hipMemGetInfo is an estimate: fragmentation) the store is consistent: out of core
      s->ooc = true;
      s->n_distinct = 0;
      s->merge_ms = r->prof.merge_ms;
      s->prof.merge_ms = s->merge_ms;
      return MGC_OK;
    }
    if (rc != MGC_OK) { s->err = r->err; return rc; }
    s->n_distinct = n;
    s->d_unique = const_cast<void *>(k);
    s->d_counts = const_cast<uint32_t *>(c);
    HIP_TRY(s, s->ensure(mgc_session::B_BLOCKS, sizeof(uint64_t) * (s->cfg.n_prefix + 1)));
    s->d_block_start = reinterpret_cast<uint64_t *>(s->buf[mgc_session::B_BLOCKS].p);
    HIP_TRY(s, mgc::launch_block_offsets(s->d_unique, s->n_distinct, s->key_words, s->cfg.w_data, s->cfg.n_prefix, s->d_block_start, st));
    HIP_TRY(s, hipStreamSynchronize(st));
  } else {
    s->ooc = true;
    s->n_distinct = 0;                                        // known once the runs have been merged (delivery)
  }
  s->merge_ms = r->prof.merge_ms;
  s->prof.merge_ms = s->merge_ms;
  return MGC_OK;
}

extern "C" int mgc_count(mgc_session *s) {
  if (!s) return MGC_EINVAL;
  if (s->borrowed) {                                         // the caller's device buffer: every call counts it again
    const int rc = count_device(s);
    if (rc == MGC_OK) s->counted = true;
    return rc;
  }
  if (s->counted) return MGC_OK;                             // nothing can have been pushed since: the result stands
  if (s->text_open) { set_err(&s->err, "mgc_count: a text file is still open (mgc_end_text)"); return MGC_ESTATE; }
  int rc = MGC_OK;
  if (!s->state_ready) { rc = input_setup(s); if (rc != MGC_OK) return rc; }     // nothing was pushed: an empty stream
  rc = flush_pinned(s);
  if (rc == MGC_OK) rc = resolve_length(s);
  const int wrc = join_worker(s);
  if (rc == MGC_OK) rc = wrc;
  if (rc != MGC_OK) return rc;
  if (!s->have_r) {                                          // everything fits in one pass
    s->d_bases = stage_ptr(s, s->fill);
    s->n_bases = s->fill_len;
    rc = count_device(s);
    s->prof.n_batches = 1;
  } else {
    if (s->fill_len) rc = count_staged_batch(s, s->fill, s->fill_len);
    if (rc == MGC_OK) rc = finalize_from_runs(s);
  }
  if (rc == MGC_OK) s->counted = true;
  s->free_garbage();
  if (getenv("MGC_IO_TRACE") && s->input_seen)
    fprintf(stderr, "[io] host pushes: memcpy into pinned %.3f s, flush (grow + async upload + wait for the other chunk) %.3f s, batch cuts %.3f s "
                    "(of which waiting for the worker %.3f s), staging-buffer growth %.3f s; %u batches, device merges %.1f ms; "
                    "arena hipMalloc/hipFree %.3f s for %.1f GB\n",
            s->tr_memcpy, s->tr_flush, s->tr_cut, s->tr_join, s->tr_grow, s->n_batches, s->merge_ms, s->tr_alloc, s->tr_alloc_bytes / 1e9);
  return rc;
}

extern "C" int mgc_staged_bases(mgc_session *s, const uint8_t **d_bases, uint64_t *n_bases) {
  if (!s || !d_bases || !n_bases) return MGC_EINVAL;
  if (s->borrowed) { *d_bases = s->d_bases; *n_bases = s->n_bases; return MGC_OK; }
  if (s->text_open) { set_err(&s->err, "mgc_staged_bases: a text file is still open (mgc_end_text)"); return MGC_ESTATE; }
  if (s->counted || s->have_r || s->n_batches) { set_err(&s->err, "mgc_staged_bases: part of the input has already been counted"); return MGC_ESTATE; }
  int rc = MGC_OK;
  if (!s->state_ready) { rc = input_setup(s); if (rc != MGC_OK) return rc; }
  rc = flush_pinned(s);
  if (rc == MGC_OK) rc = resolve_length(s);
  const int wrc = join_worker(s);
  if (rc == MGC_OK) rc = wrc;
  if (rc != MGC_OK) return rc;
  HIP_TRY(s, hipStreamSynchronize(s->st_in));
  *d_bases = stage_ptr(s, s->fill);
  *n_bases = s->fill_len;
  return MGC_OK;
}

extern "C" int mgc_count_buckets(mgc_session *s, void *d_keys, uint32_t bucket_bits, const uint64_t *bucket_counts) {
  if (!s || !bucket_counts || bucket_bits < MGC_NUM_FILES_BITS || bucket_bits > MGC_MAX_BUCKET_BITS || bucket_bits > 2 * s->cfg.k)
    return MGC_EINVAL;
  uint64_t n = 0;
  for (uint32_t b = 0; b < (1u << bucket_bits); b++) n += bucket_counts[b];
  if (n && !d_keys) return MGC_EINVAL;
  if (s->input_seen || s->borrowed) {
    set_err(&s->err, "mgc_count_buckets: the session already holds pushed bases");
    return MGC_ESTATE;
  }
  const int rc = count_device(s, d_keys, bucket_counts, bucket_bits);
  if (rc == MGC_OK) s->counted = true;
  return rc;
}

// ... with the packed result written straight to the caller's buffers when it fits (capacity in k-mers): the waves of a sharded
// count land in ONE pre-sized result instead of being copied out of the session and concatenated.  *n_distinct > capacity: the
// result stayed in the session (mgc_copy_result_device / mgc_get_result_device as usual).
extern "C" int mgc_count_buckets_into(mgc_session *s, void *d_keys, uint32_t bucket_bits, const uint64_t *bucket_counts,
                                      void *d_out_keys, uint32_t *d_out_counts, uint64_t capacity, uint64_t *n_distinct,
                                      const uint64_t *d_fine_hist) {
  if (!s || !n_distinct || (capacity && (!d_out_keys || !d_out_counts))) return MGC_EINVAL;
  s->ext_out_keys = d_out_keys; s->ext_out_counts = d_out_counts; s->ext_out_cap = capacity;
  s->ext_fine = d_fine_hist;
  const int rc = mgc_count_buckets(s, d_keys, bucket_bits, bucket_counts);
  s->ext_out_keys = nullptr; s->ext_out_counts = nullptr; s->ext_out_cap = 0;
  s->ext_fine = nullptr;
  if (rc == MGC_OK) *n_distinct = s->n_distinct;
  return rc;
}

extern "C" int mgc_count_partitioned(mgc_session *s, void *d_keys, const uint64_t *file_counts, void *reserved) {
  (void)reserved;
  return mgc_count_buckets(s, d_keys, MGC_NUM_FILES_BITS, file_counts);
}

extern "C" int mgc_get_result_info(const mgc_session *s, mgc_result_info *info) {
  if (!s || !info) return MGC_EINVAL;
  if (!s->counted) return MGC_ESTATE;
  info->n_bases = s->n_bases;
  info->n_instances = s->n_instances;
  info->n_distinct = s->n_distinct;
  info->w_prefix = s->cfg.w_prefix;
  info->w_data = s->cfg.w_data;
  info->n_prefix = s->cfg.n_prefix;
  memcpy(info->file_instances, s->file_instances, sizeof(info->file_instances));
  return MGC_OK;
}

static int refuse_ooc(mgc_session *s, const char *what) {
  set_err(&s->err, "%s: the result is out of core (larger than the device holds): stream it with mgc_write_database / mgc_finish", what);
  return MGC_ESTATE;
}

extern "C" int mgc_result_out_of_core(const mgc_session *s) { return (s && s->counted && s->ooc) ? 1 : 0; }

extern "C" int mgc_set_result_budget(mgc_session *s, uint64_t device_bytes) {
  if (!s) return MGC_EINVAL;
  s->result_budget = device_bytes;
  return MGC_OK;
}

extern "C" int mgc_get_runs_profile(const mgc_session *s, mgc_runs_profile *p) {
  if (!s || !p) return MGC_EINVAL;
  memset(p, 0, sizeof(*p));
  if (s->runs) *p = s->runs->prof;
  return MGC_OK;
}

extern "C" int mgc_copy_result_device(mgc_session *s, void *d_keys_out, uint32_t *d_counts_out) {
  if (!s) return MGC_EINVAL;
  if (!s->counted) return MGC_ESTATE;
  if (s->ooc) return refuse_ooc(s, "mgc_copy_result_device");
  const size_t kbytes = sizeof(uint64_t) * s->key_words;
  if (s->n_distinct) {
    if (d_keys_out) HIP_TRY(s, hipMemcpyAsync(d_keys_out, s->d_unique, kbytes * s->n_distinct, hipMemcpyDeviceToDevice, s->stream));
    if (d_counts_out) HIP_TRY(s, hipMemcpyAsync(d_counts_out, s->d_counts, sizeof(uint32_t) * s->n_distinct, hipMemcpyDeviceToDevice, s->stream));
  }
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  return MGC_OK;
}

extern "C" int mgc_get_result_device(const mgc_session *s, const void **d_unique, const uint32_t **d_counts,
                                     const uint64_t **d_block_start, uint32_t *key_words) {
  if (!s) return MGC_EINVAL;
  if (!s->counted) return MGC_ESTATE;
  if (s->ooc) return refuse_ooc(const_cast<mgc_session *>(s), "mgc_get_result_device");
  if (d_unique) *d_unique = s->d_unique;
  if (d_counts) *d_counts = s->d_counts;
  if (d_block_start) *d_block_start = s->d_block_start;
  if (key_words) *key_words = s->key_words;
  return MGC_OK;
}

extern "C" int mgc_copy_result(const mgc_session *cs, uint64_t *keys_lo, uint64_t *keys_hi, uint32_t *counts,
                               uint64_t *block_start) {
  mgc_session *s = const_cast<mgc_session *>(cs);
  if (!s) return MGC_EINVAL;
  if (!s->counted) return MGC_ESTATE;
  if (s->ooc) return refuse_ooc(s, "mgc_copy_result");
  HIP_TRY(s, hipSetDevice(s->device));
  return copy_device_result(s, keys_lo, keys_hi, counts, block_start);
}
