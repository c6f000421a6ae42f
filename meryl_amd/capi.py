"""ctypes binding of include/meryl_gpu_count.h.

Loads meryl_amd/libmeryl_gpu_count.so and fails loudly when it is missing or
an expected symbol is absent -- there is no Python/CPU fallback for any compute
entry point.  Device pointers are plain integers (e.g. torch_tensor.data_ptr()).
"""
import ctypes
import os

from . import build as _build

MGC_OK, MGC_EINVAL, MGC_ENOMEM, MGC_EHIP, MGC_ESTATE, MGC_EUNSUPPORTED, MGC_ETIMEOUT = 0, -1, -2, -3, -4, -5, -6
MODE_CANONICAL, MODE_FORWARD, MODE_REVERSE = 0, 1, 2
NUM_FILES = 64
NUM_STAGES = 5
STAGE_NAMES = ("histogram", "partition", "sort", "rle", "blocks")

# every extern "C" symbol include/meryl_gpu_count.h declares
SYMBOLS = (
    "mgc_configure_counting", "mgc_format_configured_line",
    "mgc_dev_partition_workspace_bytes", "mgc_dev_kmer_histogram", "mgc_dev_kmer_partition",
    "mgc_dev_sort_workspace_bytes", "mgc_dev_radix_sort", "mgc_dev_radix_group", "mgc_dev_kmer_histogram_fine",
    "mgc_dev_rle_workspace_bytes", "mgc_dev_rle_count", "mgc_dev_rle_emit", "mgc_dev_block_offsets",
    "mgc_open", "mgc_close", "mgc_last_error", "mgc_push_bases", "mgc_push_bases_device", "mgc_staged_bases", "mgc_reserve_text", "mgc_begin_text", "mgc_push_text", "mgc_end_text", "mgc_push_text_file", "mgc_push_text_file_range", "mgc_text_record_start", "mgc_push_text_bgzf_file", "mgc_is_bgzf_file", "mgc_count", "mgc_count_partitioned", "mgc_count_buckets", "mgc_count_buckets_into", "mgc_copy_result_device",
    "mgc_get_result_info", "mgc_get_result_device", "mgc_copy_result", "mgc_finish", "mgc_finish_labelled",
    "mgc_set_profiling", "mgc_get_profile", "mgc_dev_synth_reads", "mgc_dev_synth_reads_ex", "mgc_version",
    "mgc_dev_merge_workspace_bytes", "mgc_dev_merge_count", "mgc_dev_merge_count_values", "mgc_dev_merge_emit",
    "mgc_dev_select_workspace_bytes", "mgc_dev_select_count", "mgc_dev_select_emit",
    "mgc_dev_homopoly_workspace_bytes", "mgc_dev_homopoly_compress", "mgc_set_batch_bases", "mgc_prepare", "mgc_set_result_budget", "mgc_result_out_of_core",
    # include/meryl_db.h
    "mdb_writer_open", "mdb_writer_open_ex", "mdb_merge_parts", "mdb_writer_add_block", "mdb_writer_add_block_labelled",
    "mdb_writer_add_encoded", "mdb_writer_reserve_encoded", "mdb_writer_write_at", "mdb_writer_add_histogram", "mdb_writer_close", "mdb_writer_discard", "mdb_last_error",
    "mdb_reader_open", "mdb_reader_info", "mdb_reader_histogram", "mdb_reader_read_file", "mdb_reader_read_file_ex",
    "mdb_reader_file_index", "mdb_reader_block_header", "mdb_reader_read_block_raw", "mdb_reader_raw_file", "mdb_reader_close",
    "mdb_free", "mgc_write_database", "mgc_write_database_profiled",
    "mgc_db_stream_open", "mgc_db_stream_write", "mgc_db_stream_sync", "mgc_db_stream_close", "mgc_db_stream_error", "mgc_db_stream_queued", "mgc_db_stream_done", "mgc_db_stream_wait_buffers",
    "mgc_runs_open", "mgc_runs_add", "mgc_runs_write", "mgc_runs_get_profile", "mgc_runs_error", "mgc_runs_close", "mgc_get_runs_profile", "mgc_db_merge", "mgc_db_filter", "mgc_count_node", "mgc_count_node_batched", "mgc_count_node_staged", "mgc_node_plan",
    # include/meryl_lookup.h
    "mgc_lookup_load", "mgc_lookup_estimate", "mgc_lookup_from_device", "mgc_lookup_free", "mgc_lookup_get_info", "mgc_lookup_error",
    "mgc_lookup_values", "mgc_lookup_stream", "mgc_lookup_existence",
    # include/meryl_seq.h
    "msr_open", "msr_read_text", "msr_close", "msr_last_error", "msr_load_bases", "msr_load_stream", "msr_format", "msr_is_compressed", "msr_guess_number_of_kmers",
)


class CountConfig(ctypes.Structure):
    _fields_ = [
        ("k", ctypes.c_uint32),
        ("mode", ctypes.c_int32),
        ("n_kmers_estimate", ctypes.c_uint64),
        ("memory_allowed", ctypes.c_uint64),
        ("threads", ctypes.c_uint32),
        ("count_suffix_length", ctypes.c_uint32),
        ("homopoly_compress", ctypes.c_uint32),
        ("page_size", ctypes.c_uint32),
        ("sizeof_count_array", ctypes.c_uint32),
        ("use_simple", ctypes.c_int32),
        ("w_prefix", ctypes.c_uint32),
        ("n_prefix", ctypes.c_uint64),
        ("w_data", ctypes.c_uint32),
        ("n_batches", ctypes.c_uint32),
        ("memory_used", ctypes.c_uint64),
        ("count_suffix", ctypes.c_char * 36),
        ("label_size", ctypes.c_uint32),
        ("reserved0", ctypes.c_uint32),
        ("label_constant", ctypes.c_uint64),
    ]


class ResultInfo(ctypes.Structure):
    _fields_ = [
        ("n_bases", ctypes.c_uint64),
        ("n_instances", ctypes.c_uint64),
        ("n_distinct", ctypes.c_uint64),
        ("w_prefix", ctypes.c_uint32),
        ("w_data", ctypes.c_uint32),
        ("n_prefix", ctypes.c_uint64),
        ("file_instances", ctypes.c_uint64 * NUM_FILES),
    ]


class Profile(ctypes.Structure):
    _fields_ = [
        ("stage_ms", ctypes.c_double * NUM_STAGES),
        ("stage_launches", ctypes.c_uint32 * NUM_STAGES),
        ("sort_pass_ms_total", ctypes.c_double),
        ("sort_pass_launches", ctypes.c_uint32),
        ("sort_pass_keys", ctypes.c_uint64),
        ("total_ms", ctypes.c_double),
        ("merge_ms", ctypes.c_double),
        ("n_batches", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
        ("pass_ms", ctypes.c_double * 2),
        ("pass_bytes", ctypes.c_uint64 * 2),
        ("pass_keys", ctypes.c_uint64 * 2),
        ("pass_launches", ctypes.c_uint32 * 2),
        ("finish_ms", ctypes.c_double),
        ("finish_bytes", ctypes.c_uint64),
        ("finish_keys", ctypes.c_uint64),
        ("finish_launches", ctypes.c_uint32),
        ("wide_msd_files", ctypes.c_uint32),
        ("stream_files", ctypes.c_uint32),
        ("k96_files", ctypes.c_uint32),
        ("k96_widened_files", ctypes.c_uint32),
        ("hpc_mixed_files", ctypes.c_uint32),
        ("stream_retries", ctypes.c_uint64),
        ("probe_ratio", ctypes.c_double),
        ("pack_ms", ctypes.c_double),
        ("hist_bytes", ctypes.c_uint64),
        ("partition_bytes", ctypes.c_uint64),
    ]


class DbInfo(ctypes.Structure):
    _fields_ = [
        ("k", ctypes.c_uint32),
        ("prefix_size", ctypes.c_uint32),
        ("suffix_size", ctypes.c_uint32),
        ("num_files_bits", ctypes.c_uint32),
        ("num_blocks_bits", ctypes.c_uint32),
        ("flags", ctypes.c_uint32),
        ("num_unique", ctypes.c_uint64),
        ("num_distinct", ctypes.c_uint64),
        ("num_total", ctypes.c_uint64),
        ("hist_len", ctypes.c_uint64),
        ("label_size", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
    ]


class NodeProfile(ctypes.Structure):
    _fields_ = [
        ("n_ranks", ctypes.c_uint32),
        ("bucket_bits", ctypes.c_uint32),
        ("n_bases", ctypes.c_uint64),
        ("n_instances", ctypes.c_uint64),
        ("n_distinct", ctypes.c_uint64),
        ("data_bytes", ctypes.c_uint64),
        ("partition_s", ctypes.c_double),
        ("exchange_count_s", ctypes.c_double),
        ("close_s", ctypes.c_double),
        ("merge_parts_s", ctypes.c_double),
        ("total_s", ctypes.c_double),
        ("n_batches", ctypes.c_uint32),
        ("n_host_runs", ctypes.c_uint32),
        ("host_run_bytes", ctypes.c_uint64),
        ("merge_runs_s", ctypes.c_double),
        ("peak_hbm_bytes", ctypes.c_uint64),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class DbWriteProfile(ctypes.Structure):
    _fields_ = [
        ("plan_ms", ctypes.c_double),
        ("encode_ms", ctypes.c_double),
        ("copy_write_s", ctypes.c_double),
        ("total_s", ctypes.c_double),
        ("data_bytes", ctypes.c_uint64),
        ("n_kmers", ctypes.c_uint64),
        ("n_blocks", ctypes.c_uint64),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class RunsProfile(ctypes.Structure):
    _fields_ = [
        ("n_runs", ctypes.c_uint32), ("n_host_runs", ctypes.c_uint32),
        ("n_entries", ctypes.c_uint64),
        ("device_bytes", ctypes.c_uint64), ("host_bytes", ctypes.c_uint64),
        ("n_merged", ctypes.c_uint64),
        ("n_chunks", ctypes.c_uint32),
        ("spill_s", ctypes.c_double),
        ("upload_s", ctypes.c_double), ("merge_ms", ctypes.c_double), ("deliver_s", ctypes.c_double),
        ("peak_hbm_bytes", ctypes.c_uint64),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class LookupInfo(ctypes.Structure):
    _fields_ = [("k", ctypes.c_uint32), ("key_words", ctypes.c_uint32), ("index_bits", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
                ("n_kmers", ctypes.c_uint64), ("n_kmers_in_db", ctypes.c_uint64), ("device_bytes", ctypes.c_uint64)]


class IndexEntry(ctypes.Structure):
    _fields_ = [("prefix", ctypes.c_uint64), ("position", ctypes.c_uint64), ("n_kmers", ctypes.c_uint64)]


class BlockHeader(ctypes.Structure):
    _fields_ = [("prefix", ctypes.c_uint64), ("n_kmers", ctypes.c_uint64), ("k_code", ctypes.c_uint32),
                ("unary_bits", ctypes.c_uint32), ("binary_bits", ctypes.c_uint32), ("c_code", ctypes.c_uint32),
                ("k1", ctypes.c_uint64), ("c1", ctypes.c_uint64), ("c2", ctypes.c_uint64)]


BLOCK_CB2 = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
                             ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                             ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint64)
BLOCK_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
                            ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                            ctypes.POINTER(ctypes.c_uint32))


OK, EINVAL, ENOMEM, EHIP, ESTATE, EUNSUPPORTED, ETIMEOUT, EFORMAT = 0, -1, -2, -3, -4, -5, -6, -7


class MgcError(RuntimeError):
    def __init__(self, rc, what, detail=""):
        self.rc = rc
        self.code = rc
        super().__init__("%s failed rc=%d %s" % (what, rc, detail))


_lib = None


def library_path():
    return _build.LIB


def _load_hip_runtime():
    """The native library is linked without a HIP runtime of its own (-no-hip-rt)
    and binds to the one already in the process.  Under Python that must be the
    runtime torch uses (its bundled libamdhip64.so), so that torch's device
    pointers, streams and ordering are valid inside the library; without torch it
    is /opt/rocm's."""
    cands = []
    try:
        import torch  # noqa: F401  (loads its bundled ROCm runtime)
        cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except Exception:
        pass
    cands += [os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "libamdhip64.so"), "libamdhip64.so"]
    for c in cands:
        if os.path.sep in c and not os.path.exists(c):
            continue
        try:
            ctypes.CDLL(c, mode=ctypes.RTLD_GLOBAL)
            return c
        except OSError:
            continue
    raise RuntimeError("no HIP runtime (libamdhip64.so) could be loaded; the count path has no CPU fallback")


def lib():
    """The loaded C-ABI library.  Raises if it is not built -- by design."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            "native library %s is missing: run `python -m meryl_amd.build` "
            "(there is no CPU fallback for the count path)" % path)
    _load_hip_runtime()
    L = ctypes.CDLL(path)
    missing = [s for s in SYMBOLS if not hasattr(L, s)]
    if missing:
        raise RuntimeError("native library %s lacks symbols: %s" % (path, ", ".join(missing)))
    vp, u64, u32, i32, sz = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_size_t
    P = ctypes.POINTER

    def sig(name, restype, *argtypes):
        f = getattr(L, name)
        f.restype = restype
        f.argtypes = list(argtypes)

    sig("mgc_version", u32)
    sig("mgc_configure_counting", i32, P(CountConfig))
    sig("mgc_format_configured_line", i32, P(CountConfig), ctypes.c_char_p, sz)
    sig("mgc_dev_partition_workspace_bytes", sz, u32)
    sig("mgc_dev_kmer_histogram", i32, vp, u64, u32, i32, u32, vp, vp, sz, vp)
    sig("mgc_dev_kmer_partition", i32, vp, u64, u32, i32, u32, vp, vp, vp, sz, vp)
    sig("mgc_dev_sort_workspace_bytes", sz, u64)
    sig("mgc_dev_radix_sort", i32, vp, vp, u64, u32, u32, u32, vp, sz, P(i32), vp)
    sig("mgc_dev_radix_group", i32, vp, vp, u64, u32, u32, u32, vp, sz, P(i32), vp)
    sig("mgc_dev_rle_workspace_bytes", sz, u64)
    sig("mgc_dev_rle_count", i32, vp, u64, u32, vp, sz, P(u64), vp)
    sig("mgc_dev_rle_emit", i32, vp, u64, u32, vp, sz, vp, vp, vp)
    sig("mgc_dev_block_offsets", i32, vp, u64, u32, u32, u64, vp, vp)
    sig("mgc_dev_merge_workspace_bytes", sz, u64, u64)
    sig("mgc_dev_merge_count", i32, vp, u64, vp, u64, u32, i32, vp, sz, P(u64), vp)
    sig("mgc_dev_merge_emit", i32, vp, vp, u64, vp, vp, u64, u32, i32, vp, sz, vp, vp, vp)
    sig("mgc_dev_merge_count_values", i32, vp, vp, u64, vp, vp, u64, u32, i32, vp, sz, P(u64), vp)
    sig("mgc_dev_select_workspace_bytes", sz, u64)
    sig("mgc_dev_select_count", i32, vp, vp, u64, u32, i32, u64, vp, sz, P(u64), vp)
    sig("mgc_dev_select_emit", i32, vp, vp, u64, u32, i32, u64, vp, sz, vp, vp, vp)
    sig("mgc_db_filter", i32, ctypes.c_char_p, i32, u64, ctypes.c_char_p, i32, i32)
    sig("mgc_dev_homopoly_workspace_bytes", sz, u64)
    sig("mgc_dev_homopoly_compress", i32, vp, u64, vp, P(u64), vp, sz, vp)
    sig("mgc_dev_synth_reads", i32, u64, u64, u64, u64, u32, u32, u32, vp, vp)
    sig("mgc_dev_synth_reads_ex", i32, u64, u64, u64, u64, u32, u32, u32, u32, u32, u32, vp, vp)
    sig("mgc_open", vp, P(CountConfig), i32)
    sig("mgc_close", None, vp)
    sig("mgc_last_error", ctypes.c_char_p, vp)
    sig("mgc_push_bases", i32, vp, ctypes.c_char_p, sz, i32)
    sig("mgc_push_bases_device", i32, vp, vp, u64)
    sig("mgc_set_batch_bases", i32, vp, u64)
    sig("mgc_prepare", i32, vp, u64)
    sig("mgc_set_result_budget", i32, vp, u64)
    sig("mgc_result_out_of_core", i32, vp)
    sig("mgc_get_runs_profile", i32, vp, P(RunsProfile))
    sig("mgc_runs_open", vp, u32, u32, i32, u64, u64)
    sig("mgc_runs_add", i32, vp, vp, vp, u64, vp)
    sig("mgc_runs_write", i32, vp, vp, u64, u64)
    sig("mgc_runs_get_profile", i32, vp, P(RunsProfile))
    sig("mgc_runs_error", ctypes.c_char_p, vp)
    sig("mgc_runs_close", None, vp)
    sig("mgc_db_stream_queued", u64, vp)
    sig("mgc_db_stream_done", u64, vp)
    sig("mgc_db_stream_wait_buffers", i32, vp, u64)
    sig("mgc_reserve_text", i32, vp, u64)
    sig("mgc_begin_text", i32, vp, i32)
    sig("mgc_push_text", i32, vp, ctypes.c_char_p, sz)
    sig("mgc_end_text", i32, vp)
    sig("mgc_push_text_file", i32, vp, ctypes.c_char_p, i32, i32)
    sig("mgc_push_text_file_range", i32, vp, ctypes.c_char_p, i32, i32, u64, u64)
    sig("mgc_push_text_bgzf_file", i32, vp, ctypes.c_char_p, i32, i32)
    sig("mgc_is_bgzf_file", i32, ctypes.c_char_p)
    sig("mgc_text_record_start", i32, ctypes.c_char_p, i32, u64, P(u64))
    sig("mgc_count", i32, vp)
    sig("mgc_count_partitioned", i32, vp, vp, vp, vp)
    sig("mgc_count_buckets", i32, vp, vp, u32, vp)
    sig("mgc_count_buckets_into", i32, vp, vp, u32, vp, vp, vp, u64, P(u64), vp)
    sig("mgc_dev_kmer_histogram_fine", i32, vp, u64, u32, i32, u32, vp, vp, vp, sz, vp)
    sig("mgc_copy_result_device", i32, vp, vp, vp)
    sig("mgc_get_result_info", i32, vp, P(ResultInfo))
    sig("mgc_get_result_device", i32, vp, P(vp), P(vp), P(vp), P(u32))
    sig("mgc_copy_result", i32, vp, vp, vp, vp, vp)
    sig("mgc_finish", i32, vp, BLOCK_CB, vp, i32)
    sig("mgc_finish_labelled", i32, vp, BLOCK_CB2, vp, i32)
    sig("mgc_set_profiling", i32, vp, i32)
    sig("mgc_get_profile", i32, vp, P(Profile))
    sig("mdb_writer_open", vp, ctypes.c_char_p, u32, u32)
    sig("mdb_writer_open_ex", vp, ctypes.c_char_p, u32, u32, u32, u32, u32)
    sig("mdb_merge_parts", i32, ctypes.c_char_p, u32)
    sig("mdb_writer_add_block", i32, vp, u64, u64, vp, vp, vp)
    sig("mdb_writer_add_block_labelled", i32, vp, u64, u64, vp, vp, vp, vp, u64)
    sig("mdb_writer_add_encoded", i32, vp, u32, vp, u64, vp, u64)
    sig("mdb_writer_reserve_encoded", i32, vp, u32, u64, vp, u64, P(u64))
    sig("mdb_writer_write_at", i32, vp, u32, u64, vp, u64)
    sig("mdb_writer_add_histogram", i32, vp, vp, vp, u64)
    sig("mdb_writer_close", i32, vp)
    sig("mdb_last_error", ctypes.c_char_p)
    sig("mdb_reader_open", vp, ctypes.c_char_p)
    sig("mdb_reader_info", i32, vp, P(DbInfo))
    sig("mdb_reader_histogram", i32, vp, vp, vp)
    sig("mdb_reader_read_file", i32, vp, u32, P(vp), P(vp), P(vp), P(u64))
    sig("mdb_reader_read_file_ex", i32, vp, u32, P(vp), P(vp), P(vp), P(vp), P(u64))
    sig("mdb_reader_file_index", i32, vp, u32, vp)
    sig("mdb_reader_block_header", i32, vp, u32, u64, P(BlockHeader))
    sig("mdb_reader_read_block_raw", i32, vp, u32, u64, P(BlockHeader), P(vp), P(vp), P(vp), P(vp), P(vp))
    sig("mdb_reader_raw_file", i32, vp, u32, P(vp), P(u64), P(vp), P(u64), P(u64))
    sig("mdb_reader_close", None, vp)
    sig("mdb_free", None, vp)
    sig("mgc_write_database", i32, vp, ctypes.c_char_p, i32)
    sig("mgc_write_database_profiled", i32, vp, ctypes.c_char_p, i32, P(DbWriteProfile))
    sig("mgc_db_stream_open", vp, ctypes.c_char_p, u32, u32, u32, u64, u32, u32, i32, i32)
    sig("mgc_db_stream_write", i32, vp, vp, vp, u64, u64, u64)
    sig("mgc_db_stream_sync", i32, vp)
    sig("mgc_db_stream_close", i32, vp, P(DbWriteProfile))
    sig("mgc_db_stream_error", ctypes.c_char_p, vp)
    sig("mgc_db_merge", i32, P(ctypes.c_char_p), u32, i32, ctypes.c_char_p, i32, i32)
    sig("mgc_count_node", i32, P(CountConfig), u32, P(ctypes.c_int), P(vp), P(u64), ctypes.c_char_p, i32, P(NodeProfile))
    sig("mgc_count_node_batched", i32, P(CountConfig), u32, P(ctypes.c_int), P(vp), P(u64), u64, ctypes.c_char_p, i32, P(NodeProfile))
    sig("mgc_staged_bases", i32, vp, P(vp), P(u64))
    sig("mgc_count_node_staged", i32, vp, u32, P(ctypes.c_int), ctypes.c_char_p, i32, P(NodeProfile))
    sig("mgc_node_plan", i32, u32, u32, u64, u32, P(u32), P(u64), P(u32))
    sig("mgc_lookup_load", vp, ctypes.c_char_p, u64, u64, i32, i32)
    sig("mgc_lookup_estimate", i32, ctypes.c_char_p, u64, u64, P(LookupInfo))
    sig("mgc_lookup_from_device", vp, vp, vp, u64, u32, u64, u64, i32)
    sig("mgc_lookup_free", None, vp)
    sig("mgc_lookup_get_info", i32, vp, P(LookupInfo))
    sig("mgc_lookup_error", ctypes.c_char_p)
    sig("mgc_lookup_values", i32, vp, vp, u64, vp, vp)
    sig("mgc_lookup_stream", i32, vp, vp, u64, vp, vp)
    sig("mgc_lookup_existence", i32, vp, vp, u64, vp, u64, vp, vp, vp)
    sig("msr_open", vp, ctypes.c_char_p)
    sig("msr_close", None, vp)
    sig("msr_read_text", ctypes.c_int64, vp, vp, u64)
    sig("msr_last_error", ctypes.c_char_p)
    sig("msr_load_bases", i32, vp, vp, u64, P(u64), P(i32))
    sig("msr_is_compressed", i32, vp)
    sig("msr_format", i32, vp)
    sig("msr_load_stream", i32, vp, vp, u64, P(u64))
    sig("msr_guess_number_of_kmers", u64, ctypes.c_char_p)
    _lib = L
    return L


def last_error(handle=None):
    s = lib().mgc_last_error(handle)
    return s.decode("utf-8", "replace") if s else ""


def check(rc, what, handle=None):
    if rc != MGC_OK:
        raise MgcError(rc, what, last_error(handle))


def configure(k, n_kmers_estimate, memory_bytes, mode=MODE_CANONICAL, threads=0, count_suffix_length=0,
              homopoly_compress=0, count_suffix="", label_size=0, label=0):
    """mgc_configure_counting -> filled CountConfig."""
    c = CountConfig()
    c.k = k
    c.mode = mode
    c.n_kmers_estimate = int(n_kmers_estimate)
    c.memory_allowed = int(memory_bytes)
    c.threads = threads
    c.count_suffix_length = len(count_suffix) if count_suffix else count_suffix_length
    c.count_suffix = count_suffix.encode("ascii")
    c.homopoly_compress = homopoly_compress
    c.label_size = label_size
    c.label_constant = label
    check(lib().mgc_configure_counting(ctypes.byref(c)), "mgc_configure_counting")
    return c


def configured_line(cfg):
    buf = ctypes.create_string_buffer(256)
    check(lib().mgc_format_configured_line(ctypes.byref(cfg), buf, 256), "mgc_format_configured_line")
    return buf.value.decode("ascii")
