"""Python plumbing over the C-ABI: torch owns device memory and streams, the
native library (include/meryl_gpu_count.h) does all the work.

Single GPU : `Session` (mgc_open / mgc_push_bases[_device] / mgc_count / ...)
             or the stateless `dev_*` operators on torch tensors.
Multi GPU  : `count_sharded` -- one process per GPU; every rank packs its own
             reads, the 64*N top-bit buckets are split into contiguous per-rank
             ranges and k-mers are routed to their owning rank in waves of
             point-to-point messages (RCCL on GPUs, gloo in the CPU tests) while
             the owner counts the buckets that have already arrived
             (mgc_count_buckets).
"""
import ctypes
import os
import sys

import numpy as np

from . import capi

try:  # torch is plumbing only (device buffers, streams, torch.distributed)
    import torch
except Exception:  # pragma: no cover
    torch = None


def _stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() else ctypes.c_void_p(0)


def _u64(n, device):
    # torch has no general uint64 arithmetic; int64 storage is bit-identical
    return torch.empty(int(n), dtype=torch.int64, device=device)


# ---------------------------------------------------------------------------
# stateless device operators
# ---------------------------------------------------------------------------
def dev_synth_reads(seed, genome_len, first_read, n_reads, read_len=150, sub_rate_ppm=5000, n_rate_ppm=100,
                    device="cuda", repeat_ppm=0, repeat_unit=300, repeat_families=1000):
    out = torch.empty(int(n_reads) * (read_len + 1), dtype=torch.uint8, device=device)
    capi.check(capi.lib().mgc_dev_synth_reads_ex(seed, genome_len, first_read, n_reads, read_len, sub_rate_ppm,
                                                 n_rate_ppm, repeat_ppm, repeat_unit, repeat_families, _ptr(out),
                                                 _stream_ptr()), "mgc_dev_synth_reads_ex")
    return out


def dev_homopoly_compress(bases):
    """uint8 cuda tensor -> homopolymer-compressed uint8 cuda tensor (`compress`)."""
    L = capi.lib()
    n = bases.numel()
    out = torch.empty(n, dtype=torch.uint8, device=bases.device)
    ws_bytes = L.mgc_dev_homopoly_workspace_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=bases.device)
    n_out = ctypes.c_uint64(0)
    capi.check(L.mgc_dev_homopoly_compress(_ptr(bases), n, _ptr(out), ctypes.byref(n_out), _ptr(ws), ws_bytes,
                                           _stream_ptr()), "mgc_dev_homopoly_compress")
    return out[:n_out.value]


def key_words(k):
    """uint64 keys for k <= 32, 16-byte {lo, hi} pairs for k in 33..64."""
    return 2 if k > 32 else 1


def dev_kmer_partition(bases, k, mode=capi.MODE_CANONICAL, bucket_bits=6):
    """bases: uint8 cuda tensor -> (keys grouped by bucket, counts uint64[2^bucket_bits] on host).
    keys is int64[N] for k <= 32 and int64[N, 2] ({lo, hi} rows) for k > 32."""
    L = capi.lib()
    dev = bases.device
    nb = 1 << bucket_bits
    ws_bytes = L.mgc_dev_partition_workspace_bytes(bucket_bits)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    counts = _u64(nb, dev)
    capi.check(L.mgc_dev_kmer_histogram(_ptr(bases), bases.numel(), k, mode, bucket_bits, _ptr(counts), _ptr(ws),
                                        ws_bytes, _stream_ptr()), "mgc_dev_kmer_histogram")
    h_counts = counts.cpu().numpy().astype(np.uint64)
    starts = np.zeros(nb, dtype=np.uint64)
    starts[1:] = np.cumsum(h_counts)[:-1]
    n = int(h_counts.sum())
    d_starts = torch.from_numpy(starts.astype(np.int64)).to(dev)
    keys = _u64(n * key_words(k), dev)
    if k > 32:
        keys = keys.view(n, 2)
    capi.check(L.mgc_dev_kmer_partition(_ptr(bases), bases.numel(), k, mode, bucket_bits, _ptr(d_starts), _ptr(keys),
                                        _ptr(ws), ws_bytes, _stream_ptr()), "mgc_dev_kmer_partition")
    return keys, h_counts


def dev_kmer_histogram_keep(bases, k, mode=capi.MODE_CANONICAL, bucket_bits=6):
    """k-mers per bucket of a base stream -> (uint64[2^bucket_bits] on the host, token).  The token keeps the per-workgroup
    histogram rows the partition kernel takes its private cursors from (dev_kmer_partition_into)."""
    L = capi.lib()
    nb = 1 << bucket_bits
    ws_bytes = L.mgc_dev_partition_workspace_bytes(bucket_bits)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=bases.device)
    counts = _u64(nb, bases.device)
    fine = None
    if 6 <= bucket_bits <= 8 and 2 * k >= 17 and os.environ.get("MGC_SHARD_FINE", "1") != "0":
        # ... and the k-mers per top FIFTEEN bits in the same pass: summed over the ranks they are the first grouping digit of every
        # owner-side bucket (count_sharded -> mgc_count_buckets_into), so that the owner does not read its keys for a histogram
        fine = _u64(1 << 15, bases.device)
        capi.check(L.mgc_dev_kmer_histogram_fine(_ptr(bases), bases.numel(), k, mode, bucket_bits, _ptr(counts), _ptr(fine), _ptr(ws), ws_bytes,
                                                 _stream_ptr()), "mgc_dev_kmer_histogram_fine")
    else:
        capi.check(L.mgc_dev_kmer_histogram(_ptr(bases), bases.numel(), k, mode, bucket_bits, _ptr(counts), _ptr(ws), ws_bytes,
                                            _stream_ptr()), "mgc_dev_kmer_histogram")
    h_counts = counts.cpu().numpy().astype(np.uint64)
    return h_counts, (bases, k, mode, bucket_bits, ws, ws_bytes, h_counts, fine)


def dev_kmer_partition_into(token, starts, out):
    """The partition of dev_kmer_histogram_keep's base stream with EXPLICIT bucket starts (key indices into `out`, any order,
    gaps allowed): bucket b's k-mers land at out[starts[b] : starts[b] + count[b]].  What lets a sharded count write the
    buckets a rank owns itself straight into its inbox instead of copying them there (count_sharded)."""
    bases, k, mode, bucket_bits, ws, ws_bytes, h_counts = token[:7]
    starts = np.asarray(starts, dtype=np.uint64)
    # a wrong plan would be a silent out-of-bounds scatter on the device: the shape of `out` and every bucket's range are checked here
    if starts.shape != h_counts.shape:
        raise ValueError("dev_kmer_partition_into: %d starts for %d buckets" % (starts.size, h_counts.size))
    want_dim = 2 if k > 32 else 1
    if out.dtype != torch.int64 or out.dim() != want_dim or (want_dim == 2 and out.shape[1] != 2) or not out.is_contiguous():
        raise ValueError("dev_kmer_partition_into: out must be a contiguous int64[N%s] tensor for k = %d" % (", 2" if want_dim == 2 else "", k))
    if out.device != bases.device:
        raise ValueError("dev_kmer_partition_into: out lives on %s, the bases on %s" % (out.device, bases.device))
    nz = h_counts > 0
    if nz.any() and int((starts[nz] + h_counts[nz]).max()) > int(out.shape[0]):
        raise ValueError("dev_kmer_partition_into: a bucket ends at key %d, out holds %d" % (int((starts[nz] + h_counts[nz]).max()), int(out.shape[0])))
    d_starts = torch.from_numpy(starts.astype(np.int64)).to(bases.device)
    capi.check(capi.lib().mgc_dev_kmer_partition(_ptr(bases), bases.numel(), k, mode, bucket_bits, _ptr(d_starts), _ptr(out),
                                                 _ptr(ws), ws_bytes, _stream_ptr()), "mgc_dev_kmer_partition")


def dev_radix_sort(keys, begin_bit, end_bit, group=False):
    """Sorts keys on bits [begin_bit, end_bit): an int64[N] cuda tensor (as uint64) or an
    int64[N, 2] tensor of {lo, hi} rows (128-bit keys).  Returns the sorted tensor.
    group=True: the count path's grouping passes (mgc_dev_radix_group): grouped by those bits, members in any order."""
    L = capi.lib()
    kw = 2 if keys.dim() == 2 else 1
    n = keys.shape[0]
    if n == 0:
        return keys
    alt = torch.empty_like(keys)
    ws_bytes = L.mgc_dev_sort_workspace_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=keys.device)
    in_alt = ctypes.c_int(0)
    op = L.mgc_dev_radix_group if group else L.mgc_dev_radix_sort
    capi.check(op(_ptr(keys), _ptr(alt), n, kw, begin_bit, end_bit, _ptr(ws), ws_bytes,
                  ctypes.byref(in_alt), _stream_ptr()), "mgc_dev_radix_group" if group else "mgc_dev_radix_sort")
    return alt if in_alt.value else keys


def dev_run_length(sorted_keys):
    """(unique, counts int32[D]) of a sorted key tensor (int64[N] or int64[N, 2])."""
    L = capi.lib()
    kw = 2 if sorted_keys.dim() == 2 else 1
    n = sorted_keys.shape[0]
    dev = sorted_keys.device
    ws_bytes = L.mgc_dev_rle_workspace_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    nd = ctypes.c_uint64(0)
    capi.check(L.mgc_dev_rle_count(_ptr(sorted_keys), n, kw, _ptr(ws), ws_bytes, ctypes.byref(nd), _stream_ptr()),
               "mgc_dev_rle_count")
    uniq = _u64(nd.value * kw, dev)
    if kw == 2:
        uniq = uniq.view(nd.value, 2)
    cnts = torch.empty(nd.value, dtype=torch.int32, device=dev)
    capi.check(L.mgc_dev_rle_emit(_ptr(sorted_keys), n, kw, _ptr(ws), ws_bytes, _ptr(uniq), _ptr(cnts), _stream_ptr()),
               "mgc_dev_rle_emit")
    return uniq, cnts


def count_node(cfg, bases, path, devices=None, host_threads=8, batch_bases=0):
    """mgc_count_node[_batched]: ONE count over several ranks of this process -- bases[r] is rank r's reads (uint8 cuda
    tensor, on the device the rank runs on; ranks may share a device) -> the database at `path`, byte-identical to a
    single-device count of the concatenation.  batch_bases: no rank holds the k-mers of more of its bases at once (0: from
    the free HBM).  Returns the profile as a dict."""
    n = len(bases)
    ptrs = (ctypes.c_void_p * n)(*[int(b.data_ptr()) if b.numel() else None for b in bases])
    lens = (ctypes.c_uint64 * n)(*[int(b.numel()) for b in bases])
    if devices is None:
        devices = [b.device.index if b.device.index is not None else torch.cuda.current_device() for b in bases]
    devs = (ctypes.c_int * n)(*[int(d) for d in devices])
    prof = capi.NodeProfile()
    torch.cuda.synchronize()
    rc = capi.lib().mgc_count_node_batched(ctypes.byref(cfg), n, devs, ptrs, lens, int(batch_bases), os.fsencode(path), host_threads,
                                           ctypes.byref(prof))
    if rc != 0:
        raise RuntimeError("mgc_count_node failed rc=%d: %s" % (rc, (capi.lib().mgc_last_error(None) or b"").decode()))
    return prof.as_dict()


MERGE_OPS = {"union-sum": 0, "union-min": 1, "union-max": 2, "intersect-sum": 3, "intersect-min": 4, "intersect-max": 5,
             "intersect": 6, "subtract": 7, "difference": 8, "symmetric-difference": 9}


def dev_merge(keys_a, counts_a, keys_b, counts_b, op="union-sum"):
    """Two (k-mer, value) streams with distinct ascending keys (int64[N] / int64[N, 2] cuda tensors, int32 values) ->
    their union / intersection with combined values (mgc_dev_merge_*)."""
    L = capi.lib()
    kw = 2 if keys_a.dim() == 2 else 1
    na, nb = keys_a.shape[0], keys_b.shape[0]
    dev = keys_a.device
    ws_bytes = L.mgc_dev_merge_workspace_bytes(na, nb)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    n = ctypes.c_uint64(0)
    code = MERGE_OPS[op]
    capi.check(L.mgc_dev_merge_count_values(_ptr(keys_a), _ptr(counts_a), na, _ptr(keys_b), _ptr(counts_b), nb, kw, code, _ptr(ws), ws_bytes,
                                            ctypes.byref(n), _stream_ptr()), "mgc_dev_merge_count_values")
    out_k = _u64(n.value * kw, dev)
    if kw == 2:
        out_k = out_k.view(n.value, 2)
    out_c = torch.empty(n.value, dtype=torch.int32, device=dev)
    capi.check(L.mgc_dev_merge_emit(_ptr(keys_a), _ptr(counts_a), na, _ptr(keys_b), _ptr(counts_b), nb, kw, code, _ptr(ws),
                                    ws_bytes, _ptr(out_k), _ptr(out_c), _stream_ptr()), "mgc_dev_merge_emit")
    return out_k, out_c


def dev_block_offsets(unique, w_data, n_prefix):
    out = _u64(n_prefix + 1, unique.device)
    kw = 2 if unique.dim() == 2 else 1
    capi.check(capi.lib().mgc_dev_block_offsets(_ptr(unique), unique.shape[0], kw, w_data, n_prefix, _ptr(out),
                                                _stream_ptr()), "mgc_dev_block_offsets")
    return out


# ---------------------------------------------------------------------------
# session (mirrors merylOperation::countThreads, merylOp-countThreads.C:385-474)
# ---------------------------------------------------------------------------
class Session:
    def __init__(self, cfg, device=-1):
        self.cfg = cfg
        self._h = capi.lib().mgc_open(ctypes.byref(cfg), device)
        if not self._h:
            raise capi.MgcError(-1, "mgc_open", capi.last_error(None))
        self._keep = []

    def close(self):
        if self._h:
            capi.lib().mgc_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def push_bases(self, bases, end_of_sequence=True):
        b = bases.encode("ascii") if isinstance(bases, str) else bytes(bases)
        capi.check(capi.lib().mgc_push_bases(self._h, b, len(b), 1 if end_of_sequence else 0), "mgc_push_bases",
                   self._h)

    def push_text(self, text, fmt, pieces=None):
        """One whole sequence file as raw text (bytes/str), parsed on the device.  fmt: "fasta" | "fastq".
        pieces: optional chunk size (the text is fed in pieces of that many bytes -- tests use odd sizes).
        Raises MgcError(code MGC_EFORMAT) if the device parser refuses the file (its output is rolled back)."""
        b = text.encode("ascii") if isinstance(text, str) else bytes(text)
        L = capi.lib()
        capi.check(L.mgc_begin_text(self._h, 1 if fmt == "fasta" else 2), "mgc_begin_text", self._h)
        step = pieces or max(len(b), 1)
        for i in range(0, len(b), step):
            piece = b[i:i + step]
            capi.check(L.mgc_push_text(self._h, piece, len(piece)), "mgc_push_text", self._h)
        capi.check(L.mgc_end_text(self._h), "mgc_end_text", self._h)

    def push_bases_device(self, t):
        # the session runs on its own HIP stream: whatever produced `t` on torch's
        # stream must be complete before mgc_count reads it
        torch.cuda.current_stream().synchronize()
        self._keep.append(t)
        capi.check(capi.lib().mgc_push_bases_device(self._h, _ptr(t), t.numel()), "mgc_push_bases_device", self._h)

    def set_batch_bases(self, n):
        """force out-of-core batches of about n bases (default: derived from the free HBM)"""
        capi.check(capi.lib().mgc_set_batch_bases(self._h, int(n)), "mgc_set_batch_bases", self._h)

    def set_result_budget(self, nbytes):
        """bytes of batch results (runs) that may stay in HBM; beyond it they are parked in pinned host DRAM"""
        capi.check(capi.lib().mgc_set_result_budget(self._h, int(nbytes)), "mgc_set_result_budget", self._h)

    def out_of_core(self):
        """True when the counted result exists only as runs (stream it: write_database / finish)"""
        return bool(capi.lib().mgc_result_out_of_core(self._h))

    def runs_profile(self):
        p = capi.RunsProfile()
        capi.check(capi.lib().mgc_get_runs_profile(self._h, ctypes.byref(p)), "mgc_get_runs_profile", self._h)
        return p.as_dict()

    def set_profiling(self, on=True):
        capi.check(capi.lib().mgc_set_profiling(self._h, 1 if on else 0), "mgc_set_profiling", self._h)

    def count(self):
        capi.check(capi.lib().mgc_count(self._h), "mgc_count", self._h)

    def count_partitioned(self, keys, bucket_counts):
        """Owner side of a sharded count: `keys` (int64[N] or int64[N, 2] cuda tensor) already hold canonical k-mers
        laid out bucket-major; `bucket_counts` has 64 entries (files) or 2^b, b in 7..10 (finer ranges of the top bits);
        processed in place."""
        fc = np.ascontiguousarray(np.asarray(bucket_counts, dtype=np.uint64))
        bits = int(fc.size).bit_length() - 1
        assert fc.size == (1 << bits) and 6 <= bits <= 10 and int(fc.sum()) == keys.shape[0]
        torch.cuda.current_stream(keys.device).synchronize()
        capi.check(capi.lib().mgc_count_buckets(self._h, _ptr(keys) if keys.shape[0] else None, bits, fc.ctypes.data),
                   "mgc_count_buckets", self._h)

    def count_partitioned_into(self, keys, bucket_counts, out_keys, out_counts, fine=None):
        """count_partitioned with the packed result written straight into out_keys / out_counts (pre-sized cuda tensors) when it
        fits; returns (n_distinct, fitted).  fine: int64[2^15] cuda tensor, the k-mers per top fifteen bits over all ranks."""
        fc = np.ascontiguousarray(np.asarray(bucket_counts, dtype=np.uint64))
        nb = fc.size
        bits = int(nb).bit_length() - 1
        assert nb == 1 << bits and nb >= 64
        assert int(fc.sum()) == int(keys.shape[0])
        cap = int(out_keys.shape[0])
        assert out_counts.shape[0] >= cap and out_counts.dtype == torch.int32 and out_keys.dtype == torch.int64 and out_keys.is_contiguous()
        torch.cuda.current_stream(keys.device).synchronize()
        n = ctypes.c_uint64(0)
        capi.check(capi.lib().mgc_count_buckets_into(self._h, _ptr(keys) if keys.shape[0] else None, bits, fc.ctypes.data,
                                                     _ptr(out_keys) if cap else None, _ptr(out_counts) if cap else None, cap, ctypes.byref(n),
                                                     _ptr(fine) if fine is not None else None),
                   "mgc_count_buckets_into", self._h)
        return int(n.value), int(n.value) <= cap

    def result_device(self):
        """(distinct keys, counts int32) as fresh cuda tensors (device-to-device copy)."""
        r = self.info()
        dev = torch.device("cuda", torch.cuda.current_device())
        kw = 2 if self.cfg.k > 32 else 1
        keys = _u64(r.n_distinct * kw, dev)
        if kw == 2:
            keys = keys.view(r.n_distinct, 2)
        cnts = torch.empty(r.n_distinct, dtype=torch.int32, device=dev)
        capi.check(capi.lib().mgc_copy_result_device(self._h, _ptr(keys) if r.n_distinct else None,
                                                     _ptr(cnts) if r.n_distinct else None), "mgc_copy_result_device", self._h)
        return keys, cnts

    def info(self):
        r = capi.ResultInfo()
        capi.check(capi.lib().mgc_get_result_info(self._h, ctypes.byref(r)), "mgc_get_result_info", self._h)
        return r

    def profile(self):
        p = capi.Profile()
        capi.check(capi.lib().mgc_get_profile(self._h, ctypes.byref(p)), "mgc_get_profile", self._h)
        return p

    def result(self):
        """(keys uint64[D] (low 64 bits), counts uint32[D], block_start uint64[n_prefix+1]) as numpy
        arrays; for k > 32 use result_wide() to get the high halves too."""
        lo, _, counts, bstart = self.result_wide()
        return lo, counts, bstart

    def result_wide(self):
        """(keys_lo, keys_hi, counts, block_start)"""
        r = self.info()
        lo = np.zeros(r.n_distinct, dtype=np.uint64)
        hi = np.zeros(r.n_distinct, dtype=np.uint64)
        counts = np.zeros(r.n_distinct, dtype=np.uint32)
        bstart = np.zeros(r.n_prefix + 1, dtype=np.uint64)
        capi.check(capi.lib().mgc_copy_result(self._h, lo.ctypes.data, hi.ctypes.data, counts.ctypes.data,
                                              bstart.ctypes.data), "mgc_copy_result", self._h)
        return lo, hi, counts, bstart

    def write_database(self, path, host_threads=8):
        """Result -> database directory (blocks encoded on the device); returns the write profile as a dict."""
        prof = capi.DbWriteProfile()
        capi.check(capi.lib().mgc_write_database_profiled(self._h, path.encode(), host_threads, ctypes.byref(prof)),
                   "mgc_write_database", self._h)
        return prof.as_dict()

    def finish(self, callback, host_threads=1):
        """callback(prefix, n_kmers, suffix_lo uint64[n], counts uint32[n][, suffix_hi]) per block,
        addBlock order; suffix_hi is passed (5th argument) only when the callback accepts it."""
        import inspect
        err = []
        wants_hi = len(inspect.signature(callback).parameters) >= 5

        def _cb(ctx, prefix, n, slo, shi, cnt):
            try:
                s = np.ctypeslib.as_array(slo, shape=(n,)).copy() if n else np.zeros(0, np.uint64)
                c = np.ctypeslib.as_array(cnt, shape=(n,)).copy() if n else np.zeros(0, np.uint32)
                if wants_hi:
                    h = np.ctypeslib.as_array(shi, shape=(n,)).copy() if (n and shi) else np.zeros(n, np.uint64)
                    callback(int(prefix), int(n), s, c, h)
                else:
                    callback(int(prefix), int(n), s, c)
                return 0
            except Exception as e:  # pragma: no cover
                err.append(e)
                return -1

        cb = capi.BLOCK_CB(_cb)
        rc = capi.lib().mgc_finish(self._h, cb, None, host_threads)
        if err:
            raise err[0]
        capi.check(rc, "mgc_finish", self._h)


class Runs:
    """Sorted runs of partial (k-mer, count) results (mgc_runs_*, include/meryl_db.h): parked in HBM within
    `device_budget` bytes, in pinned host DRAM beyond it; write() merges them once into a database stream."""

    def __init__(self, k, w_prefix, device=-1, device_budget=0xFFFFFFFFFFFFFFFF, chunk_bytes=0):
        self._h = capi.lib().mgc_runs_open(k, w_prefix, device, device_budget, chunk_bytes)
        if not self._h:
            raise capi.MgcError(-1, "mgc_runs_open", (capi.lib().mgc_runs_error(None) or b"").decode("utf-8", "replace"))

    def _check(self, rc, what):
        if rc != 0:
            raise capi.MgcError(rc, what, (capi.lib().mgc_runs_error(self._h) or b"").decode("utf-8", "replace"))

    def add(self, keys, counts):
        """keys: int64[n] / int64[n, 2] cuda tensor (ascending, distinct), counts int32[n]; copied before the call returns"""
        torch.cuda.current_stream(keys.device).synchronize()
        self._check(capi.lib().mgc_runs_add(self._h, _ptr(keys), _ptr(counts), keys.shape[0], None), "mgc_runs_add")

    def write(self, stream, prefix_begin, prefix_end):
        self._check(capi.lib().mgc_runs_write(self._h, stream._h, int(prefix_begin), int(prefix_end)), "mgc_runs_write")

    def profile(self):
        p = capi.RunsProfile()
        self._check(capi.lib().mgc_runs_get_profile(self._h, ctypes.byref(p)), "mgc_runs_get_profile")
        return p.as_dict()

    def close(self):
        if self._h:
            capi.lib().mgc_runs_close(self._h)
            self._h = None


class DbStream:
    """Device-resident (k-mer, count) ranges -> database files, encoded on the device (mgc_db_stream_*,
    include/meryl_db.h).  One stream = one writer, or part `part` of `n_parts` of a sharded database."""

    def __init__(self, path, k, w_prefix, label_size=0, label=0, part=0, n_parts=1, host_threads=8, device=-1):
        self._h = capi.lib().mgc_db_stream_open(path.encode(), k, w_prefix, label_size, label, part, n_parts,
                                                host_threads, device)
        if not self._h:
            raise capi.MgcError(-1, "mgc_db_stream_open", capi.lib().mgc_db_stream_error(None).decode("utf-8", "replace"))
        self._keep = []

    def _check(self, rc, what):
        if rc != 0:
            msg = capi.lib().mgc_db_stream_error(self._h if self._h else None)
            raise capi.MgcError(rc, what, msg.decode("utf-8", "replace") if msg else "")

    def write(self, keys, counts, prefix_begin, prefix_end):
        """Queues the blocks of prefixes [prefix_begin, prefix_end): `keys` (int64[n] / int64[n, 2] cuda tensor, ascending)
        hold exactly the distinct k-mers of that range, `counts` (int32[n]) their counts.  Asynchronous: the tensors are
        kept alive here until sync()/close()."""
        torch.cuda.current_stream(keys.device).synchronize()       # the stream's own HIP streams read them
        self._check(capi.lib().mgc_db_stream_write(self._h, _ptr(keys), _ptr(counts), keys.shape[0], int(prefix_begin),
                                                   int(prefix_end)), "mgc_db_stream_write")
        self._keep.append((int(capi.lib().mgc_db_stream_queued(self._h)), keys, counts))

    def release_done(self):
        """drops the tensors of the ranges that have been encoded and copied out (no waiting)"""
        done = int(capi.lib().mgc_db_stream_done(self._h))
        self._keep = [e for e in self._keep if e[0] > done]

    def sync(self):
        self._check(capi.lib().mgc_db_stream_sync(self._h), "mgc_db_stream_sync")
        self._keep = []

    def close(self):
        """-> profile dict (plan_ms, encode_ms, copy_write_s, total_s, data_bytes, n_kmers, n_blocks)"""
        if not self._h:
            return None
        prof = capi.DbWriteProfile()
        h, self._h = self._h, None
        rc = capi.lib().mgc_db_stream_close(h, ctypes.byref(prof))
        self._keep = []
        if rc != 0:
            msg = capi.lib().mgc_db_stream_error(None)
            raise capi.MgcError(rc, "mgc_db_stream_close", msg.decode("utf-8", "replace") if msg else "")
        return prof.as_dict()


def count_bases(bases, k, mode=capi.MODE_CANONICAL, n_estimate=None, memory_gb=4.0, device=-1, profiling=False):
    """Host convenience: bases is str/bytes (with '.' breakers) or a uint8 cuda tensor."""
    n = bases.numel() if (torch is not None and isinstance(bases, torch.Tensor)) else len(bases)
    cfg = capi.configure(k, n_estimate if n_estimate else max(n, 1), int(memory_gb * (1 << 30)), mode)
    with Session(cfg, device) as s:
        if torch is not None and isinstance(bases, torch.Tensor):
            s.push_bases_device(bases)
        else:
            s.push_bases(bases, end_of_sequence=False)
        s.set_profiling(profiling)
        s.count()
        keys, counts, bstart = s.result()
        info = s.info()
        prof = s.profile() if profiling else None
    return keys, counts, bstart, info, prof


# ---------------------------------------------------------------------------
# multi-GPU: contiguous bucket ranges per rank + point-to-point waves
# ---------------------------------------------------------------------------
def balanced_file_ranges(file_counts, world):
    """Contiguous ranges of the 64 files, one per rank, cut so that every rank
    owns about the same number of k-mer instances (canonical prefixes are
    skewed towards A/C, so equal file counts would not balance).  Every rank
    gets at least one file.  Returns world+1 cut points in [0, 64]."""
    fc = np.asarray(file_counts, dtype=np.float64)
    nf = len(fc)
    if world > nf:
        raise ValueError("more ranks (%d) than files (%d)" % (world, nf))
    cum = np.concatenate([[0.0], np.cumsum(fc)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        lo, hi = cuts[-1] + 1, nf - (world - r)          # leave one file for every later rank
        target = total * r / world
        cand = np.arange(lo, hi + 1)
        cuts.append(int(cand[np.argmin(np.abs(cum[cand] - target))]))
    cuts.append(nf)
    return cuts


def exchange_plan(local_file_counts, cuts):
    """Per-destination send counts (in keys) for this rank's partitioned keys."""
    c = np.asarray(local_file_counts, dtype=np.int64)
    return [int(c[cuts[r]:cuts[r + 1]].sum()) for r in range(len(cuts) - 1)]


def owned_sort_bits(k, first_file, end_file):
    """Bits an owner must sort: everything below the 6 file bits plus the file
    bits in which its (contiguous) files differ."""
    if end_file - first_file <= 1:
        return 2 * k - 6
    return 2 * k - 6 + int(first_file ^ (end_file - 1)).bit_length()


EXCHANGE_CHUNK = 1 << 27          # keys per message: keeps every send/recv far below 2^31 bytes/elements


def _host_staged(t, group=None):
    """CUDA tensors over a backend that only moves host memory (gloo): two ranks sharing ONE device can run the real plan with the
    real HIP operators -- RCCL refuses duplicate GPUs, gloo does not (tests/test_dist_one_gpu.py) -- through pinned-less host
    copies.  Never taken on the product path (backend nccl = RCCL)."""
    import torch.distributed as dist
    return bool(getattr(t, "is_cuda", False)) and dist.get_backend(group) != "nccl"


class _StagedRecv:
    """an irecv into a host buffer + the copy to its device destination once it has arrived"""
    def __init__(self, req, host, dst):
        self.req, self.host, self.dst = req, host, dst

    def wait(self):
        self.req.wait()
        self.dst.copy_(self.host)


def _all_reduce(t, op, group):
    import torch.distributed as dist
    if _host_staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)


def _all_gather(outs, t, group):
    import torch.distributed as dist
    if _host_staged(t, group):
        hs = [o.cpu() for o in outs]
        dist.all_gather(hs, t.cpu(), group=group)
        for o, h in zip(outs, hs):
            o.copy_(h)
    else:
        dist.all_gather(outs, t, group=group)


def exchange_segments(sends, recvs, device, group=None, chunk=None, max_rows=None, wait=True):
    """Variable-size all-to-all as explicit point-to-point segments.
    sends: list of (peer, tensor_view) in the order the peer expects them;
    recvs: list of (peer, tensor_view) in the matching order (per peer, the i-th send of the
    source pairs with the i-th receive of the destination).  Segments to/from the own rank are
    copied locally.  Long segments are cut into rounds of `chunk` rows so that no single message
    exceeds a few GB (torch's all_to_all_single silently mishandles > 2^31-element exchanges, and a
    10 Gbp rank would post 8 GB messages) -- each round is one grouped RCCL launch.
    max_rows: the longest segment over ALL ranks when the caller knows it (every rank must run the same
    number of rounds); None -> one all_reduce finds it.  wait=False returns the outstanding requests
    instead of waiting for them, so the caller can compute while the rounds are in flight."""
    import torch.distributed as dist
    chunk = chunk or EXCHANGE_CHUNK
    rank = dist.get_rank(group)
    if max_rows is None:
        local_max = max([t.shape[0] for _, t in sends] + [t.shape[0] for _, t in recvs] + [0])
        m = torch.tensor([local_max], dtype=torch.int64, device=device)
        _all_reduce(m, dist.ReduceOp.MAX, group)                   # same number of rounds everywhere
        max_rows = int(m.item())
    rounds = max(1, -(-int(max_rows) // chunk))
    own_src = [t for p, t in sends if p == rank]
    own_dst = [t for p, t in recvs if p == rank]
    assert len(own_src) == len(own_dst)
    for a, b in zip(own_src, own_dst):
        b.copy_(a)
    pending = []
    staged = any(_host_staged(t, group) for _, t in sends + recvs)
    for j in range(rounds):
        p2p = []
        landing = []                                              # staged receives: (index in p2p, host buffer, device destination)
        for peer, t in sends:
            if peer != rank:
                piece = t[min(t.shape[0], j * chunk):min(t.shape[0], (j + 1) * chunk)]
                if piece.shape[0]:
                    g = peer if group is None else dist.get_global_rank(group, peer)
                    p2p.append(dist.P2POp(dist.isend, piece.cpu() if staged else piece, g, group))
        for peer, t in recvs:
            if peer != rank:
                piece = t[min(t.shape[0], j * chunk):min(t.shape[0], (j + 1) * chunk)]
                if piece.shape[0]:
                    g = peer if group is None else dist.get_global_rank(group, peer)
                    if staged:
                        host = torch.empty(piece.shape, dtype=piece.dtype)
                        landing.append((len(p2p), host, piece))
                        p2p.append(dist.P2POp(dist.irecv, host, g, group))
                    else:
                        p2p.append(dist.P2POp(dist.irecv, piece, g, group))
        if p2p:
            reqs = list(dist.batch_isend_irecv(p2p))
            if staged and len(reqs) == len(p2p):                  # (one request per operation: gloo)
                for idx, host, dst in landing:
                    reqs[idx] = _StagedRecv(reqs[idx], host, dst)
            elif staged:                                          # a backend that coalesces the batch: wait, then land everything
                for req in reqs:
                    req.wait()
                for _, host, dst in landing:
                    dst.copy_(host)
                reqs = []
            if wait:
                for req in reqs:
                    req.wait()
            else:
                pending.extend(reqs)
    return pending


def dev_count_files(keys, file_counts, k, mode=capi.MODE_CANONICAL, out=None, fine=None):
    """(distinct keys ascending, counts int32) of k-mers already laid out bucket-major (`file_counts`: 64 entries for
    whole files, 2^b for finer buckets).  Everything mgc_count does after the partition, in place on `keys`
    (mgc_count_buckets).  out = (keys tensor, counts tensor): the free tail of a pre-sized result -- the packed result is written
    there when it fits and VIEWS of it are returned (mgc_count_buckets_into: no copy out of the session, nothing to concatenate).
    fine: the k-mers per top fifteen bits over all ranks' reads (int64[2^15] on the device): the first grouping digit of every bucket."""
    dev = keys.device.index if keys.device.index is not None else torch.cuda.current_device()
    # A session reads the MGC_* switches ONCE, when it is opened (mgc_open): the cache is keyed by their current values too, so
    # that a switch set between two sharded counts of one process opens a new session instead of being silently ignored.
    env = tuple(sorted((n, v) for n, v in os.environ.items() if n.startswith("MGC_")))
    key = (k, mode, dev, env)
    s = _SESSIONS.get(key)
    if s is None:                        # kept: the session's device arena is grow-only, a new one would re-malloc tens of GB
        for old in [kk for kk in _SESSIONS if kk[:3] == key[:3]]:      # (the same shape under other switches: its arena goes first)
            _SESSIONS.pop(old).close()
        cfg = capi.configure(k, max(int(keys.shape[0]), 1) * max(k, 1), 64 << 30, mode)
        s = _SESSIONS[key] = Session(cfg, dev)
    if SHARD_PROFILE is not None:
        s.set_profiling(True)
    fitted = False
    if out is not None:
        n_out, fitted = s.count_partitioned_into(keys, file_counts, out[0], out[1], fine)
    elif fine is not None:
        empty_k = keys[:0]
        s.count_partitioned_into(keys, file_counts, empty_k, torch.empty(0, dtype=torch.int32, device=keys.device), fine)
    else:
        s.count_partitioned(keys, file_counts)
    if SHARD_PROFILE is not None:
        p = s.profile()
        SHARD_PROFILE["pass_ms"] = SHARD_PROFILE.get("pass_ms", 0.0) + p.sort_pass_ms_total
        SHARD_PROFILE["pass_launches"] = SHARD_PROFILE.get("pass_launches", 0) + p.sort_pass_launches
        SHARD_PROFILE["pass_keys"] = SHARD_PROFILE.get("pass_keys", 0) + p.sort_pass_keys
        SHARD_PROFILE["pass_bytes"] = SHARD_PROFILE.get("pass_bytes", 0) + p.pass_bytes[0] + p.pass_bytes[1]
        for i in range(2):
            bp = SHARD_PROFILE.setdefault("by_pass", [{"ms": 0.0, "launches": 0, "keys": 0, "bytes": 0} for _ in range(2)])[i]
            bp["ms"] += p.pass_ms[i]; bp["launches"] += p.pass_launches[i]; bp["keys"] += p.pass_keys[i]; bp["bytes"] += p.pass_bytes[i]
    if out is not None:
        if fitted:                       # the packing kernels wrote the caller's buffers (the count ends synchronised): views, no copy
            return out[0][:n_out], out[1][:n_out]
        return s.result_device()         # (did not fit: the caller grows its result and takes this copy)
    return s.result_device()


_SESSIONS = {}
SHARD_PROFILE = None        # bench.py sets this to a dict to collect the owner-side pass timings of the timed steps


def release_cached_sessions():
    """Frees the sessions (and their device arenas) dev_count_files keeps between calls."""
    for s in _SESSIONS.values():
        s.close()
    _SESSIONS.clear()


class HipOps:
    """The device operators count_sharded drives (all HIP, via the C-ABI)."""
    partition = staticmethod(dev_kmer_partition)
    histogram_keep = staticmethod(dev_kmer_histogram_keep)
    partition_into = staticmethod(dev_kmer_partition_into)
    count_files = staticmethod(dev_count_files)

    @staticmethod
    def count_files_into(keys, file_counts, k, mode, out, fine=None):
        return dev_count_files(keys, file_counts, k, mode, out=out, fine=fine)

    @staticmethod
    def empty_keys(n, like):
        if like.dim() == 2:
            return _u64(2 * n, like.device).view(int(n), 2)
        return _u64(n, like.device)

    @staticmethod
    def open_sink(path, k, w_prefix, label_size, label, part, n_parts, host_threads):
        """where the owned (k-mer, count) ranges of a sharded count go: a device-encoding database stream"""
        return DbStream(path, k, w_prefix, label_size, label, part, n_parts, host_threads)

    @staticmethod
    def histogram(bases, k, mode, bucket_bits):
        """k-mers per bucket of a base stream (uint64[2^bucket_bits] on the host): the routing plan's input"""
        L = capi.lib()
        nb = 1 << bucket_bits
        ws_bytes = L.mgc_dev_partition_workspace_bytes(bucket_bits)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=bases.device)
        counts = _u64(nb, bases.device)
        capi.check(L.mgc_dev_kmer_histogram(_ptr(bases), bases.numel(), k, mode, bucket_bits, _ptr(counts), _ptr(ws), ws_bytes,
                                            _stream_ptr()), "mgc_dev_kmer_histogram")
        return counts.cpu().numpy().astype(np.uint64)

    @staticmethod
    def open_runs(k, w_prefix, device_budget):
        """where the counted waves of a BATCHED sharded count wait for the last batch: a run store (HBM, then pinned host DRAM)"""
        return Runs(k, w_prefix, device_budget=device_budget)


def shard_bucket_bits(world, k, n_bases_local=0, w_prefix=None):
    """Top bits of the k-mer that route it in a `world`-rank count: 6 (the files) + ceil(log2(world)), one more for every
    doubling of the per-rank input beyond what keeps a bucket within two grouping digits (~180 M bases), at most 10."""
    extra = max(0, (int(world) - 1).bit_length())
    while extra < 4 and (int(world) * int(n_bases_local)) >> (6 + extra) > 180_000_000:
        extra += 1
    if os.environ.get("MGC_SHARD_BITS"):                   # experiments: the granularity of an N-rank run on fewer ranks
        extra = int(os.environ["MGC_SHARD_BITS"]) - 6
    bits = max(6, min(10, 6 + extra, 2 * int(k)))
    if w_prefix is not None:                                # a database is written: rank ranges must be cut between blocks
        bits = min(bits, int(w_prefix))
    return bits


def batch_slices(n, n_batches, k):
    """[(begin, end)] of the n_batches slices a base stream of n bytes is counted in: slice b = [cut_b - (k-1), cut_{b+1}) -- a
    window that starts in the last k-1 bases of slice b-1 is incomplete there and complete here, so the cuts may fall anywhere
    (also inside a read) and no k-mer is lost or counted twice (mgc_node.cpp:batch_range)."""
    out = []
    for b in range(n_batches):
        cut, end = n * b // n_batches, n * (b + 1) // n_batches
        a = cut - (k - 1) if (b and cut >= k - 1) else 0
        if b and cut < k - 1 and end < k:
            end = a
        out.append((a, end))
    return out


def count_sharded(bases, k, mode=capi.MODE_CANONICAL, group=None, ops=HipOps, db=None, keep_result=True, batch_bases=None,
                  runs_budget=None):
    """Collective.  Every rank passes ITS OWN reads (uint8 tensor on its GPU);
    returns this rank's share of the database: (unique keys, counts int32,
    (first_bucket, end_bucket, bucket_bits)).  The concatenation over ranks, in rank order, is
    the ascending (key, count) stream a single-GPU count of all reads gives.
    `ops` exists so the routing logic can be exercised without a GPU (the gloo
    tests inject CPU stand-ins); the product default is the HIP operators.

    The unit of routing is a BUCKET = a range of the top `bucket_bits` bits of the k-mer: the 64 files for one rank,
    every file cut into 2, 4, 8 ... ranges for 2, 4, 8 ... ranks (shard_bucket_bits), so that an owner-side bucket
    is as large as a single-GPU file however many GPUs feed it (a whole file would be N times larger: a third grouping
    pass, and past 2^30 k-mers the stable wide-granule passes).  Layout after the exchange is bucket-major -- for
    every owned bucket, the pieces of all source ranks back to back -- and each bucket goes through the grouping
    passes and the LDS finish exactly like a file of the single-GPU path (mgc_count_buckets).

    db = dict(path=..., w_prefix=..., label_size=0, label=0, host_threads=8): the ranks also WRITE THE DATABASE -- the
    reference's final dump (merylOp-countThreads.C:452-464) spread over the ranks.  Rank r streams the blocks of its
    bucket range into part r of the directory while later waves are still being exchanged and counted (ops.open_sink;
    the product sink encodes the blocks on the device); after a barrier rank 0 stitches the parts (mdb_merge_parts): the
    64+64+1 files are byte-identical to a single-GPU count of all reads.  A rank range may begin or end inside a file --
    the cut is between two blocks, which is why the routing granularity never exceeds w_prefix bits.
    keep_result=False drops the per-wave tensors once written (the return value then holds empty tensors).

    batch_bases (with db): the reads of every rank are counted in BATCHES of at most that many bases (the sharded form of
    writeBatch's spill, merylOp-countThreads.C:323-379) so that a rank never holds the k-mers of all its reads: the routing
    plan comes from one histogram of all reads; per batch every rank partitions the next slice of its stream (batch_slices:
    k-1 overlap), the waves are exchanged and counted as before, and every counted wave is parked in the owner's run store
    (ops.open_runs: HBM within runs_budget bytes, pinned host DRAM beyond); after the last batch every owner merges its runs
    once into its part.  The database does not depend on the batching."""
    import torch.distributed as dist
    import time as _time
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    prof = os.environ.get("MGC_SHARD_PROFILE") == "1" and torch is not None and bases.is_cuda
    marks = []
    stage_s = {}

    def mark(name):
        if prof:
            torch.cuda.synchronize()
            marks.append((name, _time.perf_counter()))

    mark("start")
    nb_local = torch.tensor([int(bases.numel())], dtype=torch.int64, device=bases.device)
    _all_reduce(nb_local, dist.ReduceOp.MAX, group)                              # every rank must pick the same granularity
    max_local = int(nb_local.item())
    bits = shard_bucket_bits(world, k, max_local, db["w_prefix"] if db else None)
    nbk = 1 << bits
    n_batches = 1
    if batch_bases and db is not None:
        n_batches = max(1, -(-max_local // int(batch_bases)))                   # the same on every rank
    if n_batches > 1:
        # the plan needs the histogram of ALL reads before the first batch is routed
        full = np.asarray(ops.histogram(bases, k, mode, bits)).astype(np.int64)
        fc = torch.from_numpy(full).to(bases.device)
        allf = [torch.empty_like(fc) for _ in range(world)]
        _all_gather(allf, fc, group)
        cuts = balanced_file_ranges(torch.stack(allf).cpu().numpy().sum(axis=0), world)
        slices = batch_slices(int(bases.numel()), n_batches, k)
    else:
        cuts = None
        slices = [(0, int(bases.numel()))]

    sink = None
    runs = None
    n_local_distinct = 0
    parts = []
    # ONE pre-sized result for the waves of an unbatched count (operators that can write into it: the HIP ones): sized after the
    # first wave from its distinct / instances ratio (+ 6 %), grown in the rare case a later wave does not fit.  Without it every
    # wave's result was copied out of the session and all of them concatenated at the end: 18 ms of a 10 Gbp rank's 147.
    res = {"k": None, "c": None, "at": 0, "inst": 0}
    into = getattr(ops, "count_files_into", None)
    f0 = f1 = 0
    proto = bases.new_empty((0, 2) if k > 32 else (0,), dtype=torch.int64)       # (the shape of a key tensor: {lo, hi} rows for k > 32)
    for bi, (sa, sb) in enumerate(slices):
        # Histogram first, partition after the plan (round 5): the partition takes explicit bucket starts, so the buckets this rank
        # OWNS go straight to their place in its inbox -- no copy to itself (35 ms of a 10 Gbp rank at one GPU, 1/world of the
        # exchange on a node) -- and the others into a compact send area in front of it.
        local_counts, tok = ops.histogram_keep(bases[sa:sb], k, mode, bits)
        mark("histogram")
        local_counts = np.asarray(local_counts).astype(np.int64)
        # the senders' fifteen-bit histograms, summed: every owner's first grouping digit (HIP operators; buckets of up to 8 bits)
        fine = tok[7] if (isinstance(tok, tuple) and len(tok) > 7) else None
        if fine is not None and world > 1:
            _all_reduce(fine, dist.ReduceOp.SUM, group)
        # one small all-gather gives every rank the same [rank][file] histogram -> same cut points
        fc = torch.from_numpy(local_counts).to(bases.device)
        all_counts = [torch.empty_like(fc) for _ in range(world)]
        _all_gather(all_counts, fc, group)
        per_rank = torch.stack(all_counts).cpu().numpy()                             # [world][2^bits]
        if cuts is None:
            cuts = balanced_file_ranges(per_rank.sum(axis=0), world)

        f0, f1 = cuts[rank], cuts[rank + 1]
        if db is not None and sink is None:
            sink = ops.open_sink(db["path"], k, db["w_prefix"], db.get("label_size", 0), db.get("label", 0), rank, world,
                                 db.get("host_threads", 8))
            blocks_per_bucket = 1 << (db["w_prefix"] - bits)
            if n_batches > 1:
                runs = ops.open_runs(k, db["w_prefix"], runs_budget if runs_budget is not None else 0xFFFFFFFFFFFFFFFF)
        file_total = per_rank[:, f0:f1].sum(axis=0)                                  # keys per owned file
        file_off = np.concatenate([[0], np.cumsum(file_total)]).astype(np.int64)
        send_counts = local_counts.copy()
        send_counts[f0:f1] = 0                                                       # what leaves this rank
        n_send = int(send_counts.sum())
        local_off = np.concatenate([[0], np.cumsum(send_counts)]).astype(np.int64)   # send area: the other ranks' buckets, ascending
        starts = local_off[:-1].copy()
        for f in range(f0, f1):                                                      # own buckets: behind the earlier ranks' pieces in the inbox
            starts[f] = n_send + int(file_off[f - f0]) + int(per_rank[:rank, f].sum())
        buf = ops.empty_keys(n_send + int(file_total.sum()), proto)
        mark("plan")
        ops.partition_into(tok, starts, buf)
        del tok
        keys, inbox = buf[:n_send], buf[n_send:]
        mark("partition")

        # The exchange runs in waves: wave i carries, for every rank, the pieces of the i-th group of `bpw` buckets of that
        # rank's range.  While wave i is on the links the owner counts the buckets of wave i-1 (the same grouping passes +
        # LDS finish a single-GPU count runs after its partition), so only the first wave is exposed.  Every rank
        # derives the same wave plan and segment sizes from the gathered histogram: no further collective is needed.
        # About 16 waves per rank: fewer would expose more of the exchange, more pay the per-call host overhead more often.
        most = max(cuts[r + 1] - cuts[r] for r in range(world))
        bpw = max(1, -(-most // 16))
        n_waves = -(-most // bpw)

        def post(i):
            sends, recvs = [], []
            longest = 0
            for dst in range(world):
                for f in range(cuts[dst] + i * bpw, min(cuts[dst + 1], cuts[dst] + (i + 1) * bpw)):
                    if dst != rank:                                                  # (the own pieces are in the inbox already)
                        sends.append((dst, keys[int(local_off[f]):int(local_off[f + 1])]))
                    longest = max(longest, int(per_rank[:, f].max()))
            for f in range(f0 + i * bpw, min(f1, f0 + (i + 1) * bpw)):
                for src in range(world):
                    if src == rank:
                        continue
                    a = int(file_off[f - f0] + per_rank[:src, f].sum())
                    recvs.append((src, inbox[a:a + int(per_rank[src, f])]))
            return exchange_segments(sends, recvs, keys.device, group, max_rows=longest, wait=False)

        def count_file(i):
            nonlocal n_local_distinct
            lo, hi = f0 + i * bpw, min(f1, f0 + (i + 1) * bpw)
            if lo >= hi:
                return
            if file_total[lo - f0:hi - f0].sum() == 0:
                if sink is not None and runs is None:             # the range still gets its (empty) blocks
                    e = ops.empty_keys(0, inbox)
                    sink.write(e, torch.empty(0, dtype=torch.int32, device=e.device), lo * blocks_per_bucket, hi * blocks_per_bucket)
                return
            bc = np.zeros(nbk, dtype=np.uint64)
            bc[lo:hi] = file_total[lo - f0:hi - f0]
            seg = inbox[int(file_off[lo - f0]):int(file_off[hi - f0])]
            if into is not None and runs is None and res["k"] is not None:
                part = into(seg, bc, k, mode, (res["k"][res["at"]:], res["c"][res["at"]:]), fine)
                if part[0].data_ptr() != res["k"][res["at"]:].data_ptr():        # did not fit: grow, then take the copy
                    need = res["at"] + int(part[0].shape[0])
                    left = int(file_total[hi - f0:].sum())
                    cap = need + int(left * 1.25 * need / max(1, res["inst"] + int(seg.shape[0]))) + 4096
                    nk, nc = ops.empty_keys(cap, proto), torch.empty(cap, dtype=torch.int32, device=seg.device)
                    nk[:res["at"]] = res["k"][:res["at"]]; nc[:res["at"]] = res["c"][:res["at"]]
                    nk[res["at"]:need] = part[0]; nc[res["at"]:need] = part[1]
                    res["k"], res["c"] = nk, nc
                    part = (nk[res["at"]:need], nc[res["at"]:need])
                res["at"] += int(part[0].shape[0]); res["inst"] += int(seg.shape[0])
            else:
                part = ops.count_files(seg, bc, k, mode, fine=fine) if fine is not None else ops.count_files(seg, bc, k, mode)
                if into is not None and runs is None:
                    # the first counted wave sizes the result: its distinct / instances ratio over everything this rank owns
                    total = int(file_total.sum())
                    d0, n0 = int(part[0].shape[0]), max(1, int(seg.shape[0]))
                    cap = d0 + int((total - n0) * 1.06 * d0 / n0) + 4096
                    res["k"], res["c"] = ops.empty_keys(cap, proto), torch.empty(cap, dtype=torch.int32, device=seg.device)
                    res["k"][:d0] = part[0]; res["c"][:d0] = part[1]
                    part = (res["k"][:d0], res["c"][:d0])
                    res["at"], res["inst"] = d0, n0
            if runs is not None:                                  # batched: parked (copied) until the last batch is counted
                runs.add(part[0], part[1])
                return
            n_local_distinct += int(part[0].shape[0])
            if sink is not None:
                sink.write(part[0], part[1], lo * blocks_per_bucket, hi * blocks_per_bucket)
                if not keep_result and hasattr(sink, "release_done"):
                    sink.release_done()                               # the tensors of ranges that have left the device
            if keep_result or sink is None:
                parts.append(part)

        t_first = _time.perf_counter()
        for i in range(n_waves + 1):
            reqs = post(i) if i < n_waves else []
            if i >= 1:
                count_file(i - 1)
            for r in reqs:
                r.wait()
            if i == 0 and prof:
                torch.cuda.synchronize()
                stage_s["first_wave_exposed"] = stage_s.get("first_wave_exposed", 0.0) + _time.perf_counter() - t_first
        del keys, inbox, buf
        mark("exchange+count")
    if runs is not None:
        runs.write(sink, f0 * blocks_per_bucket, f1 * blocks_per_bucket)
        rp = runs.profile()
        n_local_distinct = rp["n_merged"]
        db["runs_profile"] = rp
        runs.close()
        mark("merge runs")
    if res["k"] is not None and parts:                     # the waves were counted into ONE result: nothing to concatenate
        uniq, cnts = res["k"][:res["at"]], res["c"][:res["at"]]
    elif parts:
        uniq = torch.cat([p[0] for p in parts])
        cnts = torch.cat([p[1] for p in parts])
    else:
        like = bases.new_empty(0, dtype=torch.int64)
        uniq = like.view(0, 2) if k > 32 else like
        cnts = torch.empty(0, dtype=torch.int32, device=uniq.device)
    mark("concat")
    if sink is not None:
        db["profile"] = sink.close()                      # waits for this rank's files
        db["n_distinct_local"] = n_local_distinct
        db["n_batches"] = n_batches
        if world > 1:
            dist.barrier(group=group)
            t_st = _time.perf_counter()
            if rank == 0:
                from . import db as _db
                _db.merge_parts(db["path"], world)
            dist.barrier(group=group)
            stage_s["stitch"] = _time.perf_counter() - t_st
        mark("database")
    if prof:
        for i, (n, t) in enumerate(marks[1:]):
            stage_s[n] = stage_s.get(n, 0.0) + (t - marks[i][1])
        if db is not None:
            db["stage_s"] = stage_s
        if rank == 0:
            print("[shard profile] " + "  ".join("%s %.1f ms" % (n, v * 1e3) for n, v in stage_s.items()), file=sys.stderr, flush=True)
    return uniq, cnts, (f0, f1, bits)
