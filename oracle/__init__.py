"""ctypes loader for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke().  Nothing under meryl_amd/ may import this package.
See oracle/oracle.h for what each function restates (reference file:line).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

CANONICAL, FORWARD, REVERSE = 0, 1, 2


def build(force=False):
    """Compile liboracle.so with the committed Makefile (gcc/g++ only)."""
    srcs = [os.path.join(_HERE, f) for f in ("oracle_count.c", "oracle_port.cpp", "oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Config(ctypes.Structure):
    _fields_ = [
        ("k", ctypes.c_uint32),
        ("n_kmers_estimate", ctypes.c_uint64),
        ("memory_allowed", ctypes.c_uint64),
        ("count_suffix_length", ctypes.c_uint32),
        ("page_size", ctypes.c_uint32),
        ("sizeof_count_array", ctypes.c_uint32),
        ("use_simple", ctypes.c_int),
        ("w_prefix", ctypes.c_uint32),
        ("n_prefix", ctypes.c_uint64),
        ("w_data", ctypes.c_uint32),
        ("n_batches", ctypes.c_uint32),
        ("memory_used", ctypes.c_uint64),
        ("memory_simple", ctypes.c_uint64),
        ("memory_complex", ctypes.c_uint64),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = ctypes.CDLL(_LIB_PATH)
    u64p = ctypes.POINTER(ctypes.c_uint64)
    u32p = ctypes.POINTER(ctypes.c_uint32)
    L.orc_base_code.argtypes = [ctypes.c_char]
    L.orc_base_code.restype = ctypes.c_int
    L.orc_enumerate_kmers.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
    L.orc_enumerate_kmers.restype = ctypes.c_uint64
    L.orc_count_brute.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int,
                                  ctypes.POINTER(u64p), ctypes.POINTER(u64p), ctypes.POINTER(u32p),
                                  u64p, u64p]
    L.orc_count_brute.restype = ctypes.c_int
    L.orc_count_brute_suffix.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_char_p,
                                         ctypes.POINTER(u64p), ctypes.POINTER(u64p), ctypes.POINTER(u32p),
                                         u64p, u64p]
    L.orc_count_brute_suffix.restype = ctypes.c_int
    L.orc_count_threaded_collect.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int,
                                             ctypes.c_uint32, ctypes.c_int,
                                             ctypes.POINTER(u64p), ctypes.POINTER(u64p), ctypes.POINTER(u32p),
                                             u64p, u64p]
    L.orc_count_threaded_collect.restype = ctypes.c_int
    L.orc_count_threaded.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int,
                                     ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                     u64p, u64p]
    L.orc_count_threaded.restype = ctypes.c_int
    L.orc_count_threaded_digest.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int,
                                            ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, u64p, u64p]
    L.orc_count_threaded_digest.restype = ctypes.c_int
    L.orc_count_threaded_digest_collect.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int,
                                                    ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, u64p, u64p, ctypes.c_uint64,
                                                    ctypes.c_void_p, ctypes.POINTER(u64p), ctypes.POINTER(u64p), ctypes.POINTER(u32p)]
    L.orc_count_threaded_digest_collect.restype = ctypes.c_int
    L.orc_free.argtypes = [ctypes.c_void_p]
    L.orc_free.restype = None
    L.orc_kmer_to_string.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_char_p]
    L.orc_kmer_to_string.restype = None
    L.orc_homopoly_compress.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_char]
    L.orc_homopoly_compress.restype = ctypes.c_uint64
    L.orc_configure_counting.argtypes = [ctypes.POINTER(_Config)]
    L.orc_configure_counting.restype = ctypes.c_int
    L.orc_synth_reads.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                  ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    L.orc_synth_reads.restype = ctypes.c_uint64
    L.orc_synth_reads_ex.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                     ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                     ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    L.orc_synth_reads_ex.restype = ctypes.c_uint64
    _lib = L
    return L


def _as_bytes(bases):
    if isinstance(bases, str):
        return bases.encode("ascii")
    if isinstance(bases, np.ndarray):
        return bases.tobytes()
    return bytes(bases)


def _take(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    arr = np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)
    lib().orc_free(ctypes.cast(ptr, ctypes.c_void_p))
    return arr


def enumerate_kmers(bases, k, mode=CANONICAL):
    """All k-mer instances in input order -> (hi, lo) uint64 arrays."""
    b = _as_bytes(bases)
    L = lib()
    n = L.orc_enumerate_kmers(b, len(b), k, mode, None, None, 0)
    hi = np.zeros(n, dtype=np.uint64)
    lo = np.zeros(n, dtype=np.uint64)
    if n:
        L.orc_enumerate_kmers(b, len(b), k, mode, hi.ctypes.data, lo.ctypes.data, n)
    return hi, lo


def count_brute(bases, k, mode=CANONICAL, count_suffix=""):
    """Brute-force count -> (keys_hi, keys_lo, counts, n_instances); count_suffix: only k-mers ending in these bases."""
    b = _as_bytes(bases)
    L = lib()
    hi = ctypes.POINTER(ctypes.c_uint64)()
    lo = ctypes.POINTER(ctypes.c_uint64)()
    cn = ctypes.POINTER(ctypes.c_uint32)()
    nd = ctypes.c_uint64(0)
    ni = ctypes.c_uint64(0)
    if count_suffix:
        rc = L.orc_count_brute_suffix(b, len(b), k, mode, count_suffix.encode("ascii"), ctypes.byref(hi), ctypes.byref(lo),
                                      ctypes.byref(cn), ctypes.byref(nd), ctypes.byref(ni))
    else:
        rc = L.orc_count_brute(b, len(b), k, mode, ctypes.byref(hi), ctypes.byref(lo), ctypes.byref(cn),
                               ctypes.byref(nd), ctypes.byref(ni))
    if rc != 0:
        raise RuntimeError("orc_count_brute failed rc=%d" % rc)
    # hi/lo were malloc'd with n_instances entries; only the first nd are meaningful
    return (_take(hi, nd.value, np.uint64), _take(lo, nd.value, np.uint64),
            _take(cn, nd.value, np.uint32), ni.value)


def count_threaded(bases, k, w_prefix, mode=CANONICAL, threads=0):
    """Reference-algorithm restatement -> (keys_hi, keys_lo, counts, n_instances)."""
    b = _as_bytes(bases)
    L = lib()
    hi = ctypes.POINTER(ctypes.c_uint64)()
    lo = ctypes.POINTER(ctypes.c_uint64)()
    cn = ctypes.POINTER(ctypes.c_uint32)()
    nd = ctypes.c_uint64(0)
    ni = ctypes.c_uint64(0)
    rc = L.orc_count_threaded_collect(b, len(b), k, mode, w_prefix, threads,
                                      ctypes.byref(hi), ctypes.byref(lo), ctypes.byref(cn),
                                      ctypes.byref(nd), ctypes.byref(ni))
    if rc != 0:
        raise RuntimeError("orc_count_threaded_collect failed rc=%d" % rc)
    return (_take(hi, nd.value, np.uint64), _take(lo, nd.value, np.uint64),
            _take(cn, nd.value, np.uint32), ni.value)


def time_threaded(bases, k, w_prefix, mode=CANONICAL, threads=0):
    """Run the port with no collection (pure timing) -> (n_distinct, n_instances)."""
    b = _as_bytes(bases)
    nd = ctypes.c_uint64(0)
    ni = ctypes.c_uint64(0)
    rc = lib().orc_count_threaded(b, len(b), k, mode, w_prefix, threads, None, None,
                                  ctypes.byref(nd), ctypes.byref(ni))
    if rc != 0:
        raise RuntimeError("orc_count_threaded failed rc=%d" % rc)
    return nd.value, ni.value


DIGEST_C1, DIGEST_C2, DIGEST_C3 = 0x9E3779B97F4A7C15, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9


def digest_threaded(bases, k, w_prefix, mode=CANONICAL, threads=0):
    """The port's result as per-file digests -> (uint64[64, 4], n_distinct, n_instances).  `bases` may be a uint8 numpy
    array (used in place: no 10 GB copy).  Columns: n_distinct, sum counts, sum (x*C1)*count,
    sum ((x^C2)*(x|1))*count, x = lo ^ hi*C3, arithmetic mod 2^64."""
    if isinstance(bases, np.ndarray) and bases.dtype == np.uint8 and bases.flags["C_CONTIGUOUS"]:
        ptr, n, keep = bases.ctypes.data, bases.size, bases
    else:
        keep = _as_bytes(bases)
        ptr, n = ctypes.cast(ctypes.c_char_p(keep), ctypes.c_void_p), len(keep)
    out = np.zeros((64, 4), dtype=np.uint64)
    nd = ctypes.c_uint64(0)
    ni = ctypes.c_uint64(0)
    rc = lib().orc_count_threaded_digest(ptr, n, k, mode, w_prefix, threads, out.ctypes.data, ctypes.byref(nd), ctypes.byref(ni))
    del keep
    if rc != 0:
        raise RuntimeError("orc_count_threaded_digest failed rc=%d" % rc)
    return out, nd.value, ni.value


def digest_collect_threaded(bases, k, w_prefix, files, mode=CANONICAL, threads=0):
    """digest_threaded plus the (k-mer, count) streams of the files in `files` themselves ->
    (uint64[64, 4], n_distinct, n_instances, {file: (keys_hi, keys_lo, counts)})."""
    if isinstance(bases, np.ndarray) and bases.dtype == np.uint8 and bases.flags["C_CONTIGUOUS"]:
        ptr, n, keep = bases.ctypes.data, bases.size, bases
    else:
        keep = _as_bytes(bases)
        ptr, n = ctypes.cast(ctypes.c_char_p(keep), ctypes.c_void_p), len(keep)
    out = np.zeros((64, 4), dtype=np.uint64)
    fs = np.zeros(65, dtype=np.uint64)
    nd = ctypes.c_uint64(0)
    ni = ctypes.c_uint64(0)
    hi = ctypes.POINTER(ctypes.c_uint64)()
    lo = ctypes.POINTER(ctypes.c_uint64)()
    cn = ctypes.POINTER(ctypes.c_uint32)()
    mask = 0
    for f in files:
        mask |= 1 << int(f)
    rc = lib().orc_count_threaded_digest_collect(ptr, n, k, mode, w_prefix, threads, out.ctypes.data, ctypes.byref(nd), ctypes.byref(ni),
                                                 mask, fs.ctypes.data, ctypes.byref(hi), ctypes.byref(lo), ctypes.byref(cn))
    del keep
    if rc != 0:
        raise RuntimeError("orc_count_threaded_digest_collect failed rc=%d" % rc)
    total = int(fs[64])
    ahi, alo, acn = _take(hi, total, np.uint64), _take(lo, total, np.uint64), _take(cn, total, np.uint32)
    got = {}
    for f in files:
        a, b = int(fs[int(f)]), int(fs[int(f) + 1])
        got[int(f)] = (ahi[a:b], alo[a:b], acn[a:b])
    return out, nd.value, ni.value, got


def digest_arrays(lo, hi, counts, k):
    """The same digests from (lo, hi, counts) numpy arrays (ascending keys): the checker's own reference of the sums."""
    M = (1 << 64) - 1
    out = np.zeros((64, 4), dtype=np.uint64)
    lo = lo.astype(np.uint64); hi = hi.astype(np.uint64); c = counts.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = lo ^ (hi * np.uint64(DIGEST_C3))
        a = (x * np.uint64(DIGEST_C1)) * c
        b = ((x ^ np.uint64(DIGEST_C2)) * (x | np.uint64(1))) * c
        if 2 * k - 6 >= 64:
            f = (hi >> np.uint64(2 * k - 6 - 64)).astype(np.int64)
        else:
            f = ((lo >> np.uint64(2 * k - 6)) | ((hi << np.uint64(64 - (2 * k - 6))) if 2 * k > 64 else np.uint64(0))).astype(np.int64)
        for ff in range(64):
            m = f == ff
            out[ff, 0] = int(m.sum())
            out[ff, 1] = int(c[m].sum(dtype=np.uint64)) & M
            out[ff, 2] = int(a[m].sum(dtype=np.uint64)) & M
            out[ff, 3] = int(b[m].sum(dtype=np.uint64)) & M
    return out


def kmer_to_string(hi, lo, k):
    buf = ctypes.create_string_buffer(k + 1)
    lib().orc_kmer_to_string(int(hi), int(lo), k, buf)
    return buf.value.decode("ascii")


def homopoly_compress(bases, last_byte=b"\0"):
    b = _as_bytes(bases)
    out = ctypes.create_string_buffer(len(b) + 1)
    n = lib().orc_homopoly_compress(b, len(b), out, last_byte)
    return out.raw[:n]


def compress_stream(bases, chunk=0):
    """`compress` applied the way the reference does (merylInput.C:245-271): every sequence of the
    '.'-separated stream is homopolymer-compressed on its own, optionally in chunks of `chunk`
    bytes with the _lastByte carry between chunks of one sequence."""
    b = _as_bytes(bases)
    out = []
    for seq in b.split(b"."):
        if chunk <= 0:
            out.append(homopoly_compress(seq))
        else:
            last, parts = b"\0", []
            for i in range(0, len(seq), chunk):
                piece = seq[i:i + chunk]
                parts.append(homopoly_compress(piece, last))
                last = piece[-1:] if piece else b"\0"
            out.append(b"".join(parts))
    return b".".join(out)


def configure_counting(k, n_kmers_estimate, memory_bytes, count_suffix_length=0,
                       page_size=4096, sizeof_count_array=3232):
    c = _Config()
    c.k = k
    c.n_kmers_estimate = n_kmers_estimate
    c.memory_allowed = memory_bytes
    c.count_suffix_length = count_suffix_length
    c.page_size = page_size
    c.sizeof_count_array = sizeof_count_array
    rc = lib().orc_configure_counting(ctypes.byref(c))
    if rc != 0:
        raise RuntimeError("orc_configure_counting failed rc=%d" % rc)
    return {f: getattr(c, f) for f, _ in _Config._fields_}


def synth_reads(seed, genome_len, first_read, n_reads, read_len=150, sub_rate_ppm=5000, n_rate_ppm=100,
                repeat_ppm=0, repeat_unit=300, repeat_families=1000):
    out = np.zeros(n_reads * (read_len + 1), dtype=np.uint8)
    n = lib().orc_synth_reads_ex(seed, genome_len, first_read, n_reads, read_len, sub_rate_ppm, n_rate_ppm,
                                 repeat_ppm, repeat_unit, repeat_families, out.ctypes.data)
    assert n == out.size
    return out
