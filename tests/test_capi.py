"""CPU tests of the drop-in boundary: the C-ABI library builds (hipcc
cross-compiles gfx950 without a GPU), loads, exports every symbol the header
declares, and its host-side logic (configureCounting restatement, argument
checking, error behaviour) matches the oracle.  No kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, "include", h) for h in ("meryl_gpu_count.h", "meryl_db.h", "meryl_seq.h", "meryl_lookup.h")]


def declared_functions():
    src = "".join(open(h).read() for h in HEADERS)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(m(?:gc|db|sr)_[a-z0-9_]+)\s*\(", src)
    return sorted(set(n for n in names if not n.endswith("_cb")))


def test_header_symbols_exported(native_lib):
    from meryl_amd import capi
    names = declared_functions()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(native_lib, n)]
    assert not missing, missing
    # the binding's own list covers the header
    assert set(names) <= set(capi.SYMBOLS), set(names) - set(capi.SYMBOLS)


def test_library_is_in_tree_and_gfx950(native_lib):
    from meryl_amd import capi
    path = capi.library_path()
    assert path.startswith(ROOT) and os.path.exists(path)
    blob = open(path, "rb").read()
    assert b"gfx950" in blob, "code object for gfx950 missing from the shared library"
    assert native_lib.mgc_version() >= 1


def test_configure_matches_oracle(native_lib, oracle_lib):
    from meryl_amd import capi
    rng = np.random.default_rng(5)
    GB = 1 << 30
    combos = [(21, 4641652, 4 * GB), (21, 10_000_000_000, 64 * GB), (21, 90_000_000_000, 256 * GB),
              (31, 90_000_000_000, 256 * GB), (51, 150_000_000_000, 256 * GB), (22, 5_000_000, 1 * GB)]
    for _ in range(300):
        k = int(rng.integers(6, 65))
        n = int(10 ** rng.uniform(3, 11.5))
        mem = int(2 ** rng.uniform(28, 41))
        combos.append((k, n, mem))
    for k, n, mem in combos:
        c = capi.configure(k, n, mem)
        o = oracle_lib.configure_counting(k, n, mem)
        got = (c.use_simple, c.w_prefix, c.n_prefix, c.w_data, c.n_batches, c.memory_used)
        want = (o["use_simple"], o["w_prefix"], o["n_prefix"], o["w_data"], o["n_batches"], o["memory_used"])
        assert got == want, (k, n, mem, got, want)


def test_configured_line_format(native_lib):
    # the Canu-parsed line, src/meryl/merylOp-count.C:398-401
    from meryl_amd import capi
    c = capi.configure(21, 10_000_000_000, 64 << 30)
    line = capi.configured_line(c)
    assert re.fullmatch(r"Configured complex mode for \d+\.\d{3} GB memory per batch, and up to \d+ batch(es)?\.", line)


def test_error_codes_not_exit(native_lib):
    from meryl_amd import capi
    c = capi.CountConfig()
    c.k = 0                                   # the reference exits(1) here (merylOp-count.C:311-312)
    assert native_lib.mgc_configure_counting(ctypes.byref(c)) == capi.MGC_EINVAL
    assert b"Kmer size" in native_lib.mgc_last_error(None)
    assert native_lib.mgc_configure_counting(None) == capi.MGC_EINVAL
    c.k = 21
    c.mode = 7
    assert native_lib.mgc_configure_counting(ctypes.byref(c)) == capi.MGC_EINVAL
    # unconfigured config is refused before any device work
    c2 = capi.CountConfig()
    c2.k = 21
    assert not native_lib.mgc_open(ctypes.byref(c2), -1)
    assert b"mgc_configure_counting" in native_lib.mgc_last_error(None)
    # count-suffix= is validated before any device work too (merylOp.H:139-147, merylOp-count.C:142-145)
    for k, sfx, msg in ((21, "ACN", b"not ACGT"), (4, "AC", b"needs k >="), (31, "ACGT", b"simple mode"), (21, "A" * 33, b"count_suffix must hold")):
        c3 = capi.configure(k, 1000, 1 << 30)
        c3.count_suffix_length = len(sfx)
        c3.count_suffix = sfx[:35].encode()
        assert not native_lib.mgc_open(ctypes.byref(c3), -1) and msg in native_lib.mgc_last_error(None), (k, sfx, native_lib.mgc_last_error(None))
    c4 = capi.configure(21, 1000, 1 << 30, count_suffix="ac")
    assert c4.use_simple == 1 and c4.count_suffix_length == 2
    # argument checks of the device operators happen before any launch
    assert native_lib.mgc_dev_kmer_histogram(None, 10, 0, 0, 6, None, None, 0, None) == capi.MGC_EINVAL
    assert native_lib.mgc_dev_kmer_histogram(None, 10, 65, 0, 6, None, None, 0, None) == capi.MGC_EINVAL
    assert native_lib.mgc_dev_kmer_histogram(None, 10, 40, 0, 6, None, None, 0, None) == capi.MGC_EINVAL   # NULL bases
    ia = ctypes.c_int(0)
    assert native_lib.mgc_dev_radix_sort(None, None, 5, 1, 10, 4, None, 0, ctypes.byref(ia), None) == capi.MGC_EINVAL
    assert native_lib.mgc_dev_radix_sort(None, None, 5, 1, 0, 65, None, 0, ctypes.byref(ia), None) == capi.MGC_EINVAL
    assert native_lib.mgc_dev_radix_sort(None, None, 5, 3, 0, 64, None, 0, ctypes.byref(ia), None) == capi.MGC_EINVAL
    assert native_lib.mgc_dev_radix_sort(None, None, 0, 1, 0, 42, None, 0, ctypes.byref(ia), None) == capi.MGC_OK
    assert native_lib.mgc_dev_radix_sort(None, None, 0, 2, 0, 128, None, 0, ctypes.byref(ia), None) == capi.MGC_OK


def test_node_count_argument_checks_need_no_device(native_lib):
    """mgc_count_node / mgc_count_node_staged / mgc_staged_bases / mgc_node_plan refuse bad arguments with a code and a message
    before any device work (this box has no GPU: anything that got past the checks would fail differently)."""
    from meryl_amd import capi
    L = native_lib
    cfg = capi.configure(21, 1000, 1 << 30)
    one = (ctypes.c_void_p * 1)(None)
    n1 = (ctypes.c_uint64 * 1)(0)
    assert L.mgc_count_node(None, 1, None, one, n1, b"/tmp/x", 1, None) == capi.MGC_EINVAL
    assert L.mgc_count_node(ctypes.byref(cfg), 0, None, one, n1, b"/tmp/x", 1, None) == capi.MGC_EINVAL
    assert L.mgc_count_node(ctypes.byref(cfg), 1, None, one, n1, None, 1, None) == capi.MGC_EINVAL
    sfx = capi.configure(21, 1000, 1 << 30, count_suffix="ACG")
    assert L.mgc_count_node(ctypes.byref(sfx), 1, None, one, n1, b"/tmp/x", 1, None) == capi.MGC_EINVAL
    assert b"count-suffix" in L.mgc_last_error(None)
    raw = capi.CountConfig()
    raw.k = 21
    assert L.mgc_count_node(ctypes.byref(raw), 1, None, one, n1, b"/tmp/x", 1, None) == capi.MGC_EINVAL
    assert b"mgc_configure_counting" in L.mgc_last_error(None)
    assert L.mgc_count_node_staged(None, 2, None, b"/tmp/x", 1, None) == capi.MGC_EINVAL
    p = ctypes.c_void_p()
    n = ctypes.c_uint64(0)
    assert L.mgc_staged_bases(None, ctypes.byref(p), ctypes.byref(n)) == capi.MGC_EINVAL
    bits = ctypes.c_uint32(0)
    assert L.mgc_node_plan(0, 21, 0, 18, ctypes.byref(bits), None, None) == capi.MGC_EINVAL
    assert L.mgc_node_plan(2, 0, 0, 18, ctypes.byref(bits), None, None) == capi.MGC_EINVAL
    assert L.mgc_node_plan(100, 21, 0, 6, ctypes.byref(bits), None, None) == capi.MGC_EINVAL      # 64 ranges, 100 ranks
    assert L.mgc_node_plan(8, 21, 10 ** 10, 18, ctypes.byref(bits), None, None) == 0 and bits.value == 9


def test_workspace_sizes_monotone(native_lib):
    prev = 0
    for n in (0, 1, 4096, 10**6, 10**9):
        w = native_lib.mgc_dev_sort_workspace_bytes(n)
        assert w >= prev
        prev = w
    assert native_lib.mgc_dev_partition_workspace_bytes(6) == 16384 * 64 * 8      # rows for up to 16384 virtual workgroups (MGC_PART_VGRID)
    assert native_lib.mgc_dev_partition_workspace_bytes(10) == 2048 * 1024 * 8
    assert native_lib.mgc_dev_rle_workspace_bytes(0) > 0


def test_product_never_imports_oracle():
    # the oracle is a checker: nothing under meryl_amd/ or include/ may reference it
    bad = []
    for base in ("meryl_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M) or "liboracle" in txt or "oracle.h" in txt:
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_node_plan_equals_the_rccl_launchers_plan(native_lib):
    """mgc_count_node (one process, peer copies) and count_sharded (one process per GPU, RCCL) must route alike: the C++
    plan -- bucket granularity and the balanced contiguous bucket ranges -- against the Python launcher's on random histograms,
    skewed ones (everything in one bucket, empty tails) and the refusal when ranks outnumber the routable ranges."""
    import ctypes
    from meryl_amd import count
    rng = np.random.default_rng(5)
    L = native_lib
    for trial in range(300):
        n = int(rng.integers(1, 65))
        k = int(rng.choice([4, 8, 16, 21, 31, 51]))
        w_prefix = int(rng.integers(6, min(2 * k, 24) + 1))
        max_bases = int(rng.choice([0, 10 ** 6, 10 ** 10, 3 * 10 ** 10, 10 ** 11]))
        bits = ctypes.c_uint32(0)
        want_bits = count.shard_bucket_bits(n, k, max_bases, w_prefix)
        rc = L.mgc_node_plan(n, k, max_bases, w_prefix, ctypes.byref(bits), None, None)
        if (1 << want_bits) < n:
            assert rc != 0
            continue
        assert rc == 0 and bits.value == want_bits, (n, k, max_bases, w_prefix, bits.value, want_bits)
        nb = 1 << want_bits
        shape = trial % 4
        if shape == 0:
            tot = rng.integers(0, 10 ** 9, nb).astype(np.uint64)
        elif shape == 1:
            tot = np.zeros(nb, np.uint64); tot[int(rng.integers(0, nb))] = 10 ** 12
        elif shape == 2:
            tot = (rng.pareto(1.2, nb) * 1e6).astype(np.uint64)
        else:
            tot = np.zeros(nb, np.uint64); tot[:nb // 3] = rng.integers(1, 10 ** 6, nb // 3).astype(np.uint64)
        cuts = (ctypes.c_uint32 * (n + 1))()
        assert L.mgc_node_plan(n, k, max_bases, w_prefix, ctypes.byref(bits), tot.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), cuts) == 0
        assert list(cuts) == [int(c) for c in count.balanced_file_ranges(tot, n)]


def test_staged_slices_with_k_minus_1_overlap_lose_and_double_nothing(oracle_lib):
    """The cutting rule of mgc_count_node_staged (slice r = [cut_r - (k-1), cut_{r+1}), cuts anywhere in the stream -- inside reads,
    inside '.' runs, within k-1 bases of either end), restated in Python and held against the oracle: the k-mer counts of the
    slices add up to the counts of the whole stream, for every k and rank count tried."""
    import collections
    rng = np.random.default_rng(11)
    for trial in range(40):
        k = int(rng.choice([3, 5, 8, 13, 21, 31]))
        parts = []
        for _ in range(int(rng.integers(1, 30))):
            parts.append("".join("ACGTN"[i] for i in rng.choice(5, int(rng.integers(0, 3 * k)), p=[.24, .24, .24, .24, .04])))
            parts.append("." * int(rng.integers(1, 4)))
        stream = "".join(parts)
        n = len(stream)
        for n_ranks in (1, 2, 3, 7, 16):
            total = collections.Counter()
            for r in range(n_ranks):
                cut, end = n * r // n_ranks, n * (r + 1) // n_ranks
                a = cut - (k - 1) if (r and cut >= k - 1) else 0
                hi, lo, cn, _ = oracle_lib.count_brute(stream[a:end], k)
                for h, l, c in zip(hi, lo, cn):
                    total[(int(h), int(l))] += int(c)
            hi, lo, cn, _ = oracle_lib.count_brute(stream, k)
            want = {(int(h), int(l)): int(c) for h, l, c in zip(hi, lo, cn)}
            assert dict(total) == want, (k, n_ranks, stream)


def test_text_record_start_finds_window_cuts(native_lib, tmp_path):
    """mgc_text_record_start (host I/O only): the first record start at or after an offset -- where a rank of a node count
    may begin to read its byte window of a file.  FASTQ quality lines that start with '@' or '+' must not fool it."""
    import ctypes
    import random
    from meryl_amd import capi
    L = capi.lib()
    rnd = random.Random(5)
    recs = []
    for i in range(300):
        n = rnd.randint(1, 90)
        seq = "".join(rnd.choice("ACGTN") for _ in range(n))
        q = "".join(rnd.choice("@+I5>") for _ in range(n))
        recs.append("@r%d\n%s\n+\n%s\n" % (i, seq, q))
    fq = tmp_path / "a.fq"
    fq.write_text("".join(recs))
    starts, p = [], 0
    for r in recs:
        starts.append(p)
        p += len(r)
    size = p
    out = ctypes.c_uint64(0)
    for off in list(range(0, 400)) + [rnd.randrange(size) for _ in range(300)] + [size - 1, size, size + 10]:
        assert L.mgc_text_record_start(str(fq).encode(), 0, off, ctypes.byref(out)) == 0
        want = min([s for s in starts if s >= off] + [size])
        assert out.value == want, (off, out.value, want)
    fa = tmp_path / "a.fa"
    frecs = [">s%d x\n%s\n%s\n" % (i, "ACGT" * rnd.randint(1, 20), "GG" * rnd.randint(1, 9)) for i in range(200)]
    fa.write_text("".join(frecs))
    fstarts, p = [], 0
    for r in frecs:
        fstarts.append(p)
        p += len(r)
    for off in [0, 1, 2, 50, 51] + [rnd.randrange(p) for _ in range(200)] + [p]:
        assert L.mgc_text_record_start(str(fa).encode(), 0, off, ctypes.byref(out)) == 0
        assert out.value == min([s for s in fstarts if s >= off] + [p]), off
    assert L.mgc_text_record_start(str(tmp_path / "missing").encode(), 0, 5, ctypes.byref(out)) != 0
