"""CPU tests of the database writer/reader (host C++ behind include/meryl_db.h).

What is checked: the structural facts the reference tree pins (64+64+1 files,
names, master-index parameters, block header fields, uBits/bBits rule of
documentation/source/usage.rst:13-45) via an INDEPENDENT pure-Python parser of
the bytes, and a write -> read round trip against the oracle's counts.  The
byte layout itself is a restatement of the absent meryl-utility writer and is
PARITY UNPINNED (no reference-made database exists to diff against)."""
import os
import struct

import numpy as np
import pytest

from helpers import random_reads


class Bits:
    """independent MSB-first bit reader over little-endian uint64 words"""

    def __init__(self, words):
        self.w = [int(x) for x in words]
        self.pos = 0

    def get(self, n):
        v = 0
        for _ in range(n):
            word, off = divmod(self.pos, 64)
            v = (v << 1) | ((self.w[word] >> (63 - off)) & 1)
            self.pos += 1
        return v

    def unary(self):
        v = 0
        while self.get(1) == 0:
            v += 1
        return v


def read_stuffed(f):
    lenmax, nblocks, nmax = struct.unpack("<QII", f.read(16))
    bgn = struct.unpack("<%dQ" % nblocks, f.read(8 * nblocks))
    ln = struct.unpack("<%dQ" % nblocks, f.read(8 * nblocks))
    words = []
    for i in range(nblocks):
        nw, nalloc = struct.unpack("<QQ", f.read(16))
        assert nw == (ln[i] + 63) // 64 and nalloc == lenmax // 64
        words += list(struct.unpack("<%dQ" % nw, f.read(8 * nw)))
    return words, sum(ln)


def make_db(tmp_path, oracle_lib, k, w_prefix, bases, name="db"):
    from meryl_amd import db
    hi, lo, cn, ni = oracle_lib.count_threaded(bases, k, w_prefix, threads=2)
    keys = [(int(h) << 64) | int(l) for h, l in zip(hi, lo)]
    w_data = 2 * k - w_prefix
    path = str(tmp_path / name)
    w = db.Writer(path, k, w_prefix)
    by_prefix = {}
    for key, c in zip(keys, cn):
        by_prefix.setdefault(key >> w_data, []).append((key & ((1 << w_data) - 1), int(c)))
    for p in range(1 << w_prefix):                       # addBlock for every prefix, empty ones too
        items = by_prefix.get(p, [])
        suf = [s for s, _ in items]
        w.add_block(p, [s & 0xFFFFFFFFFFFFFFFF for s in suf], [c for _, c in items],
                    [s >> 64 for s in suf] if w_data > 64 else None)
    w.close()
    return path, keys, [int(c) for c in cn], ni


@pytest.mark.parametrize("k,wp", [(21, 10), (16, 12), (31, 10), (32, 11), (51, 10), (64, 12), (8, 10)])
def test_roundtrip_and_structure(tmp_path, native_lib, oracle_lib, k, wp):
    from meryl_amd import db
    rng = np.random.default_rng(k)
    bases = random_reads(rng, 150, 40, 300) + ("ACGT" * 50 + ".") * 5
    path, keys, counts, ni = make_db(tmp_path, oracle_lib, k, wp, bases)

    # structure: 64 data + 64 index + master (reference.rst:73-77), binary-digit names
    names = sorted(os.listdir(path))
    assert len(names) == 129 and "merylIndex" in names
    assert "0x000000.merylData" in names and "0x111111.merylIndex" in names

    # master index parameters (usage.rst:15-19), parsed independently
    with open(os.path.join(path, "merylIndex"), "rb") as f:
        words, nbits = read_stuffed(f)
    b = Bits(words)
    m1, m2 = b.get(64), b.get(64)
    assert struct.pack("<Q", m1) == b"merylInd" and struct.pack("<Q", m2)[:6] == b"ex__v."
    assert (b.get(32), b.get(32), b.get(32), b.get(32)) == (wp, 2 * k - wp, 6, wp - 6)
    b.get(32)
    n_unique, n_distinct, n_total, n_pairs = b.get(64), b.get(64), b.get(64), b.get(64)
    assert n_distinct == len(keys) and n_total == ni == sum(counts)
    assert n_unique == sum(1 for c in counts if c == 1)
    hist = {b.get(64): b.get(64) for _ in range(n_pairs)}
    vals, occ = np.unique(np.array(counts), return_counts=True)
    assert hist == {int(v): int(o) for v, o in zip(vals, occ)}

    # first data file: block headers (usage.rst:33-45) parsed independently
    nblocks = 1 << (wp - 6)
    with open(os.path.join(path, "0x000000.merylIndex"), "rb") as f:
        idx = [struct.unpack("<QQQ", f.read(24)) for _ in range(nblocks)]
    w_data = 2 * k - wp
    got = []
    with open(os.path.join(path, "0x000000.merylData"), "rb") as f:
        for bp, pos, n in idx:
            f.seek(pos)
            words, _ = read_stuffed(f)
            b = Bits(words + [0, 0])
            assert struct.pack("<Q", b.get(64)) == b"merylDat" and struct.pack("<Q", b.get(64)) == b"aFile00\n"
            prefix, nk = b.get(64), b.get(64)
            kcode, ub, bb, k1, ccode, c1, c2 = b.get(8), b.get(32), b.get(32), b.get(64), b.get(8), b.get(64), b.get(64)
            assert (prefix, nk) == (bp, n) and kcode == 1 and ccode == 1 and ub + bb == w_data
            assert (1 << ub) >= max(nk, 1) and (ub == 0 or (1 << (ub - 1)) < nk)      # uBits = ceil(log2(nKmers))
            top, sufs = 0, []
            for _ in range(nk):
                top += b.unary()
                sufs.append((top << bb) | b.get(bb))
            cnts = [b.get(32) for _ in range(nk)]
            got += [((prefix << w_data) | s, c) for s, c in zip(sufs, cnts)]
    want = [(key, c) for key, c in zip(keys, counts) if (key >> (2 * k - 6)) == 0]
    assert got == want

    # round trip through the C reader, all 64 files
    r = db.Reader(path)
    assert (r.info.k, r.info.prefix_size, r.info.suffix_size) == (k, wp, 2 * k - wp)
    lo, hi, cn = r.read_all()
    assert [(int(h) << 64) | int(l) for h, l in zip(hi, lo)] == keys
    assert [int(c) for c in cn] == counts
    hv, ho = r.histogram()
    assert {int(v): int(o) for v, o in zip(hv, ho)} == hist
    r.close()


def test_usage_rst_header_example(tmp_path, native_lib):
    # usage.rst:33-37: a block with nKmers=22363 and suffixSize=104 has uBits=15, bBits=89
    from meryl_amd import db
    k, wp = 57, 10
    path = str(tmp_path / "hdr")
    w = db.Writer(path, k, wp)
    n = 22363
    w.add_block(0, np.arange(n, dtype=np.uint64) * np.uint64(977), np.ones(n, dtype=np.uint32), np.zeros(n, dtype=np.uint64))
    w.close()
    with open(os.path.join(path, "0x000000.merylData"), "rb") as f:
        words, _ = read_stuffed(f)
    b = Bits(words)
    b.get(64); b.get(64)
    assert b.get(64) == 0 and b.get(64) == n
    assert b.get(8) == 1 and (b.get(32), b.get(32)) == (15, 89)


def test_writer_rejects_bad_use(tmp_path, native_lib):
    from meryl_amd import db
    with pytest.raises(db.DbError):
        db.Writer(str(tmp_path / "x"), 21, 5)                 # prefix must cover the 6 file bits
    w = db.Writer(str(tmp_path / "y"), 21, 10)
    w.add_block(3, [1, 2], [1, 1])
    with pytest.raises(db.DbError):
        w.add_block(2, [1], [1])                              # ascending prefixes per file
    with pytest.raises(db.DbError):
        w.add_block(1 << 10, [1], [1])                        # prefix out of range
    w.close()
    with pytest.raises(db.DbError):
        db.Reader(str(tmp_path / "nonexistent"))


def _dir_bytes(path):
    return {n: open(os.path.join(path, n), "rb").read() for n in sorted(os.listdir(path))}


def _blocks_of(keys, counts, k, wp):
    w_data = 2 * k - wp
    by_prefix = {}
    for key, c in zip(keys, counts):
        by_prefix.setdefault(key >> w_data, []).append((key & ((1 << w_data) - 1), int(c)))
    return by_prefix, w_data


def _feed(w, by_prefix, w_data, p0, p1, label=0):
    for p in range(p0, p1):
        items = by_prefix.get(p, [])
        suf = [s for s, _ in items]
        w.add_block(p, [s & 0xFFFFFFFFFFFFFFFF for s in suf], [c for _, c in items],
                    [s >> 64 for s in suf] if w_data > 64 else None, label=label)


@pytest.mark.parametrize("k,wp,cuts", [(21, 10, [0, 300, 1024]), (21, 12, [0, 64, 65, 2000, 4096]), (51, 9, [0, 1, 100, 511, 512]),
                                        (16, 8, [0, 256]), (31, 10, [0, 512, 512, 1024])])
def test_part_writers_merge_to_identical_bytes(tmp_path, native_lib, oracle_lib, k, wp, cuts):
    """A database written by several part writers over contiguous prefix ranges (cuts in the middle of files, empty
    ranges, one part only) and stitched by mdb_merge_parts is byte-identical to the single writer's."""
    from meryl_amd import db
    rng = np.random.default_rng(k + wp)
    bases = random_reads(rng, 120, 40, 250)
    path1, keys, counts, _ = make_db(tmp_path, oracle_lib, k, wp, bases, "single")
    by_prefix, w_data = _blocks_of(keys, counts, k, wp)
    pathn = str(tmp_path / "parts")
    n_parts = len(cuts) - 1
    for part in reversed(range(n_parts)):                # closing order does not matter
        w = db.Writer(pathn, k, wp, part=part, n_parts=n_parts)
        _feed(w, by_prefix, w_data, cuts[part], cuts[part + 1])
        w.close()
    if n_parts > 1:
        assert any(".part" in n for n in os.listdir(pathn))
        db.merge_parts(pathn, n_parts)
    a, b = _dir_bytes(path1), _dir_bytes(pathn)
    assert sorted(a) == sorted(b) and len(a) == 129
    for n in a:
        assert a[n] == b[n], n


def test_merge_parts_refuses_missing_or_overlapping_parts(tmp_path, native_lib):
    from meryl_amd import db
    p = str(tmp_path / "bad")
    w = db.Writer(p, 21, 10, part=0, n_parts=2)
    w.add_block(5, [1, 2], [1, 1])
    w.close()
    with pytest.raises(db.DbError):
        db.merge_parts(p, 2)                              # part 1 never written
    w = db.Writer(p, 21, 10, part=1, n_parts=2)
    w.add_block(4, [1], [1])                              # below part 0's last prefix, same file
    w.close()
    with pytest.raises(db.DbError):
        db.merge_parts(p, 2)


def test_labelled_database_roundtrip(tmp_path, native_lib, oracle_lib):
    """meryl2's constant count label (addCountedBlock(..., labels=nullptr, label)): stored per k-mer in label_size bits,
    read back by the reader; label_size 0 writes exactly the unlabelled bytes."""
    from meryl_amd import db
    k, wp = 21, 10
    rng = np.random.default_rng(3)
    bases = random_reads(rng, 80, 40, 200)
    path0, keys, counts, _ = make_db(tmp_path, oracle_lib, k, wp, bases, "plain")
    by_prefix, w_data = _blocks_of(keys, counts, k, wp)
    pl = str(tmp_path / "lab")
    w = db.Writer(pl, k, wp, label_size=7)
    _feed(w, by_prefix, w_data, 0, 1 << wp, label=0x155)   # only the low 7 bits (0x55) are stored
    w.close()
    r = db.Reader(pl)
    assert r.info.label_size == 7
    lo, hi, cn, lb = r.read_all(labels=True)
    assert [int(x) for x in lo] == [kk & 0xFFFFFFFFFFFFFFFF for kk in keys] and [int(c) for c in cn] == counts
    assert np.all(lb == 0x55) and lb.size == len(keys)
    r.close()
    r0 = db.Reader(path0)
    assert r0.info.label_size == 0
    r0.close()
    assert os.path.getsize(os.path.join(pl, "0x000000.merylData")) > os.path.getsize(os.path.join(path0, "0x000000.merylData"))


def test_reader_rejects_truncated_and_corrupt_files(tmp_path, native_lib, oracle_lib):
    """The reader trusts nothing on disk: truncated data, a damaged master index, absurd sizes -> DbError, no crash."""
    from meryl_amd import db
    import shutil
    k, wp = 21, 10
    rng = np.random.default_rng(9)
    path, keys, counts, _ = make_db(tmp_path, oracle_lib, k, wp, random_reads(rng, 60, 40, 200), "ok")
    # 1. data file cut short
    p1 = str(tmp_path / "trunc"); shutil.copytree(path, p1)
    f = os.path.join(p1, "0x000000.merylData")
    data = open(f, "rb").read()
    open(f, "wb").write(data[:len(data) // 2])
    r = db.Reader(p1)
    with pytest.raises(db.DbError):
        r.read_file(0)
    r.close()
    # 2. block payload zeroed (unary codes never terminate)
    p2 = str(tmp_path / "zero"); shutil.copytree(path, p2)
    f = os.path.join(p2, "0x000000.merylData")
    idx = np.fromfile(os.path.join(p2, "0x000000.merylIndex"), dtype=np.uint64).reshape(-1, 3)
    big = int(np.argmax(idx[:, 2]))
    if idx[big, 2] > 0:
        start = int(idx[big, 1]) + 48 + 66
        buf = bytearray(open(f, "rb").read())
        end = int(idx[big + 1, 1]) if big + 1 < len(idx) else len(buf)
        buf[start:end] = bytes(end - start)
        open(f, "wb").write(bytes(buf))
        r = db.Reader(p2)
        with pytest.raises(db.DbError):
            r.read_file(0)
        r.close()
    # 3. master index with an absurd histogram length / parameters
    p3 = str(tmp_path / "master"); shutil.copytree(path, p3)
    f = os.path.join(p3, "merylIndex")
    buf = bytearray(open(f, "rb").read())
    words = np.frombuffer(bytes(buf[48:]), dtype="<u8").copy()          # header 16+8+8 bytes, then nw, nalloc, words
    words[7] = np.uint64(0xFFFFFFFFFFFFFF)                              # nPairs
    open(f, "wb").write(bytes(buf[:48]) + words.tobytes())
    with pytest.raises(db.DbError):
        db.Reader(p3)
    buf2 = bytearray(open(os.path.join(path, "merylIndex"), "rb").read())
    w2 = np.frombuffer(bytes(buf2[48:]), dtype="<u8").copy()
    w2[2] = np.uint64((200 << 32) | 3)                                  # prefixSize 200
    open(f, "wb").write(bytes(buf2[:48]) + w2.tobytes())
    with pytest.raises(db.DbError):
        db.Reader(p3)


def _cli():
    from meryl_amd import build
    return build.build_cli()


def test_cli_dumpfile_print_and_doc_size_bound(tmp_path, native_lib):
    """`meryl dumpFile <db>/0x######` (src/meryl/meryl.C:41-45) prints the three tables of usage.rst:24-45; the doc's own
    example -- suffixSize 104, blocks of 22363 and 16486 k-mers at blkPos 0 / 345448 / 601248 -- bounds the layout
    assumptions A2-A6: our encoding of blocks of those sizes lands within 0.1 % of the documented block sizes."""
    import subprocess
    from meryl_amd import db
    k, wp = 57, 10                                            # 2k - wp = 104 = the example's suffixSize
    rng = np.random.default_rng(1)
    path = str(tmp_path / "doc.meryl")
    w = db.Writer(path, k, wp)
    wl = db.Writer(path + ".labelled", k, wp, label_size=3)
    sizes = [22363, 16486, 17345]
    blocks = []
    for p, n in enumerate(sizes):
        hi = np.sort(rng.integers(0, 1 << 40, n, dtype=np.uint64))      # suffix = hi:lo, 104 bits
        lo = rng.integers(0, 1 << 63, n, dtype=np.uint64)
        cn = rng.integers(1, 9, n).astype(np.uint32)
        w.add_block(p, lo, cn, hi)
        wl.add_block(p, lo, cn, hi, label=5)
        blocks.append((lo, hi, cn))
    for p in range(len(sizes), 16):                           # addBlock is called for every prefix of a file, empty ones too
        w.add_block(p, [], [], [])
        wl.add_block(p, [], [], [])
    w.close()
    wl.close()
    out = subprocess.run([_cli(), "-Q", "dumpFile", path + "/0x000000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.split("\n")
    i0 = lines.index("---------- --------- ---------")
    idx = [l.split() for l in lines[i0 + 1:i0 + 1 + 16]]
    assert [int(r[2]) for r in idx[:4]] == sizes + [0] and idx[0][0] == "0x00000000" and int(idx[0][1]) == 0
    pos = [int(r[1]) for r in idx]
    for got, doc in ((pos[1] - pos[0], 345448), (pos[2] - pos[1], 601248 - 345448)):
        assert abs(got - doc) / doc < 1e-3, (got, doc)        # usage.rst:29-31
    i1 = [i for i, l in enumerate(lines) if l.startswith("------------------ --------")][0]
    h0 = lines[i1 + 1].split()
    assert h0 == ["0x0000000000000000", "22363", "1", "15", "89", "0x0000000000000000", "1", "0x0000000000000000", "0x0000000000000000"]   # usage.rst:36
    i2 = [i for i, l in enumerate(lines) if l.startswith("-------- -----------")][0]
    lo, hi, cn = blocks[0]
    suffix0 = (int(hi[0]) << 64) | int(lo[0])
    f = lines[i2 + 1].split()
    top = suffix0 >> 89
    assert int(f[0]) == 0 and int(f[1]) == top and int(f[2], 16) == top and int(f[3]) == 25 and int(f[5]) == 64
    assert int(f[4], 16) == (suffix0 >> 64) & ((1 << 25) - 1) and int(f[6], 16) == suffix0 & ((1 << 64) - 1) and int(f[7]) == int(cn[0])
    assert sum(1 for l in lines[i2 + 1:] if l.strip()) == sum(sizes)
    # print: k-mer, value, label in binary (meryl2's third column)
    pr = subprocess.run([_cli(), "-Q", "print", path + ".labelled"], stdout=subprocess.PIPE, text=True)
    first = pr.stdout.split("\n")[0].split("\t")
    assert len(first[0]) == k and int(first[1]) == int(cn[0]) and first[2] == "101"
    assert len(subprocess.run([_cli(), "-Q", "print", path], stdout=subprocess.PIPE, text=True).stdout.split("\n")[0].split("\t")) == 2
    di = subprocess.run([_cli(), "-Q", "dumpIndex", path + ".labelled"], stdout=subprocess.PIPE, text=True).stdout
    assert "prefixSize     10" in di and "suffixSize     104" in di and "labelSize      3" in di


@pytest.mark.parametrize("k,label_size", [(21, 0), (40, 6)])
def test_lookup_estimate_needs_no_device(native_lib, tmp_path, k, label_size):
    """merylExactLookup::estimateMemoryUsage as meryl-lookup uses it before loading (meryl-lookup.C:62-87): k-mers kept by the value
    filter from the database's own histogram, bytes from the table layout -- on a box without a GPU -- and the CLI's -estimate /
    -memory report on top of it."""
    import ctypes
    import subprocess
    from meryl_amd import build, capi, db
    rng = np.random.default_rng(k)
    wp = 8
    n = 5000
    keys = sorted({int(x) for x in rng.integers(0, 1 << 62, n)} if k <= 32 else
                  {(int(a) << 40) | int(b) for a, b in zip(rng.integers(0, 1 << 40, n), rng.integers(0, 1 << 40, n))})
    keys = [x & ((1 << (2 * k)) - 1) for x in keys]
    keys = sorted(set(keys))
    counts = rng.integers(1, 40, len(keys)).astype(np.uint32)
    w = db.Writer(str(tmp_path / "d.meryl"), k, wp, label_size)
    wd = 2 * k - wp
    by_prefix = {}
    for x, c in zip(keys, counts):
        by_prefix.setdefault(x >> wd, []).append((x & ((1 << wd) - 1), int(c)))
    for p in range(1 << wp):
        rows = by_prefix.get(p, [])
        lo = np.array([s & ((1 << 64) - 1) for s, _ in rows], dtype=np.uint64)
        hi = np.array([s >> 64 for s, _ in rows], dtype=np.uint64)
        w.add_block(p, lo, np.array([c for _, c in rows], dtype=np.uint32), hi if wd > 64 else None, label=3)
    w.close()
    L = native_lib
    for vmin, vmax in ((0, 2 ** 64 - 1), (5, 20), (39, 39), (100, 200)):
        info = capi.LookupInfo()
        assert L.mgc_lookup_estimate(str(tmp_path / "d.meryl").encode(), vmin, vmax, ctypes.byref(info)) == 0
        kept = int(np.sum((counts >= vmin) & (counts <= vmax)))
        assert (info.k, info.n_kmers_in_db, info.n_kmers) == (k, len(keys), kept)
        kw = 2 if k > 32 else 1
        assert info.key_words == kw and info.device_bytes == (8 * kw + 4) * kept + 8 * ((1 << info.index_bits) + 1)
    assert L.mgc_lookup_estimate(str(tmp_path / "nope").encode(), 0, 1, ctypes.byref(capi.LookupInfo())) != 0
    cli = build.build_lookup_cli()
    p = subprocess.run([cli, "-existence", "-estimate", "-mers", str(tmp_path / "d.meryl"), "-min", "5", "-max", "20", "-memory", "1"],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert "Memory required:" in p.stderr and "Memory limit:     1.000 GB" in p.stderr and "-estimate option enabled" in p.stderr
    assert "%d of %d %d-mers kept" % (int(np.sum((counts >= 5) & (counts <= 20))), len(keys), k) in p.stderr
    p = subprocess.run([cli, "-existence", "-estimate", "-mers", str(tmp_path / "d.meryl"), "-memory", "0.00001"],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 1 and "Not enough memory" in p.stderr


def test_raw_file_lists_validated_blocks(native_lib, tmp_path):
    """mdb_reader_raw_file (what the device decoder is fed): the data file's bytes and one descriptor per block that holds k-mers --
    object offsets = the index's positions, k-mers and output offsets add up, a truncated file is refused."""
    import ctypes
    from meryl_amd import capi, db
    k, wp = 15, 8
    rng = np.random.default_rng(3)
    keys = np.unique(rng.integers(0, 1 << (2 * k), 20_000, dtype=np.uint64))
    keys = keys[(keys >> np.uint64(2 * k - wp)) % np.uint64(3) != 0]          # some prefixes stay empty
    cnt = rng.integers(1, 100, keys.size).astype(np.uint32)
    path = str(tmp_path / "db")
    w = db.Writer(path, k, wp)
    w_data = 2 * k - wp
    pref = (keys >> np.uint64(w_data)).astype(np.int64)
    starts = np.searchsorted(pref, np.arange(0, (1 << wp) + 1))
    for p in range(1 << wp):
        w.add_block(p, keys[starts[p]:starts[p + 1]] & np.uint64((1 << w_data) - 1), cnt[starts[p]:starts[p + 1]])
    w.close()
    L = capi.lib()
    r = db.Reader(path)
    total = 0
    for ff in (0, 5, 63):
        by, sz, bl, nb, nk = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint64()
        assert L.mdb_reader_raw_file(r._h, ff, ctypes.byref(by), ctypes.byref(sz), ctypes.byref(bl), ctypes.byref(nb), ctypes.byref(nk)) == 0
        idx = [e for e in r.file_index(ff) if e[2]]
        desc = np.ctypeslib.as_array(ctypes.cast(bl, ctypes.POINTER(ctypes.c_uint64)), shape=(max(nb.value, 1), 6))[:nb.value].copy()
        assert nb.value == len(idx) and nk.value == sum(e[2] for e in idx) == len(r.read_file(ff)[0])
        assert [int(x) for x in desc[:, 0]] == [e[1] for e in idx] and [int(x) for x in desc[:, 4]] == [e[0] for e in idx]
        assert [int(x) for x in desc[:, 5]] == list(np.cumsum([0] + [e[2] for e in idx])[:-1])
        assert sz.value == os.path.getsize(os.path.join(path, "0x%s.merylData" % format(ff, "06b")))
        L.mdb_free(by); L.mdb_free(bl)
        total += nk.value
    r.close()
    # a data file cut short: the framing check refuses it before anything is uploaded
    name = os.path.join(path, "0x%s.merylData" % format(5, "06b"))
    data = open(name, "rb").read()
    open(name, "wb").write(data[:len(data) - 40])
    r = db.Reader(path)
    by, sz, bl, nb, nk = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint64()
    assert L.mdb_reader_raw_file(r._h, 5, ctypes.byref(by), ctypes.byref(sz), ctypes.byref(bl), ctypes.byref(nb), ctypes.byref(nk)) != 0
    r.close()
