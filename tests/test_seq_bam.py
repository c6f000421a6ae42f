"""CPU tests of the SAM / BAM / BGZF side of the loader (include/meryl_seq.h).  The files are written here, from the
SAM specification (SAMv1 4.1 BGZF, 4.2 BAM), by code that shares nothing with the C++ reader."""
import ctypes
import gzip
import random
import struct
import zlib

import pytest

from test_seq import load_all

NT16 = "=ACMGRSVTWYHKDBN"


def bgzf_block(data):
    assert len(data) <= 0xff00
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    cdata = c.compress(data) + c.flush()
    bsize = 12 + 6 + len(cdata) + 8
    head = struct.pack("<BBBBIBBH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
    return head + cdata + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))


def bgzf(data, block=0xff00, eof_marker=True):
    out = b"".join(bgzf_block(data[i:i + block]) for i in range(0, len(data), block))
    return out + (bgzf_block(b"") if eof_marker else b"")


def bam_bytes(records, refs=(("chr1", 1000), ("chrUn_x", 5))):
    """records: list of (name, flag, seq) with seq over NT16 ('' = absent)"""
    text = "@HD\tVN:1.6\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    b = b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs))
    for name, ln in refs:
        b += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    rng = random.Random(7)
    for name, flag, seq in records:
        rn = name.encode() + b"\0"
        cigar = [(len(seq) << 4) | 0] if seq and rng.random() < 0.7 else []          # one M op, or none
        packed = bytearray((len(seq) + 1) // 2)
        for i, ch in enumerate(seq):
            packed[i // 2] |= NT16.index(ch) << (4 if i % 2 == 0 else 0)
        qual = bytes([0xff]) * len(seq)
        aux = b"NMC\x00" if rng.random() < 0.5 else b""
        body = struct.pack("<iiBBHHHiiii", 0, 10, len(rn), 30, 4681, len(cigar), flag, len(seq), -1, -1, 0)
        body += rn + b"".join(struct.pack("<I", c) for c in cigar) + bytes(packed) + qual + aux
        b += struct.pack("<i", len(body)) + body
    return b


def random_records(n, seed, longest=300):
    rng = random.Random(seed)
    recs = []
    for i in range(n):
        ln = rng.choice([0, 1, 2, 3, 150, 151, rng.randrange(0, longest)])
        alphabet = "ACGT" if rng.random() < 0.8 else NT16
        recs.append(("read%d/%d" % (seed, i), rng.choice([0, 16, 4, 256, 2048, 77, 141]), "".join(rng.choice(alphabet) for _ in range(ln))))
    return recs


@pytest.mark.parametrize("threads", ["0", "1", "5"])
@pytest.mark.parametrize("max_len", [1, 7, 1 << 16])
def test_bam_records_come_out_as_stored(native_lib, tmp_path, monkeypatch, max_len, threads):
    # every record -- forward, reverse-flagged, unmapped, secondary, supplementary, without SEQ -- gives its stored SEQ;
    # odd and even lengths, IUPAC codes and '=', records straddling BGZF blocks (tiny 300-byte blocks), a 200 kb read
    # longer than three blocks.  MERYL_BGZF_THREADS=0 reads the same file through zlib's single stream.
    monkeypatch.setenv("MERYL_BGZF_THREADS", threads)
    recs = random_records(400 if max_len > 1 else 40, 1) + [("long", 0, "".join(random.Random(3).choice("ACGT") for _ in range(200_001)))]
    if max_len == 1:
        recs = recs[:40]
    raw = bam_bytes(recs)
    for block in (300, 0xff00):
        p = tmp_path / ("x%d.bam" % block)
        p.write_bytes(bgzf(raw, block))
        r = native_lib.msr_open(str(p).encode())
        assert r and native_lib.msr_format(r) == 1 and native_lib.msr_is_compressed(r) == 1
        native_lib.msr_close(r)
        assert load_all(native_lib, str(p), max_len) == [s.encode() for _, _, s in recs]
    # plain gzip (not BGZF) around the same bytes, and a name without the suffix: the content decides
    g = tmp_path / "plain_gzip_container"
    with gzip.open(g, "wb") as f:
        f.write(raw)
    assert load_all(native_lib, str(g), max_len) == [s.encode() for _, _, s in recs]


def test_bam_empty_and_damaged(native_lib, tmp_path):
    p = tmp_path / "none.bam"
    p.write_bytes(bgzf(bam_bytes([])))
    assert load_all(native_lib, str(p), 100) == []
    raw = bgzf(bam_bytes(random_records(50, 2)), 300)
    cut = tmp_path / "cut.bam"
    cut.write_bytes(raw[:len(raw) // 2])                                       # ends inside a block
    r = native_lib.msr_open(str(cut).encode())                                 # a small file is one batch: refused at open
    buf = ctypes.create_string_buffer(1 << 16)
    n, eos = ctypes.c_uint64(0), ctypes.c_int(0)
    rc = 1
    while r and rc > 0:
        rc = native_lib.msr_load_bases(r, buf, 1 << 16, ctypes.byref(n), ctypes.byref(eos))
    assert ((not r) or rc < 0) and b"cut.bam" in native_lib.msr_last_error() and b"truncated" in native_lib.msr_last_error()
    if r:
        native_lib.msr_close(r)
    flipped = bytearray(raw)
    flipped[len(raw) // 3] ^= 0x55                                             # CRC or deflate stream no longer right
    bad = tmp_path / "flipped.bam"
    bad.write_bytes(bytes(flipped))
    r = native_lib.msr_open(str(bad).encode())
    rc = 1
    while r and rc > 0:
        rc = native_lib.msr_load_bases(r, buf, 1 << 16, ctypes.byref(n), ctypes.byref(eos))
    assert (not r) or rc < 0
    if r:
        native_lib.msr_close(r)
    notbam = tmp_path / "fake.bam"
    notbam.write_text(">r\nACGT\n")
    assert not native_lib.msr_open(str(notbam).encode()) and b"not a BAM" in native_lib.msr_last_error()
    assert not native_lib.msr_open(b"x.cram") and b"CRAM" in native_lib.msr_last_error()


@pytest.mark.parametrize("max_len", [1, 5, 1 << 16])
def test_sam_text(native_lib, tmp_path, max_len):
    sam = ("@HD\tVN:1.6\n@SQ\tSN:c\tLN:9\n@CO\tfree text with\ttabs\n"
           "r1\t0\tc\t1\t30\t4M\t*\t0\t0\tACGT\tIIII\tNM:i:0\n"
           "r2\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*\n"
           "\n"
           "r3\t16\tc\t2\t30\t3M\t*\t0\t0\tnnA\t*\r\n"
           "r4\t0\tc\t1\t0\t1M\t*\t0\t0\tG\tI")                                # no newline at the end
    p = tmp_path / "x.sam"
    p.write_text(sam)
    r = native_lib.msr_open(str(p).encode())
    assert r and native_lib.msr_format(r) == 2
    native_lib.msr_close(r)
    assert load_all(native_lib, str(p), max_len) == [b"ACGT", b"", b"nnA", b"G"]
    headed = tmp_path / "no_suffix"                                            # told by its @HD line
    headed.write_text(sam)
    assert load_all(native_lib, str(headed), max_len) == [b"ACGT", b"", b"nnA", b"G"]
    short = tmp_path / "short.sam"
    short.write_text("r1\t0\tc\t1\n")
    r = native_lib.msr_open(str(short).encode())
    buf = ctypes.create_string_buffer(16)
    n, eos = ctypes.c_uint64(0), ctypes.c_int(0)
    assert native_lib.msr_load_bases(r, buf, 16, ctypes.byref(n), ctypes.byref(eos)) < 0
    native_lib.msr_close(r)


@pytest.mark.parametrize("threads", ["1", "4"])
def test_bgzipped_fastq_equals_plain(native_lib, tmp_path, monkeypatch, threads):
    # bgzip output is a valid .gz: parsed (msr_load_bases) and raw (msr_read_text) reads give what the plain file gives
    monkeypatch.setenv("MERYL_BGZF_THREADS", threads)
    rng = random.Random(5)
    text = "".join("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in
                   enumerate("".join(rng.choice("ACGTN") for _ in range(rng.randrange(1, 400))) for _ in range(3000)))
    plain = tmp_path / "r.fastq"
    plain.write_text(text)
    bz = tmp_path / "r.fastq.gz"
    bz.write_bytes(bgzf(text.encode(), 20_000))
    want = load_all(native_lib, str(plain), 1 << 16)
    assert len(want) == 3000 and load_all(native_lib, str(bz), 1 << 16) == want
    r = native_lib.msr_open(str(bz).encode())
    assert r and native_lib.msr_format(r) == 0
    got = b""
    buf = ctypes.create_string_buffer(70_001)
    while True:
        k = native_lib.msr_read_text(r, buf, 70_001)
        assert k >= 0
        if k == 0:
            break
        got += buf.raw[:k]
    native_lib.msr_close(r)
    assert got == text.encode()


@pytest.mark.parametrize("max_len", [2, 3, 64, 1 << 16])
def test_load_stream_is_the_sequences_with_breakers(native_lib, tmp_path, max_len):
    # msr_load_stream = what a loop over msr_load_bases + '.' per ended sequence builds, for every format, whatever the
    # buffer size (a sequence longer than the buffer continues in the next call without a breaker)
    from test_seq import FASTA, FASTQ
    recs = random_records(60, 9, longest=200)
    files = {"x.fasta": FASTA.encode(), "x.fastq": FASTQ.encode(), "x.bam": bgzf(bam_bytes(recs), 700)}
    for name, content in files.items():
        p = tmp_path / name
        p.write_bytes(content)
        want = b"".join(s + b"." for s in load_all(native_lib, str(p), 1 << 16))
        r = native_lib.msr_open(str(p).encode())
        buf = ctypes.create_string_buffer(max_len)
        n = ctypes.c_uint64(0)
        got = b""
        while True:
            rc = native_lib.msr_load_stream(r, buf, max_len, ctypes.byref(n))
            assert rc >= 0
            if rc == 0:
                break
            assert 0 < n.value <= max_len
            got += buf.raw[:n.value]
        native_lib.msr_close(r)
        assert got == want, name


def test_bgzf_oddities(native_lib, tmp_path, monkeypatch):
    # other extra subfields before 'BC', empty (EOF-marker) blocks in the middle of a file, a file too short to hold a
    # BGZF header, and an empty file: all read like the plain text
    monkeypatch.setenv("MERYL_BGZF_THREADS", "3")
    from test_seq import FASTA
    want = [b"ACGTACGTNNacgt", b"TTTT", b"", b"GATTACA"]

    def block_with_extra(data):
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        cdata = c.compress(data) + c.flush()
        extra = b"XY" + struct.pack("<H", 3) + b"abc"                          # a foreign subfield first
        xlen = len(extra) + 6
        bsize = 12 + xlen + len(cdata) + 8
        return (struct.pack("<BBBBIBBH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, xlen) + extra + b"BC" + struct.pack("<HH", 2, bsize - 1) +
                cdata + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))
    t = FASTA.encode()
    p = tmp_path / "odd.fa.gz"
    p.write_bytes(block_with_extra(t[:10]) + bgzf_block(b"") + bgzf_block(t[10:30]) + bgzf_block(b"") + bgzf_block(b"") + block_with_extra(t[30:]) + bgzf_block(b""))
    assert load_all(native_lib, str(p), 1 << 16) == want
    tiny = tmp_path / "tiny.fa"
    tiny.write_text(">a\nAC")
    assert load_all(native_lib, str(tiny), 16) == [b"AC"]
    empty = tmp_path / "empty.fa"
    empty.write_bytes(b"")
    assert load_all(native_lib, str(empty), 16) == []
    only_eof = tmp_path / "only_eof.bam"                                        # a BGZF file of nothing but the EOF marker is not a BAM
    only_eof.write_bytes(bgzf_block(b""))
    assert not native_lib.msr_open(str(only_eof).encode()) and b"not a BAM" in native_lib.msr_last_error()


def test_bgzf_damage_deep_inside_a_file_is_an_error(native_lib, tmp_path, monkeypatch):
    # more than one batch of blocks, the damage in a later one: the sequences before it arrive, then the loader reports it
    monkeypatch.setenv("MERYL_BGZF_THREADS", "4")
    rng = random.Random(2)
    text = "".join(">r%d\n%s\n" % (i, "".join(rng.choice("ACGT") for _ in range(300))) for i in range(4000))
    raw = bytearray(bgzf(text.encode(), 1000))                                  # ~1300 blocks: three batches of 512
    raw[len(raw) * 4 // 5] ^= 0x3c
    p = tmp_path / "late_damage.fa.gz"
    p.write_bytes(bytes(raw))
    from test_seq import _drain
    seqs, rc = _drain(native_lib, str(p))
    assert rc < 0 and 0 < seqs < 4000 and b"late_damage" in native_lib.msr_last_error()


def test_bam_reader_survives_corrupt_payloads(native_lib, tmp_path, monkeypatch):
    # bit flips and truncations INSIDE the BAM payload (the BGZF wrapping stays valid, so they reach the record decoder):
    # whatever happens -- an error, fewer or odd sequences -- the loader must come back, and never hand out more bases
    # than it was given room for
    monkeypatch.setenv("MERYL_BGZF_THREADS", "2")
    rng = random.Random(11)
    good = bam_bytes(random_records(120, 4, longest=120))
    for trial in range(150):
        b = bytearray(good)
        for _ in range(rng.randrange(1, 6)):
            i = rng.randrange(len(b))
            b[i] ^= 1 << rng.randrange(8)
        if trial % 5 == 0:
            b = b[:rng.randrange(8, len(b))]
        p = tmp_path / "fuzz.bam"
        p.write_bytes(bgzf(bytes(b), rng.choice([200, 4000, 0xff00])))
        r = native_lib.msr_open(str(p).encode())
        if not r:
            continue
        room = rng.choice([2, 17, 4096])
        buf = ctypes.create_string_buffer(room + 8)
        n = ctypes.c_uint64(0)
        calls = 0
        while calls < 100000:
            buf.raw = b"\xa5" * (room + 8)
            rc = native_lib.msr_load_stream(r, buf, room, ctypes.byref(n))
            calls += 1
            assert n.value <= room and buf.raw[room:] == b"\xa5" * 8          # nothing written past the room given
            if rc <= 0:
                break
        native_lib.msr_close(r)
        assert calls < 100000
