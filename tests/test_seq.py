"""CPU tests of the FASTA/FASTQ(.gz) loader (include/meryl_seq.h), the stand-in for
dnaSeqFile::loadBases of the absent meryl-utility: contract of merylInput::loadBases
(src/meryl/merylInput.H:67-70)."""
import ctypes
import gzip
import os

import pytest


def load_all(lib, path, max_len):
    """-> list of sequences (bases only), reassembled from chunks of at most max_len"""
    r = lib.msr_open(path.encode())
    assert r, lib.msr_last_error()
    buf = ctypes.create_string_buffer(max_len)
    n = ctypes.c_uint64(0)
    eos = ctypes.c_int(0)
    seqs, cur = [], b""
    while True:
        rc = lib.msr_load_bases(r, buf, max_len, ctypes.byref(n), ctypes.byref(eos))
        assert rc >= 0, lib.msr_last_error()
        if rc == 0:
            break
        assert n.value <= max_len
        cur += buf.raw[:n.value]
        if eos.value:
            seqs.append(cur)
            cur = b""
    assert cur == b""
    lib.msr_close(r)
    return seqs


FASTA = ">r1 some description\nACGTACGT\nNNacgt\n\n>r2\nTTTT\n>empty\n>r4\nGATTACA"
FASTQ = "@q1\nACGTN\n+\nIIIII\n@q2 x\nGGGG\nCC\n+q2\nIIII\nII\n@q3\nA\n+\n@\n"


@pytest.mark.parametrize("max_len", [1, 3, 7, 1 << 16])
def test_fasta_fastq_plain_and_gz(native_lib, tmp_path, max_len):
    fa = tmp_path / "x.fasta"
    fa.write_text(FASTA)
    assert load_all(native_lib, str(fa), max_len) == [b"ACGTACGTNNacgt", b"TTTT", b"", b"GATTACA"]
    fq = tmp_path / "x.fastq"
    fq.write_text(FASTQ)
    assert load_all(native_lib, str(fq), max_len) == [b"ACGTN", b"GGGGCC", b"A"]       # '@' as a quality is not a record
    gz = tmp_path / "x.fastq.gz"
    with gzip.open(gz, "wt") as f:
        f.write(FASTQ)
    assert load_all(native_lib, str(gz), max_len) == [b"ACGTN", b"GGGGCC", b"A"]
    crlf = tmp_path / "crlf.fa"
    crlf.write_bytes(FASTA.replace("\n", "\r\n").encode())
    assert load_all(native_lib, str(crlf), max_len) == [b"ACGTACGTNNacgt", b"TTTT", b"", b"GATTACA"]


def test_reader_errors_and_guess(native_lib, tmp_path):
    assert not native_lib.msr_open(str(tmp_path / "missing.fa").encode())
    assert b"cannot open" in native_lib.msr_last_error()
    assert not native_lib.msr_open(b"reads.fq.bz2")                          # no such file
    assert b"cannot open" in native_lib.msr_last_error()
    assert not native_lib.msr_open(b"reads.cram")                            # no samtools on the PATH (or no such file)
    assert b"CRAM" in native_lib.msr_last_error() or b"cannot open" in native_lib.msr_last_error()
    bad = tmp_path / "bad.txt"
    bad.write_text("hello\n")
    r = native_lib.msr_open(str(bad).encode())
    buf = ctypes.create_string_buffer(16)
    n, eos = ctypes.c_uint64(0), ctypes.c_int(0)
    assert native_lib.msr_load_bases(r, buf, 16, ctypes.byref(n), ctypes.byref(eos)) < 0
    native_lib.msr_close(r)
    # file-size guess of merylOp-count.C:410-433
    p = tmp_path / "g.fa"
    p.write_text("A" * 1000)
    assert native_lib.msr_guess_number_of_kmers(str(p).encode()) == 1000
    pz = tmp_path / "g.fa.gz"
    pz.write_bytes(b"x" * 100)
    assert native_lib.msr_guess_number_of_kmers(str(pz).encode()) == 300
    assert native_lib.msr_guess_number_of_kmers(b"-") == 0


@pytest.mark.parametrize("max_len", [3, 1 << 16])
def test_bz2_and_xz_through_the_system_decompressors(native_lib, tmp_path, max_len):
    import bz2, lzma, shutil
    if not (shutil.which("bzip2") and shutil.which("xz")):
        pytest.skip("bzip2 / xz not installed")
    b = tmp_path / "it's x.fastq.bz2"                                        # a name the shell must not split or expand
    b.write_bytes(bz2.compress(FASTQ.encode()))
    assert load_all(native_lib, str(b), max_len) == [b"ACGTN", b"GGGGCC", b"A"]
    x = tmp_path / "x.fasta.xz"
    x.write_bytes(lzma.compress(FASTA.encode()))
    assert load_all(native_lib, str(x), max_len) == [b"ACGTACGTNNacgt", b"TTTT", b"", b"GATTACA"]
    r = native_lib.msr_open(str(x).encode())
    assert r and native_lib.msr_is_compressed(r) == 1 and native_lib.msr_format(r) == 0
    native_lib.msr_close(r)


def _drain(lib, path):
    """-> (number of sequences delivered, rc of the last call) ; rc < 0 = the loader reported an error"""
    r = lib.msr_open(path.encode())
    if not r:
        return 0, -1
    buf = ctypes.create_string_buffer(1 << 16)
    n, eos = ctypes.c_uint64(0), ctypes.c_int(0)
    seqs, rc = 0, 1
    while rc > 0:
        rc = lib.msr_load_bases(r, buf, 1 << 16, ctypes.byref(n), ctypes.byref(eos))
        seqs += 1 if (rc > 0 and eos.value) else 0
    lib.msr_close(r)
    return seqs, rc


def test_damaged_compressed_inputs_are_errors_not_short_files(native_lib, tmp_path):
    # a .gz that stops in the middle of its stream, a .bz2 with a flipped byte: the loader must say so -- ending the
    # input quietly would count fewer k-mers than the file holds
    import bz2, random, shutil
    rng = random.Random(1)
    text = "".join(">r%d\n%s\n" % (i, "".join(rng.choice("ACGT") for _ in range(200))) for i in range(5000))
    whole = tmp_path / "whole.fa.gz"
    with gzip.open(whole, "wt") as f:
        f.write(text)
    assert _drain(native_lib, str(whole)) == (5000, 0)
    cut = tmp_path / "cut.fa.gz"
    cut.write_bytes(whole.read_bytes()[:whole.stat().st_size // 2])
    seqs, rc = _drain(native_lib, str(cut))
    assert rc < 0 and seqs < 5000 and b"cut.fa.gz" in native_lib.msr_last_error()
    if shutil.which("bzip2"):
        raw = bytearray(bz2.compress(text.encode()))
        raw[len(raw) // 2] ^= 0xff
        bad = tmp_path / "bad.fa.bz2"
        bad.write_bytes(bytes(raw))
        seqs, rc = _drain(native_lib, str(bad))
        assert rc < 0 and b"bad.fa.bz2" in native_lib.msr_last_error()


def test_cram_goes_through_samtools_view_when_it_is_there(native_lib, tmp_path, monkeypatch):
    """CRAM: decoded by `samtools view -h` when the binary exists -- a stand-in script on the
    PATH plays it here (the image has no samtools) and prints SAM records, which take the SAM path: SEQ of every record as stored."""
    import os
    import stat
    fake = tmp_path / "bin"
    fake.mkdir()
    tool = fake / "samtools"
    tool.write_text("#!/bin/sh\n[ \"$1\" = view ] || exit 2\nprintf '@HD\\tVN:1.6\\n@SQ\\tSN:c\\tLN:100\\n'\n"
                    "printf 'r1\\t0\\tc\\t1\\t60\\t8M\\t*\\t0\\t0\\tACGTACGT\\tIIIIIIII\\n'\n"
                    "printf 'r2\\t4\\t*\\t0\\t0\\t*\\t*\\t0\\t0\\t*\\t*\\n'\n"
                    "printf 'r3\\t16\\tc\\t9\\t60\\t5M\\t*\\t0\\t0\\tGGNCC\\tIIIII\\n'\n")
    tool.chmod(tool.stat().st_mode | stat.S_IXUSR | stat.S_IXGRP | stat.S_IXOTH)
    cram = tmp_path / "reads.cram"
    cram.write_bytes(b"CRAM\x03\x00 not a real container: the stand-in never reads it")
    monkeypatch.setenv("PATH", str(fake) + os.pathsep + os.environ.get("PATH", ""))
    assert load_all(native_lib, str(cram), 64) == [b"ACGTACGT", b"", b"GGNCC"]     # (an absent SEQ is an empty sequence, as for SAM / BAM)
    # without the binary: refused, and the message says why
    monkeypatch.setenv("PATH", str(tmp_path / "nowhere"))
    assert not native_lib.msr_open(str(cram).encode())
    assert b"samtools" in native_lib.msr_last_error() and b"CRAM" in native_lib.msr_last_error()
