"""CPU tests of the oracle itself (test infrastructure): pinned against the
reference's known-answer vector and ordering rule, and cross-checked three
ways (C brute force, C++ reference-algorithm port, pure Python)."""
import numpy as np
import pytest

from conftest import golden_cases
from helpers import MODES, expected_arrays, join128, python_count, random_reads

CASES = golden_cases()


def test_reference_known_answer_vector(oracle_lib):
    # documentation/source/reference.rst:545-568: GGAGCT, k=3 -> AGC 2, CTC 1, TCC 1 in that order
    hi, lo, cn, ni = oracle_lib.count_brute("GGAGCT", 3)
    got = [(oracle_lib.kmer_to_string(h, l, 3), int(c)) for h, l, c in zip(hi, lo, cn)]
    assert got == [("AGC", 2), ("CTC", 1), ("TCC", 1)]
    assert ni == 4
    # per-position canonical choice of the same table: GGA->TCC, GAG->CTC, AGC->AGC, GCT->AGC
    ehi, elo = oracle_lib.enumerate_kmers("GGAGCT", 3)
    assert [oracle_lib.kmer_to_string(h, l, 3) for h, l in zip(ehi, elo)] == ["TCC", "CTC", "AGC", "AGC"]


def test_base_order_is_ACTG(oracle_lib):
    # src/tests/test-operations.pl:114-118: db order == lexicographic order after tr/GT/TG/
    hi, lo, cn, _ = oracle_lib.count_brute("ACGTTGCATGTCGCATGATGCATGAGAGCTACG.", 5, oracle_lib.FORWARD)
    strs = [oracle_lib.kmer_to_string(h, l, 5) for h, l in zip(hi, lo)]
    swapped = [s.translate(str.maketrans("GT", "TG")) for s in strs]
    assert swapped == sorted(swapped)
    assert len(set(strs)) == len(strs)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_golden_cases(oracle_lib, case):
    hi, lo, cn, ni = oracle_lib.count_brute(case["bases"], case["k"], MODES[case["mode"]])
    ek, ec = expected_arrays(case)
    assert join128(hi, lo) == ek
    assert [int(c) for c in cn] == ec
    assert ni == case["n_instances"] == sum(ec)
    # independent pure-python statement agrees too
    pk, pc = python_count(case["bases"], case["k"], MODES[case["mode"]])
    assert pk == ek and pc == ec


def test_count_suffix_restatement(oracle_lib):
    # count-suffix=<bases> (merylOp.H:139-147, merylOp-countSimple.C:50-58,88-93): the k-mer being counted is kept only if
    # it ends in the bases.  Hand-checkable: 4-mers of ACGTAC are ACGT, CGTA, GTAC; canonical (A<C<T<G): ACGT (its own
    # reverse complement), CGTA vs TACG -> CGTA, GTAC (palindrome).  Ending in "AC": GTAC only; in "T": ACGT only.
    hi, lo, cn, ni = oracle_lib.count_brute("ACGTAC.", 4, oracle_lib.CANONICAL, "AC")
    assert [oracle_lib.kmer_to_string(h, l, 4) for h, l in zip(hi, lo)] == ["GTAC"] and list(cn) == [1] and ni == 1
    hi, lo, cn, ni = oracle_lib.count_brute("ACGTAC.", 4, oracle_lib.CANONICAL, "T")
    assert [oracle_lib.kmer_to_string(h, l, 4) for h, l in zip(hi, lo)] == ["ACGT"] and ni == 1
    # forward keeps CGTA (ends in A), reverse keeps TACG's ... check against a filter of the unfiltered stream, all modes
    rng = np.random.default_rng(3)
    bases = random_reads(rng, 200, 30, 200)
    for k, sfx in ((7, "G"), (21, "TC"), (40, "ACGTA")):
        code = 0
        for ch in sfx:
            code = (code << 2) | "ACTG".index(ch)
        for mode in (oracle_lib.CANONICAL, oracle_lib.FORWARD, oracle_lib.REVERSE):
            ahi, alo, acn, _ = oracle_lib.count_brute(bases, k, mode)
            keep = (alo & np.uint64((1 << (2 * len(sfx))) - 1)) == np.uint64(code)
            fhi, flo, fcn, fni = oracle_lib.count_brute(bases, k, mode, sfx)
            assert np.array_equal(fhi, ahi[keep]) and np.array_equal(flo, alo[keep]) and np.array_equal(fcn, acn[keep])
            assert fni == int(acn[keep].sum()) and keep.sum() > 0
    with pytest.raises(RuntimeError):
        oracle_lib.count_brute("ACGT", 3, oracle_lib.CANONICAL, "AN")


@pytest.mark.parametrize("k,wp", [(6, 6), (16, 10), (21, 10), (21, 14), (31, 12), (32, 10), (33, 10), (51, 12), (64, 10)])
def test_port_equals_brute(oracle_lib, k, wp):
    # the reference-algorithm restatement (buckets, bit-packed store, std::sort, RLE,
    # 2 MiB chunks with k-1 carry) gives the brute-force stream, for any thread count
    rng = np.random.default_rng(k * 100 + wp)
    bases = random_reads(rng, 300, 1, 400)
    a = oracle_lib.count_brute(bases, k)
    for threads in (1, 4):
        b = oracle_lib.count_threaded(bases, k, wp, threads=threads)
        assert a[3] == b[3]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_port_chunk_boundary(oracle_lib):
    # a single read longer than the 2 MiB loader buffer: k-mers that straddle the
    # buffer cut are counted exactly once (merylOp-countThreads.C:149-155,221-222)
    rng = np.random.default_rng(7)
    read = "".join(np.array(list("ACGT"))[rng.integers(0, 4, 5_000_000)])
    bases = read + "." + read[:1000] + "."
    a = oracle_lib.count_brute(bases, 21)
    b = oracle_lib.count_threaded(bases, 21, 10, threads=3)
    assert a[3] == b[3] == (5_000_000 - 20) + (1000 - 20)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_spill_thread_invariance_sweep(oracle_lib):
    # the reference's config sweep (src/tests/test-build.pl:66-74: memory x threads) must not
    # change the result; here: different wPrefix (what memory= changes) and thread counts
    bases = oracle_lib.synth_reads(11, 20000, 0, 400).tobytes()
    ref = oracle_lib.count_brute(bases, 22)
    for wp in (10, 12, 17):
        for threads in (1, 2, 8):
            got = oracle_lib.count_threaded(bases, 22, wp, threads=threads)
            assert np.array_equal(ref[1], got[1]) and np.array_equal(ref[2], got[2])


def test_configure_matches_survey_table(oracle_lib):
    # SURVEY.md 3.2 (scratch restatement of merylOp-count.C:173-227,300-403)
    GB = 1 << 30
    table = {
        (4641652, 21): [10, 10, 10, 10],
        (10_000_000_000, 21): [18, 18, 18, 18],
        (90_000_000_000, 21): [17, 21, 21, 21],
        (90_000_000_000, 31): [16, 18, 21, 21],
        (150_000_000_000, 51): [16, 17, 20, 22],
    }
    for (n, k), want in table.items():
        got = [oracle_lib.configure_counting(k, n, m * GB)["w_prefix"] for m in (16, 64, 256, 1024)]
        assert got == want, (n, k, got)
    c = oracle_lib.configure_counting(21, 10_000_000_000, 64 * GB)
    assert c["w_data"] == 42 - 18 and c["n_prefix"] == 1 << 18 and c["use_simple"] == 0
    # small k: simple mode is cheaper (merylOp-count.C:368-372)
    assert oracle_lib.configure_counting(10, 1_000_000, 4 * GB)["use_simple"] == 1


def test_homopoly_compress(oracle_lib):
    assert oracle_lib.homopoly_compress(b"AAACCGTTTT") == b"ACGT"
    assert oracle_lib.homopoly_compress(b"AaAcCG") == b"AcG"
    # _lastByte carries the run across chunks of one sequence (merylInput.C:261-268)
    assert oracle_lib.homopoly_compress(b"AAAC", last_byte=b"A") == b"C"
    assert oracle_lib.homopoly_compress(b"", last_byte=b"A") == b""


def test_compress_chunk_carry_equals_whole(oracle_lib):
    # _lastByte makes chunked compression equal to compressing the whole sequence (merylInput.C:261-268)
    rng = np.random.default_rng(2)
    seq = "".join(np.repeat(np.array(list("ACGTacgtN"))[rng.integers(0, 9, 400)], rng.integers(1, 6, 400)))
    stream = seq + "." + seq[::-1] + ".AAAA.C."
    whole = oracle_lib.compress_stream(stream)
    for chunk in (1, 2, 7, 64):
        assert oracle_lib.compress_stream(stream, chunk) == whole
    assert b"AA" not in whole.upper() and b"CC" not in whole.upper()
    # runs never merge across sequences
    assert oracle_lib.compress_stream("AAA.AAA.") == b"A.A."


def test_synth_reads_deterministic(oracle_lib):
    a = oracle_lib.synth_reads(2, 1_000_000, 0, 100)
    b = oracle_lib.synth_reads(2, 1_000_000, 50, 50)
    assert a.size == 100 * 151 and np.array_equal(a[50 * 151:], b)
    assert set(np.unique(a).tolist()) <= set(b"ACGTN.")
    assert (a[150::151] == ord(".")).all()


def test_digest_collect_threaded_agrees_with_digest_and_collect(oracle_lib):
    """the full-size parity test's checker: digests of all 64 files + the streams of a few whole files from ONE port run"""
    import oracle
    b = oracle.synth_reads(3, 50_000, 0, 3000, 150, 5000, 100)
    cfg = oracle.configure_counting(21, b.size, 1 << 30)
    d1, nd1, ni1 = oracle.digest_threaded(b, 21, cfg["w_prefix"], threads=4)
    d2, nd2, ni2, files = oracle.digest_collect_threaded(b, 21, cfg["w_prefix"], (0, 21, 63), threads=4)
    assert np.array_equal(d1, d2) and (nd1, ni1) == (nd2, ni2)
    hi, lo, cn, _ = oracle.count_threaded(b, 21, cfg["w_prefix"], threads=4)
    for f in (0, 21, 63):
        m = (lo >> np.uint64(36)) == f
        assert np.array_equal(files[f][1], lo[m]) and np.array_equal(files[f][2], cn[m]) and not files[f][0].any()
