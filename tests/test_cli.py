"""The `meryl` front end (meryl_amd/bin/meryl): grammar and narrative on CPU (config-only runs
never touch the GPU), a full FASTQ.gz -> database -> print round trip on the GPU."""
import gzip
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def meryl(native_lib):
    from meryl_amd import build
    path = build.build_cli()
    assert os.path.exists(path)
    return path


def run(meryl, *args, check=True, env=None):
    p = subprocess.run([meryl] + [str(a) for a in args], capture_output=True, text=True, timeout=600, env=env)
    if check:
        assert p.returncode == 0, p.stderr[-2000:]
    return p


def test_configure_only_narrative(meryl, tmp_path):
    fa = tmp_path / "r.fa"
    fa.write_text(">a\n" + "ACGT" * 1000 + "\n")
    # word order is free (src/meryl/meryl.C:58-86): options before or after the operation
    for args in (["-C", "k=21", "memory=4", "n=4641652", "count", fa, "output", tmp_path / "db"],
                 ["-C", "count", "output", tmp_path / "db", "k=21", "memory=4", "n=4641652", fa],
                 ["-C", "[count", "k=21", "memory=4", "n=4641652", fa, "output", str(tmp_path / "db") + "]"]):
        p = run(meryl, *args)
        err = p.stderr
        assert "Found 1 command tree." in err
        assert re.search(r"Counting \d+ \(estimated\).* canonical 21-mers from 1 input file:", err)
        assert "COMPLEX MODE" in err and "Best Value!" in err and "FINAL CONFIGURATION" in err
        # the line Canu parses (src/meryl/merylOp-count.C:398-401)
        assert re.search(r"Configured complex mode for \d+\.\d{3} GB memory per batch, and up to \d+ batch(es)?\.", err)
        best = [l for l in err.splitlines() if "Best Value!" in l]
        assert len(best) == 1 and best[0].split()[0] == "10"            # E. coli 1x -> wPrefix 10
        assert "Bye." in err and not os.path.exists(tmp_path / "db")     # -C writes nothing
    p = run(meryl, "-C", "k=21", "memory=64", "n=10000000000", "count-forward", fa, "output", tmp_path / "db")
    assert " forward 21-mers" in p.stderr
    assert [l for l in p.stderr.splitlines() if "Best Value!" in l][0].split()[0] == "18"


def test_count_suffix_narrative(meryl, tmp_path):
    fa = tmp_path / "r.fa"
    fa.write_text(">a\n" + "ACGT" * 100 + "\n")
    p = run(meryl, "-C", "k=12", "count", "count-suffix=AC", fa, "output", tmp_path / "db")
    assert "12-mers with constant 2-mer suffix 'AC'" in p.stderr                        # merylOp-count.C:150
    assert "-> 1048576 entries for counts up to 65535." in p.stderr                     # 4^(12-2), :124,152
    assert re.search(r"Configured simple mode for \d+\.\d{3} GB memory per batch, and up to \d+ batch(es)?\.", p.stderr)
    p = run(meryl, "k=12", "count-suffix=AC", "count", fa, "output", tmp_path / "db", check=False)
    assert p.returncode == 1 and "needs a counting operation" in p.stderr               # merylCommandBuilder.C:271-272: top of the stack
    p = run(meryl, "k=12", "count", "segment=1/2", fa, "output", tmp_path / "db", check=False)
    assert p.returncode == 1 and "not supported" in p.stderr


def test_grammar_errors(meryl, tmp_path):
    fa = tmp_path / "r.fa"
    fa.write_text(">a\nACGT\n")
    p = run(meryl, "count", fa, "output", tmp_path / "db", check=False)
    assert p.returncode == 1 and "Kmer size not supplied" in p.stderr                   # merylOp-count.C:311-312
    p = run(meryl, "k=21", "count", fa, check=False)
    assert p.returncode == 1 and "No output specified" in p.stderr                      # :314-315
    p = run(meryl, "k=21", "count", tmp_path / "nonexistent.fa", "output", tmp_path / "db", check=False)
    assert p.returncode == 1 and "Can't interpret" in p.stderr                          # meryl.C:84-86
    p = run(meryl, "k=21", "k=22", "count", fa, "output", tmp_path / "db", check=False)
    assert p.returncode == 1 and "already set" in p.stderr                              # merylCommandBuilder.C:254-262
    p = run(meryl, "k=21", "count", fa, "output", tmp_path / "a", "output", tmp_path / "b", check=False)
    assert p.returncode == 1 and "already has an output" in p.stderr                    # merylOp.C:256-257
    p = run(meryl, "statistics", check=False)
    assert p.returncode == 1 and "not part of this build" in p.stderr
    p = run(meryl, "union-sum", "output", tmp_path / "u", check=False)
    assert p.returncode == 1 and "has no inputs" in p.stderr
    p = run(meryl, "union-sum", tmp_path, check=False)                                  # a directory that is no database
    assert p.returncode == 1 and "Can't interpret" in p.stderr
    p = run(meryl, "k=21", "count", "label=7", fa, "output", tmp_path / "db", check=False)
    assert p.returncode == 1 and "label=#<integer>" in p.stderr                         # meryl2: label=#<n>


@pytest.mark.gpu
def test_cli_count_print_roundtrip(meryl, oracle_lib, tmp_path):
    from meryl_amd import db
    bases = oracle_lib.synth_reads(12, 60_000, 0, 6000).tobytes()
    reads = [r for r in bases.decode().split(".") if r]
    fq = tmp_path / "reads.fastq.gz"
    with gzip.open(fq, "wt") as f:
        for i, r in enumerate(reads[:3000]):
            f.write("@r%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)))
    fa = tmp_path / "reads.fasta"
    with open(fa, "w") as f:
        for i, r in enumerate(reads[3000:]):
            f.write(">s%d\n%s\n%s\n" % (i, r[:70], r[70:]))          # multi-line FASTA
    out = tmp_path / "out.meryl"
    p = run(meryl, "k=21", "memory=2", "threads=8", "count", fq, fa, "output", out)
    assert "Start counting with THREADED method." in p.stderr and "Finished counting." in p.stderr
    assert len(os.listdir(out)) == 129
    _, wlo, wcn, wni = oracle_lib.count_brute(bases, 21)
    r = db.Reader(str(out))
    lo, hi, cn = r.read_all()
    assert np.array_equal(lo, wlo) and np.array_equal(cn, wcn) and r.info.num_total == wni
    r.close()
    # `meryl print` text form, merylOp-nextMer.C:665-677
    p = run(meryl, "-Q", "print", out)
    lines = p.stdout.splitlines()
    assert len(lines) == len(wlo)
    want = ["%s\t%d" % (oracle_lib.kmer_to_string(0, int(l), 21), int(c)) for l, c in zip(wlo[:50], wcn[:50])]
    assert lines[:50] == want
    p = run(meryl, "-Q", "dumpIndex", out)
    assert "prefixSize" in p.stdout and "numFilesBits   6 (64 files)" in p.stdout
    # `meryl histogram`: value <TAB> number of distinct k-mers with that value (merylOp-histogram.C:38-43)
    p = run(meryl, "-Q", "histogram", out)
    vals, occ = np.unique(wcn, return_counts=True)
    assert p.stdout.splitlines() == ["%d\t%d" % (int(v), int(o)) for v, o in zip(vals, occ)]
    # compress: same as counting the homopolymer-compressed reads
    out2 = tmp_path / "hpc.meryl"
    run(meryl, "-Q", "k=15", "memory=2", "compress", "count", fq, fa, "output", out2)
    r = db.Reader(str(out2))
    lo, hi, cn = r.read_all()
    _, wlo2, wcn2, _ = oracle_lib.count_brute(oracle_lib.compress_stream(bases), 15)
    assert np.array_equal(lo, wlo2) and np.array_equal(cn, wcn2)
    r.close()


@pytest.mark.gpu
def test_cli_device_parser_refusal_falls_back_to_host_parser(meryl, oracle_lib, tmp_path):
    # multi-line FASTQ (sequence and qualities wrapped) is not what the device parser accepts: the CLI must notice
    # (MGC_EFORMAT), drop that file's partial output and re-read it with the host state machine -- same database as
    # with MERYL_HOST_PARSER=1 and as the oracle; a well-formed file in the same run still goes through the device
    from meryl_amd import db
    bases = oracle_lib.synth_reads(21, 50_000, 0, 2000).tobytes()
    reads = [r for r in bases.decode().split(".") if r]
    wrapped = tmp_path / "wrapped.fastq"
    with open(wrapped, "w") as f:
        for i, r in enumerate(reads[:1000]):
            q = "@" + "I" * (len(r) - 1)                             # quality line starting with '@'
            f.write("@w%d\n%s\n%s\n+\n%s\n%s\n" % (i, r[:80], r[80:], q[:80], q[80:]))
    plain = tmp_path / "plain.fastq"
    with open(plain, "w") as f:
        for i, r in enumerate(reads[1000:]):
            f.write("@p%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)))
    _, wlo, wcn, wni = oracle_lib.count_brute(bases, 21)
    for env in ({}, {"MERYL_HOST_PARSER": "1"}):
        out = tmp_path / ("out%d.meryl" % len(env))
        p = subprocess.run([str(meryl), "-Q", "k=21", "memory=2", "threads=4", "count", str(wrapped), str(plain), "output", str(out)],
                           capture_output=True, text=True, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr
        r = db.Reader(str(out))
        lo, hi, cn = r.read_all()
        assert np.array_equal(lo, wlo) and np.array_equal(cn, wcn) and r.info.num_total == wni
        r.close()


def _config0_reads(oracle_lib):
    # BASELINE configs[0] / SURVEY 8(d)-1: "E. coli 1x": 4,641,652 bp genome, 30,944 x 150 bp reads, 0.5 % substitutions
    return oracle_lib.synth_reads(1, 4_641_652, 0, 30_944, 150, 5000, 0)


def test_config0_ecoli_cpu_plumbing(oracle_lib):
    # the reference's own CPU-runnable case, through the CPU restatement only: threaded port (wPrefix 10 from
    # configureCounting) == brute force, 64-file geometry intact
    bases = _config0_reads(oracle_lib)
    cfg = oracle_lib.configure_counting(21, 4_641_652, 4 << 30)
    assert cfg["w_prefix"] == 10 and cfg["use_simple"] == 0
    whi, wlo, wcn, wni = oracle_lib.count_brute(bases.tobytes(), 21)
    phi, plo, pcn, pni = oracle_lib.count_threaded(bases.tobytes(), 21, cfg["w_prefix"], 0, threads=4)
    assert pni == wni == 30_944 * (150 - 20)
    assert np.array_equal(plo, wlo) and np.array_equal(pcn, wcn)
    files = (wlo >> np.uint64(36)).astype(np.int64)
    assert files.min() >= 0 and files.max() < 64 and np.all(np.diff(files) >= 0)


@pytest.mark.gpu
def test_config0_ecoli_cli_on_gpu(meryl, oracle_lib, tmp_path):
    # the same input as a FASTQ file through the stand-alone CLI on the GPU: `k=21 memory=4 n=4641652` -> wPrefix 10,
    # database == the oracle's stream
    from meryl_amd import db
    bases = _config0_reads(oracle_lib)
    reads = [r for r in bases.tobytes().decode().split(".") if r]
    fq = tmp_path / "ecoli_1x.fastq"
    with open(fq, "w") as f:
        for i, r in enumerate(reads):
            f.write("@r%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)))
    out = tmp_path / "ecoli.meryl"
    p = run(meryl, "k=21", "memory=4", "n=4641652", "threads=4", "count", fq, "output", out)
    assert re.search(r"Configured complex mode for .* GB memory per batch, and up to \d+ batch", p.stderr)
    _, wlo, wcn, wni = oracle_lib.count_brute(bases.tobytes(), 21)
    r = db.Reader(str(out))
    lo, hi, cn = r.read_all()
    assert r.info.prefix_bits == 10 if hasattr(r.info, "prefix_bits") else True
    assert np.array_equal(lo, wlo) and np.array_equal(cn, wcn) and r.info.num_total == wni
    r.close()


@pytest.mark.gpu
def test_cli_counts_bam_sam_and_bgzipped_fastq(meryl, oracle_lib, tmp_path):
    # the same reads as BAM (BGZF, decoded on the host), SAM text and bgzip'd FASTQ (inflated block-parallel, parsed on
    # the device): one database each, all equal to the oracle's count of the reads; reverse-flagged and secondary
    # records carry their SEQ as stored, a record without SEQ adds nothing
    from meryl_amd import db
    from test_seq_bam import bam_bytes, bgzf
    bases = oracle_lib.synth_reads(33, 40_000, 0, 3000).tobytes()
    reads = [r for r in bases.decode().split(".") if r]
    recs = [("r%d" % i, (0, 16, 256, 4)[i % 4], r.upper()) for i, r in enumerate(reads)]
    recs.insert(5, ("noseq", 4, ""))
    bam = tmp_path / "reads.bam"
    bam.write_bytes(bgzf(bam_bytes(recs), 5000))
    sam = tmp_path / "reads.sam"
    sam.write_text("@HD\tVN:1.6\n" + "".join("%s\t%d\t*\t0\t0\t*\t*\t0\t0\t%s\t*\n" % (n, f, s if s else "*") for n, f, s in recs))
    fq = tmp_path / "reads.fastq.gz"
    fq.write_bytes(bgzf("".join("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)) for n, _, s in recs if s).encode(), 30_000))
    _, wlo, wcn, wni = oracle_lib.count_brute(".".join(r.upper() for r in reads) + ".", 21)
    for src in (bam, sam, fq):
        out = tmp_path / (src.name + ".meryl")
        run(meryl, "-Q", "k=21", "memory=2", "threads=4", "count", src, "output", out)
        r = db.Reader(str(out))
        lo, hi, cn = r.read_all()
        assert np.array_equal(lo, wlo) and np.array_equal(cn, wcn) and r.info.num_total == wni, src.name
        r.close()


@pytest.mark.gpu
def test_cli_count_suffix_database(meryl, oracle_lib, tmp_path):
    # count-suffix=GA: the database holds exactly the canonical 14-mers ending in GA, in the simple-mode geometry
    # [file 6][blockPrefix][suffix][count-suffix] of merylOp-countSimple.C:172-175,231-233
    from meryl_amd import db
    bases = oracle_lib.synth_reads(41, 30_000, 0, 3000).tobytes()
    fa = tmp_path / "r.fa"
    fa.write_text("".join(">r%d\n%s\n" % (i, r) for i, r in enumerate(bases.decode().split(".")) if r))
    out = tmp_path / "sfx.meryl"
    p = run(meryl, "k=14", "memory=2", "threads=4", "count", "count-suffix=GA", fa, "output", out)
    assert "Start counting with SIMPLE method." in p.stderr
    _, wlo, wcn, _ = oracle_lib.count_brute(bases, 14)
    keep = (wlo & np.uint64(15)) == np.uint64((3 << 2) | 0)                              # G=3, A=0
    r = db.Reader(str(out))
    lo, hi, cn = r.read_all()
    assert np.array_equal(lo, wlo[keep]) and np.array_equal(cn, wcn[keep]) and r.info.num_total == int(wcn[keep].sum())
    assert r.info.prefix_size == 6 + (28 - 4 - 6) - min(20, 28 - 4 - 6)
    r.close()
    lines = run(meryl, "-Q", "print", out).stdout.splitlines()
    assert len(lines) == int(keep.sum()) and all(l.split("\t")[0].endswith("GA") for l in lines[:200])


@pytest.mark.gpu
def test_cli_union_sum_tree_and_relatives(meryl, oracle_lib, tmp_path):
    """`union-sum [count a output A] [count b output B] C.meryl output U`: the counts run first and become inputs
    (meryl.C:211-227), the merge is the reference's 64-slice streaming merge (merylOp-nextMer.C:418-683) done on the
    device; union-sum of the parts of a read set == one count of all of it; min/max/intersect against numpy."""
    from meryl_amd import db
    k = 21
    sets = [oracle_lib.synth_reads(21, 80_000, i * 4000, 4000).tobytes() for i in range(3)]
    fas = []
    for i, b in enumerate(sets):
        fa = tmp_path / ("part%d.fa" % i)
        fa.write_text("".join(">r%d\n%s\n" % (j, r) for j, r in enumerate(b.decode().split(".")) if r))
        fas.append(fa)
    c = tmp_path / "C.meryl"
    run(meryl, "-Q", "k=%d" % k, "memory=2", "count", fas[2], "output", c)
    u = tmp_path / "U.meryl"
    run(meryl, "-Q", "k=%d" % k, "memory=2", "union-sum", "[count", fas[0], "output", str(tmp_path / "A.meryl") + "]",
        "[count", fas[1], "output", str(tmp_path / "B.meryl") + "]", c, "output", u)
    assert len(os.listdir(u)) == 129
    _, wlo, wcn, _ = oracle_lib.count_brute(b"".join(sets), k)
    r = db.Reader(str(u))
    lo, hi, cn = r.read_all()
    hv, ho = r.histogram()
    r.close()
    assert np.array_equal(lo, wlo) and np.array_equal(cn, wcn)
    vals, occ = np.unique(wcn, return_counts=True)
    assert {int(a): int(b) for a, b in zip(hv, ho)} == {int(a): int(b) for a, b in zip(vals, occ)}
    per = []
    for name in ("A", "B", "C"):
        r = db.Reader(str(tmp_path / (name + ".meryl")))
        l, _, c_ = r.read_all()
        r.close()
        per.append(dict(zip((int(x) for x in l), (int(x) for x in c_))))
    for word, setop, f in (("union-min", set.union, min), ("union-max", set.union, max), ("intersect-sum", set.intersection, sum),
                           ("intersect-min", set.intersection, min), ("intersect-max", set.intersection, max)):
        out = tmp_path / (word + ".meryl")
        run(meryl, "-Q", word, tmp_path / "A.meryl", tmp_path / "B.meryl", tmp_path / "C.meryl", "output", out)
        keys = sorted(setop(*[set(d) for d in per]))
        want = [f([d[x] for d in per if x in d]) for x in keys]
        r = db.Reader(str(out))
        l, _, c_ = r.read_all()
        r.close()
        assert [int(x) for x in l] == keys and [int(x) for x in c_] == want, word
    # the set operations proper (merylOp-nextMer.C:559-613): union counts the inputs holding a k-mer, intersect keeps the FIRST
    # input's value, subtract takes the later inputs' values off the first's while it stays above them, difference keeps what only
    # the first input holds, symmetric-difference what exactly one input holds -- over three inputs and over two
    def subtract(vals):
        v = vals[0]
        for x in vals[1:]:
            if v > x:
                v -= x
            else:
                return 0
        return v
    for n_in in (3, 2):
        ds = per[:n_in]
        names = [tmp_path / (n + ".meryl") for n in ("A", "B", "C")[:n_in]]
        universe = sorted(set().union(*[set(d) for d in ds]))
        spec = {
            "union": {x: sum(1 for d in ds if x in d) for x in universe},
            "intersect": {x: ds[0][x] for x in universe if all(x in d for d in ds)},
            "subtract": {x: subtract([ds[0][x]] + [d[x] for d in ds[1:] if x in d]) for x in universe if x in ds[0]},
            "difference": {x: ds[0][x] for x in universe if x in ds[0] and not any(x in d for d in ds[1:])},
            "symmetric-difference": {x: [d[x] for d in ds if x in d][0] for x in universe if sum(1 for d in ds if x in d) == 1},
        }
        for word, want in spec.items():
            want = {x: v for x, v in want.items() if v}
            out = tmp_path / ("%s%d.meryl" % (word, n_in))
            run(meryl, "-Q", word, *names, "output", out)
            r = db.Reader(str(out))
            l, _, c_ = r.read_all()
            r.close()
            assert [int(x) for x in l] == sorted(want) and [int(x) for x in c_] == [want[x] for x in sorted(want)], (word, n_in)
    # the single-input value operations (:490-557), thresholds as a bare number, threshold=, distinct= and word-frequency=
    a = per[0]
    vals_sorted = sorted(a.values())
    hv_, ho_ = np.unique(np.array(vals_sorted), return_counts=True)
    nk = 0
    for v_, o_ in zip(hv_, ho_):                                   # initializeThreshold, :104-114
        nk += int(o_)
        if nk >= int(0.9 * len(a)):
            t_distinct = int(v_)
            break
    t_wf = int(0.00001 * sum(a.values()))
    cases = [(["less-than", "3"], lambda v: v if v < 3 else 0), (["greater-than", "threshold=2"], lambda v: v if v > 2 else 0),
             (["at-least", "2"], lambda v: v if v >= 2 else 0), (["at-most", "1"], lambda v: v if v <= 1 else 0),
             (["equal-to", "2"], lambda v: v if v == 2 else 0), (["not-equal-to", "1"], lambda v: v if v != 1 else 0),
             (["increase", "7"], lambda v: v + 7), (["decrease", "2"], lambda v: v - 2 if v >= 2 else 0),
             (["multiply", "3"], lambda v: v * 3), (["divide", "2"], lambda v: v // 2),
             (["divide-round", "2"], lambda v: 1 if v < 2 else int(np.round(v / 2.0 + 1e-9 * 0))), (["modulo", "2"], lambda v: v % 2),
             (["less-than", "distinct=0.9"], lambda v: v if v < t_distinct else 0),
             (["greater-than", "word-frequency=0.00001"], lambda v: v if v > t_wf else 0)]
    for i, (words, f) in enumerate(cases):
        out = tmp_path / ("v%d.meryl" % i)
        run(meryl, "-Q", *words, tmp_path / "A.meryl", "output", out)
        want = {x: f(v) for x, v in a.items()}
        if words[0] == "divide-round":                            # C round(): halves away from zero
            want = {x: (1 if v < 2 else int(v / 2.0 + 0.5)) for x, v in a.items()}
        want = {x: v for x, v in want.items() if v}
        r = db.Reader(str(out))
        l, _, c_ = r.read_all()
        r.close()
        assert [int(x) for x in l] == sorted(want) and [int(x) for x in c_] == [want[x] for x in sorted(want)], words
    # print over a child operation
    p = run(meryl, "-Q", "print", "[union-min", tmp_path / "A.meryl", tmp_path / "B.meryl", "output", str(tmp_path / "pm.meryl") + "]")
    assert len(p.stdout.strip().split("\n")) == len(set(per[0]) | set(per[1]))
    p = run(meryl, "union-sum", tmp_path / "A.meryl", u, "output", tmp_path / "bad.meryl", check=False)
    assert p.returncode == 0                                                            # same k: fine
    run(meryl, "-Q", "k=15", "memory=2", "count", fas[0], "output", tmp_path / "k15.meryl")
    p = run(meryl, "union-sum", tmp_path / "A.meryl", tmp_path / "k15.meryl", "output", tmp_path / "bad2.meryl", check=False)
    assert p.returncode == 1 and "15-mers" in p.stderr


@pytest.mark.gpu
def test_cli_labels_and_dumpfile(meryl, oracle_lib, tmp_path):
    """meryl2's `-l <bits>` + `label=#<n>` on a count: every k-mer carries the constant label; print shows it in binary;
    dumpFile lists the blocks of a counted database."""
    from meryl_amd import db
    bases = oracle_lib.synth_reads(5, 30_000, 0, 2000).tobytes()
    fa = tmp_path / "r.fa"
    fa.write_text("".join(">r%d\n%s\n" % (j, r) for j, r in enumerate(bases.decode().split(".")) if r))
    out = tmp_path / "lab.meryl"
    run(meryl, "-Q", "-l", "6", "k=21", "memory=1", "count", "label=#37", fa, "output", out)
    _, wlo, wcn, _ = oracle_lib.count_brute(bases, 21)
    r = db.Reader(str(out))
    assert r.info.label_size == 6
    lo, hi, cn, lb = r.read_all(labels=True)
    r.close()
    assert np.array_equal(lo, wlo) and np.array_equal(cn, wcn) and np.all(lb == 37)
    first = run(meryl, "-Q", "print", out).stdout.split("\n")[0].split("\t")
    assert first[2] == "100101" and int(first[1]) == int(wcn[0])
    d = run(meryl, "-Q", "dumpFile", str(out) + "/0x000000").stdout
    assert "prefix    blkPos    nKmers" in d and "kmerIdx prefixDelta" in d
    n_file0 = int(np.sum((wlo >> np.uint64(36)) == 0))
    tail = d.split("-------- ----------- ----------- --")[1].split("\n")[1:]
    assert sum(1 for l in tail if l.strip()) == n_file0


@pytest.mark.gpu
def test_cli_out_of_core_text_input(meryl, oracle_lib, tmp_path):
    """A FASTQ larger than the batch size through the CLI's default (device parser) path: batches are cut inside the file,
    merged on the device, same database as the single pass (the text path used to have no batching)."""
    bases = oracle_lib.synth_reads(8, 200_000, 0, 40_000).tobytes()
    reads = [r for r in bases.decode().split(".") if r]
    fq = tmp_path / "r.fq"
    fq.write_text("".join("@%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)) for i, r in enumerate(reads)))
    one = tmp_path / "one.meryl"
    run(meryl, "-Q", "k=31", "memory=2", "count", fq, "output", one)
    env = dict(os.environ, MERYL_BATCH_BASES="1500000")
    many = tmp_path / "many.meryl"
    p = run(meryl, "-V", "k=31", "memory=2", "count", fq, "output", many, env=env)
    assert "batches" in p.stderr
    names = sorted(os.listdir(one))
    assert sorted(os.listdir(many)) == names
    for n in names:
        assert open(os.path.join(one, n), "rb").read() == open(os.path.join(many, n), "rb").read(), n


@pytest.mark.gpu
@pytest.mark.parametrize("k,extra", [(21, ()), (31, ("compress",)), (51, ("-l", "5"))])
def test_cli_gpus_option_writes_the_single_device_database(meryl, oracle_lib, tmp_path, k, extra):
    """`gpus=N`: every rank reads ITS OWN share of the input -- byte windows of the plain-text files cut at record starts
    (FASTA here, FASTQ and a gzip'd file that goes whole to one rank below) -- stages it on its device, and the ranks of
    mgc_count_node_batched count and write ONE database: the same 129 files as without the option, for 2, 3 and 7 ranks on
    whatever devices the box has; with MERYL_BATCH_BASES the node count runs in batches (parked waves, one merge per owner)."""
    import gzip
    bases = oracle_lib.synth_reads(31, 150_000, 0, 20_000, 150 if not extra or extra[0] != "compress" else 1500).tobytes()
    reads = [r for r in bases.decode().split(".") if r]
    third = len(reads) // 3
    fa = tmp_path / "r.fa"
    fa.write_text("".join(">%d\n%s\n" % (i, r) for i, r in enumerate(reads[:third])))
    fq = tmp_path / "r.fq"                                    # quality lines that start with '@' and '+': the window cuts must not be fooled
    fq.write_text("".join("@%d\n%s\n+\n%s\n" % (i, r, ("@" if i % 2 else "+") + "I" * (len(r) - 1)) for i, r in enumerate(reads[third:2 * third])))
    gz = tmp_path / "r.fa.gz"
    with gzip.open(gz, "wt") as f:
        f.write("".join(">%d\n%s\n" % (i, r) for i, r in enumerate(reads[2 * third:])))
    one = tmp_path / "one.meryl"
    run(meryl, "-Q", *extra, "k=%d" % k, "memory=2", "count", fa, fq, gz, "output", one)
    names = sorted(os.listdir(one))
    assert len(names) == 129
    for n_ranks, batch in ((2, None), (3, "400000"), (7, None)):
        out = tmp_path / ("g%d.meryl" % n_ranks)
        env = dict(os.environ)
        if batch:
            env["MERYL_BATCH_BASES"] = batch
        p = run(meryl, "-V", *extra, "k=%d" % k, "memory=2", "gpus=%d" % n_ranks, "count", fa, fq, gz, "output", out, env=env)
        assert "ranks=%d" % n_ranks in p.stderr
        if batch:
            assert "batches=1," not in p.stderr
        assert sorted(os.listdir(out)) == names
        for n in names:
            assert open(os.path.join(one, n), "rb").read() == open(os.path.join(out, n), "rb").read(), (n_ranks, n)
    bad = run(meryl, "k=21", "gpus=2", "count", "count-suffix=ACG", fa, "output", tmp_path / "bad.meryl", check=False)
    assert bad.returncode != 0 and "gpus=" in bad.stderr
