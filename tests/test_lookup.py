"""GPU tests of the exact lookup table (include/meryl_lookup.h): merylExactLookup's load / value
as meryl-lookup uses them (src/meryl-lookup/meryl-lookup.C:36-100, existence.C:63-82), against a dictionary built
from the oracle's counts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _keys_tensor(torch, lo, hi, k):
    if k > 32:
        return torch.from_numpy(np.stack([lo, hi], axis=1).view(np.int64).copy()).cuda()
    return torch.from_numpy(lo.view(np.int64).copy()).cuda()


@pytest.mark.parametrize("k", [21, 31, 51, 8])
def test_lookup_values_stream_and_existence(native_lib, oracle_lib, tmp_path, k):
    import torch
    from meryl_amd import capi, count, lookup
    bases = oracle_lib.synth_reads(41, 150_000, 0, 20_000, 150, 5000, 2000)          # plenty of N
    whi, wlo, wcn, _ = oracle_lib.count_brute(bases.tobytes(), k)
    table = {(int(h) << 64) | int(l): int(c) for h, l, c in zip(whi, wlo, wcn)}
    cfg = capi.configure(k, bases.size, 1 << 30)
    d = torch.from_numpy(bases).cuda()
    with count.Session(cfg, 0) as s:
        s.push_bases_device(d)
        s.count()
        keys, cnts = s.result_device()
        s.write_database(str(tmp_path / "db"), 4)
    for src in ("device", "file"):
        lk = lookup.Lookup.from_device(keys, cnts, k) if src == "device" else lookup.Lookup.load(str(tmp_path / "db"))
        assert lk.info.n_kmers == len(wlo) == lk.info.n_kmers_in_db and lk.info.k == k
        # 1. value(): every stored k-mer, and k-mers that are not there
        got = lk.values(keys).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, wcn)
        rng = np.random.default_rng(k)
        qlo = rng.integers(0, 1 << 62, 5000, dtype=np.uint64) & np.uint64((1 << min(2 * k, 64)) - 1 if 2 * k < 64 else 0xFFFFFFFFFFFFFFFF)
        qhi = (rng.integers(0, 1 << 62, 5000, dtype=np.uint64) & np.uint64((1 << (2 * k - 64)) - 1)) if k > 32 else np.zeros(5000, np.uint64)
        got = lk.values(_keys_tensor(torch, qlo, qhi, k)).cpu().numpy().view(np.uint32)
        want = [table.get((int(h) << 64) | int(l), 0) for h, l in zip(qhi, qlo)]
        assert [int(x) for x in got] == want
        # 2. every window of a base stream: reads the table was built from, plus foreign sequence
        other = oracle_lib.synth_reads(77, 50_000, 0, 300, 150, 5000, 100)
        stream = np.concatenate([bases[:60_000], other])
        ehi, elo = oracle_lib.enumerate_kmers(stream.tobytes(), k, 0)                  # canonical k-mers in input order
        vals = lk.stream(torch.from_numpy(stream).cuda()).cpu().numpy().view(np.uint32)
        # positions of valid windows, independently: k consecutive ACGT
        ok = np.isin(stream, np.frombuffer(b"ACGTacgt", dtype=np.uint8))
        run = np.zeros(stream.size + 1, dtype=np.int64)
        run[1:] = np.cumsum(~ok)
        starts = np.nonzero(run[k:] - run[:-k] == 0)[0]
        assert starts.size == elo.size
        want = np.zeros(stream.size, dtype=np.uint32)
        want[starts] = [table.get((int(h) << 64) | int(l), 0) for h, l in zip(ehi, elo)]
        assert np.array_equal(vals, want)
        # 3. -existence: per sequence (read + its breaker) total k-mers and k-mers found
        seq_start = np.concatenate([[0], np.nonzero(stream == ord("."))[0] + 1])
        if seq_start[-1] != stream.size:
            seq_start = np.append(seq_start, stream.size)
        tot, fnd = lk.existence(torch.from_numpy(stream).cuda(), seq_start)
        sid = np.searchsorted(seq_start, starts, side="right") - 1
        wt = np.bincount(sid, minlength=seq_start.size - 1)
        wf = np.bincount(sid, weights=(want[starts] > 0), minlength=seq_start.size - 1).astype(np.int64)
        assert np.array_equal(tot.astype(np.int64), wt) and np.array_equal(fnd.astype(np.int64), wf)
        assert fnd[:100].sum() == tot[:100].sum() and fnd.sum() < tot.sum()           # own reads all found, foreign ones not
        lk.close()
    # 4. the value filter of load (-min / -max): only k-mers with 2 <= value <= 5
    for lk in (lookup.Lookup.load(str(tmp_path / "db"), 2, 5), lookup.Lookup.from_device(keys, cnts, k, 2, 5)):
        keep = (wcn >= 2) & (wcn <= 5)
        assert lk.info.n_kmers == int(keep.sum()) and lk.info.n_kmers_in_db == len(wlo)
        got = lk.values(keys).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, np.where(keep, wcn, 0))
        lk.close()


def test_lookup_tiny_and_empty(native_lib):
    import torch
    from meryl_amd import lookup
    keys = torch.tensor([5, 9, 1 << 40], dtype=torch.int64).cuda()
    cnts = torch.tensor([1, 2, 3], dtype=torch.int32).cuda()
    with lookup.Lookup.from_device(keys, cnts, 21) as lk:
        assert lk.info.index_bits == 0 and lk.info.n_kmers == 3
        q = torch.tensor([0, 5, 6, 9, 1 << 40, (1 << 40) + 1], dtype=torch.int64).cuda()
        assert lk.values(q).cpu().tolist() == [0, 1, 0, 2, 3, 0]
    empty = torch.empty(0, dtype=torch.int64).cuda()
    with lookup.Lookup.from_device(empty, torch.empty(0, dtype=torch.int32).cuda(), 32) as lk:
        assert lk.info.n_kmers == 0
        assert lk.values(torch.tensor([7], dtype=torch.int64).cuda()).cpu().tolist() == [0]
        assert lk.stream(torch.from_numpy(np.frombuffer(b"ACGT" * 20, dtype=np.uint8).copy()).cuda()).sum().item() == 0


def test_meryl_lookup_existence_cli(native_lib, oracle_lib, tmp_path):
    """`meryl-lookup -existence -sequence X -mers A B` (src/meryl-lookup/existence.C:48-132): per sequence its k-mers and how
    many of them each database holds -- against a dictionary of the oracle's counts; -min filters the table."""
    import subprocess
    from meryl_amd import build
    k = 21
    reads_a = [r for r in oracle_lib.synth_reads(51, 60_000, 0, 3000).tobytes().decode().split(".") if r]
    reads_b = [r for r in oracle_lib.synth_reads(52, 60_000, 0, 3000).tobytes().decode().split(".") if r]
    fa, fb = tmp_path / "a.fa", tmp_path / "b.fq"
    fa.write_text("".join(">a%d some description\n%s\n%s\n" % (i, r[:60], r[60:]) for i, r in enumerate(reads_a)))
    fb.write_text("".join("@b%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)) for i, r in enumerate(reads_b)))
    meryl = build.build_cli()
    lookup = build.build_lookup_cli()
    for f, dbn in ((fa, "A"), (fb, "B")):
        subprocess.run([meryl, "-Q", "k=%d" % k, "memory=1", "count", str(f), "output", str(tmp_path / (dbn + ".meryl"))], check=True)
    tables = []
    for reads in (reads_a, reads_b):
        _, lo, cn, _ = oracle_lib.count_brute(".".join(reads) + ".", k)
        tables.append(dict(zip((int(x) for x in lo), (int(c) for c in cn))))
    query = reads_a[:40] + reads_b[:40] + ["ACGTNACGT", "A" * 30]
    q = tmp_path / "q.fa"
    q.write_text("".join(">q%d\n%s\n" % (i, r) for i, r in enumerate(query)))
    for vmin in (0, 3):
        args = [lookup, "-existence", "-sequence", str(q), "-mers", str(tmp_path / "A.meryl"), str(tmp_path / "B.meryl")]
        if vmin:
            args += ["-min", str(vmin)]
        p = subprocess.run(args, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        rows = [l.split("\t") for l in p.stdout.strip().split("\n")]
        assert len(rows) == len(query)
        for i, (row, seq) in enumerate(zip(rows, query)):
            _, klo = oracle_lib.enumerate_kmers(seq + ".", k, 0)
            want = [str(len(klo))]
            for t in tables:
                kept = {key for key, c in t.items() if c >= max(vmin, 1)}
                want += [str(len(kept)), str(sum(1 for x in klo if int(x) in kept))]
            assert row == ["q%d" % i] + want, (i, row, want)



@pytest.mark.parametrize("k,w_prefix,label_size,n", [(31, 6, 0, 2_600_000), (21, 18, 0, 300_000), (51, 8, 9, 400_000), (16, 12, 0, 50_000), (4, 6, 0, 200)])
def test_device_block_decoder_equals_host_decoder(native_lib, tmp_path, monkeypatch, k, w_prefix, label_size, n):
    """mgc_decode.hip against the host reader: random distinct k-mers + values -> database (device encoder) -> loaded back through
    the DEVICE decoder (the default of mgc_lookup_load: raw file bytes uploaded, one thread per block) and through the host
    decoder (MGC_DECODE_HOST=1).  The first case puts 2.6 M k-mers into ONE block (a stuffedBits object of several 16 MiB
    sub-blocks); k = 51 carries labels (skipped by both); k = 4 has blocks of a handful of k-mers and empty ones."""
    import torch
    from meryl_amd import count, db, lookup
    rng = np.random.default_rng(k * 7 + w_prefix)
    bits = 2 * k
    if k == 31:                                               # everything in prefix 0: one huge block
        lo = np.unique(rng.integers(0, 1 << (bits - w_prefix), n, dtype=np.uint64))
        hi = np.zeros_like(lo)
    elif bits <= 64:
        lo = np.unique(rng.integers(0, 1 << bits, n, dtype=np.uint64) if bits < 64 else rng.integers(0, 1 << 63, n, dtype=np.uint64))
        hi = np.zeros_like(lo)
    else:
        hi = rng.integers(0, 1 << (bits - 64), n, dtype=np.uint64)
        lo = rng.integers(0, 1 << 63, n, dtype=np.uint64)
        order = np.lexsort((lo, hi))
        hi, lo = hi[order], lo[order]
        keep = np.ones(n, bool); keep[1:] = (hi[1:] != hi[:-1]) | (lo[1:] != lo[:-1])
        hi, lo = hi[keep], lo[keep]
    vals = rng.integers(1, 0xFFFFFFFF, lo.size, dtype=np.uint64).astype(np.uint32)
    vals[::5] = 1
    keys = _keys_tensor(torch, lo, hi, k)
    cnts = torch.from_numpy(vals.view(np.int32).copy()).cuda()
    path = str(tmp_path / "db")
    st = count.DbStream(path, k, w_prefix, label_size, 0x155 if label_size else 0, host_threads=4)
    st.write(keys, cnts, 0, 1 << w_prefix)
    st.close()
    for host in ("0", "1"):
        monkeypatch.setenv("MGC_DECODE_HOST", host)
        lk = lookup.Lookup.load(path)
        assert lk.info.n_kmers == lo.size == lk.info.n_kmers_in_db
        got = lk.values(keys).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, vals), host
        lk2 = lookup.Lookup.load(path, min_value=2)           # the value filter on top of either decoder
        assert lk2.info.n_kmers == int((vals >= 2).sum())
    r = db.Reader(path)
    rlo, rhi, rcn = r.read_all()
    r.close()
    assert np.array_equal(rlo, lo) and np.array_equal(rhi, hi) and np.array_equal(rcn, vals)
