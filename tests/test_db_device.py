"""GPU tests of the device-side database encoder (meryl_amd/csrc/mgc_encode.hip behind mgc_db_stream_*): the bytes
it produces must equal, file by file, what the host encoder (mdb_writer_add_block -- the stand-in for the reference's
merylBlockWriter::addBlock, src/meryl/merylCountArray.C:472-475) writes from the same (k-mer, count) stream.  The
host encoder itself is checked against an independent Python parser and the reference's documented block shape in
tests/test_db.py; byte parity with a genuine meryl database stays unpinned."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dir_bytes(path):
    return {n: open(os.path.join(path, n), "rb").read() for n in sorted(os.listdir(path))}


def _assert_same_dirs(a, b):
    da, db_ = _dir_bytes(a), _dir_bytes(b)
    assert sorted(da) == sorted(db_) and len(da) == 129
    for n in da:
        if da[n] != db_[n]:
            x, y = np.frombuffer(da[n], np.uint8), np.frombuffer(db_[n], np.uint8)
            m = min(x.size, y.size)
            first = int(np.argmax(x[:m] != y[:m])) if np.any(x[:m] != y[:m]) else m
            raise AssertionError("%s differs: sizes %d / %d, first difference at byte %d" % (n, x.size, y.size, first))


def _random_keys(rng, k, n):
    """n (or fewer) distinct ascending k-mers as (lo, hi) uint64 arrays"""
    bits = 2 * k
    if bits <= 64:
        lo = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
        if bits < 64:
            lo &= np.uint64((1 << bits) - 1)
        lo = np.unique(lo)
        return lo, np.zeros(lo.size, np.uint64)
    lo = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    hi = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    if bits < 128:
        hi &= np.uint64((1 << (bits - 64)) - 1)
    order = np.lexsort((lo, hi))
    lo, hi = lo[order], hi[order]
    keep = np.ones(lo.size, bool)
    keep[1:] = (lo[1:] != lo[:-1]) | (hi[1:] != hi[:-1])
    return lo[keep], hi[keep]


def _prefixes(lo, hi, k, wp):
    w_data = 2 * k - wp
    if w_data >= 64:
        return (hi >> np.uint64(w_data - 64)) if w_data > 64 else hi.copy()
    p = lo >> np.uint64(w_data)
    if 2 * k > 64:
        p = p | (hi << np.uint64(64 - w_data))
    return p


def _host_write(path, lo, hi, cn, k, wp, label_size=0, label=0, part=0, n_parts=1, p0=0, p1=None):
    from meryl_amd import db
    w_data = 2 * k - wp
    p1 = (1 << wp) if p1 is None else p1
    pref = _prefixes(lo, hi, k, wp)
    starts = np.searchsorted(pref, np.arange(p0, p1 + 1, dtype=np.uint64))
    mlo = np.uint64((1 << min(w_data, 64)) - 1) if w_data < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    mhi = np.uint64((1 << (w_data - 64)) - 1) if w_data > 64 else np.uint64(0)
    w = db.Writer(path, k, wp, label_size, part, n_parts)
    for i, p in enumerate(range(p0, p1)):
        s, e = int(starts[i]), int(starts[i + 1])
        w.add_block(p, lo[s:e] & mlo, cn[s:e], (hi[s:e] & mhi) if w_data > 64 else None, label=label)
    w.close()


def _to_device(lo, hi, cn, k):
    import torch
    if k > 32:
        keys = torch.from_numpy(np.stack([lo, hi], axis=1).view(np.int64).copy()).cuda()
    else:
        keys = torch.from_numpy(lo.view(np.int64).copy()).cuda()
    return keys, torch.from_numpy(cn.view(np.int32).copy()).cuda()


def _counts(rng, n):
    cn = rng.integers(1, 60, n).astype(np.uint32)
    big = rng.random(n) < 0.01
    cn[big] = rng.integers(1000, 5000, int(big.sum())).astype(np.uint32)          # around the histogram's 1024-bin split
    huge = rng.random(n) < 0.001
    cn[huge] = rng.integers(1 << 20, 0xFFFFFFFF, int(huge.sum()), dtype=np.uint64).astype(np.uint32)
    return cn


@pytest.mark.parametrize("k,wp,n,label_size", [
    (21, 10, 200_000, 0), (21, 18, 300_000, 0), (16, 12, 150_000, 0), (8, 10, 70_000, 0), (5, 6, 1024, 0), (3, 6, 64, 0),
    (31, 10, 200_000, 7), (32, 11, 100_000, 0), (33, 6, 50_000, 0), (51, 10, 200_000, 0), (51, 20, 100_000, 64),
    (64, 12, 100_000, 3), (40, 16, 80_000, 0), (21, 10, 0, 0), (21, 10, 1, 0), (36, 8, 100_000, 0)])
def test_device_encoder_matches_host_writer(tmp_path, native_lib, k, wp, n, label_size):
    from meryl_amd import count
    rng = np.random.default_rng(k * 100 + wp)
    lo, hi = _random_keys(rng, k, n) if n else (np.zeros(0, np.uint64), np.zeros(0, np.uint64))
    cn = _counts(rng, lo.size)
    label = 0xDEADBEEFCAFEF00D
    _host_write(str(tmp_path / "host"), lo, hi, cn, k, wp, label_size, label)
    keys, cnts = _to_device(lo, hi, cn, k)
    s = count.DbStream(str(tmp_path / "dev"), k, wp, label_size, label, host_threads=5)
    s.write(keys, cnts, 0, 1 << wp)
    prof = s.close()
    assert prof["n_kmers"] == lo.size and prof["n_blocks"] == 1 << wp
    _assert_same_dirs(str(tmp_path / "host"), str(tmp_path / "dev"))


@pytest.mark.parametrize("k,wp,n", [(31, 6, 5_000_000), (21, 6, 7_000_000), (51, 6, 3_000_000)])
def test_device_encoder_huge_blocks(tmp_path, native_lib, k, wp, n):
    """Blocks larger than a pinned copy buffer (32 MiB) and than one stuffedBits sub-block (16 MiB): all k-mers in a few
    one prefix, so that pieces continue a block and the dump has several sub-blocks."""
    from meryl_amd import count
    rng = np.random.default_rng(k)
    lo, hi = _random_keys(rng, k, n)
    # squeeze everything into the first file's single block
    w_data = 2 * k - wp
    if k > 32:
        hi = hi % np.uint64(1 << (w_data - 64))
        order = np.lexsort((lo, hi)); lo, hi = lo[order], hi[order]
        keep = np.ones(lo.size, bool); keep[1:] = (lo[1:] != lo[:-1]) | (hi[1:] != hi[:-1])
        lo, hi = lo[keep], hi[keep]
    else:
        lo = np.unique(lo % np.uint64(1 << w_data))
        hi = np.zeros(lo.size, np.uint64)
    cn = _counts(rng, lo.size)
    _host_write(str(tmp_path / "host"), lo, hi, cn, k, wp)
    size0 = os.path.getsize(str(tmp_path / "host" / "0x000000.merylData"))
    assert size0 > (34 << 20), size0
    keys, cnts = _to_device(lo, hi, cn, k)
    s = count.DbStream(str(tmp_path / "dev"), k, wp, host_threads=3)
    s.write(keys, cnts, 0, 1 << wp)
    s.close()
    _assert_same_dirs(str(tmp_path / "host"), str(tmp_path / "dev"))


@pytest.mark.parametrize("k,wp,cuts,n_parts", [(21, 12, [0, 100, 100, 1777, 4096], 1), (21, 12, [0, 1000, 4096], 2),
                                               (51, 10, [0, 1, 513, 1024], 3)])
def test_device_stream_ranges_and_parts(tmp_path, native_lib, k, wp, cuts, n_parts):
    """Several ascending ranges through one stream (what a sharded count's waves do), and ranges spread over part
    writers + mdb_merge_parts: same bytes as one host writer."""
    from meryl_amd import count, db
    import torch
    rng = np.random.default_rng(wp)
    lo, hi = _random_keys(rng, k, 250_000)
    cn = _counts(rng, lo.size)
    _host_write(str(tmp_path / "host"), lo, hi, cn, k, wp)
    pref = _prefixes(lo, hi, k, wp)
    keys, cnts = _to_device(lo, hi, cn, k)
    out = str(tmp_path / "dev")
    ranges = list(zip(cuts, cuts[1:]))
    per_part = -(-len(ranges) // n_parts)
    for part in range(n_parts):
        s = count.DbStream(out, k, wp, part=part, n_parts=n_parts, host_threads=4)
        for (p0, p1) in ranges[part * per_part:(part + 1) * per_part]:
            a, b = int(np.searchsorted(pref, np.uint64(p0))), int(np.searchsorted(pref, np.uint64(p1))) if p1 < (1 << wp) else lo.size
            s.write(keys[a:b], cnts[a:b], p0, p1)
        s.close()
    if n_parts > 1:
        db.merge_parts(out, n_parts)
    _assert_same_dirs(str(tmp_path / "host"), out)
    # a range whose keys do not belong to it is refused, not written
    s = count.DbStream(str(tmp_path / "bad"), k, wp)
    from meryl_amd import capi
    with pytest.raises(capi.MgcError):
        s.write(keys, cnts, 0, 5)
        s.sync()
    with pytest.raises(capi.MgcError):
        s.close()
    torch.cuda.synchronize()


@pytest.mark.parametrize("k,label_size", [(21, 0), (31, 0), (51, 9), (16, 0)])
def test_session_database_device_equals_host_finish(tmp_path, native_lib, oracle_lib, k, label_size):
    """mgc_write_database (device-encoded) == the streamed mgc_finish_labelled callbacks fed to the host writer, and both hold
    the oracle's counts; `meryl print` reads the result back."""
    import torch
    from meryl_amd import capi, count, db
    bases = oracle_lib.synth_reads(11, 200_000, 0, 40_000, 150, 5000, 100)
    cfg = capi.configure(k, bases.size, 1 << 30, label_size=label_size, label=0x2A5)
    d = torch.from_numpy(bases).cuda()
    with count.Session(cfg, 0) as s:
        s.push_bases_device(d)
        s.count()
        prof = s.write_database(str(tmp_path / "dev"), 6)
        w = db.Writer(str(tmp_path / "host"), k, cfg.w_prefix, label_size)
        lock_free_blocks = []

        def cb(prefix, n, slo, cnt, shi):
            lock_free_blocks.append((prefix, slo, cnt, shi))
        s.finish(cb, host_threads=4)
        info = s.info()
    # the callbacks of different files arrive concurrently: feed the writer in prefix order afterwards
    lock_free_blocks.sort(key=lambda b: b[0])
    assert [b[0] for b in lock_free_blocks] == list(range(1 << cfg.w_prefix))
    for prefix, slo, cnt, shi in lock_free_blocks:
        w.add_block(prefix, slo, cnt, shi if k > 32 else None, label=0x2A5)
    w.close()
    assert prof["n_kmers"] == info.n_distinct
    _assert_same_dirs(str(tmp_path / "host"), str(tmp_path / "dev"))
    hi_w, lo_w, cn_w, _ = oracle_lib.count_threaded(bases.tobytes(), k, cfg.w_prefix, threads=2)
    r = db.Reader(str(tmp_path / "dev"))
    lo, hi, cn, lb = r.read_all(labels=True)
    assert np.array_equal(lo, lo_w) and np.array_equal(hi, hi_w) and np.array_equal(cn, cn_w)
    assert r.info.label_size == label_size and (label_size == 0 or np.all(lb == (0x2A5 & ((1 << label_size) - 1))))
    r.close()


def test_forced_sharded_single_rank_database_equals_unsharded(tmp_path, native_lib, oracle_lib):
    """The multi-GPU code path with ONE rank (RCCL group of one): partition -> waves -> owner-side count -> device-encoded
    part -> (no merge needed) must write the same bytes as the single-GPU session."""
    import socket
    import torch
    import torch.distributed as dist
    from meryl_amd import capi, count
    k = 21
    bases = oracle_lib.synth_reads(5, 400_000, 0, 80_000, 150, 5000, 100)
    cfg = capi.configure(k, 10_000_000_000, 64 << 30)                      # wPrefix 18: many empty blocks at this size
    d = torch.from_numpy(bases).cuda()
    with count.Session(cfg, 0) as s:
        s.push_bases_device(d)
        s.count()
        s.write_database(str(tmp_path / "one"), 8)
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for bits in ("", "9"):                                             # the 8-GPU granularity too
            if bits:
                os.environ["MGC_SHARD_BITS"] = bits
            out = str(tmp_path / ("sharded" + bits))
            info = dict(path=out, w_prefix=cfg.w_prefix, host_threads=8)
            count.count_sharded(d, k, db=info, keep_result=False)
            _assert_same_dirs(str(tmp_path / "one"), out)
            assert info["profile"]["n_blocks"] == 1 << cfg.w_prefix
    finally:
        os.environ.pop("MGC_SHARD_BITS", None)
        count.release_cached_sessions()
        dist.destroy_process_group()


@pytest.mark.parametrize("k,compress,label_size,ranks", [(21, 0, 0, (1, 2, 3)), (51, 0, 7, (4,)), (31, 1, 0, (2, 5)), (16, 0, 0, (8,)), (10, 0, 0, (3,))])
def test_count_node_virtual_ranks_database_equals_single_device(tmp_path, native_lib, oracle_lib, k, compress, label_size, ranks):
    """mgc_count_node -- the in-process node count: per-rank extraction, owner PULLS over peer copies in waves, owner-side
    count, device-encoded parts, stitch -- with several ranks placed on the one GPU of the test box: the 129 files
    must be the bytes a single session writes for the concatenated reads (which holds the oracle's counts).  Rank
    slices are cut at read boundaries and deliberately unequal (one rank gets no reads at all when there are >= 3)."""
    import torch
    from meryl_amd import capi, count, db
    n_reads, rl = 60_000, (3000 if compress else 150)
    if compress:
        n_reads = 3000
    bases = oracle_lib.synth_reads(21, 300_000, 0, n_reads, rl, 5000, 100)
    cfg = capi.configure(k, 3_000_000_000, 64 << 30, homopoly_compress=compress, label_size=label_size, label=0x55)
    d = torch.from_numpy(bases).cuda()
    with count.Session(cfg, 0) as s:
        s.push_bases_device(d)
        s.count()
        s.write_database(str(tmp_path / "one"), 8)
        info = s.info()
    rec = rl + 1
    for n in ranks:
        w = np.array([1.0 + (i * 7) % 5 for i in range(n)])
        if n >= 3:
            w[1] = 0.0                                                     # a rank without reads still owns a range
        cuts = np.concatenate([[0], np.floor(np.cumsum(w) / w.sum() * n_reads).astype(np.int64)]) * rec
        cuts[-1] = d.numel()
        slices = [d[int(cuts[i]):int(cuts[i + 1])] for i in range(n)]
        out = str(tmp_path / ("node%d" % n))
        prof = count.count_node(cfg, slices, out, devices=[0] * n, host_threads=4)
        _assert_same_dirs(str(tmp_path / "one"), out)
        assert prof["n_ranks"] == n and prof["n_distinct"] == info.n_distinct and prof["n_instances"] == info.n_instances
        assert not [f for f in os.listdir(out) if "part" in f.lower()]
    text = bases.tobytes()
    if compress:
        text = oracle_lib.compress_stream(text)
    hi_w, lo_w, cn_w, _ = oracle_lib.count_threaded(text, k, cfg.w_prefix, threads=2)
    r = db.Reader(str(tmp_path / ("node%d" % ranks[-1])))
    lo, hi, cn, _ = r.read_all(labels=True)
    r.close()
    assert np.array_equal(lo, lo_w) and np.array_equal(hi, hi_w) and np.array_equal(cn, cn_w)


def test_count_node_rejects_bad_arguments(tmp_path, native_lib):
    import torch
    from meryl_amd import capi, count
    d = torch.zeros(1000, dtype=torch.uint8, device="cuda") + 65
    cfg = capi.configure(21, 1000, 1 << 30)
    with pytest.raises(RuntimeError, match="device"):
        count.count_node(cfg, [d], str(tmp_path / "x"), devices=[99])
    cfg2 = capi.configure(21, 1000, 1 << 30, count_suffix="ACG")
    with pytest.raises(RuntimeError, match="count-suffix"):
        count.count_node(cfg2, [d], str(tmp_path / "y"), devices=[0])
    cfg3 = capi.configure(4, 1000, 1 << 30)                                # 8 bits of k-mer: at most 256 ranges to route
    with pytest.raises(RuntimeError, match="ranges of the k-mer space"):
        count.count_node(cfg3, [d] * 257, str(tmp_path / "z"), devices=[0] * 257)


@pytest.mark.gpu
@pytest.mark.parametrize("k,compress,label_size,ranks,batch,budget", [(21, 0, 0, 4, 350_000, None), (51, 0, 8, 4, 300_000, 1), (31, 1, 0, 3, 250_000, 1),
                                                                      (21, 0, 0, 1, 700_000, 1), (51, 0, 8, 2, 10**9, None)])
def test_count_node_batches_park_waves_and_merge_once(tmp_path, native_lib, oracle_lib, monkeypatch, k, compress, label_size, ranks, batch, budget):
    """mgc_count_node_batched: the node count in BATCHES -- the routing plan from one histogram of all reads, then per batch
    partition -> pulls -> owner count, every counted wave parked in the owner's run store (budget = 1 byte: in pinned host
    DRAM; None: in HBM), one merge per owner into its part when the last batch is done.  >= 4 batches x 4 virtual ranks on the
    one GPU, k = 51 with a label, `compress`, batch cuts inside reads (k-1 overlap), one rank without reads: the 129 files
    must be the single session's, byte for byte.  batch = 10^9: one batch, the waves stream to the writer as before."""
    import torch
    from meryl_amd import capi, count
    if budget is not None:
        monkeypatch.setenv("MGC_OOC_BUDGET", str(budget))
    monkeypatch.setenv("MGC_OOC_CHUNK", "600000")
    n_reads, rl = (2000, 3000) if compress else (40_000, 150)
    bases = oracle_lib.synth_reads(23, 300_000, 0, n_reads, rl, 5000, 100)
    cfg = capi.configure(k, 3_000_000_000, 64 << 30, homopoly_compress=compress, label_size=label_size, label=0x33)
    d = torch.from_numpy(bases).cuda()
    with count.Session(cfg, 0) as s:
        s.push_bases_device(d)
        s.count()
        s.write_database(str(tmp_path / "one"), 8)
        info = s.info()
    rec = rl + 1
    w = np.array([1.0 + (i * 7) % 5 for i in range(ranks)])
    if ranks >= 3:
        w[1] = 0.0
    cuts = np.concatenate([[0], np.floor(np.cumsum(w) / w.sum() * n_reads).astype(np.int64)]) * rec
    cuts[-1] = d.numel()
    slices = [d[int(cuts[i]):int(cuts[i + 1])] for i in range(ranks)]
    out = str(tmp_path / "node")
    prof = count.count_node(cfg, slices, out, devices=[0] * ranks, host_threads=4, batch_bases=batch)
    _assert_same_dirs(str(tmp_path / "one"), out)
    assert prof["n_distinct"] == info.n_distinct and prof["n_instances"] == info.n_instances
    if batch < 10**9:
        assert prof["n_batches"] >= 4
        assert (prof["n_host_runs"] > 0 and prof["host_run_bytes"] > 0) if budget == 1 else prof["n_host_runs"] == 0
    else:
        assert prof["n_batches"] == 1 and prof["n_host_runs"] == 0
