"""World-size-2/3 gloo tests: a sharded count that ENDS IN THE DATABASE.  `count_sharded(..., db=...)` with CPU stand-ins
for the HIP operators (built on the oracle -- test infrastructure) and a host-writer sink: every rank writes its part of
the directory, rank 0 stitches the parts, and the 64+64+1 files must be byte-identical to the directory one writer
produces from a single count of all reads (the reference dumps its 64 files in one pass,
src/meryl/merylOp-countThreads.C:452-464).  What is under test is the host logic: bucket-granular cuts that fall inside
a file, wave-by-wave prefix ranges (empty ones included), part files, merge."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class HostSink:
    """stand-in for the device-encoding stream: the host writer (mdb_writer_*) fed block by block"""

    def __init__(self, path, k, w_prefix, label_size, label, part, n_parts, host_threads):
        from meryl_amd import db
        self.w = db.Writer(path, k, w_prefix, label_size, part, n_parts)
        self.k, self.wp, self.label = k, w_prefix, label

    def write(self, keys, counts, pb, pe):
        a = keys.numpy().view(np.uint64)
        c = counts.numpy().view(np.uint32)
        w_data = 2 * self.k - self.wp
        pref = (a >> np.uint64(w_data)).astype(np.int64)
        assert a.size == 0 or (pref.min() >= pb and pref.max() < pe)
        starts = np.searchsorted(pref, np.arange(pb, pe + 1))
        mask = np.uint64((1 << w_data) - 1)
        for i, p in enumerate(range(pb, pe)):
            s, e = starts[i], starts[i + 1]
            self.w.add_block(p, a[s:e] & mask, c[s:e], label=self.label)

    def close(self):
        self.w.close()
        return {}


class HostRuns:
    """stand-in for the run store (mgc_runs_*): parked (k-mer, count) runs, merged once into the sink by prefix range"""

    def __init__(self, k, w_prefix, device_budget):
        self.k, self.wp, self.runs, self.merged = k, w_prefix, [], 0

    def add(self, keys, counts):
        self.runs.append((keys.numpy().view(np.uint64).copy(), counts.numpy().view(np.uint32).astype(np.uint64)))

    def write(self, sink, pb, pe):
        import torch
        allk = np.concatenate([r[0] for r in self.runs]) if self.runs else np.zeros(0, np.uint64)
        allc = np.concatenate([r[1] for r in self.runs]) if self.runs else np.zeros(0, np.uint64)
        u, inv = np.unique(allk, return_inverse=True)
        c = np.zeros(u.size, dtype=np.uint64)
        np.add.at(c, inv, allc)
        pref = (u >> np.uint64(2 * self.k - self.wp)).astype(np.int64)
        m = (pref >= pb) & (pref < pe)
        self.merged += int(m.sum())
        sink.write(torch.from_numpy(u[m].view(np.int64).copy()), torch.from_numpy((c[m] & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32).copy()), pb, pe)

    def profile(self):
        return {"n_merged": self.merged, "n_runs": len(self.runs)}

    def close(self):
        pass


def _cpu_ops(oracle):
    import torch

    class CpuOps:
        @staticmethod
        def partition(bases, k_, mode, bucket_bits):
            _, lo = oracle.enumerate_kmers(bases.numpy().tobytes(), k_, mode)
            b = (lo >> np.uint64(2 * k_ - bucket_bits)).astype(np.int64)
            order = np.argsort(b, kind="stable")
            counts = np.bincount(b, minlength=1 << bucket_bits).astype(np.uint64)
            return torch.from_numpy(lo[order].view(np.int64).copy()), counts

        @staticmethod
        def histogram_keep(bases, k_, mode, bucket_bits):
            _, lo = oracle.enumerate_kmers(bases.numpy().tobytes(), k_, mode)
            b = (lo >> np.uint64(2 * k_ - bucket_bits)).astype(np.int64)
            order = np.argsort(b, kind="stable")
            counts = np.bincount(b, minlength=1 << bucket_bits).astype(np.uint64)
            return counts, (lo[order], counts)

        @staticmethod
        def partition_into(token, starts, out):                      # bucket b's k-mers at out[starts[b] : starts[b] + count[b]]
            keys, counts = token
            o = out.numpy().view(np.uint64)
            at = 0
            for b, c in enumerate(counts):
                c = int(c)
                o[int(starts[b]):int(starts[b]) + c] = keys[at:at + c]
                at += c

        @staticmethod
        def count_files(keys, file_counts, k_, mode):
            a = keys.numpy().view(np.uint64)
            u, c = np.unique(a, return_counts=True)
            return torch.from_numpy(u.view(np.int64).copy()), torch.from_numpy(c.astype(np.int32))

        @staticmethod
        def empty_keys(n, like):
            return torch.empty(int(n), dtype=torch.int64)

        @staticmethod
        def histogram(bases, k_, mode, bucket_bits):
            _, lo = oracle.enumerate_kmers(bases.numpy().tobytes(), k_, mode)
            return np.bincount((lo >> np.uint64(2 * k_ - bucket_bits)).astype(np.int64), minlength=1 << bucket_bits).astype(np.uint64)

        open_sink = HostSink
        open_runs = HostRuns

    return CpuOps


def _worker(rank, world, port, k, wp, path, label_size, label, batch_bases=None):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import oracle
    from meryl_amd import count
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bases = oracle.synth_reads(7, 20000, rank * 250, 250, 100, 5000, 100)
        db = dict(path=path, w_prefix=wp, label_size=label_size, label=label)
        count.count_sharded(torch.from_numpy(bases), k, 0, ops=_cpu_ops(oracle), db=db, keep_result=(rank % 2 == 0), batch_bases=batch_bases)
        if batch_bases:
            assert db["n_batches"] >= 3
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k,wp,label_size,batch", [(2, 21, 10, 0, None), (3, 16, 8, 0, None), (2, 31, 12, 5, None), (3, 21, 6, 0, None),
                                                         (2, 21, 10, 0, 7000), (3, 31, 12, 5, 9001)])
def test_sharded_count_writes_identical_database(tmp_path, oracle_lib, native_lib, world, k, wp, label_size, batch):
    import torch.multiprocessing as mp
    from meryl_amd import db
    label = 0x13
    path = str(tmp_path / "sharded")
    ctx = mp.get_context("spawn")
    port = _free_port()
    # batch: every rank's reads (25,250 bases) go through the exchange in slices of that many bases cut anywhere (k-1 overlap);
    # the counted waves wait in the owner's run store and are merged once -- same 129 files
    procs = [ctx.Process(target=_worker, args=(r, world, port, k, wp, path, label_size, label, batch)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    # the single-writer database of all reads
    all_bases = b"".join(oracle_lib.synth_reads(7, 20000, r * 250, 250, 100, 5000, 100).tobytes() for r in range(world))
    _, wlo, wcn, _ = oracle_lib.count_brute(all_bases, k)
    one = str(tmp_path / "single")
    w = db.Writer(one, k, wp, label_size)
    w_data = 2 * k - wp
    pref = (wlo >> np.uint64(w_data)).astype(np.int64)
    starts = np.searchsorted(pref, np.arange(0, (1 << wp) + 1))
    for p in range(1 << wp):
        s, e = starts[p], starts[p + 1]
        w.add_block(p, wlo[s:e] & np.uint64((1 << w_data) - 1), wcn[s:e], label=label)
    w.close()
    names = sorted(os.listdir(one))
    assert sorted(os.listdir(path)) == names and len(names) == 129
    for n in names:
        assert open(os.path.join(one, n), "rb").read() == open(os.path.join(path, n), "rb").read(), n
    # with 3 ranks and 6 prefix bits (or finer buckets than files) at least one cut falls inside a file when it can
    r = db.Reader(path)
    lo, hi, cn = r.read_all()
    assert np.array_equal(lo, wlo) and np.array_equal(cn, wcn)
    r.close()
