"""World-size-2 (and 3) gloo tests of the multi-GPU routing logic on CPU:
`meryl_amd.count.count_sharded` with CPU stand-ins injected for the three HIP
operators (the stand-ins are built on the oracle -- test infrastructure -- so
what is under test is the host logic: balanced contiguous file ranges, the
exchange plan, all_to_all_single split sizes, the owner-side sort width).
The concatenation of the ranks' results must equal a single-process count of
all reads."""
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


def _worker(rank, world, port, k, seed, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import oracle
    from meryl_amd import count

    class CpuOps:                                        # stand-ins for the HIP operators
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
            # stand-in for mgc_count_partitioned: the keys arrive file-major, each file's pieces back to back
            a = keys.numpy().view(np.uint64)
            fc = np.asarray(file_counts, dtype=np.uint64)
            bits = int(fc.size).bit_length() - 1
            assert fc.size == 1 << bits and int(fc.sum()) == a.size
            f = (a >> np.uint64(2 * k_ - bits)).astype(np.int64)
            assert np.all(f[1:] >= f[:-1])                                    # bucket-major layout
            assert np.array_equal(np.bincount(f, minlength=fc.size).astype(np.uint64), fc)
            u, c = np.unique(a, return_counts=True)
            return torch.from_numpy(u.view(np.int64).copy()), torch.from_numpy(c.astype(np.int32))

        @staticmethod
        def empty_keys(n, like):
            return torch.empty(int(n), dtype=torch.int64)

    dist.init_process_group("gloo", rank=rank, world_size=world)
    if seed % 2 == 1:
        count.EXCHANGE_CHUNK = 97                        # force many exchange rounds with ragged tails
    try:
        reads_per_rank = 300
        bases = oracle.synth_reads(seed, 20000, rank * reads_per_rank, reads_per_rank, 100, 5000, 100)
        uniq, cnts, (f0, f1, bits) = count.count_sharded(torch.from_numpy(bases), k, 0, ops=CpuOps)
        q.put((rank, uniq.numpy().view(np.uint64).copy(), cnts.numpy().view(np.uint32).copy(), f0, f1, bits))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k", [(2, 21), (3, 16), (2, 31), (1, 21)])
def test_count_sharded_equals_single(oracle_lib, world, k):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, k, 5, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=180) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # contiguous, disjoint, complete bucket ranges in rank order
    bits = got[0][5]
    assert all(g[5] == bits for g in got) and bits == 6 + (world - 1).bit_length()
    assert got[0][3] == 0 and got[-1][4] == 1 << bits
    for a, b in zip(got, got[1:]):
        assert a[4] == b[3]
    keys = np.concatenate([g[1] for g in got])
    cnts = np.concatenate([g[2] for g in got])
    for g in got:                                        # every key sits in its owner's bucket range
        f = (g[1] >> np.uint64(2 * k - bits)).astype(np.int64)
        assert np.all((f >= g[3]) & (f < g[4]))
    all_bases = b"".join(oracle_lib.synth_reads(5, 20000, r * 300, 300, 100, 5000, 100).tobytes() for r in range(world))
    _, wlo, wcn, _ = oracle_lib.count_brute(all_bases, k)
    assert np.array_equal(keys, wlo) and np.array_equal(cnts, wcn)


def test_balanced_file_ranges_properties():
    sys.path.insert(0, ROOT)
    from meryl_amd import count
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 4, 8, 64):
        for _ in range(20):
            fc = rng.integers(0, 1000, 64) * (rng.random(64) < 0.8)
            cuts = count.balanced_file_ranges(fc, world)
            assert cuts[0] == 0 and cuts[-1] == 64 and len(cuts) == world + 1
            assert all(b > a for a, b in zip(cuts, cuts[1:]))           # every rank owns >= 1 file
            send = count.exchange_plan(fc, cuts)
            assert sum(send) == fc.sum()
    # skewed like canonical k-mers (A/C-heavy low files): the cut is by weight, not by file count
    fc = np.concatenate([np.full(16, 700), np.full(16, 200), np.full(16, 80), np.full(16, 20)])
    cuts = count.balanced_file_ranges(fc, 4)
    loads = [fc[a:b].sum() for a, b in zip(cuts, cuts[1:])]
    assert max(loads) <= 1.35 * fc.sum() / 4
    with pytest.raises(ValueError):
        count.balanced_file_ranges(np.ones(64), 65)
    assert count.owned_sort_bits(21, 0, 64) == 42 and count.owned_sort_bits(21, 5, 6) == 36
