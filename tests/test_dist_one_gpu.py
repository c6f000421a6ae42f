"""The Python plan of the sharded count and the REAL HIP operators at world = 2 and 3 on ONE GPU: RCCL
refuses two ranks on one device, gloo does not, so every rank is a process on device 0 with a gloo group and
`count_sharded(ops=HipOps, db=...)` -- histogram, balanced cuts, partition with explicit bucket starts, bucket-major waves
(host-staged here: count.exchange_segments), owner-side grouping passes + sub-bucket count (mgc_count_buckets), the
device-encoding database stream of every rank's part, the stitch.  The 64 + 64 + 1 files must be byte-identical to the database
ONE session writes from all reads.  What the gloo CPU tests cannot see: the HIP operators driven by the plan at N > 1."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _reads(count, rank, n_reads, read_len, device):
    # every rank its own reads of ONE genome (30x over the ranks together): the same k-mers arrive from every rank
    return count.dev_synth_reads(11, 400_000, rank * n_reads, n_reads, read_len, 5000, 100).to(device)


def _worker(rank, world, port, k, wp, path, label_size, label, compress, n_reads, read_len, batch_bases, env):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.update(env)
    import torch
    import torch.distributed as dist
    from meryl_amd import capi, count
    torch.cuda.set_device(0)
    capi.lib()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bases = _reads(count, rank, n_reads, read_len, "cuda:0")
        if compress:
            bases = count.dev_homopoly_compress(bases)
        db = dict(path=path, w_prefix=wp, label_size=label_size, label=label, host_threads=4)
        uniq, cnts, rng = count.count_sharded(bases, k, capi.MODE_CANONICAL, ops=count.HipOps, db=db, keep_result=(rank == 0),
                                              batch_bases=batch_bases)
        bits = int(env["MGC_SHARD_BITS"]) if "MGC_SHARD_BITS" in env else 6 + (world - 1).bit_length()
        assert rng[2] == min(bits, wp)                                               # bucket-granular routing
        if batch_bases:
            assert db["n_batches"] >= 2
        if rank == 0 and not batch_bases:
            assert uniq.shape[0] == db["n_distinct_local"] and uniq.is_cuda
        count.release_cached_sessions()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k,label_size,compress,n_reads,read_len,batch,env", [
    (2, 21, 0, 0, 40_000, 150, None, {}),                                       # the judged k
    (2, 21, 0, 0, 40_000, 150, None, {"MGC_FINISH_MIN_TOP": "16", "MGC_HASH_STREAM": "1"}),   # two-digit owner-side plan, distinct-sized count
    (3, 21, 0, 0, 30_000, 150, None, {}),                                       # three ranks: cuts inside files
    (2, 31, 0, 1, 400, 8_000, None, {}),                                        # k = 31 `compress` on long reads
    (2, 51, 8, 0, 30_000, 150, None, {}),                                       # k = 51 (16-byte keys) with an 8-bit constant label
    (2, 21, 0, 0, 40_000, 150, 1_500_000, {}),                                  # batched: waves parked in the owner's run store
    (2, 21, 0, 0, 40_000, 150, None, {"MGC_SHARD_BITS": "9"}),                     # the granularity of an EIGHT-rank run (512 buckets: no fifteen-bit histogram on the owner side)
    (3, 21, 0, 0, 30_000, 150, None, {"MGC_SHARD_BITS": "9", "MGC_FINISH_MIN_TOP": "14"}),   # ... with the owner's two-digit plan
])
def test_sharded_count_with_hip_operators_on_one_gpu(tmp_path, native_lib, world, k, label_size, compress, n_reads, read_len, batch, env):
    import torch
    import torch.multiprocessing as mp
    from meryl_amd import capi, count
    label = 0xA5 if label_size else 0
    cfg = capi.configure(k, world * n_reads * (read_len + 1), 1 << 30, capi.MODE_CANONICAL, homopoly_compress=compress, label_size=label_size, label=label)
    wp = int(cfg.w_prefix)
    path = str(tmp_path / "sharded")
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, k, wp, path, label_size, label, compress, n_reads, read_len, batch, env))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    # ONE session over all reads, its own database
    torch.cuda.set_device(0)
    all_bases = torch.cat([_reads(count, r, n_reads, read_len, "cuda:0") for r in range(world)])
    assert int(all_bases.numel()) == world * n_reads * (read_len + 1)
    one = str(tmp_path / "single")
    with count.Session(cfg, 0) as s:
        s.push_bases_device(all_bases)
        s.count()
        s.write_database(one, 4)
        n_distinct = s.info().n_distinct
    assert n_distinct > 1000
    names = sorted(os.listdir(one))
    assert sorted(os.listdir(path)) == names and len(names) == 129
    for n in names:
        assert open(os.path.join(one, n), "rb").read() == open(os.path.join(path, n), "rb").read(), n
