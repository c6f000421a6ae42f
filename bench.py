#!/usr/bin/env python3
"""bench.py -- measures BASELINE.json's metric (distinct k-mers counted / sec,
whole job) for the `meryl count` hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R]

A "step" is one full pass of the hot path (per-file histogram -> pack+scatter ->
per-file radix grouping passes on the top bits -> LDS hash-count of every sub-bucket
-> compaction -> block offsets) over one batch of synthetic reads that is already
resident in HBM when the timed region starts.
Default workload = BASELINE.json configs[1]: k=21, 10 Gbp of synthetic 150 bp
reads (30x of a 333,333,334 bp genome, 0.5 % substitutions, 0.01 % N) on one
MI355X.  With --gpus N (launched by torch.distributed.run, one rank per GPU)
every rank brings the same amount of its own reads (weak scaling: 30x of a genome
that grows with N); the 64*N top-bit buckets are cut into contiguous per-rank ranges
and k-mers are routed to their owner in point-to-point waves over RCCL/xGMI while
the owner counts the buckets that have arrived.

Prints ONE JSON line (rank 0), at every N:
  roofline      the kernel family with the largest WALL-CLOCK share of the step (computed from the stage clocks, not asserted: the
                first grouping pass on the judged workload) priced on its algorithmic bytes and its own launch durations (HIP
                events on the stream it is launched on); the other families -- second pass, sub-bucket count, partition,
                histogram -- under `kernels`, the grouping passes also under `sort_pass`; `traffic` is the PMC measurement
                committed under profiles/ for this same workload
  check         untimed sanity check of the counted result (N > 1: reduced over the ranks, rank boundaries included)
  db_write      the counted result -> the 64-file database (N > 1: count_sharded(db=...), every rank its part, stitched;
                its 129 files compared with the database mgc_count_node -- the in-process peer-copy form -- writes)
  e2e           file -> database wall clock of the stand-alone CLI (N > 1: gpus=N, every rank reading its own windows)
  cpu_baseline  the CPU restatement of the reference algorithm (oracle/, kind "port") timed on this box's host cores over a
                bounded sample of the same workload (rank 0)
N > 1 on a box with ONE GPU (MGC_BENCH_ONE_DEVICE=1: every rank is told to use device 0): RCCL refuses two ranks on one
device ("Duplicate GPU detected"), so rank 0 falls back to the peer-copy form with N virtual ranks in one process and says
so in the line (`transport`).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K = 21
GENOME_LEN = 333_333_334
READ_LEN = 150
DEFAULT_READS = 66_666_667          # 10.0 Gbp
SEED = 2
HBM_PEAK_GBS = 8000.0               # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(sample_reads, threads, dev_bases=None, whole=False):
    """The CPU leg: the restatement of the reference's threaded algorithm (oracle/oracle_port.cpp: 2 MiB chunks, spin-locked
    bit-packed prefix buckets with the reference's one/two/three-word add, std::sort, run-length count, 64-file dump) on
    this box's host cores.  It is a RESTATEMENT (kind "port"), never reference meryl itself -- the genuine binary cannot be
    built here (its utility library is an absent submodule).
    whole=True (the default at N = 1): the WHOLE workload, the very bytes the GPU counted, at ONE thread count -- 16, the
    best of 16 / 32 / 64 on this kind of box (profiles/r03_cpu_full.json: 146 / 158 / 165 s; the port gets slower with
    more threads because of the reference's spin-locked buckets); MGC_CPU_THREADS overrides it.
    whole=False: a bounded sample (the first `sample_reads` reads; lower coverage per k-mer than the whole run, which
    favours the CPU's distinct/s) at 16 / 32 / 64 threads, best reported."""
    import oracle
    oracle.build()
    if dev_bases is not None:                                         # the very bytes the GPU counted (first reads of rank 0)
        bases = (dev_bases if whole else dev_bases[:sample_reads * (READ_LEN + 1)]).cpu().numpy()
    else:
        bases = oracle.synth_reads(SEED, GENOME_LEN, 0, sample_reads, READ_LEN, 5000, 100)
    n_reads = bases.size // (READ_LEN + 1)
    cfg = oracle.configure_counting(K, 10_000_000_000, 64 << 30)      # the workload's geometry (wPrefix 18)
    tried = []
    best = None
    if whole:
        counts = [max(1, min(threads, int(os.environ.get("MGC_CPU_THREADS", "16"))))]
    else:
        counts = sorted({max(1, min(threads, t)) for t in (16, 32, 64)})
    for th in counts:
        t0 = time.perf_counter()
        _, nd, ni = oracle.digest_threaded(bases, K, cfg["w_prefix"], oracle.CANONICAL, th)
        dt = time.perf_counter() - t0
        tried.append({"threads": th, "seconds": dt, "distinct_per_s": nd / dt, "instances_per_s": ni / dt})
        if best is None or dt < best[0]:
            best = (dt, th, nd, ni)
    dt, th, nd, ni = best
    out = {
        "value": nd / dt, "unit": "distinct k-mers/s", "cores": th, "kind": "port",
        "kind_note": "a restatement of the reference's threaded count (oracle/oracle_port.cpp), timed here; NOT reference meryl "
                     "(unbuildable in this tree).  Thread count = the best of 16/32/64 measured on this kind of box "
                     "(profiles/r03_cpu_full.json): the restated spin-locked buckets get slower beyond 16 threads",
        "sample": ("the WHOLE workload (%d x %d bp reads, %.2f Gbp), the bytes the GPU counted" % (n_reads, READ_LEN, bases.size / 1e9) if whole else
                   "the first %d x %d bp reads of the workload (%.2f Gbp of the %d bp genome's reads)" % (n_reads, READ_LEN, bases.size / 1e9, GENOME_LEN))
                  + ", k=%d, wPrefix=%d; %d instances, %d distinct in %.2f s (%.3g instances/s) on %d threads" % (K, cfg["w_prefix"], ni, nd, dt, ni / dt, th),
        "whole_workload": bool(whole),
        "instances_per_s": ni / dt, "seconds": dt, "threads_tried": tried, "host_cores": os.cpu_count(),
    }
    if not whole:
        # the WHOLE workload on the port (scripts/cpu_full.py on the GPU box's host, committed under profiles/)
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_cpu_full.json")), reverse=True):
            try:
                out["full_workload"] = json.load(open(f))
                out["full_workload"]["source"] = "profiles/" + os.path.basename(f)
                break
            except (OSError, ValueError):
                continue
    return out


def hbm_model(reads, world, read_len=READ_LEN, k=K, distinct_ratio=0.16):
    """Peak HBM of ONE rank of the sharded count, in bytes, by buffer: asserted against the device's memory BEFORE
    anything is allocated, so that a configuration that cannot fit is refused with a message instead of taking the node down.
    Upper bounds: every read yields read_len - k + 1 k-mers; a rank owns 1/world of all k-mers (the cuts balance instances)."""
    n_bases = reads * (read_len + 1)
    n_kmers = reads * max(0, read_len - k + 1)
    key = 16 if k > 32 else 8
    m = {"bases": n_bases,
         "send_area": key * n_kmers * (world - 1) // max(1, world),      # the other ranks' buckets, compact (partition with explicit starts)
         "inbox": key * n_kmers,                                         # what this rank owns, bucket-major (its own pieces written in place)
         "owner_pass_buffer": 2 * key * n_kmers // 64,                   # the grouping passes' ping-pong: one bucket (as large as a 1-GPU file) + tables
         "result": int((key + 4) * distinct_ratio * 1.06 * n_kmers) if world > 1 else int((key + 4) * distinct_ratio * 1.06 * n_kmers),
         "database_stream": 3 << 30,                                     # encoded images + pinned staging of the part writer
         "histograms_and_tables": 1 << 30}
    m["total"] = sum(m.values())
    return m


def self_launch(n, argv):
    """`python bench.py --gpus N` invoked plainly: start the N ranks under torch.distributed.run (one per GPU, RCCL) and
    relay the ONE JSON line rank 0 prints."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def valid_windows(bases, k, chunk=1 << 30):
    """k-mer windows without a non-ACGT byte, computed independently of the library (torch ops on the device): the
    positions of the invalid bytes cut the stream into runs, a run of length L holds max(0, L - k + 1) windows."""
    import torch
    n = bases.numel()
    bad = []
    for a in range(0, n, chunk):
        b = bases[a:a + chunk] | 0x20
        ok = (b == ord("a")) | (b == ord("c")) | (b == ord("g")) | (b == ord("t"))
        bad.append(torch.nonzero(~ok).flatten() + a)
        del b, ok
    idx = torch.cat([torch.tensor([-1], device=bases.device)] + bad + [torch.tensor([n], device=bases.device)])
    runs = idx[1:] - idx[:-1] - 1
    return int(torch.clamp(runs - (k - 1), min=0).sum().item())


def per_file_sums(keys, cnts, k):
    import torch
    c64 = cnts.to(torch.int64) & 0xFFFFFFFF
    bounds = torch.arange(0, 65, device=keys.device, dtype=torch.int64) << (2 * k - 6)
    cut = torch.searchsorted(keys, bounds)
    csum = torch.cat([torch.zeros(1, dtype=torch.int64, device=keys.device), torch.cumsum(c64, 0)])
    return (csum[cut[1:]] - csum[cut[:-1]]), c64


def result_check(keys, cnts, bases, k, n_instances, file_instances, dist=None):
    """Untimed sanity check of the counted result at the judged size (the parity tests proper are tests/test_gpu_parity.py):
    instances == valid windows of the input, distinct k-mers strictly ascending, counts sum to the instances per file.
    With `dist` the check is reduced over the ranks: windows, instances and per-file totals are summed, and the keys must
    also ascend ACROSS the rank boundaries (rank r's last k-mer below rank r+1's first)."""
    import torch
    out = {}
    want = valid_windows(bases, k)
    per_file, c64 = per_file_sums(keys, cnts, k)
    asc = bool((keys[1:] > keys[:-1]).all().item()) if keys.numel() > 1 else True   # k <= 31: int64 order == uint64 order
    have = int(c64.sum().item())
    ge1 = bool(int(c64.min().item()) >= 1) if keys.numel() else True
    if dist is not None:
        world, rank = dist.get_world_size(), dist.get_rank()
        t = torch.tensor([want, have, int(asc), int(ge1), keys.numel()], dtype=torch.int64, device=keys.device)
        tot = t.clone()
        dist.all_reduce(tot)
        mn = t.clone()
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        want, have, asc, ge1 = int(tot[0]), int(tot[1]), bool(int(mn[2])), bool(int(mn[3]))
        dist.all_reduce(per_file)
        edge = torch.tensor([int(keys[0]) if keys.numel() else -1, int(keys[-1]) if keys.numel() else -1], dtype=torch.int64, device=keys.device)
        edges = [torch.empty_like(edge) for _ in range(world)]
        dist.all_gather(edges, edge)
        last, across = -1, True
        for e in edges:
            if int(e[0]) < 0:
                continue                                    # a rank without k-mers
            across = across and int(e[0]) > last
            last = int(e[1])
        out["keys_ascending_across_rank_boundaries"] = bool(across)
        out["ranks"] = world
        fi = torch.tensor([int(x) for x in file_instances], dtype=torch.int64, device=keys.device)
        dist.all_reduce(fi)
        file_instances = fi.cpu().tolist()
        n_instances = have if n_instances is None else n_instances
    out["valid_windows"] = want
    out["instances_equal_valid_windows"] = bool(have == want and (n_instances is None or n_instances == want))
    out["keys_strictly_ascending"] = asc
    out["sum_counts_equals_instances"] = bool(have == want)
    out["per_file_totals_match"] = bool(per_file.cpu().tolist() == [int(x) for x in file_instances])
    out["distinct_ge_1"] = ge1
    out["ok"] = all(v for kk, v in out.items() if isinstance(v, bool))
    return out


def dir_digest(path):
    """sha256 over the 129 files of a database directory (names + bytes, sorted)"""
    h = hashlib.sha256()
    names = sorted(os.listdir(path))
    for n in names:
        h.update(n.encode())
        with open(os.path.join(path, n), "rb") as f:
            while True:
                b = f.read(1 << 24)
                if not b:
                    break
                h.update(b)
    return h.hexdigest(), len(names), sum(os.path.getsize(os.path.join(path, n)) for n in names)


def shm_dir(prefix):
    import tempfile
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    return tempfile.mkdtemp(prefix=prefix, dir=shm), shm


def e2e_run(bases, reads, threads, gpus=1):
    """File -> database wall clock of the stand-alone CLI: the synthetic reads are written as a FASTQ
    file on tmpfs, `meryl count` reads it, parses it on the device, counts, encodes the blocks on the device and writes
    the 64-file database back to tmpfs.  gpus > 1: `gpus=N` -- every rank reads its own byte windows of the file through
    its own device's link.  Never part of `value`."""
    import re
    import shutil
    import subprocess
    import torch
    from meryl_amd import build
    d, shm = shm_dir("mgc_e2e_")
    st = os.statvfs(shm)
    free = st.f_bavail * st.f_frsize
    rec = READ_LEN * 2 + 7
    use = reads
    need = lambda r: r * rec * 1.45 + (1 << 30)            # FASTQ + database
    while use > 1000 and need(use) > free * 0.8:
        use //= 2
    try:
        fq = os.path.join(d, "reads.fq")
        t0 = time.perf_counter()
        with open(fq, "wb") as f:
            step = 4_000_000
            for a in range(0, use, step):
                n = min(step, use - a)
                r = torch.empty((n, rec), dtype=torch.uint8, device=bases.device)
                r[:, 0] = ord("@"); r[:, 1] = ord("r"); r[:, 2] = 10
                r[:, 3:3 + READ_LEN] = bases[a * (READ_LEN + 1):(a + n) * (READ_LEN + 1)].view(n, READ_LEN + 1)[:, :READ_LEN]
                r[:, 3 + READ_LEN] = 10; r[:, 4 + READ_LEN] = ord("+"); r[:, 5 + READ_LEN] = 10
                r[:, 6 + READ_LEN:6 + 2 * READ_LEN] = ord("I"); r[:, 6 + 2 * READ_LEN] = 10
                f.write(r.cpu().numpy().tobytes())
                del r
        t_gen = time.perf_counter() - t0
        # Two things that belong to THIS harness, not to the path (profiles/r03m_e2e_io.txt, r03n_e2e_*.txt): (1) tmpfs pages are
        # slow the first time another process reads them after they were written (1.3-1.6 GB/s per reader thread against ~10):
        # the file is read once, like a file that sits in the page cache; (2) the driver clears device memory a process has
        # released lazily, and the NEXT process's first large allocation waits for it (2-4 s after this bench's own ~130 GB):
        # the device is left alone for a few seconds first.
        t0 = time.perf_counter()
        subprocess.run("cat %s > /dev/null" % fq, shell=True)
        t_warm = time.perf_counter() - t0
        time.sleep(float(os.environ.get("MGC_E2E_SETTLE_S", "6")))
        dbp = os.path.join(d, "out.meryl")
        cli = build.build_cli()
        cmd = [cli, "-V", "k=%d" % K, "memory=64", "threads=%d" % threads, "n=10000000000"] + (["gpus=%d" % gpus] if gpus > 1 else []) + \
              ["count", fq, "output", dbp]
        t0 = time.perf_counter()
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        wall = time.perf_counter() - t0
        if p.returncode != 0:
            return {"error": "meryl CLI rc=%d: %s" % (p.returncode, p.stderr[-400:])}
        out = {"reads": use, "bases": use * READ_LEN, "fastq_bytes": os.path.getsize(fq), "wall_s": wall,
               "threads": threads, "where": shm, "fastq_generation_s": t_gen, "file_read_once_before_s": t_warm,
               "database_bytes": sum(os.path.getsize(os.path.join(dbp, n)) for n in os.listdir(dbp)),
               "command": " ".join(["meryl"] + cmd[1:-4] + ["count", "reads.fq", "output", "out.meryl"])}
        if os.environ.get("MGC_IO_TRACE"):
            out["io_trace"] = [l for l in p.stderr.splitlines() if l.startswith("[io]")]
        m = re.search(r"count on the device: ([0-9.]+) ms", p.stderr)
        if m:
            out["count_on_device_s"] = float(m.group(1)) / 1e3
        m = re.search(r"TIMING(.*)", p.stderr)
        if m:
            for name, val in re.findall(r"([a-z+_]+)=([0-9.]+)", m.group(1)):
                out[name + ("" if name.endswith("bytes") else "_s")] = float(val)
        if all(x in out for x in ("startup_s", "read+parse+stage_s", "count_s", "encode+write_s")):
            # what the wall clock holds beyond the CLI's own stamps: exec + dynamic loading before main(), and the
            # process exit after the last write (the arena and the pinned rings are not torn down: _exit)
            out["before_main_and_exit_s"] = wall - (out["startup_s"] + out["read+parse+stage_s"] + out["count_s"] + out["encode+write_s"])
        m = re.search(r"(\d+) distinct k-mers", p.stderr)
        if m:
            out["n_distinct"] = int(m.group(1))
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _gz_members(args):
    """one worker of e2e_compressed_run: a byte range of the FASTQ file -> gzip members / BGZF blocks (zlib level 1)"""
    import struct
    import zlib
    path, a, b, kind = args
    with open(path, "rb") as f:
        f.seek(a)
        data = f.read(b - a)
    if kind == "gz":                                     # one gzip member per worker range: gunzip / gzread read the concatenation as ONE stream
        c = zlib.compressobj(1, zlib.DEFLATED, 31)
        return c.compress(data) + c.flush()
    out = []
    for i in range(0, len(data), 0xff00):                # BGZF (SAMv1 4.1): <= 64 KiB members with their compressed size in a 'BC' extra field
        d = data[i:i + 0xff00]
        c = zlib.compressobj(1, zlib.DEFLATED, -15)
        cd = c.compress(d) + c.flush()
        bs = 12 + 6 + len(cd) + 8
        out.append(struct.pack("<BBBBIBBH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6) + b"BC" + struct.pack("<HH", 2, bs - 1) + cd +
                   struct.pack("<II", zlib.crc32(d) & 0xffffffff, len(d)))
    return b"".join(out)


def e2e_compressed_run(bases, reads, threads):
    """real inputs are .fastq.gz.  The same reads (a bounded sample: the single zlib stream of a .gz is host-serial)
    as one gzip stream and as BGZF (bgzip: independent <= 64 KiB members, inflated block-parallel), each through the stand-alone CLI:
    wall clock, and the rate at which uncompressed text arrives on the device (read + inflate + upload + parse).  Never part of `value`."""
    import multiprocessing
    import re
    import shutil
    import subprocess
    import torch
    from meryl_amd import build
    d, shm = shm_dir("mgc_e2ez_")
    rec = READ_LEN * 2 + 7
    try:
        fq = os.path.join(d, "reads.fq")
        with open(fq, "wb") as f:
            step = 2_000_000
            for a in range(0, reads, step):
                n = min(step, reads - a)
                r = torch.empty((n, rec), dtype=torch.uint8, device=bases.device)
                r[:, 0] = ord("@"); r[:, 1] = ord("r"); r[:, 2] = 10
                r[:, 3:3 + READ_LEN] = bases[a * (READ_LEN + 1):(a + n) * (READ_LEN + 1)].view(n, READ_LEN + 1)[:, :READ_LEN]
                r[:, 3 + READ_LEN] = 10; r[:, 4 + READ_LEN] = ord("+"); r[:, 5 + READ_LEN] = 10
                r[:, 6 + READ_LEN:6 + 2 * READ_LEN] = ord("I"); r[:, 6 + 2 * READ_LEN] = 10
                f.write(r.cpu().numpy().tobytes())
                del r
        size = os.path.getsize(fq)
        procs = max(1, min(32, (os.cpu_count() or 2) // 2))
        per = ((size // procs) // rec + 1) * rec                            # whole records per worker
        ranges = [(fq, a, min(a + per, size)) for a in range(0, size, per)]
        out = {"reads": reads, "bases": reads * READ_LEN, "fastq_bytes": size, "threads": threads, "where": shm,
               "note": "a bounded sample of the workload's reads (zlib level 1); `text_GBps` = uncompressed FASTQ bytes / the CLI's read + inflate + upload + parse phase"}
        cli = build.build_cli()
        for kind, name in (("gz", "reads.gz.fq.gz"), ("bgzf", "reads.bgzf.fq.gz")):
            t0 = time.perf_counter()
            path = os.path.join(d, name)
            with multiprocessing.get_context("fork").Pool(procs) as pool, open(path, "wb") as f:
                for piece in pool.imap(_gz_members, [r + (kind,) for r in ranges]):
                    f.write(piece)
                if kind == "bgzf":
                    f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))   # the BGZF end-of-file block
            t_make = time.perf_counter() - t0
            subprocess.run("cat %s > /dev/null" % path, shell=True)
            dbp = os.path.join(d, "out_%s.meryl" % kind)
            cmd = [cli, "-V", "k=%d" % K, "memory=64", "threads=%d" % threads, "n=%d" % (reads * READ_LEN), "count", path, "output", dbp]
            t0 = time.perf_counter()
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            wall = time.perf_counter() - t0
            o = {"file_bytes": os.path.getsize(path), "made_in_s": round(t_make, 2), "wall_s": wall}
            if p.returncode != 0:
                o["error"] = "meryl CLI rc=%d: %s" % (p.returncode, p.stderr[-300:])
            else:
                m = re.search(r"TIMING(.*)", p.stderr)
                if m:
                    for nm, val in re.findall(r"([a-z+_]+)=([0-9.]+)", m.group(1)):
                        o[nm + ("" if nm.endswith("bytes") else "_s")] = float(val)
                rd = o.get("read+parse+stage_s")
                if rd:
                    o["text_GBps"] = size / rd / 1e9
                m = re.search(r"(\d+) distinct k-mers", p.stderr)
                if m:
                    o["n_distinct"] = int(m.group(1))
            out["gz" if kind == "gz" else "bgzf"] = o
            shutil.rmtree(dbp, ignore_errors=True)
            os.remove(path)
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def pmc_traffic(reads, prefix, also=()):
    """HBM bytes per launch of a kernel from the committed PMC run of this same workload (profiles/*_pmc_traffic.json, made by
    scripts/gpu_final.sh: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes, calibrated on known-byte kernels of the same
    access width); None for any other workload -- counters cannot be collected from inside this process.
    The mean over ALL instantiations whose name starts with `prefix`, weighted by their launches (so that it describes the same
    launches as algorithmic_bytes_per_launch: VERDICT r4 item 9); `also`: further name prefixes whose bytes belong to the same
    launch of the library (the retry / streaming kernels of a file's count) -- added to the numerator only."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("reads_per_gpu") != reads:
            continue
        total, launches, names = 0.0, 0, []
        for name, v in d.get("all_kernels", {}).items():
            if v.get("fetch_bytes_per_launch") is None or v.get("write_bytes_per_launch") is None or not v.get("launches"):
                continue
            if name.startswith(prefix):
                total += (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"]
                launches += v["launches"]
                names.append(name[:48])
            elif any(name.startswith(a) for a in also):
                total += (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"]
        if launches:
            return total / launches, "profiles/%s (launch-weighted mean of %d instantiation%s: %s)" % (
                os.path.basename(f), len(names), "" if len(names) == 1 else "s", "; ".join(names))
    return None, None


def roofline_object(prof_acc, ms_per_step, steps, reads, single, where):
    """`roofline` of the line: the kernel family with the LARGEST WALL SHARE of the step -- computed, not asserted.
    A family's wall clock per step: the grouping passes and the front-end kernels run back to back on the session stream, so the sum
    of their launch durations IS their wall clock; the count kernels of consecutive files overlap on two streams, so their wall
    clock is the count stage's own (HIP events on the session stream) minus the packing tail the library times separately
    (`pack_ms`) -- never the sum of their launch durations, which exceeds it.  Every family is priced on its ALGORITHMIC bytes; the
    others stay in `kernels`, the grouping passes also under `sort_pass` (the key earlier rounds' lines carry)."""
    bp = prof_acc["by_pass"]
    if not bp[0]["launches"] and prof_acc["pass_launches"]:              # only the totals were collected
        bp = [{"ms": prof_acc["pass_ms"], "launches": prof_acc["pass_launches"], "keys": prof_acc["pass_keys"],
               "bytes": prof_acc.get("pass_bytes", 16 * prof_acc["pass_keys"])}, {"ms": 0.0, "launches": 0, "keys": 0, "bytes": 0}]
    fams = {}

    def family(key, kernel, wall_ms_total, bytes_total, launches, launch_ms_total, traffic_prefix, also=(), note=None, keys=None):
        if not launches or wall_ms_total <= 0:
            return
        achieved = bytes_total / (launch_ms_total / 1e3) / 1e9        # on the launches' own durations (HIP events on their stream)
        t, src = pmc_traffic(reads, traffic_prefix, also=also) if single else (None, None)
        f = {"kernel": kernel, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
             "traffic": t, "traffic_source": src or ("none: no committed PMC run of this workload" if single else None),
             "launches": launches, "avg_launch_ms": launch_ms_total / launches,
             "algorithmic_bytes_per_launch": bytes_total / launches,
             "wall_ms_per_step": wall_ms_total / steps, "share_of_step": (wall_ms_total / steps) / ms_per_step if single else None}
        if keys:
            f["keys_per_launch"] = keys / launches
        if abs(wall_ms_total - launch_ms_total) > 1e-9:
            # launches that overlap on two streams: the fraction on the family's WALL clock beside the one on the launches' own
            f["frac_on_wall"] = bytes_total / (wall_ms_total / 1e3) / 1e9 / HBM_PEAK_GBS
            f["launch_ms_sum_per_step"] = launch_ms_total / steps
        if note:
            f["note"] = note
        fams[key] = f

    if bp[0]["launches"]:
        # A file's FIRST grouping pass.  Algorithmic bytes = the key bytes it must read and write (the library reports them per
        # launch): 5 + 4 per k-mer from the 5-byte layout (k = 20..23), 8 + 4 when only the output is narrowed, 8 + 8 / 12 + 12 /
        # 16 + 16 for whole keys.
        per_key = bp[0]["bytes"] / max(1, bp[0]["keys"])
        narrowed = bp[0]["bytes"] < 16 * bp[0]["keys"]
        family("first_pass", "radix_group_kernel, first pass of a file (%s)" %
               (("5 B k-mers (u32 + u8 arrays per file) in, 4 B narrowed words out" if per_key < 10 else "8 B k-mers in, 4 B narrowed words out")
                if narrowed else "whole keys in and out"),
               bp[0]["ms"], bp[0]["bytes"], bp[0]["launches"], bp[0]["ms"], "radix_group_kernel<unsigned long long", keys=bp[0]["keys"])
    if bp[1]["launches"]:
        family("second_pass", "radix_group_kernel, second pass of a file", bp[1]["ms"], bp[1]["bytes"], bp[1]["launches"], bp[1]["ms"],
               "radix_group_kernel<unsigned int", keys=bp[1]["keys"])
    fin = prof_acc.get("finish")
    if fin and fin["launches"]:
        stage_wall = prof_acc["stage_ms"][3]                              # the count stage on the session stream (count kernels + packing)
        wall = stage_wall - prof_acc.get("pack_ms", 0.0) if (single and stage_wall) else fin["ms"]
        if wall <= 0:
            wall = fin["ms"]
        family("count", "hash_count_stream_kernel / hash_count_multi_kernel / hash_count_kernel (sub-bucket count: 4 B narrowed keys in, "
               "distinct 4 B suffixes + 4 B counts out), one launch per file on two alternating streams",
               wall, fin["bytes"], fin["launches"], fin["ms"], "hash_count_stream_kernel",
               also=("hash_count_multi_kernel", "hash_count_kernel", "hash_count_huge_kernel"), keys=fin["keys"],
               note="VALU/LDS-issue-bound (DESIGN.md 3.4): the fraction says how far from the HBM roof the kernel runs, not that HBM limits it")
    if single and prof_acc.get("partition_bytes") and prof_acc["stage_ms"][1]:
        family("partition", "kmer_partition_kernel (bases in, k-mers out in the files' layout)", prof_acc["stage_ms"][1], prof_acc["partition_bytes"],
               steps, prof_acc["stage_ms"][1], "kmer_partition_kernel")
    if single and prof_acc.get("hist_bytes") and prof_acc["stage_ms"][0]:
        family("histogram", "kmer_hist_fine_kernel (bases in, fifteen-bit file histogram out)", prof_acc["stage_ms"][0], prof_acc["hist_bytes"],
               steps, prof_acc["stage_ms"][0], "kmer_hist_fine_kernel")
    if not fams:
        return None
    top = max(fams, key=lambda kk: fams[kk]["wall_ms_per_step"])
    out = dict(fams[top])
    out["dominant"] = top
    out["dominant_by"] = "largest wall-clock share of the step among the kernel families (per-stage wall clock; overlapped launches never summed)"
    out["measured"] = "HIP events around every launch, on the stream it is launched on, " + where
    out["kernels"] = {kk: v for kk, v in fams.items() if kk != top}
    if "first_pass" in fams:
        sp = dict(fams["first_pass"])
        if "second_pass" in fams:
            sp["second_pass"] = dict(fams["second_pass"])
        # the same passes priced the way SURVEY 8(d) prices a radix pass (8 + 8 B per k-mer whatever is really moved): comparable
        # with earlier rounds' figures, not a roofline
        if prof_acc["pass_ms"]:
            eq = 16.0 * prof_acc["pass_keys"] / (prof_acc["pass_ms"] / 1e3) / 1e9
            sp["survey_accounting"] = {"bytes_per_key_per_pass": 16, "achieved": eq, "frac": eq / HBM_PEAK_GBS}
        out["sort_pass"] = sp
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=DEFAULT_READS, help="reads per GPU (default = 10 Gbp)")
    ap.add_argument("--cpu-sample-reads", type=int, default=14_000_000, help="reads of the workload the CPU port counts (2.1 Gbp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the untimed file -> database run of the CLI")
    ap.add_argument("--no-check", action="store_true", help="skip the untimed result check")
    ap.add_argument("--no-db", action="store_true", help="skip the untimed database write (N > 1: and the node-count comparison)")
    ap.add_argument("--cpu-sample", action="store_true", help="CPU leg over the bounded sample (three thread counts) instead of the whole workload")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    # Wall-clock guard of the UNTIMED legs (check, database, node-count comparison, CLI end to end, CPU leg): a leg that
    # would start beyond the budget (or could not finish inside it by its estimate) is skipped and named in the line --
    # a SCALE run at N = 8 must not sink under its own untimed tail.  The timed steps are never skipped.
    t_begin = time.perf_counter()
    budget_s = float(os.environ.get("MGC_BENCH_BUDGET_S", "600"))
    legs_s, skipped = {}, {}

    def leg_ok(name, estimate_s):
        spent = time.perf_counter() - t_begin
        if spent + estimate_s > budget_s:
            skipped[name] = "%.0f s spent + ~%.0f s estimated > budget %.0f s (MGC_BENCH_BUDGET_S)" % (spent, estimate_s, budget_s)
            return False
        return True

    # Libraries (RCCL prints a version banner) may write to stdout; the contract is ONE JSON line
    # there, so everything before the final print goes to stderr.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    from meryl_amd import build, capi, count

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        print("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
              % (args.gpus, args.gpus), file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (no CPU fallback exists for the count path)", file=sys.stderr)
        sys.exit(2)
    one_device = os.environ.get("MGC_BENCH_ONE_DEVICE", "0") == "1"      # every rank on device 0 (a one-GPU box)
    dev_index = 0 if one_device else local_rank
    torch.cuda.set_device(dev_index)

    # MGC_BENCH_FORCE_SHARDED=1 runs the multi-GPU code path (partition -> exchange waves -> owner-side count) even
    # with a single rank, so that it can be exercised on a 1-GPU box
    force_sharded = os.environ.get("MGC_BENCH_FORCE_SHARDED", "0") == "1"
    dist = None
    transport = "none (one GPU)"
    if world > 1 or force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        transport = "RCCL point-to-point waves, one process per GPU"
        try:                                                     # two ranks on one device: RCCL says "Duplicate GPU detected"
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
            if world > 1:
                probe = torch.ones(1, device="cuda")
                dist.all_reduce(probe)
                torch.cuda.synchronize()
        except Exception as e:                                   # noqa: BLE001 -- whatever RCCL raises, the fallback is the same
            if world == 1:
                raise
            why = str(e).strip().splitlines()[-1][:160] if str(e).strip() else type(e).__name__
            transport = "peer copies between %d virtual ranks of ONE process (mgc_count_node): RCCL refused the %d ranks (%s)" % (world, world, why)
            try:
                dist.destroy_process_group()
            except Exception:                                    # noqa: BLE001
                pass
            dist = None
            if rank != 0:
                os._exit(0)                                      # rank 0 carries the whole job in the fallback
    node_fallback = world > 1 and dist is None
    # the in-tree library is normally up to date (it travels with the snapshot); if it has to be rebuilt, one rank does it
    if local_rank == 0 or node_fallback:
        build.build()
    if dist is not None:
        dist.barrier()
    capi.lib()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the per-rank memory model, checked before anything is allocated (N > 1 and the forced-sharded form) ----
    reads = args.reads
    mem_model = None
    if world > 1 or force_sharded:
        mem_model = hbm_model(reads, world if not node_fallback else 1)
        if node_fallback:                                    # N virtual ranks share ONE device
            mem_model = {kk: v * world for kk, v in hbm_model(reads, world).items()}
        free_b, total_b = torch.cuda.mem_get_info()
        if mem_model["total"] > 0.94 * total_b:
            print("bench.py: %d reads per GPU on %d rank(s) need ~%.0f GB of HBM per device by the model of DESIGN.md 4 (%s), the device has %.0f GB: "
                  "refusing before anything is allocated (use --reads)" % (reads, world, mem_model["total"] / 1e9,
                  ", ".join("%s %.1f" % (kk, v / 1e9) for kk, v in mem_model.items() if kk != "total"), total_b / 1e9), file=sys.stderr)
            if dist is not None:
                dist.destroy_process_group()
            sys.exit(2)

    # ---- synthetic input, generated in HBM ----
    genome_len = GENOME_LEN * world              # weak scaling: every GPU brings its own 30x share of a genome that grows with N
    ranks_here = list(range(world)) if node_fallback else [rank]
    all_bases = [count.dev_synth_reads(SEED, genome_len, r * reads, reads, READ_LEN, 5000, 100) for r in ranks_here]
    bases = all_bases[0]
    torch.cuda.synchronize()
    n_bases = bases.numel()

    prof_acc = {"pass_ms": 0.0, "pass_launches": 0, "pass_keys": 0, "stage_ms": [0.0] * capi.NUM_STAGES,
                "by_pass": [{"ms": 0.0, "launches": 0, "keys": 0, "bytes": 0} for _ in range(2)],
                "finish": {"ms": 0.0, "launches": 0, "keys": 0, "bytes": 0}}
    result = {}
    single = world == 1 and not force_sharded
    cfg = capi.configure(K, 10_000_000_000 if reads == DEFAULT_READS else n_bases * world, 64 << 30)
    sess = None
    node_dir = None

    if single:
        sess = count.Session(cfg, dev_index)
        sess.push_bases_device(bases)
        sess.set_profiling(True)

        def step(timed):
            sess.count()
            if timed:
                p = sess.profile()
                prof_acc["pass_ms"] += p.sort_pass_ms_total
                prof_acc["pass_launches"] += p.sort_pass_launches
                prof_acc["pass_keys"] += p.sort_pass_keys
                for i in range(2):
                    bp = prof_acc["by_pass"][i]
                    bp["ms"] += p.pass_ms[i]; bp["launches"] += p.pass_launches[i]
                    bp["keys"] += p.pass_keys[i]; bp["bytes"] += p.pass_bytes[i]
                fin = prof_acc["finish"]
                fin["ms"] += p.finish_ms; fin["launches"] += p.finish_launches; fin["keys"] += p.finish_keys; fin["bytes"] += p.finish_bytes
                for i in range(capi.NUM_STAGES):
                    prof_acc["stage_ms"][i] += p.stage_ms[i]
                prof_acc["pack_ms"] = prof_acc.get("pack_ms", 0.0) + p.pack_ms
                prof_acc["partition_bytes"] = prof_acc.get("partition_bytes", 0) + p.partition_bytes
                prof_acc["hist_bytes"] = prof_acc.get("hist_bytes", 0) + p.hist_bytes
                prof_acc["plan"] = {"stream_files": p.stream_files, "stream_retries": p.stream_retries, "probe_ratio": p.probe_ratio}
            info = sess.info()
            result["info"] = info
            result["n_distinct"] = info.n_distinct
            result["n_instances"] = info.n_instances
            result["w_prefix"] = info.w_prefix
    elif node_fallback:
        node_dir, _ = shm_dir("mgc_node_")

        def step(timed):
            import shutil
            out = os.path.join(node_dir, "db.meryl")
            shutil.rmtree(out, ignore_errors=True)
            result["node_profile"] = count.count_node(cfg, all_bases, out, devices=[0] * world, host_threads=min(32, os.cpu_count() or 8))
            result["n_distinct"] = result["node_profile"]["n_distinct"]
            result["n_instances"] = result["node_profile"]["n_instances"]
    else:
        def step(timed, profile=False):
            # per-file profiling synchronises after every owned file; it runs in ONE extra untimed step (below)
            count.SHARD_PROFILE = prof_acc if profile else None
            uniq, cnts, rng = count.count_sharded(bases, K)
            result["uniq"], result["cnts"], result["range"] = uniq, cnts, rng
            result["n_distinct_local"] = uniq.shape[0]

    for _ in range(args.warmup):
        step(False)
    # The sharded forms hand their results back as torch tensors while the library holds its own arena: torch's caching
    # allocator keeps growing for a step or two (a step that makes it hipMalloc another 70 GB lasts 1.3 s instead of
    # 0.19 -- scripts/fsh_trace.py).  That is the harness's memory settling, not the path: up to three more untimed steps
    # until the pool stops growing (every rank runs the same number: the decision is reduced over the ranks).
    if not single:
        for _ in range(3):
            before = torch.cuda.memory_reserved()
            step(False)
            grew = torch.tensor([1 if torch.cuda.memory_reserved() > before else 0], dtype=torch.int32, device="cuda")
            if dist is not None:
                dist.all_reduce(grew, op=dist.ReduceOp.MAX)
            if not int(grew.item()):
                break
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    dt = time.perf_counter() - t0
    stage_s = None
    if dist is not None:
        os.environ["MGC_SHARD_PROFILE"] = "1"
        stage_db = {}
        step(False, profile=True)        # collective: every rank runs it; rank 0's pass timings feed the roofline object
        os.environ.pop("MGC_SHARD_PROFILE", None)
        barrier()

    # max over ranks, totals over ranks
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        d = torch.tensor([result["n_distinct_local"]], dtype=torch.int64, device="cuda")
        dist.all_reduce(d)
        result["n_distinct"] = int(d.item())

    ms_per_step = dt / args.steps * 1e3
    n_distinct = result["n_distinct"]
    threads = min(32, os.cpu_count() or 8)
    line = {
        "metric": "distinct k-mers counted/sec (whole node)",
        "value": n_distinct / (dt / args.steps),
        "unit": "distinct k-mers/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {
            "workload": "meryl count k=21 on synthetic short reads: %d x %d bp reads per GPU (%.2f Gbp per GPU, "
                        "30x of a %d bp genome, 0.5%% substitutions, 0.01%% N), inputs resident in HBM"
                        % (reads, READ_LEN, reads * READ_LEN / 1e9, genome_len),
            "k": K, "reads_per_gpu": reads, "bases_per_gpu": n_bases,
            "n_distinct": n_distinct,
            "parallelism": "1 GPU" if world == 1 else "%d GPUs: the top-bit buckets (64 files x ranges) in contiguous per-rank ranges, bucket-major point-to-point waves overlapped with the owner-side count" % world,
            "transport": transport,
        },
    }
    if mem_model:
        line["config"]["hbm_model_gb_per_rank"] = {kk: round(v / 1e9, 2) for kk, v in mem_model.items()}
        line["config"]["hbm_peak_allocated_gb"] = round(torch.cuda.max_memory_allocated() / 1e9, 2)       # torch's own (the library's arena is not in it)
    if node_fallback:
        line["config"]["note"] = ("one-GPU box: the step is mgc_count_node with %d virtual ranks sharing the device and INCLUDES writing the "
                                  "database; not a scaling number" % world)

    legs_s["setup+input+warmup+timed_steps"] = time.perf_counter() - t_begin
    # (collective legs: every rank decides alike -- rank 0's clock, broadcast)
    def leg_ok_all(name, estimate_s):
        ok = leg_ok(name, estimate_s) if rank == 0 else True
        if dist is not None:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
            dist.broadcast(flag, src=0)
            ok = bool(flag.item())
        return ok

    # ---- check (collective at N > 1) ----
    if not args.no_check and leg_ok_all("check", 10 if world == 1 else 30):
        t_leg = time.perf_counter()
        try:
            if single:
                keys, cnts = sess.result_device()
                line["check"] = result_check(keys, cnts, bases, K, result["info"].n_instances, result["info"].file_instances)
                del keys, cnts
            elif dist is not None:
                fh = count.HipOps.histogram(bases, K, capi.MODE_CANONICAL, 6)       # this rank's k-mers per file, from the base stream
                chk = result_check(result["uniq"], result["cnts"], bases, K, None, fh, dist)
                if rank == 0:
                    line["check"] = chk
            else:                                                                    # node fallback: the database itself is checked below
                want = sum(valid_windows(b, K) for b in all_bases)
                line["check"] = {"valid_windows": want, "instances_equal_valid_windows": bool(want == result["n_instances"]),
                                 "ok": bool(want == result["n_instances"])}
        except Exception as e:                                                       # noqa: BLE001 -- reported, never fatal for the metric line
            if rank == 0:
                line["check"] = {"error": str(e)[:300], "ok": False}
        legs_s["check"] = time.perf_counter() - t_leg
    result.pop("uniq", None); result.pop("cnts", None)

    # ---- the database (untimed) ----
    if not args.no_db and leg_ok_all("db_write", 10 if world == 1 else 60):
        t_leg = time.perf_counter()
        import shutil
        try:
            if single:
                dbdir, _ = shm_dir("mgc_db_")
                try:
                    t0 = time.perf_counter()
                    wp = sess.write_database(os.path.join(dbdir, "db.meryl"), threads)
                    wp["wall_s"] = time.perf_counter() - t0
                    wp["vs_count_step"] = wp["wall_s"] / (ms_per_step / 1e3)
                    line["db_write"] = wp
                finally:
                    shutil.rmtree(dbdir, ignore_errors=True)
            elif dist is not None:
                # every rank writes its part of ONE directory (rank 0 names it), rank 0 stitches; then the same reads through the
                # in-process peer-copy form (rank 0 drives all devices) and the two directories are compared
                name = [None]
                if rank == 0:
                    name[0], _ = shm_dir("mgc_sdb_")
                dist.broadcast_object_list(name, src=0)
                os.environ["MGC_SHARD_PROFILE"] = "1"
                info = dict(path=os.path.join(name[0], "db.meryl"), w_prefix=cfg.w_prefix, host_threads=max(4, threads // world))
                barrier()
                t0 = time.perf_counter()
                count.count_sharded(bases, K, db=info, keep_result=False)
                barrier()
                wall = time.perf_counter() - t0
                os.environ.pop("MGC_SHARD_PROFILE", None)
                stage_s = info.get("stage_s")
                count.release_cached_sessions()
                torch.cuda.empty_cache()
                if rank == 0:
                    dg, nfiles, nbytes = dir_digest(info["path"])
                    line["db_write"] = {"wall_s": wall, "what": "count_sharded(db=...): count + device-encoded parts + stitch, one collective call",
                                        "files": nfiles, "database_bytes": nbytes, "sha256": dg,
                                        "rank0_stream": info.get("profile"), "vs_count_step": wall / (ms_per_step / 1e3)}
                # the peer-copy form over the same devices needs their memory: every rank lets go of its reads first
                keep_first = bases[:min(args.cpu_sample_reads, reads) * (READ_LEN + 1)].clone() if rank == 0 else None
                del bases, all_bases
                torch.cuda.empty_cache()
                barrier()
                if rank == 0 and not leg_ok("db_write.node_count", 90 + 15 * world):
                    line["db_write"]["node_count"] = {"skipped": skipped["db_write.node_count"]}
                    shutil.rmtree(name[0], ignore_errors=True)
                elif rank == 0:
                    try:
                        nb = []
                        for r in range(world):
                            with torch.cuda.device(r if not one_device else 0):
                                nb.append(count.dev_synth_reads(SEED, genome_len, r * reads, reads, READ_LEN, 5000, 100))
                        torch.cuda.synchronize()
                        out = os.path.join(name[0], "node.meryl")
                        np_ = count.count_node(cfg, nb, out, devices=[(r if not one_device else 0) for r in range(world)], host_threads=threads)
                        dg2, _, _ = dir_digest(out)
                        line["db_write"]["node_count"] = {k: np_[k] for k in ("total_s", "partition_s", "exchange_count_s", "close_s", "merge_parts_s",
                                                                             "n_batches", "n_distinct", "bucket_bits")}
                        line["db_write"]["node_count"]["sha256"] = dg2
                        line["db_write"]["identical_to_node_count"] = bool(dg == dg2)
                        del nb
                    except Exception as e:                                           # noqa: BLE001
                        line["db_write"]["node_count"] = {"error": str(e)[:300]}
                    shutil.rmtree(name[0], ignore_errors=True)
                    torch.cuda.empty_cache()
                barrier()
                bases = keep_first
            else:
                out = os.path.join(node_dir, "db.meryl")
                dg, nfiles, nbytes = dir_digest(out)
                # the same reads through ONE session
                one = os.path.join(node_dir, "one.meryl")
                cat = torch.cat(all_bases)
                with count.Session(cfg, 0) as s1:
                    s1.push_bases_device(cat)
                    s1.count()
                    s1.write_database(one, threads)
                dg1, _, _ = dir_digest(one)
                del cat
                line["db_write"] = {"what": "mgc_count_node with %d virtual ranks on one device (part of every timed step)" % world, "files": nfiles,
                                    "database_bytes": nbytes, "sha256": dg, "identical_to_single_session": bool(dg == dg1),
                                    "node_profile": result.get("node_profile")}
        except Exception as e:                                                       # noqa: BLE001
            if rank == 0:
                line.setdefault("db_write", {})["error"] = str(e)[:300]
        legs_s["db_write"] = time.perf_counter() - t_leg
    if node_dir:
        import shutil
        shutil.rmtree(node_dir, ignore_errors=True)

    if rank == 0:
        if single:
            n_inst = result["n_instances"]
            line["config"]["n_instances"] = n_inst
            line["config"]["w_prefix"] = result["w_prefix"]
            line["instances_per_s"] = n_inst / (dt / args.steps)
            line["stage_ms_per_step"] = {capi.STAGE_NAMES[i]: prof_acc["stage_ms"][i] / args.steps for i in range(capi.NUM_STAGES)}
            if prof_acc.get("plan"):
                line["config"]["file_plan"] = prof_acc["plan"]      # files on the distinct-sized count, its retries, the probe file's D / N
        elif stage_s:
            line["stage_ms_one_profiled_step"] = {k: v * 1e3 for k, v in stage_s.items()}
            line["stage_note"] = ("rank 0, the database-writing call: partition / plan / exchange+count (first_wave_exposed = the part of the "
                                  "exchange nothing overlaps) / database (waiting for this rank's files) / stitch")
        elif node_fallback and result.get("node_profile"):
            npf = result["node_profile"]
            line["stage_ms_per_step"] = {"partition": npf["partition_s"] * 1e3, "exchange+owner count": npf["exchange_count_s"] * 1e3,
                                         "files": npf["close_s"] * 1e3, "stitch": npf["merge_parts_s"] * 1e3}
        rf = roofline_object(prof_acc, ms_per_step, args.steps, reads, single,
                             "over the timed steps" if single else "rank 0's owner-side count in one extra untimed step")
        if rf:
            line["roofline"] = rf
        elif node_fallback:
            # mgc_count_node owns its sessions: the kernels' launch durations are measured in ONE session counting rank 0's
            # reads on the same device (the same kernels on the same per-GPU workload)
            try:
                acc1 = {"pass_ms": 0.0, "pass_launches": 0, "pass_keys": 0, "stage_ms": [0.0] * capi.NUM_STAGES,
                        "by_pass": [{"ms": 0.0, "launches": 0, "keys": 0, "bytes": 0} for _ in range(2)],
                        "finish": {"ms": 0.0, "launches": 0, "keys": 0, "bytes": 0}}
                with count.Session(cfg, dev_index) as s1:
                    s1.push_bases_device(all_bases[0])
                    s1.set_profiling(True)
                    s1.count()
                    t1 = time.perf_counter()
                    for _ in range(2):
                        s1.count()
                        p1 = s1.profile()
                        acc1["pass_ms"] += p1.sort_pass_ms_total; acc1["pass_launches"] += p1.sort_pass_launches; acc1["pass_keys"] += p1.sort_pass_keys
                        for i in range(2):
                            bp = acc1["by_pass"][i]
                            bp["ms"] += p1.pass_ms[i]; bp["launches"] += p1.pass_launches[i]; bp["keys"] += p1.pass_keys[i]; bp["bytes"] += p1.pass_bytes[i]
                        f1 = acc1["finish"]
                        f1["ms"] += p1.finish_ms; f1["launches"] += p1.finish_launches; f1["keys"] += p1.finish_keys; f1["bytes"] += p1.finish_bytes
                        for i in range(capi.NUM_STAGES):
                            acc1["stage_ms"][i] += p1.stage_ms[i]
                        acc1["pack_ms"] = acc1.get("pack_ms", 0.0) + p1.pack_ms
                        acc1["partition_bytes"] = acc1.get("partition_bytes", 0) + p1.partition_bytes
                        acc1["hist_bytes"] = acc1.get("hist_bytes", 0) + p1.hist_bytes
                    torch.cuda.synchronize()
                    ms1 = (time.perf_counter() - t1) / 2 * 1e3
                rf = roofline_object(acc1, ms1, 2, reads, True, "in one single-session count of rank 0's reads on the same device "
                                     "(2 steps; the one-process fallback's mgc_count_node owns its sessions)")
                if rf:
                    rf["single_session_ms_per_step"] = ms1
                    line["roofline"] = rf
            except Exception as e:                                                   # noqa: BLE001
                line["roofline"] = {"error": str(e)[:300], "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": None, "traffic": None}

    # ---- file -> database through the CLI, and the CPU leg (rank 0; the other ranks have let go of their memory) ----
    if sess is not None:
        sess.close()
    count.release_cached_sessions()
    torch.cuda.empty_cache()
    if rank == 0:
        if not args.no_e2e and bases is not None and leg_ok("e2e", 45 if world == 1 else 60 + 20 * world):
            t_leg = time.perf_counter()
            try:
                line["e2e"] = e2e_run(bases, min(reads, bases.numel() // (READ_LEN + 1)), threads, gpus=1 if one_device else world)
                nd = line["e2e"].get("n_distinct")
                if nd and line["e2e"].get("wall_s"):
                    line["value_e2e"] = nd / line["e2e"]["wall_s"]
                    line["value_e2e_note"] = "distinct k-mers / wall clock of `meryl count` file -> database (process start to exit), %d reads" % line["e2e"]["reads"]
            except Exception as e:                                                   # noqa: BLE001
                line["e2e"] = {"error": str(e)[:300]}
            legs_s["e2e"] = time.perf_counter() - t_leg
        # ... and compressed inputs (one gzip stream; BGZF): a tenth of the reads -- the single stream inflates at a few hundred MB/s
        if not args.no_e2e and bases is not None and world == 1 and leg_ok("e2e_compressed", 60):
            t_leg = time.perf_counter()
            try:
                zreads = max(1000, min(reads, bases.numel() // (READ_LEN + 1)) // 10)
                line["e2e_compressed"] = e2e_compressed_run(bases, zreads, threads)
            except Exception as e:                                                   # noqa: BLE001
                line["e2e_compressed"] = {"error": str(e)[:300]}
            legs_s["e2e_compressed"] = time.perf_counter() - t_leg
        # the CPU leg: rank 0 at N = 1 only (the harness's rule); the whole workload at the best thread count when the budget
        # allows its ~150-170 s, else the bounded sample
        if not args.no_cpu_baseline and bases is not None and world == 1:
            sample = min(args.cpu_sample_reads, reads, bases.numel() // (READ_LEN + 1))
            whole = (not args.cpu_sample) and leg_ok("cpu_baseline.whole_workload", 175.0 * reads / DEFAULT_READS)
            if whole or leg_ok("cpu_baseline.sample", 100.0 * sample / 14_000_000):
                t_leg = time.perf_counter()
                line["cpu_baseline"] = cpu_baseline(sample, os.cpu_count() or 1, bases, whole=whole)
                legs_s["cpu_baseline"] = time.perf_counter() - t_leg
        elif not args.no_cpu_baseline and world > 1:
            skipped["cpu_baseline"] = "reported at N = 1 only"
        legs_s["total"] = time.perf_counter() - t_begin
        line["legs_s"] = {k: round(v, 3) for k, v in legs_s.items()}
        line["legs_skipped"] = skipped
        line["legs_budget_s"] = budget_s
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
