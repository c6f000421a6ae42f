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

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, the
radix grouping pass (algorithmic 8 B read + 8 B write per key), timed with HIP
events on the library's own stream during the timed steps; `traffic` is the PMC
measurement committed under profiles/ for this same workload.  `cpu_baseline` is
the CPU restatement of the reference algorithm (oracle/, kind "port") timed on
this box's host cores over a bounded sample of the same workload shape.
"""
import argparse
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


def cpu_baseline(sample_reads, threads, dev_bases=None):
    """Reference-algorithm port (oracle/oracle_port.cpp: 2 MiB chunks, spin-locked bit-packed prefix buckets with the
    reference's one/two/three-word add, std::sort, run-length count, 64-file dump) on the host cores, over a bounded
    sample of the SAME workload: the first `sample_reads` reads of the same 333 Mbp genome with the workload's
    geometry (wPrefix 18) -- coverage per k-mer is lower than in the whole run, which favours the CPU's distinct/s.
    Every thread count tried is reported; `value` is the best one."""
    import oracle
    oracle.build()
    if dev_bases is not None:                                         # the very bytes the GPU counted (first reads of rank 0)
        bases = dev_bases[:sample_reads * (READ_LEN + 1)].cpu().numpy()
    else:
        bases = oracle.synth_reads(SEED, GENOME_LEN, 0, sample_reads, READ_LEN, 5000, 100)
    cfg = oracle.configure_counting(K, 10_000_000_000, 64 << 30)      # the workload's geometry (wPrefix 18)
    # the reference's spin-locked buckets do not scale to every core count: time a few thread counts (all <= the
    # box's cores) and report every one
    tried = []
    best = None
    for th in sorted({max(1, min(threads, t)) for t in (16, 64, threads)}):
        t0 = time.perf_counter()
        _, nd, ni = oracle.digest_threaded(bases, K, cfg["w_prefix"], oracle.CANONICAL, th)
        dt = time.perf_counter() - t0
        tried.append({"threads": th, "seconds": dt, "distinct_per_s": nd / dt, "instances_per_s": ni / dt})
        if best is None or dt < best[0]:
            best = (dt, th, nd, ni)
    dt, th, nd, ni = best
    return {
        "value": nd / dt, "unit": "distinct k-mers/s", "cores": th, "kind": "port",
        "sample": "the first %d x %d bp reads of the workload (%.2f Gbp of the %d bp genome's reads), k=%d, wPrefix=%d; "
                  "%d instances, %d distinct in %.2f s (%.3g instances/s) on %d threads"
                  % (sample_reads, READ_LEN, bases.size / 1e9, GENOME_LEN, K, cfg["w_prefix"], ni, nd, dt, ni / dt, th),
        "instances_per_s": ni / dt, "seconds": dt, "threads_tried": tried, "host_cores": os.cpu_count(),
    }


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


def result_check(sess, bases, k, info):
    """Untimed sanity check of the counted result at the judged size (the parity tests proper are tests/test_gpu_parity.py):
    instances == valid windows of the input, distinct k-mers strictly ascending, counts sum to the instances per file."""
    import torch
    keys, cnts = sess.result_device()
    out = {}
    want = valid_windows(bases, k)
    out["valid_windows"] = want
    out["instances_equal_valid_windows"] = bool(info.n_instances == want)
    out["keys_strictly_ascending"] = bool((keys[1:] > keys[:-1]).all().item()) if keys.numel() > 1 else True   # k <= 31: int64 order == uint64 order
    c64 = cnts.to(torch.int64) & 0xFFFFFFFF
    out["sum_counts_equals_instances"] = bool(int(c64.sum().item()) == info.n_instances)
    bounds = torch.arange(0, 65, device=keys.device, dtype=torch.int64) << (2 * k - 6)
    cut = torch.searchsorted(keys, bounds)
    csum = torch.cat([torch.zeros(1, dtype=torch.int64, device=keys.device), torch.cumsum(c64, 0)])
    per_file = (csum[cut[1:]] - csum[cut[:-1]]).cpu().tolist()
    out["per_file_totals_match"] = bool(per_file == [int(x) for x in info.file_instances])
    out["distinct_ge_1"] = bool(int(c64.min().item()) >= 1) if keys.numel() else True
    out["ok"] = all(v for kk, v in out.items() if isinstance(v, bool))
    return out


def e2e_run(bases, reads, threads):
    """File -> database wall clock of the stand-alone CLI: the synthetic reads are written as a FASTQ
    file on tmpfs, `meryl count` reads it, parses it on the device, counts, encodes the blocks on the device and writes
    the 64-file database back to tmpfs.  Never part of `value`."""
    import re
    import shutil
    import subprocess
    import tempfile
    import torch
    from meryl_amd import build
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    st = os.statvfs(shm)
    free = st.f_bavail * st.f_frsize
    rec = READ_LEN * 2 + 7
    use = reads
    need = lambda r: r * rec * 1.45 + (1 << 30)            # FASTQ + database
    while use > 1000 and need(use) > free * 0.8:
        use //= 2
    d = tempfile.mkdtemp(prefix="mgc_e2e_", dir=shm)
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
        dbp = os.path.join(d, "out.meryl")
        cli = build.build_cli()
        cmd = [cli, "-V", "k=%d" % K, "memory=64", "threads=%d" % threads, "n=10000000000", "count", fq, "output", dbp]
        t0 = time.perf_counter()
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        wall = time.perf_counter() - t0
        if p.returncode != 0:
            return {"error": "meryl CLI rc=%d: %s" % (p.returncode, p.stderr[-400:])}
        out = {"reads": use, "bases": use * READ_LEN, "fastq_bytes": os.path.getsize(fq), "wall_s": wall,
               "threads": threads, "where": shm, "fastq_generation_s": t_gen,
               "database_bytes": sum(os.path.getsize(os.path.join(dbp, n)) for n in os.listdir(dbp)),
               "command": "meryl -V k=%d memory=64 threads=%d n=10000000000 count reads.fq output out.meryl" % (K, threads)}
        if os.environ.get("MGC_IO_TRACE"):
            out["io_trace"] = [l for l in p.stderr.splitlines() if l.startswith("[io]")]
        m = re.search(r"TIMING(.*)", p.stderr)
        if m:
            for name, val in re.findall(r"([a-z+_]+)=([0-9.]+)", m.group(1)):
                out[name + ("" if name.endswith("bytes") else "_s")] = float(val)
        if "count" in out and out.get("count_s"):
            pass
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def pmc_traffic(reads):
    """HBM bytes per launch of the dominant kernel from the committed PMC run of this same workload
    (profiles/*_pmc_traffic.json, made by scripts/gpu_pmc_bench.sh: FETCH_SIZE and WRITE_SIZE in separate
    rocprofv3 passes, calibrated on known-byte kernels of the same access width); None for any other workload --
    counters cannot be collected from inside this process."""
    import glob
    for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("reads_per_gpu") == reads and str(d.get("kernel", "")).startswith("radix_group_kernel"):
            return d["fetch_bytes_per_launch"] + d["write_bytes_per_launch"], "profiles/" + os.path.basename(f)
    return None, None


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
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

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
    torch.cuda.set_device(local_rank)

    # MGC_BENCH_FORCE_SHARDED=1 runs the multi-GPU code path (partition -> exchange waves -> owner-side count) even
    # with a single rank, so that it can be exercised on a 1-GPU box
    force_sharded = os.environ.get("MGC_BENCH_FORCE_SHARDED", "0") == "1"
    dist = None
    if world > 1 or force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    # the in-tree library is normally up to date (it travels with the snapshot); if it has to be rebuilt, one rank does it
    if local_rank == 0:
        build.build()
    if dist is not None:
        dist.barrier()
    capi.lib()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- synthetic input, generated in HBM ----
    reads = args.reads
    genome_len = GENOME_LEN * world              # weak scaling: every GPU brings its own 30x share of a genome that grows with N
    bases = count.dev_synth_reads(SEED, genome_len, rank * reads, reads, READ_LEN, 5000, 100)
    torch.cuda.synchronize()
    n_bases = bases.numel()

    prof_acc = {"pass_ms": 0.0, "pass_launches": 0, "pass_keys": 0, "stage_ms": [0.0] * capi.NUM_STAGES,
                "by_pass": [{"ms": 0.0, "launches": 0, "keys": 0, "bytes": 0} for _ in range(2)]}
    result = {}

    if world == 1 and not force_sharded:
        cfg = capi.configure(K, 10_000_000_000 if reads == DEFAULT_READS else n_bases, 64 << 30)
        sess = count.Session(cfg, local_rank)
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
                for i in range(capi.NUM_STAGES):
                    prof_acc["stage_ms"][i] += p.stage_ms[i]
            info = sess.info()
            result["info"] = info
            result["n_distinct"] = info.n_distinct
            result["n_instances"] = info.n_instances
            result["w_prefix"] = info.w_prefix
    else:
        def step(timed, profile=False):
            # per-file profiling synchronises after every owned file; it runs in ONE extra untimed step (below)
            count.SHARD_PROFILE = prof_acc if profile else None
            uniq, cnts, _ = count.count_sharded(bases, K)
            result["n_distinct_local"] = uniq.numel()
            result["n_instances_local"] = int(cnts.to(torch.int64).sum().item()) if not timed else 0

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        step(False, profile=True)        # collective: every rank runs it; rank 0's pass timings feed the roofline object
        barrier()

    # max over ranks, totals over ranks
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        d = torch.tensor([result["n_distinct_local"]], dtype=torch.int64, device="cuda")
        dist.all_reduce(d)
        result["n_distinct"] = int(d.item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        n_distinct = result["n_distinct"]
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
            },
        }
        single = world == 1 and not force_sharded
        if single:
            n_inst = result["n_instances"]
            line["config"]["n_instances"] = n_inst
            line["config"]["w_prefix"] = result["w_prefix"]
            line["instances_per_s"] = n_inst / (dt / args.steps)
        if prof_acc["pass_launches"]:                                    # N > 1: rank 0's owner-side passes
            # The dominant kernel is a file's FIRST grouping pass.  Algorithmic bytes = key bytes it must read and write:
            # 8 + 8 per k-mer for the wide pass, 8 + 4 when the pass narrows its output to 32-bit words (k <= ~25: the
            # digit a key was grouped by is dropped, the second pass then moves 4 + 4) -- the library reports them per launch.
            bp = prof_acc["by_pass"]
            if not bp[0]["launches"]:                                    # sharded owner side: only the totals are collected
                bp = [{"ms": prof_acc["pass_ms"], "launches": prof_acc["pass_launches"], "keys": prof_acc["pass_keys"],
                       "bytes": prof_acc.get("pass_bytes", 16 * prof_acc["pass_keys"])}, {"ms": 0.0, "launches": 0, "keys": 0, "bytes": 0}]
            achieved = bp[0]["bytes"] / (bp[0]["ms"] / 1e3) / 1e9
            narrowed = bp[0]["bytes"] < 16 * bp[0]["keys"]
            line["roofline"] = {
                "kernel": "radix_group_kernel, first pass of a file (9-bit digit; %s)" %
                          ("8 B k-mers in, 4 B narrowed words out" if narrowed else "8 B k-mers in and out"),
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(reads)[0] if single else None,
                "traffic_source": (pmc_traffic(reads)[1] or "none: no committed PMC run of this workload") if single else None,
                "measured": "HIP events around every pass launch of the timed steps" if single else
                            "HIP events around every pass launch of rank 0's owner-side count in one extra untimed step",
                "launches": bp[0]["launches"],
                "avg_launch_ms": bp[0]["ms"] / bp[0]["launches"],
                "algorithmic_bytes_per_launch": bp[0]["bytes"] / bp[0]["launches"],
                "keys_per_launch": bp[0]["keys"] / bp[0]["launches"],
            }
            if bp[1]["launches"]:
                a1 = bp[1]["bytes"] / (bp[1]["ms"] / 1e3) / 1e9
                line["roofline"]["second_pass"] = {"achieved": a1, "frac": a1 / HBM_PEAK_GBS, "launches": bp[1]["launches"],
                                                   "avg_launch_ms": bp[1]["ms"] / bp[1]["launches"],
                                                   "algorithmic_bytes_per_launch": bp[1]["bytes"] / bp[1]["launches"]}
            # the same passes priced the way SURVEY 8(d) prices a radix pass (8 + 8 B per k-mer whatever is really moved):
            # comparable with earlier rounds' 0.47
            eq = 16.0 * prof_acc["pass_keys"] / (prof_acc["pass_ms"] / 1e3) / 1e9
            line["roofline"]["survey_accounting"] = {"bytes_per_key_per_pass": 16, "achieved": eq, "frac": eq / HBM_PEAK_GBS}
        if single:
            line["stage_ms_per_step"] = {capi.STAGE_NAMES[i]: prof_acc["stage_ms"][i] / args.steps
                                         for i in range(capi.NUM_STAGES)}
            if not args.no_check:
                line["check"] = result_check(sess, bases, K, result["info"])
            if not args.no_e2e:
                # device encode + write of the timed result (no file input), then the whole CLI file -> database
                import shutil
                import tempfile
                shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
                dbdir = tempfile.mkdtemp(prefix="mgc_db_", dir=shm)
                try:
                    t0 = time.perf_counter()
                    wp = sess.write_database(os.path.join(dbdir, "db.meryl"), min(32, os.cpu_count() or 8))
                    wp["wall_s"] = time.perf_counter() - t0
                    wp["vs_count_step"] = wp["wall_s"] / (ms_per_step / 1e3)
                    line["db_write"] = wp
                except Exception as e:                                  # reported, never fatal for the metric line
                    line["db_write"] = {"error": str(e)[:300]}
                finally:
                    shutil.rmtree(dbdir, ignore_errors=True)
                sess.close()
                count.release_cached_sessions()
                torch.cuda.empty_cache()
                try:
                    line["e2e"] = e2e_run(bases, reads, min(32, os.cpu_count() or 8))
                except Exception as e:
                    line["e2e"] = {"error": str(e)[:300]}
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(min(args.cpu_sample_reads, reads), os.cpu_count() or 1, bases)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
