#!/usr/bin/env python3
"""BASELINE config 5's mechanics at a size one pass cannot hold: k=51 with a constant 8-bit label over >= 20 Gbp of 150 bp reads
(16-byte keys: ~275 GB of k-mers, beyond HBM), pushed FROM THE HOST in 64 MiB pieces (pinned double-buffered uploads) and
counted in batches by the worker thread while the next batch uploads.  Every batch result is parked as a sorted run -- in HBM
within --budget bytes, in pinned host DRAM beyond it (writeBatch's spill, merylOp-countThreads.C:323-379) -- and the runs are
merged ONCE, chunk by chunk, straight into the database stream (merylBlockWriter::finish(), :461-464).  Reports the wall
clocks, where the runs went, the merge and the peak device memory.  Checked by size-independent properties: the database's
total equals the valid k-mer windows of the input counted independently (torch ops), the per-file totals match, every file's
k-mers ascend (a sample of files is read back).
usage: python scripts/ooc_bench.py [--reads N] [--batch BASES] [--budget BYTES] [--k K] [--out JSON]"""
import argparse
import ctypes
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from meryl_amd import capi, count, db  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=134_000_000)            # 20.1 Gbp
ap.add_argument("--batch", type=int, default=2_500_000_000)
ap.add_argument("--budget", type=int, default=20 << 30, help="bytes of runs that may stay in HBM")
ap.add_argument("--k", type=int, default=51)
ap.add_argument("--out", default=None)
args = ap.parse_args()
k, reads = args.k, args.reads
rl = 150
genome = reads * rl // 50                                              # 50x

# ---- the input: generated on the device slice by slice, kept in HOST memory; its valid windows counted on the way ----
t0 = time.perf_counter()
host = np.empty(reads * (rl + 1), dtype=np.uint8)
windows = 0
step = 20_000_000
for a in range(0, reads, step):
    n = min(step, reads - a)
    d = count.dev_synth_reads(5, genome, a, n, rl, 5000, 100)
    windows += bench.valid_windows(d, k)
    host[a * (rl + 1):(a + n) * (rl + 1)] = d.cpu().numpy()
    del d
torch.cuda.empty_cache()
t_gen = time.perf_counter() - t0

cfg = capi.configure(k, host.size, 64 << 30, label_size=8, label=0x5A)
shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
outdir = tempfile.mkdtemp(prefix="mgc_ooc_", dir=shm)
res = {}
try:
    with count.Session(cfg, 0) as s:
        s.set_batch_bases(args.batch)
        s.set_result_budget(args.budget)
        L = capi.lib()
        ptr = host.ctypes.data
        piece = 64 << 20
        t0 = time.perf_counter()
        for a in range(0, host.size, piece):
            n = min(piece, host.size - a)
            capi.check(L.mgc_push_bases(s._h, ctypes.cast(ptr + a, ctypes.c_char_p), n, 0), "mgc_push_bases", s._h)
        t_push = time.perf_counter() - t0
        s.count()
        t_count = time.perf_counter() - t0
        ooc = s.out_of_core()
        info = s.info()
        p = s.profile()
        dbp = os.path.join(outdir, "db.meryl")
        t1 = time.perf_counter()
        wp = s.write_database(dbp, min(32, os.cpu_count() or 8))
        t_write = time.perf_counter() - t1
        n_distinct = s.info().n_distinct
        rp = s.runs_profile()
    r = db.Reader(dbp)
    ok_total = int(r.info.num_total) == windows == int(info.n_instances)
    ok_distinct = int(r.info.num_distinct) == n_distinct
    ok_files = True
    for ff in (0, 17, 63):                                             # a sample of files read back: ascending, totals per file
        lo, hi, cn = r.read_file(ff)
        asc = bool(np.all((hi[1:] > hi[:-1]) | ((hi[1:] == hi[:-1]) & (lo[1:] > lo[:-1]))))
        ok_files = ok_files and asc and int(cn.astype(np.int64).sum()) == int(info.file_instances[ff])
    r.close()
    res = {
        "workload": "meryl count k=%d label=#0x5A (8 bits) on %d x %d bp reads (%.2f Gbp, 50x of a %d bp genome), pushed from host memory in "
                    "64 MiB pieces, batches of %.2f Gbases, runs may keep %.1f GB of HBM" % (k, reads, rl, reads * rl / 1e9, genome, args.batch / 1e9, args.budget / 1e9),
        "ok": bool(ok_total and ok_distinct and ok_files and wp["n_kmers"] == n_distinct),
        "checks": {"database_total_equals_valid_windows": ok_total, "distinct_matches": ok_distinct, "sampled_files_ascending_and_totals": ok_files},
        "out_of_core": ooc, "n_batches": p.n_batches, "n_instances": int(info.n_instances), "n_distinct": int(n_distinct),
        "host_bytes_pushed": int(host.size),
        "push_calls_s": t_push, "push_to_counted_s": t_count, "write_database_s": t_write, "host_to_database_s": t_count + t_write,
        "runs": rp, "db_write": wp,
        "merge_once_s": rp["deliver_s"], "spill_to_host_s": rp["spill_s"], "peak_hbm_gb": rp["peak_hbm_bytes"] / 1e9,
        "result_bytes": int(n_distinct) * (8 * (2 if k > 32 else 1) + 4),
        "database_bytes": int(wp["data_bytes"]), "input_generation_s": t_gen,
        "upload_inclusive_rate_Gbases_per_s": host.size / 1e9 / t_count,
    }
finally:
    shutil.rmtree(outdir, ignore_errors=True)
line = json.dumps(res)
print(line)
if args.out:
    open(args.out, "w").write(line + "\n")
sys.exit(0 if res.get("ok") else 1)
