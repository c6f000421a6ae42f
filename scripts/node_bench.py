#!/usr/bin/env python3
"""mgc_count_node on ONE GPU with several ranks placed on it (the in-process node count: per-rank extraction, peer-copy
pulls in waves, owner-side count, device-encoded parts, stitch): what the plan costs beside a plain single-session
count + database write of the same reads.  Not a scaling number -- every rank shares the one device.
usage: python scripts/node_bench.py [reads] [k]"""
import json
import shutil
import sys
import time

sys.path.insert(0, '.')
import torch  # noqa: E402
from meryl_amd import capi, count  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 13_333_334
k = int(sys.argv[2]) if len(sys.argv) > 2 else 21
L = 150
d = count.dev_synth_reads(2, reads * L // 30, 0, reads, L, 5000, 100)
torch.cuda.synchronize()
cfg = capi.configure(k, 10_000_000_000, 64 << 30)
out = "/dev/shm/mgc_node_bench"
res = {"reads": reads, "bases": int(d.numel()), "k": k, "runs": []}
for rep in range(2):
    shutil.rmtree(out, ignore_errors=True)
    t0 = time.perf_counter()
    with count.Session(cfg, 0) as s:
        s.push_bases_device(d)
        s.count()
        prof = s.write_database(out, 16)
        nd = s.info().n_distinct
    res["single_session_s"] = time.perf_counter() - t0
res["n_distinct"] = nd
for n in (1, 2, 4, 8):
    rec = L + 1
    cuts = [(reads * i // n) * rec for i in range(n + 1)]
    slices = [d[cuts[i]:cuts[i + 1]] for i in range(n)]
    for rep in range(2):
        shutil.rmtree(out, ignore_errors=True)
        t0 = time.perf_counter()
        p = count.count_node(cfg, slices, out, devices=[0] * n, host_threads=16)
        wall = time.perf_counter() - t0
    assert p["n_distinct"] == nd
    res["runs"].append(dict(ranks=n, wall_s=wall, **{q: p[q] for q in ("bucket_bits", "partition_s", "exchange_count_s", "close_s", "merge_parts_s", "total_s")}))
shutil.rmtree(out, ignore_errors=True)
print(json.dumps(res))
