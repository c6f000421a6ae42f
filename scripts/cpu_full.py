#!/usr/bin/env python3
"""The CPU port (oracle/oracle_port.cpp, the reference's threaded algorithm restated) over the WHOLE judged workload
(66,666,667 x 150 bp, k=21, wPrefix 18) on this box's host cores -> JSON (commit it as profiles/rNN_cpu_full.json;
bench.py's cpu_baseline.full_workload reads it).  Usage: python scripts/cpu_full.py OUT.json [threads ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import oracle
from meryl_amd import count

out = sys.argv[1]
threads = [int(x) for x in sys.argv[2:]] or [32]
oracle.build()
bases = count.dev_synth_reads(bench.SEED, bench.GENOME_LEN, 0, bench.DEFAULT_READS, bench.READ_LEN, 5000, 100).cpu().numpy()
cfg = oracle.configure_counting(bench.K, 10_000_000_000, 64 << 30)
res = {"workload": "the whole judged workload: %d x %d bp reads, k=%d, wPrefix=%d" % (bench.DEFAULT_READS, bench.READ_LEN, bench.K, cfg["w_prefix"]),
       "kind": "port", "host_cores": os.cpu_count(), "runs": []}
for th in threads:
    t0 = time.perf_counter()
    _, nd, ni = oracle.digest_threaded(bases, bench.K, cfg["w_prefix"], oracle.CANONICAL, th)
    dt = time.perf_counter() - t0
    res["runs"].append({"threads": th, "seconds": dt, "n_distinct": int(nd), "n_instances": int(ni),
                        "distinct_per_s": nd / dt, "instances_per_s": ni / dt})
    print(res["runs"][-1], flush=True)
best = min(res["runs"], key=lambda r: r["seconds"])
res.update({"value": best["distinct_per_s"], "unit": "distinct k-mers/s", "cores": best["threads"], "seconds": best["seconds"],
            "instances_per_s": best["instances_per_s"]})
json.dump(res, open(out, "w"), indent=1)
