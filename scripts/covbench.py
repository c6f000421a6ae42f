#!/usr/bin/env python3
"""k = 21 count of synthetic 150 bp reads at a chosen COVERAGE (the judged workload is 30x; 1x is the low-coverage case: D ~ N),
stage times + which plan the files took.  usage: python scripts/covbench.py coverage [reads] [k]"""
import json
import sys
import time

sys.path.insert(0, '.')
import torch  # noqa: E402
from meryl_amd import capi, count  # noqa: E402

cov = float(sys.argv[1]); reads = int(sys.argv[2]) if len(sys.argv) > 2 else 66_666_667
k = int(sys.argv[3]) if len(sys.argv) > 3 else 21
steps = 3
bases = count.dev_synth_reads(7, int(reads * 150 / cov), 0, reads, 150, 5000, 100)
torch.cuda.synchronize()
cfg = capi.configure(k, reads * 150, 64 << 30)
s = count.Session(cfg, 0)
s.push_bases_device(bases)
s.set_profiling(True)
s.count(); torch.cuda.synchronize()
stage = [0.0] * capi.NUM_STAGES
t0 = time.perf_counter()
for _ in range(steps):
    s.count()
    p = s.profile()
    for i in range(capi.NUM_STAGES):
        stage[i] += p.stage_ms[i]
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
i = s.info()
print(json.dumps({"coverage": cov, "k": k, "reads": reads, "ms_per_step": dt * 1e3, "n_instances": i.n_instances, "n_distinct": i.n_distinct,
                  "distinct_over_instances": i.n_distinct / max(1, i.n_instances),
                  "stage_ms": {capi.STAGE_NAMES[j]: stage[j] / steps for j in range(capi.NUM_STAGES)},
                  "stream_files": p.stream_files, "stream_retries": p.stream_retries, "probe_ratio": p.probe_ratio}))
