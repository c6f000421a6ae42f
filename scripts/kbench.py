#!/usr/bin/env python3
"""One bench line for another k / read shape than the judged workload (BASELINE configs 3-5, single-GPU legs), in the
format of bench.py with the `roofline` object of the dominant grouping/sort pass.
usage: python scripts/kbench.py K [reads] [compress 0|1] [read_len] [repeat_ppm] [label_bits] [repeat_unit] [repeat_families]"""
import json
import sys
import time

sys.path.insert(0, '.')
import torch  # noqa: E402
from meryl_amd import capi, count  # noqa: E402

k = int(sys.argv[1]); reads = int(sys.argv[2]) if len(sys.argv) > 2 else 33_333_334
compress = int(sys.argv[3]) if len(sys.argv) > 3 else 0
read_len = int(sys.argv[4]) if len(sys.argv) > 4 else 150
repeat_ppm = int(sys.argv[5]) if len(sys.argv) > 5 else 0      # e.g. 100000 = 10 % of the genome in repeat families
label_bits = int(sys.argv[6]) if len(sys.argv) > 6 else 0
repeat_unit = int(sys.argv[7]) if len(sys.argv) > 7 else 300        # bases per repeat block; families: how many different units (1: one satellite)
repeat_families = int(sys.argv[8]) if len(sys.argv) > 8 else 1000
steps = 3
bases = count.dev_synth_reads(20240917, reads * read_len // 30, 0, reads, read_len, 5000, 100, repeat_ppm=repeat_ppm, repeat_unit=repeat_unit, repeat_families=repeat_families)
torch.cuda.synchronize()
cfg = capi.configure(k, reads * read_len, 64 << 30, homopoly_compress=compress, label_size=label_bits, label=5)
s = count.Session(cfg, 0)
s.push_bases_device(bases)
s.set_profiling(True)
s.count(); torch.cuda.synchronize()
acc = {"ms": 0.0, "launches": 0, "keys": 0, "bytes": 0, "stage": [0.0] * capi.NUM_STAGES}
t0 = time.perf_counter()
for _ in range(steps):
    s.count()
    p = s.profile()
    acc["ms"] += p.sort_pass_ms_total; acc["launches"] += p.sort_pass_launches; acc["keys"] += p.sort_pass_keys
    acc["bytes"] += p.pass_bytes[0] + p.pass_bytes[1]          # key bytes really read + written (narrowed passes move fewer)
    for i in range(capi.NUM_STAGES):
        acc["stage"][i] += p.stage_ms[i]
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
i = s.info()
kb = 16 if k > 32 else 8
line = {"metric": "distinct k-mers counted/sec", "value": i.n_distinct / dt, "unit": "distinct k-mers/s", "n_gpus": 1,
        "steps": steps, "ms_per_step": dt * 1e3, "dtype": "u128" if k > 32 else "u64", "data": "synthetic",
        "config": {"workload": "meryl count k=%d%s: %d x %d bp reads (%.2f Gbp, 30x), %d ppm of the genome in repeat families, inputs resident in HBM"
                               % (k, " compress" if compress else "", reads, read_len, reads * read_len / 1e9, repeat_ppm),
                   "k": k, "compress": compress, "n_instances": i.n_instances, "n_distinct": i.n_distinct, "w_prefix": i.w_prefix,
                   "bases": int(bases.numel())},
        "ms_per_Gbp": dt * 1e3 / (reads * read_len / 1e9),
        "stage_ms_per_step": {capi.STAGE_NAMES[j]: acc["stage"][j] / steps for j in range(capi.NUM_STAGES)}}
if acc["launches"]:
    ach = acc["bytes"] / (acc["ms"] / 1e3) / 1e9
    line["roofline"] = {"kernel": "grouping / stable radix pass over a file's k-mers (%d B keys)" % kb, "bound": "hbm", "achieved": ach,
                        "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": None, "launches": acc["launches"],
                        "avg_launch_ms": acc["ms"] / acc["launches"],
                        "algorithmic_bytes_per_key_per_pass": acc["bytes"] / acc["keys"]}
print(json.dumps(line))
