#!/usr/bin/env python3
"""ms per mgc_count for another k than the judged bench uses (developer tool).  Usage: python scripts/kbench.py K [reads]"""
import sys, time
sys.path.insert(0, '.')
import torch
from meryl_amd import capi, count
k = int(sys.argv[1]); reads = int(sys.argv[2]) if len(sys.argv) > 2 else 33_333_334
compress = int(sys.argv[3]) if len(sys.argv) > 3 else 0
read_len = int(sys.argv[4]) if len(sys.argv) > 4 else 150
repeat_ppm = int(sys.argv[5]) if len(sys.argv) > 5 else 0      # e.g. 100000 = 10 % of the genome in repeat families
bases = count.dev_synth_reads(20240917, reads * read_len // 30, 0, reads, read_len, 5000, 100, repeat_ppm=repeat_ppm)
torch.cuda.synchronize()
cfg = capi.configure(k, reads * read_len, 64 << 30, homopoly_compress=compress)
s = count.Session(cfg, 0)
s.push_bases_device(bases)
s.set_profiling(True)
s.count(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2):
    s.count()
torch.cuda.synchronize()
p = s.profile(); i = s.info()
print("k=%d compress=%d read_len=%d repeats=%dppm reads=%d: %.1f ms/step, %d instances, %d distinct, stages %s" % (k, compress, read_len, repeat_ppm, reads, (time.perf_counter() - t0) / 2 * 1e3, i.n_instances,
      i.n_distinct, ["%.1f" % x for x in list(p.stage_ms)[:capi.NUM_STAGES]]))
