#!/usr/bin/env python3
"""Time of hash_count*_huge_kernel on one oversized sub-bucket (developer tool; run under rocprofv3 --kernel-trace --stats).
Usage: python scripts/huge_bench.py K N_INST N_DISTINCT HEAVY_PERCENT"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from meryl_amd import capi, count
k = int(sys.argv[1]); n_inst = int(sys.argv[2]); n_dist = int(sys.argv[3]); heavy = int(sys.argv[4])
rng = np.random.default_rng(1)
plen = min(20, k - 10)
prefix = "AAC" + "".join("ACGT"[i] for i in rng.integers(0, 4, plen - 3))
tails = ["".join("ACGT"[i] for i in rng.integers(0, 4, k - plen)) for _ in range(n_dist)]
pick = rng.integers(0, n_dist, n_inst)
pick[rng.integers(0, 100, n_inst) < heavy] = 0
stream = ".".join(prefix + tails[int(i)] for i in pick) + "."
filler = count.dev_synth_reads(3, 2_000_000, 0, 200_000, 150, 5000, 100).cpu().numpy().tobytes().decode()
stream += filler
cfg = capi.configure(k, len(stream), 1 << 30, 1)
with count.Session(cfg, 0) as s:
    s.push_bases(stream, end_of_sequence=False)
    s.count(); torch.cuda.synchronize()
    t0 = time.perf_counter(); s.count(); torch.cuda.synchronize()
    print("k=%d n=%d distinct<=%d heavy=%d%%: count %.2f ms, %d distinct" % (k, n_inst, n_dist, heavy, (time.perf_counter() - t0) * 1e3, s.info().n_distinct))
