#!/usr/bin/env python3
"""Owner-side count of ONE file as large as an 8-GPU run makes them (developer tool).
A file at N ranks holds N x the k-mers of a single-GPU file; this times mgc_count_partitioned on such a file.
Usage: python scripts/bigfile_bench.py [n_keys] [pool]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from meryl_amd import capi, count
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_080_000_000
pool_n = int(sys.argv[2]) if len(sys.argv) > 2 else 150_000_000
k = 21
g = torch.Generator(device="cuda"); g.manual_seed(3)
pool = torch.randint(0, 1 << 36, (pool_n,), dtype=torch.int64, device="cuda", generator=g)
idx = torch.randint(0, pool_n, (n,), dtype=torch.int64, device="cuda", generator=g)
keys = pool[idx]                      # file 0: top six bits zero
del idx
want_d = int(torch.unique(pool).numel()) if pool_n <= 200_000_000 else -1
fc = np.zeros(64, dtype=np.uint64); fc[0] = n
cfg = capi.configure(k, n * 21, 64 << 30)
s = count.Session(cfg, 0)
for rep in range(3):
    kk = keys.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.count_partitioned(kk, fc)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    info = s.info()
    print("n=%d keys in one file: %.1f ms  (%.2f ms per 135M keys), distinct %d (pool distinct %d)" %
          (n, dt * 1e3, dt * 1e3 / (n / 135e6), info.n_distinct, want_d), flush=True)
