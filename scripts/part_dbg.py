#!/usr/bin/env python3
"""What the partition kernel's time is made of (measurements only): hist + partition of the judged 10 Gbp as they are, with every
key leaving as 4 bytes (MGC_PART_DBG=1: what a narrower key layout could buy on the write side) and with no global stores at all
(MGC_PART_DBG=2: read + extraction + ranking + exchange floor).  The outputs of the two debug forms are garbage."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from meryl_amd import capi, count  # noqa: E402

bases = count.dev_synth_reads(2, 333333334, 0, 66666667, 150, 5000, 100)
torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter(); h = count.HipOps.histogram(bases, 21, 0, 6); torch.cuda.synchronize(); th = time.perf_counter() - t0
print("histogram (64 files) alone: %.2f ms" % (th * 1e3))
for dbg in ("", "1", "2", ""):
    if dbg:
        os.environ["MGC_PART_DBG"] = dbg
    else:
        os.environ.pop("MGC_PART_DBG", None)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        keys, counts = count.dev_kmer_partition(bases, 21, 0, 6)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        del keys
    print("MGC_PART_DBG=%-2s hist + partition (+ torch alloc): %.2f ms  -> partition ~%.2f ms" % (dbg or "-", best * 1e3, (best - th) * 1e3), flush=True)
