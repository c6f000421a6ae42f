#!/usr/bin/env python3
"""A/B bench of the radix sort variants on one GPU (not the judged bench).
Usage: python scripts/sort_bench.py [n_keys] [bits]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meryl_amd import capi, count

n = int(sys.argv[1]) if len(sys.argv) > 1 else 135_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 36
variants = os.environ.get("SORT_VARIANTS", "0:8:16:512:1 0:8:16:512:0 0:9:16:512:1 0:8:16:1024:1 0:9:16:1024:1 1:8:16:512:1 1:9:16:512:1").split()

torch.cuda.set_device(0)
capi.lib()
g = torch.Generator(device="cuda"); g.manual_seed(1)
keys0 = torch.randint(0, 1 << bits, (n,), dtype=torch.int64, device="cuda", generator=g)
ref = None
for v in variants:
    f = (v.split(":") + ["1", "1", "0"])[:7]
    m, rb, kpt, blk, match, lb, flags = f
    os.environ.update(MGC_SORT_MODE=m, MGC_RADIX_BITS=rb, MGC_SORT_KPT=kpt, MGC_SORT_BLOCK=blk, MGC_SORT_MATCH=match,
                      MGC_SORT_LB=lb, MGC_SORT_FLAGS=flags)
    best = 1e9
    for it in range(3):
        k = keys0.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = count.dev_radix_sort(k, 0, bits)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    ok = bool((out[1:] >= out[:-1]).all().item())
    if ref is None:
        ref = out.clone()
    same = bool(torch.equal(out, ref))
    passes = (bits + int(rb) - 1) // int(rb)
    print("variant mode=%s rb=%s kpt=%s block=%s match=%s lb=%s flags=%s: %.2f ms  (%d passes, %.0f GB/s per-pass-equivalent incl. hist)  sorted=%s same=%s"
          % (m, rb, kpt, blk, match, lb, flags, best * 1e3, passes, 16.0 * n * passes / best / 1e9, ok, same), flush=True)
