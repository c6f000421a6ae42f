#!/usr/bin/env python3
"""VERDICT r5 item 1, the gate: how many DISTINCT k-mers does a sub-bucket hold when the file plan takes one or two grouping
bits fewer?  Counted from the result of the judged workload (k = 21, 10 Gbp): the sub-bucket of a distinct k-mer at `t` grouping
bits below the file is its top 6 + t bits, so the number of distinct k-mers per (6 + t)-bit prefix IS the D a count-kernel
iteration over that sub-bucket would have to hold; the instance counts summed per prefix are its keys.
usage: python scripts/gate_distinct.py [reads]          (prints one JSON object)"""
import json
import sys

sys.path.insert(0, '.')
import torch  # noqa: E402
from meryl_amd import capi, count  # noqa: E402

K = 21
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 66_666_667
bases = count.dev_synth_reads(2, 333_333_334 * reads // 66_666_667, 0, reads, 150, 5000, 100)
cfg = capi.configure(K, 10_000_000_000 if reads == 66_666_667 else reads * 150, 64 << 30)
s = count.Session(cfg, 0)
s.push_bases_device(bases)
s.count()
keys, cnts = s.result_device()
keys = keys.view(torch.int64)
info = s.info()
out = {"reads": reads, "k": K, "n_instances": int(info.n_instances), "n_distinct": int(info.n_distinct), "by_top_bits": {}}


def quant(x, qs):
    xs, _ = torch.sort(x)
    n = xs.numel()
    return {("p%g" % (q * 100)): int(xs[min(n - 1, int(q * n))]) for q in qs}


for t in (18, 17, 16, 15, 14):
    shift = 2 * K - 6 - t
    pre = keys >> shift                                       # k = 21: 42-bit keys, non-negative as int64
    nb = 1 << (6 + t)
    d = torch.bincount(pre, minlength=nb)
    n = torch.zeros(nb, dtype=torch.int64, device=keys.device)
    n.index_add_(0, pre, cnts.to(torch.int64) & 0xFFFFFFFF)
    nz = n > 0
    qs = (0.5, 0.9, 0.99, 0.999, 0.9999)
    row = {"sub_buckets": nb, "non_empty": int(nz.sum()), "suffix_bits": shift,
           "keys_mean": float(n[nz].float().mean()), "keys": quant(n[nz], qs), "keys_max": int(n.max()),
           "distinct_mean": float(d[nz].float().mean()), "distinct": quant(d[nz], qs), "distinct_max": int(d.max()),
           "sub_buckets_with_more_than_4094_keys": int((n > 4094).sum()),
           "of_the_ones_up_to_4094_keys": {"distinct_max": int(d[n <= 4094].max()),
                                           "distinct_over_1024": int(((d > 1024) & (n <= 4094)).sum()),
                                           "distinct_over_1400": int(((d > 1400) & (n <= 4094)).sum())}}
    out["by_top_bits"][str(t)] = row
    del pre, d, n, nz
print(json.dumps(out))
