#!/usr/bin/env python3
"""BASELINE config 5's mechanics at a size that means something: k=51 with a constant label, the input pushed FROM THE HOST
(pinned double-buffered uploads) in >= 3 forced batches of >= 2 Gbp each, every batch counted by the worker thread while
the next one uploads, merged on the device into the running result -- compared, by per-file digests, with the single pass
over the same bases resident in HBM.  Prints one JSON line (wall clocks, merge time, batches).
usage: python scripts/ooc_bench.py [reads=40000000] [batch_bases=2000000000] [k=51]"""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from meryl_amd import capi, count  # noqa: E402
from test_gpu_parity import device_digests  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 51
bases = count.dev_synth_reads(5, reads * 150 // 50, 0, reads, 150, 5000, 100)              # 50x of its genome
cfg = capi.configure(k, bases.numel(), 64 << 30, label_size=8, label=0x5A)
with count.Session(cfg, 0) as s:
    s.push_bases_device(bases)
    s.count(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.count()
    t_single = time.perf_counter() - t0
    info1 = s.info()
    k1, c1 = s.result_device()
    want = device_digests(torch, k1, c1, k)
    del k1, c1
host = bases.cpu().numpy()
del bases
torch.cuda.empty_cache()
piece = 64 << 20
with count.Session(cfg, 0) as s:
    s.set_batch_bases(batch)
    t0 = time.perf_counter()
    L = capi.lib()
    ptr = host.ctypes.data
    for a in range(0, host.size, piece):
        n = min(piece, host.size - a)
        capi.check(L.mgc_push_bases(s._h, ctypes.cast(ptr + a, ctypes.c_char_p), n, 0), "mgc_push_bases", s._h)
    t_push = time.perf_counter() - t0
    s.count()
    t_total = time.perf_counter() - t0
    info = s.info()
    p = s.profile()
    k2, c2 = s.result_device()
    got = device_digests(torch, k2, c2, k)
ok = bool(np.array_equal(got, want)) and info.n_instances == info1.n_instances and info.n_distinct == info1.n_distinct
print(json.dumps({
    "workload": "meryl count k=%d label=#0x5A (8 bits) on %d x 150 bp reads (%.2f Gbp, 50x), pushed from host memory in 64 MiB pieces, "
                "batches of %.2f Gbases" % (k, reads, reads * 150 / 1e9, batch / 1e9),
    "equal_to_single_pass": ok, "n_batches": p.n_batches, "device_merge_ms_total": p.merge_ms,
    "host_push_to_result_s": t_total, "of_which_push_calls_s": t_push, "single_pass_resident_s": t_single,
    "n_instances": info.n_instances, "n_distinct": info.n_distinct,
    "host_bytes_pushed": int(host.size), "upload_inclusive_rate_Gbases_per_s": host.size / 1e9 / t_total}))
sys.exit(0 if ok else 1)
