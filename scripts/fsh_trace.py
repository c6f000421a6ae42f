#!/usr/bin/env python3
"""Where a forced-sharded step (one rank of the multi-GPU path on one GPU: partition -> exchange waves to itself -> owner-side
count of every wave) spends its wall clock: cProfile over two steps + the library's stage profile of the owner-side counts.
usage: python scripts/fsh_trace.py [reads]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bench  # noqa: E402
from meryl_amd import capi, count  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else bench.DEFAULT_READS
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
capi.lib()
bases = count.dev_synth_reads(bench.SEED, bench.GENOME_LEN, 0, reads, bench.READ_LEN, 5000, 100)
torch.cuda.synchronize()
count.count_sharded(bases, bench.K)                      # warm-up: arenas
torch.cuda.synchronize()
t0 = time.perf_counter()
count.count_sharded(bases, bench.K)
torch.cuda.synchronize()
print("one step: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(2):
    count.count_sharded(bases, bench.K)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("cumulative").print_stats(28)
# the way bench.py's step holds on to the previous result while the next one is counted
keep = {}
for i in range(3):
    t0 = time.perf_counter()
    keep["u"], keep["c"], keep["r"] = count.count_sharded(bases, bench.K)
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    print("step keeping the result: %.1f ms  (free %.1f GB of %.1f, torch reserved %.1f GB)" %
          ((time.perf_counter() - t0) * 1e3, free / 1e9, total / 1e9, torch.cuda.memory_reserved() / 1e9))
keep.clear()
# the owner-side sessions' own stage clocks
acc = {}
count.SHARD_PROFILE = acc
os.environ["MGC_SHARD_PROFILE"] = "1"
count.count_sharded(bases, bench.K)
print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in acc.items() if k != "by_pass"})
dist.destroy_process_group()
