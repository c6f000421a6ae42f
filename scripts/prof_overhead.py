import sys, time
sys.path.insert(0, '.')
import torch
from meryl_amd import capi, count
bases = count.dev_synth_reads(20240917, 333333334, 0, 66666667, 150, 5000, 100)
torch.cuda.synchronize()
cfg = capi.configure(21, 10_000_000_000, 64 << 30)
s = count.Session(cfg, 0)
s.push_bases_device(bases)
for prof in (False, True, False, True):
    s.set_profiling(prof)
    s.count()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        s.count()
    torch.cuda.synchronize()
    print("profiling", prof, "%.1f ms/step" % ((time.perf_counter() - t0) / 3 * 1e3))
