import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meryl_amd import capi, count
torch.cuda.set_device(0); capi.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_200_000_000
g = torch.Generator(device="cuda"); g.manual_seed(3)
keys = torch.randint(0, 1 << 28, (n,), dtype=torch.int64, device="cuda", generator=g) * 3 + 1   # ~7x duplication
t0 = time.perf_counter()
out = count.dev_radix_sort(keys.clone(), 0, 42)
torch.cuda.synchronize(); print("sort %.1f ms" % ((time.perf_counter() - t0) * 1e3))
bad = int((out[1:] < out[:-1]).sum().item())
print("n", n, "unsorted pairs", bad, "sum equal", bool(out.sum().item() == keys.sum().item()))
u, c = count.dev_run_length(out)
ref = torch.unique_consecutive(out)
print("distinct ours", u.numel(), "torch", ref.numel(), "equal", bool(u.numel() == ref.numel() and torch.equal(u, ref)), "count sum", int(c.to(torch.int64).sum().item()))
