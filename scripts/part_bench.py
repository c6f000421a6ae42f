import sys, time
sys.path.insert(0, '.')
import torch
from meryl_amd import capi, count
bases = count.dev_synth_reads(20240917, 333333334, 0, 66666667, 150, 5000, 100)
torch.cuda.synchronize()
for bits in (6, 7, 8, 9):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        keys, counts = count.dev_kmer_partition(bases, 21, 0, bits)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        del keys
    print("partition into 2^%d buckets: %.1f ms (hist + partition + allocs)" % (bits, dt * 1e3), flush=True)
