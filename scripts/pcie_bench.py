"""Host <-> device copy rates of this box (pinned memory, hipMemcpyAsync via torch): what bounds the file -> database
wall clock once parsing, counting and encoding run on the device."""
import time
import torch

GB = 1 << 30
dev = torch.device("cuda", 0)
host = torch.empty(2 * GB, dtype=torch.uint8).pin_memory()
d = torch.empty(2 * GB, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()


def rate(fn, nbytes, reps=3):
    best = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = max(best, nbytes / (time.perf_counter() - t0) / 1e9)
    return best


print("H2D 2 GiB pinned       %.1f GB/s" % rate(lambda: d.copy_(host, non_blocking=True), 2 * GB))
print("D2H 2 GiB pinned       %.1f GB/s" % rate(lambda: host.copy_(d, non_blocking=True), 2 * GB))
for mb in (4, 32, 256):
    n = mb << 20
    k = (2 * GB) // n
    print("H2D %4d MiB pieces     %.1f GB/s" % (mb, rate(lambda: [d[i * n:(i + 1) * n].copy_(host[i * n:(i + 1) * n], non_blocking=True) for i in range(k)], 2 * GB)))
    print("D2H %4d MiB pieces     %.1f GB/s" % (mb, rate(lambda: [host[i * n:(i + 1) * n].copy_(d[i * n:(i + 1) * n], non_blocking=True) for i in range(k)], 2 * GB)))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def two_streams(h2d):
    for st, half in ((s1, 0), (s2, 1)):
        with torch.cuda.stream(st):
            a, b = half * GB, (half + 1) * GB
            if h2d:
                d[a:b].copy_(host[a:b], non_blocking=True)
            else:
                host[a:b].copy_(d[a:b], non_blocking=True)


print("H2D two streams        %.1f GB/s" % rate(lambda: two_streams(True), 2 * GB))
print("D2H two streams        %.1f GB/s" % rate(lambda: two_streams(False), 2 * GB))


def both():
    with torch.cuda.stream(s1):
        d[:GB].copy_(host[:GB], non_blocking=True)
    with torch.cuda.stream(s2):
        host[GB:].copy_(d[GB:], non_blocking=True)


print("H2D + D2H concurrently %.1f GB/s (sum)" % rate(both, 2 * GB))
pageable = torch.empty(GB, dtype=torch.uint8)
print("H2D 1 GiB pageable     %.1f GB/s" % rate(lambda: d[:GB].copy_(pageable), GB))
