"""What the host side of file -> database can do on this box: tmpfs reads/writes from/to PINNED buffers by thread count
(the readers of mgc_push_text_file and the writers of the database stream do exactly this), and plain memcpy."""
import os
import sys
import threading
import time

import numpy as np
import torch

MB = 1 << 20
CH = 32 * MB
N = 96                      # chunks -> 3 GiB
shm = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
path = os.path.join(shm, "mgc_hostio.bin")
pinned = [torch.empty(CH, dtype=torch.uint8).pin_memory().numpy() for _ in range(32)]
for p in pinned:
    p[:] = 7


def run(nthreads, fn):
    idx = [0]
    lock = threading.Lock()

    def work(t):
        while True:
            with lock:
                i = idx[0]
                idx[0] += 1
            if i >= N:
                return
            fn(t, i)
    ts = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return N * CH / (time.perf_counter() - t0) / 1e9


for nt in (1, 2, 4, 8, 16, 32):
    if os.path.exists(path):
        os.unlink(path)
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
    w = run(nt, lambda t, i: os.pwrite(fd, memoryview(pinned[t]), i * CH))
    w2 = run(nt, lambda t, i: os.pwrite(fd, memoryview(pinned[t]), i * CH))          # pages already allocated
    r = run(nt, lambda t, i: os.preadv(fd, [memoryview(pinned[t])], i * CH))
    os.close(fd)
    print("%2d threads: pwrite new pages %.1f GB/s, pwrite existing pages %.1f GB/s, pread %.1f GB/s" % (nt, w, w2, r))
os.unlink(path)
a = np.empty(CH * 8, dtype=np.uint8)
b = np.empty(CH * 8, dtype=np.uint8)
b[:] = 1
t0 = time.perf_counter()
for _ in range(4):
    np.copyto(a, b)
print("memcpy pageable->pageable 1 thread %.1f GB/s" % (4 * a.size / (time.perf_counter() - t0) / 1e9))
t0 = time.perf_counter()
for _ in range(16):
    np.copyto(pinned[0], b[:CH])
print("memcpy pageable->pinned 1 thread %.1f GB/s" % (16 * CH / (time.perf_counter() - t0) / 1e9))
