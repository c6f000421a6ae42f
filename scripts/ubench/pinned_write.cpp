// How fast can a CPU thread fill hipHostMalloc'ed memory on this box -- alone, while the GPU DMA-reads another pinned
// buffer, from a thread other than the allocating one, with different allocation flags?  (Why were the pread()s of
// mgc_push_text_file and the memcpy()s of mgc_push_bases running at 1-2 GB/s per thread?)
// hipcc -O2 pinned_write.cpp -o pinned_write -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t CH = 32u << 20; const int REP = 32;
  hipSetDevice(0);
  char *src = (char *)malloc(CH); memset(src, 3, CH);
  void *dev; hipMalloc(&dev, CH * 2);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  struct Case { const char *name; unsigned flags; };
  Case cases[] = {{"default", hipHostMallocDefault}, {"portable|mapped", hipHostMallocPortable | hipHostMallocMapped},
                  {"noncoherent", hipHostMallocNonCoherent}, {"coherent", hipHostMallocCoherent}, {"numa_user", hipHostMallocNumaUser}};
  for (auto &c : cases) {
    char *a = nullptr, *b = nullptr;
    if (hipHostMalloc((void **)&a, CH, c.flags) != hipSuccess || hipHostMalloc((void **)&b, CH, c.flags) != hipSuccess) { printf("%s: alloc failed\n", c.name); continue; }
    memset(a, 1, CH); memset(b, 1, CH);
    double t0 = now(); for (int i = 0; i < REP; i++) memcpy(a, src, CH); double alone = REP * CH / (now() - t0) / 1e9;
    // while the GPU reads b over and over
    std::atomic<bool> stop(false);
    std::thread dma([&] { hipSetDevice(0); while (!stop.load()) { hipMemcpyAsync(dev, b, CH, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); } });
    t0 = now(); for (int i = 0; i < REP; i++) memcpy(a, src, CH); double with_dma = REP * CH / (now() - t0) / 1e9;
    // ... and while the GPU reads THE SAME buffer that was just written (alternating a/b like the double buffers do)
    stop = true; dma.join();
    t0 = now();
    for (int i = 0; i < REP; i++) { char *w = (i & 1) ? a : b; memcpy(w, src, CH); hipMemcpyAsync(dev, w, CH, hipMemcpyHostToDevice, st); if (i) ; }
    hipStreamSynchronize(st);
    double pingpong = REP * CH / (now() - t0) / 1e9;
    // from another thread than the allocating one
    double other = 0; std::thread t([&] { double s0 = now(); for (int i = 0; i < REP; i++) memcpy(a, src, CH); other = REP * CH / (now() - s0) / 1e9; }); t.join();
    printf("%-16s memcpy into pinned: alone %.1f GB/s, during DMA of another buffer %.1f, write+upload ping-pong %.1f, from another thread %.1f\n",
           c.name, alone, with_dma, pingpong, other);
    hipHostFree(a); hipHostFree(b);
  }
  // allocation by a worker thread, written by it (the reader ring)
  std::thread w([&] { hipSetDevice(0); char *a = nullptr; hipHostMalloc((void **)&a, CH, hipHostMallocDefault); double t0 = now();
    for (int i = 0; i < REP; i++) memcpy(a, src, CH); printf("allocated+written by a worker thread: %.1f GB/s\n", REP * CH / (now() - t0) / 1e9); hipHostFree(a); });
  w.join();
  // 16 threads each filling their own pinned buffer
  { std::vector<char *> bufs(16); for (auto &p : bufs) hipHostMalloc((void **)&p, CH, hipHostMallocDefault);
    std::vector<std::thread> ts; double t0 = now();
    for (int i = 0; i < 16; i++) ts.emplace_back([&, i] { for (int r = 0; r < REP; r++) memcpy(bufs[i], src, CH); });
    for (auto &t : ts) t.join();
    printf("16 threads, own pinned buffers: %.1f GB/s total\n", 16.0 * REP * CH / (now() - t0) / 1e9);
    for (auto &p : bufs) hipHostFree(p); }
  return 0;
}
