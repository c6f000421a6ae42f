// Reader-ring microbenchmark: the read side of mgc_push_text_file without the device (does the ring let N readers run in
// parallel on this box, into malloc'ed and into hipHostMalloc'ed buffers?).  hipcc ring_read.cpp -o ring_read -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <mutex>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>
int main(int argc, char **argv) {
  const char *path = argv[1]; int reader_threads = atoi(argv[2]); const bool pinned = atoi(argv[3]) != 0; const int R = argc > 4 ? atoi(argv[4]) : 8;
  int fd = open(path, O_RDONLY); struct stat st; fstat(fd, &st); uint64_t size = st.st_size;
  const size_t CH = 32u << 20; const uint64_t nchunks = (size + CH - 1) / CH;
  std::vector<char *> ring(R);
  for (int i = 0; i < R; i++) {
    if (pinned) { if (hipHostMalloc((void **)&ring[i], CH, hipHostMallocDefault) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; } }
    else ring[i] = (char *)malloc(CH);
    memset(ring[i], 1, CH);
  }
  std::mutex mu; std::condition_variable cv; std::vector<uint64_t> free_gen(R, 0), ready_chunk(R, ~0ull);
  std::atomic<uint64_t> next_chunk(0);
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  std::vector<double> tread(reader_threads, 0.0);
  auto reader = [&](int t) { for (;;) { const uint64_t c = next_chunk.fetch_add(1); if (c >= nchunks) return; const int slot = (int)(c % R);
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return free_gen[slot] == c / R; }); }
      const uint64_t off = c * CH; const size_t want = (size_t)std::min<uint64_t>(CH, size - off); size_t have = 0;
      const double a = now();
      while (have < want) { ssize_t r = pread(fd, ring[slot] + have, want - have, off + have); if (r <= 0) break; have += r; }
      tread[t] += now() - a;
      std::lock_guard<std::mutex> g(mu); ready_chunk[slot] = c; cv.notify_all(); } };
  double t0 = now();
  std::vector<std::thread> readers; for (int t = 0; t < reader_threads; t++) readers.emplace_back(reader, t);
  for (uint64_t c = 0; c < nchunks; c++) { const int slot = (int)(c % R);
    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return ready_chunk[slot] == c; }); }
    if (c >= 2) { std::lock_guard<std::mutex> g(mu); free_gen[(c - 2) % R]++; cv.notify_all(); } }
  for (auto &t : readers) t.join();
  double sum = 0; for (double x : tread) sum += x;
  printf("%2d readers, ring %2d, %s: %.2f GB in %.3f s = %5.1f GB/s; per-thread pread rate %.1f GB/s\n", reader_threads, R, pinned ? "pinned" : "malloc",
         size / 1e9, now() - t0, size / 1e9 / (now() - t0), size / 1e9 / sum);
}
