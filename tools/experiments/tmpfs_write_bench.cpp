#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <chrono>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 8;
  const size_t total = (size_t)(argc > 2 ? atof(argv[2]) : 2.0) * (1u << 30), chunk = (size_t)96 << 20;   // each job ~96 MB like a slice
  std::vector<char> src(chunk, 'A');
  const char* path = "/dev/shm/wbench.bin";
  for (int mode = 0; mode < 3; ++mode) {
    unlink(path);
    int fd = open(path, O_RDWR | O_CREAT, 0644);
    const size_t njobs = total / chunk;
    if (mode == 1) { if (ftruncate(fd, (off_t)(njobs * chunk)) != 0) return 1; }
    double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t]() {
      int f2 = fd; char p2[64];
      if (mode == 2) { snprintf(p2, sizeof p2, "/dev/shm/wbench.%d.bin", t); unlink(p2); f2 = open(p2, O_RDWR | O_CREAT, 0644); }
      for (size_t j = t; j < njobs; j += T) {
        off_t at = (off_t)(j * chunk) + 13;      // unaligned like real slices
        if (mode == 2) at = (off_t)((j / T) * chunk);
        if (mode == 1) {
          off_t a0 = at & ~(off_t)4095; size_t d = (size_t)(at - a0);
          size_t len = chunk - 64;
          char* m = (char*)mmap(nullptr, len + d, PROT_READ | PROT_WRITE, MAP_SHARED, fd, a0);
          if (m == MAP_FAILED) { perror("mmap"); exit(1); }
          memcpy(m + d, src.data(), len);
          munmap(m, len + d);
        } else {
          size_t n = chunk - 64; const char* p = src.data(); off_t o = at;
          while (n) { ssize_t w = pwrite(f2, p, n, o); if (w <= 0) { perror("pwrite"); exit(1); } p += w; n -= w; o += w; }
        }
      }
      if (mode == 2) { close(f2); unlink(p2); }
    });
    for (auto& x : th) x.join();
    double dt = now() - t0;
    printf("%s: %d threads, %.1f GB in %.2f s = %.2f GB/s\n", mode == 0 ? "pwrite, one file" : mode == 1 ? "mmap+memcpy, one file" : "pwrite, one file per thread", T, total / 1e9, dt, total / 1e9 / dt);
    close(fd); unlink(path);
  }
}
