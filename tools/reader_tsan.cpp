// ThreadSanitizer driver of the native reader on a plain .gz file with several inflating threads (parallel decoder + the hand-over to the one parser,
// csrc/c3_io.cpp: gzpar_thread / gzpar_read / gzpar_swap).  CPU only (page-locking falls back to malloc without a GPU):
//   cd c3poa_amd/csrc && g++ -O1 -g -fsanitize=thread -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include ../../tools/reader_tsan.cpp c3_io.cpp -L/opt/rocm/lib -lamdhip64 -lz -lpthread -Wl,-rpath,/opt/rocm/lib -o /tmp/rd_tsan
//   C3_GZ_THREADS=6 C3_GZ_CHUNK=300000 /tmp/rd_tsan some.fastq.gz      (round 6: 40 MB of FASTQ-shaped gzip -6, 3 passes, 60 000 reads: no report)
#include "../include/c3poa.h"
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv) {
  long total = 0, bases = 0;
  for (int rep = 0; rep < 3; ++rep) {
    c3_reader* r = nullptr;
    if (c3_reader_open(argv[1], 2, &r) != 0) { printf("open failed\n"); return 1; }
    for (;;) {
      c3_host_batch b;
      const int rc = c3_reader_next(r, 3000, 1ll << 40, 0, &b);
      if (rc != 0) { printf("rc %d: %s\n", rc, c3_reader_error(r)); return 2; }
      if (b.n == 0) break;
      total += b.n;
    }
    c3_reader_close(r);
  }
  printf("reads %ld\n", total);
}
