// Vector-memory issue-rate microbenchmark (diagnostic, round 5): what does ONE vector memory instruction of each shape the DP
// kernels use cost the CU's address / L1 pipeline (TA + TCP), with 24 waves per CU (6 per SIMD) each working in a scratch region of
// its own, as k_poa / k_window do?  Reported: wave-instructions per CU-cycle and its inverse (CU-cycles per instruction).
//   hipcc --offload-arch=gfx950 -O3 -o vmem_rates vmem_rates.hip && ./vmem_rates [region KB per wave, default 256]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(64) void k(int iters, uint8_t* base, size_t region, int* out) {
  const int lane = threadIdx.x;
  uint8_t* p = base + (size_t)blockIdx.x * region;
  const unsigned mask = (unsigned)region - 1;                 // region is a power of two
  unsigned rnd = (blockIdx.x * 64 + lane) * 2654435761u + 12345u;
  unsigned off = 0;
  int acc = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      rnd = rnd * 1664525u + 1013904223u;
      const unsigned r = (rnd >> 8) & mask;
      if (MODE == 0) { p[(off + lane) & mask] = (uint8_t)i; off += 64; }                                   // byte store, 64 contiguous lanes
      if (MODE == 1) { ((unsigned*)p)[((off >> 2) + lane) & (mask >> 2)] = i; off += 256; }               // dword store, contiguous
      if (MODE == 2) { if (lane == 0) ((unsigned*)p)[(off >> 2) & (mask >> 2)] = i; off += 12; }         // dword store, one lane
      if (MODE == 3) { if (lane == 0) { unsigned* q = (unsigned*)p + ((off >> 2) & (mask >> 2) & ~3u); q[0] = i; q[1] = i; q[2] = i; } off += 16; }   // three dwords, one lane
      if (MODE == 4) { ((unsigned short*)p)[((off >> 1) + lane) & (mask >> 1)] = (unsigned short)i; off += 128; }   // short store, contiguous
      if (MODE == 5) { acc += ((const unsigned*)p)[((off >> 2) + lane) & (mask >> 2)]; off += 256; }      // dword load, contiguous, independent
      if (MODE == 6) { acc += ((const unsigned*)p)[r >> 2]; }                                              // dword gather, random in the region, independent
      if (MODE == 7) { acc += p[r]; }                                                                       // byte gather
      if (MODE == 8) { off = ((const unsigned*)p)[((off + r) & mask) >> 2] & mask; }                     // dword gather, DEPENDENT chain (latency)
      if (MODE == 9) { ((unsigned*)p)[r >> 2] = i; }                                                        // dword scatter
      if (MODE == 10) { acc += ((const unsigned*)p)[(((r >> 2) & ~63u) + lane) & (mask >> 2)]; }          // dword load, contiguous 256 B at a random place
      if (MODE == 11) { acc += ((const unsigned*)p)[(((r >> 2) & ~15u) + (lane & 15) + 1024 * (lane >> 4)) & (mask >> 2)]; }   // 4 segments of 64 B
      if (MODE == 12) { const uint4 v = ((const uint4*)p)[(((r >> 4) & ~63u) + lane) & (mask >> 4)]; acc += v.x + v.y + v.z + v.w; }   // 16 B per lane, contiguous 1 KB
    }
  }
  if (acc + (int)off == 0x7fffffff) out[0] = 1;
}
int main(int argc, char** argv) {
  const size_t region = (size_t)(argc > 1 ? atoi(argv[1]) : 256) << 10;
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const double ghz = pr.clockRate * 1e-6;
  const char* names[] = {"store byte x64 contiguous", "store dword x64 contiguous (256 B)", "store dword, lane 0 only", "store 3 dwords, lane 0 only", "store short x64 contiguous",
                         "load dword x64 contiguous, streaming", "load dword gather (random in region)", "load byte gather (random)", "load dword gather, dependent chain", "store dword scatter (random)",
                         "load dword x64 contiguous at random place", "load dword, 4 segments of 64 B", "load 16 B per lane, 1 KB contiguous at random place"};
  int* out; hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("CUs %d clock %.2f GHz, region per wave %zu KB\n", pr.multiProcessorCount, ghz, region >> 10);
  for (int wps : {6, 3, 1}) {
    const int blocks = pr.multiProcessorCount * 4 * wps;
    uint8_t* base; if (hipMalloc(&base, region * blocks) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(base, 1, region * blocks);
    for (int mode = 0; mode < 13; ++mode) {
      const int iters = mode == 8 ? 200 : 1000;
      auto launch = [&]() {
#define L(M) case M: hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(64), 0, 0, iters, base, region, out); break;
        switch (mode) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) } };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double inst_per_cu = 8.0 * iters * wps * 4 * (mode == 3 ? 1 : 1);
      const double cyc = ms * 1e-3 * ghz * 1e9;
      printf("%d waves/SIMD  %-52s %8.3f ms  %.4f instr / CU-cycle = %7.1f CU-cycles per wave-instruction\n", wps, names[mode], ms, inst_per_cu / cyc, cyc / inst_per_cu);
      fflush(stdout);
    }
    hipFree(base);
  }
  return 0;
}
