// Issue-rate microbenchmark (diagnostic): how many scalar / vector instructions per cycle does one CU sustain, alone and
// mixed, at 1..8 waves per SIMD?  hipcc --offload-arch=gfx950 -O3 -o issue_rates issue_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
template <int MODE>
__global__ void k(int iters, int* out) {
  int a = threadIdx.x, b = 1, c2 = 2, d = 3;
  int s0 = 7, s1 = 1, s2 = 2, s3 = 3;
  asm volatile("" : "+s"(s0));
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { REP64(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");) }
    if (MODE == 1) { REP64(asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1" : "+v"(a), "+v"(b), "+v"(c2), "+v"(d));) }
    if (MODE == 2) { REP64(asm volatile("v_add_u32 %0, %0, 1\n s_add_u32 %4, %4, 1\n v_add_u32 %1, %1, 1\n s_add_u32 %5, %5, 1\n v_add_u32 %2, %2, 1\n s_add_u32 %6, %6, 1\n v_add_u32 %3, %3, 1\n s_add_u32 %7, %7, 1" : "+v"(a), "+v"(b), "+v"(c2), "+v"(d), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");) }
    if (MODE == 3) { REP64(asm volatile("v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c2), "+v"(d));) }
    if (MODE == 5) { REP64(asm volatile("v_pk_add_i16 %0, %0, %1\n v_pk_max_i16 %1, %1, %2\n v_pk_sub_i16 %2, %2, %3 clamp\n v_pk_mad_i16 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c2), "+v"(d));) }
    if (MODE == 6) { REP64(asm volatile("ds_bpermute_b32 %0, %1, %0\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1" : "+v"(a), "+v"(b), "+v"(c2), "+v"(d));) asm volatile("s_waitcnt lgkmcnt(0)"); }
    if (MODE == 7) { REP64(asm volatile("v_max3_i32 %0, %0, %1, %2\n v_max3_i32 %1, %1, %2, %3\n v_max3_i32 %2, %2, %3, %0\n v_max3_i32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c2), "+v"(d));) }
    if (MODE == 8) { REP64(asm volatile("v_bfe_i32 %0, %0, %1, 8\n v_bfe_i32 %1, %1, %2, 8\n v_bfe_i32 %2, %2, %3, 8\n v_bfe_i32 %3, %3, %0, 8" : "+v"(a), "+v"(b), "+v"(c2), "+v"(d));) }
    if (MODE == 9) { REP64(asm volatile("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %0\n v_add3_u32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c2), "+v"(d));) }
    if (MODE == 10) { REP64(asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_gt_i32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(a), "+v"(b), "+v"(c2), "+v"(d) : : "vcc");) }
    if (MODE == 11) { REP64(asm volatile("v_max_i32 %0, %0, %1\n v_subrev_u32 %1, 3, %1\n v_max_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c2), "+v"(d));) }
    if (MODE == 12) { REP64(asm volatile("v_max_i32 %1, %0, %1\n v_subrev_u32 %1, 3, %1\n v_max3_i32 %0, %2, %1, 0\n v_add_u32 %2, %3, %0" : "+v"(a), "+v"(b), "+v"(c2), "+v"(d));) }
    if (MODE == 13) { REP64(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1" : "+v"(a), "+v"(b), "+v"(c2), "+v"(d));) }
    if (MODE == 14) { REP64(asm volatile("v_and_or_b32 %0, %0, %1, %2\n v_lshl_add_u32 %1, %1, 2, %3\n v_alignbit_b32 %2, %2, %3, 5\n v_perm_b32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c2), "+v"(d));) }
    if (MODE == 4) { REP64(asm volatile("v_readlane_b32 %4, %0, 3\n v_add_u32 %1, %1, 1\n v_readlane_b32 %5, %2, 5\n v_add_u32 %3, %3, 1" : "+v"(a), "+v"(b), "+v"(c2), "+v"(d), "+s"(s0), "+s"(s1));) }
  }
  if (a + b + c2 + d + s0 + s1 + s2 + s3 == 0x7fffffff) out[0] = 1;
}
int main(int argc, char** argv) {
  int* out; hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"SALU only (4 per group)", "VALU only (4 per group)", "VALU+SALU interleaved (4+4)", "VALU DPP only (4)", "readlane+VALU (2+2)", "packed 16-bit VOP3P (4)", "bpermute+3 VALU", "v_max3_i32 (4)", "v_bfe_i32 (4)", "v_add3_u32 (4)", "v_cmp+v_cndmask (2+2)", "VOP2 max/sub/max/add (4)", "dependent max,sub,max3,add", "mov_dpp wave_shr + 3 VALU", "and_or/lshl_add/alignbit/perm"};
  const int per_iter[] = {256, 256, 512, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256};
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const double ghz = p.clockRate * 1e-6;
  printf("CUs %d clock %.2f GHz\n", p.multiProcessorCount, ghz); fflush(stdout);
  for (int mode = (argc > 1 ? atoi(argv[1]) : 0); mode < 15; ++mode)
    for (int wps : {1, 2, 4, 8}) {
      const int iters = 2000, blocks = p.multiProcessorCount * wps;       // 256-thread blocks: 4 waves = one per SIMD
      auto launch = [&]() {
        switch (mode) { case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, iters, out); break;
                        case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, iters, out); break;
                        case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, iters, out); break;
                        case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, iters, out); break;
                        case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, iters, out); break;
                        case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, iters, out); break;
                        case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, iters, out); break;
                        case 7: hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(256), 0, 0, iters, out); break;
                        case 8: hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, iters, out); break;
                        case 9: hipLaunchKernelGGL(k<9>, dim3(blocks), dim3(256), 0, 0, iters, out); break;
                        case 10: hipLaunchKernelGGL(k<10>, dim3(blocks), dim3(256), 0, 0, iters, out); break;
                        case 11: hipLaunchKernelGGL(k<11>, dim3(blocks), dim3(256), 0, 0, iters, out); break;
                        case 12: hipLaunchKernelGGL(k<12>, dim3(blocks), dim3(256), 0, 0, iters, out); break;
                        case 13: hipLaunchKernelGGL(k<13>, dim3(blocks), dim3(256), 0, 0, iters, out); break;
                        default: hipLaunchKernelGGL(k<14>, dim3(blocks), dim3(256), 0, 0, iters, out); } };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double inst_per_cu = (double)per_iter[mode] * iters * wps * 4;
      printf("%-30s %d waves/SIMD: %.3f ms  -> %.3f instr / (CU-cycle @%.2f GHz)  = %.3f per SIMD-cycle\n", names[mode], wps, ms,
             inst_per_cu / (ms * 1e-3 * ghz * 1e9), ghz, inst_per_cu / (ms * 1e-3 * ghz * 1e9) / 4); fflush(stdout);
    }
  return 0;
}
