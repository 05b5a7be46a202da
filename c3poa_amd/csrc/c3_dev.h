// c3_dev.h -- shared host/device definitions of the HIP backend (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/c3poa.h"

#define C3_NEG (-(1 << 28))    // out-of-band / unreachable score
#define C3_NEG2 (-(1 << 30))   // "no left neighbour" of the horizontal-gap state
#define C3_MAX_SUB 250
#define C3_SPLINT_MAX 512
#define C3_JUMP_LEVELS 18   // binary-lifting levels of the consensus path (graphs up to 2^18 nodes)      // 64 lanes x 8 rows per lane in the conk kernel

// device view of the resident batch
struct C3Batch {
  int n;
  const uint32_t* pk;       // 2-bit packed bases, 16 per word, every read starts on a word
  const int64_t* woff;      // [n+1] word offsets
  const uint8_t* qual;      // Phred+33 bytes, concatenated
  const int64_t* off;       // [n+1] base offsets
  const uint8_t* strand;    // '+', '-', other = not assigned
  const int16_t* splint_id;
};

// per-read record on the device; mirrors c3_read_result (copied out verbatim)
typedef c3_read_result C3Info;

struct C3Params {
  int conk_match, conk_mismatch, conk_penalty;
  int sg_iters, sg_window, sg_order, mdist;
  int poa_match, poa_mismatch, o1, e1, o2, e2, band_b;
  double band_f;
  int pol_match, pol_mismatch, pol_gap, pol_window, pol_q, dang_band;
  int zero, zr_match, zr_mismatch, zr_gapo, zr_gape, zr_min_score, zr_max_cells;
};

__device__ __forceinline__ int c3_code_at(const uint32_t* pk, int64_t i) {
  return (pk[i >> 4] >> ((i & 15) * 2)) & 3;
}

// ---- wave (64-lane) primitives ----------------------------------------------------------
__device__ __forceinline__ int wave_lane() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int wave_bcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ int wave_first(int v) { return __builtin_amdgcn_readfirstlane(v); }

// value of lane-1; lane 0 receives `carry`.  wave_shr:1 is a single DPP move on gfx9-family.
__device__ __forceinline__ int wave_shr1(int v, int carry) {
  return __builtin_amdgcn_update_dpp(carry, v, 0x138, 0xf, 0xf, false);
}

// value of lane-1; lane 0 receives 0 (bound_ctrl: no register has to be preset)
__device__ __forceinline__ int wave_shr1z(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);
}

// 16-bit signed max of the low halves, zero-extended (VOP2 16-bit: twice the issue rate of v_max_i32 on gfx950)
__device__ __forceinline__ int max16(int a, int b) { int d; asm("v_max_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }

// value of lane+1; lane 63 receives `carry` (wave_shl:1)
__device__ __forceinline__ int wave_shl1(int v, int carry) {
  return __builtin_amdgcn_update_dpp(carry, v, 0x130, 0xf, 0xf, false);
}

// value of lane+1; lane 63 receives 0 (bound_ctrl: no register has to be preset)
__device__ __forceinline__ int wave_shl1z(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true);
}

// three independent inclusive max-scans, interleaved step by step: a DPP instruction reading a register written by the
// previous VALU instruction needs two wait states, which the other two chains fill
__device__ __forceinline__ void wave_scan_max3(int& a, int& b, int& c) {
  const int ID = INT32_MIN;
#define C3_SCAN_STEP(ctrl, rm) { \
    const int ta = __builtin_amdgcn_update_dpp(ID, a, ctrl, rm, 0xf, false); \
    const int tb = __builtin_amdgcn_update_dpp(ID, b, ctrl, rm, 0xf, false); \
    const int tc = __builtin_amdgcn_update_dpp(ID, c, ctrl, rm, 0xf, false); \
    a = max(a, ta); b = max(b, tb); c = max(c, tc); }
  C3_SCAN_STEP(0x111, 0xf) C3_SCAN_STEP(0x112, 0xf) C3_SCAN_STEP(0x114, 0xf) C3_SCAN_STEP(0x118, 0xf)
  C3_SCAN_STEP(0x142, 0xa) C3_SCAN_STEP(0x143, 0xc)
#undef C3_SCAN_STEP
}

// inclusive max-scan over the 64 lanes (DPP row shifts + row broadcasts)
__device__ __forceinline__ int wave_scan_max(int x) {
  const int ID = INT32_MIN;
  int t;
  t = __builtin_amdgcn_update_dpp(ID, x, 0x111, 0xf, 0xf, false); x = max(x, t);  // row_shr:1
  t = __builtin_amdgcn_update_dpp(ID, x, 0x112, 0xf, 0xf, false); x = max(x, t);  // row_shr:2
  t = __builtin_amdgcn_update_dpp(ID, x, 0x114, 0xf, 0xf, false); x = max(x, t);  // row_shr:4
  t = __builtin_amdgcn_update_dpp(ID, x, 0x118, 0xf, 0xf, false); x = max(x, t);  // row_shr:8
  t = __builtin_amdgcn_update_dpp(ID, x, 0x142, 0xa, 0xf, false); x = max(x, t);  // row_bcast:15
  t = __builtin_amdgcn_update_dpp(ID, x, 0x143, 0xc, 0xf, false); x = max(x, t);  // row_bcast:31
  return x;
}
__device__ __forceinline__ int wave_max(int x) { return wave_bcast(wave_scan_max(x), 63); }
__device__ __forceinline__ int wave_min(int x) { return -wave_max(-x); }
__device__ __forceinline__ int wave_scan_add(int x) {
  int t;
  t = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false); x += t;
  t = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false); x += t;
  t = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false); x += t;
  t = __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false); x += t;
  t = __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); x += t;
  t = __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); x += t;
  return x;
}

// diagnostic build only (-DC3_PHASE_PROF): per-phase cycle sums, never in the shipped library
#ifdef C3_PHASE_PROF
#define PH_DECL unsigned long long ph_t0_ = __builtin_readcyclecounter(), ph_acc_[16] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
#define PH_MARK(i) { unsigned long long t_ = __builtin_readcyclecounter(); ph_acc_[i] += t_ - ph_t0_; ph_t0_ = t_; }
#define PH_FLUSH(p) if (wave_lane() == 0) { for (int i_ = 0; i_ < 16; ++i_) atomicAdd((p) + i_, ph_acc_[i_]); }
#else
#define PH_DECL
#define PH_MARK(i)
#define PH_FLUSH(p)
#endif

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return c3_hip_fail(h, e_, #x, __LINE__); } while (0)
