// VALU cost table (diagnostic): SIMD cycles per wave64 instruction by operand shape, at 8 waves/SIMD.
// hipcc --offload-arch=gfx950 -O3 -o valu_cost valu_cost.hip && timeout 120 ./valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define CLOB "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "vcc", "s40", "s41", "s42", "s43"
// every mode: 4 independent instructions per group, 64 groups per iteration
#define MODES(X) \
  X(0, "add v,v,const", "v_add_u32 v20, v20, 1\n v_add_u32 v21, v21, 1\n v_add_u32 v22, v22, 1\n v_add_u32 v23, v23, 1") \
  X(1, "add v,v,v (other bank)", "v_add_u32 v20, v20, v25\n v_add_u32 v21, v21, v26\n v_add_u32 v22, v22, v27\n v_add_u32 v23, v23, v24") \
  X(2, "add v,v,v (same bank)", "v_add_u32 v20, v20, v24\n v_add_u32 v21, v21, v25\n v_add_u32 v22, v22, v26\n v_add_u32 v23, v23, v27") \
  X(3, "add v,v,v (same reg)", "v_add_u32 v20, v20, v20\n v_add_u32 v21, v21, v21\n v_add_u32 v22, v22, v22\n v_add_u32 v23, v23, v23") \
  X(4, "add v,s,v", "v_add_u32 v20, s40, v20\n v_add_u32 v21, s41, v21\n v_add_u32 v22, s42, v22\n v_add_u32 v23, s43, v23") \
  X(5, "add dst != src: v,v',const", "v_add_u32 v20, v24, 1\n v_add_u32 v21, v25, 1\n v_add_u32 v22, v26, 1\n v_add_u32 v23, v27, 1") \
  X(6, "add_e64 v,v,const (VOP3 enc)", "v_add_u32_e64 v20, v20, 1\n v_add_u32_e64 v21, v21, 1\n v_add_u32_e64 v22, v22, 1\n v_add_u32_e64 v23, v23, 1") \
  X(7, "max3 v,v,v", "v_max3_i32 v20, v20, v25, v30\n v_max3_i32 v21, v21, v26, v31\n v_max3_i32 v22, v22, v27, v28\n v_max3_i32 v23, v23, v24, v29") \
  X(8, "max3 v,v,const", "v_max3_i32 v20, v20, v25, 0\n v_max3_i32 v21, v21, v26, 0\n v_max3_i32 v22, v22, v27, 0\n v_max3_i32 v23, v23, v24, 0") \
  X(9, "max3 v,const,const", "v_max3_i32 v20, v20, 3, 0\n v_max3_i32 v21, v21, 3, 0\n v_max3_i32 v22, v22, 3, 0\n v_max3_i32 v23, v23, 3, 0") \
  X(10, "max3 v,s,v", "v_max3_i32 v20, v20, s40, v25\n v_max3_i32 v21, v21, s40, v26\n v_max3_i32 v22, v22, s40, v27\n v_max3_i32 v23, v23, s40, v24") \
  X(11, "bfe_i32 v,const,8", "v_bfe_i32 v20, v20, 8, 8\n v_bfe_i32 v21, v21, 8, 8\n v_bfe_i32 v22, v22, 8, 8\n v_bfe_i32 v23, v23, 8, 8") \
  X(12, "bfe_i32 v,v,8", "v_bfe_i32 v20, v20, v25, 8\n v_bfe_i32 v21, v21, v26, 8\n v_bfe_i32 v22, v22, v27, 8\n v_bfe_i32 v23, v23, v24, 8") \
  X(13, "cndmask v,v,vcc", "v_cndmask_b32 v20, v20, v25, vcc\n v_cndmask_b32 v21, v21, v26, vcc\n v_cndmask_b32 v22, v22, v27, vcc\n v_cndmask_b32 v23, v23, v24, vcc") \
  X(14, "cndmask const,v,vcc", "v_cndmask_b32 v20, 0, v20, vcc\n v_cndmask_b32 v21, 0, v21, vcc\n v_cndmask_b32 v22, 0, v22, vcc\n v_cndmask_b32 v23, 0, v23, vcc") \
  X(15, "cmp v,v -> vcc", "v_cmp_eq_u32 vcc, v20, v25\n v_cmp_eq_u32 vcc, v21, v26\n v_cmp_eq_u32 vcc, v22, v27\n v_cmp_eq_u32 vcc, v23, v24") \
  X(16, "cmp const,v -> vcc", "v_cmp_eq_u32 vcc, 3, v20\n v_cmp_eq_u32 vcc, 3, v21\n v_cmp_eq_u32 vcc, 3, v22\n v_cmp_eq_u32 vcc, 3, v23") \
  X(17, "mov v,v", "v_mov_b32 v20, v24\n v_mov_b32 v21, v25\n v_mov_b32 v22, v26\n v_mov_b32 v23, v27") \
  X(18, "mov_dpp wave_shr:1", "v_mov_b32_dpp v20, v24 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v21, v25 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v22, v26 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v23, v27 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1") \
  X(19, "mov_dpp row_shr:1", "v_mov_b32_dpp v20, v24 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v21, v25 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v22, v26 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v23, v27 row_shr:1 row_mask:0xf bank_mask:0xf") \
  X(20, "add_sdwa byte sext", "v_add_u32_sdwa v20, v20, v25 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa v21, v21, v26 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa v22, v22, v27 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa v23, v23, v24 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1") \
  X(21, "pk_add_u16 v,v,v", "v_pk_add_u16 v20, v20, v25\n v_pk_add_u16 v21, v21, v26\n v_pk_add_u16 v22, v22, v27\n v_pk_add_u16 v23, v23, v24") \
  X(22, "pk_max_i16 v,v,v", "v_pk_max_i16 v20, v20, v25\n v_pk_max_i16 v21, v21, v26\n v_pk_max_i16 v22, v22, v27\n v_pk_max_i16 v23, v23, v24") \
  X(23, "max_i16 v,v,v (VOP2 16-bit)", "v_max_i16 v20, v20, v25\n v_max_i16 v21, v21, v26\n v_max_i16 v22, v22, v27\n v_max_i16 v23, v23, v24") \
  X(24, "add3 v,v,v", "v_add3_u32 v20, v20, v25, v30\n v_add3_u32 v21, v21, v26, v31\n v_add3_u32 v22, v22, v27, v28\n v_add3_u32 v23, v23, v24, v29") \
  X(25, "add3 v,s,const", "v_add3_u32 v20, v20, s40, 4\n v_add3_u32 v21, v21, s40, 4\n v_add3_u32 v22, v22, s40, 4\n v_add3_u32 v23, v23, s40, 4") \
  X(26, "mad_i32_i24 v,v,v", "v_mad_i32_i24 v20, v20, v25, v30\n v_mad_i32_i24 v21, v21, v26, v31\n v_mad_i32_i24 v22, v22, v27, v28\n v_mad_i32_i24 v23, v23, v24, v29") \
  X(27, "max_i32 v,v,v (other bank)", "v_max_i32 v20, v20, v25\n v_max_i32 v21, v21, v26\n v_max_i32 v22, v22, v27\n v_max_i32 v23, v23, v24") \
  X(28, "lshrrev const,v", "v_lshrrev_b32 v20, 3, v20\n v_lshrrev_b32 v21, 3, v21\n v_lshrrev_b32 v22, 3, v22\n v_lshrrev_b32 v23, 3, v23") \
  X(29, "add_f32 v,v,v", "v_add_f32 v20, v20, v25\n v_add_f32 v21, v21, v26\n v_add_f32 v22, v22, v27\n v_add_f32 v23, v23, v24") \
  X(30, "max3_f32 v,v,v", "v_max3_f32 v20, v20, v25, v30\n v_max3_f32 v21, v21, v26, v31\n v_max3_f32 v22, v22, v27, v28\n v_max3_f32 v23, v23, v24, v29") \
  X(31, "fma_f32 v,v,v", "v_fma_f32 v20, v20, v25, v30\n v_fma_f32 v21, v21, v26, v31\n v_fma_f32 v22, v22, v27, v28\n v_fma_f32 v23, v23, v24, v29") \
  X(32, "pk_fma_f32 (2 lanes-pairs)", "v_pk_fma_f32 v[20:21], v[20:21], v[24:25], v[28:29]\n v_pk_fma_f32 v[22:23], v[22:23], v[26:27], v[30:31]\n v_pk_fma_f32 v[20:21], v[20:21], v[24:25], v[28:29]\n v_pk_fma_f32 v[22:23], v[22:23], v[26:27], v[30:31]") \
  X(33, "max_u16 + add_u16 VOP2", "v_max_u16 v20, v20, v25\n v_add_u16 v21, v21, v26\n v_max_u16 v22, v22, v27\n v_add_u16 v23, v23, v24") \
  X(35, "and v,v,v", "v_and_b32 v20, v20, v25\n v_and_b32 v21, v21, v26\n v_and_b32 v22, v22, v27\n v_and_b32 v23, v23, v24") \
  X(36, "or v,v,v", "v_or_b32 v20, v20, v25\n v_or_b32 v21, v21, v26\n v_or_b32 v22, v22, v27\n v_or_b32 v23, v23, v24") \
  X(37, "xor v,v,v", "v_xor_b32 v20, v20, v25\n v_xor_b32 v21, v21, v26\n v_xor_b32 v22, v22, v27\n v_xor_b32 v23, v23, v24") \
  X(38, "sub v,v,v", "v_sub_u32 v20, v20, v25\n v_sub_u32 v21, v21, v26\n v_sub_u32 v22, v22, v27\n v_sub_u32 v23, v23, v24") \
  X(39, "subrev const,v", "v_subrev_u32 v20, 20, v20\n v_subrev_u32 v21, 20, v21\n v_subrev_u32 v22, 20, v22\n v_subrev_u32 v23, 20, v23") \
  X(40, "add v, literal 1000", "v_add_u32 v20, 1000, v20\n v_add_u32 v21, 1000, v21\n v_add_u32 v22, 1000, v22\n v_add_u32 v23, 1000, v23") \
  X(41, "and v, literal 0xffff", "v_and_b32 v20, 0xffff, v20\n v_and_b32 v21, 0xffff, v21\n v_and_b32 v22, 0xffff, v22\n v_and_b32 v23, 0xffff, v23") \
  X(42, "max_f32 v,v,v", "v_max_f32 v20, v20, v25\n v_max_f32 v21, v21, v26\n v_max_f32 v22, v22, v27\n v_max_f32 v23, v23, v24") \
  X(43, "min_f32 v,v,v", "v_min_f32 v20, v20, v25\n v_min_f32 v21, v21, v26\n v_min_f32 v22, v22, v27\n v_min_f32 v23, v23, v24") \
  X(44, "max_f32 v,const", "v_max_f32 v20, 0, v20\n v_max_f32 v21, 0, v21\n v_max_f32 v22, 0, v22\n v_max_f32 v23, 0, v23") \
  X(45, "max_f32 s,v", "v_max_f32 v20, s40, v20\n v_max_f32 v21, s40, v21\n v_max_f32 v22, s40, v22\n v_max_f32 v23, s40, v23") \
  X(46, "sub_f32 v,v", "v_sub_f32 v20, v20, v25\n v_sub_f32 v21, v21, v26\n v_sub_f32 v22, v22, v27\n v_sub_f32 v23, v23, v24") \
  X(47, "mul_f32 v,v", "v_mul_f32 v20, v20, v25\n v_mul_f32 v21, v21, v26\n v_mul_f32 v22, v22, v27\n v_mul_f32 v23, v23, v24") \
  X(48, "min_u32 v,v", "v_min_u32 v20, v20, v25\n v_min_u32 v21, v21, v26\n v_min_u32 v22, v22, v27\n v_min_u32 v23, v23, v24") \
  X(49, "max_u32 v,v", "v_max_u32 v20, v20, v25\n v_max_u32 v21, v21, v26\n v_max_u32 v22, v22, v27\n v_max_u32 v23, v23, v24") \
  X(50, "min_i16 v,v", "v_min_i16 v20, v20, v25\n v_min_i16 v21, v21, v26\n v_min_i16 v22, v22, v27\n v_min_i16 v23, v23, v24") \
  X(51, "max_i16 s,v", "v_max_i16 v20, s40, v20\n v_max_i16 v21, s40, v21\n v_max_i16 v22, s40, v22\n v_max_i16 v23, s40, v23") \
  X(52, "max_i16 const,v", "v_max_i16 v20, 0, v20\n v_max_i16 v21, 0, v21\n v_max_i16 v22, 0, v22\n v_max_i16 v23, 0, v23") \
  X(53, "sub_u16 v,v", "v_sub_u16 v20, v20, v25\n v_sub_u16 v21, v21, v26\n v_sub_u16 v22, v22, v27\n v_sub_u16 v23, v23, v24") \
  X(54, "lshlrev v,v", "v_lshlrev_b32 v20, v25, v20\n v_lshlrev_b32 v21, v26, v21\n v_lshlrev_b32 v22, v27, v22\n v_lshlrev_b32 v23, v24, v23") \
  X(55, "lshrrev v,v", "v_lshrrev_b32 v20, v25, v20\n v_lshrrev_b32 v21, v26, v21\n v_lshrrev_b32 v22, v27, v22\n v_lshrrev_b32 v23, v24, v23") \
  X(56, "ashrrev const,v", "v_ashrrev_i32 v20, 8, v20\n v_ashrrev_i32 v21, 8, v21\n v_ashrrev_i32 v22, 8, v22\n v_ashrrev_i32 v23, 8, v23") \
  X(57, "mul_u32_u24 v,v", "v_mul_u32_u24 v20, v20, v25\n v_mul_u32_u24 v21, v21, v26\n v_mul_u32_u24 v22, v22, v27\n v_mul_u32_u24 v23, v23, v24") \
  X(58, "mad_u32_u24 v,v,v", "v_mad_u32_u24 v20, v20, v25, v30\n v_mad_u32_u24 v21, v21, v26, v31\n v_mad_u32_u24 v22, v22, v27, v28\n v_mad_u32_u24 v23, v23, v24, v29") \
  X(59, "med3_i32 v,v,v", "v_med3_i32 v20, v20, v25, v30\n v_med3_i32 v21, v21, v26, v31\n v_med3_i32 v22, v22, v27, v28\n v_med3_i32 v23, v23, v24, v29") \
  X(60, "cmp_e64 v,v -> s[42:43]", "v_cmp_eq_u32_e64 s[42:43], v20, v25\n v_cmp_eq_u32_e64 s[42:43], v21, v26\n v_cmp_eq_u32_e64 s[42:43], v22, v27\n v_cmp_eq_u32_e64 s[42:43], v23, v24") \
  X(61, "cmp_gt_i16 v,v -> vcc", "v_cmp_gt_i16 vcc, v20, v25\n v_cmp_gt_i16 vcc, v21, v26\n v_cmp_gt_i16 vcc, v22, v27\n v_cmp_gt_i16 vcc, v23, v24") \
  X(62, "cmp_f32 v,v -> vcc", "v_cmp_gt_f32 vcc, v20, v25\n v_cmp_gt_f32 vcc, v21, v26\n v_cmp_gt_f32 vcc, v22, v27\n v_cmp_gt_f32 vcc, v23, v24") \
  X(63, "cndmask_e64 v,v,s[42:43]", "v_cndmask_b32_e64 v20, v20, v25, s[42:43]\n v_cndmask_b32_e64 v21, v21, v26, s[42:43]\n v_cndmask_b32_e64 v22, v22, v27, s[42:43]\n v_cndmask_b32_e64 v23, v23, v24, s[42:43]") \
  X(64, "lshl_add v,const,v", "v_lshl_add_u32 v20, v20, 8, v25\n v_lshl_add_u32 v21, v21, 8, v26\n v_lshl_add_u32 v22, v22, 8, v27\n v_lshl_add_u32 v23, v23, 8, v24") \
  X(65, "add_lshl v,v,const", "v_add_lshl_u32 v20, v20, v25, 8\n v_add_lshl_u32 v21, v21, v26, 8\n v_add_lshl_u32 v22, v22, v27, 8\n v_add_lshl_u32 v23, v23, v24, 8") \
  X(66, "lshl_or v,const,v", "v_lshl_or_b32 v20, v20, 8, v25\n v_lshl_or_b32 v21, v21, 8, v26\n v_lshl_or_b32 v22, v22, 8, v27\n v_lshl_or_b32 v23, v23, 8, v24") \
  X(67, "or3 v,v,v", "v_or3_b32 v20, v20, v25, v30\n v_or3_b32 v21, v21, v26, v31\n v_or3_b32 v22, v22, v27, v28\n v_or3_b32 v23, v23, v24, v29") \
  X(68, "xad v,v,v", "v_xad_u32 v20, v20, v25, v30\n v_xad_u32 v21, v21, v26, v31\n v_xad_u32 v22, v22, v27, v28\n v_xad_u32 v23, v23, v24, v29") \
  X(69, "perm v,v,v", "v_perm_b32 v20, v20, v25, v30\n v_perm_b32 v21, v21, v26, v31\n v_perm_b32 v22, v22, v27, v28\n v_perm_b32 v23, v23, v24, v29") \
  X(70, "cvt_f32_i32", "v_cvt_f32_i32 v20, v20\n v_cvt_f32_i32 v21, v21\n v_cvt_f32_i32 v22, v22\n v_cvt_f32_i32 v23, v23") \
  X(71, "readlane (to s40)", "v_readlane_b32 s40, v20, 5\n v_readlane_b32 s40, v21, 5\n v_readlane_b32 s40, v22, 5\n v_readlane_b32 s40, v23, 5") \
  X(72, "readfirstlane", "v_readfirstlane_b32 s41, v20\n v_readfirstlane_b32 s41, v21\n v_readfirstlane_b32 s41, v22\n v_readfirstlane_b32 s41, v23") \
  X(73, "add_co_u32 v,v,v", "v_add_co_u32 v20, vcc, v20, v25\n v_add_co_u32 v21, vcc, v21, v26\n v_add_co_u32 v22, vcc, v22, v27\n v_add_co_u32 v23, vcc, v23, v24") \
  X(74, "sad_u32 v,v,v", "v_sad_u32 v20, v20, v25, v30\n v_sad_u32 v21, v21, v26, v31\n v_sad_u32 v22, v22, v27, v28\n v_sad_u32 v23, v23, v24, v29") \
  X(75, "bfi v,v,v", "v_bfi_b32 v20, v20, v25, v30\n v_bfi_b32 v21, v21, v26, v31\n v_bfi_b32 v22, v22, v27, v28\n v_bfi_b32 v23, v23, v24, v29") \
  X(76, "mbcnt_lo", "v_mbcnt_lo_u32_b32 v20, v20, v25\n v_mbcnt_lo_u32_b32 v21, v21, v26\n v_mbcnt_lo_u32_b32 v22, v22, v27\n v_mbcnt_lo_u32_b32 v23, v23, v24") \
  X(77, "bcnt", "v_bcnt_u32_b32 v20, v20, v25\n v_bcnt_u32_b32 v21, v21, v26\n v_bcnt_u32_b32 v22, v22, v27\n v_bcnt_u32_b32 v23, v23, v24") \
  X(78, "ffbh", "v_ffbh_u32 v20, v20\n v_ffbh_u32 v21, v21\n v_ffbh_u32 v22, v22\n v_ffbh_u32 v23, v23") \
  X(79, "dot4 i8", "v_dot4_i32_i8 v20, v20, v25, v30\n v_dot4_i32_i8 v21, v21, v26, v31\n v_dot4_i32_i8 v22, v22, v27, v28\n v_dot4_i32_i8 v23, v23, v24, v29") \
  X(80, "pk_max_i16 + pk_add", "v_pk_max_i16 v20, v20, v25\n v_pk_max_i16 v21, v21, v26\n v_pk_max_i16 v22, v22, v27\n v_pk_max_i16 v23, v23, v24") \
  X(81, "pk_mul_f32?", "v_pk_add_f32 v[20:21], v[20:21], v[24:25]\n v_pk_add_f32 v[20:21], v[20:21], v[24:25]\n v_pk_add_f32 v[20:21], v[20:21], v[24:25]\n v_pk_add_f32 v[20:21], v[20:21], v[24:25]") \
  X(34, "add v,v,v dependent chain", "v_add_u32 v20, v20, v21\n v_add_u32 v21, v21, v20\n v_add_u32 v20, v20, v21\n v_add_u32 v21, v21, v20")

template <int MODE>
__global__ void k(int iters, int* out) {
  for (int i = 0; i < iters; ++i) {
#define X(m, name, code) if (MODE == m) { REP64(asm volatile(code : : : CLOB);) }
    MODES(X)
#undef X
  }
  if (iters == 0x7fffffff) out[0] = 1;
}
typedef void (*kern_t)(int, int*);
int main(int argc, char** argv) {
  int* out; (void)hipMalloc(&out, 64);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const double ghz = p.clockRate * 1e-6;
  printf("CUs %d clock %.2f GHz; cycles per wave64 instruction per SIMD at 2 / 8 waves per SIMD\n", p.multiProcessorCount, ghz); fflush(stdout);
  struct M { int m; const char* name; kern_t f; };
#define X(m, name, code) {m, name, k<m>},
  M modes[] = {MODES(X)};
#undef X
  for (auto& md : modes) {
    if (argc > 1 && md.m < atoi(argv[1]) && md.m != 34) continue;
    double cyc[2];
    int wi = 0;
    for (int wps : {2, 8}) {
      const int iters = 1000, blocks = p.multiProcessorCount * wps;       // 256-thread blocks: 4 waves = one per SIMD
      hipLaunchKernelGGL(md.f, dim3(blocks), dim3(256), 0, 0, iters, out); (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0); hipLaunchKernelGGL(md.f, dim3(blocks), dim3(256), 0, 0, iters, out); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      cyc[wi++] = (ms * 1e-3 * ghz * 1e9) / (256.0 * iters * wps);
    }
    printf("%2d %-34s %.2f  %.2f\n", md.m, md.name, cyc[0], cyc[1]); fflush(stdout);
  }
  return 0;
}
