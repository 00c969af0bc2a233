// Shared helpers for the gfx950 kernels (wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/st_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

void st_set_error(const char* fmt, ...);
// internal, not part of the C-ABI (csrc/st_simloss.hip; used by csrc/st_aetrain.hip)
bool sl_small_views_ok(int B);
int sl_small_views(const float* ori, const float* like0, int n_views, int B, float tau, float* loss_rows, float* lse, const float* g,
                   float* d_ori_parts, float* d_like0, hipStream_t st);

#define ST_CHECK_ARG(cond, ...)   \
  do {                            \
    if (!(cond)) {                \
      st_set_error(__VA_ARGS__);  \
      return ST_EINVAL;           \
    }                             \
  } while (0)

#define ST_HIP(call)                                                                    \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      st_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return ST_EHIP;                                                                   \
    }                                                                                   \
  } while (0)

#define ST_LAUNCH_CHECK() ST_HIP(hipGetLastError())

// Kernels that need more than 64 KiB of dynamic LDS must opt in once PER DEVICE (hipFuncSetAttribute acts on the
// current device).  `done` is a per-kernel static array indexed by device ordinal.
#define ST_MAX_DEVICES 64
#define ST_ENSURE_LDS(fn, bytes)                                                                        \
  do {                                                                                                  \
    static bool done_[ST_MAX_DEVICES] = {};                                                             \
    int dev_ = 0;                                                                                       \
    ST_HIP(hipGetDevice(&dev_));                                                                        \
    if (dev_ < 0 || dev_ >= ST_MAX_DEVICES || !done_[dev_]) {                                           \
      ST_HIP(hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes))); \
      if (dev_ >= 0 && dev_ < ST_MAX_DEVICES) done_[dev_] = true;                                       \
    }                                                                                                   \
  } while (0)

// Sum over the 64 lanes, result in every lane.  DPP row shifts + row broadcasts (6 VALU ops + one readlane)
// instead of six ds_bpermute round trips through the LDS crossbar (~100 cycles each): the LayerNorm prologues
// were spending most of their time there.
__device__ __forceinline__ float wave_sum(float v) {
#define ST_DPP_ADD(ctrl, row_mask)                                                                            \
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, row_mask, 0xf, false))
  ST_DPP_ADD(0x111, 0xf);  // row_shr:1   (lanes shifted in from outside the 16-lane row read as 0)
  ST_DPP_ADD(0x112, 0xf);  // row_shr:2
  ST_DPP_ADD(0x114, 0xf);  // row_shr:4
  ST_DPP_ADD(0x118, 0xf);  // row_shr:8   -> lane 15 of each row holds the row sum
  ST_DPP_ADD(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
  ST_DPP_ADD(0x143, 0xc);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
#undef ST_DPP_ADD
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// Sum over the 16 lanes of a DPP row (lanes 16k..16k+15), result in all 16: row rotations, no LDS crossbar.
// The attention kernels put one head on one row (lane = 4 of the head's 64 dims).
__device__ __forceinline__ float row16_sum(float v) {
#define ST_ROW_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, false))
  ST_ROW_ADD(0x121);  // row_ror:1
  ST_ROW_ADD(0x122);  // row_ror:2
  ST_ROW_ADD(0x124);  // row_ror:4
  ST_ROW_ADD(0x128);  // row_ror:8
#undef ST_ROW_ADD
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// C/D register -> row of a 32x32 MFMA tile (cdna guide §3): row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
__device__ __forceinline__ int mfma32_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// D += A(32x2) * B(2x32), exact f32 (v_mfma_f32_32x32x2_f32)
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- f16x3 split products (gemm_format ST_GEMM_F16X3) -------------------------------------------
// x = hi + lo + O(2^-22 |x|) with hi = f16(x), lo = f16(x - hi)  (f16 subnormals are kept by the
// conversion and by the MFMA on gfx950, so the absolute residual is <= 2^-25 for small x);
// x.w ~ hi.whi + hi.wlo + lo.whi on v_mfma_f32_32x32x16_f16 (products exact, f32 accumulate).
__device__ __forceinline__ void split_f16(f32x4 v, h4& hi, h4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const _Float16 h = (_Float16)v[j];
    hi[j] = h;
    lo[j] = (_Float16)(v[j] - (float)h);
  }
}
// D += A(32x16) * B(16x32): lane (r = lane&31, h = lane>>5) holds A[r][8h..8h+7], B[8h..8h+7][r]
__device__ __forceinline__ f32x16 mfma16h(h8 a, h8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ h8 as_h8(f32x4 v) { return __builtin_bit_cast(h8, v); }
// D(16 x 16) += A(16 x 32) . B(32 x 16) on f16: lane l holds A[row l & 15][k = 8 (l >> 4) .. +7], B[k = same][col l & 15];
// D: col = l & 15, rows 4 (l >> 4) .. +3 in the four registers
__device__ __forceinline__ f32x4 mfma16x32h(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
