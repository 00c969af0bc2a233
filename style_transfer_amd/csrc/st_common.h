// Shared helpers for the gfx950 kernels (wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/st_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void st_set_error(const char* fmt, ...);

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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
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
