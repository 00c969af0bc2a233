// Probe: does v_mfma_f32_32x32x16_f16 keep f16 subnormal inputs on gfx950?
// Build: hipcc --offload-arch=gfx950 -O2 tools/f16_denorm_probe.cpp -o build_variants/f16_denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void probe(float a_val, float b_val, float* out) {
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)a_val; b[j] = (_Float16)b_val; }
  f16v acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; }
}
int main() {
  float* d; hipMalloc(&d, 8);
  float vals[] = {1.0f, 0x1p-14f, 0x1p-15f, 0x1p-20f, 0x1p-24f};
  for (float v : vals) {
    hipLaunchKernelGGL(probe, 1, 64, 0, 0, v, 1.0f, d);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("a=%g (as f16 %g) b=1: sum over k=16 -> %g (expected %g)\n", v, h[1], h[0], 16.0 * v);
    hipLaunchKernelGGL(probe, 1, 64, 0, 0, 1.0f, v, d);
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("a=1 b=%g: -> %g\n", v, h[0]);
  }
  return 0;
}
