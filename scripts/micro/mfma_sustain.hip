// Sustained bf16 MFMA throughput of the whole chip (development aid): 8 independent accumulators per wavefront,
// 4 or 8 wavefronts per CU-resident workgroup, back-to-back launches for ~0.2 s, timed with HIP events.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(512) k(float *sink, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (short)(0x3f80 + threadIdx.x + i); b[i] = (short)(0x3f80 + threadIdx.x * 3 + i); }
  f32x16 acc[8];
  for (int n = 0; n < 8; n++) for (int i = 0; i < 16; i++) acc[n][i] = 0.f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int n = 0; n < 8; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
  }
  float s = 0; for (int n = 0; n < 8; n++) s += acc[n][0];
  if (s == 123.f) sink[0] = s;
}
int main() {
  float *s; hipMalloc(&s, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves = 4; waves <= 8; waves += 4) {
    for (int wg = 1; wg <= 2; wg++) {
      const int iters = 4000, launches = 60;
      hipLaunchKernelGGL(k, dim3(256 * wg), dim3(64 * waves), 0, 0, s, iters);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int l = 0; l < launches; l++) hipLaunchKernelGGL(k, dim3(256 * wg), dim3(64 * waves), 0, 0, s, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fl = (double)launches * 256 * wg * waves * iters * 8 * 32768.0;
      printf("waves/WG=%d WGs/CU=%d: %.1f ms, %.0f TFLOP/s bf16 sustained\n", waves, wg, ms, fl / ms / 1e9);
    }
  }
  return 0;
}
