// MFMA issue-rate microbenchmark (development aid): cycles per v_mfma_f32_32x32x16_bf16 for
// NACC independent accumulators, WAVES wavefronts per workgroup, one workgroup per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void k(unsigned long long *out, float *sink, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; n++) for (int i = 0; i < 16; i++) acc[n][i] = 0.f;
  __syncthreads();
  const unsigned long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < 4; rep++)
#pragma unroll
      for (int n = 0; n < NACC; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
  }
  const unsigned long long t1 = clock64(), w1 = wall_clock64();
  float s = 0; for (int n = 0; n < NACC; n++) s += acc[n][0];
  if (s == 123.f) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
}
template <int NACC> void run(int waves) {
  unsigned long long *d, h[2]; float *s;
  hipMalloc(&d, 16); hipMalloc(&s, 4);
  const int iters = 2000;
  hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(64 * waves), 0, 0, d, s, iters);
  hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(64 * waves), 0, 0, d, s, iters);
  hipDeviceSynchronize();
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  const double n = (double)iters * 4 * NACC;
  printf("NACC=%d waves/WG=%d: %.1f clock64 ticks per MFMA per wave, %.1f ns per MFMA per wave, clock64 rate %.2f GHz; chip rate if all CUs: %.0f TF\n",
         NACC, waves, h[0] / n, h[1] * 10.0 / n, h[0] / (h[1] * 10.0), 256.0 * waves * 32768.0 / (h[1] * 10.0 / n) / 1e3);
}
int main() { run<1>(4); run<2>(4); run<4>(4); run<8>(4); run<2>(8); run<4>(8); return 0; }
