// Sustained v_mfma_f32_32x32x2_f32 rate with everything in registers: the practical ceiling for the conv kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x = a + threadIdx.x * 1e-9f, y = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int wpsimd) {
  float* out; hipMalloc(&out, 1 << 26);
  const int iters = 20000, blocks = 256 * wpsimd;     // 4 waves per block = one per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.f, 1.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * NACC * 4096.0;
  printf("acc=%d waves/SIMD=%d : %.1f TF/s  (%.2f ms)\n", NACC, wpsimd, flops / ms / 1e9, ms);
  hipFree(out);
}
int main() { run<4>(1); run<8>(1); run<9>(1); run<9>(2); run<4>(2); run<8>(2); return 0; }
