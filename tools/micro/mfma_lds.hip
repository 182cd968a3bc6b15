// Sustained MFMA rate of the wgrad_lds inner loop in isolation: 9 accumulators, operands from LDS
// (4 ds_read per 9 MFMAs), no global traffic, no barriers.  Variants: LDS reads on/off, 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mf(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
template <int CQ, bool LDS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k(float* out, int trips) {
  extern __shared__ float wl[];
  for (int i = threadIdx.x; i < 8192; i += 256) wl[i] = 1e-3f * i;
  __syncthreads();
  const int lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
  f32x16 acc[3][3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int rep = 0; rep < trips; ++rep) {
    const float* p0 = wl + li + lk * CQ;
    const float* p1 = p0 + 18 * CQ;
    const float* p2 = p1 + 18 * CQ;
    const float* pg = wl + 4096 + li + lk * CQ;
    float c0 = p0[0], c1 = p1[0], c2 = p2[0], pA = pg[0];
    float a01 = p0[CQ], a02 = p0[2 * CQ], a11 = p1[CQ], a12 = p1[2 * CQ], a21 = p2[CQ], a22 = p2[2 * CQ];
    for (int ks = 0; ks < 8; ks += 2) {
      acc[0][0] = mf(c0, pA, acc[0][0]);
      __builtin_amdgcn_sched_barrier(0);
      float pB, b01, b02, b11, b12, b21, b22;
      if (LDS) { pB = pg[2 * CQ]; b01 = p0[3 * CQ]; b02 = p0[4 * CQ]; b11 = p1[3 * CQ]; b12 = p1[4 * CQ]; b21 = p2[3 * CQ]; b22 = p2[4 * CQ]; }
      else { pB = pA; b01 = a01; b02 = a02; b11 = a11; b12 = a12; b21 = a21; b22 = a22; }
      __builtin_amdgcn_sched_barrier(0);
      acc[1][0] = mf(c1, pA, acc[1][0]); acc[2][0] = mf(c2, pA, acc[2][0]);
      acc[0][1] = mf(a01, pA, acc[0][1]); acc[1][1] = mf(a11, pA, acc[1][1]); acc[2][1] = mf(a21, pA, acc[2][1]);
      acc[0][2] = mf(a02, pA, acc[0][2]); acc[1][2] = mf(a12, pA, acc[1][2]); acc[2][2] = mf(a22, pA, acc[2][2]);
      __builtin_amdgcn_sched_barrier(0);
      c0 = a02; c1 = a12; c2 = a22;
      acc[0][0] = mf(c0, pB, acc[0][0]);
      __builtin_amdgcn_sched_barrier(0);
      if (LDS) { pA = pg[4 * CQ]; a01 = p0[5 * CQ]; a02 = p0[6 * CQ]; a11 = p1[5 * CQ]; a12 = p1[6 * CQ]; a21 = p2[5 * CQ]; a22 = p2[6 * CQ]; }
      __builtin_amdgcn_sched_barrier(0);
      acc[1][0] = mf(c1, pB, acc[1][0]); acc[2][0] = mf(c2, pB, acc[2][0]);
      acc[0][1] = mf(b01, pB, acc[0][1]); acc[1][1] = mf(b11, pB, acc[1][1]); acc[2][1] = mf(b21, pB, acc[2][1]);
      acc[0][2] = mf(b02, pB, acc[0][2]); acc[1][2] = mf(b12, pB, acc[1][2]); acc[2][2] = mf(b22, pB, acc[2][2]);
      __builtin_amdgcn_sched_barrier(0);
      c0 = b02; c1 = b12; c2 = b22;
      p0 += 4 * CQ; p1 += 4 * CQ; p2 += 4 * CQ; pg += 4 * CQ;
    }
  }
  float s = 0.f;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CQ, bool LDS> void run(int bpc) {
  float* out; hipMalloc(&out, 1 << 26);
  const int trips = 20000, blocks = 256 * bpc;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CQ, LDS>), dim3(blocks), dim3(256), 40960, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<CQ, LDS>), dim3(blocks), dim3(256), 40960, 0, out, trips);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * trips * 8 * 9 * 4096.0;
  printf("CQ=%d lds=%d blocks/CU=%d : %.1f TF/s\n", CQ, (int)LDS, bpc, flops / ms / 1e9);
  hipFree(out);
}
int main() { run<64, false>(1); run<64, false>(2); run<64, true>(1); run<64, true>(2); run<32, true>(2); return 0; }
