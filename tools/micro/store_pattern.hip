// HBM write bandwidth of the cost-volume forward's store pattern without any of its arithmetic: is 4.2 TB/s the
// pattern's ceiling?  Volume [D = 192 planes][H = 48 rows][W x G = 5120 floats]; a workgroup owns (row, chunk of planes).
//   fill        : plain grid-stride float4 fill of the same 188.7 MB (the reference)
//   item_outer  : gwc_warp_fwd4_kernel's order -- a thread finishes all planes of one float4 item, then its next item
//   plane_outer : per plane the workgroup writes its whole 20 KB row, then moves to the next plane
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/store_pattern.hip -o build/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int D = 192, H = 48, ROW4 = 160 * 32 / 4;            // float4 per plane row
constexpr long PLANE4 = (long)H * ROW4;

template <bool NT> __device__ __forceinline__ void st(v4f* p, v4f v) {
  if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
template <bool NT> __global__ void __launch_bounds__(256) fill_k(v4f* out, long n4) {
  const v4f v = {1.f, 2.f, 3.f, 4.f};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) st<NT>(out + i, v);
}
template <bool NT> __global__ void __launch_bounds__(512) item_outer_k(v4f* out, int planes_per_chunk) {
  const int row = blockIdx.x, k0 = blockIdx.y * planes_per_chunk, k1 = min(D, k0 + planes_per_chunk);
  const v4f v = {1.f, 2.f, 3.f, 4.f};
  for (int item = threadIdx.x; item < ROW4; item += blockDim.x) {
    v4f* dst = out + (long)k0 * PLANE4 + (long)row * ROW4 + item;
    for (int k = k0; k < k1; ++k) { st<NT>(dst, v); dst += PLANE4; }
  }
}
template <bool NT> __global__ void __launch_bounds__(512) plane_outer_k(v4f* out, int planes_per_chunk) {
  const int row = blockIdx.x, k0 = blockIdx.y * planes_per_chunk, k1 = min(D, k0 + planes_per_chunk);
  const v4f v = {1.f, 2.f, 3.f, 4.f};
  for (int k = k0; k < k1; ++k) {
    v4f* dst = out + (long)k * PLANE4 + (long)row * ROW4;
    for (int item = threadIdx.x; item < ROW4; item += blockDim.x) st<NT>(dst + item, v);
  }
}
template <class F> void timeit(const char* name, F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  float best = 1e9f;
  for (int i = 0; i < 10; ++i) {
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
  }
  const double bytes = (double)D * PLANE4 * 16.0;
  printf("%-44s %7.1f us  %5.2f TB/s\n", name, best * 1e3, bytes / best / 1e9);
}
int main() {
  v4f* out; const long n4 = (long)D * PLANE4; hipMalloc(&out, n4 * 16);
  timeit("fill", [&] { hipLaunchKernelGGL(fill_k<false>, dim3(8192), dim3(256), 0, 0, out, n4); });
  timeit("fill, nontemporal", [&] { hipLaunchKernelGGL(fill_k<true>, dim3(8192), dim3(256), 0, 0, out, n4); });
  for (int ppc : {12, 6, 24, 48}) {
    for (int threads : {256, 512}) {
      char nm[96];
      const dim3 grid(H, (D + ppc - 1) / ppc);
      snprintf(nm, sizeof nm, "item_outer  ppc=%2d threads=%d", ppc, threads);
      timeit(nm, [&] { hipLaunchKernelGGL(item_outer_k<false>, grid, dim3(threads), 0, 0, out, ppc); });
      snprintf(nm, sizeof nm, "item_outer  ppc=%2d threads=%d nontemporal", ppc, threads);
      timeit(nm, [&] { hipLaunchKernelGGL(item_outer_k<true>, grid, dim3(threads), 0, 0, out, ppc); });
      snprintf(nm, sizeof nm, "plane_outer ppc=%2d threads=%d", ppc, threads);
      timeit(nm, [&] { hipLaunchKernelGGL(plane_outer_k<false>, grid, dim3(threads), 0, 0, out, ppc); });
      snprintf(nm, sizeof nm, "plane_outer ppc=%2d threads=%d nontemporal", ppc, threads);
      timeit(nm, [&] { hipLaunchKernelGGL(plane_outer_k<true>, grid, dim3(threads), 0, 0, out, ppc); });
    }
  }
  hipFree(out);
  return 0;
}
