#include <hip/hip_runtime.h>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 b4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short l[64 * 16];
  for (int i = threadIdx.x; i < 1024; i += 64) l[i] = in[i];
  __syncthreads();
  // lane p of a 16-lane group supplies the address of 4 contiguous elements
  const int lane = threadIdx.x;
  const int addr_el = ((lane & 15) >> 2) * 32 + (lane & 3) * 4 + (lane >> 4) * 128;   // rows of 32 elements (64 B)
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(l + addr_el));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short h[1024], *din, *dout, o[256];
  for (int i = 0; i < 1024; ++i) h[i] = i;
  hipMalloc(&din, 2048); hipMalloc(&dout, 512);
  hipMemcpy(din, h, 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
  hipMemcpy(o, dout, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, o[4*l], o[4*l+1], o[4*l+2], o[4*l+3]);
  return 0;
}
