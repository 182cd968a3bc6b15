// Shared helpers for the ssbev HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssbev.h"

#define SSBEV_WAVE 64

static inline int ssbev_launch_status() {
  return hipGetLastError() == hipSuccess ? SSBEV_OK : SSBEV_ELAUNCH;
}

static inline hipStream_t as_stream(ssbev_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// 64-lane butterfly sum (all lanes receive the total).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
