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

// Environment switches of the library (SSBEV_*): each name is looked up in the process environment ONCE (first use, under a
// mutex) and answered from a table afterwards -- no getenv per launch (the launch helpers run hundreds of times per step, from
// the host thread and from autograd's) and no race with setenv on the host-language side.  A process that changes a switch after
// the library has read it calls ssbev_env_refresh() (include/ssbev.h).  Returns nullptr when the variable is unset.  capi.hip.
const char* ssbev_env(const char* name);
// Tuning hooks (forced tile shapes, grid sizes, kernel-family A/B switches, phase-clock dumps: ~50 names) exist only in a tuning
// build (`python -m stereoscene_amd.build --tuning` = -DSSBEV_TUNING); in the product build the lookup is a null constant and
// the branches behind it fold away.  What stays switchable at run time is what the parity tests flip (SSBEV_IGEMM, SSBEV_IGEMM16,
// SSBEV_POOL_MAX_DIGIT_BITS); everything a USER chooses (precision, ablation, DP exchange, streams) is host-side.
#ifdef SSBEV_TUNING
static inline const char* ssbev_tune(const char* name) { return ssbev_env(name); }
#else
static inline const char* ssbev_tune(const char*) { return nullptr; }
#endif

// 64-lane butterfly sum (all lanes receive the total).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- activation storage types (BASELINE configs[3]: bf16 activations between layers, fp32 arithmetic inside the kernels) ----
// bf16 tensors cross the C ABI as raw 16-bit patterns; loads widen exactly, stores round to nearest even (v_cvt_pk_bf16_f32).
typedef unsigned short bf16_t;
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ bf16_t f2bf(float v) { return __builtin_bit_cast(bf16_t, (__bf16)v); }
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) { return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16); }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return bf2f(*p); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = f2bf(v); }
// four consecutive elements (16-byte / 8-byte aligned)
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(float* p, const float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, const float4 v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
}

// One 1 KiB LDS-DMA (global_load_lds_dwordx4: lane l copies 16 bytes from gsrc to lds_dst + 16 l) issued behind the compiler's
// back.  hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the first ds_read that follows a __builtin_amdgcn_global_load_lds
// it cannot prove disjoint -- in the LDS-ring kernels that is the first operand read of every walk, so the rows "in flight
// during the walk" were waited for before it (round 4: the .s of conv_taph_kernel).  An asm statement is outside its
// bookkeeping: the kernel waits (`s_waitcnt vmcnt(0)` + barrier) exactly where the ring protocol needs it.  lds_dst must be
// wave-uniform; M0 is saved and restored (the compiler keeps its own values there).
__device__ __forceinline__ void glds16(const float* gsrc, const float* lds_dst) {
  unsigned keep;
  const unsigned ldsaddr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(ldsaddr)
               : "memory");
}

// `s_waitcnt vmcnt(0)` as an INSTRUCTION hipcc sees (the builtin; simm16 = vmcnt 0, expcnt 7, lgkmcnt 15 on gfx9): it retires
// the compiler's own pending loads in its bookkeeping -- behind an asm wait it keeps believing they are in flight and waits
// for them (and for every glds16 issued since) at their next register hazard -- and it is a hard wait, so it also covers
// the glds16 copies the compiler knows nothing about.
__device__ __forceinline__ void wait_vm0() {
  __builtin_amdgcn_s_waitcnt(0x0F70);
  asm volatile("" ::: "memory");
}

// Workgroup barrier that retires this wave's LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier).  __syncthreads() is a fence:
// it also waits vmcnt(0) for the wave's pending global STORES -- the write acknowledgement of an epilogue store issued just
// before it (1.8 k clocks per block in conv_tapdh_kernel).  Use where the barrier orders LDS accesses only.
__device__ __forceinline__ void barrier_lds() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0); vmcnt / expcnt untouched
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
