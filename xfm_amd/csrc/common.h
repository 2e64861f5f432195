// Shared device/host helpers for the XFM hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define XFM_INTERNAL_BF16
#include "../../include/xfm_hip.h"

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
#define GLB_PTR(T, p) ((const __attribute__((address_space(1))) T*)(p))

// host side: thread-local error message (C-ABI: xfm_last_error)
void xfm_set_error(const char* fmt, ...);
int xfm_check_launch(const char* what);

#define XFM_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      xfm_set_error(__VA_ARGS__);         \
      return XFM_E_ARG;                   \
    }                                     \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

#ifdef __HIPCC__
// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }

// exact erf GELU (transformers ACT2FN["gelu"], torch.nn.GELU default) and its derivative
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.39894228040143267f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// stateless 32-bit mixer for dropout masks: keep(element) is a pure function of (seed, element id)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t rng_u32(uint32_t seed_lo, uint32_t seed_hi, uint32_t idx_lo, uint32_t idx_hi) {
  return mix32(mix32(idx_lo ^ seed_lo) + 0x9E3779B9U * (idx_hi ^ seed_hi) + 0x85EBCA6BU);
}
// keep with probability (1-p): thresh = p * 2^32
__device__ __forceinline__ bool rng_keep(uint32_t r, uint32_t thresh) { return r >= thresh; }

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

// Bijective XCD-aware block remap: blocks b and b+8 share an XCD (round-robin dispatch), so give each
// XCD a contiguous chunk of the logical tile space (speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
#endif
