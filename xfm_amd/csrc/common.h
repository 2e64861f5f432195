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
int xfm_cu_count();

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

// erf-GELU (transformers ACT2FN["gelu"], torch.nn.GELU default) and its derivative.  The GEMM epilogues evaluate them on
// 64 elements per lane, so libm's erff (~45 VALU instructions per element) costs as much as the whole K = 768 MFMA loop.
// erf is evaluated with Abramowitz-Stegun 7.1.25, |delta erf| <= 2.5e-5 -- two orders below the bf16 resolution (2^-9)
// of the stored activation -- and shares its exp(-x^2/2) with the Gaussian density of the derivative.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.33267f, ax, 1.0f));              // 1 / (1 + 0.47047 |x| / sqrt(2))
  const float e = __builtin_amdgcn_exp2f(-0.72134752f * x * x);                 // exp(-x^2 / 2)
  const float poly = t * fmaf(t, fmaf(t, 0.7478556f, -0.0958798f), 0.3480242f);
  const float erf_abs = fmaf(-poly, e, 1.0f);                                   // erf(|x| / sqrt(2))
  cdf = fmaf(copysignf(0.5f, x), erf_abs, 0.5f);
  pdf = 0.39894228040143267f * e;
}
__device__ __forceinline__ float gelu_f(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return x * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return fmaf(x, pdf, cdf);
}

// stateless 32-bit mixer for dropout masks: keep(element) is a pure function of (seed, element id)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// Counter-based dropout stream over (row, column): a per-row key (two lowbias32 rounds over the row id and the 64-bit seed,
// computed once per row) and ONE round per element, mix32(column + key) -- 2 integer multiplies, which are quarter-rate on
// CDNA.  (A single 64-bit linear index made the key depend on the index's high word, which the compiler must recompute per
// element: 5 multiplies per attention score, two thirds of the VALU time of the text-side attention kernels.)
__device__ __forceinline__ uint32_t rng_row_key(uint32_t seed_lo, uint32_t seed_hi, uint32_t row) {
  return mix32(mix32(row ^ seed_lo) + 0x9E3779B9U * seed_hi + 0x85EBCA6BU);
}
__device__ __forceinline__ uint32_t rng_u32(uint32_t row_key, uint32_t col) { return mix32(col + row_key); }
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
// Grouped tile order (after xcd_remap): consecutive logical ids walk GM row-panels down M before stepping to the next
// column tile, so the ~32-64 tiles an XCD runs at once share a few A panels AND a few B tiles -- their live working set fits
// the XCD's private 4 MiB L2 instead of streaming every B tile from the Infinity Cache.
__device__ __forceinline__ void grouped_tile(int wg, int tiles_m, int tiles_n, int GM, int& tm, int& tn) {
  const int gsz = GM * tiles_n;
  const int gid = wg / gsz, in = wg - gid * gsz;
  const int first = gid * GM;
  const int gm = (tiles_m - first) < GM ? (tiles_m - first) : GM;
  tm = first + in % gm;
  tn = in / gm;
}
#endif
