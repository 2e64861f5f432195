// Fused LayerNorm family (gfx950).  One 64-lane wavefront owns one row; lane l holds elements
// (i*64 + l)*4 .. +3 of every 256-element chunk i, so loads are 16 B (fp32) / 8 B (bf16) per lane, row statistics
// are two wave reductions, and per-column gradient sums stay in registers while a wave walks its rows.
//
//   mode PLAIN : y = LN(x)                                   beit2.py:191-206 norm1 / fc_norm :460, xroberta.py:1329,
//                                                            xfm.py:118 (itm_head LayerNorm)
//   mode POST  : z = dropout(h) + res ; y = LN(z)            xroberta.py:300-304, :381-385 (post-LN residual blocks)
//   mode LS    : x' = x + s_b * gamma_ls * h ; y = LN(x')    beit2.py:203-204 (layer-scale + drop-path residual)
//                                                            fused with the LayerNorm that consumes x' next
#include "common.h"
#include <stdlib.h>

enum { LN_PLAIN = 0, LN_POST = 1, LN_LS = 2 };

typedef xfm_ln_fwd_args LnFwd;

template <int NCH, int MODE>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnFwd p) {
  constexpr int D = NCH * 256;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  float wv[NCH][4], bv[NCH][4], lsg[NCH][4];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int e = (i * 64 + lane) * 4;
    const f32x4 a = *reinterpret_cast<const f32x4*>(p.w + e), c = *reinterpret_cast<const f32x4*>(p.b + e);
#pragma unroll
    for (int j = 0; j < 4; ++j) { wv[i][j] = a[j]; bv[i][j] = c[j]; lsg[i][j] = 0.f; }
    if (MODE == LN_LS) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(p.ls_gamma + e);
#pragma unroll
      for (int j = 0; j < 4; ++j) lsg[i][j] = g[j];
    }
  }
  for (int row = wave; row < p.rows; row += nwaves) {
    const long base = (long)row * D;
    const uint32_t rkey = rng_row_key(p.seed_lo, p.seed_hi, (uint32_t)row);  // dropout stream of this row
    float v[NCH][4];
    float s = 0.f;
    float rs = 1.f;
    if (MODE == LN_LS && p.row_scale != nullptr) rs = p.row_scale[row / p.rows_per_sample];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e = (i * 64 + lane) * 4;
      if (MODE == LN_PLAIN) {
        if (p.x32 != nullptr) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(p.x32 + base + e);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[i][j] = a[j];
        } else {
          const bf16x4 a = *reinterpret_cast<const bf16x4*>(p.x16 + base + e);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[i][j] = bf2f(a[j]);
        }
      } else if (MODE == LN_POST) {
        const bf16x4 hh = *reinterpret_cast<const bf16x4*>(p.h + base + e);
        float rv[4];
        if (p.res32 != nullptr) {  // fp32 residual stream: the previous LayerNorm's un-rounded output (what the reference's autocast keeps)
          const f32x4 rr = *reinterpret_cast<const f32x4*>(p.res32 + base + e);
#pragma unroll
          for (int j = 0; j < 4; ++j) rv[j] = rr[j];
        } else {
          const bf16x4 rr = *reinterpret_cast<const bf16x4*>(p.res + base + e);
#pragma unroll
          for (int j = 0; j < 4; ++j) rv[j] = bf2f(rr[j]);
        }
        bf16x4 zz;
        f32x4 z32;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float hv = bf2f(hh[j]);
          if (p.drop_thresh != 0u) {
            const uint32_t r = rng_u32(rkey, (uint32_t)(e + j));
            hv = rng_keep(r, p.drop_thresh) ? hv * p.drop_scale : 0.f;
          }
          v[i][j] = hv + rv[j];
          zz[j] = f2bf(v[i][j]);
          z32[j] = v[i][j];
        }
        if (p.z_out != nullptr) *reinterpret_cast<bf16x4*>(p.z_out + base + e) = zz;
        if (p.z32_out != nullptr) *reinterpret_cast<f32x4*>(p.z32_out + base + e) = z32;
      } else {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p.x32 + base + e);
        const bf16x4 hh = *reinterpret_cast<const bf16x4*>(p.h + base + e);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[i][j] = a[j] + rs * lsg[i][j] * bf2f(hh[j]); o[j] = v[i][j]; }
        *reinterpret_cast<f32x4*>(p.x_out + base + e) = o;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) s += v[i][j];
    }
    const float mu = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mu; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + p.eps);
    if (lane == 0) { p.mean[row] = mu; p.rstd[row] = rstd; }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e = (i * 64 + lane) * 4;
      bf16x4 o;
      f32x4 o32;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o32[j] = (v[i][j] - mu) * rstd * wv[i][j] + bv[i][j];
        if (p.gelu) o32[j] = gelu_f(o32[j]);
        o[j] = f2bf(o32[j]);
      }
      *reinterpret_cast<bf16x4*>(p.y + base + e) = o;
      if (p.y32 != nullptr) *reinterpret_cast<f32x4*>(p.y32 + base + e) = o32;
    }
  }
}

static int ln_grid(int rows) {
  int blocks = cdiv(rows, 4);
  if (blocks > 2048) blocks = 2048;
  return blocks;
}

// Wide rows (D a multiple of 1024 up to 8192, PLAIN mode): the 3072 / 6144-wide LayerNorms of the classification head
// (model_classification.py:33-48).  One 256-thread workgroup per row, the row stays in registers (<= 8 x 4 values per thread),
// block reductions through LDS.  Few rows (one per sample), so no attempt at the multi-row tiling of the main kernels.
__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void ln_wide_fwd_kernel(LnFwd p, int D) {
  __shared__ float red[4];
  const int row = blockIdx.x, nv = D / 1024;
  const long base = (long)row * D;
  float v[8][4];
  float s = 0.f;
  for (int i = 0; i < nv; ++i) {
    const int e = (i * 256 + threadIdx.x) * 4;
    if (p.x32 != nullptr) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p.x32 + base + e);
      for (int j = 0; j < 4; ++j) v[i][j] = a[j];
    } else {
      const bf16x4 a = *reinterpret_cast<const bf16x4*>(p.x16 + base + e);
      for (int j = 0; j < 4; ++j) v[i][j] = bf2f(a[j]);
    }
    for (int j = 0; j < 4; ++j) s += v[i][j];
  }
  const float mu = block_sum256(s, red) / D;
  float q = 0.f;
  for (int i = 0; i < nv; ++i)
    for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mu; q += d * d; }
  const float rstd = rsqrtf(block_sum256(q, red) / D + p.eps);
  if (threadIdx.x == 0) { p.mean[row] = mu; p.rstd[row] = rstd; }
  for (int i = 0; i < nv; ++i) {
    const int e = (i * 256 + threadIdx.x) * 4;
    const f32x4 w = *reinterpret_cast<const f32x4*>(p.w + e), b = *reinterpret_cast<const f32x4*>(p.b + e);
    bf16x4 o;
    f32x4 o32;
    for (int j = 0; j < 4; ++j) {
      o32[j] = (v[i][j] - mu) * rstd * w[j] + b[j];
      if (p.gelu) o32[j] = gelu_f(o32[j]);
      o[j] = f2bf(o32[j]);
    }
    if (p.y != nullptr) *reinterpret_cast<bf16x4*>(p.y + base + e) = o;
    if (p.y32 != nullptr) *reinterpret_cast<f32x4*>(p.y32 + base + e) = o32;
  }
}

static bool ln_wide(int D, int mode) { return mode == LN_PLAIN && D > 1536 && D % 1024 == 0 && D <= 8192; }

int xfm_ln_fwd_impl(const LnFwd& p, int D, int mode, hipStream_t st) {
  XFM_REQUIRE(p.rows > 0, "ln_fwd: no rows");
  if (ln_wide(D, mode)) {
    hipLaunchKernelGGL(ln_wide_fwd_kernel, dim3(p.rows), dim3(256), 0, st, p, D);
    return xfm_check_launch("ln_fwd_wide");
  }
  XFM_REQUIRE(D == 768 || D == 1536 || D == 1024 || D == 256 || D == 512, "ln_fwd: unsupported width %d", D);
  const int grid = ln_grid(p.rows);
#define LN_CASE(NCH, MD) hipLaunchKernelGGL((ln_fwd_kernel<NCH, MD>), dim3(grid), dim3(256), 0, st, p); break;
#define LN_MODES(NCH)                                                       \
  switch (mode) {                                                           \
    case LN_PLAIN: LN_CASE(NCH, LN_PLAIN)                                   \
    case LN_POST: LN_CASE(NCH, LN_POST)                                     \
    case LN_LS: LN_CASE(NCH, LN_LS)                                         \
    default: xfm_set_error("ln_fwd: bad mode %d", mode); return XFM_E_ARG;  \
  }
  switch (D) {
    case 256: LN_MODES(1) break;
    case 512: LN_MODES(2) break;
    case 768: LN_MODES(3) break;
    case 1024: LN_MODES(4) break;
    default: LN_MODES(6) break;
  }
#undef LN_MODES
#undef LN_CASE
  return xfm_check_launch("ln_fwd");
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
typedef xfm_ln_bwd_args LnBwd;

template <int NCH, int MODE>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwd p) {
  constexpr int D = NCH * 256;
  constexpr int NSET = (MODE == LN_PLAIN) ? 2 : (MODE == LN_POST ? 3 : 4);
  // Column sums (dgamma, dbeta, dbias, dls) live in LDS, one private slice per wave, updated by 16-byte read-modify-writes of the
  // lane's own slots (no atomics: ds_add_f32 serialises per lane -- measured ~200 clocks per wave instruction, 4x slower kernels):
  // 12 * NSET fewer VGPRs per lane than register accumulators (LS mode at D = 768: 170 -> under 128, i.e. 4 waves per SIMD
  // instead of 2, which is what an HBM-bound kernel with a load -> reduce -> load -> store chain per row needs).
  __shared__ f32x4 colsum[4][NSET][NCH][64];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
#pragma unroll
  for (int s = 0; s < NSET; ++s)
#pragma unroll
    for (int i = 0; i < NCH; ++i) colsum[wib][s][i][lane] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto col_add4 = [&](int s, int i, const f32x4& v) {
    f32x4 t = colsum[wib][s][i][lane];
    t += v;
    colsum[wib][s][i][lane] = t;
  };
  float wv[NCH][4], lsg[NCH][4];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int e = (i * 64 + lane) * 4;
    const f32x4 a = *reinterpret_cast<const f32x4*>(p.w + e);
#pragma unroll
    for (int j = 0; j < 4; ++j) { wv[i][j] = a[j]; lsg[i][j] = 0.f; }
    if (MODE == LN_LS) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(p.ls_gamma + e);
#pragma unroll
      for (int j = 0; j < 4; ++j) lsg[i][j] = g[j];
    }
  }
  for (int row = wave; row < p.rows; row += nwaves) {
    const long base = (long)row * D;
    const uint32_t rkey = rng_row_key(p.seed_lo, p.seed_hi, (uint32_t)row);  // dropout stream of this row
    const float mu = p.mean[row], rstd = p.rstd[row];
    float dy[NCH][4], xh[NCH][4];
    float c1 = 0.f, c2 = 0.f;
    // LS mode: the stream gradient and the branch output of this row are needed only after the row reductions -- fetch them now,
    // with everything else (one round trip to HBM per row instead of two)
    f32x4 ds_in[MODE == LN_LS ? NCH : 1];
    bf16x4 h_in[MODE == LN_LS ? NCH : 1];
    if (MODE == LN_LS) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int e = (i * 64 + lane) * 4;
        ds_in[i] = *reinterpret_cast<const f32x4*>(p.dstream + base + e);
        h_in[i] = *reinterpret_cast<const bf16x4*>(p.h + base + e);
      }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e = (i * 64 + lane) * 4;
      const bf16x4 a = *reinterpret_cast<const bf16x4*>(p.dy1 + base + e);
#pragma unroll
      for (int j = 0; j < 4; ++j) dy[i][j] = bf2f(a[j]);
      if (p.dy2 != nullptr) {
        const bf16x4 b2 = *reinterpret_cast<const bf16x4*>(p.dy2 + base + e);
#pragma unroll
        for (int j = 0; j < 4; ++j) dy[i][j] += bf2f(b2[j]);
      }
      if (p.dy32 != nullptr) {
        const f32x4 b3 = *reinterpret_cast<const f32x4*>(p.dy32 + base + e);
#pragma unroll
        for (int j = 0; j < 4; ++j) dy[i][j] += b3[j];
      }
      if (p.x32 != nullptr) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(p.x32 + base + e);
#pragma unroll
        for (int j = 0; j < 4; ++j) xh[i][j] = (xv[j] - mu) * rstd;
      } else {
        const bf16x4 xv = *reinterpret_cast<const bf16x4*>(p.x16 + base + e);
#pragma unroll
        for (int j = 0; j < 4; ++j) xh[i][j] = (bf2f(xv[j]) - mu) * rstd;
      }
      if (p.gelu_b != nullptr) {  // the forward output was GELU(xhat * w + b): back through the activation first
        const f32x4 bb = *reinterpret_cast<const f32x4*>(p.gelu_b + e);
#pragma unroll
        for (int j = 0; j < 4; ++j) dy[i][j] *= gelu_grad_f(fmaf(xh[i][j], wv[i][j], bb[j]));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g = dy[i][j] * wv[i][j];
        c1 += g;
        c2 += g * xh[i][j];
      }
    }
    c1 = wave_sum(c1) * (1.0f / D);
    c2 = wave_sum(c2) * (1.0f / D);
    float rs = 1.f;
    if (MODE == LN_LS && p.row_scale != nullptr) rs = p.row_scale[row / p.rows_per_sample];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e = (i * 64 + lane) * 4;
      float dz[4];
      asm volatile("" ::: "memory");  // keep the column-sum read-modify-writes of chunk i here (hoisted, they cost 12 * NSET VGPRs)
      col_add4(0, i, f32x4{dy[i][0] * xh[i][0], dy[i][1] * xh[i][1], dy[i][2] * xh[i][2], dy[i][3] * xh[i][3]});
      col_add4(1, i, f32x4{dy[i][0], dy[i][1], dy[i][2], dy[i][3]});
#pragma unroll
      for (int j = 0; j < 4; ++j) dz[j] = rstd * (dy[i][j] * wv[i][j] - c1 - xh[i][j] * c2);
      if (MODE == LN_PLAIN) {
        if (p.dx32 != nullptr) {
          f32x4 o;
          if (p.dx_accum) {
            o = *reinterpret_cast<const f32x4*>(p.dx32 + base + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += dz[j];
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = dz[j];
          }
          *reinterpret_cast<f32x4*>(p.dx32 + base + e) = o;
        }
        if (p.dx16 != nullptr) {
          bf16x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = f2bf(dz[j]);
          *reinterpret_cast<bf16x4*>(p.dx16 + base + e) = o;
        }
      } else if (MODE == LN_POST) {
        bf16x4 oh, orr;
        f32x4 a2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float dhv = dz[j];
          if (p.drop_thresh != 0u) {
            const uint32_t r = rng_u32(rkey, (uint32_t)(e + j));
            dhv = rng_keep(r, p.drop_thresh) ? dhv * p.drop_scale : 0.f;
          }
          orr[j] = f2bf(dz[j]);
          oh[j] = f2bf(dhv);
          a2[j] = bf2f(oh[j]);
        }
        col_add4(2, i, a2);
        *reinterpret_cast<bf16x4*>(p.dh + base + e) = oh;
        if (p.dres != nullptr && p.dres != p.dh) *reinterpret_cast<bf16x4*>(p.dres + base + e) = orr;
        if (p.dres32 != nullptr) *reinterpret_cast<f32x4*>(p.dres32 + base + e) = f32x4{dz[0], dz[1], dz[2], dz[3]};  // fp32 gradient stream
      } else {
        const f32x4 ds = ds_in[MODE == LN_LS ? i : 0];
        const bf16x4 hh = h_in[MODE == LN_LS ? i : 0];
        f32x4 o, a2, a3;
        bf16x4 oh;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = ds[j] + dz[j];
          const float dhv = rs * lsg[i][j] * o[j];
          oh[j] = f2bf(dhv);
          a2[j] = bf2f(oh[j]);
          a3[j] = rs * bf2f(hh[j]) * o[j];
        }
        col_add4(2, i, a2);
        col_add4(3, i, a3);
        *reinterpret_cast<f32x4*>(p.dstream + base + e) = o;
        *reinterpret_cast<bf16x4*>(p.dh + base + e) = oh;
      }
    }
  }
  // one slab row per block and set
  __syncthreads();
  const float* cs = reinterpret_cast<const float*>(&colsum[0][0][0][0]);  // [wave][set][column]
#pragma unroll
  for (int s = 0; s < NSET; ++s) {
    float* dst = p.partial + ((long)s * gridDim.x + blockIdx.x) * D;
    for (int c = threadIdx.x; c < D; c += 256)
      dst[c] = (cs[(0 * NSET + s) * D + c] + cs[(1 * NSET + s) * D + c]) + (cs[(2 * NSET + s) * D + c] + cs[(3 * NSET + s) * D + c]);
  }
}

// out[s][c] += sum_b partial[s][b][c]   (each set s has its own destination pointer; null = skip)
// grid (D/64, sets, row groups): a block folds its group's partial rows (64 columns x 4 row-slices, LDS folds the slices); every CU
// takes part (a (D/64) x sets grid alone leaves 200 of the 256 CUs idle while ~48 blocks crawl through up to 9 MB of partials).
// With more than one row group the groups' sums are NOT added with float atomics (the order of the eight adds moved the last bits of
// every LayerNorm / layer-scale / output-bias gradient from run to run): a group parks its sum in the first partial row of its own
// range -- rows and columns no other block reads -- and reduce_sets_fold_kernel adds the groups in group order.  `partial` is scratch
// the caller hands over: it is clobbered.
#define REDUCE_SETS_GROUPS 8
struct ReduceSets { float* partial; float* out[4]; int nblocks; int D; };
__global__ __launch_bounds__(256) void reduce_sets_kernel(ReduceSets r) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int s = blockIdx.y;
  const int per = (r.nblocks + gridDim.z - 1) / gridDim.z;
  const int b0 = blockIdx.z * per;
  int b1 = b0 + per;
  b1 = b1 < r.nblocks ? b1 : r.nblocks;
  float t = 0.f;
  if (c < r.D && r.out[s] != nullptr) {
    const float* src = r.partial + (long)s * r.nblocks * r.D + c;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    int b = b0 + sl;
    for (; b + 12 < b1; b += 16) {
      t0 += src[(long)b * r.D];
      t1 += src[(long)(b + 4) * r.D];
      t2 += src[(long)(b + 8) * r.D];
      t3 += src[(long)(b + 12) * r.D];
    }
    for (; b < b1; b += 4) t0 += src[(long)b * r.D];
    t = (t0 + t1) + (t2 + t3);
  }
  red[sl][cl] = t;
  __syncthreads();
  if (sl == 0 && c < r.D && r.out[s] != nullptr && b0 < b1) {
    const float v = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
    if (gridDim.z == 1) r.out[s][c] += v;
    else r.partial[((long)s * r.nblocks + b0) * r.D + c] = v;   // (every read of this column of this group happened before the barrier)
  }
}
// second half for `groups` > 1: out[s][c] += the groups' parked sums, in group order.  grid (D/256, sets)
__global__ __launch_bounds__(256) void reduce_sets_fold_kernel(ReduceSets r, int groups) {
  const int c = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
  if (c >= r.D || r.out[s] == nullptr) return;
  const int per = (r.nblocks + groups - 1) / groups;
  float v = 0.f;
  for (int g = 0; g < groups; ++g) {
    const int b0 = g * per;
    if (b0 < r.nblocks) v += r.partial[((long)s * r.nblocks + b0) * r.D + c];
  }
  r.out[s][c] += v;
}
static int reduce_sets_groups(int nblocks) { return nblocks >= 64 * REDUCE_SETS_GROUPS ? REDUCE_SETS_GROUPS : 1; }
static int launch_reduce_sets(const ReduceSets& r, int nset, hipStream_t st, const char* what) {
  const int groups = reduce_sets_groups(r.nblocks);
  hipLaunchKernelGGL(reduce_sets_kernel, dim3(cdiv(r.D, 64), nset, groups), dim3(256), 0, st, r);
  int rc = xfm_check_launch(what);
  if (rc != XFM_OK || groups == 1) return rc;
  hipLaunchKernelGGL(reduce_sets_fold_kernel, dim3(cdiv(r.D, 256), nset), dim3(256), 0, st, r, groups);
  return xfm_check_launch(what);
}

// The same fold for a TABLE of deferred reduces (xfm_reduce_sets_batch): grid (max D / 64, 4 sets, items x REDUCE_SETS_GROUPS); an item
// with few partial rows leaves most of its row groups empty.  One launch for the 36 LayerNorm backwards of a 12-layer tower, and a
// second, small one (reduce_sets_batch_fold_kernel) when some item was split over row groups: group order, no float atomics between
// the groups.  The one atomic add per item and column that remains is there because two items of a table may name the same
// destination (a gradient shared by two LayerNorms): two commuting adds onto a zeroed gradient, still order-independent.
#define REDUCE_BATCH_MAX 56
struct ReduceBatch { int n; xfm_reduce_item it[REDUCE_BATCH_MAX]; };
__global__ __launch_bounds__(256) void reduce_sets_batch_kernel(ReduceBatch tb) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int s = blockIdx.y;
  const int e = blockIdx.z / REDUCE_SETS_GROUPS, grp = blockIdx.z % REDUCE_SETS_GROUPS;
  const xfm_reduce_item& r = tb.it[e];
  float* out = (s < r.nset) ? r.out[s] : nullptr;
  const int groups = r.nblocks >= 64 * REDUCE_SETS_GROUPS ? REDUCE_SETS_GROUPS : 1;   // (the per-call kernel's rule: same sums)
  if (grp >= groups) return;
  const int per = (r.nblocks + groups - 1) / groups;
  const int b0 = grp * per;
  int b1 = b0 + per;
  b1 = b1 < r.nblocks ? b1 : r.nblocks;
  float t = 0.f;
  if (c < r.D && out != nullptr) {
    const float* src = r.partial + (long)s * r.nblocks * r.D + c;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    int b = b0 + sl;
    for (; b + 12 < b1; b += 16) {
      t0 += src[(long)b * r.D];
      t1 += src[(long)(b + 4) * r.D];
      t2 += src[(long)(b + 8) * r.D];
      t3 += src[(long)(b + 12) * r.D];
    }
    for (; b < b1; b += 4) t0 += src[(long)b * r.D];
    t = (t0 + t1) + (t2 + t3);
  }
  red[sl][cl] = t;
  __syncthreads();
  if (sl == 0 && c < r.D && out != nullptr && b0 < b1) {
    const float v = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
    if (groups == 1) atomicAdd(out + c, v);
    else const_cast<float*>(r.partial)[((long)s * r.nblocks + b0) * r.D + c] = v;   // parked for the fold (see reduce_sets_kernel)
  }
}
// grid (max D / 256, 4 sets, items): the items that were split over row groups
__global__ __launch_bounds__(256) void reduce_sets_batch_fold_kernel(ReduceBatch tb) {
  const int c = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
  const xfm_reduce_item& r = tb.it[blockIdx.z];
  if (r.nblocks < 64 * REDUCE_SETS_GROUPS || c >= r.D || s >= r.nset || r.out[s] == nullptr) return;
  const int per = (r.nblocks + REDUCE_SETS_GROUPS - 1) / REDUCE_SETS_GROUPS;
  float v = 0.f;
  for (int g = 0; g < REDUCE_SETS_GROUPS; ++g) {
    const int b0 = g * per;
    if (b0 < r.nblocks) v += r.partial[((long)s * r.nblocks + b0) * r.D + c];
  }
  atomicAdd(r.out[s] + c, v);
}
int xfm_reduce_sets_batch_impl(int n, const xfm_reduce_item* items, hipStream_t st) {
  XFM_REQUIRE(n >= 0 && (n == 0 || items != nullptr), "reduce_sets_batch: bad arguments");
  for (int i0 = 0; i0 < n; i0 += REDUCE_BATCH_MAX) {
    ReduceBatch tb;
    tb.n = n - i0 < REDUCE_BATCH_MAX ? n - i0 : REDUCE_BATCH_MAX;
    int dmax = 0;
    for (int i = 0; i < tb.n; ++i) {
      tb.it[i] = items[i0 + i];
      XFM_REQUIRE(tb.it[i].partial != nullptr && tb.it[i].nblocks > 0 && tb.it[i].D > 0 && tb.it[i].nset >= 1 && tb.it[i].nset <= 4,
                  "reduce_sets_batch: bad item %d", i0 + i);
      dmax = tb.it[i].D > dmax ? tb.it[i].D : dmax;
    }
    hipLaunchKernelGGL(reduce_sets_batch_kernel, dim3(cdiv(dmax, 64), 4, tb.n * REDUCE_SETS_GROUPS), dim3(256), 0, st, tb);
    int rc = xfm_check_launch("reduce_sets_batch");
    if (rc != XFM_OK) return rc;
    bool split = false;
    for (int i = 0; i < tb.n; ++i) split = split || tb.it[i].nblocks >= 64 * REDUCE_SETS_GROUPS;
    if (split) {
      hipLaunchKernelGGL(reduce_sets_batch_fold_kernel, dim3(cdiv(dmax, 256), 4, tb.n), dim3(256), 0, st, tb);
      rc = xfm_check_launch("reduce_sets_batch_fold");
      if (rc != XFM_OK) return rc;
    }
  }
  return XFM_OK;
}

int xfm_ln_bwd_grid(int rows) {
  static const int rpb = getenv("XFM_LN_BWD_ROWS") ? atoi(getenv("XFM_LN_BWD_ROWS")) : 8;  // tuning knob (8 measured: fusion tower 13.95 -> 13.56 ms)
  int blocks = cdiv(rows, rpb);  // rows per workgroup: enough waves per CU for an HBM-bound kernel at M = 7680
  static const int cap = getenv("XFM_LN_BWD_BLOCKS") ? atoi(getenv("XFM_LN_BWD_BLOCKS")) : 768;  // tuning knob
  if (blocks > cap) blocks = cap;  // 3 blocks (12 waves) per CU: the kernel is HBM-bound and needs the loads in flight
  if (blocks < 1) blocks = 1;
  return blocks;
}

// backward of the wide-row form: dx = rstd (dy w - mean(dy w) - xhat mean(dy w xhat)); dgamma / dbeta by fp32 atomics per column
// (rows are few: one per sample of a classification batch)
__global__ __launch_bounds__(256) void ln_wide_bwd_kernel(LnBwd p, int D, float* dgamma, float* dbeta) {
  __shared__ float red[4];
  const int row = blockIdx.x, nv = D / 1024;
  const long base = (long)row * D;
  const float mu = p.mean[row], rstd = p.rstd[row];
  float dy[8][4], xh[8][4];
  float c1 = 0.f, c2 = 0.f;
  for (int i = 0; i < nv; ++i) {
    const int e = (i * 256 + threadIdx.x) * 4;
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(p.dy1 + base + e);
    const f32x4 w = *reinterpret_cast<const f32x4*>(p.w + e);
    for (int j = 0; j < 4; ++j) dy[i][j] = bf2f(a[j]);
    if (p.dy2 != nullptr) {
      const bf16x4 b2 = *reinterpret_cast<const bf16x4*>(p.dy2 + base + e);
      for (int j = 0; j < 4; ++j) dy[i][j] += bf2f(b2[j]);
    }
    if (p.dy32 != nullptr) {
      const f32x4 b3 = *reinterpret_cast<const f32x4*>(p.dy32 + base + e);
      for (int j = 0; j < 4; ++j) dy[i][j] += b3[j];
    }
    if (p.x32 != nullptr) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(p.x32 + base + e);
      for (int j = 0; j < 4; ++j) xh[i][j] = (x[j] - mu) * rstd;
    } else {
      const bf16x4 x = *reinterpret_cast<const bf16x4*>(p.x16 + base + e);
      for (int j = 0; j < 4; ++j) xh[i][j] = (bf2f(x[j]) - mu) * rstd;
    }
    if (p.gelu_b != nullptr) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(p.gelu_b + e);
      for (int j = 0; j < 4; ++j) dy[i][j] *= gelu_grad_f(fmaf(xh[i][j], w[j], bb[j]));
    }
    for (int j = 0; j < 4; ++j) {
      const float gw = dy[i][j] * w[j];
      c1 += gw;
      c2 += gw * xh[i][j];
      if (dgamma != nullptr) atomicAdd(dgamma + e + j, dy[i][j] * xh[i][j]);
      if (dbeta != nullptr) atomicAdd(dbeta + e + j, dy[i][j]);
    }
  }
  c1 = block_sum256(c1, red) / D;
  c2 = block_sum256(c2, red) / D;
  for (int i = 0; i < nv; ++i) {
    const int e = (i * 256 + threadIdx.x) * 4;
    const f32x4 w = *reinterpret_cast<const f32x4*>(p.w + e);
    f32x4 dz;
    for (int j = 0; j < 4; ++j) dz[j] = rstd * (dy[i][j] * w[j] - c1 - xh[i][j] * c2);
    if (p.dx32 != nullptr) {
      f32x4 o = dz;
      if (p.dx_accum) {
        const f32x4 old = *reinterpret_cast<const f32x4*>(p.dx32 + base + e);
        for (int j = 0; j < 4; ++j) o[j] += old[j];
      }
      *reinterpret_cast<f32x4*>(p.dx32 + base + e) = o;
    }
    if (p.dx16 != nullptr) {
      bf16x4 o;
      for (int j = 0; j < 4; ++j) o[j] = f2bf(dz[j]);
      *reinterpret_cast<bf16x4*>(p.dx16 + base + e) = o;
    }
  }
}

int xfm_ln_bwd_impl(LnBwd p, int D, int mode, float* dgamma, float* dbeta, float* dbias, float* dls, float* workspace,
                    long workspace_bytes, hipStream_t st) {
  XFM_REQUIRE(p.rows > 0, "ln_bwd: no rows");
  if (ln_wide(D, mode)) {
    hipLaunchKernelGGL(ln_wide_bwd_kernel, dim3(p.rows), dim3(256), 0, st, p, D, dgamma, dbeta);
    return xfm_check_launch("ln_bwd_wide");
  }
  XFM_REQUIRE(D == 768 || D == 1536 || D == 1024 || D == 256 || D == 512, "ln_bwd: unsupported width %d", D);
  const int grid = xfm_ln_bwd_grid(p.rows);
  const int nset = mode == LN_PLAIN ? 2 : (mode == LN_POST ? 3 : 4);
  XFM_REQUIRE(workspace != nullptr && workspace_bytes >= (long)nset * grid * D * 4, "ln_bwd: workspace too small (%ld bytes)",
              workspace_bytes);
  p.partial = workspace;
#define LNB_CASE(NCH, MD) hipLaunchKernelGGL((ln_bwd_kernel<NCH, MD>), dim3(grid), dim3(256), 0, st, p); break;
#define LNB_MODES(NCH)                                                      \
  switch (mode) {                                                           \
    case LN_PLAIN: LNB_CASE(NCH, LN_PLAIN)                                  \
    case LN_POST: LNB_CASE(NCH, LN_POST)                                    \
    case LN_LS: LNB_CASE(NCH, LN_LS)                                        \
    default: xfm_set_error("ln_bwd: bad mode %d", mode); return XFM_E_ARG;  \
  }
  switch (D) {
    case 256: LNB_MODES(1) break;
    case 512: LNB_MODES(2) break;
    case 768: LNB_MODES(3) break;
    case 1024: LNB_MODES(4) break;
    default: LNB_MODES(6) break;
  }
#undef LNB_MODES
#undef LNB_CASE
  int rc = xfm_check_launch("ln_bwd");
  if (rc != XFM_OK) return rc;
  if (p.defer != nullptr) {   // the caller folds the partials later, with the other LayerNorms of its tower (xfm_reduce_sets_batch)
    *p.defer = xfm_reduce_item{workspace, {dgamma, dbeta, dbias, dls}, grid, D, nset, 0};
    return XFM_OK;
  }
  ReduceSets r{workspace, {dgamma, dbeta, dbias, dls}, grid, D};
  return launch_reduce_sets(r, nset, st, "ln_bwd_reduce");
}

// ---------------------------------------------------------------------------------------------
// column sums via block partials (bias gradients of GELU / QKV linears): out[n] += sum_m Y[m,n]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16* __restrict__ y, long ldy, int M, int N,
                                                             float* __restrict__ partial, int rows_per_block) {
  const int c8 = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (c8 >= N) return;
  const int mb = blockIdx.y * rows_per_block;
  int me = mb + rows_per_block;
  me = me < M ? me : M;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c8 + 8 <= N) {
    for (int m = mb; m < me; ++m) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(y + (long)m * ldy + c8);
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] += bf2f(v[i]);
    }
  } else {
    for (int m = mb; m < me; ++m)
      for (int i = 0; i < 8; ++i)
        if (c8 + i < N) s[i] += bf2f(y[(long)m * ldy + c8 + i]);
  }
  float* dst = partial + (long)blockIdx.y * N;
  for (int i = 0; i < 8; ++i)
    if (c8 + i < N) dst[c8 + i] = s[i];
}

int xfm_colsum_impl(const void* y, long ldy, int M, int N, float* out, float* workspace, long workspace_bytes,
                    hipStream_t st) {
  XFM_REQUIRE(M > 0 && N > 0 && ldy % 8 == 0, "colsum: bad shape M=%d N=%d ldy=%ld", M, N, ldy);
  const int bx = cdiv(N, 256 * 8);
  int by = cdiv(1024, bx);
  if (by > cdiv(M, 32)) by = cdiv(M, 32);
  if (by > 256) by = 256;
  const int rpb = cdiv(M, by);
  by = cdiv(M, rpb);
  XFM_REQUIRE(workspace != nullptr && workspace_bytes >= (long)by * N * 4, "colsum: workspace too small");
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(bx, by), dim3(256), 0, st, (const bf16*)y, ldy, M, N, workspace, rpb);
  int rc = xfm_check_launch("colsum");
  if (rc != XFM_OK) return rc;
  ReduceSets r{workspace, {out, nullptr, nullptr, nullptr}, by, N};
  return launch_reduce_sets(r, 1, st, "colsum_reduce");
}
