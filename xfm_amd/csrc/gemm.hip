// bf16 MFMA GEMMs for the XFM hot path (gfx950).
//
//   gemm_nt : C[M,N] = A[M,K] . B[N,K]^T (+bias, +GELU ...)   forward Linear (B = W) and dgrad (B = W^T copy)
//   gemm_tn : dW[N,K] += dY[M,N]^T . X[M,K]                     wgrad, split over M, fp32 atomics into the grad arena
//
// Covers every Linear on the path: beit2.py:131 (qkv), :162 (proj), :64-68 (fc1/fc2), :229 (patch-embed conv as
// GEMM); xroberta.py:211,224-234 (query/key/value), :301 (attention output), :368 (intermediate), :382 (output),
// :1326,1331 (LM head); xfm.py:117-120 (itm_head), :617-620 (vision_proj/text_proj).
//
// Tiling is for 64-wide wavefronts: 256 threads = 2x2 waves, v_mfma_f32_16x16x32_bf16.  The NT kernel computes
// C^T tiles (A-operand = weight rows, B-operand = activation rows) so that after the K loop every lane owns 8
// CONSECUTIVE output columns of one output row: bias/GELU are applied in registers and the row is stored 16 B per lane.
#include "common.h"
#include <stdlib.h>

enum { EPI_BF16 = 0, EPI_F32 = 1, EPI_GELU = 2, EPI_DGELU = 3, EPI_F32_ACC = 4 };

// raw workgroup barrier fenced for the compiler only: direct-to-LDS loads stay in flight across it (no vmcnt(0) drain)
#define XFM_FENCE() asm volatile("" ::: "memory")
#define XFM_BAR()                    \
  do {                               \
    XFM_FENCE();                     \
    __builtin_amdgcn_s_barrier();    \
    XFM_FENCE();                     \
  } while (0)


struct GemmNT {
  const bf16* A; long lda;
  const bf16* B; long ldb;
  void* C; long ldc;
  const float* bias;
  bf16* aux; long ldaux;
  int M, N, K;
  int group_m;  // row-panels per tile group (L2 locality of the block order)
  int k_splits; // small-tile kernels, EPI_F32_ACC only: gridDim.y K-slices, fp32 atomics into C (1 = off)
  int k_rot;    // 256 x 256 kernel: column phases of the K rotation (0 / 1 = every tile starts at K-tile 0)
  long split_stride;  // EPI_F32 with k_splits > 1: K-slice y stores its partial tile to C + y * split_stride (elements), plain stores
  int skew_from, skew_ticks;  // persistent 256 x 256 kernel: workgroup i >= skew_from starts (i - skew_from) / (grid - skew_from) * skew_ticks
                              // later (10-ns ticks of the constant clock); 0 = all at once.  See launch_nt_256.
  long long* dbg;             // diagnostic (XFM_GEMM_DBG_PTR, tools/tile_timeline.py): wave 0 of every workgroup writes 10-ns timestamps
                              // [tile index, start, K loop done, epilogue done] per tile it walks; NULL in every product call
};

// LDS swizzles (16-B chunk index XOR) for 128-B tile rows read with ds_read_b128.
// X tile: a 16-lane group reads 16 consecutive rows; W tile: rows {0-3,8-11,16-19,24-27}(+4) (see header comment).
__device__ __forceinline__ int swz_x(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ int swz_w(int r) { return ((r >> 1) & 1) | (((r >> 3) & 3) << 1); }

// Epilogue of one wave's (MT*16) x (NT*16) sub-tile: lane (lg, lr) owns row m_base + mt*16 + lr and the 8 consecutive
// columns n_base + np*32 + 8*lg .. +7 of every (mt, np): bias / GELU in registers, one 16-B store per (mt, np).
// lds_bias (optional): the wave's NT*16 bias values already in LDS (fp32, index = column - n_base), zeros when there is no bias -- for
// the persistent 256 x 256 kernel, whose epilogue runs with the next tile's staging loads in flight: a load issued here has to wait
// for all of them (loads return in order).
template <int MT, int NT, int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmNT& g, f32x4 (&acc)[MT][NT], int m_base, int n_base, int lr, int lg,
                                              const float* lds_bias = nullptr) {
  const bool vec_c = (g.ldc % 8) == 0;
  // DGELU: all gelu'(x) loads of the sub-tile go out first, so their latency is paid once, not once per (mt, np)
  bf16x8 pre_all[EPI == EPI_DGELU ? NT / 2 : 1][EPI == EPI_DGELU ? MT : 1];
  const bool vec_aux = (g.ldaux % 8) == 0;
  if (EPI == EPI_DGELU && vec_aux) {
#pragma unroll
    for (int np = 0; np < NT / 2; ++np)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int nb = n_base + np * 32 + 8 * lg, m = m_base + mt * 16 + lr;
        const int mc = m < g.M ? m : g.M - 1, nc = nb + 8 <= g.N ? nb : 0;  // clamped: out-of-range lanes are never stored
        pre_all[np][mt] = *reinterpret_cast<const bf16x8*>(g.aux + (long)mc * g.ldaux + nc);
      }
    // every chunk counts as read here: a chunk left pending on the paths that skip its rows would make the compiler drain all
    // memory operations before the persistent kernel's next tile may reuse the register
#pragma unroll
    for (int np = 0; np < NT / 2; ++np)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(pre_all[np][mt]));
  }
#pragma unroll
  for (int np = 0; np < NT / 2; ++np) {
    const int nb = n_base + np * 32 + 8 * lg;
    float bv[8];
    if (lds_bias != nullptr) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(lds_bias + np * 32 + 8 * lg), b1 = *reinterpret_cast<const f32x4*>(lds_bias + np * 32 + 8 * lg + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { bv[i] = b0[i]; bv[4 + i] = b1[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) bv[i] = (g.bias != nullptr && nb + i < g.N) ? g.bias[nb + i] : 0.f;
    }
    if (EPI == EPI_F32 && g.k_splits > 1 && blockIdx.y != 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) bv[i] = 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m_base + mt * 16 + lr;
      if (m >= g.M || nb >= g.N) continue;
      float v[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = acc[mt][2 * np][i] + bv[i];
        v[4 + i] = acc[mt][2 * np + 1][i] + bv[4 + i];
      }
      const bool full = (nb + 8 <= g.N) && vec_c;
      if (EPI == EPI_F32 || EPI == EPI_F32_ACC) {
        float* cp = reinterpret_cast<float*>(g.C) + (long)m * g.ldc + nb;
        if (EPI == EPI_F32 && g.k_splits > 1) {  // K-slices leave as separate planes, summed in a fixed order by ksplit_reduce_kernel
          cp += (long)blockIdx.y * g.split_stride;  // (slice 0 carries the bias: bv is zero on the others)
        }
        if (EPI == EPI_F32_ACC && g.k_splits > 1) {  // K-slices meet in C through fp32 atomics; slice 0 carried the bias
          for (int i = 0; i < 8; ++i)
            if (nb + i < g.N) atomicAdd(cp + i, v[i] - (blockIdx.y == 0 ? 0.f : bv[i]));
          continue;
        }
        if (EPI == EPI_F32_ACC) {
          for (int i = 0; i < 8; ++i)
            if (nb + i < g.N) v[i] += cp[i];
        }
        if (full) {
          *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(cp + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
          for (int i = 0; i < 8; ++i)
            if (nb + i < g.N) cp[i] = v[i];
        }
      } else {
        bf16* cp = reinterpret_cast<bf16*>(g.C) + (long)m * g.ldc + nb;
        bf16x8 o;
        if (EPI == EPI_GELU) {
          // GELU and its derivative share one erf / exp evaluation, so the forward stores gelu'(x) for the backward (x = the
          // bf16-rounded pre-activation, the value the reference's autocast GELU sees): the dgrad epilogue is then one
          // multiply per element instead of a second erf evaluation that costs as much as a whole K = 768 MFMA loop.
          bf16* ap = g.aux + (long)m * g.ldaux + nb;
          bf16x8 dact;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float x = bf2f(f2bf(v[i]));
            float cdf, pdf;
            gelu_parts(x, cdf, pdf);
            o[i] = f2bf(x * cdf);
            dact[i] = f2bf(fmaf(x, pdf, cdf));
          }
          if (full) *reinterpret_cast<bf16x8*>(ap) = dact;
          else
            for (int i = 0; i < 8; ++i)
              if (nb + i < g.N) ap[i] = dact[i];
        } else if (EPI == EPI_DGELU) {
          const bf16* ap = g.aux + (long)m * g.ldaux + nb;
          if (full && vec_aux) {
            const bf16x8 dact = pre_all[np][mt];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i] * bf2f(dact[i]));
          } else {
            for (int i = 0; i < 8; ++i) o[i] = (nb + i < g.N) ? f2bf(v[i] * bf2f(ap[i])) : f2bf(0.f);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
        }
        if (full) *reinterpret_cast<bf16x8*>(cp) = o;
        else
          for (int i = 0; i < 8; ++i)
            if (nb + i < g.N) cp[i] = o[i];
      }
    }
  }
}


// NS = LDS stages: 2 = load K-tile kt+1 while computing kt (enough when several workgroups share a CU); 3 / 4 keep one / two
// more K-tiles in flight behind a counted s_waitcnt -- for the small-M problems (text tower) where a CU holds one or two
// workgroups and each K-step would otherwise expose the full L2 latency.
template <int BM, int BN, int EPI, int NS>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNT g) {
  constexpr int MT = BM / 32, NT = BN / 32;
  constexpr int LOADS = (BM + BN) / 32;  // direct-to-LDS loads per thread per stage
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int lr = lane & 15, lg = lane >> 4;
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  int tm, tn;
  grouped_tile(wg, tiles_m, tiles_n, g.group_m, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  // K-slice of this workgroup (k_splits > 1: the LM-head dgrad, K = 50304 against 90 output tiles)
  const int nk_all = g.K / 64;
  const int nk_per = (nk_all + g.k_splits - 1) / g.k_splits;
  const int kt0 = blockIdx.y * nk_per;
  const int nk = nk_all - kt0 < nk_per ? nk_all - kt0 : nk_per;
  if (nk <= 0) return;  // workgroup-uniform, before any barrier
  auto stage = [&](int buf, int kt) {
    char* sA = smem + buf * STAGE;
    char* sB = sA + A_BYTES;
    const int k0 = (kt0 + kt) * 64;
#pragma unroll
    for (int i = 0; i < BM / 32; ++i) {
      const int blk = i * 4 + w;  // one wave-instruction fills 1 KiB = 8 rows x 128 B, lane-linear
      const int r = blk * 8 + (lane >> 3);
      const int c = (lane & 7) ^ swz_x(r);
      int gr = m0 + r;
      gr = gr < g.M ? gr : g.M - 1;
      const bf16* src = g.A + (long)gr * g.lda + k0 + c * 8;
      __builtin_amdgcn_global_load_lds(GLB_PTR(void, src), LDS_PTR(void, sA + blk * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) {
      const int blk = i * 4 + w;
      const int r = blk * 8 + (lane >> 3);
      const int c = (lane & 7) ^ swz_w(r);
      int gr = n0 + r;
      gr = gr < g.N ? gr : g.N - 1;
      const bf16* src = g.B + (long)gr * g.ldb + k0 + c * 8;
      __builtin_amdgcn_global_load_lds(GLB_PTR(void, src), LDS_PTR(void, sB + blk * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane LDS row of each fragment (constant over the K loop)
  int xrow[MT], wrow[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) xrow[mt] = wm * (BM / 2) + mt * 16 + lr;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wrow[nt] = wn * (BN / 2) + (nt >> 1) * 32 + 8 * (lr >> 2) + 4 * (nt & 1) + (lr & 3);

  if (NS == 2) {
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  } else {
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
      if (s0 < nk) stage(s0, s0);
  }
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (NS == 2) {
      if (kt + 1 < nk) stage((kt & 1) ^ 1, kt + 1);
    } else {
      // stage kt must have landed; the younger stages kt+1 .. kt+NS-2 (where they exist) stay in flight across the barrier
      const int younger = nk - 1 - kt < NS - 2 ? nk - 1 - kt : NS - 2;
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      XFM_BAR();  // everyone's share of stage kt has landed; everyone is done reading the slot of stage kt-1
      if (kt + NS - 1 < nk) stage(slot == 0 ? NS - 1 : slot - 1, kt + NS - 1);
    }
    const int cur = NS == 2 ? (kt & 1) : slot;
    const char* sA = smem + cur * STAGE;
    const char* sB = sA + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks * 4 + lg;
      bf16x8 xf[MT], wf[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        xf[mt] = *reinterpret_cast<const bf16x8*>(sA + xrow[mt] * 128 + ((c ^ swz_x(xrow[mt])) << 4));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        wf[nt] = *reinterpret_cast<const bf16x8*>(sB + wrow[nt] * 128 + ((c ^ swz_w(wrow[nt])) << 4));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[mt][nt], 0, 0, 0);
    }
    if (NS == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    } else {
      slot = slot == NS - 1 ? 0 : slot + 1;
    }
  }

  gemm_epilogue<MT, NT, EPI>(g, acc, m0 + wm * (BM / 2), n0 + wn * (BN / 2), lr, lg);
}

// ---------------------------------------------------------------------------------------------
// Large-M variant: 256 x 128 tile, 8 waves (4 x 2, the same 64 x 64 micro-kernel per wave), one workgroup per CU, and a
// 3-slot LDS ring (3 x 48 KiB) filled by direct-to-LDS loads that stay in flight ACROSS the per-step barrier: a counted
// s_waitcnt vmcnt(6) retires only the slot about to be read while the next slot's 6 loads per wave keep flying, and
// the slot after that is issued right behind the barrier.  The projections of this model sit at the MI355X ridge
// (N = 768, K = 768: ~370 FLOP/B), so bytes in flight per CU, not MFMA issue, decide their speed.
// ---------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt_ring_kernel(GemmNT g) {
  constexpr int BM = 256, BN = 128, MT = 4, NT = 4;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;  // 48 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int lr = lane & 15, lg = lane >> 4;
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  int tm, tn;
  grouped_tile(wg, tiles_m, tiles_n, g.group_m, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  auto stage = [&](int slot, int kt) {  // 48 wave-instructions of 1 KiB: 6 per wave (4 of A, 2 of B)
    char* sA = smem + slot * STAGE;
    char* sB = sA + A_BYTES;
    const int k0 = kt * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int blk = i * 8 + w;
      const int r = blk * 8 + (lane >> 3);
      const int c = (lane & 7) ^ swz_x(r);
      int gr = m0 + r;
      gr = gr < g.M ? gr : g.M - 1;
      __builtin_amdgcn_global_load_lds(GLB_PTR(void, g.A + (long)gr * g.lda + k0 + c * 8), LDS_PTR(void, sA + blk * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int blk = i * 8 + w;
      const int r = blk * 8 + (lane >> 3);
      const int c = (lane & 7) ^ swz_w(r);
      int gr = n0 + r;
      gr = gr < g.N ? gr : g.N - 1;
      __builtin_amdgcn_global_load_lds(GLB_PTR(void, g.B + (long)gr * g.ldb + k0 + c * 8), LDS_PTR(void, sB + blk * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int xrow[MT], wrow[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) xrow[mt] = wm * 64 + mt * 16 + lr;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wrow[nt] = wn * 64 + (nt >> 1) * 32 + 8 * (lr >> 2) + 4 * (nt & 1) + (lr & 3);

  const int nk = g.K / 64;
  stage(0, 0);
  if (nk > 1) stage(1, 1);
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // retire slot `kt` (this wave's share), keep the 6 loads of slot kt+1 in flight across the barrier
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's share of slot kt has landed; everyone is done reading slot kt-1
    if (kt + 2 < nk) stage(slot == 0 ? 2 : slot - 1, kt + 2);  // (kt+2) % 3 == (kt-1) % 3: the slot just released
    const char* sA = smem + slot * STAGE;
    const char* sB = sA + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks * 4 + lg;
      bf16x8 xf[MT], wf[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        xf[mt] = *reinterpret_cast<const bf16x8*>(sA + xrow[mt] * 128 + ((c ^ swz_x(xrow[mt])) << 4));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        wf[nt] = *reinterpret_cast<const bf16x8*>(sB + wrow[nt] * 128 + ((c ^ swz_w(wrow[nt])) << 4));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[mt][nt], 0, 0, 0);
    }
    slot = slot == 2 ? 0 : slot + 1;
  }
  gemm_epilogue<MT, NT, EPI>(g, acc, m0 + wm * 64, n0 + wn * 64, lr, lg);
}

static int launch_nt_ring(const GemmNT& g, int epi, hipStream_t st) {
  const int tiles = cdiv(g.M, 256) * cdiv(g.N, 128);
  const size_t smem = 3 * (256 + 128) * 128;
#define XFM_RING_CASE(E)                                                                                       \
  case E: {                                                                                                    \
    static bool attr_set = false;                                                                              \
    if (!attr_set) {                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_ring_kernel<E>),                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                        \
      attr_set = true;                                                                                         \
    }                                                                                                          \
    hipLaunchKernelGGL((gemm_nt_ring_kernel<E>), dim3(tiles), dim3(512), smem, st, g);                         \
    break;                                                                                                     \
  }
  switch (epi) {
    XFM_RING_CASE(EPI_BF16)
    XFM_RING_CASE(EPI_F32)
    XFM_RING_CASE(EPI_GELU)
    XFM_RING_CASE(EPI_DGELU)
    XFM_RING_CASE(EPI_F32_ACC)
    default:
      xfm_set_error("gemm_nt: bad epilogue %d", epi);
      return XFM_E_ARG;
  }
#undef XFM_RING_CASE
  return xfm_check_launch("gemm_nt_ring");
}

// ---------------------------------------------------------------------------------------------
// 256 x 256 tile, 8 waves as 2 (M) x 4 (N), each wave a 128 x 64 output block (128 accumulator VGPRs).  Against the
// 256 x 128 ring this halves the LDS fragment bytes read per MFMA (24 ds_read_b128 per 64 MFMA) and the global->LDS
// bytes per FLOP -- the two rates that bound the ring kernel.
//
// LDS = 2 K-tile buffers x (X 256 rows + W 256 rows) x 128 B = 128 KiB, filled by direct-to-LDS loads in UNITS of 128 rows
// (16 KiB = 2 wave-instructions per wave), ordered by when the compute phases need them:
//   U0 = X rows {0-63, 128-191} (the "a0" half of both M-wave rows)      U1 = W rows {wc*64 + 0-31}  ("b0")
//   U2 = W rows {wc*64 + 32-63} ("b1")                                    U3 = X rows {64-127, 192-255} ("a1")
// A K-tile is computed in 4 phases of 16 MFMA: P0 reads a0,b0 -> (a0,b0); P1 reads b1 -> (a0,b1); P2 reads a1 ->
// (a1,b1); P3 reads nothing -> (a1,b0).  Phase index ph = 4*kt + p issues unit ph+5, so every unit flies >= 4 phases and
// three units (6 loads per wave) stay in flight across every barrier: s_waitcnt vmcnt(6) at the end of a phase retires
// exactly the unit(s) the NEXT phase reads.  A unit overwrites the unit 8 places back, whose last ds_read was >= 3
// phases earlier.
// The two M-wave groups (one wave of each per SIMD) run half a phase apart -- group 1 takes one extra barrier up
// front -- so one group's MFMA section overlaps the other's ds_read/glds section (two barriers per phase).
// RAW: a wave's share of a unit is retired by its own counted vmcnt before barrier #1 of phase ph; readers touch it
// in phase ph+1, i.e. after barrier #2 of phase ph, which every wave of both groups reaches after that wait.
// ---------------------------------------------------------------------------------------------
template <int J>
__device__ __forceinline__ int unit_row(int u) {  // row of the X (J = 0, 3) or W (J = 1, 2) tile held by unit row u
  if (J == 0) return u + (u & 64);
  if (J == 3) return u + 64 + (u & 64);
  if (J == 1) return ((u >> 5) << 6) + (u & 31);
  return ((u >> 5) << 6) + 32 + (u & 31);
}


__device__ __forceinline__ void wait_younger(int y) {  // leave the y youngest units (2 loads each) in flight
  if (y >= 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (y == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (y == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// leave the y youngest staging units (2 loads each) in flight, plus NS more memory operations (the previous tile's output stores,
// issued between the staging units of the prologue and those of the K loop)
template <int NS, int YMAX>
__device__ __forceinline__ void wait_units(int y) {
  if (YMAX >= 7 && y >= 7) wait_vm<14 + NS>();
  else if (YMAX >= 6 && y == 6) wait_vm<12 + NS>();
  else if (YMAX >= 5 && y == 5) wait_vm<10 + NS>();
  else if (YMAX >= 4 && y == 4) wait_vm<8 + NS>();
  else if (y >= 3) wait_vm<6 + NS>();
  else if (y == 2) wait_vm<4 + NS>();
  else if (y == 1) wait_vm<2 + NS>();
  else wait_vm<NS>();
}
// Ring form of the pipeline above: the staging units live in R = D + 3 slots of 16 KiB (unit s in slot s mod R, its 128 rows
// contiguous), phase ph issues unit ph + D, and D - 2 units (2 (D - 2) loads per wave) stay in flight across every barrier.  D = 5 is
// the schedule described above in 128 KiB; D = 7 uses all 160 KiB of LDS and keeps 80 KiB per CU in flight: the K loop runs at the
// pace (operand latency) / (look-ahead) -- the first fetch of every operand line comes from HBM or the Infinity Cache, and the
// workgroups that share it ask for it at the same time -- so two more units in flight shorten every phase.
//
// PERSIST: one workgroup per CU walks the tiles blockIdx.x, blockIdx.x + gridDim.x, ... (the same tile -> XCD assignment as one
// workgroup per tile, gridDim.x being a multiple of 8).  The first D units of the NEXT tile are issued before the epilogue of this
// one -- LDS is free once the K loop is over -- so the next tile's first-K-tile latency and this tile's output stores (whose
// acknowledgement a terminating wave would have to wait for) overlap instead of adding up with a workgroup launch in between.
// CDNA counts stores in vmcnt, in issue order with the loads: an interior tile issues exactly NS output stores per lane between
// unit D - 1 and unit D of the next tile, and the waits that retire units 1..3 allow for them; the wait that retires unit 5 (P3 of
// K-tile 0) is the first that needs the stores acknowledged.
template <int EPI, bool PERSIST, int D>
__global__ __launch_bounds__(512) void gemm_nt_256_kernel(GemmNT g, int tiles) {
  constexpr int BM = 256, BN = 256, MT = 8, NT = 4;
  constexpr int R = D + 3, UNIT = 128 * 128;
  // output stores per lane of an interior tile (16-B stores; fp32 output: two per 8 columns; GELU also stores gelu')
  constexpr int NS = !PERSIST ? 0 : EPI == EPI_GELU || EPI == EPI_F32 ? 32 : EPI == EPI_F32_ACC ? 0 : 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int lr = lane & 15, lg = lane >> 4;
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  const int nk = g.K / 64;
  const int total = 4 * nk;  // staging units

  // per-lane global element offsets of the 8 (unit, instruction) loads of a tile; the K offset is added per K-tile.  Instruction i
  // of wave w fills the 1-KiB block (i*8 + w) of the unit's slot: unit rows (i*8 + w)*8 + (lane >> 3), 16-B chunk lane & 7.
  unsigned soff[4][2];
  // K rotation: the tiles of one row panel that run at the same time on an XCD (GM row panels x 32 / GM columns) would ask for the
  // same A lines at the same moment, and every one of them would wait out the full miss (requests merged on an in-flight fill are
  // "hits" that cost a miss).  Column tn therefore starts its K loop at K-tile (tn mod P) * nk / P: each of the P tiles is the
  // first to touch 1/P of the panel, and reads the rest from L2 several K-steps after a neighbour brought it in.
  int kt0 = 0;
  const int rot_p = g.k_rot < tiles_n ? g.k_rot : tiles_n;
  auto tile_origin = [&](int v, int& m0, int& n0) {
    int tm, tn;
    grouped_tile(xcd_remap(v, tiles), tiles_m, tiles_n, g.group_m, tm, tn);
    m0 = tm * BM;
    n0 = tn * BN;
    kt0 = rot_p > 1 ? (tn % rot_p) * (nk / rot_p) : 0;
  };
  auto tile_offsets = [&](int m0, int n0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = (i * 8 + w) * 8 + (lane >> 3);
      {
        const int r = unit_row<0>(u);
        int gr = m0 + r; gr = gr < g.M ? gr : g.M - 1;
        soff[0][i] = (unsigned)gr * (unsigned)g.lda + (((lane & 7) ^ swz_x(r)) << 3);
      }
      {
        const int r = unit_row<3>(u);
        int gr = m0 + r; gr = gr < g.M ? gr : g.M - 1;
        soff[3][i] = (unsigned)gr * (unsigned)g.lda + (((lane & 7) ^ swz_x(r)) << 3);
      }
      {
        const int r = unit_row<1>(u);
        int gr = n0 + r; gr = gr < g.N ? gr : g.N - 1;
        soff[1][i] = (unsigned)gr * (unsigned)g.ldb + (((lane & 7) ^ swz_w(r)) << 3);
      }
      {
        const int r = unit_row<2>(u);
        int gr = n0 + r; gr = gr < g.N ? gr : g.N - 1;
        soff[2][i] = (unsigned)gr * (unsigned)g.ldb + (((lane & 7) ^ swz_w(r)) << 3);
      }
    }
  };
  int iss = 0;  // ring slot of the next unit to issue
  auto issue = [&](int s) {  // unit s = (K-tile s >> 2, part s & 3) into slot iss; wave-uniform branch
    char* base = smem + iss * UNIT + w * 1024;
    iss = iss + 1 == R ? 0 : iss + 1;
    if (s >= total) return;
    const int j = s & 3;
    int kt = (s >> 2) + kt0;
    kt = kt >= nk ? kt - nk : kt;
    const bf16* src = ((j == 0 || j == 3) ? g.A : g.B) + kt * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned so = j == 0 ? soff[0][i] : j == 1 ? soff[1][i] : j == 2 ? soff[2][i] : soff[3][i];
      // inline asm on purpose: a direct-to-LDS load the compiler can see is drained (s_waitcnt vmcnt(0)) in front of the ds_reads
      // it cannot prove disjoint from it -- every read of a ring slot
      const unsigned lds_addr = (unsigned)(uintptr_t)LDS_PTR(void, base) + (unsigned)(i * 8192);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src + (size_t)so), "s"(lds_addr) : "memory", "m0");
    }
  };

  f32x4 acc[MT][NT];

  // per-lane byte offsets of the fragment reads inside a unit's slot (the 16-B chunk index ks*4 + lg is XOR-ed with the row swizzle,
  // which only depends on the low row bits the unit row shares with the tile row):
  //   X units (a0 / a1): tile rows wr*128 + half*64 + m*16 + lr -> unit rows wr*64 + m*16 + lr
  //   W units (b0 / b1): tile rows wc*64 + half*32 + 4*n + 8*(lr>>2) + (lr&3) -> unit rows wc*32 + 4*n + 8*(lr>>2) + (lr&3)
  const int xs = swz_x(lr);
  const int wu = 8 * (lr >> 2) + (lr & 3);
  const int ws = swz_w(wu);
  const int xbase0 = (wr * 64 + lr) * 128 + (((0 + lg) ^ xs) << 4), xbase1 = (wr * 64 + lr) * 128 + (((4 + lg) ^ xs) << 4);
  const int wbase0 = (wc * 32 + wu) * 128 + (((0 + lg) ^ ws) << 4), wbase1 = (wc * 32 + wu) * 128 + (((4 + lg) ^ ws) << 4);

  bf16x8 xa[4][2], wb0[2][2], wb1[2][2];
  auto read_x = [&](const char* slot) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      xa[m][0] = *reinterpret_cast<const bf16x8*>(slot + xbase0 + m * 2048);
      xa[m][1] = *reinterpret_cast<const bf16x8*>(slot + xbase1 + m * 2048);
    }
  };
  auto read_w = [&](const char* slot, bf16x8 (&wb)[2][2]) {
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      wb[n][0] = *reinterpret_cast<const bf16x8*>(slot + wbase0 + n * 512);
      wb[n][1] = *reinterpret_cast<const bf16x8*>(slot + wbase1 + n * 512);
    }
  };
#define XFM_QUAD(MH, NH, WB)                                                                                     \
  do {                                                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                             \
    _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                                \
    _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                                \
      acc[MH * 4 + m][NH * 2 + n] =                                                                              \
          __builtin_amdgcn_mfma_f32_16x16x32_bf16(WB[n][ks], xa[m][ks], acc[MH * 4 + m][NH * 2 + n], 0, 0, 0);  \
    __builtin_amdgcn_s_setprio(0);                                                                               \
  } while (0)

  int v = blockIdx.x, m0, n0;
  if (PERSIST && g.skew_ticks > 0 && (int)blockIdx.x >= g.skew_from) {
    // start-time skew (launch_nt_256): the workgroups that walk one tile fewer than the others start late, spread over one tile time
    const long long t0 = wall_clock64();
    const long long d = (long long)g.skew_ticks * ((int)blockIdx.x - g.skew_from) / ((int)gridDim.x - g.skew_from);
    while (wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(8);
  }
  tile_origin(v, m0, n0);
  tile_offsets(m0, n0);
  // prologue: units 0..D-1 in flight
#pragma unroll
  for (int s = 0; s < D; ++s) issue(s);
  bool stores_behind = false;  // NS output stores of the previous tile were issued after the units 0..D-1 of this one

  int dbg_n = 0;
  while (true) {
    if (g.dbg != nullptr && tid == 0) {
      long long* d = g.dbg + ((long)blockIdx.x * 8 + dbg_n) * 4;
      d[0] = v; d[1] = wall_clock64();
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // units 0 and 1 (a0, b0 of K-tile 0) retired and visible
    {
      const int y = (total - 1 < D - 1 ? total - 1 : D - 1) - 1;
      if (NS > 0 && stores_behind) wait_units<NS, D - 2>(y);
      else wait_units<0, D - 2>(y);
    }
    XFM_BAR();
    if (wr == 1) XFM_BAR();  // stagger the second M-wave group by one barrier

    int rs = 0;  // ring slot of unit 4*kt
    for (int kt = 0; kt < nk; ++kt) {
      const int ph = 4 * kt;
      const bool plus = NS > 0 && stores_behind && kt == 0;
      const char* s_a0 = smem + rs * UNIT;
      const char* s_b0 = smem + (rs + 1 >= R ? rs + 1 - R : rs + 1) * UNIT;
      const char* s_b1 = smem + (rs + 2 >= R ? rs + 2 - R : rs + 2) * UNIT;
      const char* s_a1 = smem + (rs + 3 >= R ? rs + 3 - R : rs + 3) * UNIT;
      rs = rs + 4 >= R ? rs + 4 - R : rs + 4;
      int last;
      // ---- P0: (a0, b0)
      issue(ph + D);
      read_x(s_a0);
      read_w(s_b0, wb0);
      last = ph + D < total ? ph + D : total - 1;
      if (plus) wait_units<NS, D - 2>(last - (ph + 2));
      else wait_units<0, D - 2>(last - (ph + 2));
      XFM_BAR();
      XFM_QUAD(0, 0, wb0);
      XFM_BAR();
      // ---- P1: (a0, b1)
      issue(ph + D + 1);
      read_w(s_b1, wb1);
      last = ph + D + 1 < total ? ph + D + 1 : total - 1;
      if (plus) wait_units<NS, D - 2>(last - (ph + 3));
      else wait_units<0, D - 2>(last - (ph + 3));
      XFM_BAR();
      XFM_QUAD(0, 1, wb1);
      XFM_BAR();
      // ---- P2: (a1, b1)
      issue(ph + D + 2);
      read_x(s_a1);
      XFM_BAR();
      XFM_QUAD(1, 1, wb1);
      XFM_BAR();
      // ---- P3: (a1, b0); retire a0, b0 of the next K-tile
      issue(ph + D + 3);
      last = ph + D + 3 < total ? ph + D + 3 : total - 1;
      if (D > 5 && plus) wait_units<NS, D - 2>(last - (ph + 5) < 0 ? 0 : last - (ph + 5));  // D > 5: units 6.. of the prologue are older than the stores too
      else wait_units<0, D - 2>(last - (ph + 5) < 0 ? 0 : last - (ph + 5));
      XFM_BAR();
      XFM_QUAD(1, 0, wb0);
      XFM_BAR();
    }
    if (wr == 0) XFM_BAR();  // both groups are past their last LDS read
    if (g.dbg != nullptr && tid == 0) g.dbg[((long)blockIdx.x * 8 + dbg_n) * 4 + 2] = wall_clock64();
    const int cm0 = m0, cn0 = n0;
    // The bias goes out BEFORE the next tile's staging loads and is waited for with a count that leaves exactly those in flight
    // (loads return in order): this wave's 64 values, into the last ring slot (free until P2 of the next tile's first K-tile).
    // (DGELU's gelu'(x) chunks are still loaded inside the epilogue, behind the staging loads: 64 more live registers do not fit.)
    float* lds_bias = reinterpret_cast<float*>(smem + (R - 1) * UNIT + w * 256);
    if (g.bias != nullptr) {
      int col = cn0 + wc * 64 + lane;
      col = col < g.N ? col : g.N - 1;
      const unsigned lds_addr = (unsigned)(uintptr_t)LDS_PTR(void, lds_bias);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g.bias + col), "s"(lds_addr) : "memory", "m0");
    } else {
      lds_bias[lane] = 0.f;
    }
    v += gridDim.x;
    const bool more = PERSIST && v < tiles;
    int ahead = 0;  // staging units of the next tile in flight
    if (more) {
      tile_origin(v, m0, n0);
      tile_offsets(m0, n0);
      iss = 0;
#pragma unroll
      for (int s = 0; s < D; ++s) issue(s);
      ahead = total < D ? total : D;
    }
    wait_units<0, D>(ahead);
    if (g.bias == nullptr) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    gemm_epilogue<MT, NT, EPI>(g, acc, cm0 + wr * 128, cn0 + wc * 64, lr, lg, lds_bias);
    if (g.dbg != nullptr && tid == 0) {
      g.dbg[((long)blockIdx.x * 8 + dbg_n) * 4 + 3] = wall_clock64();
      dbg_n = dbg_n < 7 ? dbg_n + 1 : 7;
    }
    if (!more) break;
    XFM_FENCE();
    // exactly NS stores per lane only when every lane stored every (mt, np) with one 16-B (2 x 16-B for fp32) instruction
    stores_behind = cm0 + BM <= g.M && cn0 + BN <= g.N && (g.ldc % 8) == 0 && (EPI != EPI_GELU || (g.ldaux % 8) == 0);
  }
#undef XFM_QUAD
}

template <int E, bool P, int D>
static void launch_nt_256_as(const GemmNT& g, int grid, int tiles, hipStream_t st) {
  constexpr int smem = (D + 3) * 128 * 128;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_256_kernel<E, P, D>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_nt_256_kernel<E, P, D>), dim3(grid), dim3(512), smem, st, g, tiles);
}

static int launch_nt_256(const GemmNT& g_in, int epi, hipStream_t st) {
  GemmNT g = g_in;
  const int tiles = cdiv(g.M, 256) * cdiv(g.N, 256);
  if ((unsigned long)g.M * (unsigned long)g.lda >= (1ul << 32) || (unsigned long)g.N * (unsigned long)g.ldb >= (1ul << 32)) {
    xfm_set_error("gemm_nt: operand too large for the 256x256 kernel's 32-bit element offsets");
    return XFM_E_ARG;
  }
  // more tiles than CUs: one persistent workgroup per CU (XFM_GEMM_PERSIST=0: one workgroup per tile); XFM_GEMM_NT_D = look-ahead
  // in staging units (5: 128 KiB of LDS, 7: 160 KiB)
  static const int persist_env = getenv("XFM_GEMM_PERSIST") ? atoi(getenv("XFM_GEMM_PERSIST")) : 1;
  static const int d_env = getenv("XFM_GEMM_NT_D") ? atoi(getenv("XFM_GEMM_NT_D")) : 5;
  static const int cus = xfm_cu_count();
  const bool persist = persist_env && cus >= 8 && tiles > cus;
  const int grid = persist ? cus & ~7 : tiles;
  // Start-time skew.  A round of 256 tiles ends with every CU storing its 128 KB (256 KB with gelu') of output at the same moment:
  // 33 - 66 MB that drain at the chip's write bandwidth (7 - 15 us) while no matrix core works, and the next tile's staging loads
  // queue behind the store acknowledgements (in-order vmcnt).  tools/kstep_probe.py: 1.4 - 1.5 us per K-step but 8 - 9 us of fixed cost
  // per round.  The workgroups that walk one tile FEWER than the others (tiles % grid != 0) can start up to one tile time later for
  // free; spread over that time they stay out of step with the rest for the whole launch, and the stores of one group drain under the
  // K loops of the others.  XFM_GEMM_SKEW_US: the spread in microseconds per K-step of the problem (0 = off).
  static const float skew_env = getenv("XFM_GEMM_SKEW_US") ? (float)atof(getenv("XFM_GEMM_SKEW_US")) : 0.f;
  g.skew_from = 0;
  g.skew_ticks = 0;
  g.dbg = nullptr;
  {
    const char* dp = getenv("XFM_GEMM_DBG_PTR");   // (read per launch: the tool sets it around the one call it wants a timeline of)
    if (dp != nullptr && dp[0] != 0) g.dbg = reinterpret_cast<long long*>(strtoull(dp, nullptr, 0));
  }
  if (persist && skew_env > 0.f && tiles % grid != 0) {
    g.skew_from = tiles % grid;
    g.skew_ticks = (int)(skew_env * 100.f * (float)(g.K / 64));
  }
#define XFM_256_CASE(E)                                                                 \
  case E:                                                                               \
    if (d_env == 5) {                                                                   \
      if (persist) launch_nt_256_as<E, true, 5>(g, grid, tiles, st);                    \
      else launch_nt_256_as<E, false, 5>(g, grid, tiles, st);                           \
    } else {                                                                            \
      if (persist) launch_nt_256_as<E, true, 7>(g, grid, tiles, st);                    \
      else launch_nt_256_as<E, false, 7>(g, grid, tiles, st);                           \
    }                                                                                   \
    break;
  switch (epi) {
    XFM_256_CASE(EPI_BF16)
    XFM_256_CASE(EPI_F32)
    XFM_256_CASE(EPI_GELU)
    XFM_256_CASE(EPI_DGELU)
    XFM_256_CASE(EPI_F32_ACC)
    default:
      xfm_set_error("gemm_nt: bad epilogue %d", epi);
      return XFM_E_ARG;
  }
#undef XFM_256_CASE
  return xfm_check_launch("gemm_nt_256");
}

template <int BM, int BN, int NS>
static int launch_nt(const GemmNT& g, int epi, hipStream_t st) {
  const int tiles = cdiv(g.M, BM) * cdiv(g.N, BN);
  const size_t smem = NS * (BM + BN) * 128;
#define XFM_NT_CASE(E)                                                                                         \
  case E: {                                                                                                    \
    static bool attr_set = false;                                                                              \
    if (!attr_set) {                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<BM, BN, E, NS>),                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                        \
      attr_set = true;                                                                                         \
    }                                                                                                          \
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, E, NS>), dim3(tiles, g.k_splits), dim3(256), smem, st, g);      \
    break;                                                                                                     \
  }
  switch (epi) {
    XFM_NT_CASE(EPI_BF16)
    XFM_NT_CASE(EPI_F32)
    XFM_NT_CASE(EPI_GELU)
    XFM_NT_CASE(EPI_DGELU)
    XFM_NT_CASE(EPI_F32_ACC)
    default:
      xfm_set_error("gemm_nt: bad epilogue %d", epi);
      return XFM_E_ARG;
  }
#undef XFM_NT_CASE
  return xfm_check_launch("gemm_nt");
}

// Launch plan of xfm_gemm_nt for a shape: the tile configuration (1 = 128x128, 2 = 64x128, 3 = 64x64, 4 = 256x128 ring, 5 = 256x256
// phase pipeline, 7 = 64x128 on 3 LDS stages, 8 = 64x64 on 4) and, for the tail split, the leading rows that run as whole rounds of
// 256x256 tiles (rows_a > 0: those rows go to configuration 5, the rest is planned again with hint -1).  Exported as
// xfm_gemm_nt_plan so that a profiler / benchmark can attribute a call to the kernels it launches.
static int nt_plan(int M, int N, int K, int epi, int tile_hint, int* rows_a_out, int* k_splits_out) {
  *rows_a_out = 0;
  *k_splits_out = 1;
  int cfg = tile_hint;
  if (cfg <= 0) {  // measured on MI355X (tools/tune_gemm.py): 128x128 pays from ~3 workgroups per CU, else go smaller
    // Tail split: one workgroup per CU, so T tiles of 256x256 cost ceil(T / 256) rounds.  When the last round would be
    // nearly empty (N = 768: 99 x 3 = 297 tiles = 1.16 rounds), the whole rounds run as 256x256 tiles and the remaining
    // rows go to the small-tile kernels, which fill every CU for a fraction of a big-tile time.
    static const int split_env = getenv("XFM_GEMM_TAIL_SPLIT") ? atoi(getenv("XFM_GEMM_TAIL_SPLIT")) : 35;  // tuning knob: max tail % (35 measured best, tools/split_sweep.sh)
    const long tn256 = cdiv(N, 256), t256 = (long)cdiv(M, 256) * tn256;
    if (split_env && tile_hint == 0 && M >= 2048 && t256 > 256 && t256 % 256 != 0 && (t256 % 256) * 100 < split_env * 256) {
      const int rows_a = (int)((t256 / 256) * 256 / tn256) * 256;  // row tiles that exactly fill the whole rounds
      if (rows_a > 0 && rows_a < M) {
        *rows_a_out = rows_a;
        return 5;
      }
    }
    // ~half a round of 256x256 tiles already beats the rest on the tall problems (M = 12608 / 25216); on the packed token rows of
    // the text / fusion towers (M ~ 2600 .. 5300, tools/tune_small.py) it needs ~0.7 of a round, below that 128x128 tiles win
    // (5248x1536x768: 25.2 -> 17.4 us, 2624x3072x768: 24.0 -> 17.1 us)
    if (M >= 2048 && (t256 >= 180 || (t256 >= 120 && M >= 8192))) cfg = 5;
    else if (M >= 2048 && t256 >= 120) cfg = 1;
    else if ((long)cdiv(M, 256) * cdiv(N, 128) >= 768) cfg = 4;  // >= 3 rounds of 256x128 tiles: the 3-slot ring wins on cold operands
    else if ((long)cdiv(M, 128) * cdiv(N, 128) >= 800) cfg = 1;
    else if ((long)cdiv(M, 64) * cdiv(N, 128) >= 512) cfg = 2;
    else if (K >= 1536 && epi != EPI_F32_ACC && (long)cdiv(M, 64) * cdiv(N, 128) < 300) {
      // a long K loop against about one 64x128 tile per CU or fewer (the FFN / QKV dgrads of the text tower and of a 2B-sequence
      // fusion pass): 64x64 tiles double the workgroups and four LDS stages keep three K-tiles in flight
      // (2624x768x3072: 24.4 -> 21.9 us, 1344x768x3072: 24.2 -> 18.5 us, 1344x768x2304: 19.3 -> 14.5 us)
      cfg = 8;
    }
    else if ((long)cdiv(M, 64) * cdiv(N, 128) >= 128 || K >= 1536) {  // under two workgroups per CU (tail-split row blocks, text
      // tower) the 2-stage loop exposes the load latency of every K-step: keep two K-tiles in flight (3-stage 64x128; measured
      // 45.9 -> 31.7 us on 3456x768x3072, 18.9 -> 20.6 us on the 720-tile 7680x768x768 which therefore stays 2-stage)
      cfg = 7;
      // ... and when the loop is VERY long against very few tiles (LM-head dgrad: K = 50304, 90 tiles) slice K over gridDim.y;
      // only the fp32-accumulate epilogue can merge slices (atomics), so callers ask for it with a zeroed fp32 C
      const long t = (long)cdiv(M, 64) * cdiv(N, 128);
      if (epi == EPI_F32_ACC && K >= 8192 && t < 192) {
        int sp = (int)(512 / t);
        if (sp > K / 1024) sp = K / 1024;
        *k_splits_out = sp < 1 ? 1 : sp;
      }
    }
    else cfg = 3;
  }
  return cfg;
}

int xfm_gemm_nt_impl(const void* A, long lda, const void* B, long ldb, void* C, long ldc, const float* bias,
                     void* aux, long ldaux, int M, int N, int K, int epi, int tile_hint, hipStream_t st) {
  XFM_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_nt: empty problem M=%d N=%d K=%d", M, N, K);
  XFM_REQUIRE(K % 64 == 0, "gemm_nt: K=%d must be a multiple of 64", K);
  XFM_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, "gemm_nt: lda=%ld ldb=%ld must be multiples of 8", lda, ldb);
  XFM_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0,
              "gemm_nt: operands must be 16-byte aligned");
  XFM_REQUIRE((epi != EPI_GELU && epi != EPI_DGELU) || aux != nullptr, "gemm_nt: epilogue %d needs aux", epi);
  static const int gm_env = getenv("XFM_GEMM_GROUP_M") ? atoi(getenv("XFM_GEMM_GROUP_M")) : 0;  // tuning knob
  static const int rot_env = getenv("XFM_GEMM_KROT") ? atoi(getenv("XFM_GEMM_KROT")) : 0;  // tuning knob (measured neutral)
  GemmNT g{(const bf16*)A, lda, (const bf16*)B, ldb, C, ldc, bias, (bf16*)aux, ldaux, M, N, K, gm_env > 0 ? gm_env : 8, 1, rot_env, 0};
  int rows_a = 0, k_splits = 1;
  const int cfg = nt_plan(M, N, K, epi, tile_hint, &rows_a, &k_splits);
  if (rows_a > 0) {  // tail split: whole rounds of 256x256 tiles first, the remaining rows on the small-tile kernels
    int rc = xfm_gemm_nt_impl(A, lda, B, ldb, C, ldc, bias, aux, ldaux, rows_a, N, K, epi, 5, st);
    if (rc != XFM_OK) return rc;
    const long esz = (epi == EPI_F32 || epi == EPI_F32_ACC) ? 4 : 2;
    return xfm_gemm_nt_impl((const bf16*)A + (long)rows_a * lda, lda, B, ldb, (char*)C + (long)rows_a * ldc * esz, ldc, bias,
                            aux ? (void*)((bf16*)aux + (long)rows_a * ldaux) : nullptr, ldaux, M - rows_a, N, K, epi, -1, st);
  }
  g.k_splits = k_splits;
  static const int exp_cfg = getenv("XFM_GEMM_EXP_CFG") ? atoi(getenv("XFM_GEMM_EXP_CFG")) : 0;  // experiment: replaces config 7 at M >= 4096
  if (exp_cfg > 0 && tile_hint == 0 && cfg == 7 && M >= 4096 && k_splits == 1) return exp_cfg == 9 ? launch_nt<128, 128, 3>(g, epi, st) : exp_cfg == 10 ? launch_nt<128, 128, 4>(g, epi, st) : launch_nt<128, 128, 2>(g, epi, st);
  switch (cfg) {
    case 1: return launch_nt<128, 128, 2>(g, epi, st);
    case 2: return launch_nt<64, 128, 2>(g, epi, st);
    case 7: return launch_nt<64, 128, 3>(g, epi, st);
    case 8: return launch_nt<64, 64, 4>(g, epi, st);
    case 9: return launch_nt<128, 128, 3>(g, epi, st);
    case 10: return launch_nt<128, 128, 4>(g, epi, st);
    case 4: return launch_nt_ring(g, epi, st);
    case 5: return launch_nt_256(g, epi, st);
    default: return launch_nt<64, 64, 2>(g, epi, st);
  }
}

// ---------------------------------------------------------------------------------------------
// Deterministic K-sliced C = A . B^T for a very long K against few output tiles (the LM-head dgrad of xroberta.py:1325-1333 /
// xbert.py:680-697: K = the padded vocabulary, 50304, against a [rows, 768] output).  The slices of gridDim.y write fp32 partial
// planes to a workspace with plain stores and ksplit_reduce_kernel sums them IN SLICE ORDER, rounding once to the output type.
// Why not the fp32-atomic merge of EPI_F32_ACC (round 1-3): the order of the atomic adds changes the last bits of the sum from run
// to run, the sum is an ACTIVATION gradient that is rounded to bf16 next, and an element that sits on a rounding boundary then comes
// out one bf16 ulp apart -- a 2e-5 perturbation that the remaining backward (bf16 roundings at every layer) amplifies to 1e-3 of the
// gradient norm (tools/cold_probe.py: 2 of 28 cold runs, both landing on the same second value).  Atomics stay where their sum is a
// final fp32 parameter gradient.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ksplit_reduce_kernel(const float* __restrict__ ws, int slices, long plane, int M, int N, void* __restrict__ out,
                                                            long ldo, int out_bf16) {
  const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 8;  // 8 consecutive columns of one row (N % 8 == 0)
  if (e >= (long)M * N) return;
  const int m = (int)(e / N), n = (int)(e - (long)m * N);
  f32x4 a0 = *reinterpret_cast<const f32x4*>(ws + e), a1 = *reinterpret_cast<const f32x4*>(ws + e + 4);
  for (int s = 1; s < slices; ++s) {
    a0 += *reinterpret_cast<const f32x4*>(ws + (long)s * plane + e);
    a1 += *reinterpret_cast<const f32x4*>(ws + (long)s * plane + e + 4);
  }
  if (out_bf16) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = f2bf(a0[i]); o[4 + i] = f2bf(a1[i]); }
    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(out) + (long)m * ldo + n) = o;
  } else {
    float* op = reinterpret_cast<float*>(out) + (long)m * ldo + n;
    *reinterpret_cast<f32x4*>(op) = a0;
    *reinterpret_cast<f32x4*>(op + 4) = a1;
  }
}

// K-slices for a shape (1 = the plain kernels are the better plan) and the K-tiles per slice
static int ksplit_plan(int M, int N, int K, int* nk_per_out) {
  const long t = (long)cdiv(M, 64) * cdiv(N, 128);
  int sp = 1;
  if (K >= 8192 && t < 192) {
    sp = (int)(512 / t);
    if (sp > K / 1024) sp = K / 1024;
    if (sp < 1) sp = 1;
  }
  static const int force_env = getenv("XFM_KSPLIT_FORCE") ? atoi(getenv("XFM_KSPLIT_FORCE")) : 0;   // experiment knob
  if (force_env > 0 && force_env <= K / 64) sp = force_env;
  const int nk_all = K / 64, nk_per = cdiv(nk_all, sp);
  *nk_per_out = nk_per;
  return cdiv(nk_all, nk_per);  // slices that own at least one K-tile
}

long xfm_gemm_nt_ksplit_workspace_impl(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int nk_per;
  const int slices = ksplit_plan(M, N, K, &nk_per);
  return slices > 1 ? (long)slices * M * N * 4 : 0;
}

int xfm_gemm_nt_ksplit_impl(const void* A, long lda, const void* B, long ldb, void* out, long ldo, int out_bf16, const float* bias, int M, int N,
                            int K, float* ws, long ws_bytes, hipStream_t st) {
  XFM_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0, "gemm_nt_ksplit: bad shape M=%d N=%d K=%d", M, N, K);
  int nk_per;
  const int slices = ksplit_plan(M, N, K, &nk_per);
  if (slices <= 1) return xfm_gemm_nt_impl(A, lda, B, ldb, out, ldo, bias, nullptr, 0, M, N, K, out_bf16 ? EPI_BF16 : EPI_F32, 0, st);
  XFM_REQUIRE(N % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)out % 16) == 0, "gemm_nt_ksplit: N=%d, ldo=%ld must be multiples of 8 and out 16-byte aligned", N, ldo);
  XFM_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "gemm_nt_ksplit: operands must be 16-byte aligned rows");
  XFM_REQUIRE(ws != nullptr && ((uintptr_t)ws % 16) == 0 && ws_bytes >= (long)slices * M * N * 4,
              "gemm_nt_ksplit: workspace of %ld bytes needed (xfm_gemm_nt_ksplit_workspace)", (long)slices * M * N * 4);
  static const int gm_env = getenv("XFM_GEMM_GROUP_M") ? atoi(getenv("XFM_GEMM_GROUP_M")) : 0;
  GemmNT g{(const bf16*)A, lda, (const bf16*)B, ldb, ws, (long)N, bias, nullptr, 0, M, N, K, gm_env > 0 ? gm_env : 8, slices, 0, (long)M * N};
  // k_splits = the slices that own K-tiles; the kernel re-derives nk_per = ceil(nk_all / k_splits) <= the planned one, under which
  // exactly those slices stay non-empty, so every plane of the workspace is written in full
  XFM_REQUIRE(cdiv(K / 64, cdiv(K / 64, slices)) == slices, "gemm_nt_ksplit: slice plan mismatch");
  static const int tile_env = getenv("XFM_KSPLIT_TILE") ? atoi(getenv("XFM_KSPLIT_TILE")) : 0;   // experiment knob: 1 = 128 x 128 tiles
  int rc = tile_env == 1 ? launch_nt<128, 128, 3>(g, EPI_F32, st) : launch_nt<64, 128, 3>(g, EPI_F32, st);
  if (rc != XFM_OK) return rc;
  hipLaunchKernelGGL(ksplit_reduce_kernel, dim3(cdiv((long)M * N / 8, 256)), dim3(256), 0, st, ws, slices, (long)M * N, M, N, out, ldo, out_bf16);
  return xfm_check_launch("ksplit_reduce");
}

// ---------------------------------------------------------------------------------------------
// wgrad: dW[N,K] += dY[M,N]^T . X[M,K]   (contraction over the row index of both operands)
// Both tiles are staged row-major ([m][n], [m][k], 256-B rows) and consumed with the gfx950 transposed LDS
// read ds_read_b64_tr_b16, which hands each lane 4 consecutive m for its own column.
// ---------------------------------------------------------------------------------------------
struct GemmTN {
  const bf16* dY; long ldy;
  const bf16* X; long ldx;
  float* dW; long ldw;
  float* dbias;  // optional: dbias[n] += sum_m dY[m,n] (bias gradient), folded into the k-tile-0 workgroups
  int M, N, K;
  int m_per_split;
  float* ws;  // per-(split, tile) partial tiles in accumulator-register order, summed by the reduce kernels (null: see direct)
  int direct; // no workspace: 1 = single split, every dW element has one owner -> plain read-modify-write; 0 = fp32 atomics
};

// Several weight gradients of ONE shape in one launch (the three 768 x 768 projections of a fusion layer: 36 tiles each would
// need 9 splits apiece to fill the chip; together they are 108 tiles x 4 splits, one launch and one reduce instead of three).
#define TN_BATCH_MAX 4
struct TnBatch {
  int nb;           // 1 = plain call (the arrays are unused)
  int wg_per;       // workgroups per problem
  long ws_stride;   // floats of workspace per problem
  const bf16* dY[TN_BATCH_MAX];
  const bf16* X[TN_BATCH_MAX];
  float* dW[TN_BATCH_MAX];
  float* dbias[TN_BATCH_MAX];
};

__device__ __forceinline__ int swz_t(int r) { return ((r & 3) | (((r >> 3) & 1) << 2)) << 1; }  // XOR on the 16-B chunk idx

__device__ __forceinline__ bf16x8 tr_read_pair(const char* tile, int row0, int col0, int lr) {
  // rows row0..row0+3 then row0+4..row0+7, columns col0..col0+15; lane lr (0..15 in its 16-lane group) gets column lr
  const int r = row0 + (lr >> 2);
  const int col = col0 + 4 * (lr & 3);
  const int off0 = r * 256 + ((((col >> 3)) ^ swz_t(r)) << 4) + (col & 7) * 2;
  const int r2 = r + 4;
  const int off1 = r2 * 256 + ((((col >> 3)) ^ swz_t(r2)) << 4) + (col & 7) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + off0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + off1));
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo;
  u.s.b = hi;
  return u.v;
}

__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTN g) {
  constexpr int TILE = 64 * 256;  // 64 m-rows x 128 columns bf16
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wn = w >> 1, wk = w & 1;
  const int lr = lane & 15, lg = lane >> 4;
  const int tiles_k = (g.K + 127) / 128, tiles_n = (g.N + 127) / 128;
  const int per_split = tiles_k * tiles_n;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int split = wg / per_split, t = wg % per_split;
  const int n0 = (t / tiles_k) * 128, k0 = (t % tiles_k) * 128;
  const int mbeg = split * g.m_per_split;
  int mend = mbeg + g.m_per_split;
  mend = mend < g.M ? mend : g.M;
  const int nsteps = (mend - mbeg + 63) / 64;

  const bool do_bias = g.dbias != nullptr && k0 == 0;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // this thread's 8 dY columns (chunk tid & 15), over its rows
  u32x4 ry[4], rx[4];
  auto gload = [&](int step) {
    const int mb = mbeg + step * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = i * 256 + tid, r = q >> 4, c = q & 15;
      const int m = mb + r;
      const bool okm = m < mend;
      const int nn = n0 + c * 8, kk = k0 + c * 8;
      ry[i] = u32x4{0, 0, 0, 0};
      rx[i] = u32x4{0, 0, 0, 0};
      if (okm && nn < g.N) {
        if (nn + 8 <= g.N) ry[i] = *reinterpret_cast<const u32x4*>(g.dY + (long)m * g.ldy + nn);
        else {
          union { bf16 h[8]; u32x4 v; } u; u.v = u32x4{0, 0, 0, 0};
          for (int e = 0; e < 8; ++e) if (nn + e < g.N) u.h[e] = g.dY[(long)m * g.ldy + nn + e];
          ry[i] = u.v;
        }
      }
      if (okm && kk < g.K) rx[i] = *reinterpret_cast<const u32x4*>(g.X + (long)m * g.ldx + kk);  // K % 8 == 0
    }
  };
  auto lstore = [&](int buf) {
    char* sY = smem + buf * 2 * TILE;
    char* sX = sY + TILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = i * 256 + tid, r = q >> 4, c = q & 15;
      const int off = r * 256 + ((c ^ swz_t(r)) << 4);
      *reinterpret_cast<u32x4*>(sY + off) = ry[i];
      *reinterpret_cast<u32x4*>(sX + off) = rx[i];
      if (do_bias) {  // consumed here (after the MFMAs), never at the load site: the loads must stay in flight
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bsum[2 * e] += __uint_as_float(ry[i][e] << 16);
          bsum[2 * e + 1] += __uint_as_float(ry[i][e] & 0xFFFF0000u);
        }
      }
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nsteps > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int cur = s & 1;
    if (s + 1 < nsteps) gload(s + 1);
    const char* sY = smem + cur * 2 * TILE;
    const char* sX = sY + TILE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) af[nt] = tr_read_pair(sY, ks * 32 + 8 * lg, wn * 64 + nt * 16, lr);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) bfr[kt] = tr_read_pair(sX, ks * 32 + 8 * lg, wk * 64 + kt * 16, lr);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
          acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt], bfr[kt], acc[nt][kt], 0, 0, 0);
    }
    if (s + 1 < nsteps) lstore(cur ^ 1);
    __syncthreads();
  }

  if (do_bias) {  // fold the 16 row-slices of each column chunk through LDS (the tiles are no longer needed)
    float* red = reinterpret_cast<float*>(smem);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(tid >> 4) * 128 + (tid & 15) * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < 128 && n0 + tid < g.N) {
      float t2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t2 += red[r * 128 + tid];
      atomicAdd(g.dbias + n0 + tid, t2);
    }
  }
  if (g.ws != nullptr) {  // split partial in accumulator-register order (coalesced 16-B stores); tn_reduce128_kernel sums them
    f32x4* wsp = reinterpret_cast<f32x4*>(g.ws) + ((((long)split * per_split + t) * 4 + w) * 16) * 64 + lane;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) wsp[(nt * 4 + kt) * 64] = acc[nt][kt];
    return;
  }
  // D[i = n slot][j = k col]: lane (lg, lr) holds k = ..+lr and n = ..+4*lg+reg
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int k = k0 + wk * 64 + kt * 16 + lr;
#pragma unroll
      for (int rgi = 0; rgi < 4; ++rgi) {
        const int n = n0 + wn * 64 + nt * 16 + 4 * lg + rgi;
        if (n < g.N && k < g.K) {
          float* dst = g.dW + (long)n * g.ldw + k;
          if (g.direct) *dst += acc[nt][kt][rgi];  // single split: this workgroup is the element's only writer
          else atomicAdd(dst, acc[nt][kt][rgi]);
        }
      }
    }
}


// ---------------------------------------------------------------------------------------------
// wgrad, 128 x 128 tile on a 4-slot LDS ring (the mid-size problems: M of a few thousand token rows, the text / fusion towers).
// The register-staged kernel above keeps ONE K-step in flight, and at two workgroups per CU every 64-row step exposes the
// load latency (measured 1.7 us per step against 0.22 us of MFMA).  Here a step is 32 rows of M (one MFMA k-slice): two
// [32 m][128 col] images (256-B rows, swz_t on the SOURCE address) = 16 KB, filled by direct-to-LDS loads issued as inline
// asm (see gemm_tn_256_kernel: the compiler would drain them in front of every transposed LDS read); THREE steps stay in
// flight behind a counted s_waitcnt vmcnt(8) and one raw barrier per step.  4 waves as 2 (n) x 2 (k), 64 x 64 each -- the
// register layout of gemm_tn_kernel, so the split partials go through the same tn_reduce128_kernel.  Rows past the end of a
// split read a zero row (a wgrad must not see clamped rows).  Needs N % 128 == 0 and K % 128 == 0; the bias gradient rides on
// the matrix cores (dY fragment x ones) in the k-tile-0 workgroups.
// ---------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(256))) static const uint32_t g_zero_row[64] = {0};

__global__ __launch_bounds__(256) void gemm_tn_ring_kernel(GemmTN g, TnBatch bt) {
  constexpr int IMG = 32 * 256, STG = 2 * IMG, NS = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wn = w >> 1, wk = w & 1;
  const int lr = lane & 15, lg = lane >> 4;
  const int tiles_k = g.K / 128, tiles_n = g.N / 128;
  const int per_split = tiles_k * tiles_n;
  int wg = xcd_remap(blockIdx.x, gridDim.x);
  if (bt.nb > 1) {  // which problem of the batch (wave-uniform: scalar loads from the argument arrays)
    const int bi = wg / bt.wg_per;
    wg -= bi * bt.wg_per;
    g.dY = bt.dY[bi];
    g.X = bt.X[bi];
    g.dW = bt.dW[bi];
    g.dbias = bt.dbias[bi];
    g.ws += (long)bi * bt.ws_stride;
  }
  const int split = wg / per_split, t = wg % per_split;
  const int n0 = (t / tiles_k) * 128, k0 = (t % tiles_k) * 128;
  const int mbeg = split * g.m_per_split;
  int mend = mbeg + g.m_per_split;
  mend = mend < g.M ? mend : g.M;
  const int nsteps = (mend - mbeg + 31) / 32;
  const bool do_bias = g.dbias != nullptr && k0 == 0 && wk == 0;

  // this wave's 4 loads of a step: blocks {w, w + 4} of the dY image and of the X image (a block = 4 rows x 256 B = 1 KiB)
  const bf16* zrow = reinterpret_cast<const bf16*>(g_zero_row);
  int lrow[2], lcol[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    lrow[i] = 4 * (i * 4 + w) + (lane >> 4);
    lcol[i] = ((lane & 15) ^ swz_t(lrow[i])) * 8;  // logical column stored at this lane's 16-B slot
  }
  auto issue = [&](int s) {
    if (s >= nsteps) return;
    char* base = smem + (s & (NS - 1)) * STG;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = mbeg + s * 32 + lrow[i];
      const bool ok = m < mend;
      const bf16* sy = ok ? g.dY + (long)m * g.ldy + n0 + lcol[i] : zrow + lcol[i];
      const bf16* sx = ok ? g.X + (long)m * g.ldx + k0 + lcol[i] : zrow + lcol[i];
      const unsigned dy_lds = (unsigned)(uintptr_t)LDS_PTR(void, base) + (unsigned)__builtin_amdgcn_readfirstlane((i * 4 + w) * 1024);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(sy), "s"(dy_lds) : "memory", "m0");
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(sx), "s"(dy_lds + (unsigned)IMG) : "memory", "m0");
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = f2bf(1.0f);

  issue(0);
  issue(1);
  issue(2);
  for (int s = 0; s < nsteps; ++s) {
    // step s has landed (this wave's share); the younger steps s+1, s+2 (4 loads each, where they exist) stay in flight
    const int younger = nsteps - 1 - s < 2 ? nsteps - 1 - s : 2;
    if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    XFM_BAR();      // every wave's share of step s is in LDS; everyone is done reading step s-1, whose slot step s+3 reuses
    issue(s + 3);
    const char* sY = smem + (s & (NS - 1)) * STG;
    const char* sX = sY + IMG;
    bf16x8 af[4], bfr[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) af[nt] = tr_read_pair(sY, 8 * lg, wn * 64 + nt * 16, lr);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) bfr[kt] = tr_read_pair(sX, 8 * lg, wk * 64 + kt * 16, lr);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt], bfr[kt], acc[nt][kt], 0, 0, 0);
    if (do_bias) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt], ones, bacc[nt], 0, 0, 0);
    }
  }

  if (do_bias && lr == 0) {  // D[i = n][j]: every column j holds the same sum; lane (lg, lr = 0) owns rows 4*lg .. 4*lg+3
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) atomicAdd(g.dbias + n0 + wn * 64 + nt * 16 + 4 * lg + i, bacc[nt][i]);
  }
  if (g.ws != nullptr) {  // split partial in accumulator-register order (coalesced 16-B stores); tn_reduce128_kernel sums them
    f32x4* wsp = reinterpret_cast<f32x4*>(g.ws) + ((((long)split * per_split + t) * 4 + w) * 16) * 64 + lane;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) wsp[(nt * 4 + kt) * 64] = acc[nt][kt];
    return;
  }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int k = k0 + wk * 64 + kt * 16 + lr;
#pragma unroll
      for (int rgi = 0; rgi < 4; ++rgi) {
        const int n = n0 + wn * 64 + nt * 16 + 4 * lg + rgi;
        float* dst = g.dW + (long)n * g.ldw + k;
        if (g.direct) *dst += acc[nt][kt][rgi];  // single split: this workgroup is the element's only writer
        else atomicAdd(dst, acc[nt][kt][rgi]);
      }
    }
}

// dW += sum over splits of the 128 x 128 partial tiles (register order of gemm_tn_kernel), fixed summation order.
__global__ __launch_bounds__(256) void tn_reduce128_kernel(const float* __restrict__ ws, float* __restrict__ dW, long ldw, int N, int K,
                                                           int tiles_k, int per_split, int splits, TnBatch bt) {
  if (bt.nb > 1) {  // grid.y = problem
    ws += (long)blockIdx.y * bt.ws_stride;
    dW = bt.dW[blockIdx.y];
  }
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // (tile t, wave w, quad q = nt*4+kt, lane)
  const int lane = (int)(idx & 63), q = (int)((idx >> 6) & 15), w = (int)((idx >> 10) & 3);
  const int t = (int)(idx >> 12);
  if (t >= per_split) return;
  const f32x4* p = reinterpret_cast<const f32x4*>(ws) + idx;
  const long stride = (long)per_split * 4 * 16 * 64;
  f32x4 sum = p[0];
  for (int sp = 1; sp < splits; ++sp) sum += p[sp * stride];
  const int n0 = (t / tiles_k) * 128, k0 = (t % tiles_k) * 128;
  const int wn = w >> 1, wk = w & 1, lr = lane & 15, lg = lane >> 4, nt = q >> 2, kt = q & 3;
  const int k = k0 + wk * 64 + kt * 16 + lr;
  const int n = n0 + wn * 64 + nt * 16 + 4 * lg;
  if (k >= K) return;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (n + i < N) dW[(long)(n + i) * ldw + k] += sum[i];
}

// ---------------------------------------------------------------------------------------------
// wgrad on the 256 x 256 phase pipeline (see gemm_nt_256_kernel): dW tile 256 (n) x 256 (k), 8 waves as 2 (n) x 4 (k),
// a K-tile = 64 rows of M.  Staging units are [64 m][128 col] images (256-B rows, swz_t on the source address) of the
// column subsets each phase consumes: U0 = dY cols {wr*128 + 0..63}, U1 = X cols {wc*64 + 0..31}, U2 = X cols
// {wc*64 + 32..63}, U3 = dY cols {wr*128 + 64..127}; fragments come out with ds_read_b64_tr_b16.  No bounds checks:
// the launcher only picks this kernel for M % 64 == 0 and N, K % 256 == 0.  The bias gradient (column sums of dY) rides
// on the matrix cores: in the k-tile-0 workgroups wave wc multiplies its wc-th dY fragment of each half with a ones
// fragment (4 extra MFMA per K-tile per wave).
// ---------------------------------------------------------------------------------------------
// One (tile, M-range) of the pipeline: rows [mbeg, mbeg + 64 nk) of dY columns [n0, n0 + 256) against X columns [k0, k0 + 256).
struct Tn256Seg {
  const bf16* dY; const bf16* X;
  unsigned ldy, ldx;
  int n0, k0, mbeg, nk;
  bool do_bias;
  int rows;   // RAGGED: rows of this piece that exist (the last K-step of a problem whose M is no multiple of 64 is short)
  // Progress throttle between the workgroups of ONE XCD (grouped kernel, whole tiles; window = 0: off).  The 32 workgroups of an XCD
  // walk 32 consecutive tiles, which share their dY / X panels in groups -- 15 panels instead of 64 -- but only while they are within
  // a few K-steps of each other: a K-step of those panels is 480 KB of the XCD's 4 MB of L2, and left alone the workgroups drift
  // apart over the 394 K-steps of a tile (27 GB per step beyond L2 against ~7 if every shared panel were read once per XCD).  Every
  // 4th K-step wave 0 publishes base + K-steps done in row[slot] and holds the workgroup's next barrier back while it is more than
  // `window` K-steps ahead of the slowest of the row's n workgroups (bounded spin: a workgroup that is not resident yet, or a stale
  // row, costs at most the bound, never a hang).
  unsigned* row; int slot, n, window; unsigned prog0;
};
__device__ __attribute__((aligned(16))) const unsigned tn_zero16[4] = {0u, 0u, 0u, 0u};

// RAGGED: rows at or past sg.rows are staged as zeros (their lanes point the direct-to-LDS load at a 16-byte zero constant).
template <bool RAGGED, bool SYNC = false>
__device__ __forceinline__ void tn256_mainloop(const Tn256Seg& sg, char* smem, f32x4 (&acc)[8][4], f32x4 (&bacc)[2]) {
  constexpr int UNIT = 64 * 256, BUF = 4 * UNIT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 2, wc = w & 3;
  const int lr = lane & 15, lg = lane >> 4;
  const int n0 = sg.n0, k0 = sg.k0, mbeg = sg.mbeg, nk = sg.nk;
  const int total = 4 * nk;
  const bool do_bias = sg.do_bias;

  // per-lane source element offsets (K-tile 0) and wave-uniform LDS destinations of the 8 (unit, instruction) loads
  unsigned soff[4][2];
  int doff[4][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int b = i * 8 + w;                 // 1-KiB block of the unit: rows 4b .. 4b+3
    const int r = 4 * b + (lane >> 4);
    const int c = ((lane & 15) ^ swz_t(r)) * 8;  // logical column (0..127) stored at this lane's 16-B slot
    const unsigned rowy = (unsigned)(mbeg + r) * (unsigned)sg.ldy, rowx = (unsigned)(mbeg + r) * (unsigned)sg.ldx;
    soff[0][i] = rowy + n0 + (c >> 6) * 128 + (c & 63);
    soff[3][i] = rowy + n0 + (c >> 6) * 128 + 64 + (c & 63);
    soff[1][i] = rowx + k0 + (c >> 5) * 64 + (c & 31);
    soff[2][i] = rowx + k0 + (c >> 5) * 64 + 32 + (c & 31);
    doff[0][i] = 0 * UNIT + b * 1024;
    doff[1][i] = 1 * UNIT + b * 1024;
    doff[2][i] = 2 * UNIT + b * 1024;
    doff[3][i] = 3 * UNIT + b * 1024;
  }
  auto issue = [&](int s) {
    if (s >= total) return;
    const int kt = s >> 2, j = s & 3;
    char* base = smem + (kt & 1) * BUF;
    const bool isy = (j == 0 || j == 3);
    const bf16* src = (isy ? sg.dY : sg.X) + (size_t)kt * 64 * (size_t)(isy ? sg.ldy : sg.ldx);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned so = j == 0 ? soff[0][i] : j == 1 ? soff[1][i] : j == 2 ? soff[2][i] : soff[3][i];
      const int dofs = j == 0 ? doff[0][i] : j == 1 ? doff[1][i] : j == 2 ? doff[2][i] : doff[3][i];
      // Issued as inline asm on purpose: when the compiler sees a direct-to-LDS load it drains it (s_waitcnt vmcnt(0)) in
      // front of every ds_read_b64_tr_b16, whose intrinsic carries no alias information -- that serialises the pipeline.
      const unsigned lds_addr = (unsigned)(uintptr_t)LDS_PTR(void, base) + (unsigned)__builtin_amdgcn_readfirstlane(dofs);
      const bf16* ptr = src + (size_t)so;
      if (RAGGED && kt * 64 + 4 * (i * 8 + w) + (lane >> 4) >= sg.rows) ptr = reinterpret_cast<const bf16*>(tn_zero16);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(ptr), "s"(lds_addr) : "memory", "m0");
    }
  };

#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bacc[0] = bacc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = f2bf(1.0f);

  // per-lane byte offsets of the transposed fragment reads inside a unit image (+ ks * 8192, + 1024 for rows +4)
  const int lrow = 8 * lg + (lr >> 2);
  const int tsw = swz_t(lrow);
  int offa[4], offb[2];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int col = wr * 64 + f * 16 + 4 * (lr & 3);
    offa[f] = lrow * 256 + (((col >> 3) ^ tsw) << 4) + (col & 7) * 2;
  }
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int col = wc * 32 + f * 16 + 4 * (lr & 3);
    offb[f] = lrow * 256 + (((col >> 3) ^ tsw) << 4) + (col & 7) * 2;
  }
  auto tr_pair = [&](const char* p) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, p + 1024));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
  };
  bf16x8 xa[4][2], wb0[2][2], wb1[2][2];
  auto read_a = [&](const char* unit) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      xa[f][0] = tr_pair(unit + offa[f]);
      xa[f][1] = tr_pair(unit + offa[f] + 8192);
    }
  };
  auto read_b = [&](const char* unit, bf16x8 (&wb)[2][2]) {
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      wb[f][0] = tr_pair(unit + offb[f]);
      wb[f][1] = tr_pair(unit + offb[f] + 8192);
    }
  };
#define XFM_TQUAD(MH, NH, WB)                                                                                    \
  do {                                                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                             \
    _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                                \
    _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                                \
      acc[MH * 4 + m][NH * 2 + n] =                                                                              \
          __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[m][ks], WB[n][ks], acc[MH * 4 + m][NH * 2 + n], 0, 0, 0);  \
    __builtin_amdgcn_s_setprio(0);                                                                               \
  } while (0)
#define XFM_TBIAS(H)                                                                                             \
  do {                                                                                                           \
    if (do_bias) {                                                                                               \
      _Pragma("unroll") for (int f = 0; f < 4; ++f)                                                              \
        if (f == wc) {                                                                                           \
          bacc[H] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[f][0], ones, bacc[H], 0, 0, 0);                  \
          bacc[H] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[f][1], ones, bacc[H], 0, 0, 0);                  \
        }                                                                                                        \
    }                                                                                                            \
  } while (0)

#pragma unroll
  for (int s = 0; s < 5; ++s) issue(s);
  wait_younger((total - 1 < 4 ? total - 1 : 4) - 1);
  XFM_BAR();
  if (wr == 1) XFM_BAR();

  for (int kt = 0; kt < nk; ++kt) {
    const char* buf = smem + (kt & 1) * BUF;
    const int ph = 4 * kt;
    int last;
    if (SYNC && sg.window > 0 && (kt & 3) == 0 && w == 0) {   // (see Tn256Seg; compiled into the throttled build only)
      const unsigned mine = sg.prog0 + (unsigned)kt;
      if (lane == 0) __hip_atomic_store(sg.row + sg.slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int spin = 0; spin < 3000; ++spin) {
        unsigned v = lane < sg.n ? __hip_atomic_load(sg.row + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
          const unsigned o = (unsigned)__shfl_xor((int)v, off, 64);
          v = o < v ? o : v;
        }
        if (mine <= v + (unsigned)sg.window) break;
        __builtin_amdgcn_s_sleep(8);
      }
    }
    // ---- P0: (a0, b0)
    issue(ph + 5);
    read_a(buf + 0 * UNIT);
    read_b(buf + 1 * UNIT, wb0);
    last = ph + 5 < total ? ph + 5 : total - 1;
    wait_younger(last - (ph + 2));
    XFM_BAR();
    XFM_TQUAD(0, 0, wb0);
    XFM_TBIAS(0);
    XFM_BAR();
    // ---- P1: (a0, b1)
    issue(ph + 6);
    read_b(buf + 2 * UNIT, wb1);
    last = ph + 6 < total ? ph + 6 : total - 1;
    wait_younger(last - (ph + 3));
    XFM_BAR();
    XFM_TQUAD(0, 1, wb1);
    XFM_BAR();
    // ---- P2: (a1, b1)
    issue(ph + 7);
    read_a(buf + 3 * UNIT);
    XFM_BAR();
    XFM_TQUAD(1, 1, wb1);
    XFM_TBIAS(1);
    XFM_BAR();
    // ---- P3: (a1, b0)
    issue(ph + 8);
    last = ph + 8 < total ? ph + 8 : total - 1;
    wait_younger(last - (ph + 5) < 0 ? 0 : last - (ph + 5));
    XFM_BAR();
    XFM_TQUAD(1, 0, wb0);
    XFM_BAR();
  }
  if (wr == 0) XFM_BAR();
#undef XFM_TQUAD
#undef XFM_TBIAS

}

// the bias gradient of the tile's 256 dY columns: D[i = n][j], every column j holds the same sum; lane (lg, lr = 0) owns rows 4 lg .. 4 lg + 3
template <bool ATOMIC>
__device__ __forceinline__ void tn256_bias_out(float* dbias, int n0, const f32x4 (&bacc)[2]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 2, wc = w & 3, lr = lane & 15, lg = lane >> 4;
  if (lr != 0) return;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* d = dbias + n0 + wr * 128 + h * 64 + wc * 16 + 4 * lg + i;
      if (ATOMIC) atomicAdd(d, bacc[h][i]);
      else *d += bacc[h][i];
    }
}
// a cut tile's piece: its share of the 256 column sums parked in the piece's slot (tn_group_fixup_kernel adds the pieces in
// workgroup order: a float atomic per piece moved the last bits of the bias gradient of whichever problem the cut tiles belong to)
__device__ __forceinline__ void tn256_bias_part(float* part, const f32x4 (&bacc)[2]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 2, wc = w & 3, lr = lane & 15, lg = lane >> 4;
  if (lr != 0) return;
#pragma unroll
  for (int h = 0; h < 2; ++h) *reinterpret_cast<f32x4*>(part + wr * 128 + h * 64 + wc * 16 + 4 * lg) = bacc[h];
}
// partial tile -> workspace slot, one coalesced 16-B store per accumulator register quad (the reduce kernels read the same order)
__device__ __forceinline__ void tn256_store_partial(float* ws, long slot, const f32x4 (&acc)[8][4]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  f32x4* wsp = reinterpret_cast<f32x4*>(ws) + ((slot * 8 + w) * 32) * 64 + lane;
#pragma unroll
  for (int nt = 0; nt < 8; ++nt)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) wsp[(nt * 4 + kt) * 64] = acc[nt][kt];
}
template <bool ATOMIC>
__device__ __forceinline__ void tn256_add_out(float* dW, long ldw, int n0, int k0, const f32x4 (&acc)[8][4]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 2, wc = w & 3, lr = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < 8; ++nt)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int k = k0 + wc * 64 + kt * 16 + lr;
#pragma unroll
      for (int rgi = 0; rgi < 4; ++rgi) {
        const int n = n0 + wr * 128 + nt * 16 + 4 * lg + rgi;
        if (ATOMIC) atomicAdd(dW + (long)n * ldw + k, acc[nt][kt][rgi]);
        else dW[(long)n * ldw + k] += acc[nt][kt][rgi];
      }
    }
}

__global__ __launch_bounds__(512) void gemm_tn_256_kernel(GemmTN g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_k = g.K / 256, tiles_n = g.N / 256;
  const int per_split = tiles_k * tiles_n;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int split = wg / per_split, t = wg % per_split;
  const int n0 = (t / tiles_k) * 256, k0 = (t % tiles_k) * 256;
  const int mbeg = split * g.m_per_split;
  int mend = mbeg + g.m_per_split;
  mend = mend < g.M ? mend : g.M;
  const Tn256Seg sg{g.dY, g.X, (unsigned)g.ldy, (unsigned)g.ldx, n0, k0, mbeg, (mend - mbeg) / 64, g.dbias != nullptr && k0 == 0, 0};
  f32x4 acc[8][4], bacc[2];
  tn256_mainloop<false>(sg, smem, acc, bacc);
  if (sg.do_bias) tn256_bias_out<true>(g.dbias, n0, bacc);
  if (g.ws != nullptr) {  // tn_reduce_kernel sums the splits
    tn256_store_partial(g.ws, (long)split * per_split + t, acc);
    return;
  }
  tn256_add_out<true>(g.dW, g.ldw, n0, k0, acc);
}

// ---------------------------------------------------------------------------------------------
// GROUPED weight gradients (round 4): many problems of one M (every projection of several layers) in ONE persistent launch.
// A single wgrad has 9-36 output tiles of 256 x 256 and needs 7+ M-splits to fill 256 CUs: each split writes a 256-KB fp32 partial
// per tile (64 MB per GEMM whatever its shape -- one accumulator tile per CU) that a reduce kernel reads back, ~20 % on top of the MFMA
// loop.  With the tiles of ALL queued problems in one list, workgroup i walks WHOLE tiles i, i + G, ... over the full M (one owner per
// dW element: plain += into the fp32 gradient, bias gradient included) and only the last total % G tiles are cut stream-K style:
// their R = r * nk K-steps are dealt out evenly over the G workgroups (boundaries snapped so that no piece is shorter than 4 steps),
// at most two partial pieces per workgroup go to workspace slots 2 i / 2 i + 1, and a fix-up kernel adds each cut tile's pieces in
// workgroup order -- the same bits on every run.
// ---------------------------------------------------------------------------------------------
#define TN_GROUP_MAX 48
struct TnGroupProb {
  const bf16* dY; const bf16* X;
  float* dW; float* dbias;
  unsigned ldy, ldx;
  long ldw;
  int tiles_k;
  int tile_end;   // prefix: this problem owns tiles [previous tile_end, tile_end)
};
struct TnGroup {
  int nprob, nk;             // problems; K-steps (64 rows of M, the last one possibly short) per tile
  int M;                     // rows
  int total_tiles, full_tiles;
  int sk_wgs;                // workgroups that share the cut tiles (0: none)
  long sk_iters;             // (total_tiles - full_tiles) * nk
  float* ws;
  float* ws_bias;            // 256 floats per partial-piece slot, behind the slots' tiles
  unsigned* prog;            // progress rows of the XCDs (8 x 64 words; Tn256Seg::row), epoch << 20 | K-steps done
  unsigned epoch;
  int window;                // 0: no throttle
  TnGroupProb p[TN_GROUP_MAX];
};
__host__ __device__ __forceinline__ long tn_sk_bound(long R, int nk, int sk_wgs, int i) {
  long raw = (long)i * R / sk_wgs;
  const int rem = (int)(raw % nk);
  if (rem < 4) raw -= rem;
  else if (nk - rem < 4) raw += nk - rem;
  return raw;
}

// SYNC: the build with the XCD progress throttle (XFM_TN_SYNC_WINDOW; the default build carries none of its registers)
template <bool SYNC>
__global__ __launch_bounds__(512) void gemm_tn_group_kernel(TnGroup G) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wg = xcd_remap(blockIdx.x, gridDim.x), nwg = gridDim.x;
  int dp_t = wg;
  long sk_pos = 0, sk_end = 0;
  if (wg < G.sk_wgs) {
    sk_pos = tn_sk_bound(G.sk_iters, G.nk, G.sk_wgs, wg);
    sk_end = tn_sk_bound(G.sk_iters, G.nk, G.sk_wgs, wg + 1);
  }
  const long sk_a = sk_pos;
  bool first = true;
  int tiles_done = 0;
  bool left = false;
  auto leave_rounds = [&]() {
    if (SYNC && G.window > 0 && !left && threadIdx.x == 0) {
      const int per_xcd = nwg >> 3;
      __hip_atomic_store(G.prog + (wg / per_xcd) * 64 + wg % per_xcd, (G.epoch << 20) + 0xFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    left = true;
  };
  for (;;) {   // (everything that steers this loop is a function of blockIdx: uniform over the workgroup)
    int tile, it0, it1;
    long slot = -1;   // >= 0: partial piece
    if (dp_t < G.full_tiles) {
      tile = dp_t;
      dp_t += nwg;
      it0 = 0;
      it1 = G.nk;
    } else if (sk_pos < sk_end) {
      const int rt = (int)(sk_pos / G.nk);
      const long t_end = (long)(rt + 1) * G.nk;
      it0 = (int)(sk_pos - (long)rt * G.nk);
      it1 = (int)((sk_end < t_end ? sk_end : t_end) - (long)rt * G.nk);
      tile = G.full_tiles + rt;
      if (it0 != 0 || it1 != G.nk) slot = 2l * wg + (sk_pos == sk_a ? 0 : 1);
      sk_pos = (long)rt * G.nk + it1;
    } else {
      break;
    }
    if (tile >= G.full_tiles) leave_rounds();   // (past its whole tiles: nobody waits for this workgroup any more)
    if (!first) __syncthreads();   // the previous piece's LDS reads are over before this one's staging lands
    first = false;
    int pi = 0;
    while (pi + 1 < G.nprob && tile >= G.p[pi].tile_end) ++pi;
    const TnGroupProb& P = G.p[pi];
    const int tl = tile - (pi > 0 ? G.p[pi - 1].tile_end : 0);
    const int n0 = (tl / P.tiles_k) * 256, k0 = (tl % P.tiles_k) * 256;
    Tn256Seg sg{P.dY, P.X, P.ldy, P.ldx, n0, k0, it0 * 64, it1 - it0, P.dbias != nullptr && k0 == 0, G.M - it0 * 64, nullptr, 0, 0, 0, 0u};
    if (SYNC && G.window > 0 && slot < 0 && it0 == 0 && it1 == G.nk && tile < G.full_tiles) {   // a whole tile of the data-parallel rounds
      const int per_xcd = nwg >> 3;
      sg.row = G.prog + (wg / per_xcd) * 64;
      sg.slot = wg % per_xcd;
      sg.n = per_xcd < 64 ? per_xcd : 64;
      sg.window = G.window;
      sg.prog0 = (G.epoch << 20) + (unsigned)(tiles_done * G.nk);
    }
    f32x4 acc[8][4], bacc[2];
    tn256_mainloop<true, SYNC>(sg, smem, acc, bacc);
    if (SYNC && sg.window > 0) ++tiles_done;
    if (slot >= 0) {
      if (sg.do_bias) tn256_bias_part(G.ws_bias + slot * 256, bacc);
      tn256_store_partial(G.ws, slot, acc);
    } else {
      if (sg.do_bias) tn256_bias_out<false>(P.dbias, n0, bacc);
      tn256_add_out<false>(P.dW, P.ldw, n0, k0, acc);
    }
  }
  leave_rounds();
}

// the cut tiles: dW += the pieces in workgroup order.  grid (64, cut tiles): one thread per accumulator quad, as tn_reduce_kernel.
__global__ __launch_bounds__(256) void tn_group_fixup_kernel(TnGroup G) {
  const int rt = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;   // < 8 * 32 * 64
  const long R = G.sk_iters;
  const long a = (long)rt * G.nk, b = a + G.nk;
  int i0 = (int)(a * G.sk_wgs / R);
  i0 = i0 < G.sk_wgs - 1 ? i0 : G.sk_wgs - 1;
  while (i0 > 0 && tn_sk_bound(R, G.nk, G.sk_wgs, i0) > a) --i0;
  while (i0 + 1 < G.sk_wgs && tn_sk_bound(R, G.nk, G.sk_wgs, i0 + 1) <= a) ++i0;
  int i1 = i0;
  while (i1 + 1 < G.sk_wgs && tn_sk_bound(R, G.nk, G.sk_wgs, i1 + 1) < b) ++i1;
  if (i0 == i1) return;   // one workgroup walked the whole tile and added it to dW itself
  const f32x4* ws = reinterpret_cast<const f32x4*>(G.ws);
  f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = i0; i <= i1; ++i) {
    const long slot = 2l * i + (tn_sk_bound(R, G.nk, G.sk_wgs, i) >= a ? 0 : 1);
    sum += ws[slot * (8 * 32 * 64) + idx];
  }
  const int tile = G.full_tiles + rt;
  int pi = 0;
  while (pi + 1 < G.nprob && tile >= G.p[pi].tile_end) ++pi;
  const TnGroupProb& P = G.p[pi];
  const int tl = tile - (pi > 0 ? G.p[pi - 1].tile_end : 0);
  const int n0 = (tl / P.tiles_k) * 256, k0 = (tl % P.tiles_k) * 256;
  const int lane = idx & 63, q = (idx >> 6) & 31, w = idx >> 11;
  const int wr = w >> 2, wc = w & 3, lr = lane & 15, lg = lane >> 4, nt = q >> 2, kt = q & 3;
  const int k = k0 + wc * 64 + kt * 16 + lr;
  const int n = n0 + wr * 128 + nt * 16 + 4 * lg;
#pragma unroll
  for (int i = 0; i < 4; ++i) P.dW[(long)(n + i) * P.ldw + k] += sum[i];
  if (blockIdx.x == 0 && P.dbias != nullptr && k0 == 0) {   // the pieces' column sums, same order (256 threads, one column each)
    float b = 0.f;
    for (int i = i0; i <= i1; ++i) {
      const long slot = 2l * i + (tn_sk_bound(R, G.nk, G.sk_wgs, i) >= a ? 0 : 1);
      b += G.ws_bias[slot * 256 + threadIdx.x];
    }
    P.dbias[n0 + threadIdx.x] += b;
  }
}

// dW += sum over splits of the partial tiles written by gemm_tn_256_kernel (deterministic: fixed summation order).
// One thread per accumulator quad: (tile t, wave w, quad q = nt*4+kt, lane) -> rows n..n+3 at column k.
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dW, long ldw, int tiles_k,
                                                        int per_split, int splits) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // < per_split * 8 * 32 * 64
  const int lane = (int)(idx & 63), q = (int)((idx >> 6) & 31), w = (int)((idx >> 11) & 7);
  const int t = (int)(idx >> 14);
  if (t >= per_split) return;
  const f32x4* p = reinterpret_cast<const f32x4*>(ws) + idx;
  const long stride = (long)per_split * 8 * 32 * 64;
  f32x4 sum = p[0];
  for (int sp = 1; sp < splits; ++sp) {
    const f32x4 v = p[sp * stride];
    sum += v;
  }
  const int n0 = (t / tiles_k) * 256, k0 = (t % tiles_k) * 256;
  const int wr = w >> 2, wc = w & 3, lr = lane & 15, lg = lane >> 4, nt = q >> 2, kt = q & 3;
  const int k = k0 + wc * 64 + kt * 16 + lr;
  const int n = n0 + wr * 128 + nt * 16 + 4 * lg;
#pragma unroll
  for (int i = 0; i < 4; ++i) dW[(long)(n + i) * ldw + k] += sum[i];
}

static int tn256_plan(int M, int N, int K, int& splits, int& mps) {  // -> number of 256x256 tiles
  const int t256 = (N / 256) * (K / 256);
  splits = 256 / t256;
  const int steps = M / 64;
  if (splits > steps / 8) splits = steps / 8;
  if (splits < 1) splits = 1;
  mps = cdiv(steps, splits) * 64;
  splits = cdiv(M, mps);
  return t256;
}

static bool tn256_eligible(long ldy, long ldx, int M, int N, int K) {
  return M % 64 == 0 && N % 256 == 0 && K % 256 == 0 && N >= 256 && K >= 256 &&
         (unsigned long)M * (unsigned long)(ldy > ldx ? ldy : ldx) < (1ul << 32);
}

// Below this many bytes of split partials the atomics' ~12 us/10 MB beat the extra reduce launch (measured, tools/tune_gemm.py).
#define TN_WS_MIN_BYTES (16l << 20)

static int tn128_plan(int M, int N, int K, int splits_hint, int& splits, int& mps) {  // -> number of 128x128 tiles
  const int tiles = cdiv(N, 128) * cdiv(K, 128);
  splits = splits_hint < 0 ? 0 : splits_hint;
  if (splits <= 0) {
    splits = (432 + tiles / 2) / tiles;        // measured optimum: ~432 workgroups in total (tools/tune_gemm.py)
    int max_splits = M / 480;                  // ... while every split still walks >= ~8 K-steps
    if (max_splits < 2 && M >= 640) max_splits = 2;  // (M = 928, the VQA / retrieval text rows: two splits 15-24 us, one 25-33 us)
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
  }
  mps = cdiv(cdiv(M, splits), 64) * 64;
  splits = cdiv(M, mps);
  return tiles;
}

// Workspace (bytes) xfm_gemm_tn wants for this shape with splits_hint = 0: the split partials of whichever kernel the
// heuristic picks (0 when a single split writes dW directly).
// Ragged M on an otherwise 256-tileable problem (the 577 / 901-token ViT of the 384 / 480 px configurations: M = B * tokens is no
// multiple of 64): the leading rows run on the 256 x 256 kernel and the last M % 64 rows as a second, single-split call that adds
// into dW -- the 128 x 128 ring kernel on all of M costs 30-40 % more (591 vs 900+ TFLOP/s at M = 21624).
static int tn256_body_rows(int M, int N, int K) {
  const int M0 = M - M % 64;
  // (from 8192 rows: at the fusion tower's packed M ~ 5200 the 7 x 36 workgroups of the 256 x 256 plan walk 12 K-steps each and lose to
  // the ring kernel -- 73 vs 56 us at 5252 x 3072 x 768; round 3 had put those calls here too)
  static const int min_rows = getenv("XFM_TN256_RAGGED_MIN") ? atoi(getenv("XFM_TN256_RAGGED_MIN")) : 8192;  // A/B knob
  if (M % 64 == 0 || M0 < min_rows || N % 256 != 0 || K % 256 != 0) return 0;
  int splits, mps;
  return tn256_plan(M0, N, K, splits, mps) >= 18 ? M0 : 0;
}

long xfm_gemm_tn_workspace_impl(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int splits, mps;
  if (const int M0 = tn256_body_rows(M, N, K)) return xfm_gemm_tn_workspace_impl(M0, N, K);
  if (tn256_eligible(N, K, M, N, K)) {
    const int t256 = tn256_plan(M, N, K, splits, mps);
    if (t256 >= 18 && M >= 4096) return (long)splits * N * K * 4;
  }
  const int tiles = tn128_plan(M, N, K, 0, splits, mps);
  const long need = (long)splits * tiles * 128 * 128 * 4;
  return (splits > 1 && need >= TN_WS_MIN_BYTES) ? need : 0;
}

int xfm_gemm_tn_impl(const void* dY, long ldy, const void* X, long ldx, float* dW, long ldw, float* dbias, int M, int N, int K,
                     int splits_hint, float* workspace, long workspace_bytes, hipStream_t st) {
  XFM_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_tn: empty problem M=%d N=%d K=%d", M, N, K);
  XFM_REQUIRE(K % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "gemm_tn: K=%d ldx=%ld ldy=%ld must be multiples of 8", K, ldx, ldy);
  XFM_REQUIRE(((uintptr_t)dY % 16) == 0 && ((uintptr_t)X % 16) == 0, "gemm_tn: operands must be 16-byte aligned");
  // Split partials go to the caller's workspace with plain coalesced stores and are summed by a reduce kernel: fp32 atomics
  // into dW run at ~0.8 TB/s on this part (28 MB of them cost more than the MFMA loop of a mid-size wgrad).  A single
  // split updates dW with plain read-modify-writes.  Atomics remain only when splits > 1 and no workspace was passed.
  if (splits_hint == 0) {
    if (const int M0 = tn256_body_rows(M, N, K)) {
      if (tn256_eligible(ldy, ldx, M0, N, K) && workspace != nullptr && workspace_bytes >= xfm_gemm_tn_workspace_impl(M0, N, K)) {
        int rc = xfm_gemm_tn_impl(dY, ldy, X, ldx, dW, ldw, dbias, M0, N, K, 0, workspace, workspace_bytes, st);
        if (rc != XFM_OK) return rc;
        return xfm_gemm_tn_impl((const bf16*)dY + (long)M0 * ldy, ldy, (const bf16*)X + (long)M0 * ldx, ldx, dW, ldw, dbias, M - M0, N, K, 1,
                                workspace, workspace_bytes, st);
      }
    }
  }
  // 256 x 256 phase-pipelined kernel: shapes without edges and with enough output tiles (splits_hint -3 forces, -4 forbids).
  if (tn256_eligible(ldy, ldx, M, N, K) && splits_hint != -4) {
    int splits, mps;
    const int t256 = tn256_plan(M, N, K, splits, mps);
    const long need = (long)splits * N * K * 4;
    const bool have_ws = workspace != nullptr && workspace_bytes >= need;
    const bool want = splits_hint == -3 || (splits_hint == 0 && t256 >= 18 && M >= 4096 && have_ws);
    if (want) {
      GemmTN g{(const bf16*)dY, ldy, (const bf16*)X, ldx, dW, ldw, dbias, M, N, K, mps, have_ws ? workspace : nullptr, 0};
      static bool attr256 = false;
      const size_t smem256 = 2 * 4 * 64 * 256;
      if (!attr256) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)smem256);
        attr256 = true;
      }
      hipLaunchKernelGGL(gemm_tn_256_kernel, dim3(t256 * splits), dim3(512), smem256, st, g);
      int rc = xfm_check_launch("gemm_tn_256");
      if (rc != XFM_OK || !have_ws) return rc;
      const long quads = (long)t256 * 8 * 32 * 64;
      hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)cdiv(quads, 256)), dim3(256), 0, st, workspace, dW, ldw, K / 256, t256, splits);
      return xfm_check_launch("gemm_tn_reduce");
    }
  }
  int splits, mps;
  const int tiles = tn128_plan(M, N, K, splits_hint, splits, mps);
  const long need = (long)splits * tiles * 128 * 128 * 4;
  const bool use_ws = splits > 1 && workspace != nullptr && workspace_bytes >= need && (need >= TN_WS_MIN_BYTES || splits_hint > 0);
  GemmTN g{(const bf16*)dY, ldy, (const bf16*)X, ldx, dW, ldw, dbias, M, N, K, mps, use_ws ? workspace : nullptr, splits == 1 ? 1 : 0};
  static bool attr_set = false;
  const size_t smem = 4 * 64 * 256;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_ring_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem);
    attr_set = true;
  }
  static const int ring_env = getenv("XFM_TN_RING") ? atoi(getenv("XFM_TN_RING")) : 1;  // A/B knob
  const bool ring = ring_env && N % 128 == 0 && K % 128 == 0 && splits_hint != -5;
  TnBatch one{};
  one.nb = 1;
  if (ring) hipLaunchKernelGGL(gemm_tn_ring_kernel, dim3(tiles * splits), dim3(256), smem, st, g, one);
  else hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles * splits), dim3(256), smem, st, g);
  int rc = xfm_check_launch("gemm_tn");
  if (rc != XFM_OK || !use_ws) return rc;
  const long quads = (long)tiles * 4 * 16 * 64;
  hipLaunchKernelGGL(tn_reduce128_kernel, dim3((unsigned)cdiv(quads, 256)), dim3(256), 0, st, workspace, dW, ldw, N, K, cdiv(K, 128), tiles,
                     splits, one);
  return xfm_check_launch("gemm_tn_reduce128");
}

// nb (<= TN_BATCH_MAX) weight gradients of one shape and one set of leading dimensions in ONE launch of the ring kernel + ONE reduce.
// Falls back to nb plain calls when the shape is not the ring kernel's (N, K multiples of 128) or the workspace is too small.
long xfm_gemm_tn_batch_workspace_impl(int nb, int M, int N, int K) {
  if (nb <= 0 || M <= 0 || N <= 0 || K <= 0) return 0;
  int splits, mps;
  const long single = xfm_gemm_tn_workspace_impl(M, N, K);
  if (nb == 1 || nb > TN_BATCH_MAX || N % 128 != 0 || K % 128 != 0) return single;
  const int tiles = cdiv(N, 128) * cdiv(K, 128);
  int hint = (432 + nb * tiles / 2) / (nb * tiles);
  hint = hint < 1 ? 1 : hint;
  tn128_plan(M, N, K, hint, splits, mps);
  const long need = (long)nb * splits * tiles * 128 * 128 * 4;
  return need > single ? need : single;
}
int xfm_gemm_tn_batch_impl(int nb, const void* const* dY, long ldy, const void* const* X, long ldx, float* const* dW, long ldw,
                           float* const* dbias, int M, int N, int K, float* workspace, long workspace_bytes, hipStream_t st) {
  XFM_REQUIRE(nb >= 1 && dY != nullptr && X != nullptr && dW != nullptr, "gemm_tn_batch: bad arguments");
  const int tiles = cdiv(N, 128) * cdiv(K, 128);
  int splits = 1, mps = M;
  bool batched = nb > 1 && nb <= TN_BATCH_MAX && N % 128 == 0 && K % 128 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && M > 0;
  if (batched) {
    int hint = (432 + nb * tiles / 2) / (nb * tiles);  // ~432 workgroups in total, as for a single problem
    hint = hint < 1 ? 1 : hint;
    tn128_plan(M, N, K, hint, splits, mps);
    batched = splits > 1 && workspace != nullptr && workspace_bytes >= (long)nb * splits * tiles * 128 * 128 * 4;
    for (int i = 0; i < nb && batched; ++i) batched = ((uintptr_t)dY[i] % 16) == 0 && ((uintptr_t)X[i] % 16) == 0;
  }
  if (!batched) {
    for (int i = 0; i < nb; ++i) {
      int rc = xfm_gemm_tn_impl(dY[i], ldy, X[i], ldx, dW[i], ldw, dbias ? dbias[i] : nullptr, M, N, K, 0, workspace, workspace_bytes, st);
      if (rc != XFM_OK) return rc;
    }
    return XFM_OK;
  }
  TnBatch bt{};
  bt.nb = nb;
  bt.wg_per = tiles * splits;
  bt.ws_stride = (long)splits * tiles * 128 * 128;
  for (int i = 0; i < nb; ++i) {
    bt.dY[i] = (const bf16*)dY[i];
    bt.X[i] = (const bf16*)X[i];
    bt.dW[i] = dW[i];
    bt.dbias[i] = dbias ? dbias[i] : nullptr;
  }
  GemmTN g{bt.dY[0], ldy, bt.X[0], ldx, bt.dW[0], ldw, bt.dbias[0], M, N, K, mps, workspace, 0};
  const size_t smem = 4 * 64 * 256;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_ring_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_tn_ring_kernel, dim3(nb * tiles * splits), dim3(256), smem, st, g, bt);
  int rc = xfm_check_launch("gemm_tn_batch");
  if (rc != XFM_OK) return rc;
  const long quads = (long)tiles * 4 * 16 * 64;
  hipLaunchKernelGGL(tn_reduce128_kernel, dim3((unsigned)cdiv(quads, 256), nb), dim3(256), 0, st, workspace, bt.dW[0], ldw, N, K, cdiv(K, 128),
                     tiles, splits, bt);
  return xfm_check_launch("gemm_tn_batch_reduce");
}

// Any number of weight gradients over the SAME M rows (all the projections of the layers whose dY are still alive) -> persistent
// grouped launches of gemm_tn_group_kernel, TN_GROUP_MAX problems each.  Problems the 256 x 256 pipeline does not take (N or K not a
// multiple of 256, fewer than 1024 rows) go through xfm_gemm_tn one by one.
static int tn_group_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    n = n / 8 * 8 > 0 ? n / 8 * 8 : 8;
  }
  return n;
}
static bool tn_group_item_ok(const xfm_tn_item& it, int M) {   // (any M from 1024 rows: the kernel zero-fills a short last K-step)
  return M >= 1024 && it.N % 256 == 0 && it.K % 256 == 0 && it.N >= 256 && it.K >= 256 && it.ldy % 8 == 0 && it.ldx % 8 == 0 &&
         (unsigned long)(M + 64) * (unsigned long)(it.ldy > it.ldx ? it.ldy : it.ldx) < (1ul << 32) && ((uintptr_t)it.dY % 16) == 0 &&
         ((uintptr_t)it.X % 16) == 0;
}
static void tn_group_plan(int tiles, int nk, int G, int& full, int& sk_wgs, long& R) {
  full = tiles / G * G;
  // a last round that is at least 90 % full runs as whole tiles too: cutting it would even out the last 10 % of one round at the price
  // of two partial planes + a fix-up pass for every tile of it (the fusion tower: 1512 tiles = 5 rounds + 232)
  if ((tiles - full) * 10 >= G * 9) full = tiles;
  R = (long)(tiles - full) * nk;
  sk_wgs = 0;
  if (R > 0) {
    const long m = R / 16;   // >= 16 K-steps per workgroup: pieces stay >= 4 steps after the boundaries snap to tile edges
    sk_wgs = (int)(m < G ? m : G);
    if (sk_wgs < 1) sk_wgs = 1;
  }
}
#define TN_PROG_BYTES 2048   // 8 XCDs x 64 progress words (TnGroup::prog), at the END of the caller's workspace
long xfm_gemm_tn_group_workspace_impl(int n, const xfm_tn_item* items, int M) {
  if (n <= 0 || items == nullptr || M <= 0) return 0;
  const int G = tn_group_cus();
  long need = 0;
  int tiles = 0, np = 0;
  auto close = [&]() {
    if (np == 0) return;
    int full, sk;
    long R;
    tn_group_plan(tiles, cdiv(M, 64), G, full, sk, R);
    const long b = (sk > 1 ? 2l * sk * (256 * 256 + 256) * 4 : 0) + TN_PROG_BYTES;   // two partial-piece slots per sharing workgroup (tile + column sums) + the progress rows
    need = b > need ? b : need;
    tiles = np = 0;
  };
  for (int i = 0; i < n; ++i) {
    const xfm_tn_item& it = items[i];
    if (tn_group_item_ok(it, M)) {
      tiles += (it.N / 256) * (it.K / 256);
      if (++np == TN_GROUP_MAX) close();
    } else {
      const long b = xfm_gemm_tn_workspace_impl(M, it.N, it.K);
      need = b > need ? b : need;
    }
  }
  close();
  return need;
}
int xfm_gemm_tn_group_impl(int n, const xfm_tn_item* items, int M, float* workspace, long workspace_bytes, hipStream_t st) {
  XFM_REQUIRE(n >= 0 && (n == 0 || items != nullptr) && M > 0, "gemm_tn_group: bad arguments");
  XFM_REQUIRE(workspace_bytes >= xfm_gemm_tn_group_workspace_impl(n, items, M) && (workspace != nullptr || workspace_bytes == 0),
              "gemm_tn_group: workspace smaller than xfm_gemm_tn_group_workspace()");
  const int G = tn_group_cus();
  static bool attr_set = false;
  const size_t smem = 2 * 4 * 64 * 256;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_group_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_group_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  TnGroup g{};
  g.nk = cdiv(M, 64);
  g.M = M;
  g.ws = workspace;
  auto flush = [&]() -> int {
    if (g.nprob == 0) return XFM_OK;
    tn_group_plan(g.total_tiles, g.nk, G, g.full_tiles, g.sk_wgs, g.sk_iters);
    g.ws_bias = g.ws != nullptr ? g.ws + 2l * g.sk_wgs * 256 * 256 : nullptr;
    const int grid = g.full_tiles > 0 ? G : g.sk_wgs;
    // progress throttle of the data-parallel rounds (XFM_TN_SYNC_WINDOW K-steps; 0 = off): needs whole rounds, 8 equal XCD shares and
    // the progress rows at the end of the workspace
    static const int window_env = getenv("XFM_TN_SYNC_WINDOW") ? atoi(getenv("XFM_TN_SYNC_WINDOW")) : 0;
    static unsigned epoch = 0;
    g.window = 0;
    if (window_env > 0 && g.full_tiles >= grid && grid % 8 == 0 && grid / 8 <= 64 && workspace != nullptr && workspace_bytes >= TN_PROG_BYTES) {
      g.window = window_env;
      g.prog = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(workspace) + (workspace_bytes - TN_PROG_BYTES) / 16 * 16);
      g.epoch = (++epoch) & 0xFFFu;
    }
    if (g.window > 0) hipLaunchKernelGGL(gemm_tn_group_kernel<true>, dim3(grid), dim3(512), smem, st, g);
    else hipLaunchKernelGGL(gemm_tn_group_kernel<false>, dim3(grid), dim3(512), smem, st, g);
    int rc = xfm_check_launch("gemm_tn_group");
    if (rc == XFM_OK && g.sk_wgs > 1) {
      hipLaunchKernelGGL(tn_group_fixup_kernel, dim3(64, g.total_tiles - g.full_tiles), dim3(256), 0, st, g);
      rc = xfm_check_launch("gemm_tn_group_fixup");
    }
    g.nprob = g.total_tiles = 0;
    return rc;
  };
  for (int i = 0; i < n; ++i) {
    const xfm_tn_item& it = items[i];
    XFM_REQUIRE(it.dY && it.X && it.dW && it.N > 0 && it.K > 0, "gemm_tn_group: bad problem %d", i);
    if (!tn_group_item_ok(it, M)) continue;
    TnGroupProb& P = g.p[g.nprob++];
    P.dY = (const bf16*)it.dY; P.X = (const bf16*)it.X; P.dW = it.dW; P.dbias = it.dbias;
    P.ldy = (unsigned)it.ldy; P.ldx = (unsigned)it.ldx; P.ldw = it.ldw;
    P.tiles_k = it.K / 256;
    g.total_tiles += (it.N / 256) * (it.K / 256);
    P.tile_end = g.total_tiles;
    if (g.nprob == TN_GROUP_MAX) {
      const int rc = flush();
      if (rc != XFM_OK) return rc;
    }
  }
  int rc = flush();
  if (rc != XFM_OK) return rc;
  for (int i = 0; i < n; ++i) {   // what the grouped kernel did not take
    const xfm_tn_item& it = items[i];
    if (!tn_group_item_ok(it, M))
      rc = xfm_gemm_tn_impl(it.dY, it.ldy, it.X, it.ldx, it.dW, it.ldw, it.dbias, M, it.N, it.K, 0, workspace, workspace_bytes, st);
    if (rc != XFM_OK) return rc;
  }
  return XFM_OK;
}

// ---------------------------------------------------------------------------------------------
// fp32 master weight [N,K] -> bf16 copy [N,Kp] and transposed bf16 copy [K,Np] (zero padded to ld multiples)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ w, int N, int K, bf16* __restrict__ wb,
                                                             long ldb, bf16* __restrict__ wt, long ldt) {
  __shared__ float tile[32][33];
  const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ty + i * 8, k = k0 + tx;
    const float v = (n < N && k < K) ? w[(long)n * K + k] : 0.f;
    tile[ty + i * 8][tx] = v;
    if (wb != nullptr && n < N && k < ldb) wb[(long)n * ldb + k] = f2bf(v);
  }
  __syncthreads();
  if (wt != nullptr) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + ty + i * 8, n = n0 + tx;
      if (k < K && n < ldt) wt[(long)k * ldt + n] = f2bf(tile[tx][ty + i * 8]);
    }
  }
}

// The same for a whole table of weights in ONE launch (a training step refreshes ~200 operands after the optimiser; at 7 us
// a launch that is 1.3 ms of 32x32-tile kernels that each fill a fraction of the chip).  64x64 tiles; workgroup -> item by
// binary search over the items' first tile.
__global__ __launch_bounds__(256) void cast_transpose_batch_kernel(const xfm_cast_item* __restrict__ items, int n_items) {
  __shared__ float tile[64][65];
  const long t = blockIdx.x;
  int lo = 0, hi = n_items - 1;
  while (lo < hi) {  // last item whose tile_start <= t
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].tile_start <= t) lo = mid; else hi = mid - 1;
  }
  const xfm_cast_item it = items[lo];
  const long local = t - it.tile_start;
  const int n0 = (int)(local / it.tiles_x) * 64, k0 = (int)(local % it.tiles_x) * 64;
  const int N = it.N, K = it.K;
  const float* __restrict__ w = it.w;
  bf16* __restrict__ wb = reinterpret_cast<bf16*>(it.wb);
  bf16* __restrict__ wt = reinterpret_cast<bf16*>(it.wt);
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 x 16: a thread owns 4 consecutive elements of a row
  // fast path: a full interior tile with 16-B / 8-B aligned rows -> float4 loads, 8-B bf16x4 stores (128-B row segments both ways)
  const bool fast = n0 + 64 <= N && k0 + 64 <= K && (K & 3) == 0 && ((uintptr_t)w & 15) == 0 &&
                    (wb == nullptr || ((it.ldb & 3) == 0 && ((uintptr_t)wb & 7) == 0)) &&
                    (wt == nullptr || ((it.ldt & 3) == 0 && ((uintptr_t)wt & 7) == 0));
  if (fast) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = ty + i * 16;
      const f32x4 v = *reinterpret_cast<const f32x4*>(w + (long)(n0 + n) * K + k0 + 4 * tx);
      tile[n][4 * tx] = v[0]; tile[n][4 * tx + 1] = v[1]; tile[n][4 * tx + 2] = v[2]; tile[n][4 * tx + 3] = v[3];
      if (wb != nullptr) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = f2bf(v[j]);
        *reinterpret_cast<bf16x4*>(wb + (long)(n0 + n) * it.ldb + k0 + 4 * tx) = o;
      }
    }
    __syncthreads();
    if (wt != nullptr) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = ty + i * 16;
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = f2bf(tile[4 * tx + j][k]);
        *reinterpret_cast<bf16x4*>(wt + (long)(k0 + k) * it.ldt + n0 + 4 * tx) = o;
      }
    }
    return;
  }
  // edge tiles (and the zero padding out to ldb / ldt): element-wise
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int n = n0 + (e >> 6), k = k0 + (e & 63);
    const float v = (n < N && k < K) ? w[(long)n * K + k] : 0.f;
    tile[e >> 6][e & 63] = v;
    if (wb != nullptr && n < N && k < it.ldb) wb[(long)n * it.ldb + k] = f2bf(v);
  }
  __syncthreads();
  if (wt != nullptr) {
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
      const int k = k0 + (e >> 6), n = n0 + (e & 63);
      if (k < K && n < it.ldt) wt[(long)k * it.ldt + n] = f2bf(tile[e & 63][e >> 6]);
    }
  }
}

int xfm_cast_transpose_batch_impl(const xfm_cast_item* items, int n_items, long total_tiles, hipStream_t st) {
  XFM_REQUIRE(items != nullptr && n_items > 0 && total_tiles > 0 && total_tiles < (1L << 31), "cast_transpose_batch: bad table");
  hipLaunchKernelGGL(cast_transpose_batch_kernel, dim3((unsigned)total_tiles), dim3(256), 0, st, items, n_items);
  return xfm_check_launch("cast_transpose_batch");
}

int xfm_cast_transpose_impl(const float* w, int N, int K, void* wb, long ldb, void* wt, long ldt, hipStream_t st) {
  XFM_REQUIRE(N > 0 && K > 0, "cast_transpose: empty");
  XFM_REQUIRE(wb == nullptr || ldb >= K, "cast_transpose: ldb < K");
  XFM_REQUIRE(wt == nullptr || ldt >= N, "cast_transpose: ldt < N");
  // grid covers the padded extents so the zero padding is (re)written too
  const long kext = (wb != nullptr && ldb > K) ? ldb : K, next = (wt != nullptr && ldt > N) ? ldt : N;
  hipLaunchKernelGGL(cast_transpose_kernel, dim3(cdiv(kext, 32), cdiv(next, 32)), dim3(256), 0, st, w, N, K, (bf16*)wb, ldb,
                     (bf16*)wt, ldt);
  return xfm_check_launch("cast_transpose");
}


int xfm_gemm_nt_plan_impl(int M, int N, int K, int epi, int tile_hint, int* cfg, int* rows_a) {
  XFM_REQUIRE(M > 0 && N > 0 && K > 0 && cfg != nullptr && rows_a != nullptr, "gemm_nt_plan: bad arguments");
  int ks = 1;
  *cfg = nt_plan(M, N, K, epi, tile_hint, rows_a, &ks);
  return XFM_OK;
}
