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

struct GemmNT {
  const bf16* A; long lda;
  const bf16* B; long ldb;
  void* C; long ldc;
  const float* bias;
  bf16* aux; long ldaux;
  int M, N, K;
  int group_m;  // row-panels per tile group (L2 locality of the block order)
};

// LDS swizzles (16-B chunk index XOR) for 128-B tile rows read with ds_read_b128.
// X tile: a 16-lane group reads 16 consecutive rows; W tile: rows {0-3,8-11,16-19,24-27}(+4) (see header comment).
__device__ __forceinline__ int swz_x(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ int swz_w(int r) { return ((r >> 1) & 1) | (((r >> 3) & 3) << 1); }

// Epilogue of one wave's (MT*16) x (NT*16) sub-tile: lane (lg, lr) owns row m_base + mt*16 + lr and the 8 consecutive
// columns n_base + np*32 + 8*lg .. +7 of every (mt, np): bias / GELU in registers, one 16-B store per (mt, np).
template <int MT, int NT, int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmNT& g, f32x4 (&acc)[MT][NT], int m_base, int n_base, int lr, int lg) {
  const bool vec_c = (g.ldc % 8) == 0;
#pragma unroll
  for (int np = 0; np < NT / 2; ++np) {
    const int nb = n_base + np * 32 + 8 * lg;
    float bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bv[i] = (g.bias != nullptr && nb + i < g.N) ? g.bias[nb + i] : 0.f;
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
          bf16* ap = g.aux + (long)m * g.ldaux + nb;
          bf16x8 pre;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            pre[i] = f2bf(v[i]);
            o[i] = f2bf(gelu_f(bf2f(pre[i])));  // GELU of the bf16-rounded pre-activation: bwd recomputes from `aux`
          }
          if (full) *reinterpret_cast<bf16x8*>(ap) = pre;
          else
            for (int i = 0; i < 8; ++i)
              if (nb + i < g.N) ap[i] = pre[i];
        } else if (EPI == EPI_DGELU) {
          const bf16* ap = g.aux + (long)m * g.ldaux + nb;
          if (full) {
            const bf16x8 pre = *reinterpret_cast<const bf16x8*>(ap);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i] * gelu_grad_f(bf2f(pre[i])));
          } else {
            for (int i = 0; i < 8; ++i) o[i] = (nb + i < g.N) ? f2bf(v[i] * gelu_grad_f(bf2f(ap[i]))) : f2bf(0.f);
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


template <int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNT g) {
  constexpr int MT = BM / 32, NT = BN / 32;
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

  auto stage = [&](int buf, int kt) {
    char* sA = smem + buf * STAGE;
    char* sB = sA + A_BYTES;
    const int k0 = kt * 64;
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

  const int nk = g.K / 64;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
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

template <int BM, int BN>
static int launch_nt(const GemmNT& g, int epi, hipStream_t st) {
  const int tiles = cdiv(g.M, BM) * cdiv(g.N, BN);
  const size_t smem = 2 * (BM + BN) * 128;
#define XFM_NT_CASE(E)                                                                                         \
  case E: {                                                                                                    \
    static bool attr_set = false;                                                                              \
    if (!attr_set) {                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<BM, BN, E>),                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                        \
      attr_set = true;                                                                                         \
    }                                                                                                          \
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, E>), dim3(tiles), dim3(256), smem, st, g);                      \
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

int xfm_gemm_nt_impl(const void* A, long lda, const void* B, long ldb, void* C, long ldc, const float* bias,
                     void* aux, long ldaux, int M, int N, int K, int epi, int tile_hint, hipStream_t st) {
  XFM_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_nt: empty problem M=%d N=%d K=%d", M, N, K);
  XFM_REQUIRE(K % 64 == 0, "gemm_nt: K=%d must be a multiple of 64", K);
  XFM_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, "gemm_nt: lda=%ld ldb=%ld must be multiples of 8", lda, ldb);
  XFM_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0,
              "gemm_nt: operands must be 16-byte aligned");
  XFM_REQUIRE((epi != EPI_GELU && epi != EPI_DGELU) || aux != nullptr, "gemm_nt: epilogue %d needs aux", epi);
  static const int gm_env = getenv("XFM_GEMM_GROUP_M") ? atoi(getenv("XFM_GEMM_GROUP_M")) : 0;  // tuning knob
  GemmNT g{(const bf16*)A, lda, (const bf16*)B, ldb, C, ldc, bias, (bf16*)aux, ldaux, M, N, K, gm_env > 0 ? gm_env : 8};
  int cfg = tile_hint;
  if (cfg <= 0) {  // measured on MI355X (tools/tune_gemm.py): 128x128 pays from ~3 workgroups per CU, else go smaller
    if ((long)cdiv(M, 256) * cdiv(N, 128) >= 768) cfg = 4;  // >= 3 rounds of 256x128 tiles: the 3-slot ring wins on cold operands
    else if ((long)cdiv(M, 128) * cdiv(N, 128) >= 800) cfg = 1;
    else if ((long)cdiv(M, 64) * cdiv(N, 128) >= 256) cfg = 2;
    else cfg = 3;
  }
  switch (cfg) {
    case 1: return launch_nt<128, 128>(g, epi, st);
    case 2: return launch_nt<64, 128>(g, epi, st);
    case 4: return launch_nt_ring(g, epi, st);
    default: return launch_nt<64, 64>(g, epi, st);
  }
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
  // D[i = n slot][j = k col]: lane (lg, lr) holds k = ..+lr and n = ..+4*lg+reg
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int k = k0 + wk * 64 + kt * 16 + lr;
#pragma unroll
      for (int rgi = 0; rgi < 4; ++rgi) {
        const int n = n0 + wn * 64 + nt * 16 + 4 * lg + rgi;
        if (n < g.N && k < g.K) atomicAdd(g.dW + (long)n * g.ldw + k, acc[nt][kt][rgi]);
      }
    }
}

// ---------------------------------------------------------------------------------------------
// Large-M wgrad: 256 (n) x 128 (k) output tile, 8 waves, one workgroup per CU, 3-slot LDS ring (dY 64 x 256 + X 64 x 128
// per slot = 48 KiB) filled by direct-to-LDS loads that stay in flight across the per-step barrier (counted vmcnt(6)).
// The bias gradient rides along as MFMAs against an all-ones B fragment (sum_m dY[m,n] = dY^T . 1) in the k-tile-0
// workgroups -- no extra pass over dY and no LDS re-reads.
// ---------------------------------------------------------------------------------------------
template <int ROW_BYTES>
__device__ __forceinline__ bf16x8 tr_read_pair_rs(const char* tile, int row0, int col0, int lr) {
  const int r = row0 + (lr >> 2);
  const int col = col0 + 4 * (lr & 3);
  const int off0 = r * ROW_BYTES + ((((col >> 3)) ^ swz_t(r)) << 4) + (col & 7) * 2;
  const int r2 = r + 4;
  const int off1 = r2 * ROW_BYTES + ((((col >> 3)) ^ swz_t(r2)) << 4) + (col & 7) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + off0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + off1));
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo;
  u.s.b = hi;
  return u.v;
}

__global__ __launch_bounds__(512) void gemm_tn_ring_kernel(GemmTN g) {
  constexpr int Y_BYTES = 64 * 512, X_BYTES = 64 * 256, STAGE = Y_BYTES + X_BYTES;  // 48 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wn = w >> 1, wk = w & 1;
  const int lr = lane & 15, lg = lane >> 4;
  const int tiles_k = (g.K + 127) / 128, tiles_n = (g.N + 255) / 256;
  const int per_split = tiles_k * tiles_n;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int split = wg / per_split, t = wg % per_split;
  const int n0 = (t / tiles_k) * 256, k0 = (t % tiles_k) * 128;
  const int mbeg = split * g.m_per_split;
  int mend = mbeg + g.m_per_split;
  mend = mend < g.M ? mend : g.M;
  const int nsteps = (mend - mbeg + 63) / 64;
  const bool do_bias = g.dbias != nullptr && k0 == 0 && wk == 0;

  auto stage = [&](int slot, int step) {  // 32 wave-instructions for dY (2 rows each) + 16 for X (4 rows each): 6 per wave
    char* sY = smem + slot * STAGE;
    char* sX = sY + Y_BYTES;
    const int mb = mbeg + step * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int blk = i * 8 + w;
      const int r = blk * 2 + (lane >> 5);
      const int c = (lane & 31) ^ swz_t(r);
      int m = mb + r;
      m = m < mend ? m : mend - 1;                 // tail rows are zeroed in LDS after they land (see below)
      int nn = n0 + c * 8;
      nn = nn < g.N ? nn : 0;                      // columns past N only feed masked outputs; keep the address in range
      __builtin_amdgcn_global_load_lds(GLB_PTR(void, g.dY + (long)m * g.ldy + nn), LDS_PTR(void, sY + blk * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int blk = i * 8 + w;
      const int r = blk * 4 + (lane >> 4);
      const int c = (lane & 15) ^ swz_t(r);
      int m = mb + r;
      m = m < mend ? m : mend - 1;
      int kk = k0 + c * 8;
      kk = kk < g.K ? kk : 0;
      __builtin_amdgcn_global_load_lds(GLB_PTR(void, g.X + (long)m * g.ldx + kk), LDS_PTR(void, sX + blk * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[4][4], bacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = f2bf(1.0f);

  if (nsteps > 0) stage(0, 0);
  if (nsteps > 1) stage(1, 1);
  int slot = 0;
  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + 2 < nsteps) stage(slot == 0 ? 2 : slot - 1, s + 2);
    char* sY = smem + slot * STAGE;
    char* sX = sY + Y_BYTES;
    const int valid = mend - (mbeg + s * 64);  // rows of this step that exist
    if (valid < 64) {                            // ragged last step: zero the duplicated rows of the dY tile
      for (int q = tid; q < (64 - valid) * 32; q += 512) {
        const int r = valid + (q >> 5), c = q & 31;
        *reinterpret_cast<u32x4*>(sY + r * 512 + (c << 4)) = u32x4{0, 0, 0, 0};
      }
      __syncthreads();
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) af[nt] = tr_read_pair_rs<512>(sY, ks * 32 + 8 * lg, wn * 64 + nt * 16, lr);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) bfr[kt] = tr_read_pair_rs<256>(sX, ks * 32 + 8 * lg, wk * 64 + kt * 16, lr);
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
    slot = slot == 2 ? 0 : slot + 1;
  }

#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int k = k0 + wk * 64 + kt * 16 + lr;
#pragma unroll
      for (int rgi = 0; rgi < 4; ++rgi) {
        const int n = n0 + wn * 64 + nt * 16 + 4 * lg + rgi;
        if (n < g.N && k < g.K) atomicAdd(g.dW + (long)n * g.ldw + k, acc[nt][kt][rgi]);
      }
    }
    if (do_bias && lr == 0) {  // every column of the ones-product holds the row sum: take column 0
#pragma unroll
      for (int rgi = 0; rgi < 4; ++rgi) {
        const int n = n0 + wn * 64 + nt * 16 + 4 * lg + rgi;
        if (n < g.N) atomicAdd(g.dbias + n, bacc[nt][rgi]);
      }
    }
  }
}

int xfm_gemm_tn_impl(const void* dY, long ldy, const void* X, long ldx, float* dW, long ldw, float* dbias, int M, int N, int K,
                     int splits_hint, hipStream_t st) {
  XFM_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_tn: empty problem M=%d N=%d K=%d", M, N, K);
  XFM_REQUIRE(K % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "gemm_tn: K=%d ldx=%ld ldy=%ld must be multiples of 8", K, ldx, ldy);
  XFM_REQUIRE(((uintptr_t)dY % 16) == 0 && ((uintptr_t)X % 16) == 0, "gemm_tn: operands must be 16-byte aligned");
  // splits_hint == -2 selects the experimental 256 x 128 ring variant (one workgroup per CU).  Measured on MI355X it is
  // 10-25 % SLOWER than two co-resident 128 x 128 workgroups per CU: with a single round of lock-stepped workgroups the fp32
  // atomic epilogue (128 KiB per workgroup) is fully exposed instead of overlapping the neighbour's MFMA loop.  Kept for
  // tuning; the default is the register-staged 128 x 128 kernel below.
  const int rtiles = cdiv(N, 256) * cdiv(K, 128);
  if (splits_hint == -2 && M >= 4096 && rtiles <= 128 && N >= 256) {
    int splits = 256 / rtiles;
    const int max_splits = M / 512;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int mps = cdiv(cdiv(M, splits), 64) * 64;
    splits = cdiv(M, mps);
    GemmTN g{(const bf16*)dY, ldy, (const bf16*)X, ldx, dW, ldw, dbias, M, N, K, mps};
    static bool ring_attr = false;
    const size_t rsmem = 3 * 48 * 1024;
    if (!ring_attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_ring_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)rsmem);
      ring_attr = true;
    }
    hipLaunchKernelGGL(gemm_tn_ring_kernel, dim3(rtiles * splits), dim3(512), rsmem, st, g);
    return xfm_check_launch("gemm_tn_ring");
  }
  const int tiles = cdiv(N, 128) * cdiv(K, 128);
  int splits = splits_hint < 0 ? 0 : splits_hint;
  if (splits <= 0) {
    splits = (432 + tiles / 2) / tiles;        // measured optimum: ~432 workgroups in total (tools/tune_gemm.py)
    const int max_splits = M / 480;            // ... while every split still walks >= ~8 K-steps
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
  }
  int mps = cdiv(cdiv(M, splits), 64) * 64;
  splits = cdiv(M, mps);
  GemmTN g{(const bf16*)dY, ldy, (const bf16*)X, ldx, dW, ldw, dbias, M, N, K, mps};
  static bool attr_set = false;
  const size_t smem = 4 * 64 * 256;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles * splits), dim3(256), smem, st, g);
  return xfm_check_launch("gemm_tn");
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

