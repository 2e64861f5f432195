// Contrastive / matching glue of XFMBase as a handful of small kernels (xfm.py:614-621 get_features, :683-715 get_contrastive_loss,
// :717-746 get_hard_negatives).  The reference (and round 1 of this build) ran them as a few dozen ATen launches per step: a [B,256] x
// [256,B] matmul through the vendor BLAS, softmax / log_softmax / nll_loss kernels, fills and a multinomial.  The arithmetic is a few
// MFLOP; what it costs is launches on the step's critical path between the towers and the fusion encoder.  fp32 throughout.
#include "common.h"

// one workgroup = one row: dot products of the row's vector with all N vectors of the other side into LDS, 4 waves striding the
// columns, each lane holding E/64 elements
template <int EPL>
__device__ __forceinline__ void row_logits(const float* __restrict__ own, const float* __restrict__ other, int N, int E, float inv_temp,
                                           float* __restrict__ lds_logits) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float a[EPL];
#pragma unroll
  for (int k = 0; k < EPL; ++k) a[k] = own[k * 64 + lane];
  for (int j = w; j < N; j += 4) {
    const float* o = other + (long)j * E;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < EPL; ++k) t = fmaf(a[k], o[k * 64 + lane], t);
    t = wave_sum(t);
    if (lane == 0) lds_logits[j] = t * inv_temp;
  }
  __syncthreads();
}

__device__ __forceinline__ float block_max256(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float block_add256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// the logits row lives in dynamic LDS (N floats + 4 reduction slots): 16000 rows = 62.5 KiB, inside the default 64 KiB dynamic limit.
// The gathered row count is world_size * B (the reference pre-trains at 128 x 24 = 3072).
#define XFM_LOSS_MAXN 16000

// blocks [0, N): rows of logits (image i against every text); blocks [N, 2N): rows of logits^T.  loss_sum += (lse - logit[r, r]) / (2N)
// idx != NULL (retrieval fine-tuning, xfm.py:705-713): soft labels -- every row j with idx[j] == idx[r] is a positive of row r with
// weight 1 / cnt_r (cnt_r = the number of such rows, written to cnt[r]): loss_sum += (lse - mean of the positives' logits) / (2N)
template <int EPL>
__global__ __launch_bounds__(256) void itc_fwd_kernel(const float* __restrict__ I, const float* __restrict__ T, const float* __restrict__ temp,
                                                       int N, int E, float* __restrict__ lse, float* __restrict__ loss_sum,
                                                       const int64_t* __restrict__ idx, float* __restrict__ cnt) {
  extern __shared__ float lg[];
  float* red = lg + N;
  const int r = blockIdx.x < N ? blockIdx.x : blockIdx.x - N;
  const bool rows = blockIdx.x < N;
  const float inv_temp = 1.0f / temp[0];
  row_logits<EPL>((rows ? I : T) + (long)r * E, rows ? T : I, N, E, inv_temp, lg);
  float mx = -3.0e38f;
  for (int j = threadIdx.x; j < N; j += 256) mx = fmaxf(mx, lg[j]);
  mx = block_max256(mx, red);
  float s = 0.f;
  for (int j = threadIdx.x; j < N; j += 256) s += __expf(lg[j] - mx);
  s = block_add256(s, red);
  float pos = 0.f, npos = 0.f;
  if (idx != nullptr) {
    const int64_t mine = idx[r];
    for (int j = threadIdx.x; j < N; j += 256)
      if (idx[j] == mine) { pos += lg[j]; npos += 1.f; }
    pos = block_add256(pos, red);
    npos = block_add256(npos, red);
  }
  // The loss is the sum of the 2N row terms in a FIXED order (a float atomicAdd per block made two runs of the same step differ in the
  // last bit of loss_itc): every block parks its term behind the statistics (lse[2N + block]); the block that takes the last ticket
  // (loss_sum[1], an integer counter the caller zeroed with loss_sum[0]) adds them up with a fixed tree.
  __shared__ int last;
  if (threadIdx.x == 0) {
    const float l = mx + __logf(s);
    lse[blockIdx.x] = l;
    if (idx != nullptr && rows) cnt[r] = npos;
    lse[2 * N + blockIdx.x] = (idx != nullptr ? (l - pos / npos) : (l - lg[r])) / (2.0f * N);
    __threadfence();
    last = atomicAdd(reinterpret_cast<int*>(loss_sum) + 1, 1) == 2 * N - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  float t = 0.f;
  for (int j = threadIdx.x; j < 2 * N; j += 256) t += __builtin_nontemporal_load(lse + 2 * N + j);
  t = block_add256(t, red);
  if (threadIdx.x == 0) loss_sum[0] += t;
}

// blocks [0, N): dI_i and the temperature gradient; blocks [N, 2N): dT_j.  With L = logits, G_ij = softmax_row(L)_ij +
// softmax_col(L)_ij - 2 delta_ij:  dL = G * g / (2N);  dI = dL . T / temp;  dT = dL^T . I / temp;  dtemp = - sum dL_ij L_ij / temp
template <int EPL>
__global__ __launch_bounds__(256) void itc_bwd_kernel(const float* __restrict__ I, const float* __restrict__ T, const float* __restrict__ temp,
                                                       const float* __restrict__ lse, const float* __restrict__ g, int N, int E,
                                                       float* __restrict__ dI, float* __restrict__ dT, float* __restrict__ dtemp,
                                                       const int64_t* __restrict__ idx, const float* __restrict__ cnt) {
  extern __shared__ float lg[];
  float* red = lg + N;
  const int r = blockIdx.x < N ? blockIdx.x : blockIdx.x - N;
  const bool rows = blockIdx.x < N;
  const float tv = temp[0], inv_temp = 1.0f / tv, gs = g[0] / (2.0f * N);
  const float* other = rows ? T : I;
  row_logits<EPL>((rows ? I : T) + (long)r * E, other, N, E, inv_temp, lg);
  const float* lse_own = lse + (rows ? 0 : N);     // statistics of this block's own direction ...
  const float* lse_oth = lse + (rows ? N : 0);     // ... and of the other direction, indexed by the column
  float tpart = 0.f;
  const float own = lse_own[r];
  // soft labels (idx): the positives of (r, j) weigh 1 / cnt_r in the row direction and 1 / cnt_j in the column direction
  const int64_t mine = idx != nullptr ? idx[r] : 0;
  const float inv_cnt_r = idx != nullptr ? 1.0f / cnt[r] : 0.f;
  for (int j = threadIdx.x; j < N; j += 256) {
    const float L = lg[j];
    float lab;
    if (idx != nullptr) lab = idx[j] == mine ? inv_cnt_r + 1.0f / cnt[j] : 0.f;
    else lab = j == r ? 2.0f : 0.0f;
    const float G = __expf(L - own) + __expf(L - lse_oth[j]) - lab;
    tpart += G * L;
    lg[j] = G * gs;  // dL of (r, j) in place
  }
  __syncthreads();
  if (rows) {
    tpart = block_add256(tpart, red);
    // (the row's share of the temperature gradient is parked behind the statistics -- lse[2N + r], the forward's dead row terms -- and
    // itc_dtemp_kernel adds the N shares in a fixed order: a float atomic per row made `temp`'s gradient differ from run to run)
    if (threadIdx.x == 0) const_cast<float*>(lse)[2 * N + r] = -tpart * gs * inv_temp;
  }
  float* out = (rows ? dI : dT) + (long)r * E;
  for (int e = threadIdx.x; e < E; e += 256) {
    float acc = 0.f;
    for (int j = 0; j < N; ++j) acc = fmaf(lg[j], other[(long)j * E + e], acc);
    out[e] = acc * inv_temp;
  }
}

// blocks [0, B): image i draws its negative text; blocks [B, 2B): text j draws its negative image (xfm.py:727-744)
template <int EPL>
__global__ __launch_bounds__(256) void hard_neg_kernel(const float* __restrict__ I, const float* __restrict__ T, const float* __restrict__ temp,
                                                        int B, int E, uint32_t seed_lo, uint32_t seed_hi, int64_t* __restrict__ image_neg,
                                                        int64_t* __restrict__ text_neg, const int64_t* __restrict__ idx) {
  extern __shared__ float lg[];
  float* red = lg + B;
  const int r = blockIdx.x < B ? blockIdx.x : blockIdx.x - B;
  const bool rows = blockIdx.x < B;
  row_logits<EPL>((rows ? I : T) + (long)r * E, rows ? T : I, B, E, 1.0f / temp[0], lg);
  float mx = -3.0e38f;
  for (int j = threadIdx.x; j < B; j += 256) mx = fmaxf(mx, lg[j]);
  mx = block_max256(mx, red);
  float s = 0.f;
  for (int j = threadIdx.x; j < B; j += 256) s += __expf(lg[j] - mx);
  s = block_add256(s, red);
  const float inv = 1.0f / s;
  float wsum = 0.f;
  for (int j = threadIdx.x; j < B; j += 256) {
    // softmax + 1e-5, own entry zeroed -- with idx (xfm.py:731-734) every entry of the same image id
    const bool same = idx != nullptr ? idx[j] == idx[r] : j == r;
    const float wj = same ? 0.f : __expf(lg[j] - mx) * inv + 1e-5f;
    lg[j] = wj;
    wsum += wj;
  }
  wsum = block_add256(wsum, red);
  if (threadIdx.x == 0) {  // one categorical draw by the inverse CDF (B <= a few hundred: a serial scan)
    const uint32_t key = rng_row_key(seed_lo, seed_hi, blockIdx.x);
    const float u = (float)(rng_u32(key, 0u) >> 8) * (1.0f / 16777216.0f) * wsum;
    float c = 0.f;
    int pick = -1, last = -1;
    for (int j = 0; j < B; ++j) {
      if (lg[j] <= 0.f) continue;
      last = j;
      c += lg[j];
      if (u < c) { pick = j; break; }
    }
    if (pick < 0) pick = last < 0 ? (r + 1) % B : last;  // rounding at the top of the CDF
    (rows ? text_neg : image_neg)[r] = pick;
  }
}

template <int EPL>
__global__ __launch_bounds__(256) void rownorm_fwd_kernel(const float* __restrict__ x, int R, int E, float* __restrict__ y, float* __restrict__ inv) {
  const int lane = threadIdx.x & 63, row = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= R) return;
  float v[EPL], s = 0.f;
#pragma unroll
  for (int k = 0; k < EPL; ++k) { v[k] = x[(long)row * E + k * 64 + lane]; s = fmaf(v[k], v[k], s); }
  const float n = fmaxf(sqrtf(wave_sum(s)), 1e-12f), iv = 1.0f / n;
#pragma unroll
  for (int k = 0; k < EPL; ++k) y[(long)row * E + k * 64 + lane] = v[k] * iv;
  if (lane == 0) inv[row] = iv;
}
template <int EPL>
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ inv,
                                                           int R, int E, float* __restrict__ dx) {
  const int lane = threadIdx.x & 63, row = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= R) return;
  float a[EPL], b[EPL], s = 0.f;
#pragma unroll
  for (int k = 0; k < EPL; ++k) { a[k] = dy[(long)row * E + k * 64 + lane]; b[k] = y[(long)row * E + k * 64 + lane]; s = fmaf(a[k], b[k], s); }
  s = wave_sum(s);
  const float iv = inv[row];
#pragma unroll
  for (int k = 0; k < EPL; ++k) dx[(long)row * E + k * 64 + lane] = (a[k] - b[k] * s) * iv;
}

#define XFM_EPL_DISPATCH(E, CALL)                                                        \
  switch ((E) / 64) {                                                                    \
    case 1: { constexpr int EPL = 1; CALL; break; }                                      \
    case 2: { constexpr int EPL = 2; CALL; break; }                                      \
    case 4: { constexpr int EPL = 4; CALL; break; }                                      \
    case 8: { constexpr int EPL = 8; CALL; break; }                                      \
    case 12: { constexpr int EPL = 12; CALL; break; }                                    \
    case 16: { constexpr int EPL = 16; CALL; break; }                                    \
    default: xfm_set_error("loss kernels: feature width %d not in {64,128,256,512,768,1024}", (E)); return XFM_E_UNSUPPORTED; \
  }

static int loss_check(int N, int E) {
  XFM_REQUIRE(N > 0 && N <= XFM_LOSS_MAXN && E > 0 && E % 64 == 0, "loss kernels: need 0 < rows <= %d and a width that is a multiple of 64 (got %d, %d)", XFM_LOSS_MAXN, N, E);
  return XFM_OK;
}

int xfm_rownorm_fwd_impl(const float* x, int R, int E, float* y, float* inv, hipStream_t st) {
  XFM_REQUIRE(R > 0 && E % 64 == 0, "rownorm: bad shape");
  XFM_EPL_DISPATCH(E, hipLaunchKernelGGL((rownorm_fwd_kernel<EPL>), dim3(cdiv(R, 4)), dim3(256), 0, st, x, R, E, y, inv));
  return xfm_check_launch("rownorm_fwd");
}
int xfm_rownorm_bwd_impl(const float* dy, const float* y, const float* inv, int R, int E, float* dx, hipStream_t st) {
  XFM_REQUIRE(R > 0 && E % 64 == 0, "rownorm: bad shape");
  XFM_EPL_DISPATCH(E, hipLaunchKernelGGL((rownorm_bwd_kernel<EPL>), dim3(cdiv(R, 4)), dim3(256), 0, st, dy, y, inv, R, E, dx));
  return xfm_check_launch("rownorm_bwd");
}
int xfm_itc_fwd_impl(const float* I, const float* T, const float* temp, int N, int E, float* lse, float* loss_sum, const int64_t* idx,
                     float* cnt, hipStream_t st) {
  int rc = loss_check(N, E);
  if (rc != XFM_OK) return rc;
  XFM_REQUIRE(idx == nullptr || cnt != nullptr, "itc_fwd: idx needs the cnt output");
  XFM_EPL_DISPATCH(E, hipLaunchKernelGGL((itc_fwd_kernel<EPL>), dim3(2 * N), dim3(256), (N + 4) * sizeof(float), st, I, T, temp, N, E, lse, loss_sum, idx, cnt));
  return xfm_check_launch("itc_fwd");
}
__global__ __launch_bounds__(256) void itc_dtemp_kernel(const float* __restrict__ share, int N, float* __restrict__ dtemp) {
  __shared__ float red[4];
  float t = 0.f;
  for (int j = threadIdx.x; j < N; j += 256) t += share[j];
  t = block_add256(t, red);
  if (threadIdx.x == 0) dtemp[0] += t;
}

int xfm_itc_bwd_impl(const float* I, const float* T, const float* temp, const float* lse, const float* g, int N, int E, float* dI,
                     float* dT, float* dtemp, const int64_t* idx, const float* cnt, hipStream_t st) {
  int rc = loss_check(N, E);
  if (rc != XFM_OK) return rc;
  XFM_REQUIRE(idx == nullptr || cnt != nullptr, "itc_bwd: idx needs the cnt the forward wrote");
  XFM_EPL_DISPATCH(E, hipLaunchKernelGGL((itc_bwd_kernel<EPL>), dim3(2 * N), dim3(256), (N + 4) * sizeof(float), st, I, T, temp, lse, g, N, E, dI, dT, dtemp, idx, cnt));
  rc = xfm_check_launch("itc_bwd");
  if (rc != XFM_OK) return rc;
  hipLaunchKernelGGL(itc_dtemp_kernel, dim3(1), dim3(256), 0, st, lse + 2 * N, N, dtemp);
  return xfm_check_launch("itc_dtemp");
}
int xfm_hard_negatives_impl(const float* I, const float* T, const float* temp, int B, int E, uint64_t seed, int64_t* image_neg,
                            int64_t* text_neg, const int64_t* idx, hipStream_t st) {
  int rc = loss_check(B, E);
  if (rc != XFM_OK) return rc;
  XFM_REQUIRE(B >= 2, "hard_negatives: a batch of one has no negative");
  XFM_EPL_DISPATCH(E, hipLaunchKernelGGL((hard_neg_kernel<EPL>), dim3(2 * B), dim3(256), (B + 4) * sizeof(float), st, I, T, temp, B, E, (uint32_t)(seed & 0xFFFFFFFFu),
                                         (uint32_t)(seed >> 32), image_neg, text_neg, idx));
  return xfm_check_launch("hard_negatives");
}
