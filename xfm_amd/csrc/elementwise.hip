// HBM-bound helpers of the XFM hot path (gfx950): patch gather, embedding + LayerNorm, vocabulary cross-entropy,
// flat-arena AdamW.  All vectorised 8-16 B per lane, one wavefront per row where a row reduction is needed.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// Patch gather: NCHW fp32 image -> bf16 patch matrix [B*gh*gw, C*P*P] with column order (c, ky, kx), i.e. the A
// operand of the patch-embed GEMM against Conv2d.weight.view(D, C*P*P)  (beit2.py:224-230).
// Each thread converts 8 consecutive kx of one (patch, c, ky): a 32-B coalesced read, a 16-B store.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, int B, int C, int Himg, int Wimg, int P,
                                                       bf16* __restrict__ out) {
  const int gh = Himg / P, gw = Wimg / P;
  const int kcols = C * P * P;
  const int per_row = kcols / 8;
  const long total = (long)B * gh * gw * per_row;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int c8 = (int)(t % per_row);
    const long prow = t / per_row;
    const int col = c8 * 8;
    const int c = col / (P * P), ky = (col / P) % P, kx = col % P;
    const int px = (int)(prow % gw), py = (int)((prow / gw) % gh), b = (int)(prow / ((long)gw * gh));
    const float* src = img + (((long)b * C + c) * Himg + py * P + ky) * Wimg + px * P + kx;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 4);
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = f2bf(a0[i]); o[4 + i] = f2bf(a1[i]); }
    *reinterpret_cast<bf16x8*>(out + prow * kcols + col) = o;
  }
}

int xfm_patchify_impl(const float* img, int B, int C, int H, int W, int P, void* out, hipStream_t st) {
  XFM_REQUIRE(B > 0 && C > 0 && P % 8 == 0 && H % P == 0 && W % P == 0 && W % 4 == 0, "patchify: bad geometry B=%d C=%d H=%d W=%d P=%d", B, C, H, W, P);
  const long total = (long)B * (H / P) * (W / P) * (C * P * P / 8);
  int grid = cdiv(total, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(patchify_kernel, dim3(grid), dim3(256), 0, st, img, B, C, H, W, P, (bf16*)out);
  return xfm_check_launch("patchify");
}

// ---------------------------------------------------------------------------------------------
// ViT token assembly (beit2.py:432-446): x0[b] = [cls | tok[b mod Bt] with masked patches replaced by mask_token], fp32.
// Bx = reps * Bt output rows read the SAME Bt patch-embedded images (the pre-training step runs the clean and the MIM-masked
// view of an image in one 2B pass: the patch-embed GEMM, its weight gradient and the image gather are done once per image).
// Replaces tok * (1 - w) + mask_token * w, the cls concat and their five autograd kernels.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vit_tokens_fwd_kernel(const float* __restrict__ tok, const float* __restrict__ cls,
                                                             const float* __restrict__ mask_token, const uint8_t* __restrict__ mask,
                                                             int Bt, int Bx, int P, int D, float* __restrict__ x0) {
  const int d4 = D / 4;
  const long total = (long)Bx * (P + 1) * d4;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int c = (int)(t % d4) * 4;
    const long row = t / d4;
    const int i = (int)(row % (P + 1)), b = (int)(row / (P + 1));
    const float* src;
    if (i == 0) src = cls + c;
    else if (mask != nullptr && mask[(long)b * P + i - 1]) src = mask_token + c;
    else src = tok + ((long)(b % Bt) * P + i - 1) * D + c;
    *reinterpret_cast<f32x4*>(x0 + row * D + c) = *reinterpret_cast<const f32x4*>(src);
  }
}

// dtok[s, i] = sum over the rows b = s (mod Bt) that kept patch i of dx0[b, 1 + i]   (written, not accumulated)
__global__ __launch_bounds__(256) void vit_tokens_bwd_tok_kernel(const float* __restrict__ dx0, const uint8_t* __restrict__ mask, int Bt,
                                                                 int Bx, int P, int D, float* __restrict__ dtok) {
  const int d4 = D / 4;
  const long total = (long)Bt * P * d4;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int c = (int)(t % d4) * 4;
    const long row = t / d4;
    const int i = (int)(row % P), s0 = (int)(row / P);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int b = s0; b < Bx; b += Bt) {
      if (mask != nullptr && mask[(long)b * P + i]) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(dx0 + ((long)b * (P + 1) + 1 + i) * D + c);
      acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    *reinterpret_cast<f32x4*>(dtok + row * D + c) = acc;
  }
}

// dcls += sum_b dx0[b, 0];  dmask_token += sum over masked (b, i) of dx0[b, 1 + i].  One workgroup per (batch row, chunk of the patches),
// thread = 4 columns (D <= 1024), row skips are workgroup-uniform.  The workgroups' sums are PARKED -- part[set][wg][D], the layout of
// the LayerNorm column-sum partials, in the head of dtok, which the token kernel overwrites afterwards -- and folded by reduce_sets in
// workgroup order: the 128 float atomics per element this used to end in moved the last bits of both gradients from run to run (and
// 128 workgroups walking 196 patches each took 103 us at the end of the ViT's backward chain; 4 chunks per row: a quarter of that).
__global__ __launch_bounds__(256) void vit_tokens_bwd_vec_kernel(const float* __restrict__ dx0, const uint8_t* __restrict__ mask, int Bx,
                                                                 int P, int D, int chunks, float* __restrict__ part) {
  const int c = threadIdx.x * 4;
  if (c >= D) return;
  const int np = Bx * chunks, b = blockIdx.x / chunks, ch = blockIdx.x % chunks;
  const int per = (P + chunks - 1) / chunks, i0 = ch * per, i1 = i0 + per < P ? i0 + per : P;
  const float* base = dx0 + (long)b * (P + 1) * D + c;
  f32x4 ac = {0.f, 0.f, 0.f, 0.f}, am = {0.f, 0.f, 0.f, 0.f};
  if (ch == 0) ac = *reinterpret_cast<const f32x4*>(base);
  if (mask != nullptr) {
    for (int i = i0; i < i1; ++i) {
      if (!mask[(long)b * P + i]) continue;
      const f32x4 u = *reinterpret_cast<const f32x4*>(base + (long)(1 + i) * D);
      am[0] += u[0]; am[1] += u[1]; am[2] += u[2]; am[3] += u[3];
    }
  }
  *reinterpret_cast<f32x4*>(part + ((long)0 * np + blockIdx.x) * D + c) = ac;
  *reinterpret_cast<f32x4*>(part + ((long)1 * np + blockIdx.x) * D + c) = am;
}

// ---------------------------------------------------------------------------------------------
// MIM loss (xfm.py:624-635): MSE over the masked patch rows + MSE over the pooled cls row, between the masked view's embeddings x and
// the (detached) clean view's t, both bf16 [B, N, D].  Forward: sums = {sum (x-t)^2 over masked patch rows, the same over cls rows,
// number of masked patches}; backward: dx = g * 2 (x - t) / (count * D) on masked rows, g * 2 (x - t) / (B * D) on cls rows, 0 elsewhere.
// One read of x and t each way instead of eight fp32 elementwise passes.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mim_loss_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ t,
                                                           const uint8_t* __restrict__ mask, int B, int N, int D, float* __restrict__ partial) {
  // block partials (fixed grid, fixed order inside a block) -> mim_loss_final_kernel: bit-reproducible, and no 8192 waves queueing on
  // three atomic addresses (round 3: 110 us for a 38 MB read)
  __shared__ float red[4][3];
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * 256) >> 6;
  float sp = 0.f, sc = 0.f, cnt = 0.f;
  for (long r = wave; r < (long)B * N; r += nwaves) {
    const int n = (int)(r % N), b = (int)(r / N);
    const bool cls = n == 0;
    if (!cls && !mask[(long)b * (N - 1) + n - 1]) continue;  // wave-uniform
    float s = 0.f;
    for (int c = lane * 8; c < D; c += 512) {
      const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + r * D + c);
      const bf16x8 tv = *reinterpret_cast<const bf16x8*>(t + r * D + c);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float d = bf2f(xv[i]) - bf2f(tv[i]);
        s = fmaf(d, d, s);
      }
    }
    s = wave_sum(s);
    if (cls) sc += s; else { sp += s; cnt += 1.f; }
  }
  if (lane == 0) { red[threadIdx.x >> 6][0] = sp; red[threadIdx.x >> 6][1] = sc; red[threadIdx.x >> 6][2] = cnt; }
  __syncthreads();
  if (threadIdx.x < 3) partial[blockIdx.x * 3 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(64) void mim_loss_final_kernel(const float* __restrict__ partial, int nparts, float* __restrict__ sums) {
  for (int q = 0; q < 3; ++q) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 64) s += partial[i * 3 + q];
    s = wave_sum(s);
    if (threadIdx.x == 0) sums[q] += s;
  }
}

__global__ __launch_bounds__(256) void mim_loss_bwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ t,
                                                           const uint8_t* __restrict__ mask, const float* __restrict__ sums,
                                                           const float* __restrict__ gout, int cls_term, int B, int N, int D,
                                                           bf16* __restrict__ dx) {
  const int d8 = D / 8;
  const long total = (long)B * N * d8;
  const float g = gout[0];
  const float patch_den = fmaxf(sums[2] * (float)D, 1.0f);
  const float kp = 2.0f * g / patch_den, kc = cls_term ? 2.0f * g / ((float)B * (float)D) : 0.f;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long r = e / d8;
    const int c = (int)(e % d8) * 8;
    const int n = (int)(r % N), b = (int)(r / N);
    const float k = n == 0 ? kc : (mask[(long)b * (N - 1) + n - 1] ? kp : 0.f);
    bf16x8 o;
    if (k != 0.f) {
      const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + r * D + c);
      const bf16x8 tv = *reinterpret_cast<const bf16x8*>(t + r * D + c);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = f2bf(k * (bf2f(xv[i]) - bf2f(tv[i])));
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = f2bf(0.f);
    }
    *reinterpret_cast<bf16x8*>(dx + r * D + c) = o;
  }
}

int xfm_mim_loss_fwd_impl(const void* x, const void* t, const uint8_t* mask, int B, int N, int D, float* sums, hipStream_t st) {
  XFM_REQUIRE(B > 0 && N > 1 && D > 0 && D % 8 == 0, "mim_loss: bad shape B=%d N=%d D=%d", B, N, D);
  int grid = cdiv((long)B * N, 4);
  if (grid > XFM_MIM_PARTIALS) grid = XFM_MIM_PARTIALS;
  hipLaunchKernelGGL(mim_loss_fwd_kernel, dim3(grid), dim3(256), 0, st, (const bf16*)x, (const bf16*)t, mask, B, N, D, sums + 3);
  int rc = xfm_check_launch("mim_loss_fwd");
  if (rc != XFM_OK) return rc;
  hipLaunchKernelGGL(mim_loss_final_kernel, dim3(1), dim3(64), 0, st, sums + 3, grid, sums);
  return xfm_check_launch("mim_loss_final");
}

int xfm_mim_loss_bwd_impl(const void* x, const void* t, const uint8_t* mask, const float* sums, const float* gout, int cls_term, int B,
                          int N, int D, void* dx, hipStream_t st) {
  XFM_REQUIRE(B > 0 && N > 1 && D > 0 && D % 8 == 0, "mim_loss: bad shape B=%d N=%d D=%d", B, N, D);
  int grid = cdiv((long)B * N * (D / 8), 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(mim_loss_bwd_kernel, dim3(grid), dim3(256), 0, st, (const bf16*)x, (const bf16*)t, mask, sums, gout, cls_term, B, N, D,
                     (bf16*)dx);
  return xfm_check_launch("mim_loss_bwd");
}

// ---------------------------------------------------------------------------------------------
// BEiT pooled-cls tail (beit2.py:455-466): y[b, 0, :] <- mean_i y[b, 1 + i, :] in place (bf16 rows, fp32 mean), and its backward
// dy'[b, 0] = 0, dy'[b, 1 + i] = dy[b, 1 + i] + dy[b, 0] / P.  Replaces float() / mean / cast / cat and their autograd kernels.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_rows_fwd_kernel(bf16* __restrict__ y, int N, int D) {
  __shared__ float red[2][1024];
  const int b = blockIdx.x, d8 = D / 8;
  const int cg = threadIdx.x % d8, ph = threadIdx.x / d8;  // column group (8 columns), row phase; blockDim = 2 * d8
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  bf16* base = y + (long)b * N * D + cg * 8;
  for (int i = 1 + ph; i < N; i += 2) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(base + (long)i * D);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += bf2f(v[j]);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ph][cg * 8 + j] = acc[j];
  __syncthreads();
  if (ph == 0) {
    const float inv = 1.0f / (float)(N - 1);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf((red[0][cg * 8 + j] + red[1][cg * 8 + j]) * inv);
    *reinterpret_cast<bf16x8*>(base) = o;
  }
}

__global__ __launch_bounds__(256) void pool_rows_bwd_kernel(const bf16* __restrict__ dy, int B, int N, int D, bf16* __restrict__ out) {
  const int d8 = D / 8;
  const long total = (long)B * N * d8;
  const float inv = 1.0f / (float)(N - 1);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long r = e / d8;
    const int c = (int)(e % d8) * 8;
    const int n = (int)(r % N);
    bf16x8 o;
    if (n == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f2bf(0.f);
    } else {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(dy + r * D + c);
      const bf16x8 g0 = *reinterpret_cast<const bf16x8*>(dy + (r - n) * D + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f2bf(fmaf(bf2f(g0[j]), inv, bf2f(v[j])));
    }
    *reinterpret_cast<bf16x8*>(out + r * D + c) = o;
  }
}

int xfm_pool_rows_fwd_impl(void* y, int B, int N, int D, hipStream_t st) {
  XFM_REQUIRE(B > 0 && N > 1 && D % 8 == 0 && D <= 1024 && D >= 8, "pool_rows: bad shape B=%d N=%d D=%d (D a multiple of 8, <= 1024)", B, N, D);
  hipLaunchKernelGGL(pool_rows_fwd_kernel, dim3(B), dim3(2 * (D / 8)), 0, st, (bf16*)y, N, D);
  return xfm_check_launch("pool_rows_fwd");
}

int xfm_pool_rows_bwd_impl(const void* dy, int B, int N, int D, void* out, hipStream_t st) {
  XFM_REQUIRE(B > 0 && N > 1 && D % 8 == 0, "pool_rows: bad shape B=%d N=%d D=%d", B, N, D);
  int grid = cdiv((long)B * N * (D / 8), 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(pool_rows_bwd_kernel, dim3(grid), dim3(256), 0, st, (const bf16*)dy, B, N, D, (bf16*)out);
  return xfm_check_launch("pool_rows_bwd");
}

static int vit_tokens_check(int Bt, int Bx, int P, int D) {
  XFM_REQUIRE(Bt > 0 && Bx >= Bt && Bx % Bt == 0 && P > 0 && D > 0 && D % 4 == 0 && D <= 1024,
              "vit_tokens: bad shape Bt=%d Bx=%d P=%d D=%d (Bx must be a multiple of Bt, D a multiple of 4 and <= 1024)", Bt, Bx, P, D);
  return XFM_OK;
}

int xfm_vit_tokens_fwd_impl(const float* tok, const float* cls, const float* mask_token, const uint8_t* mask, int Bt, int Bx, int P,
                            int D, float* x0, hipStream_t st) {
  int rc = vit_tokens_check(Bt, Bx, P, D);
  if (rc != XFM_OK) return rc;
  const long total = (long)Bx * (P + 1) * (D / 4);
  int grid = cdiv(total, 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(vit_tokens_fwd_kernel, dim3(grid), dim3(256), 0, st, tok, cls, mask_token, mask, Bt, Bx, P, D, x0);
  return xfm_check_launch("vit_tokens_fwd");
}

int xfm_vit_tokens_bwd_impl(const float* dx0, const uint8_t* mask, int Bt, int Bx, int P, int D, float* dtok, float* dcls,
                            float* dmask_token, hipStream_t st) {
  int rc = vit_tokens_check(Bt, Bx, P, D);
  if (rc != XFM_OK) return rc;
  const long total = (long)Bt * P * (D / 4);
  int grid = cdiv(total, 256);
  if (grid > 8192) grid = 8192;
  // the cls / mask-token sums first: their per-workgroup parts borrow the head of dtok (Bt * P * D floats), which the token kernel then
  // writes in full
  int chunks = 4;
  while (chunks > 1 && (long)Bx * chunks * 2 > (long)Bt * P) chunks >>= 1;
  XFM_REQUIRE((long)Bx * chunks * 2 <= (long)Bt * P, "vit_tokens_bwd: dtok too small to lend the scratch");
  hipLaunchKernelGGL(vit_tokens_bwd_vec_kernel, dim3(Bx * chunks), dim3(256), 0, st, dx0, mask, Bx, P, D, chunks, dtok);
  rc = xfm_check_launch("vit_tokens_bwd_vec");
  if (rc != XFM_OK) return rc;
  ReduceSets rs{dtok, {dcls, mask != nullptr ? dmask_token : nullptr, nullptr, nullptr}, Bx * chunks, D};
  rc = launch_reduce_sets(rs, 2, st, "vit_tokens_bwd_fold");
  if (rc != XFM_OK) return rc;
  hipLaunchKernelGGL(vit_tokens_bwd_tok_kernel, dim3(grid), dim3(256), 0, st, dx0, mask, Bt, Bx, P, D, dtok);
  return xfm_check_launch("vit_tokens_bwd");
}

// ---------------------------------------------------------------------------------------------
// RoBERTa embeddings: y = dropout(LN(word[id] + type[0] + pos[p])), p = cumsum(id != pad) * (id != pad) + pad
// (xroberta.py:104-137, :1747-1757); pos_mode 1 = BERT: p = t, no padding row in the position table (xbert.py:188-215).
// One wave per token.
// ---------------------------------------------------------------------------------------------
typedef xfm_embed_args EmbArgs;

__device__ __forceinline__ int roberta_pos(const int64_t* ids_row, int t, int pad, int lane) {
  int cnt = 0;
  for (int j0 = 0; j0 <= t; j0 += 64) {
    const int j = j0 + lane;
    const bool nz = (j <= t) && (ids_row[j] != pad);
    cnt += __popcll(__ballot(nz));
  }
  return (ids_row[t] != pad) ? cnt + pad : pad;
}

template <int NCH>
__global__ __launch_bounds__(256) void emb_fwd_kernel(EmbArgs p) {
  constexpr int D = NCH * 256;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int rows = p.B * p.T;
  for (int row = wave; row < rows; row += nwaves) {
    const int b = row / p.T, t = row % p.T;
    const int orow = p.row_map != nullptr ? p.row_map[row] : row;  // packed output row (wave-uniform)
    if (orow < 0) continue;
    const uint32_t rkey = rng_row_key(p.seed_lo, p.seed_hi, (uint32_t)row);
    const int64_t* ids_row = p.ids + (long)b * p.T;
    const int pid = p.pos_mode ? t : roberta_pos(ids_row, t, p.pad_id, lane);
    const long wid = ids_row[t];
    if (lane == 0) p.pos_ids[row] = pid;
    float v[NCH][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e = (i * 64 + lane) * 4;
      const f32x4 a = *reinterpret_cast<const f32x4*>(p.word + wid * D + e);
      const f32x4 c = *reinterpret_cast<const f32x4*>(p.pos + (long)pid * D + e);
      const f32x4 d = *reinterpret_cast<const f32x4*>(p.type + e);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[i][j] = a[j] + d[j] + c[j]; s += v[i][j]; }
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
      const f32x4 wv = *reinterpret_cast<const f32x4*>(p.w + e), bv = *reinterpret_cast<const f32x4*>(p.b + e);
      bf16x4 o;
      f32x4 o32;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float yv = (v[i][j] - mu) * rstd * wv[j] + bv[j];
        if (p.drop_thresh != 0u) {
          yv = rng_keep(rng_u32(rkey, (uint32_t)(e + j)), p.drop_thresh) ? yv * p.drop_scale : 0.f;
        }
        o[j] = f2bf(yv);
        o32[j] = yv;
      }
      *reinterpret_cast<bf16x4*>(p.y + (long)orow * D + e) = o;
      if (p.y32 != nullptr) *reinterpret_cast<f32x4*>(p.y32 + (long)orow * D + e) = o32;
    }
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void emb_bwd_kernel(EmbArgs p) {
  constexpr int D = NCH * 256;
  __shared__ float red[4][D];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int rows = p.B * p.T;
  float acc[3][NCH][4];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[s][i][j] = 0.f;
  for (int row = wave; row < rows; row += nwaves) {
    const int orow = p.row_map != nullptr ? p.row_map[row] : row;
    if (orow < 0) continue;
    const uint32_t rkey = rng_row_key(p.seed_lo, p.seed_hi, (uint32_t)row);
    const long wid = p.ids[row];
    const int pid = p.pos_ids[row];
    const float mu = p.mean[row], rstd = p.rstd[row];
    float dy[NCH][4], xh[NCH][4], wv[NCH][4];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e = (i * 64 + lane) * 4;
      const f32x4 a = *reinterpret_cast<const f32x4*>(p.word + wid * D + e);
      const f32x4 c = *reinterpret_cast<const f32x4*>(p.pos + (long)pid * D + e);
      const f32x4 d = *reinterpret_cast<const f32x4*>(p.type + e);
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(p.w + e);
      const bf16x4 g = *reinterpret_cast<const bf16x4*>(p.dy + (long)orow * D + e);
      f32x4 g32 = {0.f, 0.f, 0.f, 0.f};
      if (p.dy32 != nullptr) g32 = *reinterpret_cast<const f32x4*>(p.dy32 + (long)orow * D + e);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float gv = bf2f(g[j]) + g32[j];
        if (p.drop_thresh != 0u) {
          gv = rng_keep(rng_u32(rkey, (uint32_t)(e + j)), p.drop_thresh) ? gv * p.drop_scale : 0.f;
        }
        dy[i][j] = gv;
        wv[i][j] = w4[j];
        xh[i][j] = (a[j] + d[j] + c[j] - mu) * rstd;
        const float gw = gv * w4[j];
        c1 += gw;
        c2 += gw * xh[i][j];
        acc[0][i][j] += gv * xh[i][j];
        acc[1][i][j] += gv;
      }
    }
    c1 = wave_sum(c1) * (1.0f / D);
    c2 = wave_sum(c2) * (1.0f / D);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e = (i * 64 + lane) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float dz = rstd * (dy[i][j] * wv[i][j] - c1 - xh[i][j] * c2);
        acc[2][i][j] += dz;
        if (p.dz_out != nullptr) {   // ordered scatter by the caller (xfm_rows_segment_sum)
          p.dz_out[(long)row * D + e + j] = dz;
          continue;
        }
        // nn.Embedding(padding_idx): the pad row receives no gradient (xroberta.py:80,100-102)
        if (wid != p.pad_id) atomicAdd(p.dword + wid * D + e + j, dz);
        if (p.pos_mode || pid != p.pad_id) atomicAdd(p.dpos + (long)pid * D + e + j, dz);
      }
    }
  }
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wib][(i * 64 + lane) * 4 + j] = acc[s][i][j];
    __syncthreads();
    float* dst = p.partial + ((long)s * gridDim.x + blockIdx.x) * D;
    for (int c = threadIdx.x; c < D; c += 256) dst[c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
  }
}


static int emb_grid(int rows) {
  int g = cdiv(rows, 16);
  if (g > 256) g = 256;
  return g < 1 ? 1 : g;
}

int xfm_emb_fwd_impl(const EmbArgs& p, int D, hipStream_t st) {
  XFM_REQUIRE(D == 768 || D == 1024, "embedding: unsupported width %d", D);
  XFM_REQUIRE(p.B > 0 && p.T > 0, "embedding: empty batch");
  int grid = cdiv(p.B * p.T, 4);
  if (grid > 2048) grid = 2048;
  if (D == 768) hipLaunchKernelGGL(emb_fwd_kernel<3>, dim3(grid), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(emb_fwd_kernel<4>, dim3(grid), dim3(256), 0, st, p);
  return xfm_check_launch("emb_fwd");
}

int xfm_emb_bwd_impl(EmbArgs p, int D, float* dgamma, float* dbeta, float* dtype, float* workspace, long workspace_bytes,
                     hipStream_t st) {
  XFM_REQUIRE(D == 768 || D == 1024, "embedding: unsupported width %d", D);
  const int grid = emb_grid(p.B * p.T);
  XFM_REQUIRE(workspace != nullptr && workspace_bytes >= (long)3 * grid * D * 4, "embedding bwd: workspace too small");
  p.partial = workspace;
  if (D == 768) hipLaunchKernelGGL(emb_bwd_kernel<3>, dim3(grid), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(emb_bwd_kernel<4>, dim3(grid), dim3(256), 0, st, p);
  int rc = xfm_check_launch("emb_bwd");
  if (rc != XFM_OK) return rc;
  ReduceSets r{workspace, {dgamma, dbeta, dtype, nullptr}, grid, D};
  hipLaunchKernelGGL(reduce_sets_kernel, dim3(cdiv(D, 64), 3), dim3(256), 0, st, r);
  return xfm_check_launch("emb_bwd_reduce");
}

// ---------------------------------------------------------------------------------------------
// Vocabulary cross-entropy on fp32 logits [R, ld] (xroberta.py:1296-1297, CrossEntropyLoss(ignore_index=-100)).
// forward: per-row logsumexp and loss (0 for ignored rows).  backward: dlogits (bf16) = (softmax - onehot) * scale[0],
// zero in ignored rows and in the padding columns [V, ldd).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, long ld, int V, const int64_t* __restrict__ labels,
                                                     float* __restrict__ lse, float* __restrict__ loss) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* x = logits + (long)row * ld;
  float mx = -3.0e38f;
  for (int c = tid * 4; c < V; c += 1024) {
    if (c + 4 <= V) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(x + c);
      mx = fmaxf(fmaxf(mx, fmaxf(a[0], a[1])), fmaxf(a[2], a[3]));
    } else {
      for (int i = c; i < V; ++i) mx = fmaxf(mx, x[i]);
    }
  }
  mx = wave_max(mx);
  if (lane == 0) red[w] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = tid * 4; c < V; c += 1024) {
    if (c + 4 <= V) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(x + c);
      s += __expf(a[0] - mx) + __expf(a[1] - mx) + __expf(a[2] - mx) + __expf(a[3] - mx);
    } else {
      for (int i = c; i < V; ++i) s += __expf(x[i] - mx);
    }
  }
  s = wave_sum(s);
  if (lane == 0) red[w] = s;
  __syncthreads();
  if (tid == 0) {
    const float l = mx + __logf(red[0] + red[1] + red[2] + red[3]);
    lse[row] = l;
    const int64_t lab = labels[row];
    loss[row] = (lab >= 0 && lab < V) ? l - x[lab] : 0.f;
  }
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, long ld, int V, const int64_t* __restrict__ labels,
                                                     const float* __restrict__ lse, const float* __restrict__ scale,
                                                     int per_row_scale, bf16* __restrict__ dlogits, long ldd) {
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* x = logits + (long)row * ld;
  bf16* d = dlogits + (long)row * ldd;
  const int64_t lab = labels[row];
  const bool valid = lab >= 0 && lab < V;
  const float l = lse[row], sc = per_row_scale ? scale[row] : scale[0];
  for (int c = tid * 8; c < ldd; c += 2048) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int col = c + i;
      float g = 0.f;
      if (valid && col < V) g = (__expf(x[col] - l) - (col == lab ? 1.f : 0.f)) * sc;
      o[i] = f2bf(g);
    }
    if (c + 8 <= ldd) *reinterpret_cast<bf16x8*>(d + c) = o;
    else
      for (int i = 0; c + i < ldd; ++i) d[c + i] = o[i];
  }
}

int xfm_ce_fwd_impl(const float* logits, long ld, int R, int V, const int64_t* labels, float* lse, float* loss, hipStream_t st) {
  XFM_REQUIRE(R > 0 && V > 0 && ld >= V && (ld % 4 == 0 || V < 4), "ce_fwd: bad shape R=%d V=%d ld=%ld", R, V, ld);
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(R), dim3(256), 0, st, logits, ld, V, labels, lse, loss);
  return xfm_check_launch("ce_fwd");
}
int xfm_ce_bwd_impl(const float* logits, long ld, int R, int V, const int64_t* labels, const float* lse, const float* scale,
                    int per_row_scale, void* dlogits, long ldd, hipStream_t st) {
  XFM_REQUIRE(R > 0 && V > 0 && ld >= V && ldd >= V && ldd % 8 == 0, "ce_bwd: bad shape R=%d V=%d ld=%ld ldd=%ld", R, V, ld, ldd);
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(R), dim3(256), 0, st, logits, ld, V, labels, lse, scale, per_row_scale, (bf16*)dlogits, ldd);
  return xfm_check_launch("ce_bwd");
}

// ---------------------------------------------------------------------------------------------
// Flat-arena AdamW (optim.py:4-50 parameter groups; transformers AdamW: correct_bias=True, decoupled decay) with the
// global-norm clip factor folded in (apex_ddp_accelerator.py:100-110).  Per-element group id selects lr / decay.
// ---------------------------------------------------------------------------------------------
typedef xfm_adamw_args AdamArgs;
__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
  const float cc = a.clip_coef ? a.clip_coef[0] : 1.f;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < a.n; i += (long)gridDim.x * 1024) {
    const int gid = a.group[i >> 8];
    const float lr = a.lr[gid], wd = a.wd[gid];
    f32x4 p = *reinterpret_cast<f32x4*>(a.p + i);
    const f32x4 g = *reinterpret_cast<const f32x4*>(a.g + i);
    f32x4 m = *reinterpret_cast<f32x4*>(a.m + i), v = *reinterpret_cast<f32x4*>(a.v + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gj = g[j] * cc;
      m[j] = a.beta1 * m[j] + (1.f - a.beta1) * gj;
      v[j] = a.beta2 * v[j] + (1.f - a.beta2) * gj * gj;
      const float step = lr * sqrtf(a.bc2) / a.bc1;
      p[j] -= step * m[j] / (sqrtf(v[j]) + a.eps);
      p[j] -= lr * wd * p[j];
    }
    *reinterpret_cast<f32x4*>(a.p + i) = p;
    *reinterpret_cast<f32x4*>(a.m + i) = m;
    *reinterpret_cast<f32x4*>(a.v + i) = v;
    if (a.zero_grad) *reinterpret_cast<f32x4*>(a.g + i) = f32x4{0.f, 0.f, 0.f, 0.f};  // zero_grad() in the same sweep (Pretrain.py:76)
  }
}
int xfm_adamw_impl(const AdamArgs& a, hipStream_t st) {
  XFM_REQUIRE(a.n > 0 && a.n % 256 == 0, "adamw: arena length %ld must be a positive multiple of 256", a.n);
  int grid = cdiv(a.n, 1024);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, st, a);
  return xfm_check_launch("adamw");
}

// out[0] += sum of squares of an fp32 vector, DETERMINISTIC: block partials in a fixed grid, then one workgroup adds them in a
// fixed order.  The gradient norm feeds the clip coefficient of the optimizer, i.e. it is evaluated after the all-reduce on
// every data-parallel rank: an atomic accumulation (rank-dependent rounding) makes the replicas' weights drift apart by an ulp
// per step, which nothing ever re-synchronises.
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long n, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + i);
    s += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int nparts, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] += (red[0] + red[1]) + (red[2] + red[3]);
}
int xfm_sumsq_impl(const float* x, long n, float* out, float* workspace, hipStream_t st) {
  XFM_REQUIRE(n > 0 && n % 4 == 0, "sumsq: length %ld must be a positive multiple of 4", n);
  int grid = cdiv(n, 1024);
  if (grid > XFM_SUMSQ_WORKSPACE_FLOATS) grid = XFM_SUMSQ_WORKSPACE_FLOATS;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(grid), dim3(256), 0, st, x, n, workspace);
  int rc = xfm_check_launch("sumsq");
  if (rc != XFM_OK) return rc;
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, workspace, grid, out);
  return xfm_check_launch("sumsq_final");
}

// ---------------------------------------------------------------------------------------------
// Row gather / scatter-add on bf16 rows (packed token rows: [CLS] and masked-position gathers, fusion batch assembly).
// One 16-B chunk per thread; index < 0 -> zero row (gather) / skipped (scatter).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rows_gather_kernel(const bf16* __restrict__ src, const int* __restrict__ index, int R, int D,
                                                          bf16* __restrict__ dst) {
  const int cpr = D >> 3;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)R * cpr) return;
  const int r = (int)(t / cpr), c = (int)(t % cpr);
  const int s = index[r];
  u32x4 v = u32x4{0, 0, 0, 0};
  if (s >= 0) v = *reinterpret_cast<const u32x4*>(src + (long)s * D + c * 8);
  *reinterpret_cast<u32x4*>(dst + (long)r * D + c * 8) = v;
}
__global__ __launch_bounds__(256) void rows_scatter_add_kernel(const bf16* __restrict__ src, const int* __restrict__ index, int R, int D,
                                                               float* __restrict__ dst) {
  const int cpr = D >> 3;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)R * cpr) return;
  const int r = (int)(t / cpr), c = (int)(t % cpr);
  const int s = index[r];
  if (s < 0) return;
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (long)r * D + c * 8);
  float* d = dst + (long)s * D + c * 8;
#pragma unroll
  for (int i = 0; i < 8; ++i) atomicAdd(d + i, bf2f(v[i]));
}
int xfm_rows_gather_impl(const bf16* src, const int* index, int R, int D, bf16* dst, hipStream_t st) {
  XFM_REQUIRE(R > 0 && D > 0 && D % 8 == 0, "rows_gather: bad shape R=%d D=%d", R, D);
  hipLaunchKernelGGL(rows_gather_kernel, dim3(cdiv((long)R * (D >> 3), 256)), dim3(256), 0, st, src, index, R, D, dst);
  return xfm_check_launch("rows_gather");
}
int xfm_rows_scatter_add_impl(const bf16* src, const int* index, int R, int D, float* dst, hipStream_t st) {
  XFM_REQUIRE(R > 0 && D > 0 && D % 8 == 0, "rows_scatter_add: bad shape R=%d D=%d", R, D);
  hipLaunchKernelGGL(rows_scatter_add_kernel, dim3(cdiv((long)R * (D >> 3), 256)), dim3(256), 0, st, src, index, R, D, dst);
  return xfm_check_launch("rows_scatter_add");
}

// out[key] += the rows of one run of equal sorted keys, in position order: block i owns the run that STARTS at position i (others exit)
__global__ __launch_bounds__(256) void rows_segment_sum_kernel(const float* __restrict__ src, const int64_t* __restrict__ perm,
                                                               const int64_t* __restrict__ key, long R, int D, long skip_key,
                                                               float* __restrict__ out) {
  const long i = blockIdx.x;
  const int64_t k = key[i];
  if (k < 0 || k == skip_key || (i > 0 && key[i - 1] == k)) return;
  for (int c = threadIdx.x * 4; c < D; c += 1024) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long j = i; j < R && key[j] == k; ++j) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + perm[j] * D + c);
      acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    f32x4* dst = reinterpret_cast<f32x4*>(out + k * D + c);
    *dst = *dst + acc;
  }
}
int xfm_rows_segment_sum_impl(const float* src, const int64_t* perm, const int64_t* key, long R, int D, long skip_key, float* out,
                              hipStream_t st) {
  XFM_REQUIRE(R >= 0 && D > 0 && D % 4 == 0, "rows_segment_sum: bad shape R=%ld D=%d", R, D);
  if (R == 0) return XFM_OK;
  hipLaunchKernelGGL(rows_segment_sum_kernel, dim3((unsigned)R), dim3(256), 0, st, src, perm, key, R, D, skip_key, out);
  return xfm_check_launch("rows_segment_sum");
}

// ---------------------------------------------------------------------------------------------
// Block-wise MIM mask sampler on the device (masking_generator.py:27-105 of the reference: MaskingGenerator.__call__ / _mask).
// The reference draws every image's mask with Python `random` on the host, B times per step (beit2.py:432-439); here one wavefront
// owns one image and runs the same rejection loop with a counter-based generator (every draw is a pure function of (seed, image,
// draw index), so a launch is reproducible): rectangles of area U[min, remaining] and log-uniform aspect in [lo, hi], at most ten
// attempts per block, accepted when they add between 1 and `remaining` new patches; then the uniform top-up to exactly `num`.
// The patch grid lives in registers, lane l holding patches l, l + 64, ... ; counting a rectangle's overlap is a ballot + popcount.
// delta_hist (optional, int32 [H*W + 1]): histogram of the new patches each accepted block added (the distribution pin of the tests).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float mim_u01(uint32_t seed_lo, uint32_t seed_hi, uint32_t img, uint32_t& ctr) {
  const uint32_t key = rng_row_key(seed_lo, seed_hi, img);
  const uint32_t r = rng_u32(key, ctr++);
  return (float)(r >> 8) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(256) void mim_masks_kernel(int B, int GH, int GW, int num, int min_num, float log_lo, float log_hi,
                                                         uint32_t seed_lo, uint32_t seed_hi, uint8_t* __restrict__ out,
                                                         int* __restrict__ delta_hist) {
  const int lane = threadIdx.x & 63;
  const int img = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (img >= B) return;  // whole waves leave together
  const int P = GH * GW;
  constexpr int MAXW = 16;  // up to 1024 patches (32 x 32 grid)
  bool m[MAXW];
#pragma unroll
  for (int i = 0; i < MAXW; ++i) m[i] = false;
  uint32_t ctr = 0;
  int count = 0;
  while (count < num) {
    const int maxp = num - count;
    int delta = 0;
    for (int attempt = 0; attempt < 10 && delta == 0; ++attempt) {
      const float area = (float)min_num + ((float)maxp - (float)min_num) * mim_u01(seed_lo, seed_hi, img, ctr);
      const float ar = __expf(log_lo + (log_hi - log_lo) * mim_u01(seed_lo, seed_hi, img, ctr));
      const int h = (int)rintf(sqrtf(area * ar)), w = (int)rintf(sqrtf(area / ar));
      // (the two position draws are consumed whether or not the rectangle fits, so the stream stays aligned per attempt)
      const float ut = mim_u01(seed_lo, seed_hi, img, ctr), ul = mim_u01(seed_lo, seed_hi, img, ctr);
      if (!(w < GW && h < GH)) continue;
      int top = (int)(ut * (float)(GH - h + 1)), left = (int)(ul * (float)(GW - w + 1));
      top = top > GH - h ? GH - h : top;
      left = left > GW - w ? GW - w : left;
      int masked = 0;
      bool in[MAXW];
#pragma unroll
      for (int i = 0; i < MAXW; ++i) {
        const int p = i * 64 + lane, y = p / GW, x = p - y * GW;
        in[i] = p < P && y >= top && y < top + h && x >= left && x < left + w;
        masked += __popcll(__ballot(in[i] && m[i]));
      }
      const int fresh = h * w - masked;
      if (fresh > 0 && fresh <= maxp) {
#pragma unroll
        for (int i = 0; i < MAXW; ++i) m[i] = m[i] || in[i];
        delta = fresh;
      }
    }
    if (delta == 0) break;
    count += delta;
    if (delta_hist != nullptr && lane == 0) atomicAdd(delta_hist + delta, 1);
  }
  // top-up: `num - count` of the free patches, uniformly without replacement (np.random.choice(..., replace=False))
  int nfree = P - count;
  while (count < num) {
    int r = (int)(mim_u01(seed_lo, seed_hi, img, ctr) * (float)nfree);
    r = r >= nfree ? nfree - 1 : r;
#pragma unroll
    for (int i = 0; i < MAXW; ++i) {
      const int p = i * 64 + lane;
      const bool fr = p < P && !m[i];
      const unsigned long long bal = __ballot(fr);
      const int before = __popcll(bal & ((1ull << lane) - 1ull));
      const int tot = __popcll(bal);
      if (r >= 0 && r < tot && fr && before == r) m[i] = true;
      r -= tot;  // (negative once the patch was found in an earlier word: no later word matches)
    }
    ++count;
    --nfree;
  }
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    const int p = i * 64 + lane;
    if (p < P) out[(long)img * P + p] = m[i] ? 1 : 0;
  }
}

int xfm_mim_masks_impl(int B, int GH, int GW, int num, int min_num, float min_aspect, float max_aspect, uint64_t seed, uint8_t* out,
                       int* delta_hist, hipStream_t st) {
  XFM_REQUIRE(B > 0 && GH > 0 && GW > 0 && GH * GW <= 1024 && num >= 0 && num <= GH * GW && min_num >= 0, "mim_masks: bad geometry");
  XFM_REQUIRE(min_aspect > 0.f && max_aspect >= min_aspect, "mim_masks: bad aspect range");
  hipLaunchKernelGGL(mim_masks_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, B, GH, GW, num, min_num, logf(min_aspect), logf(max_aspect),
                     (uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32), out, delta_hist);
  return xfm_check_launch("mim_masks");
}
