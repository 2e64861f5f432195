// Fused multi-head attention forward / backward for head_dim 64 (gfx950, bf16 MFMA, fp32 softmax).
//
// Reference arithmetic: beit2.py:126-166 (q*scale, + relative-position bias, softmax, attn_drop, @v) and
// xroberta.py:201-289 (q/sqrt(d) BEFORE q@k^T, + additive -10000 key mask, softmax, dropout, @v; causal variant
// :772-792).  Scores are never materialised in HBM: each wave owns 16 query rows, keys stream through LDS in chunks
// of 64 with an online softmax, and S is computed TRANSPOSED (S^T = K.Q^T) so that the fp32 accumulator of one MFMA
// is already laid out as the B operand of the next one (P^T for O^T = V^T.P^T), with no lane movement.  V is staged
// row-major and consumed through the transposed LDS read (ds_read_b64_tr_b16).
//
// Layouts: q/k/v/o are addressed as ptr[(b*S + s)*row_stride + h*64 + d], i.e. straight out of / into the fused
// projection GEMM buffers ([B*S, 3*768] for self-attention, [B*Sk, 2*768] for the cross-attention K/V).
#include <type_traits>
#include "common.h"
#include <stdlib.h>

#define MASK_NEG (-10000.0f)
#define EXCL_NEG (-1.0e30f)
#define ATTN_TILE (64 * 128)       // one 64-row x 64-column bf16 tile
#define ATTN_SLOT (2 * ATTN_TILE)  // K tile + V tile (or Q tile + dO tile)
#define ATTN_RES_MAX 4             // up to 256 rows stay LDS-resident (64 KiB); longer sequences stream chunk by chunk

typedef xfm_attn_args AttnArgs;

__device__ __forceinline__ int swz_a(int r) { return (r >> 1) & 7; }

// Two ROWS x 64 bf16 tiles (K and V, or Q and dO), 128-B rows, rows >= nvalid zero filled.  All global loads of the
// pair are issued before the first LDS store, so one thread keeps up to 8 x 16 B in flight instead of paying the
// memory latency once per 16 B (the kernels are HBM/L2-bound: 64-row tiles of the fused projection buffers).
template <int ROWS, int MAXIT>
__device__ __forceinline__ void stage_pair(char* lds0, const bf16* g0, long rs0, char* lds1, const bf16* g1, long rs1, int row0,
                                           int nvalid, int tid, int nthreads) {
  constexpr int CH = ROWS * 8;
  for (int base = 0; base < CH; base += MAXIT * nthreads) {  // one trip unless the workgroup is a single wave
    u32x4 v0[MAXIT], v1[MAXIT];
#pragma unroll
    for (int i = 0; i < MAXIT; ++i) {
      const int q = base + tid + i * nthreads;
      v0[i] = u32x4{0, 0, 0, 0};
      v1[i] = u32x4{0, 0, 0, 0};
      if (q < CH) {
        const int r = q >> 3, c = q & 7;
        if (row0 + r < nvalid) {
          v0[i] = *reinterpret_cast<const u32x4*>(g0 + (long)(row0 + r) * rs0 + c * 8);
          v1[i] = *reinterpret_cast<const u32x4*>(g1 + (long)(row0 + r) * rs1 + c * 8);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MAXIT; ++i) {
      const int q = base + tid + i * nthreads;
      if (q < CH) {
        const int r = q >> 3, c = q & 7;
        const int off = r * 128 + ((c ^ swz_a(r)) << 4);
        *reinterpret_cast<u32x4*>(lds0 + off) = v0[i];
        *reinterpret_cast<u32x4*>(lds1 + off) = v1[i];
      }
    }
  }
}

// One LDS slot = two 64 x 64 bf16 tiles (K|V or Q|dO), filled by direct-to-LDS loads (global_load_lds_dwordx4): no
// staging registers, every wave's loads for the whole slot are in flight together.  Wave-instruction j (0..15)
// fills rows 8*(j&7).. +7 of tile j>>3, lane-linear; the XOR swizzle is applied on the SOURCE chunk.  Rows past
// `nvalid` re-read the last valid row (finite data; their scores / probabilities are masked to exactly 0).
__device__ __forceinline__ void stage_slot(char* slot, const bf16* g0, long rs0, const bf16* g1, long rs1, int row0, int nvalid,
                                           int w, int nw, int lane) {
  for (int j = w; j < 16; j += nw) {
    const int r = (j & 7) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz_a(r);
    int gr = row0 + r;
    gr = gr < nvalid ? gr : nvalid - 1;
    const bf16* src = ((j >> 3) ? g1 + (long)gr * rs1 : g0 + (long)gr * rs0) + c * 8;
    __builtin_amdgcn_global_load_lds(GLB_PTR(void, src), LDS_PTR(void, slot + (j >> 3) * ATTN_TILE + (j & 7) * 1024), 16, 0, 0);
  }
}
__device__ __forceinline__ void stage_wait() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// A/B fragment of a row-major tile: lane (lg, lr) -> row (row0 + lr), elements [ks*32 + 8*lg, +8)
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int row0, int ks, int lr, int lg) {
  const int r = row0 + lr, c = ks * 4 + lg;
  return *reinterpret_cast<const bf16x8*>(tile + r * 128 + ((c ^ swz_a(r)) << 4));
}

// transposed fragment: k-slots (lg, j<4) -> rows rowA + 4*lg + j ; (lg, j>=4) -> rows rowB + 4*lg + (j-4); column col0 + lr
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int rowA, int rowB, int col0, int lr, int lg) {
  const int col = col0 + 4 * (lr & 3);
  const int ra = rowA + 4 * lg + (lr >> 2), rb = rowB + 4 * lg + (lr >> 2);
  const int offa = ra * 128 + (((col >> 3) ^ swz_a(ra)) << 4) + (col & 7) * 2;
  const int offb = rb * 128 + (((col >> 3) ^ swz_a(rb)) << 4) + (col & 7) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + offa));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + offb));
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo;
  u.s.b = hi;
  return u.v;
}

__device__ __forceinline__ bf16x8 pack_pair(const f32x4& a, const f32x4& b) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) { r[j] = f2bf(a[j]); r[4 + j] = f2bf(b[j]); }
  return r;
}

__device__ __forceinline__ float group4_max(float v) {  // across the 4 lanes that share lane&15
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group4_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// Packed (unpadded) token rows: batch entry b's queries are rows q_start[b] .. q_start[b] + q_len[b] of q / o / dout / dq (q_len <= Sq,
// Sq stays the padded length: statistics, dropout counters and grids are laid out for it); the same for keys / values through
// k_start / k_len.  NULL = dense [B, S] rows.  Keys past k_len are excluded exactly (probability 0), which is what the additive
// -10000 mask of a padded batch gives in fp32 (xroberta.py:751-807), so a prefix-masked batch needs no key_keep when packed.
__device__ __forceinline__ void q_seq(const AttnArgs& a, int b, long& base, int& len) {
  if (a.q_start != nullptr) { base = a.q_start[b]; len = a.q_len[b]; }
  else { base = (long)b * a.Sq; len = a.Sq; }
}
__device__ __forceinline__ void k_seq(const AttnArgs& a, int b, long& base, int& len) {
  if (a.k_start != nullptr) { base = a.k_start[b]; len = a.k_len[b]; }
  else { base = (long)b * a.Sk; len = a.Sk; }
}

// Score post-processing shared by forward and dQ: s = raw*scale + bias (+ MASK_NEG when the key is masked or causally hidden),
// EXCL_NEG past the last key.  Branch-free: the key-keep flags of a chunk are fetched up front with the bias (16 dwords in
// flight, one wait) and every condition becomes a select; chunks with nothing to mask (`plain`) are a bare FMA.
__device__ __forceinline__ void load_keep(const AttnArgs& a, int kvb, int kc, int lg, int (&kk)[4][4]) {
  const int* row = a.key_keep + (long)kvb * a.Sk;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kj = kc * 64 + t * 16 + 4 * lg + r;
      kk[t][r] = row[kj < a.Sk ? kj : a.Sk - 1];
    }
}
__device__ __forceinline__ float score_masked(const AttnArgs& a, float raw, float biasv, bool has_mask, int keep, bool causal, int qi, int kj, int sk) {
  const bool masked = (has_mask & (keep == 0)) | (causal & (kj > qi));
  const float s = fmaf(raw, a.scale, biasv) + (masked ? MASK_NEG : 0.f);
  return kj >= sk ? EXCL_NEG : s;
}

// dropout decision of score (b, h, qi, kj): row = the query row, column = the key.  `drop_key` is loop-invariant wherever a
// lane keeps its query row (forward, dQ); dK/dV walks query rows and pays the key per element.
__device__ __forceinline__ uint32_t drop_key(const AttnArgs& a, int b, int h, int qi) {
  return rng_row_key(a.seed_lo, a.seed_hi, (uint32_t)((b * a.H + h) * a.Sq + qi));
}
__device__ __forceinline__ bool drop_keep(const AttnArgs& a, uint32_t key, int kj) { return rng_keep(rng_u32(key, (uint32_t)kj), a.drop_thresh); }

// additive bias row segment of chunk kc for this lane's query row: keys kc*64 + t*16 + 4*lg .. +3, t = 0..3 (zeros past Sk)
__device__ __forceinline__ void load_bias(const AttnArgs& a, int h, int qc, int kc, int lg, f32x4 (&bv)[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int kj0 = kc * 64 + t * 16 + 4 * lg;
    bv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias != nullptr && kj0 < a.Sk) bv[t] = *reinterpret_cast<const f32x4*>(a.bias + ((long)h * a.Sq + qc) * a.bias_ld + kj0);
  }
}

// Output row of one lane: 4 consecutive columns per d-tile, normalised; o_lo (optional) takes the bf16 of what the bf16 of O lost.
__device__ __forceinline__ void store_out(const AttnArgs& a, long row, int h, int lg, const f32x4 (&oacc)[4], float inv) {
  bf16* op = a.o + row * a.o_rs + h * 64;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    bf16x4 ov, ol;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = oacc[dt][r] * inv;
      ov[r] = f2bf(v);
      ol[r] = f2bf(v - bf2f(ov[r]));
    }
    *reinterpret_cast<bf16x4*>(op + dt * 16 + 4 * lg) = ov;
    if (a.o_lo != nullptr) *reinterpret_cast<bf16x4*>(a.o_lo + row * a.o_rs + h * 64 + dt * 16 + 4 * lg) = ol;
  }
}
// delta_i = dO_i . (O_i + Olo_i) for the query row of lane (lg, lr): each of the 4 lanes sharing lr holds 16 of the 64 columns
// (the two 8-column fragments it already loaded of dO), so the row sum is one group4_sum.
__device__ __forceinline__ float delta_from_out(const AttnArgs& a, long row, int h, int lg, const bf16x8& df0, const bf16x8& df1) {
  const bf16* op = a.o + row * a.o_rs + h * 64;
  const bf16* lp = a.o_lo + row * a.o_rs + h * 64;
  const bf16x8 o0 = *reinterpret_cast<const bf16x8*>(op + 8 * lg), o1 = *reinterpret_cast<const bf16x8*>(op + 32 + 8 * lg);
  const bf16x8 l0 = *reinterpret_cast<const bf16x8*>(lp + 8 * lg), l1 = *reinterpret_cast<const bf16x8*>(lp + 32 + 8 * lg);
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    t = fmaf(bf2f(df0[i]), bf2f(o0[i]) + bf2f(l0[i]), t);
    t = fmaf(bf2f(df1[i]), bf2f(o1[i]) + bf2f(l1[i]), t);
  }
  return group4_sum(t);
}

// ---------------------------------------------------------------------------------------------
// forward: grid (q blocks, H, B); block = NW waves, wave w owns query rows [qblk*16*NW + 16*w, +16)
// ---------------------------------------------------------------------------------------------
// PLAIN: no key mask, no causal mask, no dropout (the ViT towers) -- those code paths and their registers are compiled out.
template <bool RES, bool PLAIN>
__global__ __launch_bounds__(1024) void attn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), nthreads = blockDim.x;
  const int lr = lane & 15, lg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  long qbase, kbase;
  int sq, sk;
  q_seq(a, b, qbase, sq);
  const int q0 = (blockIdx.x * (nthreads >> 6) + w) * 16;
  const bool wave_active = q0 < sq;
  const int qi = q0 + lr;
  const int qc = qi < sq ? qi : sq - 1;
  const uint32_t dkey = drop_key(a, b, h, qi);
  const bf16* qp = a.q + (qbase + qc) * a.q_rs + h * 64;
  const bf16x8 qf0 = *reinterpret_cast<const bf16x8*>(qp + 8 * lg);
  const bf16x8 qf1 = *reinterpret_cast<const bf16x8*>(qp + 32 + 8 * lg);
  const int kvb = a.kv_index ? a.kv_index[b] : b;  // several query rows may share one key/value source (deduplicated images)
  k_seq(a, kvb, kbase, sk);
  const bf16* kb = a.k + kbase * a.k_rs + h * 64;
  const bf16* vb = a.v + kbase * a.v_rs + h * 64;

  f32x4 oacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = EXCL_NEG, l_run = 0.f;

  const int nchunks = (sk + 63) / 64;
  constexpr bool resident = RES;  // all chunks of this (b,h) staged once, one barrier (host: Sk <= 256, >= 4 waves)
  const int nw = nthreads >> 6;
  if (resident) {
    for (int kc = 0; kc < nchunks; ++kc) stage_slot(lds + kc * ATTN_SLOT, kb, a.k_rs, vb, a.v_rs, kc * 64, sk, w, nw, lane);
    stage_wait();
  } else {
    stage_slot(lds, kb, a.k_rs, vb, a.v_rs, 0, sk, w, nw, lane);
  }
  for (int kc = 0; kc < nchunks; ++kc) {
    if (!resident) {  // double buffer: chunk kc has landed, everyone is done with chunk kc-1 -> refill its slot
      stage_wait();
      if (kc + 1 < nchunks) stage_slot(lds + ((kc + 1) & 1) * ATTN_SLOT, kb, a.k_rs, vb, a.v_rs, (kc + 1) * 64, sk, w, nw, lane);
    }
    const char* sK = lds + (resident ? kc : (kc & 1)) * ATTN_SLOT;
    const char* sV = sK + ATTN_TILE;
    if (!wave_active) continue;
    f32x4 st[4], bvs[4];
    int kk[PLAIN ? 1 : 4][4];
    const bool has_mask = !PLAIN && a.key_keep != nullptr;
    const bool causal = !PLAIN && a.causal != 0;
    const bool plain = !has_mask && !causal && kc * 64 + 64 <= sk;  // wave-uniform: nothing to mask in this chunk
    load_bias(a, h, qc, kc, lg, bvs);  // bias (and key-keep) loads first: their L2 latency hides under the QK^T MFMAs
    if constexpr (!PLAIN) {
      if (has_mask) load_keep(a, kvb, kc, lg, kk);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sK, t * 16, 0, lr, lg), qf0, st[t], 0, 0, 0);
      st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sK, t * 16, 1, lr, lg), qf1, st[t], 0, 0, 0);
    }
    float mx = EXCL_NEG;
    if (plain) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          st[t][r] = fmaf(st[t][r], a.scale, bvs[t][r]);
          mx = fmaxf(mx, st[t][r]);
        }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          st[t][r] = score_masked(a, st[t][r], bvs[t][r], has_mask, has_mask ? kk[PLAIN ? 0 : t][r] : 1, causal, qi, kc * 64 + t * 16 + 4 * lg + r, sk);
          mx = fmaxf(mx, st[t][r]);
        }
    }
    mx = group4_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st[t][r] = __expf(st[t][r] - m_new);
        psum += st[t][r];
      }
    if (!PLAIN && a.drop_thresh != 0u) {  // one wave-uniform branch per chunk; the row sum above is of the undropped probabilities
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          st[t][r] = drop_keep(a, dkey, kc * 64 + t * 16 + 4 * lg + r) ? st[t][r] * a.drop_scale : 0.f;
    }
    psum = group4_sum(psum);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[dt][r] *= alpha;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 pf = pack_pair(st[2 * s], st[2 * s + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sV, 32 * s, 32 * s + 16, dt * 16, lr, lg), pf, oacc[dt], 0, 0, 0);
    }
  }
  if (!wave_active || qi >= sq) return;
  store_out(a, qbase + qi, h, lg, oacc, 1.0f / l_run);
  if (lg == 0) a.lse[((long)b * a.H + h) * a.stat_ld + qi] = m_run + __logf(l_run);
}

// ---------------------------------------------------------------------------------------------
// backward 1/2: dQ (+ delta, + dbias).  Same decomposition as the forward.
// NKC > 0 selects the bias-gradient variant (Sk <= 64*NKC): one workgroup walks `nb_per_block` batch entries and keeps
// sum_b dS in registers, then flushes it through a wave-private LDS transpose so that every atomic wave-instruction
// adds 64 consecutive keys of one bias row (256 contiguous bytes; MI355X_MICROARCH "Global float atomics").
// delta_i is recomputed exactly as sum_j P_ij dP_ij in a first pass over the keys (see below).
// ---------------------------------------------------------------------------------------------
template <int NKC, bool RES, bool PLAIN>
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(AttnArgs a, int nb_per_block) {
  constexpr bool DBIAS = NKC > 0;
  constexpr int NACC = DBIAS ? NKC : 1;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), nthreads = blockDim.x;
  const int lr = lane & 15, lg = lane >> 4;
  const int h = blockIdx.y;
  const int q0 = (blockIdx.x * (nthreads >> 6) + w) * 16;
  const int qi = q0 + lr;
  constexpr bool resident = RES;

  f32x4 dsacc[NACC][4];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) dsacc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int bi = 0; bi < nb_per_block; ++bi) {
    const int b = blockIdx.z * nb_per_block + bi;
    if (b >= a.B) break;
    long qbase, kbase;
    int sq, sk;
    q_seq(a, b, qbase, sq);
    const bool wave_active = q0 < sq;
    const bool qvalid = qi < sq;
    const int qc = qvalid ? qi : sq - 1;
    const uint32_t dkey = drop_key(a, b, h, qi);
    const bf16* qp = a.q + (qbase + qc) * a.q_rs + h * 64;
    const bf16* dop = a.dout + (qbase + qc) * a.do_rs + h * 64;
    const bf16x8 qf0 = *reinterpret_cast<const bf16x8*>(qp + 8 * lg);
    const bf16x8 qf1 = *reinterpret_cast<const bf16x8*>(qp + 32 + 8 * lg);
    const bf16x8 df0 = *reinterpret_cast<const bf16x8*>(dop + 8 * lg);
    const bf16x8 df1 = *reinterpret_cast<const bf16x8*>(dop + 32 + 8 * lg);
    const long stat_idx = ((long)b * a.H + h) * a.stat_ld + qc;
    const float lse = a.lse[stat_idx];
    const int kvb = a.kv_index ? a.kv_index[b] : b;
    k_seq(a, kvb, kbase, sk);
    const int nchunks = (sk + 63) / 64;
    const bf16* kb = a.k + kbase * a.k_rs + h * 64;
    const bf16* vb = a.v + kbase * a.v_rs + h * 64;

    const int nw = nthreads >> 6;
    if (resident) {
      __syncthreads();  // previous batch entry's readers are done
      for (int kc = 0; kc < nchunks; ++kc) stage_slot(lds + kc * ATTN_SLOT, kb, a.k_rs, vb, a.v_rs, kc * 64, sk, w, nw, lane);
      stage_wait();
    }
    // streaming mode: double-buffered slots; `first` issues chunk 0 of a pass, `next` waits for chunk kc and refills
    auto stream_first = [&]() {
      __syncthreads();
      stage_slot(lds, kb, a.k_rs, vb, a.v_rs, 0, sk, w, nw, lane);
    };
    auto stream_next = [&](int kc) {
      stage_wait();
      if (kc + 1 < nchunks) stage_slot(lds + ((kc + 1) & 1) * ATTN_SLOT, kb, a.k_rs, vb, a.v_rs, (kc + 1) * 64, sk, w, nw, lane);
    };
    // probabilities P (recomputed from the forward's log-sum-exp) and dropped dP = (dO . V^T) * keep/(1-p) of one chunk
    auto probs = [&](int kc, f32x4 (&st)[4], f32x4 (&dp)[4]) {
      const char* sK = lds + (resident ? kc : (kc & 1)) * ATTN_SLOT;
      const char* sV = sK + ATTN_TILE;
      f32x4 bvs[4];
      int kk[PLAIN ? 1 : 4][4];
      const bool has_mask = !PLAIN && a.key_keep != nullptr;
      const bool causal = !PLAIN && a.causal != 0;
      const bool plain = !has_mask && !causal && kc * 64 + 64 <= sk;
      load_bias(a, h, qc, kc, lg, bvs);
      if constexpr (!PLAIN) {
        if (has_mask) load_keep(a, kvb, kc, lg, kk);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sK, t * 16, 0, lr, lg), qf0, st[t], 0, 0, 0);
        st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sK, t * 16, 1, lr, lg), qf1, st[t], 0, 0, 0);
        dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sV, t * 16, 0, lr, lg), df0, dp[t], 0, 0, 0);
        dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sV, t * 16, 1, lr, lg), df1, dp[t], 0, 0, 0);
      }
      const float lse_q = qvalid ? lse : 3.0e38f;  // rows past Sq: exp(s - 3e38) = 0, no per-element select
      if (plain) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) st[t][r] = __expf(fmaf(st[t][r], a.scale, bvs[t][r]) - lse_q);
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)  // excluded keys: exp(EXCL_NEG - lse) = 0
            st[t][r] = __expf(score_masked(a, st[t][r], bvs[t][r], has_mask, has_mask ? kk[PLAIN ? 0 : t][r] : 1, causal, qi, kc * 64 + t * 16 + 4 * lg + r, sk) - lse_q);
      }
      if (!PLAIN && a.drop_thresh != 0u) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            dp[t][r] = drop_keep(a, dkey, kc * 64 + t * 16 + 4 * lg + r) ? dp[t][r] * a.drop_scale : 0.f;
      }
    };

    // pass 1: delta_i = sum_j P_ij dP_ij from the SAME P and dP that form dS below, so that sum_j dS_ij = 0 holds to
    // fp32 rounding (rowsum(dO*O) with a bf16-rounded O breaks it by ~2^-9 |dO||O| and swamps small dS)
    float delta = 0.f;
    f32x4 st[4], dp[4];
    const bool fast_delta = a.o_lo != nullptr;  // delta = dO . (O + Olo): no first pass over the keys (uniform over the launch)
    if (fast_delta) {
      delta = delta_from_out(a, qbase + qc, h, lg, df0, df1);
    } else {
      if (!resident) stream_first();
      for (int kc = 0; kc < nchunks; ++kc) {
        if (!resident) stream_next(kc);
        if (wave_active) {
          probs(kc, st, dp);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) delta += st[t][r] * dp[t][r];
        }
      }
      delta = group4_sum(delta);
    }
    if (wave_active && qvalid && lg == 0) a.delta[stat_idx] = delta;

    f32x4 dqacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dqacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // pass 2: dS, dbias, dQ   (a single-chunk problem keeps pass 1's registers and its staged tile)
    if (!resident && (nchunks > 1 || fast_delta)) stream_first();
    for (int kc = 0; kc < nchunks; ++kc) {
      if (nchunks > 1 || fast_delta) {
        if (!resident) stream_next(kc);
        if (wave_active) probs(kc, st, dp);
      }
      if (!wave_active) continue;
      const char* sK = lds + (resident ? kc : (kc & 1)) * ATTN_SLOT;
      // bias gradient without the in-register sums (NKC = 0): this entry's dS goes to the workspace [B,H,Sq,bias_ld] when there is
      // one (dbias_reduce_kernel adds the batch sum to dbias afterwards), else one float atomic per score
      float* wsrow = (!DBIAS && a.dbias != nullptr && a.dbias_ws != nullptr) ? a.dbias_ws + (((long)b * a.H + h) * a.Sq + qi) * a.bias_ld : nullptr;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int kj0 = kc * 64 + t * 16 + 4 * lg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kj = kj0 + r;
          const float ds = st[t][r] * (dp[t][r] - delta);
          st[t][r] = ds;
          if (!DBIAS && a.dbias != nullptr && wsrow == nullptr && kj < sk && qvalid) atomicAdd(a.dbias + ((long)h * a.Sq + qi) * a.bias_ld + kj, ds);
        }
        if (!DBIAS && wsrow != nullptr && qvalid) {
          if (kj0 + 4 <= a.bias_ld) *reinterpret_cast<f32x4*>(wsrow + kj0) = st[t];  // (columns in [Sk, bias_ld) get exact zeros: P = 0 there)
          else
            for (int r = 0; r < 4; ++r)
              if (kj0 + r < a.bias_ld) wsrow[kj0 + r] = st[t][r];
        }
      }
      if (DBIAS) {  // static register indices only: a wave-uniform compare selects the chunk's accumulator
#pragma unroll
        for (int c = 0; c < NACC; ++c)
          if (c == kc) {
#pragma unroll
            for (int t = 0; t < 4; ++t) dsacc[c][t] += st[t];
          }
      }
      // dQ^T[d, q] += K^T[d, key] . dS^T[key, q]
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = pack_pair(st[2 * s2], st[2 * s2 + 1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          dqacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sK, 32 * s2, 32 * s2 + 16, dt * 16, lr, lg), pf, dqacc[dt], 0, 0, 0);
      }
    }
    if (wave_active && qvalid) {
      bf16* dqp = a.dq + (qbase + qi) * a.dq_rs + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = f2bf(dqacc[dt][r] * a.scale);
        *reinterpret_cast<bf16x4*>(dqp + dt * 16 + 4 * lg) = ov;
      }
    }
  }

  if (DBIAS) {  // (dense rows only: the launcher never pairs the bias-gradient variant with packed rows)
    const bool wave_active = q0 < a.Sq;
    const int nchunks = (a.Sk + 63) / 64;
    float* fl = reinterpret_cast<float*>(lds + w * 4096);  // wave-private [16 q][64 keys], aliases the K/V slots (done with)
#pragma unroll
    for (int kc = 0; kc < NACC; ++kc) {
      if (kc < nchunks) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(fl + lr * 64 + t * 16 + 4 * lg) = dsacc[kc][t];
        __syncthreads();
        if (wave_active && a.dbias != nullptr) {
          const int kj = kc * 64 + lane;
          for (int row = 0; row < 16; ++row) {
            const int q = q0 + row;
            if (q < a.Sq && kj < a.Sk) atomicAdd(a.dbias + ((long)h * a.Sq + q) * a.bias_ld + kj, fl[row * 64 + lane]);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward 1/2 for short, dense, unmasked problems (the 224-px ViT: Sq = Sk = 197; any Sk <= 256): dQ, delta, dbias.
// The kernel above gives a wave 16 queries against ALL keys; with the bias gradient that is 253 VGPRs (one 7-wave workgroup per
// CU), a bias row segment fetched per key chunk, and two passes over the keys for delta: 202 us per ViT layer, 12 us per (batch,
// head, query block), against 1 us of MFMA time.  Here a workgroup is (query group of <= 4 tiles, head) and its 16 waves are
// (query tile, key range): a wave owns 16 queries x <= 64 keys of EVERY batch entry the workgroup walks, so
//   * its bias tile is loaded once (16 VGPRs, batch-invariant) and its bias-gradient sum is 16 VGPRs, not 64;
//   * S and dP are computed once: the four key-range waves of a query tile exchange their partial delta through LDS;
//   * dS crosses LDS once (bf16, 8 B per lane and tile) and the wave with key range w sums d-tile w of dQ over all keys;
//   * K, V (second LDS buffers) and the Q / dO fragments (registers) of the next entry are fetched while this one computes.
// LDS: K (2 x 32 KB) | V (2 x 32 KB) | dS exchange (8 KB per query tile) | delta partials.  One workgroup (12 waves) per CU.
// ---------------------------------------------------------------------------------------------
#define VB_KBUF (ATTN_RES_MAX * ATTN_TILE)
#define VB_EXCH(QT) ((QT) * 16 * 512)
#define VB_LDS(QT) (4 * VB_KBUF + VB_EXCH(QT) + (QT) * 4 * 16 * 4)
#define VB_LDS_QL(QT, NP) (4 * (2 * (NP) * 2048) + (QT) * (NP) * 1024 + (QT) * 4 * 16 * 4 + 2 * (QT) * 2 * 2048)   // Q / dO through LDS (PRE, NP <= 7)

// rows [0, nrows) of a [*, 64] bf16 operand into consecutive 64-row tiles (direct-to-LDS, same image as stage_slot's tiles).
// Issued as inline asm on purpose: when the compiler sees a direct-to-LDS load it drains it (s_waitcnt vmcnt(0)) in front of
// every later LDS read, which would serialise the prefetch of the next entry with the MFMAs of this one.  The caller waits
// (s_waitcnt vmcnt) before the barrier that publishes the tiles.
__device__ __forceinline__ void stage_rows(char* tiles, const bf16* g, long rs, int nrows, int w, int nw, int lane) {
  const int n = ((nrows + 63) >> 6) * 8;
  for (int j = w; j < n; j += nw) {
    const int r = (j & 7) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz_a(r);
    int gr = (j >> 3) * 64 + r;
    gr = gr < nrows ? gr : nrows - 1;
    const bf16* src = g + (long)gr * rs + c * 8;
    const unsigned dst = (unsigned)(uintptr_t)LDS_PTR(void, tiles) + (unsigned)__builtin_amdgcn_readfirstlane((j >> 3) * ATTN_TILE + (j & 7) * 1024);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory", "m0");
  }
}
// tr_frag with the two 16-row halves of the k dimension in (possibly) different tiles
__device__ __forceinline__ bf16x8 tr_frag2(const char* tileA, int rowA, const char* tileB, int rowB, int col0, int lr, int lg) {
  const int col = col0 + 4 * (lr & 3);
  const int ra = rowA + 4 * lg + (lr >> 2), rb = rowB + 4 * lg + (lr >> 2);
  const int offa = ra * 128 + (((col >> 3) ^ swz_a(ra)) << 4) + (col & 7) * 2;
  const int offb = rb * 128 + (((col >> 3) ^ swz_a(rb)) << 4) + (col & 7) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tileA + offa));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tileB + offb));
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo;
  u.s.b = hi;
  return u.v;
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for vmcnt(0), i.e. for the prefetch DMA in flight
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// PRE: the row term delta_i = dO_i . (O_i + Olo_i) is taken from the forward's output (a.o, a.o_lo) at the top of an entry, from
// fragments fetched one entry ahead -- no delta exchange, no second barrier, and dS leaves in the same phase as the scores.
// !PRE (no o_lo): delta_i = sum_j P_ij dP_ij from the very P and dP that form dS, exchanged between the four key-range waves.
template <int QT, int NP, bool PRE, bool DBG>
__global__ __launch_bounds__(QT * 256) void attn_bwd_dq_short_kernel(AttnArgs a, int nb_per_block, int G, long long* dbg) {
  constexpr int NW = QT * 4;
  // QL (PRE and <= 14 key tiles): the Q / dO tiles of the workgroup's queries come through LDS too -- 4 QT one-KB pieces per entry instead
  // of four fragment loads in each of the 4 QT waves (every key-range wave of a query tile fetched the same rows) -- in the room
  // that 28-KB K / V buffers leave.  The vector-memory path moves ~64 B/clk per CU and every wave-load holds its wave at issue
  // while the queue is full: the loads, not the arithmetic, set the entry period (tools/attn_timeline.py).
  constexpr bool QL = PRE && NP <= 7;
  constexpr int KBUF = QL ? 2 * NP * 2048 : VB_KBUF;           // one K or V image
  constexpr int EXQ = QL ? NP * 1024 : 16 * 512;               // dS exchange of one query tile
  constexpr int QIMG = QT * 2 * 2048;                          // Q tiles | dO tiles of one entry (QL)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  // (w through readfirstlane: the tile counts below are wave-uniform and the compiler must know it -- see the entry loop)
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int qt = w >> 2, kw = w & 3;
  // 1-D grid, logical id = group + G * (head + H * batch slice), an XCD takes a contiguous range of logical ids: the G groups of one
  // (head, batch slice) read the same K / V rows at about the same time and now do so through ONE L2
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int grp = wg % G, h = (wg / G) % a.H, zslice = wg / (G * a.H);
  const int sk = a.Sk, sq = a.Sq;
  const int KT = (sk + 15) >> 4;  // key tiles of 16 (<= 2 * NP), dealt to the four key-range waves as evenly as they go
  const int kbase = KT >> 2, krem = KT & 3;
  const int nt = kbase + (kw < krem ? 1 : 0);
  const int kt0 = kw * kbase + (kw < krem ? kw : krem);
  const int QTILES = (sq + 15) >> 4;  // query tiles dealt to the G groups the same way
  const int qbase_t = QTILES / G, qrem = QTILES % G;
  const int nqt = qbase_t + (grp < qrem ? 1 : 0);
  const int q0 = (grp * qbase_t + (grp < qrem ? grp : qrem) + qt) * 16;
  const bool wave_active = qt < nqt && q0 < sq;
  const int qi = q0 + lr;
  const bool qvalid = wave_active && qi < sq;
  const int qc = qi < sq ? qi : sq - 1;

  char* const sK0 = lds;
  char* const ex = lds + 4 * KBUF;
  float* const dred = reinterpret_cast<float*>(ex + QT * EXQ);
  char* const qimg = reinterpret_cast<char*>(dred) + QT * 4 * 16 * 4;   // (QL) two Q | dO images

  // LDS addressing.  A 16-row tile t of an image starts 2048 B after tile t-1 (four to a 64-row, 8 KB staging tile) and the XOR
  // swizzle of a row depends on (row >> 1) & 7 only, i.e. not on the tile: every fragment address is ONE per-lane offset plus a
  // multiple of 2048 -- an immediate -- instead of a register per fragment.
  const int sw_r = (lr >> 1) & 7;
  const int rf0 = lr * 128 + ((lg ^ sw_r) << 4), rf1 = lr * 128 + (((4 + lg) ^ sw_r) << 4);  // row fragments, k-steps 0 / 1
  const int tr_row = 4 * lg + (lr >> 2), tr_col = kw * 16 + 4 * (lr & 3);                   // transposed fragment of d-tile kw
  const int tro = tr_row * 128 + ((((tr_col >> 3) ^ ((tr_row >> 1) & 7))) << 4) + (tr_col & 7) * 2;
  // dS exchange of a query tile: key tiles in PAIRS, 16 B per lane and pair -- a lane's values of tile 2p in the low, of tile 2p + 1 in
  // the high 8 bytes: the dQ loop reads a pair as ONE ds_read_b128 (256 B/clk; the two 8-byte reads 512 B apart it used to take
  // were fused by the compiler into ds_read2st64_b64, 128 B/clk, and made its phase LDS-bound), and the value read IS the MFMA operand.
  char* const ex_q = ex + qt * EXQ + lane * 16;
  auto ex_slot = [&](int tile) { return ex_q + ((tile >> 1) << 10) + ((tile & 1) << 3); };
  // key tiles past the last one stay zero for the whole kernel (the dQ loop runs over NP pairs)
  for (int i = tid; i < QT * EXQ / 16; i += NW * 64) reinterpret_cast<u32x4*>(ex)[i] = u32x4{0, 0, 0, 0};

  // Softmax in the exponent of 2, the bias folded into the accumulator the score MFMAs start from and the row's log-sum-exp into the
  // exponent's fma:  S' = K.q + bias / scale,  P = exp2(S' * scale * log2 e - lse * log2 e)   (no subtraction, no select: keys past Sk
  // carry bias -1e30, query rows past Sq carry lse = +1e30)
  const float inv_scale = 1.0f / a.scale, c2 = a.scale * 1.44269504088896341f;
  f32x4 bvs[4], dsacc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    dsacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kj0 = (kt0 + t) * 16 + 4 * lg;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias != nullptr && t < nt && kj0 < sk) bv = *reinterpret_cast<const f32x4*>(a.bias + ((long)h * sq + qc) * a.bias_ld + kj0);
#pragma unroll
    for (int r = 0; r < 4; ++r) bvs[t][r] = (kj0 + r < sk ? bv[r] : -1.0e30f) * inv_scale;
  }

  // batch-invariant per-lane byte offsets; an entry adds one scalar stride to the (scalar) base pointers
  // (unsigned 32-bit lane offsets against SCALAR per-entry base pointers: every access is `global_* v_off, s[base]`; as signed offsets
  // the loop-strength reducer turned each into a per-lane 64-bit pointer carried round the loop, 26 VGPRs of them)
  const unsigned q_off = (unsigned)(((long)qc * a.q_rs + h * 64 + 8 * lg) * 2), do_off = (unsigned)(((long)qc * a.do_rs + h * 64 + 8 * lg) * 2);
  // (PRE: wave kw takes a QUARTER of the row's 64 columns of O / O_lo -- the 4 of this lane's 16 dO columns with index
  // 32 (kw >> 1) + 8 lg + 4 (kw & 1) + 0..3 -- so the four key-range waves together read O once, not four times)
  const unsigned do_q4 = (unsigned)(((long)qc * a.do_rs + h * 64 + 32 * (kw >> 1) + 8 * lg + 4 * (kw & 1)) * 2);
  const unsigned o_off = PRE ? (unsigned)(((long)qc * a.o_rs + h * 64 + 32 * (kw >> 1) + 8 * lg + 4 * (kw & 1)) * 2) : 0u;
  const unsigned dq_off = (unsigned)(((long)(qvalid ? qi : 0) * a.dq_rs + h * 64 + kw * 16 + 4 * lg) * 2), stat_off = (unsigned)((long)h * a.stat_ld + qc);
  const long q_bs = (long)sq * a.q_rs * 2, do_bs = (long)sq * a.do_rs * 2, dq_bs = (long)sq * a.dq_rs * 2, stat_bs = (long)a.H * a.stat_ld;
  const long o_bs = (long)sq * a.o_rs * 2;
  const long k_bs = (long)sk * a.k_rs * 2, v_bs = (long)sk * a.v_rs * 2;
  // this wave's pieces of a K / V image (<= 3 of the 1-KB, 8-row direct-to-LDS instructions): source offset per lane, LDS offset per
  // wave.  The 2 NP tiles the dQ loop reads are staged (rows past Sk repeat the last key: finite, their dS is zero).
  const int b_begin = zslice * nb_per_block;
  int b_end = b_begin + nb_per_block;
  b_end = b_end < a.B ? b_end : a.B;
  // Running (scalar) base pointers instead of `base + b * stride` at every use: the staging / fetch cursors point at the entry being
  // REQUESTED (one ahead of the entry computed), pdelta at the entry computed, pdq at the one before it (whose dQ is stored late);
  // one 64-bit add each per entry (the multiplications were ~100 scalar instructions per entry in a kernel that is issue-bound).
  const char *pk = reinterpret_cast<const char*>(a.k) + (long)b_begin * k_bs, *pv = reinterpret_cast<const char*>(a.v) + (long)b_begin * v_bs;
  const char *pq = reinterpret_cast<const char*>(a.q) + (long)b_begin * q_bs, *pdo = reinterpret_cast<const char*>(a.dout) + (long)b_begin * do_bs;
  const char *po = PRE ? reinterpret_cast<const char*>(a.o) + (long)b_begin * o_bs : nullptr;
  const char *plo = PRE ? reinterpret_cast<const char*>(a.o_lo) + (long)b_begin * o_bs : nullptr;
  const float* plse = a.lse + (long)b_begin * stat_bs;
  float* pdelta = a.delta + (long)b_begin * stat_bs;
  char* pdq = reinterpret_cast<char*>(a.dq) + (long)(b_begin - 1) * dq_bs;
  // (diagnostic build only: bit 3 of the pointer's low bits pins every request on the slice's first entry -- the whole walk then runs
  // from cache, which prices the memory side of the entry period; results are garbage)
  const bool pin = DBG && ((uintptr_t)dbg & 8) != 0;
  auto advance = [&]() {
    if (DBG && pin) return;
    pk += k_bs; pv += v_bs; pq += q_bs; pdo += do_bs; plse += stat_bs;
    if constexpr (PRE) { po += o_bs; plo += o_bs; }
  };
  unsigned pc_dst[3];
  constexpr int n_pc = 4 * NP;
  static_assert(n_pc <= 3 * NW, "three pieces per wave");
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int j = w + i * NW;
    pc_dst[i] = (unsigned)__builtin_amdgcn_readfirstlane((j >> 3) * ATTN_TILE + (j & 7) * 1024);
  }
  unsigned k_offs[3] = {0, 0, 0}, v_offs[3] = {0, 0, 0};
  if constexpr (!PRE) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int j = w + i * NW;
      const int r = (j & 7) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ swz_a(r);
      int gr = (j >> 3) * 64 + r;
      gr = gr < sk ? gr : sk - 1;
      k_offs[i] = (unsigned)(((long)gr * a.k_rs + h * 64 + c * 8) * 2);
      v_offs[i] = (unsigned)(((long)gr * a.v_rs + h * 64 + c * 8) * 2);
    }
  }
  auto stage_piece = [&](int b, int buf, int i) {   // K and V piece i of this wave (inline asm: see stage_rows)
    if (w + i * NW < n_pc) {
      unsigned k_off, v_off;
      if constexpr (PRE) {
        // (the lane offsets are recomputed per piece from an opaque copy of the lane id -- a dozen integer instructions -- instead of
        // living in six VGPRs for the whole kernel: with the O quarters in flight the scores' bias tile would be spilled for them)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int j = w + i * NW;
        const int r = (j & 7) * 8 + (ln >> 3);
        const int c = (ln & 7) ^ swz_a(r);
        int gr = (j >> 3) * 64 + r;
        gr = gr < sk ? gr : sk - 1;
        k_off = (unsigned)(((long)gr * a.k_rs + h * 64 + c * 8) * 2);
        v_off = (unsigned)(((long)gr * a.v_rs + h * 64 + c * 8) * 2);
      } else {
        k_off = k_offs[i];
        v_off = v_offs[i];
      }
      const char* kb = pk;
      const char* vb = pv;
      const unsigned dk = (unsigned)(uintptr_t)LDS_PTR(void, sK0) + (unsigned)buf * KBUF, dv = dk + 2 * KBUF;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(k_off), "s"(kb), "s"(dk + pc_dst[i]) : "memory", "m0");
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(v_off), "s"(vb), "s"(dv + pc_dst[i]) : "memory", "m0");
    }
  };
  auto stage_kv = [&](int b, int buf) {
#pragma unroll
    for (int i = 0; i < 3; ++i) stage_piece(b, buf, i);
  };

  // Q / dO (/ O, O_lo) fragments and the log-sum-exp of the NEXT entry are fetched while the current one computes (a load issued at
  // the top of an entry and waited for there costs the whole HBM latency per entry: every wave of the CU sits behind the same barrier)
  bf16x8 qf0, qf1, df0, df1;
  bf16x4 oq, lq, dq4;
  bf16x4 dq_hold = bf16x4{0, 0, 0, 0};   // this wave's dQ of the entry just finished (stored one entry later)
  float lse_n = 0.f;
  // (QL) piece w of the 4 QT: 8 query rows of Q (w < 2 QT) or dO into image `buf`, rows past Sq repeat the last one
  auto stage_q = [&](int b, int buf) {
    if (w < 4 * QT) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int isd = w >= 2 * QT ? 1 : 0, jj = w - isd * 2 * QT;
      const int r = (jj & 1) * 8 + (ln >> 3);
      const int c = (ln & 7) ^ swz_a(r);
      int gr = q0 - qt * 16 + (jj >> 1) * 16 + r;
      gr = gr < sq ? gr : sq - 1;
      const unsigned off = (unsigned)(((long)gr * (isd ? a.do_rs : a.q_rs) + h * 64 + c * 8) * 2);
      const char* base = isd ? pdo : pq;
      const unsigned dst = (unsigned)(uintptr_t)LDS_PTR(void, qimg) + (unsigned)(buf * QIMG + isd * (QT * 2048) + jj * 1024);
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(dst) : "memory", "m0");
    }
  };
  auto fetch_q = [&](int b) {
    if constexpr (!QL) {
      const char* qp = pq + q_off;   // (scalar base + zero-extended lane offset)
      const char* dop = pdo + do_off;
      qf0 = *reinterpret_cast<const bf16x8*>(qp);
      qf1 = *reinterpret_cast<const bf16x8*>(qp + 64);
      df0 = *reinterpret_cast<const bf16x8*>(dop);
      df1 = *reinterpret_cast<const bf16x8*>(dop + 64);
    } else {   // the dO quarter that meets this wave's O quarter (the row's fragments themselves arrive through LDS)
      dq4 = *reinterpret_cast<const bf16x4*>(pdo + do_q4);
    }
    if constexpr (PRE) {
      const char* op = po + o_off;
      const char* lp = plo + o_off;
      oq = *reinterpret_cast<const bf16x4*>(op);
      lq = *reinterpret_cast<const bf16x4*>(lp);
    }
    lse_n = plse[stat_off];
  };
  if (b_begin < b_end) {
    stage_kv(b_begin, 0);
    if constexpr (QL) stage_q(b_begin, 0);
    fetch_q(b_begin);
    advance();
  }
  const int dbg_wave = (int)((uintptr_t)dbg & 7);   // (the stamping wave, 0..7, rides in the low bits of the 256-B aligned diagnostic pointer)
  dbg = reinterpret_cast<long long*>((uintptr_t)dbg & ~(uintptr_t)15);
  // The number of key tiles of a wave (nt, 0..4) is a run-time, wave-uniform value.  Written as `if (t < nt)` inside the tile loops it
  // made every tile its own exec-masked basic block -- read, wait, MFMA, read, wait, MFMA: 16 LDS round trips in series.  So the whole
  // walk is straight-line code per tile COUNT (NT; -1 = a wave without a query tile: barriers and its share of the staging only),
  // picked by ONE scalar branch per kernel.  (A switch per phase inside one loop made the register allocator merge five versions of
  // the score registers: 38 spilled VGPRs.)
  auto walk = [&](auto NTc) {
    constexpr int NT = decltype(NTc)::value;
    constexpr bool ACT = NT >= 0;
    constexpr int NTS = NT > 0 ? NT : 1;
    for (int b = b_begin; b < b_end; ++b) {
      const int cur = (b - b_begin) & 1;
      const char* sK = sK0 + cur * KBUF;
      // diagnostic (XFM_ATTN_DBG_PTR, tools/attn_timeline.py): one wave stamps the phases of every entry (10-ns clock); NULL in every product call
      long long* const dbe = DBG && dbg != nullptr && tid == dbg_wave * 64 && b - b_begin < 32 ? dbg + ((long)blockIdx.x * 32 + (b - b_begin)) * 16 : nullptr;
      if (DBG && dbe) dbe[0] = wall_clock64();
      // everything up to the fetches of this entry must have landed.  (The compiler cannot see this wait: the empty asm makes it place
      // its own wait for the fetched registers HERE, before this entry's direct-to-LDS loads are issued, rather than at their first
      // use, where a counted wait would also drain those.)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // dQ of the PREVIOUS entry leaves here, behind the wait: vmcnt counts stores too (until L2 has them), and a store issued at the
      // end of an entry made this wait, a few instructions later, sit out its whole acknowledgement
      if constexpr (ACT) {
        if (b > b_begin && qvalid) *reinterpret_cast<bf16x4*>(pdq + dq_off) = dq_hold;
      }
      if constexpr (!QL) asm volatile("" : "+v"(qf0), "+v"(qf1), "+v"(df0), "+v"(df1));
      else asm volatile("" : "+v"(dq4));
      asm volatile("" : "+v"(lse_n));
      if constexpr (PRE) asm volatile("" : "+v"(oq), "+v"(lq));
      const float nlse = qvalid ? -lse_n * 1.44269504088896341f : -1.0e30f;  // rows past Sq: P = exp2(-huge) = 0
      float delta = 0.f;
      if constexpr (PRE && ACT) {
        // this wave's quarter of dO . (O + O_lo): packed bf16 dot products with fp32 accumulation; the four quarters meet in LDS
        bf16x2 d0, d1;
        if constexpr (QL) {
          d0 = bf16x2{dq4[0], dq4[1]};
          d1 = bf16x2{dq4[2], dq4[3]};
        } else {
          const bf16x8 dh = (kw & 2) ? df1 : df0;
          d0 = (kw & 1) ? bf16x2{dh[4], dh[5]} : bf16x2{dh[0], dh[1]};
          d1 = (kw & 1) ? bf16x2{dh[6], dh[7]} : bf16x2{dh[2], dh[3]};
        }
        float t0 = __builtin_amdgcn_fdot2_f32_bf16(d0, bf16x2{oq[0], oq[1]}, 0.f, false);
        float t1 = __builtin_amdgcn_fdot2_f32_bf16(d1, bf16x2{oq[2], oq[3]}, 0.f, false);
        t0 = __builtin_amdgcn_fdot2_f32_bf16(d0, bf16x2{lq[0], lq[1]}, t0, false);
        t1 = __builtin_amdgcn_fdot2_f32_bf16(d1, bf16x2{lq[2], lq[3]}, t1, false);
        const float part = group4_sum(t0 + t1);
        if (lg == 0) dred[(qt * 16 + lr) * 4 + kw] = part;
      }
      if (DBG && dbe) dbe[1] = wall_clock64();
      lds_barrier();  // K(b), V(b) have landed; every wave is done with entry b-1 (its K / V buffers, the exchange tiles)
      if (DBG && dbe) dbe[2] = wall_clock64();
      if constexpr (PRE && ACT) {
        const f32x4 dq4 = *reinterpret_cast<const f32x4*>(dred + (qt * 16 + lr) * 4);
        delta = (dq4[0] + dq4[1]) + (dq4[2] + dq4[3]);
        if (kw == 0 && lg == 0 && qvalid) pdelta[stat_off] = delta;
      }
      const bool more = b + 1 < b_end;
      if constexpr (QL) {
        // everything of the next entry is requested HERE, a whole entry ahead of its use: its Q / dO pieces, the small per-lane loads
        // (their registers are free: this entry's went into delta and nlse above), then K / V
        if (more) {
          stage_q(b + 1, cur ^ 1);
          fetch_q(b + 1);
        }
        if constexpr (ACT) {
          const char* qi_ = qimg + cur * QIMG + qt * 2048;
          qf0 = *reinterpret_cast<const bf16x8*>(qi_ + rf0);
          qf1 = *reinterpret_cast<const bf16x8*>(qi_ + rf1);
          df0 = *reinterpret_cast<const bf16x8*>(qi_ + QT * 2048 + rf0);
          df1 = *reinterpret_cast<const bf16x8*>(qi_ + QT * 2048 + rf1);
        }
      }
      // (placing the K / V pieces between the tiles of the score phase instead -- one K + V piece per tile -- measured the same entry period
      // with ~50 more scalar instructions per entry; all of them go out here)
      if (more) stage_kv(b + 1, cur ^ 1);
      if (DBG && dbe) dbe[8] = wall_clock64();

      f32x4 st[NTS], dp[NTS];
      if constexpr (ACT) {
        const char* ka = sK + kt0 * 2048;
        float dpart = 0.f;
        // fragments of tile t + 1 are requested before the MFMAs of tile t (two tiles' worth, 32 VGPRs, in flight)
        bf16x8 fr[2][4];
        auto frags = [&](int t, bf16x8 (&f)[4]) {
          f[0] = *reinterpret_cast<const bf16x8*>(ka + t * 2048 + rf0);
          f[1] = *reinterpret_cast<const bf16x8*>(ka + t * 2048 + rf1);
          f[2] = *reinterpret_cast<const bf16x8*>(ka + 2 * KBUF + t * 2048 + rf0);
          f[3] = *reinterpret_cast<const bf16x8*>(ka + 2 * KBUF + t * 2048 + rf1);
        };
        auto mfmas = [&](int t) {
          st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[t & 1][0], qf0, bvs[t], 0, 0, 0);
          st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[t & 1][1], qf1, st[t], 0, 0, 0);
          dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[t & 1][2], df0, dp[t], 0, 0, 0);
          dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[t & 1][3], df1, dp[t], 0, 0, 0);
        };
        if constexpr (PRE) {
          // Software pipeline with scheduling fences between the steps: the K (V) fragments of tile t + 1 are requested as soon as the
          // score (dP) MFMAs of tile t have read theirs (16 VGPRs of fragments), and the exponentials / dS of tile t - 1 are written
          // after the MFMAs of tile t and execute beside them (two tiles of scores live).  A free schedule hoists every read and MFMA
          // to the top and spills the bias tiles, whose reloads (vmcnt) would wait behind the K / V prefetch.
          bf16x8 fk[2], fv[2];
          if constexpr (NT > 0) {
            fk[0] = *reinterpret_cast<const bf16x8*>(ka + rf0);
            fk[1] = *reinterpret_cast<const bf16x8*>(ka + rf1);
            fv[0] = *reinterpret_cast<const bf16x8*>(ka + 2 * KBUF + rf0);
            fv[1] = *reinterpret_cast<const bf16x8*>(ka + 2 * KBUF + rf1);
          }
#pragma unroll
          for (int t = 0; t <= NT; ++t) {
            if (t < NT) {
              st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[0], qf0, bvs[t], 0, 0, 0);
              st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[1], qf1, st[t], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              if (t + 1 < NT) {
                fk[0] = *reinterpret_cast<const bf16x8*>(ka + (t + 1) * 2048 + rf0);
                fk[1] = *reinterpret_cast<const bf16x8*>(ka + (t + 1) * 2048 + rf1);
              }
              dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
              dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv[0], df0, dp[t], 0, 0, 0);
              dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv[1], df1, dp[t], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              if (t + 1 < NT) {
                fv[0] = *reinterpret_cast<const bf16x8*>(ka + 2 * KBUF + (t + 1) * 2048 + rf0);
                fv[1] = *reinterpret_cast<const bf16x8*>(ka + 2 * KBUF + (t + 1) * 2048 + rf1);
              }
            }
            if (t > 0) {
              bf16x4 pk;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(st[t - 1][r], c2, nlse));
                const float ds = pv * (dp[t - 1][r] - delta);
                dsacc[t - 1][r] += ds;
                pk[r] = f2bf(ds);
              }
              *reinterpret_cast<bf16x4*>(ex_slot(kt0 + t - 1)) = pk;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          if constexpr (NT > 0) frags(0, fr[0]);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            if (t + 1 < NT) frags(t + 1, fr[(t + 1) & 1]);
            mfmas(t);
            __builtin_amdgcn_sched_barrier(0xF);   // (ALU and MFMA instructions may cross, memory instructions may not)
          }
        }
        if constexpr (!PRE) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pv = __builtin_amdgcn_exp2f(fmaf(st[t][r], c2, nlse));
              st[t][r] = pv;
              dpart = fmaf(pv, dp[t][r], dpart);
            }
          }
          dpart = group4_sum(dpart);
          if (lg == 0) dred[(qt * 4 + kw) * 16 + lr] = dpart;
        }
      }
      if (DBG && dbe) dbe[3] = wall_clock64();
      if constexpr (!QL) if (more) fetch_q(b + 1);  // (here, not at the top: this entry's fragments are dead now and lend their registers)
      if constexpr (!PRE) {
        lds_barrier();  // delta partials are in
        if (DBG && dbe) dbe[4] = wall_clock64();
        if constexpr (ACT) {
          // delta_i = sum_j P_ij dP_ij from the SAME P and dP that form dS, so that sum_j dS_ij = 0 holds to fp32 rounding
          delta = (dred[(qt * 4 + 0) * 16 + lr] + dred[(qt * 4 + 1) * 16 + lr]) + (dred[(qt * 4 + 2) * 16 + lr] + dred[(qt * 4 + 3) * 16 + lr]);
          if (kw == 0 && lg == 0 && qvalid) pdelta[stat_off] = delta;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            bf16x4 pk;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float ds = st[t][r] * (dp[t][r] - delta);
              dsacc[t][r] += ds;
              pk[r] = f2bf(ds);
            }
            *reinterpret_cast<bf16x4*>(ex_slot(kt0 + t)) = pk;
          }
        }
      } else if (DBG && dbe) dbe[4] = wall_clock64();
      if (DBG && dbe) dbe[5] = wall_clock64();
      lds_barrier();  // the query tile's dS tiles of all keys are in
      if (DBG && dbe) dbe[6] = wall_clock64();
      if constexpr (ACT) {
        // dQ^T[d, q] = sum_keys K^T[d, key] dS^T[key, q] for d-tile kw, two key tiles per MFMA.  Straight-line over NP pairs (tiles past
        // the last one hold zeros) so that the LDS reads of several pairs are in flight together; two chains of dependent MFMAs.
        const char* kb = sK + tro;
        f32x4 acc2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s2 = 0; s2 < NP; ++s2) {
          const bf16x8 pf = *reinterpret_cast<const bf16x8*>(ex_q + s2 * 1024);
          union { struct { s16x4 a, b; } s; bf16x8 v; } kf;
          kf.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, kb + (2 * s2) * 2048));
          kf.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, kb + (2 * s2 + 1) * 2048));
          acc2[s2 & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf.v, pf, acc2[s2 & 1], 0, 0, 0);
          if ((s2 & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // four pairs' fragments in flight at a time
        }
        const f32x4 acc = acc2[0] + acc2[1];
#pragma unroll
        for (int r = 0; r < 4; ++r) dq_hold[r] = f2bf(acc[r] * a.scale);
      }
      if (DBG && dbe) dbe[7] = wall_clock64();
      advance();
      pdelta += stat_bs;
      pdq += dq_bs;
    }
  };
  if (!wave_active) walk(std::integral_constant<int, -1>{});
  else switch (nt) {
    case 4: walk(std::integral_constant<int, 4>{}); break;
    case 3: walk(std::integral_constant<int, 3>{}); break;
    case 2: walk(std::integral_constant<int, 2>{}); break;
    case 1: walk(std::integral_constant<int, 1>{}); break;
    default: walk(std::integral_constant<int, 0>{}); break;
  }
  if (qvalid && b_begin < b_end) *reinterpret_cast<bf16x4*>(pdq + dq_off) = dq_hold;

  if (a.dbias != nullptr) {  // flush sum_b dS: a wave-private LDS transpose makes every atomic wave-instruction one run of keys of one row
    __syncthreads();
    float* fl = reinterpret_cast<float*>(lds + w * 4096);  // [16 q][64 keys], aliases the K buffers (done with)
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(fl + lr * 64 + t * 16 + 4 * lg) = dsacc[t];
    __syncthreads();
    if (wave_active && b_begin < b_end) {
      const int kj = kt0 * 16 + lane;
      // a.dbias_ws != NULL (XFM_DETERMINISTIC=1): this batch slice's sums go to its own plane [slice][H][Sq][ld] with plain stores,
      // every column below ld written (zero past the last key), and dbias_reduce_kernel adds the planes in slice order; else one float
      // atomic per element and slice straight into dbias
      float* const plane = a.dbias_ws != nullptr ? a.dbias_ws + ((long)zslice * a.H + h) * sq * a.bias_ld : nullptr;
      for (int row = 0; row < 16; ++row) {
        const int q = q0 + row;
        if (q >= sq || lane >= nt * 16) continue;
        if (plane != nullptr) {
          if (kj < a.bias_ld) plane[(long)q * a.bias_ld + kj] = kj < sk ? fl[row * 64 + lane] : 0.f;
        } else if (kj < sk) {
          atomicAdd(a.dbias + ((long)h * sq + q) * a.bias_ld + kj, fl[row * 64 + lane]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward 2/2 for the same short, dense, unmasked problems: dK, dV.  Mirror image of the kernel above: a workgroup is (group of
// <= 3 key tiles, head), its 12 waves are (key tile, query range), and it walks batch entries with Q, dO and the row statistics
// (log-sum-exp, delta) of the NEXT entry landing in second LDS buffers while this one computes.  A wave holds its key tile's K / V
// fragments as the B operands (fetched one entry ahead), computes S and dP for its <= 4 query tiles once, hands P and dS to its
// three sibling waves through LDS (bf16, 8 B per lane and tile) and sums d-tile `qw` of dV^T = dO^T P and dK^T = Q^T dS over all
// queries.  Its bias tile [<= 64 queries x 16 keys] is batch-invariant: 16 VGPRs, loaded once.
// LDS: Q, dO (2 images of 2 NP tiles each) | P, dS exchange (3 x 2 NP tiles x 512 B each) | statistics (2 x 2 x 1 KB).
// NP = query-tile pairs the dK / dV loops run over (tiles past the last query hold zeros in the exchange and finite rows in the
// images): 4 for Sq <= 128, 7 for Sq <= 224 (157 KB of LDS; longer sequences take the general kernel).
// ---------------------------------------------------------------------------------------------
#define VK_KT 3
#define VK_IMG(NP) (2 * (NP) * 2048)
#define VK_EXCH(NP) (VK_KT * 2 * (NP) * 512)
#define VK_LDS(NP) (4 * VK_IMG(NP) + 2 * VK_EXCH(NP) + 4 * 1024)

template <int NP>
__global__ __launch_bounds__(VK_KT * 256) void attn_bwd_dkv_short_kernel(AttnArgs a, int nb_per_block, int G) {
  constexpr int NW = VK_KT * 4, IMG = VK_IMG(NP), EXCH = VK_EXCH(NP), PIECES = 4 * NP;  // 1-KB (8-row) pieces per image
  constexpr int NPC = (PIECES + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: see the dQ kernel)
  const int lr = lane & 15, lg = lane >> 4;
  const int ktl = w >> 2, qw = w & 3;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);  // (see the dQ kernel: the groups of one (head, batch slice) share an XCD)
  const int grp = wg % G, h = (wg / G) % a.H, zslice = wg / (G * a.H);
  const int sk = a.Sk, sq = a.Sq;
  const int QTILES = (sq + 15) >> 4;  // query tiles (<= 2 NP), dealt to the four query-range waves as evenly as they go
  const int qb4 = QTILES >> 2, qr4 = QTILES & 3;
  const int nqt = qb4 + (qw < qr4 ? 1 : 0);
  const int qt0 = qw * qb4 + (qw < qr4 ? qw : qr4);
  const int KT = (sk + 15) >> 4;  // key tiles dealt to the G groups the same way
  const int kb_t = KT / G, kr_t = KT % G;
  const int nkt = kb_t + (grp < kr_t ? 1 : 0);
  const int k0 = (grp * kb_t + (grp < kr_t ? grp : kr_t) + ktl) * 16;
  const bool wave_active = ktl < nkt && k0 < sk;
  const int kj = k0 + lr;
  const bool kvalid = wave_active && kj < sk;
  const int kc = kj < sk ? kj : sk - 1;

  char* const sQ0 = lds;                // images: Q0 | Q1 | dO0 | dO1
  char* const exP = lds + 4 * IMG;      // exchange: P | dS
  char* const stat0 = exP + 2 * EXCH;   // statistics: [buffer][lse | delta][256]

  // (see the dQ kernel: one per-lane offset per fragment kind, tiles are immediates)
  const int sw_r = (lr >> 1) & 7;
  const int rf0 = lr * 128 + ((lg ^ sw_r) << 4), rf1 = lr * 128 + (((4 + lg) ^ sw_r) << 4);
  const int tr_row = 4 * lg + (lr >> 2), tr_col = qw * 16 + 4 * (lr & 3);
  const int tro = tr_row * 128 + ((((tr_col >> 3) ^ ((tr_row >> 1) & 7))) << 4) + (tr_col & 7) * 2;
  // P / dS exchange of a key tile: query tiles in PAIRS, 16 B per lane and pair (tile 2p in the low, 2p + 1 in the high 8 bytes): the
  // dK / dV loop reads a pair as ONE ds_read_b128 -- the MFMA operand as it stands -- instead of a ds_read2st64_b64 at half the LDS rate
  const int ex_r = ktl * (2 * NP * 512) + lane * 16;
  auto ex_slot = [&](int tile) { return ex_r + ((tile >> 1) << 10) + ((tile & 1) << 3); };

  for (int i = tid; i < 2 * EXCH / 16; i += NW * 64) reinterpret_cast<u32x4*>(exP)[i] = u32x4{0, 0, 0, 0};  // tiles past the last query stay zero

  const float inv_scale = 1.0f / a.scale, c2 = a.scale * 1.44269504088896341f;
  f32x4 bvs[4];  // (bias[q, key] / scale; -1e30 past the last key or query) for lane (lg, lr): queries 16 t + 4 lg + r, key lr
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = (qt0 + t) * 16 + 4 * lg + r;
      float bv = 0.f;
      if (a.bias != nullptr && t < nqt && qi < sq && kj < sk) bv = a.bias[((long)h * sq + qi) * a.bias_ld + kj];
      bvs[t][r] = (qi < sq && kj < sk ? bv : -1.0e30f) * inv_scale;
    }

  const int k_off = (int)(((long)kc * a.k_rs + h * 64 + 8 * lg) * 2), v_off = (int)(((long)kc * a.v_rs + h * 64 + 8 * lg) * 2);
  const int dk_off = (int)(((long)kj * a.dk_rs + h * 64 + qw * 16 + 4 * lg) * 2), dv_off = (int)(((long)kj * a.dv_rs + h * 64 + qw * 16 + 4 * lg) * 2);
  const long k_bs = (long)sk * a.k_rs * 2, v_bs = (long)sk * a.v_rs * 2, dk_bs = (long)sk * a.dk_rs * 2, dv_bs = (long)sk * a.dv_rs * 2;
  const long q_bs = (long)sq * a.q_rs * 2, do_bs = (long)sq * a.do_rs * 2, stat_bs = (long)a.H * a.stat_ld;
  int q_off[NPC], do_off[NPC];
  unsigned pc_dst[NPC];
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int j = w + i * NW;  // piece j: rows 8 j .. 8 j + 7 of the image
    const int r = (j & 7) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz_a(r);
    int gr = (j >> 3) * 64 + r;
    gr = gr < sq ? gr : sq - 1;
    q_off[i] = (int)(((long)gr * a.q_rs + h * 64 + c * 8) * 2);
    do_off[i] = (int)(((long)gr * a.do_rs + h * 64 + c * 8) * 2);
    pc_dst[i] = (unsigned)__builtin_amdgcn_readfirstlane(j * 1024);
  }
  // statistics: waves 0..3 stage 64 log-sum-exps each, waves 4..7 64 deltas each (4 B per lane); rows past Sq repeat the last one
  const int st_q = (w & 3) * 64 + lane;
  const int st_src = (int)((long)h * a.stat_ld + (st_q < sq ? st_q : sq - 1)) * 4;
  const unsigned st_dst = (unsigned)__builtin_amdgcn_readfirstlane(((w >> 2) & 1) * 1024 + (w & 3) * 256);
  auto stage_q = [&](int b, int buf) {
    const char* qb = reinterpret_cast<const char*>(a.q) + (long)b * q_bs;
    const char* db = reinterpret_cast<const char*>(a.dout) + (long)b * do_bs;
    const unsigned dq = (unsigned)(uintptr_t)LDS_PTR(void, sQ0) + (unsigned)buf * IMG, dd = dq + 2 * IMG;
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      if (w + i * NW < PIECES) {  // (inline asm: see stage_rows)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(qb + q_off[i]), "s"(dq + pc_dst[i]) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(db + do_off[i]), "s"(dd + pc_dst[i]) : "memory", "m0");
      }
    }
    if (w < 8) {
      const char* sp = reinterpret_cast<const char*>(w < 4 ? a.lse : a.delta) + (long)b * stat_bs * 4 + st_src;
      const unsigned sd = (unsigned)(uintptr_t)LDS_PTR(void, stat0) + (unsigned)buf * 2048 + st_dst;
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(sp), "s"(sd) : "memory", "m0");
    }
  };

  const int b_begin = zslice * nb_per_block;
  int b_end = b_begin + nb_per_block;
  b_end = b_end < a.B ? b_end : a.B;
  bf16x8 kf0, kf1, vf0, vf1;
  auto fetch_k = [&](int b) {
    const char* kp = reinterpret_cast<const char*>(a.k) + (long)b * k_bs + k_off;
    const char* vp = reinterpret_cast<const char*>(a.v) + (long)b * v_bs + v_off;
    kf0 = *reinterpret_cast<const bf16x8*>(kp);
    kf1 = *reinterpret_cast<const bf16x8*>(kp + 64);
    vf0 = *reinterpret_cast<const bf16x8*>(vp);
    vf1 = *reinterpret_cast<const bf16x8*>(vp + 64);
  };
  if (b_begin < b_end) {
    stage_q(b_begin, 0);
    fetch_k(b_begin);
  }
  for (int b = b_begin; b < b_end; ++b) {
    const int cur = (b - b_begin) & 1;
    const char* sQ = sQ0 + cur * IMG;
    const char* sD = sQ + 2 * IMG;
    const float* sL = reinterpret_cast<const float*>(stat0 + cur * 2048);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (see the dQ kernel)
    asm volatile("" : "+v"(kf0), "+v"(kf1), "+v"(vf0), "+v"(vf1));
    lds_barrier();  // Q(b), dO(b), statistics(b) have landed; every wave is done with entry b-1
    if (b + 1 < b_end) stage_q(b + 1, cur ^ 1);

    // (straight-line code per tile count, one scalar branch: see the dQ kernel)
    auto scores = [&](auto NTc) {
      constexpr int NT = decltype(NTc)::value;
      f32x4 st[NT > 0 ? NT : 1], dp[NT > 0 ? NT : 1];
      const char* qa = sQ + qt0 * 2048;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4 lsv = *reinterpret_cast<const f32x4*>(sL + (qt0 + t) * 16 + 4 * lg);
        dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) st[t][r] = fmaf(-lsv[r], inv_scale, bvs[t][r]);
        st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(qa + t * 2048 + rf0), kf0, st[t], 0, 0, 0);
        st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(qa + t * 2048 + rf1), kf1, st[t], 0, 0, 0);
        dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(qa + 2 * IMG + t * 2048 + rf0), vf0, dp[t], 0, 0, 0);
        dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(qa + 2 * IMG + t * 2048 + rf1), vf1, dp[t], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4 dlv = *reinterpret_cast<const f32x4*>(sL + 256 + (qt0 + t) * 16 + 4 * lg);
        bf16x4 pp, ps;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f(st[t][r] * c2);
          pp[r] = f2bf(pv);
          ps[r] = f2bf(pv * (dp[t][r] - dlv[r]));
        }
        *reinterpret_cast<bf16x4*>(exP + ex_slot(qt0 + t)) = pp;
        *reinterpret_cast<bf16x4*>(exP + EXCH + ex_slot(qt0 + t)) = ps;
      }
    };
    if (wave_active) {
      switch (nqt) {
        case 4: scores(std::integral_constant<int, 4>{}); break;
        case 3: scores(std::integral_constant<int, 3>{}); break;
        case 2: scores(std::integral_constant<int, 2>{}); break;
        case 1: scores(std::integral_constant<int, 1>{}); break;
        default: break;
      }
    }
    if (b + 1 < b_end) fetch_k(b + 1);  // (this entry's K / V fragments are dead: the next ones take their registers)
    lds_barrier();  // P and dS of all queries against this key tile are in
    if (wave_active) {
      // dV^T[d, key] = sum_q dO^T[d, q] P[q, key],  dK^T[d, key] = sum_q Q^T[d, q] dS[q, key]  for d-tile qw, two query tiles per MFMA
      const char* qb = sQ + tro;
      f32x4 av[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, ak[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int s2 = 0; s2 < NP; ++s2) {
        union { struct { s16x4 a, b; } s; bf16x8 v; } qf, df;
        const bf16x8 pfv = *reinterpret_cast<const bf16x8*>(exP + ex_r + s2 * 1024);
        const bf16x8 sfv = *reinterpret_cast<const bf16x8*>(exP + EXCH + ex_r + s2 * 1024);
        df.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, qb + 2 * IMG + (2 * s2) * 2048));
        df.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, qb + 2 * IMG + (2 * s2 + 1) * 2048));
        qf.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, qb + (2 * s2) * 2048));
        qf.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, qb + (2 * s2 + 1) * 2048));
        av[s2 & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df.v, pfv, av[s2 & 1], 0, 0, 0);
        ak[s2 & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf.v, sfv, ak[s2 & 1], 0, 0, 0);
        if ((s2 & 1) == 1) __builtin_amdgcn_sched_barrier(0);  // two pairs' fragments in flight at a time
      }
      if (kvalid) {
        bf16x4 ok_, ov_;
#pragma unroll
        for (int r = 0; r < 4; ++r) { ok_[r] = f2bf((ak[0][r] + ak[1][r]) * a.scale); ov_[r] = f2bf(av[0][r] + av[1][r]); }
        *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(a.dk) + (long)b * dk_bs + dk_off) = ok_;
        *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(a.dv) + (long)b * dv_bs + dv_off) = ov_;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward 2/2: dK, dV.  grid (key blocks, H, B); wave w owns keys [kblk*16*NW + 16*w, +16); queries stream in chunks
// of 64 (Q and dO staged in LDS, read by rows for S / dP and transposed for dK^T / dV^T).
// ---------------------------------------------------------------------------------------------
template <bool RES, bool PLAIN>
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), nthreads = blockDim.x;
  const int lr = lane & 15, lg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int kvb = a.kv_index ? a.kv_index[b] : b;
  long qbase, kbase, kout;
  int sq, sk, sk_unused;
  q_seq(a, b, qbase, sq);
  k_seq(a, kvb, kbase, sk);
  k_seq(a, b, kout, sk_unused);  // dk / dv rows belong to the QUERY batch entry (kv_index folds them afterwards)
  const int k0 = (blockIdx.x * (nthreads >> 6) + w) * 16;
  const bool wave_active = k0 < sk;
  const int kj = k0 + lr;
  const bool kvalid = kj < sk;
  const int kcl = kvalid ? kj : sk - 1;
  const bf16* kp = a.k + (kbase + kcl) * a.k_rs + h * 64;
  const bf16* vp = a.v + (kbase + kcl) * a.v_rs + h * 64;
  (void)sk_unused;
  const bf16x8 kf0 = *reinterpret_cast<const bf16x8*>(kp + 8 * lg);
  const bf16x8 kf1 = *reinterpret_cast<const bf16x8*>(kp + 32 + 8 * lg);
  const bf16x8 vf0 = *reinterpret_cast<const bf16x8*>(vp + 8 * lg);
  const bf16x8 vf1 = *reinterpret_cast<const bf16x8*>(vp + 32 + 8 * lg);
  const bf16* qb = a.q + qbase * a.q_rs + h * 64;
  const bf16* db = a.dout + qbase * a.do_rs + h * 64;
  const float* lse_b = a.lse + ((long)b * a.H + h) * a.stat_ld;
  const float* del_b = a.delta + ((long)b * a.H + h) * a.stat_ld;
  bool key_masked = false;
  if (!PLAIN && a.key_keep != nullptr) key_masked = a.key_keep[(long)kvb * a.Sk + kcl] == 0;
  const bool causal = !PLAIN && a.causal != 0;

  f32x4 dkacc[4], dvacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { dkacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int nchunks = (sq + 63) / 64;
  constexpr bool resident = RES;
  const int nw = nthreads >> 6;
  // row statistics / transposed bias of this lane's 4 consecutive queries in 32-query step `step`: 16-B loads
  auto load_stats = [&](int step, f32x4 (&l)[2], f32x4 (&d)[2], f32x4 (&bt)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int qi0 = step * 32 + u * 16 + 4 * lg;
      l[u] = d[u] = bt[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (qi0 < a.Sq) {
        l[u] = *reinterpret_cast<const f32x4*>(lse_b + qi0);
        d[u] = *reinterpret_cast<const f32x4*>(del_b + qi0);
        if (a.bias_t != nullptr && kvalid) bt[u] = *reinterpret_cast<const f32x4*>(a.bias_t + ((long)h * a.Sk + kj) * a.bias_t_ld + qi0);
      }
    }
  };
  if (resident) {
    for (int qc = 0; qc < nchunks; ++qc) stage_slot(lds + qc * ATTN_SLOT, qb, a.q_rs, db, a.do_rs, qc * 64, sq, w, nw, lane);
    stage_wait();
  } else {
    stage_slot(lds, qb, a.q_rs, db, a.do_rs, 0, sq, w, nw, lane);
  }
  _Pragma("unroll 1") for (int qc = 0; qc < nchunks; ++qc) {
    if (!resident) {
      stage_wait();
      if (qc + 1 < nchunks) stage_slot(lds + ((qc + 1) & 1) * ATTN_SLOT, qb, a.q_rs, db, a.do_rs, (qc + 1) * 64, sq, w, nw, lane);
    }
    const char* sQ = lds + (resident ? qc : (qc & 1)) * ATTN_SLOT;
    const char* sD = sQ + ATTN_TILE;
    if (!wave_active) continue;
    // one 32-query k-step at a time (two 16-query tiles): D[i = query row][j = key col], lane (lg, lr) -> query
    // 16t + 4lg + r, key lr.  Half the live registers of a whole-chunk formulation, so two workgroups fit per CU.
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      f32x4 st[2], dp[2], pd[2], lsev[2], delv[2], bvt[2];
      load_stats(qc * 2 + s2, lsev, delv, bvt);  // issued early: their latency hides under the MFMAs below
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * s2 + u;
        st[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        st[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sQ, t * 16, 0, lr, lg), kf0, st[u], 0, 0, 0);
        st[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sQ, t * 16, 1, lr, lg), kf1, st[u], 0, 0, 0);
        dp[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sD, t * 16, 0, lr, lg), vf0, dp[u], 0, 0, 0);
        dp[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sD, t * 16, 1, lr, lg), vf1, dp[u], 0, 0, 0);
      }
      if (a.bias != nullptr && a.bias_t == nullptr) {  // no transposed bias copy: strided gather (slow path, wave-uniform)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qi = qc * 64 + (2 * s2 + u) * 16 + 4 * lg + r;
            if (qi < a.Sq && kvalid) bvt[u][r] = a.bias[((long)h * a.Sq + qi) * a.bias_ld + kj];
          }
      }
      const float key_add = key_masked ? MASK_NEG : 0.f;
      const bool tail = qc * 64 + 64 > sq;  // wave-uniform: this chunk holds rows past the last query
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int qi0 = qc * 64 + (2 * s2 + u) * 16 + 4 * lg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qi = qi0 + r;
          float sc = fmaf(st[u][r], a.scale, bvt[u][r]);
          if (!PLAIN) sc += (causal & (kj > qi)) ? MASK_NEG : key_add;  // masked once, whichever reason (xroberta.py:772-807)
          float pv = __expf(sc - lsev[u][r]);
          float dl = delv[u][r];
          if (tail) {  // the statistics past the last query are unwritten padding: select, never multiply
            pv = qi < sq ? pv : 0.f;
            dl = qi < sq ? dl : 0.f;
          }
          pv = kvalid ? pv : 0.f;
          delv[u][r] = dl;
          pd[u][r] = pv;
          st[u][r] = pv * (dp[u][r] - dl);
        }
      }
      if (!PLAIN && a.drop_thresh != 0u) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qi = qc * 64 + (2 * s2 + u) * 16 + 4 * lg + r;
            const float keepf = drop_keep(a, drop_key(a, b, h, qi), kj) ? a.drop_scale : 0.f;
            st[u][r] = pd[u][r] * (dp[u][r] * keepf - delv[u][r]);
            pd[u][r] *= keepf;
          }
      }
      // dV^T[d, key] += dO^T[d, q] . Pd[q, key] ;  dK^T[d, key] += Q^T[d, q] . dS[q, key]
      const bf16x8 pf = pack_pair(pd[0], pd[1]);
      const bf16x8 sf = pack_pair(st[0], st[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dvacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sD, 32 * s2, 32 * s2 + 16, dt * 16, lr, lg), pf, dvacc[dt], 0, 0, 0);
        dkacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sQ, 32 * s2, 32 * s2 + 16, dt * 16, lr, lg), sf, dkacc[dt], 0, 0, 0);
      }
    }
  }
  if (!wave_active || !kvalid) return;
  bf16* dkp = a.dk + (kout + kj) * a.dk_rs + h * 64;
  bf16* dvp = a.dv + (kout + kj) * a.dv_rs + h * 64;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    bf16x4 ok_, ov_;
#pragma unroll
    for (int r = 0; r < 4; ++r) { ok_[r] = f2bf(dkacc[dt][r] * a.scale); ov_[r] = f2bf(dvacc[dt][r]); }
    *reinterpret_cast<bf16x4*>(dkp + dt * 16 + 4 * lg) = ok_;
    *reinterpret_cast<bf16x4*>(dvp + dt * 16 + 4 * lg) = ov_;
  }
}

// ---------------------------------------------------------------------------------------------
// Grouped cross-attention (fusion towers: Sq = 30 text queries against Sk = 197 image tokens, xroberta.py:201-289 with
// encoder_hidden_states).  Several query batch rows read the SAME key/value source (XFM's ITM negatives and MLM pass reuse
// the batch's images, xfm.py:749-802), so a workgroup is one (source, head): K/V are staged once and stay LDS-resident
// while the 8 waves walk every (row, 16-query tile) of the group; dK/dV are accumulated over the group's rows in
// registers and written once per SOURCE (no per-row copies, no fold pass).  group g = rows grp_rows[grp_start[g] ..
// grp_start[g+1]) and reads source g.  Sq <= 64, Sk <= 256, no additive bias, no causal mask.
// ---------------------------------------------------------------------------------------------
// PACK: the same wave layout serves small self-attention (Sq, Sk <= 64, e.g. the 30-token text rows): a workgroup takes nw/tq
// CONSECUTIVE batch rows, each with its own key/value source in its own LDS slot -- 8 waves per workgroup instead of 2.
// MASK / DROP (grouped mode only; the packed mode keeps its run-time switches): key-keep flags / dropout present.  Grouped mode works in
// the exponent of 2 like its backward kernels: scores scaled by scale*log2(e), the key's additive term (mask, past-the-end) from an LDS
// vector.
template <bool PACK, bool MASK, bool DROP>
__global__ __launch_bounds__(512, 4) void xattn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int g = blockIdx.z, h = blockIdx.y;
  const int tq = (a.Sq + 15) / 16, rpp = nw / tq;  // waves per row, rows per pass
  int rstart, nrows;
  if (PACK) {
    rstart = g * rpp;
    nrows = a.B - rstart < rpp ? a.B - rstart : rpp;
  } else {
    rstart = a.grp_start[g];
    nrows = a.grp_start[g + 1] - rstart;
  }
  if (nrows <= 0) return;  // uniform: before any barrier
  const int nchunks = PACK ? 1 : (a.Sk + 63) / 64;
  if (PACK) {
    // the (start, length) pairs of the <= ATTN_RES_MAX rows first, so their scalar loads overlap instead of one load -> stage
    // chain per row
    long kb_[ATTN_RES_MAX];
    int sk_[ATTN_RES_MAX];
#pragma unroll
    for (int jj = 0; jj < ATTN_RES_MAX; ++jj) {
      const int row = rstart + (jj < nrows ? jj : 0);
      const int src = a.kv_index ? a.kv_index[row] : row;
      k_seq(a, src, kb_[jj], sk_[jj]);
    }
#pragma unroll
    for (int jj = 0; jj < ATTN_RES_MAX; ++jj)
      if (jj < nrows)
        stage_slot(lds + jj * ATTN_SLOT, a.k + kb_[jj] * a.k_rs + h * 64, a.k_rs, a.v + kb_[jj] * a.v_rs + h * 64, a.v_rs, 0, sk_[jj], w, nw, lane);
  } else {
    const bf16* kb = a.k + (long)g * a.Sk * a.k_rs + h * 64;
    const bf16* vb = a.v + (long)g * a.Sk * a.v_rs + h * 64;
    for (int kc = 0; kc < nchunks; ++kc) stage_slot(lds + kc * ATTN_SLOT, kb, a.k_rs, vb, a.v_rs, kc * 64, a.Sk, w, nw, lane);
  }
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  float* madd = reinterpret_cast<float*>(lds + nchunks * ATTN_SLOT);  // grouped + MASK: see xattn_dq_kernel
  if (!PACK && tid < nchunks * 64)
    madd[tid] = tid < a.Sk ? (MASK && a.key_keep[(long)g * a.Sk + tid] == 0 ? MASK_NEG * LOG2E : 0.f) : -3.0e38f;
  stage_wait();
  const int jr = w / tq, tile = w - jr * tq;
  const bool has_mask = a.key_keep != nullptr;
  const float c2 = a.scale * LOG2E;
  const bool causal = PACK && a.causal != 0;
  if (jr >= rpp) return;  // no barriers below
  for (int j = jr; j < nrows; j += rpp) {
    const int b = PACK ? rstart + j : a.grp_rows[rstart + j];
    const int kvb = PACK ? (a.kv_index ? a.kv_index[b] : b) : g;
    long qbase, kbase_unused;
    int sq, sk = a.Sk;
    q_seq(a, b, qbase, sq);
    if (tile * 16 >= sq) continue;  // packed rows: this 16-query tile lies past the sequence's end (wave-uniform, no barrier below)
    if (PACK) k_seq(a, kvb, kbase_unused, sk);
    const int qi = tile * 16 + lr;
    const int qc = qi < sq ? qi : sq - 1;
    const uint32_t dkey = drop_key(a, b, h, qi);
    const bf16* qp = a.q + (qbase + qc) * a.q_rs + h * 64;
    const bf16x8 qf0 = *reinterpret_cast<const bf16x8*>(qp + 8 * lg);
    const bf16x8 qf1 = *reinterpret_cast<const bf16x8*>(qp + 32 + 8 * lg);
    f32x4 oacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = EXCL_NEG, l_run = 0.f;
    for (int kc = 0; kc < nchunks; ++kc) {
      const char* sK = lds + (PACK ? j : kc) * ATTN_SLOT;
      const char* sV = sK + ATTN_TILE;
      f32x4 st[4];
      int kk[4][4];
      if (PACK && has_mask) load_keep(a, kvb, kc, lg, kk);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sK, t * 16, 0, lr, lg), qf0, st[t], 0, 0, 0);
        st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sK, t * 16, 1, lr, lg), qf1, st[t], 0, 0, 0);
      }
      float mx = EXCL_NEG;
      if constexpr (!PACK) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f32x4 ma = *reinterpret_cast<const f32x4*>(madd + kc * 64 + t * 16 + 4 * lg);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            st[t][r] = fmaf(st[t][r], c2, ma[r]);
            mx = fmaxf(mx, st[t][r]);
          }
        }
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            st[t][r] = score_masked(a, st[t][r], 0.f, has_mask, has_mask ? kk[t][r] : 1, causal, qi, kc * 64 + t * 16 + 4 * lg + r, sk);
            mx = fmaxf(mx, st[t][r]);
          }
      }
      mx = group4_max(mx);
      const float m_new = fmaxf(m_run, mx);
      const float alpha = PACK ? __expf(m_run - m_new) : __builtin_amdgcn_exp2f(m_run - m_new);
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          st[t][r] = PACK ? __expf(st[t][r] - m_new) : __builtin_amdgcn_exp2f(st[t][r] - m_new);
          psum += st[t][r];
        }
      if (PACK ? a.drop_thresh != 0u : DROP) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            st[t][r] = drop_keep(a, dkey, kc * 64 + t * 16 + 4 * lg + r) ? st[t][r] * a.drop_scale : 0.f;
      }
      psum = group4_sum(psum);
      l_run = l_run * alpha + psum;
      m_run = m_new;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) oacc[dt][r] *= alpha;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = pack_pair(st[2 * s2], st[2 * s2 + 1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sV, 32 * s2, 32 * s2 + 16, dt * 16, lr, lg), pf, oacc[dt], 0, 0, 0);
      }
    }
    if (qi < sq) {
      store_out(a, qbase + qi, h, lg, oacc, 1.0f / l_run);
      if (lg == 0) a.lse[((long)b * a.H + h) * a.stat_ld + qi] = PACK ? m_run + __logf(l_run) : m_run * LN2 + __logf(l_run);
    }
  }
}

// MASK / DROP: key-keep flags / dropout present (compiled out otherwise: the packed fusion tower has dropout and no mask).  The
// probabilities are taken in the exponent of 2 (one FMA with scale*log2(e) and -lse*log2(e) + the key's additive term, then v_exp), and
// the dropout decisions of the first sweep (delta) are kept as 16 bits per chunk and
// lane for the second (dS): the counter hash -- two quarter-rate integer multiplies per score -- was half of this kernel's VALU time.
template <bool MASK, bool DROP>
__global__ __launch_bounds__(512, 4) void xattn_dq_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int g = blockIdx.z, h = blockIdx.y;
  const int rstart = a.grp_start[g], nrows = a.grp_start[g + 1] - rstart;
  if (nrows <= 0) return;
  const int kvb = g;
  const bf16* kb = a.k + (long)kvb * a.Sk * a.k_rs + h * 64;
  const bf16* vb = a.v + (long)kvb * a.Sk * a.v_rs + h * 64;
  const int nchunks = (a.Sk + 63) / 64;
  constexpr float LOG2E = 1.4426950408889634f;
  for (int kc = 0; kc < nchunks; ++kc) stage_slot(lds + kc * ATTN_SLOT, kb, a.k_rs, vb, a.v_rs, kc * 64, a.Sk, w, nw, lane);
  // what a key adds to every score of its column, in the exponent of 2 (-10000 when masked, "minus infinity" past the last key), once
  // per workgroup in LDS behind the K / V slots: one ds_read_b128 + 4 adds per 16-key tile instead of 16 mask registers and selects
  float* madd = reinterpret_cast<float*>(lds + nchunks * ATTN_SLOT);
  if (tid < nchunks * 64)
    madd[tid] = tid < a.Sk ? (MASK && a.key_keep[(long)kvb * a.Sk + tid] == 0 ? MASK_NEG * LOG2E : 0.f) : -3.0e38f;
  stage_wait();
  const int tq = (a.Sq + 15) / 16, rpp = nw / tq;
  const int jr = w / tq, tile = w - jr * tq;
  if (jr >= rpp) return;
  const float c2 = a.scale * LOG2E;
  for (int j = jr; j < nrows; j += rpp) {
    const int b = a.grp_rows[rstart + j];
    long qbase;
    int sq;
    q_seq(a, b, qbase, sq);
    if (tile * 16 >= sq) continue;  // nothing of this tile belongs to the sequence
    const int qi = tile * 16 + lr;
    const bool qvalid = qi < sq;
    const int qc = qvalid ? qi : sq - 1;
    const uint32_t dkey = DROP ? drop_key(a, b, h, qi) : 0u;
    const bf16* qp = a.q + (qbase + qc) * a.q_rs + h * 64;
    const bf16* dop = a.dout + (qbase + qc) * a.do_rs + h * 64;
    const bf16x8 qf0 = *reinterpret_cast<const bf16x8*>(qp + 8 * lg);
    const bf16x8 qf1 = *reinterpret_cast<const bf16x8*>(qp + 32 + 8 * lg);
    const bf16x8 df0 = *reinterpret_cast<const bf16x8*>(dop + 8 * lg);
    const bf16x8 df1 = *reinterpret_cast<const bf16x8*>(dop + 32 + 8 * lg);
    const long stat_idx = ((long)b * a.H + h) * a.stat_ld + qc;
    const float nlse2 = qvalid ? -a.lse[stat_idx] * LOG2E : -3.0e38f;   // rows past the sequence: every probability 0
    uint32_t keep_lo = 0u, keep_hi = 0u;  // dropout decisions of chunks 0,1 / 2,3: bit (t*4 + r) of the 16-bit field (kc & 1)
    // FIRST: the sweep that draws the dropout decisions (and stores them); later sweeps read them back
    auto probs = [&](int kc, f32x4 (&st)[4], f32x4 (&dp)[4], bool first) {
      const char* sK = lds + kc * ATTN_SLOT;
      const char* sV = sK + ATTN_TILE;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sK, t * 16, 0, lr, lg), qf0, st[t], 0, 0, 0);
        st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sK, t * 16, 1, lr, lg), qf1, st[t], 0, 0, 0);
        dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sV, t * 16, 0, lr, lg), df0, dp[t], 0, 0, 0);
        dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sV, t * 16, 1, lr, lg), df1, dp[t], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 ma = *reinterpret_cast<const f32x4*>(madd + kc * 64 + t * 16 + 4 * lg);
#pragma unroll
        for (int r = 0; r < 4; ++r) st[t][r] = __builtin_amdgcn_exp2f(fmaf(st[t][r], c2, nlse2 + ma[r]));
      }
      if (DROP) {
        const int sh = (kc & 1) * 16;
        uint32_t bits;
        if (first) {
          bits = 0u;
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) bits |= drop_keep(a, dkey, kc * 64 + t * 16 + 4 * lg + r) ? (1u << (t * 4 + r)) : 0u;
          if (kc < 2) keep_lo |= bits << sh;
          else keep_hi |= bits << sh;
        } else {
          bits = (kc < 2 ? keep_lo : keep_hi) >> sh;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) dp[t][r] = (bits & (1u << (t * 4 + r))) ? dp[t][r] * a.drop_scale : 0.f;
      }
    };
    float delta = 0.f;
    f32x4 st[4], dp[4];
    const bool fast_delta = a.o_lo != nullptr;
    if (fast_delta) {
      delta = delta_from_out(a, qbase + qc, h, lg, df0, df1);
    } else {
      for (int kc = 0; kc < nchunks; ++kc) {
        probs(kc, st, dp, true);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) delta += st[t][r] * dp[t][r];
      }
      delta = group4_sum(delta);
    }
    if (qvalid && lg == 0) a.delta[stat_idx] = delta;
    f32x4 dqacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dqacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kc = 0; kc < nchunks; ++kc) {
      if (nchunks > 1 || fast_delta) probs(kc, st, dp, fast_delta);
      const char* sK = lds + kc * ATTN_SLOT;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[t][r] = st[t][r] * (dp[t][r] - delta);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = pack_pair(st[2 * s2], st[2 * s2 + 1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          dqacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sK, 32 * s2, 32 * s2 + 16, dt * 16, lr, lg), pf, dqacc[dt], 0, 0, 0);
      }
    }
    if (qvalid) {
      bf16* dqp = a.dq + (qbase + qi) * a.dq_rs + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = f2bf(dqacc[dt][r] * a.scale);
        *reinterpret_cast<bf16x4*>(dqp + dt * 16 + 4 * lg) = ov;
      }
    }
  }
}

// dK/dV of one (source, head): wave w owns 16 keys; the group's rows go through LDS four at a time (one 64-slot query chunk
// per row, Sq <= 64), gradients accumulate in registers across ALL rows and are written once, at the source's rows.
// DROP: the dropout stream is keyed per query row (two hash rounds) and a lane walks query rows here, so the row keys of the staged
// sequences are computed once per workgroup into LDS (behind the query slots) instead of once per score; probabilities in the
// exponent of 2 as in the dQ kernel.
template <bool DROP>
__global__ __launch_bounds__(1024) void xattn_dkv_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int g = blockIdx.z, h = blockIdx.y;
  const int rstart = a.grp_start[g], nrows = a.grp_start[g + 1] - rstart;
  const int kvb = g;
  const int k0 = (blockIdx.x * nw + w) * 16;
  const bool wave_active = k0 < a.Sk;
  const int kj = k0 + lr;
  const bool kvalid = kj < a.Sk;
  const int kcl = kvalid ? kj : a.Sk - 1;
  const bf16* kp = a.k + ((long)kvb * a.Sk + kcl) * a.k_rs + h * 64;
  const bf16* vp = a.v + ((long)kvb * a.Sk + kcl) * a.v_rs + h * 64;
  const bf16x8 kf0 = *reinterpret_cast<const bf16x8*>(kp + 8 * lg);
  const bf16x8 kf1 = *reinterpret_cast<const bf16x8*>(kp + 32 + 8 * lg);
  const bf16x8 vf0 = *reinterpret_cast<const bf16x8*>(vp + 8 * lg);
  const bf16x8 vf1 = *reinterpret_cast<const bf16x8*>(vp + 32 + 8 * lg);
  constexpr float LOG2E = 1.4426950408889634f;
  const float c2 = a.scale * LOG2E;
  // what this lane's key adds to its scores in the exponent of 2: -10000 when masked, "minus infinity" for a lane past the last key
  const float key_add2 = !kvalid ? -3.0e38f : (a.key_keep != nullptr && a.key_keep[(long)kvb * a.Sk + kcl] == 0) ? MASK_NEG * LOG2E : 0.f;
  // per staged sequence and query row, behind the query slots: dropout row key | -lse * log2(e) | delta  ([ATTN_RES_MAX][64] each)
  uint32_t* rowkeys = reinterpret_cast<uint32_t*>(lds + ATTN_RES_MAX * ATTN_SLOT);
  float* nlse2s = reinterpret_cast<float*>(rowkeys + ATTN_RES_MAX * 64);
  float* deltas = nlse2s + ATTN_RES_MAX * 64;
  f32x4 dkacc[4], dvacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { dkacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  for (int j0 = 0; j0 < nrows; j0 += ATTN_RES_MAX) {  // nrows is workgroup-uniform: every wave takes the same barriers
    const int nb = nrows - j0 < ATTN_RES_MAX ? nrows - j0 : ATTN_RES_MAX;
    __syncthreads();  // readers of the previous batch are done
    for (int jj = 0; jj < nb; ++jj) {
      const int b = a.grp_rows[rstart + j0 + jj];
      long qbase;
      int sq;
      q_seq(a, b, qbase, sq);
      stage_slot(lds + jj * ATTN_SLOT, a.q + qbase * a.q_rs + h * 64, a.q_rs, a.dout + qbase * a.do_rs + h * 64, a.do_rs, 0, sq, w, nw, lane);
    }
    if (tid < nb * 64) {
      const int jj = tid >> 6, qi = tid & 63;
      const int b = a.grp_rows[rstart + j0 + jj];
      if (DROP) rowkeys[tid] = drop_key(a, b, h, qi);
      const long si = ((long)b * a.H + h) * a.stat_ld + qi;
      nlse2s[tid] = qi < a.Sq ? a.lse[si] * -LOG2E : 0.f;
      deltas[tid] = qi < a.Sq ? a.delta[si] : 0.f;
    }
    stage_wait();
    if (!wave_active) continue;
    for (int jj = 0; jj < nb; ++jj) {
      const int b = a.grp_rows[rstart + j0 + jj];
      const char* sQ = lds + jj * ATTN_SLOT;
      const char* sD = sQ + ATTN_TILE;
      const int sq = a.q_len != nullptr ? a.q_len[b] : a.Sq;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if (s2 * 32 >= a.Sq) continue;  // uniform: no query in this half (Sq = 30 lives in the first)
        f32x4 st[2], pd[2];
        if (s2 * 32 >= sq) continue;  // uniform per row: the whole 32-query step lies past the sequence's end
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = 2 * s2 + u;
          const int qi0 = t * 16 + 4 * lg;
          st[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          f32x4 dpu = f32x4{0.f, 0.f, 0.f, 0.f};
          if (t * 16 < sq) {  // (an empty second tile contributes zero probabilities below: skip its four products)
            st[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sQ, t * 16, 0, lr, lg), kf0, st[u], 0, 0, 0);
            st[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sQ, t * 16, 1, lr, lg), kf1, st[u], 0, 0, 0);
            dpu = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sD, t * 16, 0, lr, lg), vf0, dpu, 0, 0, 0);
            dpu = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sD, t * 16, 1, lr, lg), vf1, dpu, 0, 0, 0);
          }
          u32x4 rk = u32x4{0u, 0u, 0u, 0u};
          if (DROP) rk = *reinterpret_cast<const u32x4*>(rowkeys + jj * 64 + qi0);
          const f32x4 nl = *reinterpret_cast<const f32x4*>(nlse2s + jj * 64 + qi0);
          const f32x4 dlv = *reinterpret_cast<const f32x4*>(deltas + jj * 64 + qi0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = qi0 + r < sq;
            float pv = __builtin_amdgcn_exp2f(fmaf(st[u][r], c2, nl[r] + key_add2));
            pv = ok ? pv : 0.f;
            const float dl = ok ? dlv[r] : 0.f;
            float keepf = 1.f;
            if (DROP) keepf = rng_keep(rng_u32(rk[r], (uint32_t)kj), a.drop_thresh) ? a.drop_scale : 0.f;
            pd[u][r] = pv * keepf;
            st[u][r] = pv * (dpu[r] * keepf - dl);
          }
        }
        const bf16x8 pf = pack_pair(pd[0], pd[1]);
        const bf16x8 sf = pack_pair(st[0], st[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dvacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sD, 32 * s2, 32 * s2 + 16, dt * 16, lr, lg), pf, dvacc[dt], 0, 0, 0);
          dkacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sQ, 32 * s2, 32 * s2 + 16, dt * 16, lr, lg), sf, dkacc[dt], 0, 0, 0);
        }
      }
    }
  }
  if (!wave_active || !kvalid) return;
  bf16* dkp = a.dk + ((long)kvb * a.Sk + kj) * a.dk_rs + h * 64;  // per SOURCE (zeros when the group is empty)
  bf16* dvp = a.dv + ((long)kvb * a.Sk + kj) * a.dv_rs + h * 64;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    bf16x4 ok_, ov_;
#pragma unroll
    for (int r = 0; r < 4; ++r) { ok_[r] = f2bf(dkacc[dt][r] * a.scale); ov_[r] = f2bf(dvacc[dt][r]); }
    *reinterpret_cast<bf16x4*>(dkp + dt * 16 + 4 * lg) = ok_;
    *reinterpret_cast<bf16x4*>(dvp + dt * 16 + 4 * lg) = ov_;
  }
}

// ---------------------------------------------------------------------------------------------
// Grouped cross-attention with STREAMED keys (Sk > 256: the 577 image tokens of 384-px retrieval fine-tuning, the 901 of 480-px VQA,
// model_retrieval.py:25-36, model_generation.py:93-130): the image's K / V no longer fit LDS whole, so the 64-key chunks go through a
// two-slot ring and the chunk loop is the OUTER one -- every (row, 16-query tile) of the group keeps its running maximum, sum and
// output (forward) or its dQ (backward) in registers across the chunks, up to XS_SLOTS per wave; a group with more than 8 * XS_SLOTS
// tiles takes another pass over the chunks.  One workgroup per (source, head), 32 KB of LDS: several workgroups share a CU and cover
// each other's staging.  dK / dV come from xattn_dkv_kernel above, which already walks any number of keys.
// ---------------------------------------------------------------------------------------------
#define XS_SLOTS 2
#define XS_RING 3   // K | V chunks of 64 keys in a 3-slot ring filled by inline-asm direct-to-LDS loads (round 4; two slots + the
// compiler-visible builtin before: hipcc drains a visible LDS-DMA in front of every LDS read, so each chunk paid its whole fetch latency)
__device__ __forceinline__ void xs_wait_vm(int n) {  // wave-uniform
  if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// this wave's two 1-KiB pieces (rows 8 w .. 8 w + 7 of the K tile and of the V tile) of chunk kc -> ring slot kc mod 3 (8 waves)
__device__ __forceinline__ void xs_stage(char* lds, const bf16* kb, long k_rs, const bf16* vb, long v_rs, int kc, int sk, int w, int lane) {
  char* slot = lds + (kc % XS_RING) * ATTN_SLOT;
  const int r = w * 8 + (lane >> 3);
  const int c = (lane & 7) ^ swz_a(r);
  int gr = kc * 64 + r;
  gr = gr < sk ? gr : sk - 1;
  const bf16* s0 = kb + (long)gr * k_rs + c * 8;
  const bf16* s1 = vb + (long)gr * v_rs + c * 8;
  const unsigned d0 = (unsigned)(uintptr_t)LDS_PTR(void, slot) + (unsigned)__builtin_amdgcn_readfirstlane(w * 1024);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(s0), "s"(d0) : "memory", "m0");
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(s1), "s"(d0 + (unsigned)ATTN_TILE) : "memory", "m0");
}
__device__ __forceinline__ void xs_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__global__ __launch_bounds__(512, 2) void xattn_fwd_stream_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int g = blockIdx.z, h = blockIdx.y;
  const int rstart = a.grp_start[g], nrows = a.grp_start[g + 1] - rstart;
  if (nrows <= 0) return;
  const int tq = (a.Sq + 15) / 16, n_slots = nrows * tq, nchunks = (a.Sk + 63) / 64;
  const bf16* kb = a.k + (long)g * a.Sk * a.k_rs + h * 64;
  const bf16* vb = a.v + (long)g * a.Sk * a.v_rs + h * 64;
  const bool has_mask = a.key_keep != nullptr;
  for (int base = 0; base < n_slots; base += nw * XS_SLOTS) {   // workgroup-uniform: every wave takes the same barriers
    bf16x8 qf[XS_SLOTS][2];
    f32x4 oacc[XS_SLOTS][4];
    float m_run[XS_SLOTS], l_run[XS_SLOTS];
    int qi_[XS_SLOTS], sq_[XS_SLOTS], b_[XS_SLOTS];
    long qb_[XS_SLOTS];
    uint32_t dkey[XS_SLOTS];
    bool ok[XS_SLOTS];
#pragma unroll
    for (int i = 0; i < XS_SLOTS; ++i) {
      const int s = base + w + nw * i;
      const int j = s / tq, tile = s - j * tq;
      ok[i] = s < n_slots;
      b_[i] = a.grp_rows[rstart + (ok[i] ? j : 0)];
      q_seq(a, b_[i], qb_[i], sq_[i]);
      ok[i] = ok[i] && tile * 16 < sq_[i];
      qi_[i] = tile * 16 + lr;
      const int qc = qi_[i] < sq_[i] ? qi_[i] : sq_[i] - 1;
      const bf16* qp = a.q + (qb_[i] + qc) * a.q_rs + h * 64;
      qf[i][0] = *reinterpret_cast<const bf16x8*>(qp + 8 * lg);
      qf[i][1] = *reinterpret_cast<const bf16x8*>(qp + 32 + 8 * lg);
      dkey[i] = drop_key(a, b_[i], h, qi_[i]);
      m_run[i] = EXCL_NEG;
      l_run[i] = 0.f;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) oacc[i][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // (the compiler's wait for the fragments above lands here, not inside the chunk loop where it would drain the ring)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < XS_SLOTS; ++i) asm volatile("" : "+v"(qf[i][0]), "+v"(qf[i][1]));
    xs_stage(lds, kb, a.k_rs, vb, a.v_rs, 0, a.Sk, w, lane);
    if (nchunks > 1) xs_stage(lds, kb, a.k_rs, vb, a.v_rs, 1, a.Sk, w, lane);
    for (int kc = 0; kc < nchunks; ++kc) {
      xs_wait_vm(kc + 1 < nchunks ? 2 : 0);  // this wave's pieces of chunk kc are in (chunk kc + 1 may still fly)
      xs_barrier();                          // ... everyone's; everyone is done with chunk kc - 1, whose slot chunk kc + 2 takes
      if (kc + 2 < nchunks) xs_stage(lds, kb, a.k_rs, vb, a.v_rs, kc + 2, a.Sk, w, lane);
      const char* sK = lds + (kc % XS_RING) * ATTN_SLOT;
      const char* sV = sK + ATTN_TILE;
      int kk[4][4];
      if (has_mask) load_keep(a, g, kc, lg, kk);
#pragma unroll
      for (int i = 0; i < XS_SLOTS; ++i) {
        if (!ok[i]) continue;   // wave-uniform
        f32x4 st[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sK, t * 16, 0, lr, lg), qf[i][0], st[t], 0, 0, 0);
          st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sK, t * 16, 1, lr, lg), qf[i][1], st[t], 0, 0, 0);
        }
        float mx = EXCL_NEG;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            st[t][r] = score_masked(a, st[t][r], 0.f, has_mask, has_mask ? kk[t][r] : 1, false, qi_[i], kc * 64 + t * 16 + 4 * lg + r, a.Sk);
            mx = fmaxf(mx, st[t][r]);
          }
        mx = group4_max(mx);
        const float m_new = fmaxf(m_run[i], mx);
        const float alpha = __expf(m_run[i] - m_new);
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            st[t][r] = __expf(st[t][r] - m_new);
            psum += st[t][r];
          }
        if (a.drop_thresh != 0u) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              st[t][r] = drop_keep(a, dkey[i], kc * 64 + t * 16 + 4 * lg + r) ? st[t][r] * a.drop_scale : 0.f;
        }
        psum = group4_sum(psum);
        l_run[i] = l_run[i] * alpha + psum;
        m_run[i] = m_new;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) oacc[i][dt][r] *= alpha;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pf = pack_pair(st[2 * s2], st[2 * s2 + 1]);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
            oacc[i][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sV, 32 * s2, 32 * s2 + 16, dt * 16, lr, lg), pf, oacc[i][dt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < XS_SLOTS; ++i) {
      if (ok[i] && qi_[i] < sq_[i]) {
        store_out(a, qb_[i] + qi_[i], h, lg, oacc[i], 1.0f / l_run[i]);
        if (lg == 0) a.lse[((long)b_[i] * a.H + h) * a.stat_ld + qi_[i]] = m_run[i] + __logf(l_run[i]);
      }
    }
    __syncthreads();  // the next pass refills the ring
  }
}

// dQ (and delta) with streamed keys: two sweeps over the chunks per pass -- delta_i = sum_j P_ij dP_ij first (the exact two-pass form
// of the kernels above), then dS and dQ -- unless the forward left o_lo (one sweep).
__global__ __launch_bounds__(512, 2) void xattn_dq_stream_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int g = blockIdx.z, h = blockIdx.y;
  const int rstart = a.grp_start[g], nrows = a.grp_start[g + 1] - rstart;
  if (nrows <= 0) return;
  const int tq = (a.Sq + 15) / 16, n_slots = nrows * tq, nchunks = (a.Sk + 63) / 64;
  const bf16* kb = a.k + (long)g * a.Sk * a.k_rs + h * 64;
  const bf16* vb = a.v + (long)g * a.Sk * a.v_rs + h * 64;
  const bool has_mask = a.key_keep != nullptr;
  const bool fast_delta = a.o_lo != nullptr;
  for (int base = 0; base < n_slots; base += nw * XS_SLOTS) {
    bf16x8 qf[XS_SLOTS][2], df[XS_SLOTS][2];
    f32x4 dqacc[XS_SLOTS][4];
    float lse_q[XS_SLOTS], delta[XS_SLOTS];
    int qi_[XS_SLOTS], sq_[XS_SLOTS];
    long qb_[XS_SLOTS], stat_[XS_SLOTS];
    uint32_t dkey[XS_SLOTS];
    bool ok[XS_SLOTS];
#pragma unroll
    for (int i = 0; i < XS_SLOTS; ++i) {
      const int s = base + w + nw * i;
      const int j = s / tq, tile = s - j * tq;
      ok[i] = s < n_slots;
      const int b = a.grp_rows[rstart + (ok[i] ? j : 0)];
      q_seq(a, b, qb_[i], sq_[i]);
      ok[i] = ok[i] && tile * 16 < sq_[i];
      qi_[i] = tile * 16 + lr;
      const bool qvalid = qi_[i] < sq_[i];
      const int qc = qvalid ? qi_[i] : sq_[i] - 1;
      const bf16* qp = a.q + (qb_[i] + qc) * a.q_rs + h * 64;
      const bf16* dop = a.dout + (qb_[i] + qc) * a.do_rs + h * 64;
      qf[i][0] = *reinterpret_cast<const bf16x8*>(qp + 8 * lg);
      qf[i][1] = *reinterpret_cast<const bf16x8*>(qp + 32 + 8 * lg);
      df[i][0] = *reinterpret_cast<const bf16x8*>(dop + 8 * lg);
      df[i][1] = *reinterpret_cast<const bf16x8*>(dop + 32 + 8 * lg);
      stat_[i] = ((long)b * a.H + h) * a.stat_ld + qc;
      lse_q[i] = qvalid ? a.lse[stat_[i]] : 3.0e38f;
      dkey[i] = drop_key(a, b, h, qi_[i]);
      delta[i] = 0.f;
      if (fast_delta && ok[i]) delta[i] = delta_from_out(a, qb_[i] + qc, h, lg, df[i][0], df[i][1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dqacc[i][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (see the forward kernel)
#pragma unroll
    for (int i = 0; i < XS_SLOTS; ++i)
      asm volatile("" : "+v"(qf[i][0]), "+v"(qf[i][1]), "+v"(df[i][0]), "+v"(df[i][1]), "+v"(lse_q[i]), "+v"(delta[i]));
    for (int sweep = fast_delta ? 1 : 0; sweep < 2; ++sweep) {
      xs_stage(lds, kb, a.k_rs, vb, a.v_rs, 0, a.Sk, w, lane);
      if (nchunks > 1) xs_stage(lds, kb, a.k_rs, vb, a.v_rs, 1, a.Sk, w, lane);
      for (int kc = 0; kc < nchunks; ++kc) {
        xs_wait_vm(kc + 1 < nchunks ? 2 : 0);
        xs_barrier();
        if (kc + 2 < nchunks) xs_stage(lds, kb, a.k_rs, vb, a.v_rs, kc + 2, a.Sk, w, lane);
        const char* sK = lds + (kc % XS_RING) * ATTN_SLOT;
        const char* sV = sK + ATTN_TILE;
        int kk[4][4];
        if (has_mask) load_keep(a, g, kc, lg, kk);
#pragma unroll
        for (int i = 0; i < XS_SLOTS; ++i) {
          if (!ok[i]) continue;
          f32x4 st[4], dp[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sK, t * 16, 0, lr, lg), qf[i][0], st[t], 0, 0, 0);
            st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sK, t * 16, 1, lr, lg), qf[i][1], st[t], 0, 0, 0);
            dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sV, t * 16, 0, lr, lg), df[i][0], dp[t], 0, 0, 0);
            dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sV, t * 16, 1, lr, lg), df[i][1], dp[t], 0, 0, 0);
          }
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              st[t][r] = __expf(score_masked(a, st[t][r], 0.f, has_mask, has_mask ? kk[t][r] : 1, false, qi_[i], kc * 64 + t * 16 + 4 * lg + r, a.Sk) - lse_q[i]);
          if (a.drop_thresh != 0u) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                dp[t][r] = drop_keep(a, dkey[i], kc * 64 + t * 16 + 4 * lg + r) ? dp[t][r] * a.drop_scale : 0.f;
          }
          if (sweep == 0) {
            float d = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) d += st[t][r] * dp[t][r];
            delta[i] += d;
          } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) st[t][r] = st[t][r] * (dp[t][r] - delta[i]);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              const bf16x8 pf = pack_pair(st[2 * s2], st[2 * s2 + 1]);
#pragma unroll
              for (int dt = 0; dt < 4; ++dt)
                dqacc[i][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sK, 32 * s2, 32 * s2 + 16, dt * 16, lr, lg), pf, dqacc[i][dt], 0, 0, 0);
            }
          }
        }
      }
      if (sweep == 0) {
#pragma unroll
        for (int i = 0; i < XS_SLOTS; ++i) delta[i] = group4_sum(delta[i]);
      }
      __syncthreads();  // the next sweep / pass refills slot 0
    }
#pragma unroll
    for (int i = 0; i < XS_SLOTS; ++i) {
      if (ok[i] && qi_[i] < sq_[i]) {
        if (lg == 0) a.delta[stat_[i]] = delta[i];
        bf16* dqp = a.dq + (qb_[i] + qi_[i]) * a.dq_rs + h * 64;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          bf16x4 ov;
#pragma unroll
          for (int r = 0; r < 4; ++r) ov[r] = f2bf(dqacc[i][dt][r] * a.scale);
          *reinterpret_cast<bf16x4*>(dqp + dt * 16 + 4 * lg) = ov;
        }
      }
    }
  }
}

#include "attention_vit.hip"

static int attn_check(const AttnArgs& a, bool bwd) {
  XFM_REQUIRE(a.B > 0 && a.H > 0 && a.Sq > 0 && a.Sk > 0, "attention: empty problem B=%d H=%d Sq=%d Sk=%d", a.B, a.H, a.Sq, a.Sk);
  XFM_REQUIRE(a.q_rs % 8 == 0 && a.k_rs % 8 == 0 && a.v_rs % 8 == 0 && a.o_rs % 4 == 0, "attention: row strides must be multiples of 8");
  XFM_REQUIRE(((uintptr_t)a.q % 16) == 0 && ((uintptr_t)a.k % 16) == 0 && ((uintptr_t)a.v % 16) == 0 && ((uintptr_t)a.o % 8) == 0,
              "attention: q/k/v must be 16-byte aligned");
  XFM_REQUIRE(a.bias == nullptr || (a.bias_ld % 4 == 0 && a.bias_ld >= a.Sk), "attention: bias_ld must be a multiple of 4 and >= Sk");
  XFM_REQUIRE(a.B <= 65535 && a.H <= 65535, "attention: B/H exceed grid limits");
  XFM_REQUIRE(a.stat_ld >= a.Sq && a.stat_ld % 4 == 0 && ((uintptr_t)a.lse % 16) == 0, "attention: stat_ld must be a multiple of 4 and >= Sq, lse 16-byte aligned");
  XFM_REQUIRE(a.bias_t == nullptr || (a.bias != nullptr && a.bias_t_ld % 4 == 0 && a.bias_t_ld >= a.Sq), "attention: bad transposed bias");
  XFM_REQUIRE((a.q_start == nullptr) == (a.q_len == nullptr) && (a.k_start == nullptr) == (a.k_len == nullptr), "attention: packed rows need both start and len");
  XFM_REQUIRE((a.q_start == nullptr && a.k_start == nullptr) || (a.bias == nullptr && a.dbias == nullptr && a.kv_index == nullptr),
              "attention: packed rows take no additive bias and no kv_index");
  if (bwd) {
    XFM_REQUIRE(a.bwd_phase >= 0 && a.bwd_phase <= 2, "attention bwd: bwd_phase must be 0, 1 or 2");
    XFM_REQUIRE(a.dout && a.dq && a.dk && a.dv && a.delta && a.lse, "attention bwd: missing buffers");
    XFM_REQUIRE(a.do_rs % 8 == 0 && a.dq_rs % 4 == 0 && a.dk_rs % 4 == 0 && a.dv_rs % 4 == 0, "attention bwd: bad strides");
  }
  return XFM_OK;
}

static bool attn_resident(int S, int nw) { return cdiv(S, 64) <= ATTN_RES_MAX && nw >= 4; }

static size_t attn_lds_bytes(int S, int nw, size_t at_least) {
  static bool attr_set = false;
  if (!attr_set) {
    const int mx = ATTN_RES_MAX * ATTN_SLOT;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<0, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<0, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<4, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    attr_set = true;
  }
  size_t b = (size_t)(attn_resident(S, nw) ? cdiv(S, 64) : 2) * ATTN_SLOT;
  return b > at_least ? b : at_least;
}

static bool attn_plain(const AttnArgs& a) { return a.key_keep == nullptr && a.causal == 0 && a.drop_thresh == 0u; }

static void attn_geom(int S, int& nw, int& blocks, int max_nw = 8) {
  const int tiles = cdiv(S, 16);
  nw = tiles < max_nw ? tiles : max_nw;
  // balance waves over blocks (e.g. 13 tiles -> 2 blocks of 7 waves)
  blocks = cdiv(tiles, nw);
  nw = cdiv(tiles, blocks);
}

// Small self-attention forward (both sequence lengths within one 64-row chunk, no additive bias): the packed kernel puts up to
// four batch rows in one 8-wave workgroup (15 us instead of 18 us at B=256, S=30).  The backward kernels measured the same
// packed or not (they are bound by each wave's dependent load -> MFMA -> exp -> MFMA chain, not by occupancy) and stay
// one row per workgroup.  XFM_ATTN_PACK=0 is the A/B knob.
static bool attn_packable(const AttnArgs& a) {
  static const bool on = getenv("XFM_ATTN_PACK") ? atoi(getenv("XFM_ATTN_PACK")) != 0 : true;
  return on && a.Sq <= 64 && a.Sk <= 64 && a.bias == nullptr && a.bias_t == nullptr && a.B >= 2;
}

static int attn_check_grouped(const AttnArgs& a) {
  XFM_REQUIRE(a.grp_rows != nullptr && a.n_groups > 0 && a.n_groups <= 65535, "grouped attention: grp_rows / n_groups missing");
  XFM_REQUIRE(a.Sq <= 64, "grouped attention needs Sq <= 64 (got %d)", a.Sq);   // (Sk > 256: the streamed-key kernels)
  XFM_REQUIRE(a.bias == nullptr && a.dbias == nullptr && a.causal == 0 && a.kv_index == nullptr,
              "grouped attention: no bias / causal mask / kv_index (the group index IS the key/value source)");
  return XFM_OK;
}

static void attn_grouped_lds() {
  static bool attr_set = false;
  if (!attr_set) {
    const int mx = ATTN_RES_MAX * ATTN_SLOT + 3 * ATTN_RES_MAX * 64 * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_fwd_kernel<false, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_fwd_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_fwd_kernel<false, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_fwd_kernel<false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_fwd_kernel<true, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_dq_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_dq_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_dq_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_dq_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_dkv_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_dkv_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    attr_set = true;
  }
}

int xfm_attn_fwd_impl(const AttnArgs& a, hipStream_t st) {
  int rc = attn_check(a, false);
  if (rc != XFM_OK) return rc;
  if (a.grp_start != nullptr) {
    rc = attn_check_grouped(a);
    if (rc != XFM_OK) return rc;
    attn_grouped_lds();
    if (a.Sk > 64 * ATTN_RES_MAX) {
      hipLaunchKernelGGL(xattn_fwd_stream_kernel, dim3(1, a.H, a.n_groups), dim3(512), (size_t)XS_RING * ATTN_SLOT, st, a);
      return xfm_check_launch("xattn_fwd_stream");
    }
    {
      const dim3 grid(1, a.H, a.n_groups);
      const bool mask = a.key_keep != nullptr, drop = a.drop_thresh != 0u;
      const size_t lds = (size_t)cdiv(a.Sk, 64) * ATTN_SLOT + 1024;
      if (mask && drop) hipLaunchKernelGGL((xattn_fwd_kernel<false, true, true>), grid, dim3(512), lds, st, a);
      else if (mask) hipLaunchKernelGGL((xattn_fwd_kernel<false, true, false>), grid, dim3(512), lds, st, a);
      else if (drop) hipLaunchKernelGGL((xattn_fwd_kernel<false, false, true>), grid, dim3(512), lds, st, a);
      else hipLaunchKernelGGL((xattn_fwd_kernel<false, false, false>), grid, dim3(512), lds, st, a);
    }
    return xfm_check_launch("xattn_fwd");
  }
  if (attn_packable(a)) {
    const int tq = cdiv(a.Sq, 16);
    int rpb = 8 / tq < ATTN_RES_MAX ? 8 / tq : ATTN_RES_MAX;
    if (rpb > a.B) rpb = a.B;
    attn_grouped_lds();
    hipLaunchKernelGGL((xattn_fwd_kernel<true, false, false>), dim3(1, a.H, cdiv(a.B, rpb)), dim3(rpb * tq * 64), (size_t)rpb * ATTN_SLOT, st, a);
    return xfm_check_launch("xattn_fwd<pack>");
  }
  if (attn_vit_shape(a)) return launch_attn_fwd_vit(a, st);
  int nw, blocks;
  static const int fwd_nw = getenv("XFM_ATTN_FWD_NW") ? atoi(getenv("XFM_ATTN_FWD_NW")) : 8;  // tuning knob
  attn_geom(a.Sq, nw, blocks, fwd_nw);
  const dim3 grid(blocks, a.H, a.B), blk(nw * 64);
  const size_t lds = attn_lds_bytes(a.Sk, nw, 0);
  if (attn_resident(a.Sk, nw)) {
    if (attn_plain(a)) hipLaunchKernelGGL((attn_fwd_kernel<true, true>), grid, blk, lds, st, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<true, false>), grid, blk, lds, st, a);
  } else {
    if (attn_plain(a)) hipLaunchKernelGGL((attn_fwd_kernel<false, true>), grid, blk, lds, st, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<false, false>), grid, blk, lds, st, a);
  }
  return xfm_check_launch("attn_fwd");
}

// dbias[h, q, :] += sum_b ws[b, h, q, :]   (fp32, rows of ld floats; one thread per 4 columns, fixed summation order)
__global__ __launch_bounds__(256) void dbias_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dbias, int B, long per_entry) {
  const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= per_entry) return;
  f32x4 acc = *reinterpret_cast<const f32x4*>(dbias + e);
  for (int b = 0; b < B; ++b) acc += *reinterpret_cast<const f32x4*>(ws + (long)b * per_entry + e);
  *reinterpret_cast<f32x4*>(dbias + e) = acc;
}

// the short dense backward pair (attn_bwd_dq_short_kernel / attn_bwd_dkv_short_kernel) takes this problem
static bool attn_short_dq_ok(const AttnArgs& a) {
  static const bool short_env = getenv("XFM_ATTN_SHORT_BWD") ? atoi(getenv("XFM_ATTN_SHORT_BWD")) != 0 : true;  // A/B knob
  return short_env && attn_plain(a) && a.Sk <= 64 * ATTN_RES_MAX && a.q_start == nullptr && a.k_start == nullptr && a.kv_index == nullptr &&
         (a.bias == nullptr || a.bias_ld >= (long)cdiv(a.Sk, 16) * 16) && (a.dbias == nullptr || a.bias_ld >= a.Sk);
}
// its dQ kernel: query groups per (head, batch slice), batch entries per slice; -> number of slices
static int attn_short_dq_slices(const AttnArgs& a, int& groups, int& nb) {
  groups = cdiv(cdiv(a.Sq, 16), 3);
  int z = 256 / (groups * a.H);
  z = z < 1 ? 1 : (z > a.B ? a.B : z);
  nb = cdiv(a.B, z);
  return cdiv(a.B, nb);
}
// XFM_DETERMINISTIC=1: the reductions that still end in float atomics by default because the ordered form costs a launch or a pass
// (the bias gradient of the short attention backward: one plane per batch slice + dbias_reduce_kernel; the embedding gradients) take
// the ordered form.  Read per call.
static bool xfm_deterministic() {
  const char* e = getenv("XFM_DETERMINISTIC");
  return e != nullptr && atoi(e) != 0;
}
// planes for the short dQ kernel's bias gradient: deterministic mode, more than one batch slice, and rows it can cover completely
static bool attn_short_dbias_planes(const AttnArgs& a) {
  int groups, nb;
  return xfm_deterministic() && a.dbias != nullptr && a.bwd_phase != 2 && attn_short_dq_ok(a) && a.bias_ld <= (long)cdiv(a.Sk, 16) * 16 &&
         attn_short_dq_slices(a, groups, nb) > 1;
}
#include "attention_long.hip"

int xfm_attn_bwd_impl(const AttnArgs& a_in, hipStream_t st) {
  AttnArgs a = a_in;
  // the workspace path stores / sums 16-byte pieces of whole bias rows: every column of a row must lie in a key chunk the kernel visits
  if (a.dbias == nullptr || a.bias_ld % 4 != 0 || (long)cdiv(a.Sk, 64) * 64 < a.bias_ld || a.q_start != nullptr) a.dbias_ws = nullptr;
  int rc = attn_check(a, true);
  if (rc != XFM_OK) return rc;
  if (a.grp_start != nullptr) {
    rc = attn_check_grouped(a);
    if (rc != XFM_OK) return rc;
    attn_grouped_lds();
    if (a.bwd_phase != 2) {
      if (a.Sk > 64 * ATTN_RES_MAX) hipLaunchKernelGGL(xattn_dq_stream_kernel, dim3(1, a.H, a.n_groups), dim3(512), (size_t)XS_RING * ATTN_SLOT, st, a);
      else {
        const dim3 grid(1, a.H, a.n_groups);
        const bool mask = a.key_keep != nullptr, drop = a.drop_thresh != 0u;
        const size_t lds = (size_t)cdiv(a.Sk, 64) * ATTN_SLOT + 1024;
        if (mask && drop) hipLaunchKernelGGL((xattn_dq_kernel<true, true>), grid, dim3(512), lds, st, a);
        else if (mask) hipLaunchKernelGGL((xattn_dq_kernel<true, false>), grid, dim3(512), lds, st, a);
        else if (drop) hipLaunchKernelGGL((xattn_dq_kernel<false, true>), grid, dim3(512), lds, st, a);
        else hipLaunchKernelGGL((xattn_dq_kernel<false, false>), grid, dim3(512), lds, st, a);
      }
      rc = xfm_check_launch("xattn_dq");
      if (rc != XFM_OK || a.bwd_phase == 1) return rc;
    }
    int knw, kblocks;
    static const int dkv_nw = getenv("XFM_XATTN_DKV_NW") ? atoi(getenv("XFM_XATTN_DKV_NW")) : 16;  // tuning knob: waves (16-key tiles) per workgroup
    attn_geom(a.Sk, knw, kblocks, dkv_nw);
    const size_t dkv_lds = (size_t)ATTN_RES_MAX * ATTN_SLOT + 3 * ATTN_RES_MAX * 64 * 4;
    if (a.drop_thresh != 0u) hipLaunchKernelGGL(xattn_dkv_kernel<true>, dim3(kblocks, a.H, a.n_groups), dim3(knw * 64), dkv_lds, st, a);
    else hipLaunchKernelGGL(xattn_dkv_kernel<false>, dim3(kblocks, a.H, a.n_groups), dim3(knw * 64), dkv_lds, st, a);
    return xfm_check_launch("xattn_dkv");
  }
  if (attn_vit_split_dbias(a)) {
    AttnArgs a2 = a;
    a2.dbias = nullptr;
    a2.dbias_ws = nullptr;
    rc = launch_attn_bwd_vit3(a2, st);
    if (rc != XFM_OK) return rc;
    return launch_attn_dbias_blocks(a, st);
  }
  if (attn_vit3_shape(a)) return launch_attn_bwd_vit3(a, st);
  // long dense unmasked problems (577 / 901 image tokens): attention_long.hip.  The dK/dV kernel reads the transposed bias copy;
  // without one that half stays on the general kernel.
  const bool long_ok = attn_long_shape(a);
  const bool long_dkv = long_ok && (a.bias == nullptr || a.bias_t != nullptr || long_tiled(a, a.bias_t_tiled));
  if (long_ok) {
    if (a.bwd_phase != 2) {
      rc = launch_attn_bwd_long_dq(a, st);
      if (rc != XFM_OK || a.bwd_phase == 1) return rc;
    }
    if (long_dkv) return launch_attn_bwd_long_dkv(a, st);
    a.bwd_phase = 2;  // (fall through to the general dK/dV kernel)
    a.dbias_ws = nullptr;
  }
  int nw, blocks;
  attn_geom(a.Sq, nw, blocks);
  const bool res = attn_resident(a.Sk, nw);
  const bool plain = attn_plain(a);
  const bool short_dq = attn_short_dq_ok(a);
  float* const short_planes = (a.dbias_ws != nullptr && attn_short_dbias_planes(a)) ? a.dbias_ws : nullptr;
  // the dQ kernel without in-register bias-gradient sums (NKC = 0) runs, and has a workspace to put each entry's dS into
  const bool dbias_via_ws = a.bwd_phase != 2 && !short_dq && !(a.dbias != nullptr && res && plain) && a.dbias != nullptr && a.dbias_ws != nullptr;
  if (!dbias_via_ws) a.dbias_ws = nullptr;
  if (a.bwd_phase == 2) {
    // dK/dV alone: `delta` was written by an earlier phase-1 call
  } else if (short_dq) {
    static bool attr_set = false;
    if (!attr_set) {
#define XFM_DQS_BYTES(NP, PRE) ((PRE) && (NP) <= 7 ? VB_LDS_QL(3, NP) : VB_LDS(3))
#define XFM_DQS_ATTR(NP, PRE) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_short_kernel<3, NP, PRE, false>), hipFuncAttributeMaxDynamicSharedMemorySize, XFM_DQS_BYTES(NP, PRE))
      XFM_DQS_ATTR(4, false); XFM_DQS_ATTR(7, false); XFM_DQS_ATTR(8, false);
      XFM_DQS_ATTR(4, true); XFM_DQS_ATTR(7, true); XFM_DQS_ATTR(8, true);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_short_kernel<3, 7, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, XFM_DQS_BYTES(7, false));
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_short_kernel<3, 7, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, XFM_DQS_BYTES(7, true));
#undef XFM_DQS_ATTR
      attr_set = true;
    }
    // (query groups x heads) workgroups per batch slice; slices so that one round of <= 256 workgroups covers the batch
    // three query tiles (12 waves) per workgroup: four would need 128-VGPR waves (measured: 40 spilled registers) and 161 KB of LDS
    int groups, nb;
    const int slices = attn_short_dq_slices(a, groups, nb);
    const dim3 grid(groups * a.H * slices);
    a.dbias_ws = short_planes;   // (NULL: float atomics into dbias)
    long long* dbg = nullptr;
    {
      const char* dp = getenv("XFM_ATTN_DBG_PTR");   // (read per launch: tools/attn_timeline.py sets it around the one call it wants a timeline of)
      if (dp != nullptr && dp[0] != 0) dbg = reinterpret_cast<long long*>(strtoull(dp, nullptr, 0));   // (| stamping wave, 0..11)
    }
    // NP = key-tile pairs the dQ loop runs over.  XFM_ATTN_SHORT_PRE=1 (opt-in): the row term delta from dO . (O + O_lo) when the forward
    // kept the low half of O -- one barrier and the delta exchange less per entry, Q / dO through LDS, and MEASURED SLOWER (dQ 121 us
    // against 97.5 at B = 128, 197 tokens: profiles/round5_attn_short.md), so the exchange form stays the default
    const char* pe = getenv("XFM_ATTN_SHORT_PRE");   // (read per call: the tests switch it inside one process)
    const bool pre_env = pe != nullptr && atoi(pe) != 0;
    const bool pre = pre_env && a.o != nullptr && a.o_lo != nullptr;
#define XFM_DQS_LAUNCH(NP)                                                                                                      \
  do {                                                                                                                          \
    if (pre) hipLaunchKernelGGL((attn_bwd_dq_short_kernel<3, NP, true, false>), grid, dim3(768), XFM_DQS_BYTES(NP, true), st, a, nb, groups, dbg);    \
    else hipLaunchKernelGGL((attn_bwd_dq_short_kernel<3, NP, false, false>), grid, dim3(768), XFM_DQS_BYTES(NP, false), st, a, nb, groups, dbg);      \
  } while (0)
    if (dbg != nullptr && a.Sk > 128 && a.Sk <= 224) {   // the stamped build exists for the ViT shape only (tools/attn_timeline.py)
      if (pre) hipLaunchKernelGGL((attn_bwd_dq_short_kernel<3, 7, true, true>), grid, dim3(768), XFM_DQS_BYTES(7, true), st, a, nb, groups, dbg);
      else hipLaunchKernelGGL((attn_bwd_dq_short_kernel<3, 7, false, true>), grid, dim3(768), XFM_DQS_BYTES(7, false), st, a, nb, groups, dbg);
    } else if (a.Sk <= 128) XFM_DQS_LAUNCH(4);
    else if (a.Sk <= 224) XFM_DQS_LAUNCH(7);
    else XFM_DQS_LAUNCH(8);
#undef XFM_DQS_LAUNCH
  } else if (a.dbias != nullptr && res && plain) {
    // batch entries whose dS one workgroup sums before touching HBM.  The kernel holds 230+ VGPRs (sum_b dS of four chunks), i.e.
    // one workgroup per CU: of 4 and 8 entries take the one with fewer (rounds of 256 workgroups) x entries, ties to 8
    // (half the atomics) -- B = 64: 192 workgroups x 8 entries beats 384 x 4 (1.5 rounds) by 6 %.
    int nb = a.B >= 8 ? 2 : 1;
    if (a.B >= 32) {
      const long c4 = (long)cdiv(blocks * a.H * cdiv(a.B, 4), 256) * 4, c8 = (long)cdiv(blocks * a.H * cdiv(a.B, 8), 256) * 8;
      nb = c8 <= c4 ? 8 : 4;
    }
    static const int nb_env = getenv("XFM_ATTN_DBIAS_NB") ? atoi(getenv("XFM_ATTN_DBIAS_NB")) : 0;  // tuning knob
    if (nb_env > 0) nb = nb_env;
    hipLaunchKernelGGL((attn_bwd_dq_kernel<4, true, true>), dim3(blocks, a.H, cdiv(a.B, nb)), dim3(nw * 64),
                       attn_lds_bytes(a.Sk, nw, 8 * 4096), st, a, nb);
  } else if (res) {  // (a masked / causal / dropped problem with a bias gradient falls back to per-element atomics here)
    if (plain) hipLaunchKernelGGL((attn_bwd_dq_kernel<0, true, true>), dim3(blocks, a.H, a.B), dim3(nw * 64), attn_lds_bytes(a.Sk, nw, 0), st, a, 1);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<0, true, false>), dim3(blocks, a.H, a.B), dim3(nw * 64), attn_lds_bytes(a.Sk, nw, 0), st, a, 1);
  } else {
    if (plain) hipLaunchKernelGGL((attn_bwd_dq_kernel<0, false, true>), dim3(blocks, a.H, a.B), dim3(nw * 64), attn_lds_bytes(a.Sk, nw, 0), st, a, 1);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<0, false, false>), dim3(blocks, a.H, a.B), dim3(nw * 64), attn_lds_bytes(a.Sk, nw, 0), st, a, 1);
  }
  rc = xfm_check_launch("attn_bwd_dq");
  if (rc != XFM_OK) return rc;
  if (dbias_via_ws) {
    const long per_entry = (long)a.H * a.Sq * a.bias_ld;
    hipLaunchKernelGGL(dbias_reduce_kernel, dim3(cdiv(per_entry / 4, 256)), dim3(256), 0, st, a.dbias_ws, a.dbias, a.B, per_entry);
    rc = xfm_check_launch("dbias_reduce");
    if (rc != XFM_OK) return rc;
  }
  if (a.bwd_phase != 2 && short_dq && short_planes != nullptr) {   // the slices' planes, in slice order
    int groups, nb;
    const int slices = attn_short_dq_slices(a, groups, nb);
    const long per_entry = (long)a.H * a.Sq * a.bias_ld;
    hipLaunchKernelGGL(dbias_reduce_kernel, dim3(cdiv(per_entry / 4, 256)), dim3(256), 0, st, short_planes, a.dbias, slices, per_entry);
    rc = xfm_check_launch("dbias_reduce");
    if (rc != XFM_OK) return rc;
  }
  if (a.bwd_phase == 1) return rc;
  if (short_dq && a.Sq <= 224) {  // (same preconditions as the short dQ kernel; Sq bounded by the LDS images)
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_short_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, VK_LDS(4));
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_short_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, VK_LDS(7));
      attr_set = true;
    }
    const int groups = cdiv(cdiv(a.Sk, 16), VK_KT);
    int z = 256 / (groups * a.H);
    z = z < 1 ? 1 : (z > a.B ? a.B : z);
    const int nb = cdiv(a.B, z);
    const dim3 grid(groups * a.H * cdiv(a.B, nb));
    if (a.Sq <= 128) hipLaunchKernelGGL(attn_bwd_dkv_short_kernel<4>, grid, dim3(VK_KT * 256), VK_LDS(4), st, a, nb, groups);
    else hipLaunchKernelGGL(attn_bwd_dkv_short_kernel<7>, grid, dim3(VK_KT * 256), VK_LDS(7), st, a, nb, groups);
    return xfm_check_launch("attn_bwd_dkv_short");
  }
  attn_geom(a.Sk, nw, blocks);
  const dim3 grid(blocks, a.H, a.B), blk(nw * 64);
  static const bool dkv_res = getenv("XFM_ATTN_DKV_RES") ? atoi(getenv("XFM_ATTN_DKV_RES")) != 0 : true;  // tuning knob
  if (dkv_res && attn_resident(a.Sq, nw)) {
    if (plain) hipLaunchKernelGGL((attn_bwd_dkv_kernel<true, true>), grid, blk, attn_lds_bytes(a.Sq, nw, 0), st, a);
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<true, false>), grid, blk, attn_lds_bytes(a.Sq, nw, 0), st, a);
  } else {
    if (plain) hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, true>), grid, blk, 2 * ATTN_SLOT, st, a);
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, false>), grid, blk, 2 * ATTN_SLOT, st, a);
  }
  return xfm_check_launch("attn_bwd_dkv");
}

// ---------------------------------------------------------------------------------------------
// dst[u, :] = sum over r with index[r] == u of src[r, :]   (bf16 in/out, fp32 accumulation; rows of `len` elements).
// Folds the per-query-row dK/dV of deduplicated key/value sources back onto the unique sources.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rows_index_sum_kernel(const bf16* __restrict__ src, const int* __restrict__ index, int R,
                                                             long len, bf16* __restrict__ dst) {
  const int u = blockIdx.y;
  const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (e >= len) return;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < R; ++r) {
    if (index[r] != u) continue;  // block-uniform
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (long)r * len + e);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += bf2f(v[i]);
  }
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = f2bf(acc[i]);
  *reinterpret_cast<bf16x8*>(dst + (long)u * len + e) = o;
}

int xfm_rows_index_sum_impl(const void* src, const int* index, int R, int U, long len, void* dst, hipStream_t st) {
  XFM_REQUIRE(R > 0 && U > 0 && len > 0 && len % 8 == 0 && U <= 65535, "rows_index_sum: bad shape R=%d U=%d len=%ld", R, U, len);
  hipLaunchKernelGGL(rows_index_sum_kernel, dim3(cdiv(len, 256 * 8), U), dim3(256), 0, st, (const bf16*)src, index, R, len, (bf16*)dst);
  return xfm_check_launch("rows_index_sum");
}

// ---------------------------------------------------------------------------------------------
// relative-position bias: dense[h,i,j] = table[index[i,j], h]  (beit2.py:139-145) and its transpose-scatter gradient
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relpos_gather_kernel(const float* __restrict__ table, const int* __restrict__ index, int H,
                                                            int N, long ld, float* __restrict__ dense, float* __restrict__ dense_t) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)H * N * ld;
  if (t >= total) return;
  const int j = (int)(t % ld);
  const int i = (int)((t / ld) % N);
  const int h = (int)(t / (ld * N));
  dense[t] = (j < N) ? table[(long)index[i * N + j] * H + h] : 0.f;
  if (dense_t != nullptr) dense_t[t] = (j < N) ? table[(long)index[j * N + i] * H + h] : 0.f;  // [h][key i][query j]
}
__global__ __launch_bounds__(256) void relpos_scatter_kernel(const float* __restrict__ ddense, const int* __restrict__ index, int H,
                                                             int N, long ld, float* __restrict__ dtable) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)H * N * N;
  if (t >= total) return;
  const int j = (int)(t % N);
  const int i = (int)((t / N) % N);
  const int h = (int)(t / ((long)N * N));
  atomicAdd(dtable + (long)index[i * N + j] * H + h, ddense[((long)h * N + i) * ld + j]);
}

int xfm_relpos_gather_impl(const float* table, const int* index, int H, int N, long ld, float* dense, float* dense_t, hipStream_t st) {
  XFM_REQUIRE(H > 0 && N > 0 && ld >= N && ld % 4 == 0, "relpos_gather: bad shape H=%d N=%d ld=%ld", H, N, ld);
  const long total = (long)H * N * ld;
  hipLaunchKernelGGL(relpos_gather_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, table, index, H, N, ld, dense, dense_t);
  return xfm_check_launch("relpos_gather");
}
// Gather form of the same gradient: positions (i*N + j) pre-sorted by table entry (order, start[e] .. start[e+1]), one workgroup
// per entry, wave w sums heads w, w+4, ... over the entry's positions -- no atomics (the scatter above piles ~53, and for the
// three cls entries up to 196, colliding fp32 atomics on each of the 732 x H addresses and takes 58 us for 0.5 M elements).
__global__ __launch_bounds__(256) void relpos_gather_grad_kernel(const float* __restrict__ ddense, const int* __restrict__ order,
                                                                 const int* __restrict__ start, int H, int N, long ld,
                                                                 float* __restrict__ dtable) {
  const int e = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int p0 = start[e], p1 = start[e + 1];
  for (int h = w; h < H; h += 4) {
    const float* src = ddense + (long)h * N * ld;
    float t = 0.f;
    for (int p = p0 + lane; p < p1; p += 64) {
      const int pos = order[p];
      t += src[(long)(pos / N) * ld + pos % N];
    }
    t = wave_sum(t);
    if (lane == 0) dtable[(long)e * H + h] += t;
  }
}

// The same gradient for the STANDARD index of a G x G patch grid plus a cls token (build_relative_position_index, beit2.py:92-116):
// entry e = (yi - yj + G - 1)(2G - 1) + (xi - xj + G - 1) for patch query (yi, xi) and patch key (yj, xj); the last three entries are
// cls -> patch, patch -> cls, cls -> cls.  The sorted gather above gives every lane one position of an entry: consecutive positions of
// an entry are ld + 1 floats apart, so each 4-byte read costs a 64-byte sector (144 us per layer at 901 tokens).  Here a workgroup is one
// (head, dy) row of the table and lane = dx: for a query (yi, xi) the lanes read the keys (yi - dy, xi - dx), 2G - 1 CONSECUTIVE
// floats of one bias row (reversed) -- every element of ddense is read once, coalesced.  The four waves split the query rows yi and are
// summed in a fixed order (no atomics: a table entry has one owner).
__global__ __launch_bounds__(256) void relpos_grid_grad_kernel(const float* __restrict__ ddense, int H, int G, long ld, float* __restrict__ dtable) {
  __shared__ float red[4][64];
  const int h = blockIdx.y, row = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int N = G * G + 1, W = 2 * G - 1, nrd = W * W + 3;
  const float* src = ddense + (long)h * N * ld;
  if (row == W) {  // the three cls entries
    float a = 0.f, b = 0.f;
    for (int j = 1 + threadIdx.x; j < N; j += 256) a += src[j];
    for (int i = 1 + threadIdx.x; i < N; i += 256) b += src[(long)i * ld];
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) { red[w][0] = a; red[w][1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
      dtable[(long)(nrd - 3) * H + h] += (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
      dtable[(long)(nrd - 2) * H + h] += (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
      dtable[(long)(nrd - 1) * H + h] += src[0];
    }
    return;
  }
  const int dy = row - (G - 1);
  const int y0 = dy > 0 ? dy : 0, y1 = dy < 0 ? G + dy : G;  // query rows whose key row yi - dy exists
  for (int dx0 = 0; dx0 < W; dx0 += 64) {  // (2G - 1 <= 64 up to a 32 x 32 grid: one trip)
    const int dxl = dx0 + lane, dx = dxl - (G - 1);
    float acc = 0.f;
    for (int yi = y0 + w; yi < y1; yi += 4) {
      const float* rowp = src + (long)(1 + yi * G) * ld + 1 + (yi - dy) * G - dx;  // + xi * ld + xi per query column
#pragma unroll 6
      for (int xi = 0; xi < G; ++xi) {
        const int xj = xi - dx;
        if (dxl < W && xj >= 0 && xj < G) acc += rowp[(long)xi * ld + xi];
      }
    }
    red[w][lane] = acc;
    __syncthreads();
    if (w == 0 && dxl < W) dtable[((long)row * W + dxl) * H + h] += (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    __syncthreads();
  }
}
int xfm_relpos_grid_grad_impl(const float* ddense, int H, int G, long ld, float* dtable, hipStream_t st) {
  XFM_REQUIRE(H > 0 && G > 0 && ld >= (long)G * G + 1, "relpos_grid_grad: bad shape H=%d G=%d ld=%ld", H, G, ld);
  hipLaunchKernelGGL(relpos_grid_grad_kernel, dim3(2 * G, H), dim3(256), 0, st, ddense, H, G, ld, dtable);
  return xfm_check_launch("relpos_grid_grad");
}

int xfm_relpos_scatter_sorted_impl(const float* ddense, const int* order, const int* start, int entries, int H, int N, long ld,
                                   float* dtable, hipStream_t st) {
  XFM_REQUIRE(H > 0 && N > 0 && ld >= N && entries > 0, "relpos_scatter_sorted: bad shape");
  hipLaunchKernelGGL(relpos_gather_grad_kernel, dim3(entries), dim3(256), 0, st, ddense, order, start, H, N, ld, dtable);
  return xfm_check_launch("relpos_scatter_sorted");
}

int xfm_relpos_scatter_impl(const float* ddense, const int* index, int H, int N, long ld, float* dtable, hipStream_t st) {
  XFM_REQUIRE(H > 0 && N > 0 && ld >= N, "relpos_scatter: bad shape");
  const long total = (long)H * N * N;
  hipLaunchKernelGGL(relpos_scatter_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, ddense, index, H, N, ld, dtable);
  return xfm_check_launch("relpos_scatter");
}
