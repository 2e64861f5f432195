// Self-attention of the 224-px ViT trunk (beit2.py:126-166: S = Sq = Sk = 197, dense rows, relative-position bias, no mask /
// dropout) as ONE forward kernel and ONE backward kernel whose workgroups walk whole (batch entry, head) problems.  Included by
// attention.hip (shares its LDS image helpers).
//
// A 197-token problem is small enough to live in one CU: K and V (forward) or Q, dO, K and V (backward) of one (b, h) are 25 KB
// each.  The general kernels above split a problem over several workgroups -- every one of them re-stages K / V (2.6 x the
// algorithmic bytes beyond L2, profiles/round2_hbm_traffic.json), the backward computes S and dP in two kernels, and the forward
// rescales an online softmax per 64-key chunk although a whole score row fits in registers.  Here:
//   * forward (attn_fwd_vit_kernel): 7 waves; a wave owns a 16-query tile against ALL keys (<= 14 key tiles = 56 accumulator VGPRs),
//     two tiles per item; the scores start from bias / scale as the MFMA accumulator, one row maximum, p = exp2(fma), no rescale;
//     K / V of the NEXT item land in the second LDS buffers while this one computes;
//   * backward (attn_bwd_vit_kernel): 8 waves; waves 0..6 own 32 keys each ("key on the lane": S = Q K^T and dP = dO V^T leave the
//     MFMA with the key on the lane, so P and dS are already the B operands of dV^T += dO^T P and dK^T += Q^T dS, whose accumulators
//     stay in registers for the whole item); only dS crosses LDS (bf16, once) for dQ^T = K^T dS^T, which wave 7 (and wave 6, whose
//     second key tile is past the sequence) sums over all keys; the bias gradient sum_b dS stays in the owners' registers across the
//     items of one head (104 VGPRs) and leaves through 128-byte runs of float atomics when the head changes.  5 matrix products, each
//     (b, h) read from HBM once.
// Items are (head, batch entry) pairs in head-major order, `ipw` consecutive ones per workgroup: 1536 items = 6 per CU at B = 128.
#define VF_MAXT 14                  // 16-row tiles per image (S <= 224)
#define VF_IMG (VF_MAXT * 2048)     // one [224 x 64] bf16 image
#define VF_LDS (4 * VF_IMG)         // forward: K | K' | V | V'
#define VF_NW 7

__device__ __forceinline__ void lds_barrier_v() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// direct-to-LDS load of one 1-KB piece (8 rows x 128 B): per-lane source address, wave-uniform LDS destination (M0).  Inline asm: see
// stage_rows in attention.hip (the compiler would drain a visible LDS-DMA before every later LDS read).
__device__ __forceinline__ void lds_dma16(const char* src, unsigned dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory", "m0");
}

// KTC: the number of 16-row tiles as a compile-time constant (13 = the 224-px ViT: every tile guard folds and the score / product loops
// become straight-line code the scheduler can pipeline), or 0 = taken from the sequence length at run time (any S <= 224).
// Work items are (head, batch entry) problems.  Which workgroup runs which, and WHEN, decides the DRAM access pattern: a head's slice of
// a token row is 128 bytes of a 4608-byte row of the fused QKV buffer, and workgroups that walk different batch entries read isolated
// 128-byte lines (measured: 2.9 TB/s).  So the H workgroups of a batch SLICE walk the same entries in step -- workgroup (slice s,
// head h) takes entries s*c .. s*c + c - 1 -- and the H heads' lines of a row are requested within the same few microseconds.  The
// entries that do not fill a slice (B - c * slices) go to a few extra workgroups, c items each, entry-major.
struct VitMap { int H, B, slices, c, main_wgs, grid; };
static VitMap vit_map(int B, int H) {
  VitMap m;
  m.H = H; m.B = B;
  m.slices = 256 / H > 0 ? 256 / H : 1;
  // default: head-major flat ranges (measured: the slice-synchronous map is no faster for the forward -- 61.0 vs 58.9 us -- and its
  // leftover workgroups, with several heads each, flush the backward's bias gradient several times: 316 vs 271 us)
  static const int flat = getenv("XFM_ATTN_VIT_FLAT") ? atoi(getenv("XFM_ATTN_VIT_FLAT")) : 1;
  m.c = B / m.slices;
  if (m.c == 0 || flat) {   // fewer entries than slices: one entry per slice
    m.slices = flat ? 0 : B;
    m.c = flat ? cdiv(B * H, 256) : 1;
  }
  m.main_wgs = m.slices * H;
  const int left = (B - m.c * m.slices) * H;
  m.grid = m.main_wgs + cdiv(left, m.c);
  return m;
}
// item i (0 <= i < c) of workgroup wg -> (h, b); false past the end
__device__ __forceinline__ bool vit_item(const VitMap& m, int wg, int i, int& h, int& b) {
  if (wg < m.main_wgs) {
    h = wg % m.H;
    b = (wg / m.H) * m.c + i;
    return true;
  }
  const int e = (wg - m.main_wgs) * m.c + i;   // leftover entries, entry-major (flat mode: everything, head-major)
  if (m.slices == 0) { h = e / m.B; b = e - h * m.B; return e < m.B * m.H; }
  b = m.c * m.slices + e / m.H;
  h = e % m.H;
  return b < m.B;
}

template <bool HAS_BIAS, int KTC, bool TILED>
__global__ __launch_bounds__(VF_NW * 64) void attn_fwd_vit_kernel(AttnArgs a, VitMap vm) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: wave-uniform address math stays on the SALU)
  const int lr = lane & 15, lg = lane >> 4;
  const int S = a.Sq, KT = KTC ? KTC : (S + 15) >> 4, NP = (KT + 1) >> 1;
  const int wg = blockIdx.x;
  int h, b, hn, bn;
  if (!vit_item(vm, wg, 0, h, b)) return;

  const int sw_r = (lr >> 1) & 7;
  const int rf0 = lr * 128 + ((lg ^ sw_r) << 4), rf1 = lr * 128 + (((4 + lg) ^ sw_r) << 4);  // row fragments, k-steps 0 / 1
  const int tr_row = 4 * lg + (lr >> 2);
  int tro[4];  // transposed fragment of d-tile dt (tile-relative; a 16-row tile is 2048 B and the swizzle ignores the tile)
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    const int tr_col = dt * 16 + 4 * (lr & 3);
    tro[dt] = tr_row * 128 + (((tr_col >> 3) ^ ((tr_row >> 1) & 7)) << 4) + (tr_col & 7) * 2;
  }
  const float inv_scale = 1.0f / a.scale, c2 = a.scale * 1.44269504088896341f;
  constexpr bool tiled = TILED;
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(void, lds);
  const long k_bs = (long)S * a.k_rs * 2, v_bs = (long)S * a.v_rs * 2, q_bs = (long)S * a.q_rs * 2;
  const int PC = 4 * NP;  // 1-KB pieces per image: every tile a PV pair touches is staged (rows past S repeat the last row: finite)

  auto stage = [&](int h, int b, int buf) {
    const char* kb = reinterpret_cast<const char*>(a.k) + (long)b * k_bs + h * 128;
    const char* vb = reinterpret_cast<const char*>(a.v) + (long)b * v_bs + h * 128;
    int ln = lane;
    asm volatile("" : "+v"(ln));   // (opaque: the per-piece row offsets are recomputed per item instead of living in 16 VGPRs)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = w + i * VF_NW;
      if (idx < 2 * PC) {
        const int isv = idx >= PC ? 1 : 0;
        const int j = idx - isv * PC;
        const int r = j * 8 + (ln >> 3);
        const int c = (ln & 7) ^ swz_a(r);
        const int gr = r < S ? r : S - 1;
        const char* src = (isv ? vb + (long)gr * a.v_rs * 2 : kb + (long)gr * a.k_rs * 2) + c * 16;
        const unsigned dst = lds0 + (unsigned)__builtin_amdgcn_readfirstlane((isv * 2 + buf) * VF_IMG + j * 1024);
        lds_dma16(src, dst);
      }
    }
  };

  // The bias rows of this wave's two query tiles are batch-invariant: they are loaded when the workgroup's head changes (once per
  // workgroup in the head-major item order), pre-divided by the scale, -1e30 past the last key, and stay in 104 VGPRs.  The score
  // MFMAs read them as their C operand and write the scores to other registers.  (Fetched per item they cost 182 KB of L2 reads per
  // item and CU, 15 us of a 61-us kernel.)
  auto load_bias = [&](f32x4 (&bt)[VF_MAXT], int qt, int h) {
    const int qi = qt * 16 + lr;
    const int qc = qi < S ? qi : S - 1;
#pragma unroll
    for (int t = 0; t < VF_MAXT; ++t) {
      bt[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t < KT && qt < KT) {
        if (HAS_BIAS) {
          if (tiled) bt[t] = *reinterpret_cast<const f32x4*>(a.bias_tiled + (((long)h * KT + qt) * KT + t) * 256 + lane * 4);
          else bt[t] = *reinterpret_cast<const f32x4*>(a.bias + ((long)h * S + qc) * a.bias_ld + t * 16 + 4 * lg) * inv_scale;
        }
        if (t == KT - 1 && !(HAS_BIAS && tiled)) {   // (a tiled bias carries -1e30 past the last key)
#pragma unroll
          for (int r = 0; r < 4; ++r) bt[t][r] = t * 16 + 4 * lg + r < S ? bt[t][r] : -1.0e30f;
        }
      }
    }
  };
  auto run_pass = [&](const f32x4 (&bt)[VF_MAXT], const bf16x8& q0, const bf16x8& q1, int qt, int h, int b, const char* sK, const char* sV) {
    const int qi = qt * 16 + lr;
    f32x4 st[VF_MAXT];
#pragma unroll
    for (int t = 0; t < VF_MAXT; ++t) {
      st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t < KT) {
        st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(sK + t * 2048 + rf0), q0, bt[t], 0, 0, 0);
        st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(sK + t * 2048 + rf1), q1, st[t], 0, 0, 0);
      }
      if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // at most four tiles' K fragments (32 VGPRs) in flight: the bias rows hold 104
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int t = 0; t < VF_MAXT; ++t)
      if (t < KT) mx = fmaxf(mx, fmaxf(fmaxf(st[t][0], st[t][1]), fmaxf(st[t][2], st[t][3])));
    mx = group4_max(mx);
    const float mc = -mx * c2;
    float l = 0.f;
    bf16x8 pf[VF_MAXT / 2];
#pragma unroll
    for (int s2 = 0; s2 < VF_MAXT / 2; ++s2) {
      f32x4 pa = f32x4{0.f, 0.f, 0.f, 0.f}, pb = f32x4{0.f, 0.f, 0.f, 0.f};
      if (2 * s2 < KT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { pa[r] = __builtin_amdgcn_exp2f(fmaf(st[2 * s2][r], c2, mc)); l += pa[r]; }
      }
      if (2 * s2 + 1 < KT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { pb[r] = __builtin_amdgcn_exp2f(fmaf(st[2 * s2 + 1][r], c2, mc)); l += pb[r]; }
      }
      pf[s2] = pack_pair(pa, pb);
    }
    l = group4_sum(l);
    f32x4 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < VF_MAXT / 2; ++s2) {
      if (s2 < NP) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          union { struct { s16x4 a, b; } s; bf16x8 v; } vf;
          vf.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, sV + (2 * s2) * 2048 + tro[dt]));
          vf.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, sV + (2 * s2 + 1) * 2048 + tro[dt]));
          oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf[s2], oacc[dt], 0, 0, 0);
        }
      }
      if (s2 & 1) __builtin_amdgcn_sched_barrier(0);
    }
    if (qi < S) {
      store_out(a, (long)b * S + qi, h, lg, oacc, 1.0f / l);
      if (lg == 0) a.lse[((long)b * a.H + h) * a.stat_ld + qi] = mx * a.scale + __logf(l);
    }
  };

  f32x4 biasA[VF_MAXT], biasB[VF_MAXT];
  int bias_h = -1;
  stage(h, b, 0);
  for (int it = 0; it < vm.c; ++it) {
    const int cur = it & 1;
    const bool more = it + 1 < vm.c && vit_item(vm, wg, it + 1, hn, bn);
    const char* sK = lds + cur * VF_IMG;
    const char* sV = lds + (2 + cur) * VF_IMG;
    // Q fragments of this wave's two query tiles (B operand: query on the lane), straight from global memory
    bf16x8 qf[2][2];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int qi = (w + VF_NW * pass) * 16 + lr;
      const int qc = qi < S ? qi : S - 1;
      const char* qp = reinterpret_cast<const char*>(a.q) + (long)b * q_bs + ((long)qc * a.q_rs + h * 64 + 8 * lg) * 2;
      qf[pass][0] = *reinterpret_cast<const bf16x8*>(qp);
      qf[pass][1] = *reinterpret_cast<const bf16x8*>(qp + 64);
    }
    if (h != bias_h) {   // (issued before the next item's LDS-DMA: vmcnt retires in order)
      load_bias(biasA, w, h);
      load_bias(biasB, w + VF_NW, h);
      bias_h = h;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[1][0]), "+v"(qf[1][1]));
#pragma unroll
    for (int t = 0; t < VF_MAXT; ++t) asm volatile("" : "+v"(biasA[t]), "+v"(biasB[t]));
    lds_barrier_v();  // K(it), V(it) have landed; every wave is done with item it-1's buffers
    if (more) stage(hn, bn, cur ^ 1);
    run_pass(biasA, qf[0][0], qf[0][1], w, h, b, sK, sV);
    if (w + VF_NW < KT) run_pass(biasB, qf[1][0], qf[1][1], w + VF_NW, h, b, sK, sV);
    if (!more) break;
    h = hn;
    b = bn;
  }
}

// One 64-thread workgroup per (head, tile a, tile b): the tile in the accumulator layout of both kernels (see include/xfm_hip.h).
__global__ __launch_bounds__(64) void bias_tile_kernel(const float* __restrict__ bias, int S, long ld, float inv_scale, float* __restrict__ tiled,
                                                       float* __restrict__ tiled_t) {
  const int T = gridDim.x, a_ = blockIdx.y, b_ = blockIdx.x, h = blockIdx.z;
  const int lane = threadIdx.x, lr = lane & 15, lg = lane >> 4;
  const float* bh = bias + (long)h * S * ld;
  const long tile = (((long)h * T + a_) * T + b_) * 256 + lane * 4;
  if (tiled != nullptr) {   // query 16a + lr, keys 16b + 4lg + r
    const int q = a_ * 16 + lr;
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = b_ * 16 + 4 * lg + r;
      v[r] = k < S ? (q < S ? bh[(long)q * ld + k] * inv_scale : 0.f) : -1.0e30f;
    }
    *reinterpret_cast<f32x4*>(tiled + tile) = v;
  }
  if (tiled_t != nullptr) {  // key 16a + lr, queries 16b + 4lg + r
    const int k = a_ * 16 + lr;
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = b_ * 16 + 4 * lg + r;
      v[r] = k < S ? (q < S ? bh[(long)q * ld + k] * inv_scale : 0.f) : -1.0e30f;
    }
    *reinterpret_cast<f32x4*>(tiled_t + tile) = v;
  }
}

int xfm_bias_tile_impl(const float* bias, int H, int S, long ld, float scale, float* tiled, float* tiled_t, hipStream_t st) {
  XFM_REQUIRE(bias != nullptr && H > 0 && S > 0 && ld >= S && scale > 0.f, "bias_tile: bad arguments");
  XFM_REQUIRE(((uintptr_t)tiled % 16) == 0 && ((uintptr_t)tiled_t % 16) == 0, "bias_tile: outputs must be 16-byte aligned");
  const int T = cdiv(S, 16);
  hipLaunchKernelGGL(bias_tile_kernel, dim3(T, T, H), dim3(64), 0, st, bias, S, ld, 1.0f / scale, tiled, tiled_t);
  return xfm_check_launch("bias_tile");
}

static bool attn_vit_shape(const AttnArgs& a) {
  static const bool on = getenv("XFM_ATTN_VIT") ? atoi(getenv("XFM_ATTN_VIT")) != 0 : true;  // A/B knob: 0 = the general kernels
  return on && a.Sq == a.Sk && a.Sq > 64 && a.Sq <= 16 * VF_MAXT && a.key_keep == nullptr && a.causal == 0 && a.drop_thresh == 0u &&
         a.q_start == nullptr && a.k_start == nullptr && a.kv_index == nullptr && a.grp_start == nullptr &&
         (a.bias == nullptr || a.bias_ld >= (long)cdiv(a.Sk, 16) * 16) && ((uintptr_t)a.bias % 16) == 0;
}

// (template instantiation + the one-time dynamic-LDS attribute of a kernel)
template <typename K>
static void vit_launch(K kernel, int lds, dim3 grid, dim3 blk, hipStream_t st, const AttnArgs& a, const VitMap& vm) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(kernel, grid, blk, lds, st, a, vm);
}

static int launch_attn_fwd_vit(const AttnArgs& a, hipStream_t st) {
  const VitMap vm = vit_map(a.B, a.H);
  const dim3 grid(vm.grid), blk(VF_NW * 64);
  const bool k13 = cdiv(a.Sq, 16) == 13;
  if (a.bias == nullptr) {
    if (k13) vit_launch(attn_fwd_vit_kernel<false, 13, false>, VF_LDS, grid, blk, st, a, vm);
    else vit_launch(attn_fwd_vit_kernel<false, 0, false>, VF_LDS, grid, blk, st, a, vm);
  } else if (a.bias_tiled != nullptr) {
    if (k13) vit_launch(attn_fwd_vit_kernel<true, 13, true>, VF_LDS, grid, blk, st, a, vm);
    else vit_launch(attn_fwd_vit_kernel<true, 0, true>, VF_LDS, grid, blk, st, a, vm);
  } else {
    if (k13) vit_launch(attn_fwd_vit_kernel<true, 13, false>, VF_LDS, grid, blk, st, a, vm);
    else vit_launch(attn_fwd_vit_kernel<true, 0, false>, VF_LDS, grid, blk, st, a, vm);
  }
  return xfm_check_launch("attn_fwd_vit");
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward: dQ, dK, dV, delta and the bias gradient of one (b, h) per step of a workgroup's item walk (see the file header).
// LDS: Q | dO | K | V images (28 KB each) | dS exchange, two query-tile pairs of [224 keys][16 q] bf16 x 2 | lse / scale and delta
// (256 floats each) | a 2-KB transposition scratch per owner wave for the bias-gradient flush.
// delta_i = dO_i . O_i (+ o_lo: what bf16 rounding of O lost): one pass over the keys; the general kernels' exact two-pass delta
// (sum_j P_ij dP_ij) is what XFM_ATTN_VIT=0 still runs.
// ---------------------------------------------------------------------------------------------------------------------------
#define VB2_QTM 13                        // query / key tiles (S <= 208): the bias-gradient accumulators are 8 VGPRs per query tile
#define VB2_NPM 7
#define VB2_EXQ (VF_MAXT * 16 * 32)       // one query tile of the exchange: [224 keys][16 q] bf16
#define VB2_EXCH (2 * VB2_EXQ)
#define VB2_OFF_EX (4 * VF_IMG)
#define VB2_OFF_ST (VB2_OFF_EX + 2 * VB2_EXCH)
#define VB2_OFF_SCR (VB2_OFF_ST + 2 * 1024)
#define VB2_OFF_SINK (VB2_OFF_SCR + 7 * 2048)   // 256 B that the L2-prefetch loads land in (never read)
#define VB2_LDS (VB2_OFF_SINK + 256)

template <bool HAS_BIAS, int KTC, bool TILED>
__global__ __launch_bounds__(512) void attn_bwd_vit_kernel(AttnArgs a, VitMap vm, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: wave-uniform address math stays on the SALU)
  const int lr = lane & 15, lg = lane >> 4;
  const int S = a.Sq, KT = KTC ? KTC : (S + 15) >> 4, NP = (KT + 1) >> 1;
  const int wg = blockIdx.x;
  int h, b, hn = 0, bn = 0;
  if (!vit_item(vm, wg, 0, h, b)) return;

  char* const sQ = lds;
  char* const sD = lds + VF_IMG;
  char* const sK = lds + 2 * VF_IMG;
  char* const sV = lds + 3 * VF_IMG;
  char* const ex = lds + VB2_OFF_EX;
  float* const sL = reinterpret_cast<float*>(lds + VB2_OFF_ST);   // lse / scale (1e30 past the last query: P = 0)
  float* const sDl = sL + 256;                                    // delta
  // key tiles past the sequence are never written by an owner: they must read as zeros in the dQ product
  for (int i = tid; i < 2 * VB2_EXCH / 16; i += 512) reinterpret_cast<u32x4*>(ex)[i] = u32x4{0, 0, 0, 0};

  const bool owner = w < 7;
  const int kt0 = 2 * w;
  int nt = KT - kt0;
  nt = !owner || nt < 0 ? 0 : (nt > 2 ? 2 : nt);
  const int sw_r = (lr >> 1) & 7;
  const int rf0 = lr * 128 + ((lg ^ sw_r) << 4), rf1 = lr * 128 + (((4 + lg) ^ sw_r) << 4);
  const int tr_row = 4 * lg + (lr >> 2);
  auto tro_of = [&](int dt) {
    const int tr_col = dt * 16 + 4 * (lr & 3);
    return tr_row * 128 + (((tr_col >> 3) ^ ((tr_row >> 1) & 7)) << 4) + (tr_col & 7) * 2;
  };
  constexpr bool tiled = TILED;
  const int ex_w = (kt0 * 16 + lr) * 32 + lg * 8;              // owner: dS[q = 4 lg .. +3][key = lr] of key tile kt0 (+512 per tile)
  const int ex_r = (4 * lg + (lr >> 2)) * 32 + (lr & 3) * 8;   // dQ: transposed read of a [4 keys][16 q] block (+512 per key tile)
  const float inv_scale = 1.0f / a.scale, c2 = a.scale * 1.44269504088896341f;
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(void, lds);
  const long q_bs = (long)S * a.q_rs * 2, k_bs = (long)S * a.k_rs * 2, v_bs = (long)S * a.v_rs * 2, do_bs = (long)S * a.do_rs * 2;
  const int PC = 4 * NP;  // 1-KB pieces per image (rows past S repeat the last row: finite)
  bool kvalid[2];
  int keyc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int key = (kt0 + t) * 16 + lr;
    kvalid[t] = key < S;
    keyc[t] = key < S ? key : S - 1;
  }

  f32x4 dsacc[VB2_QTM][2];
#pragma unroll
  for (int i = 0; i < VB2_QTM; ++i) dsacc[i][0] = dsacc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  // sum_b dS of this wave's 32 keys leaves through a wave-private LDS transpose: every atomic wave-instruction adds two 128-byte
  // runs (32 keys of two bias rows)
  auto flush = [&](int h) {
    if (a.dbias == nullptr || nt == 0 || (dbg & 4)) return;
    float* scr = reinterpret_cast<float*>(lds + VB2_OFF_SCR + w * 2048);
    const int col = lane & 31, key = kt0 * 16 + col;
    const bool kok = key < S && col < nt * 16;
    float* const dst0 = a.dbias + (long)h * S * a.bias_ld + key;
#pragma unroll
    for (int i = 0; i < VB2_QTM; ++i) {
      if (i < KT) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) scr[(4 * lg + r) * 32 + t * 16 + lr] = dsacc[i][t][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
        for (int step = 0; step < 8; ++step) {   // a real loop: unrolled, the 104 address computations of a flush spill to scratch
          const int row = 2 * step + (lane >> 5);
          const int q = i * 16 + row;
          const float v = scr[row * 32 + col];
          if (kok && q < S) atomicAdd(dst0 + (long)q * a.bias_ld, v);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  };

  int cur_h = -1;
  for (int it = 0; it < vm.c; ++it) {
    const bool more = it + 1 < vm.c && vit_item(vm, wg, it + 1, hn, bn);
    if (h != cur_h) {
      if (cur_h >= 0) {
        flush(cur_h);
#pragma unroll
        for (int i = 0; i < VB2_QTM; ++i) dsacc[i][0] = dsacc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      cur_h = h;
    }
    // ---- stage Q, dO, K, V of this item (every wave is past the previous item's last barrier)
    if (!(dbg & 8) || it == 0) {
      const char* src_b[4] = {reinterpret_cast<const char*>(a.q) + (long)b * q_bs, reinterpret_cast<const char*>(a.dout) + (long)b * do_bs,
                              reinterpret_cast<const char*>(a.k) + (long)b * k_bs, reinterpret_cast<const char*>(a.v) + (long)b * v_bs};
      const long rs_b[4] = {a.q_rs * 2, a.do_rs * 2, a.k_rs * 2, a.v_rs * 2};
#pragma unroll
      for (int img = 0; img < 4; ++img) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int j = w + i * 8;
          if (j < PC) {
            const int r = j * 8 + (lane >> 3);
            const int c = (lane & 7) ^ swz_a(r);
            const int gr = r < S ? r : S - 1;
            lds_dma16(src_b[img] + (long)gr * rs_b[img] + h * 128 + c * 16, lds0 + (unsigned)__builtin_amdgcn_readfirstlane(img * VF_IMG + j * 1024));
          }
        }
      }
    }
    // ---- row statistics: thread pair (2 r, 2 r + 1) owns row r: delta_r = dO_r . (O_r + Olo_r), lse_r / scale
    if (!(dbg & 16) || it == 0) {
      const int row = tid >> 1, half = tid & 1;
      float d = 0.f;
      if (row < S) {
        const bf16* dp_ = a.dout + ((long)b * S + row) * a.do_rs + h * 64 + half * 32;
        const bf16* op_ = a.o + ((long)b * S + row) * a.o_rs + h * 64 + half * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bf16x8 dv = *reinterpret_cast<const bf16x8*>(dp_ + c * 8), ov = *reinterpret_cast<const bf16x8*>(op_ + c * 8);
          if (a.o_lo != nullptr) {
            const bf16x8 lv = *reinterpret_cast<const bf16x8*>(a.o_lo + ((long)b * S + row) * a.o_rs + h * 64 + half * 32 + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) d = fmaf(bf2f(dv[e]), bf2f(ov[e]) + bf2f(lv[e]), d);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) d = fmaf(bf2f(dv[e]), bf2f(ov[e]), d);
          }
        }
      }
      d += __shfl_xor(d, 1, 64);
      if (half == 0) {
        sDl[row] = d;
        if (row < S && a.delta != nullptr) a.delta[((long)b * a.H + h) * a.stat_ld + row] = d;
      } else {
        sL[row] = row < S ? a.lse[((long)b * a.H + h) * a.stat_ld + row] * inv_scale : 1.0e30f;
      }
    }
    // bias of this wave's keys against the first query tile (transposed dense bias: 4 consecutive queries of one key = 16 bytes)
    f32x4 bcur[2];
    auto load_bias_t = [&](f32x4 (&bv)[2], int qt) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (HAS_BIAS && t < nt && qt < KT) {
          if (tiled) bv[t] = *reinterpret_cast<const f32x4*>(a.bias_t_tiled + (((long)h * KT + kt0 + t) * KT + qt) * 256 + lane * 4);
          else bv[t] = *reinterpret_cast<const f32x4*>(a.bias_t + ((long)h * S + keyc[t]) * a.bias_t_ld + qt * 16 + 4 * lg);
        }
      }
    };
    load_bias_t(bcur, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(bcur[0]), "+v"(bcur[1]));
    lds_barrier_v();  // images and statistics of this item are in

    f32x4 dKa[2][4], dVa[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dKa[t][dt] = dVa[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The images leave no LDS for a second set, so the next item cannot be staged under this one's products.  Wave 7 (the dQ wave: no
    // loads of its own to wait for) touches one dword of every 128-byte line the next item will stage or read: they are in L2 when
    // the direct-to-LDS loads ask for them (each lane one line per instruction; the values are never used).
    if (w == 7 && more && !(dbg & 32)) {
      const int h2 = hn, b2 = bn;
      const char* pb[5] = {reinterpret_cast<const char*>(a.q) + (long)b2 * q_bs, reinterpret_cast<const char*>(a.dout) + (long)b2 * do_bs,
                           reinterpret_cast<const char*>(a.k) + (long)b2 * k_bs, reinterpret_cast<const char*>(a.v) + (long)b2 * v_bs,
                           reinterpret_cast<const char*>(a.o) + (long)b2 * S * a.o_rs * 2};
      const long pr[5] = {a.q_rs * 2, a.do_rs * 2, a.k_rs * 2, a.v_rs * 2, a.o_rs * 2};
#pragma unroll
      for (int img = 0; img < 5; ++img) {
#pragma unroll 1
        for (int r0 = 0; r0 < S; r0 += 64) {
          const int r = r0 + lane < S ? r0 + lane : S - 1;
          // (a direct-to-LDS dword: a load into a VGPR would land, some microseconds later, in a register the compiler has reused)
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(pb[img] + (long)r * pr[img] + h2 * 128), "s"(lds0 + VB2_OFF_SINK) : "memory", "m0");
        }
      }
    }

#pragma unroll
    for (int p = 0; p <= VB2_NPM; ++p) {
      if (p < NP && nt > 0 && !(dbg & 2)) {
        // ---- owner: S, dP, P, dS of query tiles 2p, 2p+1 against this wave's keys; dS to the exchange; dV^T, dK^T accumulate
        char* const exb = ex + (p & 1) * VB2_EXCH;
        bf16x4 pp[2][2], ps[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int qt = 2 * p + u;
#pragma unroll
          for (int t = 0; t < 2; ++t) pp[u][t] = ps[u][t] = bf16x4{0, 0, 0, 0};
          if (qt < VB2_QTM && qt < KT) {
            f32x4 bnx[2];
            load_bias_t(bnx, qt + 1);  // one query tile ahead: its L2 latency hides under this tile's products
            const f32x4 lsv = *reinterpret_cast<const f32x4*>(sL + qt * 16 + 4 * lg);
            const f32x4 dlv = *reinterpret_cast<const f32x4*>(sDl + qt * 16 + 4 * lg);
            const bf16x8 qa0 = *reinterpret_cast<const bf16x8*>(sQ + qt * 2048 + rf0), qa1 = *reinterpret_cast<const bf16x8*>(sQ + qt * 2048 + rf1);
            const bf16x8 da0 = *reinterpret_cast<const bf16x8*>(sD + qt * 2048 + rf0), da1 = *reinterpret_cast<const bf16x8*>(sD + qt * 2048 + rf1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              if (t < nt) {
                const char* kp = sK + (kt0 + t) * 2048;
                const char* vp = sV + (kt0 + t) * 2048;
                f32x4 st, dp;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  float bv = bcur[t][r];
                  if (HAS_BIAS && tiled) {   // pre-divided by the scale, -1e30 past the last key, 0 past the last query
                    st[r] = bv - lsv[r];
                  } else {
                    if (HAS_BIAS && qt == KT - 1) bv = qt * 16 + 4 * lg + r < S ? bv : 0.f;   // (the padding of a bias row is not initialised)
                    bv = kvalid[t] ? bv : -1.0e30f;
                    st[r] = fmaf(bv, inv_scale, -lsv[r]);
                  }
                  dp[r] = -dlv[r];
                }
                st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa0, *reinterpret_cast<const bf16x8*>(kp + rf0), st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa1, *reinterpret_cast<const bf16x8*>(kp + rf1), st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da0, *reinterpret_cast<const bf16x8*>(vp + rf0), dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da1, *reinterpret_cast<const bf16x8*>(vp + rf1), dp, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const float pv = __builtin_amdgcn_exp2f(st[r] * c2);
                  const float ds = pv * dp[r];
                  dsacc[qt < VB2_QTM ? qt : 0][t][r] += ds;
                  pp[u][t][r] = f2bf(pv);
                  ps[u][t][r] = f2bf(ds);
                }
                *reinterpret_cast<bf16x4*>(exb + u * VB2_EXQ + ex_w + t * 512) = ps[u][t];
              }
            }
            bcur[0] = bnx[0];
            bcur[1] = bnx[1];
          }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int tro = tro_of(dt);
          union { struct { s16x4 a, b; } s; bf16x8 v; } df, qf;
          df.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, sD + (2 * p) * 2048 + tro));
          df.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, sD + (2 * p + 1) * 2048 + tro));
          qf.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, sQ + (2 * p) * 2048 + tro));
          qf.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, sQ + (2 * p + 1) * 2048 + tro));
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            if (t < nt) {
              union { struct { bf16x4 a, b; } s; bf16x8 v; } pf, sf;
              pf.s.a = pp[0][t]; pf.s.b = pp[1][t];
              sf.s.a = ps[0][t]; sf.s.b = ps[1][t];
              dVa[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df.v, pf.v, dVa[t][dt], 0, 0, 0);
              dKa[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf.v, sf.v, dKa[t][dt], 0, 0, 0);
            }
          }
        }
      }
      if (p >= 1 && p <= NP && w >= 6 && !(dbg & 1)) {
        // ---- dQ^T[d, q] = sum_keys K^T[d, key] dS^T[key, q] of the PREVIOUS pair: wave 7 takes six (query tile, d-tile) units, wave 6
        // (at most 16 keys of its own at S = 197) the other two
        const int pq = p - 1;
        const char* exb = ex + (pq & 1) * VB2_EXCH;
        const int u_begin = w == 7 ? 0 : 6, u_end = w == 7 ? 6 : 8;
#pragma unroll 1
        for (int un = u_begin; un < u_end; ++un) {
          const int u = un >> 2, dt = un & 3, qt = 2 * pq + u;
          if (qt >= KT) continue;
          const int tro = tro_of(dt);
          const char* xb = exb + u * VB2_EXQ + ex_r;
          f32x4 acc2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
          for (int s2 = 0; s2 < VB2_NPM; ++s2) {
            if (s2 < NP) {
              union { struct { s16x4 a, b; } s; bf16x8 v; } kf, sf;
              sf.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, xb + (2 * s2) * 512));
              sf.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, xb + (2 * s2 + 1) * 512));
              kf.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, sK + (2 * s2) * 2048 + tro));
              kf.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, sK + (2 * s2 + 1) * 2048 + tro));
              acc2[s2 & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf.v, sf.v, acc2[s2 & 1], 0, 0, 0);
            }
          }
          const int qi = qt * 16 + lr;
          if (qi < S) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = f2bf((acc2[0][r] + acc2[1][r]) * a.scale);
            *reinterpret_cast<bf16x4*>(a.dq + ((long)b * S + qi) * a.dq_rs + h * 64 + dt * 16 + 4 * lg) = ov;
          }
        }
      }
      lds_barrier_v();
    }
    // ---- dK, dV of this wave's keys
    if (nt > 0) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int key = (kt0 + t) * 16 + lr;
        if (t < nt && key < S) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            bf16x4 ok_, ov_;
#pragma unroll
            for (int r = 0; r < 4; ++r) { ok_[r] = f2bf(dKa[t][dt][r] * a.scale); ov_[r] = f2bf(dVa[t][dt][r]); }
            *reinterpret_cast<bf16x4*>(a.dk + ((long)b * S + key) * a.dk_rs + h * 64 + dt * 16 + 4 * lg) = ok_;
            *reinterpret_cast<bf16x4*>(a.dv + ((long)b * S + key) * a.dv_rs + h * 64 + dt * 16 + 4 * lg) = ov_;
          }
        }
      }
    }
    if (!more) break;
    h = hn;
    b = bn;
  }
  if (cur_h >= 0) flush(cur_h);
}

// Measured at B = 128 (1536 problems, tools/bench_attn_vit.py): 271 us against 234 us for the pair of split kernels (dQ + dK/dV).  The
// single pass does 5 products instead of 7 and reads each problem once, but with the bias-gradient sums resident it needs 256 VGPRs (two
// waves per SIMD) and its four LDS images leave no room to stage the next item under the current one: 15 us of exposed staging per
// item round, 40 us of atomics for the bias gradient, owners bound by VALU issue.  Kept as an opt-in (XFM_ATTN_VIT_BWD=1) with its
// tests; the default backward stays the split pair.
static bool attn_vit_bwd_shape(const AttnArgs& a) {
  const char* e = getenv("XFM_ATTN_VIT_BWD");   // (read per call: the tests switch it inside one process)
  const bool on = e != nullptr && atoi(e) != 0;
  return on && attn_vit_shape(a) && a.Sq <= 16 * VB2_QTM && a.bwd_phase == 0 &&
         (a.bias == nullptr || a.bias_t_tiled != nullptr ||
          (a.bias_t != nullptr && a.bias_t_ld >= (long)cdiv(a.Sq, 16) * 16 && ((uintptr_t)a.bias_t % 16) == 0)) &&
         (a.dbias == nullptr || a.bias != nullptr) && a.o != nullptr;
}

template <typename K>
static void vit_launch_bwd(K kernel, dim3 grid, hipStream_t st, const AttnArgs& a, const VitMap& vm, int dbg) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, VB2_LDS);
  hipLaunchKernelGGL(kernel, grid, dim3(512), VB2_LDS, st, a, vm, dbg);
}

static int launch_attn_bwd_vit(const AttnArgs& a, hipStream_t st) {
  const VitMap vm = vit_map(a.B, a.H);
  const dim3 grid(vm.grid);
  const bool k13 = cdiv(a.Sq, 16) == 13;
  static const int dbg = getenv("XFM_VIT_DBG") ? atoi(getenv("XFM_VIT_DBG")) : 0;  // timing experiments only (results are wrong when set)
  if (a.bias == nullptr) {
    if (k13) vit_launch_bwd(attn_bwd_vit_kernel<false, 13, false>, grid, st, a, vm, dbg);
    else vit_launch_bwd(attn_bwd_vit_kernel<false, 0, false>, grid, st, a, vm, dbg);
  } else if (a.bias_t_tiled != nullptr) {
    if (k13) vit_launch_bwd(attn_bwd_vit_kernel<true, 13, true>, grid, st, a, vm, dbg);
    else vit_launch_bwd(attn_bwd_vit_kernel<true, 0, true>, grid, st, a, vm, dbg);
  } else {
    if (k13) vit_launch_bwd(attn_bwd_vit_kernel<true, 13, false>, grid, st, a, vm, dbg);
    else vit_launch_bwd(attn_bwd_vit_kernel<true, 0, false>, grid, st, a, vm, dbg);
  }
  return xfm_check_launch("attn_bwd_vit");
}
