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

template <bool HAS_BIAS>
__global__ __launch_bounds__(VF_NW * 64) void attn_fwd_vit_kernel(AttnArgs a, int ipw) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int S = a.Sq, KT = (S + 15) >> 4, NP = (KT + 1) >> 1;
  const int n_items = a.B * a.H;
  const int it0 = blockIdx.x * ipw;
  int it1 = it0 + ipw;
  it1 = it1 < n_items ? it1 : n_items;
  if (it0 >= it1) return;

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
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(void, lds);
  const long k_bs = (long)S * a.k_rs * 2, v_bs = (long)S * a.v_rs * 2, q_bs = (long)S * a.q_rs * 2;
  const int PC = 4 * NP;  // 1-KB pieces per image: every tile a PV pair touches is staged (rows past S repeat the last row: finite)

  auto stage = [&](int it, int buf) {
    const int h = it / a.B, b = it - h * a.B;
    const char* kb = reinterpret_cast<const char*>(a.k) + (long)b * k_bs + h * 128;
    const char* vb = reinterpret_cast<const char*>(a.v) + (long)b * v_bs + h * 128;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = w + i * VF_NW;
      if (idx < 2 * PC) {
        const int isv = idx >= PC ? 1 : 0;
        const int j = idx - isv * PC;
        const int r = j * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz_a(r);
        const int gr = r < S ? r : S - 1;
        const char* src = (isv ? vb + (long)gr * a.v_rs * 2 : kb + (long)gr * a.k_rs * 2) + c * 16;
        const unsigned dst = lds0 + (unsigned)__builtin_amdgcn_readfirstlane((isv * 2 + buf) * VF_IMG + j * 1024);
        lds_dma16(src, dst);
      }
    }
  };

  // the score accumulators of one query tile start from bias / scale (keys past S: -1e30, i.e. probability 0).  The loads are
  // issued at the top of an item, BEFORE the next item's LDS-DMA: vmcnt retires in order, so a load issued behind the DMA could
  // only be consumed once the whole prefetch has landed.
  auto load_bias = [&](f32x4 (&st)[VF_MAXT], int qt, int h) {
    const int qi = qt * 16 + lr;
    const int qc = qi < S ? qi : S - 1;
#pragma unroll
    for (int t = 0; t < VF_MAXT; ++t) {
      st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (HAS_BIAS && t < KT && qt < KT) st[t] = *reinterpret_cast<const f32x4*>(a.bias + ((long)h * S + qc) * a.bias_ld + t * 16 + 4 * lg);
    }
  };
  auto run_pass = [&](f32x4 (&st)[VF_MAXT], const bf16x8& q0, const bf16x8& q1, int qt, int h, int b, const char* sK, const char* sV) {
    const int qi = qt * 16 + lr;
#pragma unroll
    for (int t = 0; t < VF_MAXT; ++t) {
      if (t < KT) {
        if (HAS_BIAS) st[t] *= inv_scale;
        if (t == KT - 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) st[t][r] = t * 16 + 4 * lg + r < S ? st[t][r] : -1.0e30f;
        }
        st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(sK + t * 2048 + rf0), q0, st[t], 0, 0, 0);
        st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(sK + t * 2048 + rf1), q1, st[t], 0, 0, 0);
      }
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
    }
    if (qi < S) {
      store_out(a, (long)b * S + qi, h, lg, oacc, 1.0f / l);
      if (lg == 0) a.lse[((long)b * a.H + h) * a.stat_ld + qi] = mx * a.scale + __logf(l);
    }
  };

  stage(it0, 0);
  for (int it = it0; it < it1; ++it) {
    const int cur = (it - it0) & 1;
    const int h = it / a.B, b = it - h * a.B;
    const char* sK = lds + cur * VF_IMG;
    const char* sV = lds + (2 + cur) * VF_IMG;
    // Q fragments of this wave's two query tiles (B operand: query on the lane) and their bias rows, straight from global memory
    bf16x8 qf[2][2];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int qi = (w + VF_NW * pass) * 16 + lr;
      const int qc = qi < S ? qi : S - 1;
      const char* qp = reinterpret_cast<const char*>(a.q) + (long)b * q_bs + ((long)qc * a.q_rs + h * 64 + 8 * lg) * 2;
      qf[pass][0] = *reinterpret_cast<const bf16x8*>(qp);
      qf[pass][1] = *reinterpret_cast<const bf16x8*>(qp + 64);
    }
    f32x4 stA[VF_MAXT], stB[VF_MAXT];
    load_bias(stA, w, h);
    load_bias(stB, w + VF_NW, h);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[1][0]), "+v"(qf[1][1]));
#pragma unroll
    for (int t = 0; t < VF_MAXT; ++t) asm volatile("" : "+v"(stA[t]), "+v"(stB[t]));
    lds_barrier_v();  // K(it), V(it) have landed; every wave is done with item it-1's buffers
    if (it + 1 < it1) stage(it + 1, cur ^ 1);
    run_pass(stA, qf[0][0], qf[0][1], w, h, b, sK, sV);
    if (w + VF_NW < KT) run_pass(stB, qf[1][0], qf[1][1], w + VF_NW, h, b, sK, sV);
  }
}

static bool attn_vit_shape(const AttnArgs& a) {
  static const bool on = getenv("XFM_ATTN_VIT") ? atoi(getenv("XFM_ATTN_VIT")) != 0 : true;  // A/B knob: 0 = the general kernels
  return on && a.Sq == a.Sk && a.Sq > 64 && a.Sq <= 16 * VF_MAXT && a.key_keep == nullptr && a.causal == 0 && a.drop_thresh == 0u &&
         a.q_start == nullptr && a.k_start == nullptr && a.kv_index == nullptr && a.grp_start == nullptr &&
         (a.bias == nullptr || a.bias_ld >= (long)cdiv(a.Sk, 16) * 16) && ((uintptr_t)a.bias % 16) == 0;
}

// items per workgroup: one round of <= 256 workgroups (one per CU; the LDS images leave room for one)
static int attn_vit_ipw(int items) { return cdiv(items, 256); }

static int launch_attn_fwd_vit(const AttnArgs& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_vit_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, VF_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_vit_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, VF_LDS);
    attr_set = true;
  }
  const int items = a.B * a.H, ipw = attn_vit_ipw(items);
  const dim3 grid(cdiv(items, ipw)), blk(VF_NW * 64);
  if (a.bias != nullptr) hipLaunchKernelGGL(attn_fwd_vit_kernel<true>, grid, blk, VF_LDS, st, a, ipw);
  else hipLaunchKernelGGL(attn_fwd_vit_kernel<false>, grid, blk, VF_LDS, st, a, ipw);
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
#define VB2_LDS (VB2_OFF_SCR + 7 * 2048)

template <bool HAS_BIAS>
__global__ __launch_bounds__(512) void attn_bwd_vit_kernel(AttnArgs a, int ipw) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int S = a.Sq, KT = (S + 15) >> 4, NP = (KT + 1) >> 1;
  const int n_items = a.B * a.H;
  const int it0 = blockIdx.x * ipw;
  int it1 = it0 + ipw;
  it1 = it1 < n_items ? it1 : n_items;
  if (it0 >= it1) return;

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
    if (a.dbias == nullptr || nt == 0) return;
    float* scr = reinterpret_cast<float*>(lds + VB2_OFF_SCR + w * 2048);
#pragma unroll
    for (int i = 0; i < VB2_QTM; ++i) {
      if (i < KT) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) scr[(4 * lg + r) * 32 + t * 16 + lr] = dsacc[i][t][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int step = 0; step < 8; ++step) {
          const int row = 2 * step + (lane >> 5), col = lane & 31;
          const int q = i * 16 + row, key = kt0 * 16 + col;
          const float v = scr[row * 32 + col];
          if (q < S && key < S && col < nt * 16) atomicAdd(a.dbias + ((long)h * S + q) * a.bias_ld + key, v);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  };

  int cur_h = -1;
  for (int it = it0; it < it1; ++it) {
    const int h = it / a.B, b = it - h * a.B;
    if (h != cur_h) {
      if (cur_h >= 0) {
        flush(cur_h);
#pragma unroll
        for (int i = 0; i < VB2_QTM; ++i) dsacc[i][0] = dsacc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      cur_h = h;
    }
    // ---- stage Q, dO, K, V of this item (every wave is past the previous item's last barrier)
    {
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
    {
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
        if (HAS_BIAS && t < nt && qt < KT) bv[t] = *reinterpret_cast<const f32x4*>(a.bias_t + ((long)h * S + keyc[t]) * a.bias_t_ld + qt * 16 + 4 * lg);
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

#pragma unroll
    for (int p = 0; p <= VB2_NPM; ++p) {
      if (p < NP && nt > 0) {
        // ---- owner: S, dP, P, dS of query tiles 2p, 2p+1 against this wave's keys; dS to the exchange; dV^T, dK^T accumulate
        char* const exb = ex + (p & 1) * VB2_EXCH;
        bf16x4 pp[2][2], ps[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int qt = 2 * p + u;
#pragma unroll
          for (int t = 0; t < 2; ++t) pp[u][t] = ps[u][t] = bf16x4{0, 0, 0, 0};
          __builtin_amdgcn_sched_barrier(0);
          if (qt < VB2_QTM && qt < KT) {
            f32x4 bnx[2];
            load_bias_t(bnx, qt + 1);  // one query tile ahead: its L2 latency hides under this tile's products
            const f32x4 lsv = *reinterpret_cast<const f32x4*>(sL + qt * 16 + 4 * lg);
            const f32x4 dlv = *reinterpret_cast<const f32x4*>(sDl + qt * 16 + 4 * lg);
            const bf16x8 qa0 = *reinterpret_cast<const bf16x8*>(sQ + qt * 2048 + rf0), qa1 = *reinterpret_cast<const bf16x8*>(sQ + qt * 2048 + rf1);
            const bf16x8 da0 = *reinterpret_cast<const bf16x8*>(sD + qt * 2048 + rf0), da1 = *reinterpret_cast<const bf16x8*>(sD + qt * 2048 + rf1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              __builtin_amdgcn_sched_barrier(0);  // (keeps the scheduler from hoisting every LDS fragment of the pair to its top: 160 VGPRs)
              if (t < nt) {
                const char* kp = sK + (kt0 + t) * 2048;
                const char* vp = sV + (kt0 + t) * 2048;
                f32x4 st, dp;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  float bv = bcur[t][r];
                  if (HAS_BIAS && qt == KT - 1) bv = qt * 16 + 4 * lg + r < S ? bv : 0.f;   // (the padding of a bias row is not initialised)
                  bv = kvalid[t] ? bv : -1.0e30f;
                  st[r] = fmaf(bv, inv_scale, -lsv[r]);
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
          __builtin_amdgcn_sched_barrier(0);
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
      if (p >= 1 && p <= NP && w >= 6) {
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
  }
  if (cur_h >= 0) flush(cur_h);
}

static bool attn_vit_bwd_shape(const AttnArgs& a) {
  return attn_vit_shape(a) && a.Sq <= 16 * VB2_QTM && a.bwd_phase == 0 &&
         (a.bias == nullptr || (a.bias_t != nullptr && a.bias_t_ld >= (long)cdiv(a.Sq, 16) * 16 && ((uintptr_t)a.bias_t % 16) == 0)) &&
         (a.dbias == nullptr || a.bias != nullptr) && a.o != nullptr;
}

static int launch_attn_bwd_vit(const AttnArgs& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_vit_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, VB2_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_vit_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, VB2_LDS);
    attr_set = true;
  }
  const int items = a.B * a.H, ipw = attn_vit_ipw(items);
  const dim3 grid(cdiv(items, ipw)), blk(512);
  if (a.bias != nullptr) hipLaunchKernelGGL(attn_bwd_vit_kernel<true>, grid, blk, VB2_LDS, st, a, ipw);
  else hipLaunchKernelGGL(attn_bwd_vit_kernel<false>, grid, blk, VB2_LDS, st, a, ipw);
  return xfm_check_launch("attn_bwd_vit");
}
