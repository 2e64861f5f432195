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
//   * backward (attn_bwd_vit3_kernel, opt-in): one pass for dQ, dK, dV, delta and the bias gradient -- see its header for the design and
//     for why the split dQ + dK/dV kernels of attention.hip remain the default.
// Items are (head, batch entry) pairs in head-major order, consecutive ones per workgroup: 1536 items = 6 per CU at B = 128.
#include <type_traits>
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
  const int T = gridDim.x, a_ = blockIdx.y, b_ = blockIdx.x, h = blockIdx.z;   // a_ in [0, T]: tiled_t has one more key-tile row, all -1e30
  const int lane = threadIdx.x, lr = lane & 15, lg = lane >> 4;
  const float* bh = bias + (long)h * S * ld;
  const long tile = (((long)h * T + a_) * T + b_) * 256 + lane * 4;
  if (tiled != nullptr && a_ < T) {   // query 16a + lr, keys 16b + 4lg + r
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
    *reinterpret_cast<f32x4*>(tiled_t + (((long)h * (T + 1) + a_) * T + b_) * 256 + lane * 4) = v;
  }
}

int xfm_bias_tile_impl(const float* bias, int H, int S, long ld, float scale, float* tiled, float* tiled_t, hipStream_t st) {
  XFM_REQUIRE(bias != nullptr && H > 0 && S > 0 && ld >= S && scale > 0.f, "bias_tile: bad arguments");
  XFM_REQUIRE(((uintptr_t)tiled % 16) == 0 && ((uintptr_t)tiled_t % 16) == 0, "bias_tile: outputs must be 16-byte aligned");
  const int T = cdiv(S, 16);
  hipLaunchKernelGGL(bias_tile_kernel, dim3(T, T + 1, H), dim3(64), 0, st, bias, S, ld, 1.0f / scale, tiled, tiled_t);
  return xfm_check_launch("bias_tile");
}

static bool attn_vit_shape(const AttnArgs& a) {
  static const bool on = getenv("XFM_ATTN_VIT") ? atoi(getenv("XFM_ATTN_VIT")) != 0 : true;  // A/B knob: 0 = the general kernels
  return on && a.Sq == a.Sk && a.Sq > 64 && a.Sq <= 16 * VF_MAXT && a.key_keep == nullptr && a.causal == 0 && a.drop_thresh == 0u &&
         a.q_start == nullptr && a.k_start == nullptr && a.kv_index == nullptr && a.grp_start == nullptr &&
         (a.bias == nullptr || (a.bias_ld >= (long)cdiv(a.Sk, 16) * 16 && a.bias_ld % 4 == 0)) &&  // (untiled bias rows are read 16 B at a time)
         ((uintptr_t)a.bias % 16) == 0;
}

// (template instantiation + the one-time dynamic-LDS attribute of a kernel: TAG makes one guard per kernel instantiation)
template <int TAG, typename K>
static void vit_launch(K kernel, int lds, dim3 grid, dim3 blk, hipStream_t st, const AttnArgs& a, const VitMap& vm) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kernel, grid, blk, lds, st, a, vm);
}

static int launch_attn_fwd_vit(const AttnArgs& a, hipStream_t st) {
  const VitMap vm = vit_map(a.B, a.H);
  const dim3 grid(vm.grid), blk(VF_NW * 64);
  const bool k13 = cdiv(a.Sq, 16) == 13;
  if (a.bias == nullptr) {
    if (k13) vit_launch<1>(attn_fwd_vit_kernel<false, 13, false>, VF_LDS, grid, blk, st, a, vm);
    else vit_launch<2>(attn_fwd_vit_kernel<false, 0, false>, VF_LDS, grid, blk, st, a, vm);
  } else if (a.bias_tiled != nullptr) {
    if (k13) vit_launch<3>(attn_fwd_vit_kernel<true, 13, true>, VF_LDS, grid, blk, st, a, vm);
    else vit_launch<4>(attn_fwd_vit_kernel<true, 0, true>, VF_LDS, grid, blk, st, a, vm);
  } else {
    if (k13) vit_launch<5>(attn_fwd_vit_kernel<true, 13, false>, VF_LDS, grid, blk, st, a, vm);
    else vit_launch<6>(attn_fwd_vit_kernel<true, 0, false>, VF_LDS, grid, blk, st, a, vm);
  }
  return xfm_check_launch("attn_fwd_vit");
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward, single pass (S in (192, 208]: 13 tiles, 7 query-tile pairs; opt-in, XFM_ATTN_VIT_BWD=3): dQ, dK, dV, delta and the bias
// gradient of one (b, h) from ONE evaluation of S and dP -- 5 matrix products instead of the 7 of the split dQ + dK/dV kernels, each
// problem read from HBM once.  Waves 0..6 own 32 keys each ("key on the lane": S = Q K^T and dP = dO V^T leave the MFMA with the key on
// the lane, so P and dS are already the B operands of dV^T += dO^T P and dK^T += Q^T dS, whose accumulators stay in registers for the
// whole item); only dS crosses LDS (bf16, once) for dQ^T = K^T dS^T; the bias gradient sum_b dS stays in the owners' registers across the
// items of one head (104 VGPRs) and leaves through 128-byte runs of float atomics when the head changes.  delta_i = dO_i . O_i (bf16 O).
// Nothing of an item is staged while the workgroup waits: K and V images are double-buffered across items; Q, dO and O arrive one
// query-tile PAIR ahead in a two-slot ring (they are only ever needed pair by pair: row fragments for S / dP, transposed fragments for
// dV / dK, dO . O for delta); wave 7 issues every direct-to-LDS load, waits for them behind counted vmcnt, turns dO . O into delta and
// publishes the pair's row statistics before the barrier that opens the pair.  The owner waves issue no LDS-DMA at all, so their own
// loads (bias tiles, one query tile ahead) never queue behind a prefetch.  dQ of a pair is summed one iteration later, one
// (query tile, d-tile) unit per wave, the last pair of an item inside the first iteration of the next one.
// MEASURED (B = 128: 1536 problems, tools/bench_attn_vit.py, tools/pmc_attn_vit.sh): 285 us against 231 us for the split pair, although it
// issues 17 % fewer VALU instructions and keeps the matrix pipe busy for 30 % fewer cycles: with the bias-gradient sums resident a wave
// needs 256 VGPRs, so a CU holds 8 waves (the split kernels: 12), and they spend 72 % of their cycles parked (SQ_WAIT_ANY) -- one
// barrier per query-tile pair whose period is set by wave 7's 22 LDS-DMA issues and their landing, not by the 55 MFMAs of the pair.
// A variant that staged a whole item at its top (no ring) measured the same 287 us.  Kept as an opt-in with its tests; the default
// backward is the split pair.
// LDS: K | K' | V | V' (13 tiles each) | ring: 2 x (Q | dO | O of 32 rows) | exchange: 2 x 2 x [208 keys][16 q] bf16 | statistics.
// ---------------------------------------------------------------------------------------------------------------------------
#define V3_KT 13
#define V3_NP 7
#define V3_IMG (V3_KT * 2048)                 // 26,624
#define V3_SLOT (3 * 4096)                    // Q | dO | O rows of one pair
#define V3_EXQ (V3_KT * 16 * 32)              // 6,656: one query tile of the exchange
#define V3_EXCH (2 * V3_EXQ)
#define V3_OFF_RING (4 * V3_IMG)
#define V3_OFF_EX (V3_OFF_RING + 2 * V3_SLOT)
#define V3_OFF_ST (V3_OFF_EX + 2 * V3_EXCH)    // 2 x { lse / scale [32] | delta [32] }
#define V3_LDS (V3_OFF_ST + 2 * 256)

template <bool HAS_BIAS, bool DBIAS>
__global__ __launch_bounds__(512) void attn_bwd_vit3_kernel(AttnArgs a, VitMap vm) {
  constexpr int KT = V3_KT, NP = V3_NP;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int S = a.Sq;
  const int wg = blockIdx.x;
  int h, b, hn = 0, bn = 0;
  if (!vit_item(vm, wg, 0, h, b)) return;

  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(void, lds);
  char* const ring = lds + V3_OFF_RING;
  char* const ex = lds + V3_OFF_EX;
  float* const st = reinterpret_cast<float*>(lds + V3_OFF_ST);
  const bool owner = w < 7;
  const int kt0 = 2 * w;
  const int sw_r = (lr >> 1) & 7;
  const int rf0 = lr * 128 + ((lg ^ sw_r) << 4), rf1 = lr * 128 + (((4 + lg) ^ sw_r) << 4);
  const int tr_row = 4 * lg + (lr >> 2);
  auto tro_of = [&](int dt) {
    const int tr_col = dt * 16 + 4 * (lr & 3);
    return tr_row * 128 + (((tr_col >> 3) ^ ((tr_row >> 1) & 7)) << 4) + (tr_col & 7) * 2;
  };
  const int ex_w = (kt0 * 16 + lr) * 32 + lg * 8;
  const int ex_r = (4 * lg + (lr >> 2)) * 32 + (lr & 3) * 8;
  const float inv_scale = 1.0f / a.scale, c2 = a.scale * 1.44269504088896341f;
  const long q_bs = (long)S * a.q_rs * 2, k_bs = (long)S * a.k_rs * 2, v_bs = (long)S * a.v_rs * 2, do_bs = (long)S * a.do_rs * 2,
             o_bs = (long)S * a.o_rs * 2;
  bool kvalid[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) kvalid[t] = (kt0 + t) * 16 + lr < S;

  // ---- direct-to-LDS staging (issued by wave 7, except the first item's K / V which every wave shares)
  // one 1-KB piece: rows r0 .. r0 + 7 (clamped to the last row) of a [S, 64]-per-head operand, swizzled like every image here
  auto piece = [&](const char* base, long rs2, int hh, int r0, unsigned dst) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int r = r0 + (ln >> 3);
    const int c = (ln & 7) ^ swz_a(r);
    const int gr = r < S ? r : S - 1;
    lds_dma16(base + (long)gr * rs2 + hh * 128 + c * 16, dst);
  };
  auto stage_kv_pieces = [&](int hh, int bb, int buf, int j0, int j1) {   // pieces j0 .. j1 - 1 of the 52 (K: 0..25, V: 26..51)
    const char* kb = reinterpret_cast<const char*>(a.k) + (long)bb * k_bs;
    const char* vb = reinterpret_cast<const char*>(a.v) + (long)bb * v_bs;
    for (int j = j0; j < j1; ++j) {
      const int isv = j >= 26 ? 1 : 0, jj = j - isv * 26;
      piece(isv ? vb : kb, isv ? a.v_rs * 2 : a.k_rs * 2, hh, jj * 8, lds0 + (unsigned)((isv * 2 + buf) * V3_IMG + jj * 1024));
    }
  };
  // Q | dO | O rows and the log-sum-exps of pair p of item (hh, bb) into ring / statistics slot `slot`: 13 wave-instructions
  auto stage_pair = [&](int hh, int bb, int p, int slot) {
    const char* qb = reinterpret_cast<const char*>(a.q) + (long)bb * q_bs;
    const char* db = reinterpret_cast<const char*>(a.dout) + (long)bb * do_bs;
    const char* ob = reinterpret_cast<const char*>(a.o) + (long)bb * o_bs;
    const unsigned dst = lds0 + V3_OFF_RING + slot * V3_SLOT;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      piece(qb, a.q_rs * 2, hh, p * 32 + j * 8, dst + j * 1024);
      piece(db, a.do_rs * 2, hh, p * 32 + j * 8, dst + 4096 + j * 1024);
      piece(ob, a.o_rs * 2, hh, p * 32 + j * 8, dst + 8192 + j * 1024);
    }
    int row = p * 32 + (lane & 31);
    row = row < S ? row : S - 1;
    const char* lp = reinterpret_cast<const char*>(a.lse + ((long)bb * a.H + hh) * a.stat_ld + row);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(lp), "s"(lds0 + V3_OFF_ST + slot * 256) : "memory", "m0");
  };
  // after the pair's loads have landed (wave 7's own vmcnt): delta_r = dO_r . O_r, lse_r / scale (1e30 past the last query: P = 0)
  auto publish_stats = [&](int hh, int bb, int p, int slot) {
    const char* sl = ring + slot * V3_SLOT;
    float* ss = st + slot * 64;
    const int rr = lane >> 1, half = lane & 1;          // row of the pair, half of its 64 columns
    const char* dp_ = sl + 4096 + (rr >> 4) * 2048 + (rr & 15) * 128 + half * 64;
    const char* op_ = sl + 8192 + (rr >> 4) * 2048 + (rr & 15) * 128 + half * 64;
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {   // (both rows carry the same chunk swizzle: the dot product pairs the right elements)
      const bf16x8 dv = *reinterpret_cast<const bf16x8*>(dp_ + c * 16), ov = *reinterpret_cast<const bf16x8*>(op_ + c * 16);
#pragma unroll
      for (int e = 0; e < 8; ++e) d = fmaf(bf2f(dv[e]), bf2f(ov[e]), d);
    }
    d += __shfl_xor(d, 1, 64);
    const int row = p * 32 + rr;
    float ls = 0.f;
    if (lane < 32) ls = ss[lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (half == 0) {
      ss[32 + rr] = d;
      if (row < S && a.delta != nullptr) a.delta[((long)bb * a.H + hh) * a.stat_ld + row] = d;
    }
    if (lane < 32) ss[lane] = p * 32 + lane < S ? ls * inv_scale : 1.0e30f;
  };

  f32x4 dsacc[DBIAS ? KT : 1][2];
#pragma unroll
  for (int i = 0; i < (DBIAS ? KT : 1); ++i) dsacc[i][0] = dsacc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  // sum_b dS of this wave's 32 keys leaves through a wave-private 1-KB transpose in exchange buffer `xs` (idle when this runs)
  auto flush = [&](int hh, int xs) {
    if (!DBIAS || a.dbias == nullptr || !owner) return;
    float* scr = reinterpret_cast<float*>(ex + xs * V3_EXCH + w * 1024);
    const int col = lane & 31, key = kt0 * 16 + col;
    float* const dst0 = a.dbias + (long)hh * S * a.bias_ld + key;
#pragma unroll
    for (int i = 0; i < KT; ++i) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {   // query rows 8 hf .. 8 hf + 7 of tile i: lanes with lg >> 1 == hf hold them
        if ((lg >> 1) == hf) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) scr[(4 * (lg & 1) + r) * 32 + t * 16 + lr] = dsacc[i][t][r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
        for (int step = 0; step < 4; ++step) {
          const int row = 2 * step + (lane >> 5);
          const int q = i * 16 + 8 * hf + row;
          const float v = scr[row * 32 + col];
          if (key < S && q < S) atomicAdd(dst0 + (long)q * a.bias_ld, v);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  };

  // ---- prologue: K / V of the first item (all waves), its first pair (wave 7)
  stage_kv_pieces(h, b, 0, w * 7, w * 7 + 7 < 52 ? w * 7 + 7 : 52);
  if (w == 7) stage_pair(h, b, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (w == 7) publish_stats(h, b, 0, 0);
  lds_barrier_v();

  f32x4 dKa[2][4], dVa[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dKa[t][dt] = dVa[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bcur[2];
  auto load_bias_t = [&](f32x4 (&bv)[2], int hh, int qt) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (HAS_BIAS && owner && qt < KT)
        bv[t] = *reinterpret_cast<const f32x4*>(a.bias_t_tiled + (((long)hh * (KT + 1) + kt0 + t) * KT + qt) * 256 + lane * 4);
    }
  };
  load_bias_t(bcur, h, 0);

  // dQ^T[d, q] of one (query tile, d-tile) unit of the pair whose dS sits in exchange buffer `xs`, keys from K buffer `kbuf`
  auto dq_unit = [&](int hh, int bb, int pq, int xs, int kbuf, int un) {
    const int u = un >> 2, dt = un & 3, qt = 2 * pq + u;
    if (qt >= KT) return;
    const int tro = tro_of(dt);
    const char* xb = ex + xs * V3_EXCH + u * V3_EXQ + ex_r;
    const char* kb = lds + kbuf * V3_IMG + tro;
    f32x4 acc2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s2 = 0; s2 < NP; ++s2) {
      union { struct { s16x4 a, b; } s; bf16x8 v; } kf, sf;
      sf.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, xb + (2 * s2) * 512));
      kf.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, kb + (2 * s2) * 2048));
      if (2 * s2 + 1 < KT) {
        sf.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, xb + (2 * s2 + 1) * 512));
        kf.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, kb + (2 * s2 + 1) * 2048));
      } else {   // the 13th key tile has no partner: a zero half
        sf.s.b = s16x4{0, 0, 0, 0};
        kf.s.b = kf.s.a;
      }
      acc2[s2 & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf.v, sf.v, acc2[s2 & 1], 0, 0, 0);
    }
    const int qi = qt * 16 + lr;
    if (qi < S) {
      bf16x4 ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = f2bf((acc2[0][r] + acc2[1][r]) * a.scale);
      *reinterpret_cast<bf16x4*>(a.dq + ((long)bb * S + qi) * a.dq_rs + hh * 64 + dt * 16 + 4 * lg) = ov;
    }
  };

  int hp = h, bp = b, last_par = 0;   // the item whose last pair still waits for its dQ
  bool have_prev = false;
  for (int it = 0; it < vm.c; ++it) {
    const int par = it & 1;
    const bool more = it + 1 < vm.c && vit_item(vm, wg, it + 1, hn, bn);
    const char* sK = lds + par * V3_IMG;
    const char* sV = lds + (2 + par) * V3_IMG;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int slot = (p & 1) ^ par, nslot = slot ^ 1;
      // ---------------- wave 7: the next pair's rows, a share of the next item's K / V, then the pair's statistics
      if (w == 7) {
        const bool next_here = p + 1 < NP;
        if (next_here) stage_pair(h, b, p + 1, nslot);
        else if (more) stage_pair(hn, bn, 0, nslot);
        int nkv = 0;
        if (more && p >= 1) {   // (not in p = 0: the previous item's last dQ still reads the K buffer these pieces overwrite)
          const int j0 = (p - 1) * 9, j1 = p * 9 < 52 ? p * 9 : 52;
          stage_kv_pieces(hn, bn, par ^ 1, j0, j1);
          nkv = j1 - j0;
        }
        if (next_here || more) {
          // the 13 loads of the pair are older than this iteration's K / V pieces: wait for all but those
          if (nkv == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
          else if (nkv == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (next_here) publish_stats(h, b, p + 1, nslot);
          else publish_stats(hn, bn, 0, nslot);
        }
        if (p == NP - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next item's K / V are complete before its first pair
      }
      // ---------------- owners: S, dP, P, dS of query tiles 2p, 2p+1 against this wave's keys
      if (owner) {
        const char* sl = ring + slot * V3_SLOT;
        const float* ss = st + slot * 64;
        char* const exb = ex + slot * V3_EXCH;
        auto body = [&](auto NTc) {
          constexpr int NT = decltype(NTc)::value;
          bf16x4 pp[2][2], ps[2][2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int qt = 2 * p + u;
#pragma unroll
            for (int t = 0; t < 2; ++t) pp[u][t] = ps[u][t] = bf16x4{0, 0, 0, 0};
            if (qt < KT) {
              f32x4 bnx[2];
              if (qt + 1 < KT) load_bias_t(bnx, h, qt + 1);
              else load_bias_t(bnx, more ? hn : h, 0);   // the first tile of the next item
              const f32x4 lsv = *reinterpret_cast<const f32x4*>(ss + u * 16 + 4 * lg);
              const f32x4 dlv = *reinterpret_cast<const f32x4*>(ss + 32 + u * 16 + 4 * lg);
              const bf16x8 qa0 = *reinterpret_cast<const bf16x8*>(sl + u * 2048 + rf0), qa1 = *reinterpret_cast<const bf16x8*>(sl + u * 2048 + rf1);
              const bf16x8 da0 = *reinterpret_cast<const bf16x8*>(sl + 4096 + u * 2048 + rf0), da1 = *reinterpret_cast<const bf16x8*>(sl + 4096 + u * 2048 + rf1);
#pragma unroll
              for (int t = 0; t < NT; ++t) {
                const char* kp = sK + (kt0 + t) * 2048;
                const char* vp = sV + (kt0 + t) * 2048;
                f32x4 sc, dp;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  if (HAS_BIAS) sc[r] = bcur[t][r] - lsv[r];   // tiled: pre-divided by the scale, -1e30 past the last key
                  else sc[r] = kvalid[t] ? -lsv[r] : -1.0e30f;
                  dp[r] = -dlv[r];
                }
                sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa0, *reinterpret_cast<const bf16x8*>(kp + rf0), sc, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa1, *reinterpret_cast<const bf16x8*>(kp + rf1), sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da0, *reinterpret_cast<const bf16x8*>(vp + rf0), dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da1, *reinterpret_cast<const bf16x8*>(vp + rf1), dp, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const float pv = __builtin_amdgcn_exp2f(sc[r] * c2);
                  const float ds = pv * dp[r];
                  if (DBIAS) dsacc[qt < KT ? qt : 0][t][r] += ds;
                  pp[u][t][r] = f2bf(pv);
                  ps[u][t][r] = f2bf(ds);
                }
                *reinterpret_cast<bf16x4*>(exb + u * V3_EXQ + ex_w + t * 512) = ps[u][t];
              }
              bcur[0] = bnx[0];
              bcur[1] = bnx[1];
            }
          }
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const int tro = tro_of(dt);
            union { struct { s16x4 a, b; } s; bf16x8 v; } df, qf;
            df.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, sl + 4096 + tro));
            df.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, sl + 4096 + 2048 + tro));
            qf.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, sl + tro));
            qf.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, sl + 2048 + tro));
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              union { struct { bf16x4 a, b; } s; bf16x8 v; } pf, sf;
              pf.s.a = pp[0][t]; pf.s.b = pp[1][t];
              sf.s.a = ps[0][t]; sf.s.b = ps[1][t];
              dVa[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df.v, pf.v, dVa[t][dt], 0, 0, 0);
              dKa[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf.v, sf.v, dKa[t][dt], 0, 0, 0);
            }
          }
        };
        if (w == 6) body(std::integral_constant<int, 1>{});   // keys 192 .. 207 only: its second tile is past the sequence
        else body(std::integral_constant<int, 2>{});
      }
      // ---------------- dQ of the previous pair (exchange buffer slot ^ 1): one unit per wave
      if (p > 0) dq_unit(h, b, p - 1, slot ^ 1, par, w);
      else if (have_prev) dq_unit(hp, bp, NP - 1, slot ^ 1, par ^ 1, w);
      lds_barrier_v();
    }
    // ---- dK, dV of this wave's keys; the bias gradient leaves when the head changes
    if (owner) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int key = (kt0 + t) * 16 + lr;
        if (key < S) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            bf16x4 ok_, ov_;
#pragma unroll
            for (int r = 0; r < 4; ++r) { ok_[r] = f2bf(dKa[t][dt][r] * a.scale); ov_[r] = f2bf(dVa[t][dt][r]); }
            *reinterpret_cast<bf16x4*>(a.dk + ((long)b * S + key) * a.dk_rs + h * 64 + dt * 16 + 4 * lg) = ok_;
            *reinterpret_cast<bf16x4*>(a.dv + ((long)b * S + key) * a.dv_rs + h * 64 + dt * 16 + 4 * lg) = ov_;
          }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dKa[t][dt] = dVa[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    hp = h; bp = b; have_prev = true; last_par = par;
    if (!more) break;
    if (hn != h) {
      // exchange buffer (par ^ 1) ^ 1... : the last pair (p = 6) wrote buffer ((6 & 1) ^ par) = par; its dQ is still pending, so the
      // transposition scratch takes the other one
      flush(h, par ^ 1);
#pragma unroll
      for (int i = 0; i < (DBIAS ? KT : 1); ++i) dsacc[i][0] = dsacc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      lds_barrier_v();
    }
    h = hn;
    b = bn;
  }
  // ---- drain: dQ of the very last pair (its dS sits in exchange buffer `last_par`, its keys in K buffer `last_par`), then the bias
  // gradient through the other exchange buffer
  dq_unit(hp, bp, NP - 1, last_par, last_par, w);
  flush(hp, last_par ^ 1);
}

static bool attn_vit3_shape(const AttnArgs& a) {
  const char* e = getenv("XFM_ATTN_VIT_BWD");   // (read per call: the tests switch it inside one process)
  const int mode = e != nullptr ? atoi(e) : 0;
  return mode != 0 && attn_vit_shape(a) && cdiv(a.Sq, 16) == V3_KT && a.bwd_phase == 0 && a.o != nullptr && a.o_lo == nullptr &&
         (a.bias == nullptr || a.bias_t_tiled != nullptr) && (a.dbias == nullptr || a.bias != nullptr) &&
         ((uintptr_t)a.dout % 16) == 0 && ((uintptr_t)a.o % 16) == 0 && a.o_rs % 8 == 0;
}

static int launch_attn_bwd_vit3(const AttnArgs& a, hipStream_t st) {
  const VitMap vm = vit_map(a.B, a.H);
  if (a.bias != nullptr && a.dbias != nullptr) vit_launch<7>(attn_bwd_vit3_kernel<true, true>, V3_LDS, dim3(vm.grid), dim3(512), st, a, vm);
  else if (a.bias != nullptr) vit_launch<9>(attn_bwd_vit3_kernel<true, false>, V3_LDS, dim3(vm.grid), dim3(512), st, a, vm);
  else vit_launch<8>(attn_bwd_vit3_kernel<false, false>, V3_LDS, dim3(vm.grid), dim3(512), st, a, vm);
  return xfm_check_launch("attn_bwd_vit3");
}
