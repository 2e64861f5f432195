// Attention backward for LONG dense, unmasked problems with a relative-position bias: the BEiT tower at 384 px (577 tokens, BASELINE
// configs[2]) and 480 px (901 tokens, configs[3]), beit2.py:126-166.  Included by attention.hip.
//
// Rounds 1-3 ran these shapes on the general kernels: a wave owned 16 queries (or 16 keys), waited `vmcnt(0)` + a barrier for every
// 64-row chunk, and the bias gradient left the dQ kernel as one [B, H, S, S] fp32 workspace (0.95 GB at B = 24, 901 tokens) that a
// second kernel summed over the batch: 1.29 ms per layer at (24, 901), 4.6 % of the MFMA peak, a quarter of the VQA step.
//
// Here the backward is three kernels, none of which touches an S x S buffer per batch entry:
//   attn_bwd_dq_long_kernel    workgroup = (256-query block, head, batch entry): a wave owns 32 queries (two tiles: every K / V fragment
//                              it reads from LDS feeds two MFMAs), the keys stream through a 3-slot ring of 64-key chunks filled by
//                              direct-to-LDS loads that stay in flight across the one barrier per chunk (counted vmcnt); the bias row
//                              segments arrive by asynchronous register loads issued before the score MFMAs.  delta = dO . (O + O_lo)
//                              when the forward kept the low half of O, else a first pass over the keys (exact).
//   attn_bwd_dkv_long_kernel   the mirror image: a wave owns 32 keys (dK^T, dV^T accumulators), Q / dO chunks and their row statistics
//                              stream through the ring, the transposed bias copy gives 16-byte loads.
//   attn_dbias_long_kernel     the bias gradient sum_b dS: workgroup = (128 x 128 block of one head's S x S bias, batch slice) WALKS
//                              the batch entries of its slice with the block's gradient in registers (32 VGPRs per lane) -- S and dP
//                              are recomputed (2 of the backward's matrix products again), but the sum over the batch never leaves the
//                              register file: one plain store per slice into a [slices, H, S, ld] plane buffer (slices = 1-8, a few MB)
//                              that dbias_reduce_kernel folds into dbias in slice order.  No atomics, bit-reproducible.
#define LB_RING 3
#define LB_STAT 512                                   // per ring slot: 64 log-sum-exps + 64 deltas (dK/dV kernel)
#define LB_SLOT_DQ ATTN_SLOT                          // K | V tile of 64 keys
#define LB_SLOT_DKV (ATTN_SLOT + LB_STAT)             // Q | dO tile of 64 queries | lse | delta
#define LB_LOG2E 1.44269504088896341f

__device__ __forceinline__ void lb_wait_vm(int n) {  // wave-uniform n
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
// A 16-byte register load the COMPILER DOES NOT TRACK (it would place `s_waitcnt vmcnt(0)` in front of the first use and drain the
// direct-to-LDS prefetch that was issued after it).  The caller waits with lb_wait_vm(...) and then passes the value through
// lb_use() so that no use can be scheduled ahead of the wait.
__device__ __forceinline__ f32x4 lb_load_f32x4(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void lb_use(f32x4& v) { asm volatile("" : "+v"(v)::"memory"); }
// one 1-KiB piece (8 rows x 128 B, XOR-swizzled source chunks: the image of stage_slot) of a 64-row tile, direct to LDS
__device__ __forceinline__ void lb_piece(char* tile, const bf16* g, long rs, int row0, int nvalid, int piece, int lane) {
  const int r = piece * 8 + (lane >> 3);
  const int c = (lane & 7) ^ swz_a(r);
  int gr = row0 + r;
  gr = gr < nvalid ? gr : nvalid - 1;
  const bf16* src = g + (long)gr * rs + c * 8;
  const unsigned dst = (unsigned)(uintptr_t)LDS_PTR(void, tile) + (unsigned)__builtin_amdgcn_readfirstlane(piece * 1024);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory", "m0");
}
__device__ __forceinline__ void lb_barrier() {  // raw barrier: LDS-DMA stays in flight across it
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dQ (+ delta).  512 threads: wave w owns queries (qblk * 8 + w) * 32 .. + 31.
// ---------------------------------------------------------------------------------------------------------------------------------
template <bool BIAS, bool TILED>
__global__ __launch_bounds__(512) void attn_bwd_dq_long_kernel(AttnArgs a, int qblocks) {
  constexpr int NW = 8, NP = 16 / NW;  // direct-to-LDS wave-instructions per wave and chunk (8 K pieces + 8 V pieces over NW waves)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);  // the query blocks of one (b, h) are neighbours: one XCD's L2 serves their K / V
  const int qblk = wg % qblocks, h = (wg / qblocks) % a.H, b = wg / (qblocks * a.H);
  const int sq = a.Sq, sk = a.Sk;
  const long qbase = (long)b * sq, kbase = (long)b * sk;
  const int q0 = (qblk * NW + w) * 32;
  const bool wave_active = q0 < sq;
  const bf16* kb = a.k + kbase * a.k_rs + h * 64;
  const bf16* vb = a.v + kbase * a.v_rs + h * 64;
  const int nchunks = (sk + 63) >> 6;
  const float c2 = a.scale * LB_LOG2E;
  const int T = (sk + 15) >> 4;  // TILED (Sq == Sk): tiles per side of the accumulator-layout bias copy

  bf16x8 qf[2][2], df[2][2];
  float nl[2], dlt[2];
  long stat_idx[2];
  bool qvalid[2];
  const float* brow[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qi = q0 + qt * 16 + lr;
    qvalid[qt] = qi < sq;
    const int qc = qvalid[qt] ? qi : sq - 1;
    const bf16* qp = a.q + (qbase + qc) * a.q_rs + h * 64;
    const bf16* dop = a.dout + (qbase + qc) * a.do_rs + h * 64;
    qf[qt][0] = *reinterpret_cast<const bf16x8*>(qp + 8 * lg);
    qf[qt][1] = *reinterpret_cast<const bf16x8*>(qp + 32 + 8 * lg);
    df[qt][0] = *reinterpret_cast<const bf16x8*>(dop + 8 * lg);
    df[qt][1] = *reinterpret_cast<const bf16x8*>(dop + 32 + 8 * lg);
    stat_idx[qt] = ((long)b * a.H + h) * a.stat_ld + qc;
    nl[qt] = qvalid[qt] ? -a.lse[stat_idx[qt]] * LB_LOG2E : -3.0e38f;  // rows past Sq: P = exp2(-huge) = 0
    dlt[qt] = 0.f;
    if (TILED) {  // bias_tiled[h][query tile][key tile][lane][4] (xfm_bias_tile): one contiguous 1-KB wave load per tile, pre-divided by
      // the scale, -1e30 past the last key; a query tile past the end re-reads the last one (its rows have P = 0)
      int qtile = (q0 >> 4) + qt;
      qtile = qtile < T ? qtile : T - 1;
      brow[qt] = a.bias_tiled + ((long)h * T + qtile) * T * 256 + lane * 4;
    } else {
      brow[qt] = BIAS ? a.bias + ((long)h * sq + qc) * a.bias_ld : nullptr;
    }
  }
  const bool fast = a.o_lo != nullptr;  // (uniform over the launch)
  if (fast) {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int qi = q0 + qt * 16 + lr;
      dlt[qt] = delta_from_out(a, qbase + (qi < sq ? qi : sq - 1), h, lg, df[qt][0], df[qt][1]);
    }
  }
  // every load above has returned before the first counted wait below is relied on.  (The compiler cannot see this wait: the empty
  // asm makes it place ITS wait for the loaded registers here, not at their first use inside the loop, where it would also drain the
  // direct-to-LDS loads in flight.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
    asm volatile("" : "+v"(qf[qt][0]), "+v"(qf[qt][1]), "+v"(df[qt][0]), "+v"(df[qt][1]), "+v"(nl[qt]), "+v"(dlt[qt]));

  f32x4 dqacc[2][4];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dqacc[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int kc) {  // chunk kc -> ring slot kc mod 3: this wave's K pieces and V pieces (NP wave-instructions)
    char* slot = lds + (kc % LB_RING) * LB_SLOT_DQ;
#pragma unroll
    for (int i = 0; i < 8 / NW; ++i) {
      lb_piece(slot, kb, a.k_rs, kc * 64, sk, w + i * NW, lane);
      lb_piece(slot + ATTN_TILE, vb, a.v_rs, kc * 64, sk, w + i * NW, lane);
    }
  };

  for (int pass = fast ? 1 : 0; pass < 2; ++pass) {
    // pass 0 (no O_lo): delta_i = sum_j P_ij dP_ij, from the same P and dP that form dS in pass 1; pass 1: dS, dQ
    if (pass == 1 && !fast) lb_barrier();  // every wave is done with the ring slots of pass 0
    stage(0);
    if (nchunks > 1) stage(1);
    for (int kc = 0; kc < nchunks; ++kc) {
      lb_wait_vm(kc + 1 < nchunks ? NP : 0);  // this wave's pieces of chunk kc have landed (chunk kc + 1 may still fly)
      lb_barrier();                           // ... everyone's; and everyone is done reading the slot chunk kc + 2 goes to
      const bool more = kc + 2 < nchunks;
      // the bias segments go out before the score MFMAs and are waited for after them (prefetching them a whole chunk ahead, in a
      // second register set, measured 13 % SLOWER: 64 more live registers, K / V fragments read twice)
      f32x4 bv[2][4];
      if (BIAS && wave_active) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if (TILED) {
              int kt = kc * 4 + t;
              kt = kt < T ? kt : T - 1;  // (a key tile past the end: excluded below)
              bv[qt][t] = lb_load_f32x4(brow[qt] + (long)kt * 256);
            } else {
              int kj0 = kc * 64 + t * 16 + 4 * lg;
              kj0 = kj0 + 4 <= a.bias_ld ? kj0 : 0;  // (a tile that starts past the padded row: its scores are excluded below)
              bv[qt][t] = lb_load_f32x4(brow[qt] + kj0);
            }
          }
      }
      if (more) stage(kc + 2);
      if (!wave_active) continue;
      const char* sK = lds + (kc % LB_RING) * LB_SLOT_DQ;
      const char* sV = sK + ATTN_TILE;
      f32x4 st[2][4], dp[2][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bf16x8 k0 = row_frag(sK, t * 16, 0, lr, lg), k1 = row_frag(sK, t * 16, 1, lr, lg);
        const bf16x8 v0 = row_frag(sV, t * 16, 0, lr, lg), v1 = row_frag(sV, t * 16, 1, lr, lg);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          st[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[qt][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          st[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[qt][1], st[qt][t], 0, 0, 0);
          dp[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, df[qt][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          dp[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, df[qt][1], dp[qt][t], 0, 0, 0);
        }
      }
      if (BIAS) {
        lb_wait_vm(more ? NP : 0);  // the bias segments are in (they are older than chunk kc + 2's pieces)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int t = 0; t < 4; ++t) lb_use(bv[qt][t]);
      }
      const bool tail = kc * 64 + 64 > sk;  // wave-uniform: this chunk holds rows past the last key
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // TILED: the copy holds bias / scale (-1e30 past the last key of a partial tile)
            const float nb = BIAS ? fmaf(bv[qt][t][r], TILED ? c2 : LB_LOG2E, nl[qt]) : nl[qt];
            float p = __builtin_amdgcn_exp2f(fmaf(st[qt][t][r], c2, nb));
            if (tail) p = kc * 64 + t * 16 + 4 * lg + r < sk ? p : 0.f;
            st[qt][t][r] = p;
          }
      if (pass == 0) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) dlt[qt] = fmaf(st[qt][t][r], dp[qt][t][r], dlt[qt]);
        continue;
      }
      bf16x8 pf[2][2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) st[qt][t][r] *= dp[qt][t][r] - dlt[qt];
        pf[qt][0] = pack_pair(st[qt][0], st[qt][1]);
        pf[qt][1] = pack_pair(st[qt][2], st[qt][3]);
      }
      // dQ^T[d, q] += K^T[d, key] . dS^T[key, q]: one transposed K fragment per (d-tile, 32 keys), two query tiles each
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const bf16x8 kt = tr_frag(sK, 32 * s2, 32 * s2 + 16, dt * 16, lr, lg);
          dqacc[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, pf[0][s2], dqacc[0][dt], 0, 0, 0);
          dqacc[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, pf[1][s2], dqacc[1][dt], 0, 0, 0);
        }
    }
    if (pass == 0) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) dlt[qt] = group4_sum(dlt[qt]);
    }
  }
  if (!wave_active) return;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qi = q0 + qt * 16 + lr;
    if (!qvalid[qt]) continue;
    if (lg == 0) a.delta[stat_idx[qt]] = dlt[qt];
    bf16* dqp = a.dq + (qbase + qi) * a.dq_rs + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16x4 ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = f2bf(dqacc[qt][dt][r] * a.scale);
      *reinterpret_cast<bf16x4*>(dqp + dt * 16 + 4 * lg) = ov;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dK, dV.  512 threads: wave w owns keys (kblk * 8 + w) * 32 .. + 31; Q | dO | lse | delta stream in 64-query chunks.
// D[i = query row][j = key col]: lane (lg, lr) holds queries 16 t + 4 lg + r of key lr, so the transposed bias copy is one 16-byte
// load per tile and the accumulators of S and dP are the B operands of dV^T += dO^T P and dK^T += Q^T dS as they stand.
// ---------------------------------------------------------------------------------------------------------------------------------
template <bool BIAS, bool TILED>
__global__ __launch_bounds__(512) void attn_bwd_dkv_long_kernel(AttnArgs a, int kblocks) {
  constexpr int NW = 8;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int kblk = wg % kblocks, h = (wg / kblocks) % a.H, b = wg / (kblocks * a.H);
  const int sq = a.Sq, sk = a.Sk;
  const long qbase = (long)b * sq, kbase = (long)b * sk;
  const int k0 = (kblk * NW + w) * 32;
  const bool wave_active = k0 < sk;
  const bf16* qb = a.q + qbase * a.q_rs + h * 64;
  const bf16* db = a.dout + qbase * a.do_rs + h * 64;
  const float* lse_b = a.lse + ((long)b * a.H + h) * a.stat_ld;
  const float* del_b = a.delta + ((long)b * a.H + h) * a.stat_ld;
  const int nchunks = (sq + 63) >> 6;
  const float c2 = a.scale * LB_LOG2E;
  const int T = (sq + 15) >> 4;  // TILED (Sq == Sk)

  bf16x8 kf[2][2], vf[2][2];
  bool kvalid[2];
  const float* btrow[2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int kj = k0 + kt * 16 + lr;
    kvalid[kt] = kj < sk;
    const int kc = kvalid[kt] ? kj : sk - 1;
    const bf16* kp = a.k + (kbase + kc) * a.k_rs + h * 64;
    const bf16* vp = a.v + (kbase + kc) * a.v_rs + h * 64;
    kf[kt][0] = *reinterpret_cast<const bf16x8*>(kp + 8 * lg);
    kf[kt][1] = *reinterpret_cast<const bf16x8*>(kp + 32 + 8 * lg);
    vf[kt][0] = *reinterpret_cast<const bf16x8*>(vp + 8 * lg);
    vf[kt][1] = *reinterpret_cast<const bf16x8*>(vp + 32 + 8 * lg);
    if (TILED) {  // bias_t_tiled[h][key tile (T + 1: the last is all -1e30)][query tile][lane][4]: key on the lane, pre-divided by the scale
      int ktile = (k0 >> 4) + kt;
      ktile = ktile < T ? ktile : T;
      btrow[kt] = a.bias_t_tiled + ((long)h * (T + 1) + ktile) * T * 256 + lane * 4;
    } else {
      btrow[kt] = BIAS ? a.bias_t + ((long)h * sk + kc) * a.bias_t_ld : nullptr;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (see the dQ kernel)
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) asm volatile("" : "+v"(kf[kt][0]), "+v"(kf[kt][1]), "+v"(vf[kt][0]), "+v"(vf[kt][1]));

  f32x4 dkacc[2][4], dvacc[2][4];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dkacc[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // waves 0 / 1 also bring the chunk's 64 log-sum-exps / deltas (4 B per lane; rows past Sq repeat the last one: their P is zeroed)
  const int n_dma = 16 / NW + (w < 2 ? 1 : 0);
  auto stage = [&](int qc) {
    char* slot = lds + (qc % LB_RING) * LB_SLOT_DKV;
#pragma unroll
    for (int i = 0; i < 8 / NW; ++i) {
      lb_piece(slot, qb, a.q_rs, qc * 64, sq, w + i * NW, lane);
      lb_piece(slot + ATTN_TILE, db, a.do_rs, qc * 64, sq, w + i * NW, lane);
    }
    if (w < 2) {
      int qi = qc * 64 + lane;
      qi = qi < sq ? qi : sq - 1;
      const float* sp = (w == 0 ? lse_b : del_b) + qi;
      const unsigned dst = (unsigned)(uintptr_t)LDS_PTR(void, slot + ATTN_SLOT) + (unsigned)__builtin_amdgcn_readfirstlane(w * 256);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(sp), "s"(dst) : "memory", "m0");
    }
  };
  stage(0);
  if (nchunks > 1) stage(1);
  for (int qc = 0; qc < nchunks; ++qc) {
    lb_wait_vm(qc + 1 < nchunks ? n_dma : 0);
    lb_barrier();
    const bool more = qc + 2 < nchunks;
    // transposed-bias segments of both 32-query halves of the chunk go out first (8 loads), then the next chunk's pieces
    f32x4 bt[2][2][2];  // [half][key tile][query tile of the half]
    if (BIAS && wave_active) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (TILED) {
              int qtile = qc * 4 + 2 * s2 + u;
              qtile = qtile < T ? qtile : T - 1;  // (a query tile past the end: zeroed below)
              bt[s2][kt][u] = lb_load_f32x4(btrow[kt] + (long)qtile * 256);
            } else {
              int qi0 = qc * 64 + (2 * s2 + u) * 16 + 4 * lg;
              qi0 = qi0 + 4 <= a.bias_t_ld ? qi0 : 0;
              bt[s2][kt][u] = lb_load_f32x4(btrow[kt] + qi0);
            }
          }
    }
    if (more) stage(qc + 2);
    if (!wave_active) continue;
    const char* sQ = lds + (qc % LB_RING) * LB_SLOT_DKV;
    const char* sD = sQ + ATTN_TILE;
    const float* sL = reinterpret_cast<const float*>(sQ + ATTN_SLOT);
    const bool tail = qc * 64 + 64 > sq;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      f32x4 st[2][2], dp[2][2];  // [key tile][query tile]
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * s2 + u;
        const bf16x8 q0f = row_frag(sQ, t * 16, 0, lr, lg), q1f = row_frag(sQ, t * 16, 1, lr, lg);
        const bf16x8 d0f = row_frag(sD, t * 16, 0, lr, lg), d1f = row_frag(sD, t * 16, 1, lr, lg);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          st[kt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0f, kf[kt][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          st[kt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1f, kf[kt][1], st[kt][u], 0, 0, 0);
          dp[kt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d0f, vf[kt][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          dp[kt][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d1f, vf[kt][1], dp[kt][u], 0, 0, 0);
        }
      }
      if (BIAS) {
        // half 0: the 4 loads of half 1 and chunk qc + 2's pieces are younger; half 1: only the pieces
        lb_wait_vm((s2 == 0 ? 4 : 0) + (more ? n_dma : 0));
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int u = 0; u < 2; ++u) lb_use(bt[s2][kt][u]);
      }
      f32x4 pd[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int qi0 = qc * 64 + (2 * s2 + u) * 16 + 4 * lg;
        const f32x4 lsv = *reinterpret_cast<const f32x4*>(sL + (2 * s2 + u) * 16 + 4 * lg);
        const f32x4 dlv = *reinterpret_cast<const f32x4*>(sL + 64 + (2 * s2 + u) * 16 + 4 * lg);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float nlr = -lsv[r] * LB_LOG2E;
            const float nb = BIAS ? fmaf(bt[s2][kt][u][r], TILED ? c2 : LB_LOG2E, nlr) : nlr;
            float p = __builtin_amdgcn_exp2f(fmaf(st[kt][u][r], c2, nb));
            if (tail) p = qi0 + r < sq ? p : 0.f;
            p = kvalid[kt] ? p : 0.f;
            pd[kt][u][r] = p;
            st[kt][u][r] = p * (dp[kt][u][r] - dlv[r]);
          }
      }
      // dV^T[d, key] += dO^T[d, q] . P[q, key] ;  dK^T[d, key] += Q^T[d, q] . dS[q, key]   (one transposed fragment per d-tile, two key tiles)
      const bf16x8 pf0 = pack_pair(pd[0][0], pd[0][1]), pf1 = pack_pair(pd[1][0], pd[1][1]);
      const bf16x8 sf0 = pack_pair(st[0][0], st[0][1]), sf1 = pack_pair(st[1][0], st[1][1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 dT = tr_frag(sD, 32 * s2, 32 * s2 + 16, dt * 16, lr, lg);
        const bf16x8 qT = tr_frag(sQ, 32 * s2, 32 * s2 + 16, dt * 16, lr, lg);
        dvacc[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dT, pf0, dvacc[0][dt], 0, 0, 0);
        dvacc[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dT, pf1, dvacc[1][dt], 0, 0, 0);
        dkacc[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT, sf0, dkacc[0][dt], 0, 0, 0);
        dkacc[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT, sf1, dkacc[1][dt], 0, 0, 0);
      }
    }
  }
  if (!wave_active) return;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    if (!kvalid[kt]) continue;
    const int kj = k0 + kt * 16 + lr;
    bf16* dkp = a.dk + (kbase + kj) * a.dk_rs + h * 64;
    bf16* dvp = a.dv + (kbase + kj) * a.dv_rs + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16x4 ok_, ov_;
#pragma unroll
      for (int r = 0; r < 4; ++r) { ok_[r] = f2bf(dkacc[kt][dt][r] * a.scale); ov_[r] = f2bf(dvacc[kt][dt][r]); }
      *reinterpret_cast<bf16x4*>(dkp + dt * 16 + 4 * lg) = ok_;
      *reinterpret_cast<bf16x4*>(dvp + dt * 16 + 4 * lg) = ov_;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dbias[h] block (128 queries x 128 keys) = sum over a slice of the batch of dS.  512 threads as 4 (query groups of 32) x 2 (key
// groups of 64): a wave holds 2 x 4 tiles of S^T (key on the MFMA row, query on the lane: the accumulator's four registers are four
// consecutive keys of one bias row -- 16-byte loads and stores of the bias / its gradient).  Per batch entry the block's Q, dO, K, V
// rows (4 x 16 KB) and the 128 + 128 row statistics land in the other half of a double buffer while this entry computes.
// ---------------------------------------------------------------------------------------------------------------------------------
#define LD_STAGE (4 * 2 * ATTN_TILE + 1024)   // Q (2 tiles) | dO | K | V | lse (128) | delta (128)
template <bool BIAS>
__global__ __launch_bounds__(512) void attn_dbias_long_kernel(AttnArgs a, int qblocks, int kblocks, int slices, float* __restrict__ planes) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int wq = w >> 1, wk = w & 1;
  // logical id = key block + kblocks * (query block + qblocks * (slice + slices * head)): the blocks of one (head, slice) are neighbours
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int kb_ = wg % kblocks, qb_ = (wg / kblocks) % qblocks, sl = (wg / (kblocks * qblocks)) % slices, h = wg / (kblocks * qblocks * slices);
  const int sq = a.Sq, sk = a.Sk;
  const int per = (a.B + slices - 1) / slices;
  const int b_begin = sl * per, b_end = b_begin + per < a.B ? b_begin + per : a.B;
  const int q0 = qb_ * 128, k0 = kb_ * 128;
  const float c2 = a.scale * LB_LOG2E;

  // batch-invariant: bias * log2(e) of this wave's tiles (-1e30 past the last key: P = 0), validity of its query rows
  f32x4 b2[2][4], acc[2][4];
  bool qvalid[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qi = q0 + wq * 32 + qt * 16 + lr;
    qvalid[qt] = qi < sq;
    const int qc = qvalid[qt] ? qi : sq - 1;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[qt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int kj0 = k0 + wk * 64 + t * 16 + 4 * lg;
      f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
      if (BIAS && kj0 + 4 <= a.bias_ld) bv = *reinterpret_cast<const f32x4*>(a.bias + ((long)h * sq + qc) * a.bias_ld + kj0);
#pragma unroll
      for (int r = 0; r < 4; ++r) b2[qt][t][r] = kj0 + r < sk ? bv[r] * LB_LOG2E : -1.0e30f;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (see the dQ kernel)
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(b2[qt][t]));

  // pieces of a stage: 8 per 64-row tile, 8 tiles (Q0 Q1 dO0 dO1 K0 K1 V0 V1) -> 64 pieces, 8 per wave: wave w brings piece w of every tile
  auto stage = [&](int b, int buf) {
    char* st_ = lds + buf * LD_STAGE;
    const bf16* qg = a.q + (long)b * sq * a.q_rs + h * 64;
    const bf16* dg = a.dout + (long)b * sq * a.do_rs + h * 64;
    const bf16* kg = a.k + (long)b * sk * a.k_rs + h * 64;
    const bf16* vg = a.v + (long)b * sk * a.v_rs + h * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      lb_piece(st_ + (0 + i) * ATTN_TILE, qg, a.q_rs, q0 + i * 64, sq, w, lane);
      lb_piece(st_ + (2 + i) * ATTN_TILE, dg, a.do_rs, q0 + i * 64, sq, w, lane);
      lb_piece(st_ + (4 + i) * ATTN_TILE, kg, a.k_rs, k0 + i * 64, sk, w, lane);
      lb_piece(st_ + (6 + i) * ATTN_TILE, vg, a.v_rs, k0 + i * 64, sk, w, lane);
    }
    if (w < 4) {  // lse rows 0-63 / 64-127, delta rows 0-63 / 64-127
      int qi = q0 + (w & 1) * 64 + lane;
      qi = qi < sq ? qi : sq - 1;
      const float* sp = (w < 2 ? a.lse : a.delta) + ((long)b * a.H + h) * a.stat_ld + qi;
      const unsigned dst = (unsigned)(uintptr_t)LDS_PTR(void, st_ + 8 * ATTN_TILE) + (unsigned)__builtin_amdgcn_readfirstlane(w * 256);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(sp), "s"(dst) : "memory", "m0");
    }
  };
  const int n_dma = 8 + (w < 4 ? 1 : 0);
  if (b_begin < b_end) stage(b_begin, 0);
  for (int b = b_begin; b < b_end; ++b) {
    const int buf = (b - b_begin) & 1;
    lb_wait_vm(0);   // entry b has landed (nothing younger is in flight here)
    lb_barrier();    // ... everyone's pieces; everyone is done with entry b - 1 (the other buffer)
    if (b + 1 < b_end) stage(b + 1, buf ^ 1);
    (void)n_dma;
    const char* base = lds + buf * LD_STAGE;
    const char* sQ = base + (wq >> 1) * ATTN_TILE;           // this wave's 32 queries: rows (wq & 1) * 32 .. of Q tile wq >> 1
    const char* sD = base + (2 + (wq >> 1)) * ATTN_TILE;
    const char* sK = base + (4 + wk) * ATTN_TILE;            // its 64 keys: tile wk
    const char* sV = base + (6 + wk) * ATTN_TILE;
    const float* sL = reinterpret_cast<const float*>(base + 8 * ATTN_TILE);
    bf16x8 qf[2][2], df[2][2];
    float nl[2], dl[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int row = (wq & 1) * 32 + qt * 16;
      qf[qt][0] = row_frag(sQ, row, 0, lr, lg); qf[qt][1] = row_frag(sQ, row, 1, lr, lg);
      df[qt][0] = row_frag(sD, row, 0, lr, lg); df[qt][1] = row_frag(sD, row, 1, lr, lg);
      const int srow = wq * 32 + qt * 16 + lr;
      nl[qt] = qvalid[qt] ? -sL[srow] * LB_LOG2E : -3.0e38f;
      dl[qt] = sL[128 + srow];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bf16x8 k0f = row_frag(sK, t * 16, 0, lr, lg), k1f = row_frag(sK, t * 16, 1, lr, lg);
      const bf16x8 v0f = row_frag(sV, t * 16, 0, lr, lg), v1f = row_frag(sV, t * 16, 1, lr, lg);
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0f, qf[qt][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1f, qf[qt][1], s, 0, 0, 0);
        f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0f, df[qt][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1f, df[qt][1], d, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[r], c2, b2[qt][t][r] + nl[qt]));
          acc[qt][t][r] = fmaf(p, d[r] - dl[qt], acc[qt][t][r]);
        }
      }
    }
  }
  // the block's sum over the slice: plain 16-byte stores into this slice's plane (rows of bias_ld floats); a single slice owns its
  // block of dbias outright and adds to it in place (planes == NULL)
  float* plane = planes != nullptr ? planes + ((long)sl * a.H + h) * sq * a.bias_ld : a.dbias + (long)h * sq * a.bias_ld;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qi = q0 + wq * 32 + qt * 16 + lr;
    if (!qvalid[qt]) continue;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int kj0 = k0 + wk * 64 + t * 16 + 4 * lg;
      if (kj0 + 4 > a.bias_ld) continue;
      f32x4* dst = reinterpret_cast<f32x4*>(plane + (long)qi * a.bias_ld + kj0);
      *dst = planes != nullptr ? acc[qt][t] : *dst + acc[qt][t];
    }
  }
}

// Slices of the batch for the bias-gradient kernel.  Model (us, measured at 901 / 577 tokens): a workgroup spends ~2 per batch entry
// (bound by staging 64 KB per entry into LDS) plus ~3 of prologue / epilogue; every slice beyond the first costs one plane written
// and read back (~4 TB/s).
static int dbias_long_slices(int blocks, int B, long plane_bytes) {
  int best = 1;
  double best_cost = -1.0;
  for (int s = 1; s <= 16 && s <= B; ++s) {
    const double cost = (double)cdiv((long)blocks * s, 256) * (2.0 * cdiv(B, s) + 3.0) + (s > 1 ? s * (double)plane_bytes * 2.0 / 4.0e6 : 0.0);
    if (best_cost < 0 || cost < best_cost - 1e-9) { best_cost = cost; best = s; }
  }
  return best;
}

static bool attn_long_shape(const AttnArgs& a) {
  static const bool on = getenv("XFM_ATTN_LONG") ? atoi(getenv("XFM_ATTN_LONG")) != 0 : true;  // A/B knob: 0 = the general kernels
  return on && a.Sk > 64 * ATTN_RES_MAX && a.key_keep == nullptr && a.causal == 0 && a.drop_thresh == 0u && a.q_start == nullptr &&
         a.k_start == nullptr && a.kv_index == nullptr && a.grp_start == nullptr &&
         (a.bias == nullptr || (a.bias_ld % 4 == 0 && a.bias_ld >= (long)cdiv(a.Sk, 4) * 4 && ((uintptr_t)a.bias % 16) == 0)) &&
         (a.bias_t == nullptr || (a.bias_t_ld % 4 == 0 && a.bias_t_ld >= (long)cdiv(a.Sq, 4) * 4 && ((uintptr_t)a.bias_t % 16) == 0)) &&
         // the bias-gradient kernel writes every column of a plane row: the row must end inside the last 128-key block
         (a.dbias == nullptr || (a.bias != nullptr && a.bias_ld <= (long)cdiv(a.Sk, 128) * 128 && ((uintptr_t)a.dbias % 16) == 0));
}
// XFM_ATTN_VIT_BWD=4 (experiment): the single-pass backward WITHOUT its bias-gradient sums (they are what spills it) + this file's
// block-walking bias-gradient kernel on the delta it wrote
static bool attn_vit_split_dbias(const AttnArgs& a) {
  const char* e = getenv("XFM_ATTN_VIT_BWD");
  return e != nullptr && atoi(e) == 4 && a.dbias != nullptr && attn_vit3_shape(a) && a.bias_ld % 4 == 0 && a.bias_ld <= (long)cdiv(a.Sk, 128) * 128 &&
         ((uintptr_t)a.dbias % 16) == 0 && ((uintptr_t)a.bias % 16) == 0;
}
long xfm_attn_bwd_workspace_impl(const AttnArgs& a) {
  if (attn_vit_split_dbias(a)) {
    const int blocks = cdiv(a.Sq, 128) * cdiv(a.Sk, 128) * a.H;
    const int slices = dbias_long_slices(blocks, a.B, (long)a.H * a.Sq * a.bias_ld * 4);
    return slices > 1 ? (long)slices * a.H * a.Sq * a.bias_ld * 4 : 0;
  }
  if (attn_short_dbias_planes(a)) {   // XFM_DETERMINISTIC: one plane per batch slice of the short dQ kernel
    int groups, nb;
    return (long)attn_short_dq_slices(a, groups, nb) * a.H * a.Sq * a.bias_ld * 4;
  }
  if (a.dbias == nullptr || a.Sk <= 64 * ATTN_RES_MAX) return 0;
  if (attn_long_shape(a)) {
    const int blocks = cdiv(a.Sq, 128) * cdiv(a.Sk, 128) * a.H;
    const int slices = dbias_long_slices(blocks, a.B, (long)a.H * a.Sq * a.bias_ld * 4);
    return slices > 1 ? (long)slices * a.H * a.Sq * a.bias_ld * 4 : 0;
  }
  return (long)a.B * a.H * a.Sq * a.bias_ld * 4;  // the general dQ kernel's per-entry dS (xfm_attn_args.dbias_ws)
}

template <typename K>
static void long_attr(K kernel, int bytes) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
// the accumulator-layout bias copies (xfm_bias_tile) serve square problems
static bool long_tiled(const AttnArgs& a, const float* tiles) { return a.bias != nullptr && tiles != nullptr && a.Sq == a.Sk && ((uintptr_t)tiles % 16) == 0; }

static int launch_attn_dbias_blocks(const AttnArgs& a, hipStream_t st);
static int launch_attn_bwd_long_dq(const AttnArgs& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    long_attr(attn_bwd_dq_long_kernel<true, true>, LB_RING * LB_SLOT_DQ);
    long_attr(attn_bwd_dq_long_kernel<true, false>, LB_RING * LB_SLOT_DQ);
    long_attr(attn_bwd_dq_long_kernel<false, false>, LB_RING * LB_SLOT_DQ);
    long_attr(attn_dbias_long_kernel<true>, 2 * LD_STAGE);
    attr_set = true;
  }
  const int qblocks = cdiv(a.Sq, 256);
  const dim3 grid(qblocks * a.H * a.B), blk(512);
  const size_t lds_b = LB_RING * LB_SLOT_DQ;
  if (long_tiled(a, a.bias_tiled)) hipLaunchKernelGGL((attn_bwd_dq_long_kernel<true, true>), grid, blk, lds_b, st, a, qblocks);
  else if (a.bias != nullptr) hipLaunchKernelGGL((attn_bwd_dq_long_kernel<true, false>), grid, blk, lds_b, st, a, qblocks);
  else hipLaunchKernelGGL((attn_bwd_dq_long_kernel<false, false>), grid, blk, lds_b, st, a, qblocks);
  int rc = xfm_check_launch("attn_bwd_dq_long");
  if (rc != XFM_OK || a.dbias == nullptr) return rc;
  return launch_attn_dbias_blocks(a, st);
}

// bias gradient (reads the delta a dQ kernel wrote): blocks of 128 x 128, the batch in `slices`
static int launch_attn_dbias_blocks(const AttnArgs& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    long_attr(attn_dbias_long_kernel<true>, 2 * LD_STAGE);
    attr_set = true;
  }
  int rc;
  const int qb = cdiv(a.Sq, 128), kb = cdiv(a.Sk, 128);
  const long per_entry = (long)a.H * a.Sq * a.bias_ld;
  int slices = dbias_long_slices(qb * kb * a.H, a.B, per_entry * 4);
  if (slices > 1 && a.dbias_ws == nullptr) slices = 1;  // no plane buffer: one workgroup per block walks the whole batch
  float* planes = slices > 1 ? a.dbias_ws : nullptr;
  hipLaunchKernelGGL(attn_dbias_long_kernel<true>, dim3(qb * kb * slices * a.H), dim3(512), 2 * LD_STAGE, st, a, qb, kb, slices, planes);
  rc = xfm_check_launch("attn_dbias_long");
  if (rc != XFM_OK || slices == 1) return rc;
  hipLaunchKernelGGL(dbias_reduce_kernel, dim3(cdiv(per_entry / 4, 256)), dim3(256), 0, st, planes, a.dbias, slices, per_entry);
  return xfm_check_launch("dbias_reduce");
}

static int launch_attn_bwd_long_dkv(const AttnArgs& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    long_attr(attn_bwd_dkv_long_kernel<true, true>, LB_RING * LB_SLOT_DKV);
    long_attr(attn_bwd_dkv_long_kernel<true, false>, LB_RING * LB_SLOT_DKV);
    long_attr(attn_bwd_dkv_long_kernel<false, false>, LB_RING * LB_SLOT_DKV);
    attr_set = true;
  }
  const int kblocks = cdiv(a.Sk, 256);
  const dim3 grid(kblocks * a.H * a.B), blk(512);
  const size_t lds_b = LB_RING * LB_SLOT_DKV;
  if (long_tiled(a, a.bias_t_tiled)) hipLaunchKernelGGL((attn_bwd_dkv_long_kernel<true, true>), grid, blk, lds_b, st, a, kblocks);
  else if (a.bias != nullptr) hipLaunchKernelGGL((attn_bwd_dkv_long_kernel<true, false>), grid, blk, lds_b, st, a, kblocks);
  else hipLaunchKernelGGL((attn_bwd_dkv_long_kernel<false, false>), grid, blk, lds_b, st, a, kblocks);
  return xfm_check_launch("attn_bwd_dkv_long");
}
