// One RobertaLayer per call (xroberta.py:405-473 of the reference): the launch sequence of the layer's forward and of its backward,
// on the native side.  No new arithmetic lives here -- every step is one of the kernels of gemm.hip / attention.hip / layernorm.hip,
// called through the same *_impl entry points as the one-kernel C ABI -- what this file owns is ORDER and STREAMS:
//
//   forward   qkv GEMM -> self-attention -> output GEMM -> dropout+residual+LayerNorm
//             [-> query GEMM -> (wait for the layer's K/V projection of the image states) -> grouped cross-attention -> output GEMM
//              -> dropout+residual+LayerNorm] -> FFN-in GEMM + GELU (gelu' stored) -> FFN-out GEMM -> dropout+residual+LayerNorm
//   backward  the mirror image on the launch stream; every weight-gradient GEMM, the cross-attention dK/dV kernel and the fold of
//             dK/dV into the image-state gradient go to `side_stream` behind an event recorded right after their dY exists.
//
// A text / fusion tower is ~50 launches per layer of 10-30 us kernels; driven kernel by kernel from Python (7-24 us of interpreter
// and ctypes work per wrapper call, tools/host_call_cost.py) the HOST bounds the tower: 8.4 ms of enqueue time against 9.8 ms
// until the GPU is done for the 12-layer fusion encoder, tools/fusion_host.py.  One call per layer leaves the launches themselves.
#include "common.h"
#include <string.h>

typedef xfm_rlayer_params RLP;
typedef xfm_rlayer_io RLIO;
typedef xfm_rlayer_bwd_args RLB;
typedef xfm_rlayer_layout_t RLL;

static long al256(long x) { return (x + 255) & ~255L; }

int xfm_rlayer_layout_impl(int R, int B, int T, int D, int H, int FF, int has_cross, int Nenc, int U, int xq_max, int flags, RLL* o) {
  XFM_REQUIRE(R > 0 && B > 0 && T > 0 && D > 0 && H > 0 && FF > 0 && o != nullptr, "rlayer_layout: bad geometry");
  const int dropout = flags & XFM_RL_DROPOUT;
  const bool f32s = (flags & XFM_RL_F32_STREAM) != 0;   // fp32 residual stream: z_k and the residual-branch gradients are fp32, y_k has an fp32 twin
  const long statld = (T + 3) / 4 * 4;
  const long rd = (long)R * D * 2, stat = (long)B * H * statld * 4, rvec = (long)R * 4;
  const long rz = f32s ? 2 * rd : rd;
  // cross-attention row statistics: per sequence, or per image in range mode
  const long xstat = xq_max > 0 ? (long)U * H * ((xq_max + 3) / 4 * 4) * 4 : stat;
  long off = 0;
  auto take = [&](long bytes) { const long r = off; off += al256(bytes); return r; };
  o->qkv = take(3 * rd); o->c1 = take(rd); o->lse1 = take(stat); o->h = take(rd);
  o->z1 = take(rz); o->m1 = take(rvec); o->r1 = take(rvec); o->y1 = take(rd);
  o->y1_32 = f32s ? take(2 * rd) : -1;
  if (has_cross) {
    o->q2 = take(rd); o->c2 = take(rd); o->lse2 = take(xstat);
    o->c2lo = take(rd);   // what the bf16 rounding of c2 lost: the backward gets delta = dO . (O + Olo) without a sweep over the keys
    o->z2 = take(rz); o->m2 = take(rvec); o->r2 = take(rvec); o->y2 = take(rd);
    o->y2_32 = f32s ? take(2 * rd) : -1;
  } else {
    o->q2 = o->c2 = o->c2lo = o->lse2 = o->z2 = o->m2 = o->r2 = o->y2 = o->y2_32 = -1;
  }
  o->hact = take((long)R * FF * 2); o->u = take((long)R * FF * 2);
  o->z3 = take(rz); o->m3 = take(rvec); o->r3 = take(rvec); o->y3 = take(rd);
  o->y3_32 = f32s ? take(2 * rd) : -1;
  o->fwd_bytes = off;
  off = 0;
  o->dh3 = take(rd); o->dres3 = f32s ? take(2 * rd) : (dropout ? take(rd) : o->dh3);
  o->du = take((long)R * FF * 2); o->d1a = take(rd);
  if (has_cross) {
    o->dh2 = take(rd); o->dres2 = f32s ? take(2 * rd) : (dropout ? take(rd) : o->dh2);
    o->dc2 = take(rd); o->dq2 = take(rd); o->delta2 = take(xstat); o->d2a = take(rd);
  } else {
    o->dh2 = o->dres2 = o->dc2 = o->dq2 = o->delta2 = o->d2a = -1;
  }
  o->dh1 = take(rd); o->dres1 = f32s ? take(2 * rd) : (dropout ? take(rd) : o->dh1);
  o->dc1 = take(rd); o->dqkv = take(3 * rd); o->delta1 = take(stat); o->dprev = take(rd);
  o->bwd_bytes = off;
  o->ws_main_bytes = (long)3 * xfm_ln_bwd_grid(R) * D * 4;
  long ws = xfm_gemm_tn_workspace_impl(R, D, FF);
  auto mx = [&](long v) { ws = v > ws ? v : ws; };
  mx(xfm_gemm_tn_workspace_impl(R, FF, D));
  mx(xfm_gemm_tn_workspace_impl(R, D, D));
  mx(xfm_gemm_tn_workspace_impl(R, 3 * D, D));
  if (has_cross && Nenc > 0 && U > 0) mx(xfm_gemm_tn_workspace_impl(U * Nenc, 2 * D, D));
  if (has_cross) mx(xfm_gemm_tn_batch_workspace_impl(3, R, D, D));
  o->ws_side_bytes = ws;
  return XFM_OK;
}

#define RL_TRY(expr)           \
  do {                         \
    const int rc_ = (expr);    \
    if (rc_ != XFM_OK) return rc_; \
  } while (0)

static void rl_drop(uint32_t thresh, float scale, uint32_t seed_hi, uint32_t ctr, uint32_t& t, float& s, uint32_t& lo, uint32_t& hi) {
  if (thresh == 0u) { t = 0u; s = 1.0f; lo = 0u; hi = 0u; }
  else { t = thresh; s = scale; lo = ctr; hi = seed_hi; }
}

static AttnArgs rl_self_attn(const RLP& p, const RLIO& io, const RLL& L, char* s) {
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  const int D = p.D;
  bf16* qkv = reinterpret_cast<bf16*>(s + L.qkv);
  a.q = qkv; a.k = qkv + D; a.v = qkv + 2 * D;
  a.q_rs = a.k_rs = a.v_rs = 3L * D;
  a.o = reinterpret_cast<bf16*>(s + L.c1); a.o_rs = D;
  a.lse = reinterpret_cast<float*>(s + L.lse1);
  a.key_keep = io.key_keep;
  a.B = io.B; a.H = p.H; a.Sq = io.T; a.Sk = io.T;
  a.scale = io.scale; a.causal = io.causal;
  rl_drop(io.att_thresh, io.att_scale, io.seed_hi, io.seed_ctr + 1, a.drop_thresh, a.drop_scale, a.seed_lo, a.seed_hi);
  a.stat_ld = (io.T + 3) / 4 * 4;
  a.q_start = a.k_start = io.seq_start;
  a.q_len = a.k_len = io.seq_len;
  return a;
}

static AttnArgs rl_cross_attn(const RLP& p, const RLIO& io, const RLL& L, char* s) {
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  const int D = p.D;
  a.q = reinterpret_cast<bf16*>(s + L.q2); a.q_rs = D;
  a.k = io.kv; a.v = io.kv + D; a.k_rs = a.v_rs = io.kv_ld;
  a.o = reinterpret_cast<bf16*>(s + L.c2); a.o_rs = D;
  a.o_lo = reinterpret_cast<bf16*>(s + L.c2lo);
  a.lse = reinterpret_cast<float*>(s + L.lse2);
  a.key_keep = io.enc_keep;
  a.B = io.B; a.H = p.H; a.Sq = io.T; a.Sk = io.Nenc;
  a.scale = io.scale; a.causal = 0;
  rl_drop(io.att_thresh, io.att_scale, io.seed_hi, io.seed_ctr + 3, a.drop_thresh, a.drop_scale, a.seed_lo, a.seed_hi);
  if (io.xq_start != nullptr) {  // range mode: one ragged problem per image (B = images, queries = the image's contiguous rows)
    a.B = io.U; a.Sq = io.xq_max;
    a.stat_ld = (io.xq_max + 3) / 4 * 4;
    a.q_start = io.xq_start; a.q_len = io.xq_len;
    return a;
  }
  a.stat_ld = (io.T + 3) / 4 * 4;
  a.grp_start = io.grp_start; a.grp_rows = io.grp_rows; a.n_groups = io.U;
  a.q_start = io.seq_start; a.q_len = io.seq_len;
  return a;
}

static LnFwd rl_ln_fwd(const RLIO& io, const bf16* h, const bf16* res, const float* res32, const float* w, const float* b, char* s, long z,
                       long y, long y32, long m, long r, float eps, uint32_t ctr) {
  LnFwd f;
  memset(&f, 0, sizeof(f));
  f.h = h; f.res = res; f.w = w; f.b = b;
  if (io.f32_stream) {   // fp32 residual stream: fp32 z for the backward, fp32 twin of y for the next residual; a layer input without an
    f.res32 = res32;     // fp32 twin (res32 NULL) enters the stream through its bf16 values
    f.z32_out = reinterpret_cast<float*>(s + z); f.y32 = reinterpret_cast<float*>(s + y32);
  } else {
    f.z_out = reinterpret_cast<bf16*>(s + z);
  }
  f.y = reinterpret_cast<bf16*>(s + y);
  f.mean = reinterpret_cast<float*>(s + m); f.rstd = reinterpret_cast<float*>(s + r);
  f.rows = io.R; f.rows_per_sample = 1; f.eps = eps;
  rl_drop(io.hid_thresh, io.hid_scale, io.seed_hi, ctr, f.drop_thresh, f.drop_scale, f.seed_lo, f.seed_hi);
  return f;
}

static bool rl_cross(const RLP& p, const RLIO& io) { return p.has_cross && io.kv != nullptr && io.Nenc > 0 && io.U > 0; }

static int rl_check(const RLP& p, const RLIO& io) {
  XFM_REQUIRE(io.R > 0 && io.B > 0 && io.T > 0 && io.x != nullptr && io.slab != nullptr, "rlayer: bad geometry / null buffers");
  XFM_REQUIRE(p.D == p.H * 64, "rlayer: head_dim must be 64");
  XFM_REQUIRE((io.seq_start == nullptr) == (io.seq_len == nullptr), "rlayer: packed rows need both start and len");
  XFM_REQUIRE(!rl_cross(p, io) || (io.grp_start != nullptr && io.grp_rows != nullptr) || (io.xq_start != nullptr && io.xq_len != nullptr && io.xq_max > 0),
              "rlayer: cross-attention needs the grouped row lists or the per-image row ranges");
  return XFM_OK;
}

int xfm_rlayer_fwd_impl(const RLP& p, const RLIO& io, hipStream_t st) {
  RL_TRY(rl_check(p, io));
  const bool cross = rl_cross(p, io);
  RLL L;
  RL_TRY(xfm_rlayer_layout_impl(io.R_alloc > 0 ? io.R_alloc : io.R, io.B_alloc > 0 ? io.B_alloc : io.B, io.T, p.D, p.H, p.FF, p.has_cross,
                                io.Nenc, io.U, io.xq_start != nullptr ? io.xq_max : 0,
                                (io.hid_thresh != 0u ? XFM_RL_DROPOUT : 0) | (io.f32_stream ? XFM_RL_F32_STREAM : 0), &L));
  char* s = reinterpret_cast<char*>(io.slab);
  const int R = io.R, D = p.D, FF = p.FF;
  const bool zf = io.zero_fill && io.seq_start != nullptr;
  bf16* h = reinterpret_cast<bf16*>(s + L.h);
  auto B16 = [&](long off) { return reinterpret_cast<bf16*>(s + off); };

  // ---- self-attention block (xroberta.py:419-425, 201-304)
  RL_TRY(xfm_gemm_nt_impl(io.x, D, p.wqkv, D, B16(L.qkv), 3L * D, p.bqkv, nullptr, 0, R, 3 * D, D, EPI_BF16, 0, st));
  if (zf) (void)hipMemsetAsync(s + L.c1, 0, (size_t)R * D * 2, st);
  RL_TRY(xfm_attn_fwd_impl(rl_self_attn(p, io, L, s), st));
  RL_TRY(xfm_gemm_nt_impl(B16(L.c1), D, p.wo, D, h, D, p.bo, nullptr, 0, R, D, D, EPI_BF16, 0, st));
  auto F32 = [&](long off) { return io.f32_stream ? reinterpret_cast<const float*>(s + off) : nullptr; };
  RL_TRY(xfm_ln_fwd_impl(rl_ln_fwd(io, h, io.x, io.x32, p.ln1_w, p.ln1_b, s, L.z1, L.y1, L.y1_32, L.m1, L.r1, p.eps, io.seed_ctr + 2), D, LN_POST, st));
  const bf16* y = B16(L.y1);
  const float* y32 = F32(L.y1_32);
  uint32_t ctr3 = io.seed_ctr + 3;
  // ---- cross-attention block (xroberta.py:431-458)
  if (cross) {
    RL_TRY(xfm_gemm_nt_impl(y, D, p.wq2, D, B16(L.q2), D, p.bq2, nullptr, 0, R, D, D, EPI_BF16, 0, st));
    if (io.kv_event != nullptr) (void)hipStreamWaitEvent(st, reinterpret_cast<hipEvent_t>(io.kv_event), 0);
    if (zf) (void)hipMemsetAsync(s + L.c2, 0, (size_t)R * D * 2, st);
    RL_TRY(xfm_attn_fwd_impl(rl_cross_attn(p, io, L, s), st));
    RL_TRY(xfm_gemm_nt_impl(B16(L.c2), D, p.wo2, D, h, D, p.bo2, nullptr, 0, R, D, D, EPI_BF16, 0, st));
    RL_TRY(xfm_ln_fwd_impl(rl_ln_fwd(io, h, y, y32, p.ln2_w, p.ln2_b, s, L.z2, L.y2, L.y2_32, L.m2, L.r2, p.eps, io.seed_ctr + 4), D, LN_POST, st));
    y = B16(L.y2);
    y32 = F32(L.y2_32);
    ctr3 = io.seed_ctr + 5;
  }
  // ---- feed-forward block (xroberta.py:460-473): GELU fused into the first GEMM, gelu'(x) kept for the backward
  RL_TRY(xfm_gemm_nt_impl(y, D, p.wi, D, B16(L.hact), FF, p.bi, B16(L.u), FF, R, FF, D, EPI_GELU, 0, st));
  RL_TRY(xfm_gemm_nt_impl(B16(L.hact), FF, p.wout, FF, h, D, p.bout, nullptr, 0, R, D, FF, EPI_BF16, 0, st));
  RL_TRY(xfm_ln_fwd_impl(rl_ln_fwd(io, h, y, y32, p.ln3_w, p.ln3_b, s, L.z3, L.y3, L.y3_32, L.m3, L.r3, p.eps, ctr3), D, LN_POST, st));
  return XFM_OK;
}

// Events that order the second stream behind the launch stream: a ring (a wait captures the event's state at the time of the call,
// so an event can be re-recorded as soon as its wait has been enqueued).
static hipEvent_t rl_event() {
  static hipEvent_t ring[256];
  static bool made = false;
  static unsigned next = 0;
  if (!made) {
    for (auto& e : ring) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    made = true;
  }
  return ring[next++ & 255u];
}

static LnBwd rl_ln_bwd(const RLIO& io, const bf16* dy1, const void* dy2, char* s, long z, long m, long r, const float* w, bf16* dh,
                       void* dres, uint32_t ctr) {
  LnBwd g;
  memset(&g, 0, sizeof(g));
  g.dy1 = dy1;
  if (io.f32_stream) {   // the residual-branch gradient travels in fp32 from LayerNorm to LayerNorm (dy2 may be NULL: the tower's last layer)
    g.dy32 = reinterpret_cast<const float*>(dy2);
    g.x32 = reinterpret_cast<const float*>(s + z);
    g.dres32 = reinterpret_cast<float*>(dres);
  } else {
    g.dy2 = reinterpret_cast<const bf16*>(dy2);
    g.x16 = reinterpret_cast<const bf16*>(s + z);
    g.dres = reinterpret_cast<bf16*>(dres);
  }
  g.mean = reinterpret_cast<const float*>(s + m); g.rstd = reinterpret_cast<const float*>(s + r);
  g.w = w; g.dh = dh;
  g.rows = io.R; g.rows_per_sample = 1;
  rl_drop(io.hid_thresh, io.hid_scale, io.seed_hi, ctr, g.drop_thresh, g.drop_scale, g.seed_lo, g.seed_hi);
  return g;
}

int xfm_rlayer_bwd_impl(const RLP& p, const RLIO& io, const RLB& b, hipStream_t st) {
  RL_TRY(rl_check(p, io));
  XFM_REQUIRE(b.bslab != nullptr && b.dy_a != nullptr && b.ws_main != nullptr, "rlayer_bwd: null buffers");
  const bool cross = rl_cross(p, io);
  XFM_REQUIRE(!cross || (b.enc != nullptr && b.dkv != nullptr), "rlayer_bwd: cross-attention needs enc and dkv");
  RLL L;
  RL_TRY(xfm_rlayer_layout_impl(io.R_alloc > 0 ? io.R_alloc : io.R, io.B_alloc > 0 ? io.B_alloc : io.B, io.T, p.D, p.H, p.FF, p.has_cross,
                                io.Nenc, io.U, io.xq_start != nullptr ? io.xq_max : 0,
                                (io.hid_thresh != 0u ? XFM_RL_DROPOUT : 0) | (io.f32_stream ? XFM_RL_F32_STREAM : 0), &L));
  XFM_REQUIRE(b.ws_main_bytes >= L.ws_main_bytes && (L.ws_side_bytes == 0 || (b.ws_side != nullptr && b.ws_side_bytes >= L.ws_side_bytes)),
              "rlayer_bwd: workspaces too small");
  char* s = reinterpret_cast<char*>(io.slab);
  char* g = reinterpret_cast<char*>(b.bslab);
  const int R = io.R, D = p.D, FF = p.FF;
  const bool zf = io.zero_fill && io.seq_start != nullptr;
  hipStream_t side = b.side_stream != nullptr ? reinterpret_cast<hipStream_t>(b.side_stream) : st;
  const bool two = side != st;
  auto S16 = [&](long off) { return reinterpret_cast<bf16*>(s + off); };
  auto G16 = [&](long off) { return reinterpret_cast<bf16*>(g + off); };
  auto fork = [&]() {  // what the launch stream has enqueued so far happens before what the second stream gets next
    if (!two) return;
    hipEvent_t e = rl_event();
    (void)hipEventRecord(e, st);
    (void)hipStreamWaitEvent(side, e, 0);
  };
  static const bool skip_wgrad = getenv("XFM_RL_SKIP_WGRAD") != nullptr;   // timing experiment only: WRONG gradients
  auto wgrad = [&](const bf16* dY, long ldy, const bf16* X, long ldx, float* dW, long ldw, float* db, int M, int N, int K) {
    if (skip_wgrad || b.defer_wgrad) return (int)XFM_OK;
    fork();
    return xfm_gemm_tn_impl(dY, ldy, X, ldx, dW, ldw, db, M, N, K, 0, b.ws_side, b.ws_side_bytes, side);
  };
  // XFM_TN_BATCH=1: the D x D weight gradients of a cross-attention layer (out-projection of both attention blocks, cross query) go
  // out as ONE batched launch once the last of their dY exists.  Measured on the step: 1 ms SLOWER than one launch each as their dY
  // appear (the three then trail the activation-gradient chain in one 432-workgroup lump instead of filling its gaps) -- off.
  static const bool tn_batch = getenv("XFM_TN_BATCH") ? atoi(getenv("XFM_TN_BATCH")) != 0 : false;
  const bool batch3 = cross && tn_batch && !b.defer_wgrad;
  const uint32_t c_att = io.seed_ctr + 1, c_h1 = io.seed_ctr + 2, c_att2 = io.seed_ctr + 3, c_h2 = io.seed_ctr + 4;
  const uint32_t c_h3 = cross ? io.seed_ctr + 5 : io.seed_ctr + 3;
  const bf16* y_in = cross ? S16(L.y2) : S16(L.y1);  // input of the feed-forward block
  // LayerNorm backward column sums: folded right behind each kernel (b.ws_main), or -- b.ln_items -- left as partials in the caller's
  // per-LayerNorm slices for ONE batched reduce over the whole tower (three 7-us kernels less per layer on this chain)
  const bool ln_defer = b.ln_items != nullptr && b.ln_count != nullptr && b.ln_ws != nullptr;
  auto ln_bwd = [&](LnBwd a, int k, float* dg, float* db, float* dbias) {
    if (!ln_defer) return xfm_ln_bwd_impl(a, D, LN_POST, dg, db, dbias, nullptr, b.ws_main, b.ws_main_bytes, st);
    a.defer = b.ln_items + (*b.ln_count)++;
    return xfm_ln_bwd_impl(a, D, LN_POST, dg, db, dbias, nullptr, b.ln_ws + (long)k * b.ln_ws_stride, b.ln_ws_stride * 4, st);
  };
  (void)c_att2;

  // ---- feed-forward block
  XFM_REQUIRE(!io.f32_stream || b.dy_b == nullptr, "rlayer_bwd: with the fp32 stream the second gradient is dy_b32");
  RL_TRY(ln_bwd(rl_ln_bwd(io, b.dy_a, io.f32_stream ? (const void*)b.dy_b32 : (const void*)b.dy_b, s, L.z3, L.m3, L.r3, p.ln3_w, G16(L.dh3),
                          g + L.dres3, c_h3), 0, p.dln3_w, p.dln3_b, p.dbout));
  RL_TRY(wgrad(G16(L.dh3), D, S16(L.hact), FF, p.dwout, FF, nullptr, R, D, FF));
  RL_TRY(xfm_gemm_nt_impl(G16(L.dh3), D, p.wout_t, p.ld_wout_t, G16(L.du), FF, nullptr, S16(L.u), FF, R, FF, D, EPI_DGELU, 0, st));
  RL_TRY(wgrad(G16(L.du), FF, y_in, D, p.dwi, D, p.dbi, R, FF, D));
  RL_TRY(xfm_gemm_nt_impl(G16(L.du), FF, p.wi_t, p.ld_wi_t, G16(L.d1a), D, nullptr, nullptr, 0, R, D, FF, EPI_BF16, 0, st));
  const bf16* in_a = G16(L.d1a);
  const void* in_b = g + L.dres3;
  // ---- cross-attention block
  if (cross) {
    RL_TRY(ln_bwd(rl_ln_bwd(io, in_a, in_b, s, L.z2, L.m2, L.r2, p.ln2_w, G16(L.dh2), g + L.dres2, c_h2), 1, p.dln2_w, p.dln2_b, p.dbo2));
    if (!batch3) RL_TRY(wgrad(G16(L.dh2), D, S16(L.c2), D, p.dwo2, D, nullptr, R, D, D));
    RL_TRY(xfm_gemm_nt_impl(G16(L.dh2), D, p.wo2_t, p.ld_wo2_t, G16(L.dc2), D, nullptr, nullptr, 0, R, D, D, EPI_BF16, 0, st));
    if (zf) (void)hipMemsetAsync(g + L.dq2, 0, (size_t)R * D * 2, st);
    AttnArgs a = rl_cross_attn(p, io, L, s);
    a.dout = G16(L.dc2); a.do_rs = D;
    a.dq = G16(L.dq2); a.dq_rs = D;
    a.dk = b.dkv; a.dv = b.dkv + D; a.dk_rs = a.dv_rs = b.dkv_ld;
    a.delta = reinterpret_cast<float*>(g + L.delta2);
    // dQ stays on the activation-gradient chain; dK/dV only feed the K/V weight gradient and the image-state gradient
    a.bwd_phase = 1;
    RL_TRY(xfm_attn_bwd_impl(a, st));
    fork();
    a.bwd_phase = 2;
    RL_TRY(xfm_attn_bwd_impl(a, side));
    if (!batch3) RL_TRY(wgrad(G16(L.dq2), D, S16(L.y1), D, p.dwq2, D, p.dbq2, R, D, D));
    if (!skip_wgrad && !b.defer_wgrad)
      RL_TRY(xfm_gemm_tn_impl(b.dkv, b.dkv_ld, b.enc, D, p.dwkv2, D, p.dbkv2, io.U * io.Nenc, 2 * D, D, 0, b.ws_side, b.ws_side_bytes, side));
    if (b.denc32 != nullptr)
      RL_TRY(xfm_gemm_nt_impl(b.dkv, b.dkv_ld, p.wkv2_t, p.ld_wkv2_t, b.denc32, D, nullptr, nullptr, 0, io.U * io.Nenc, D, 2 * D, EPI_F32_ACC, 0, side));
    RL_TRY(xfm_gemm_nt_impl(G16(L.dq2), D, p.wq2_t, p.ld_wq2_t, G16(L.d2a), D, nullptr, nullptr, 0, R, D, D, EPI_BF16, 0, st));
    in_a = G16(L.d2a);
    in_b = g + L.dres2;
  }
  // ---- self-attention block
  RL_TRY(ln_bwd(rl_ln_bwd(io, in_a, in_b, s, L.z1, L.m1, L.r1, p.ln1_w, G16(L.dh1), g + L.dres1, c_h1), 2, p.dln1_w, p.dln1_b, p.dbo));
  if (batch3) {
    const void* dys[3] = {G16(L.dh2), G16(L.dq2), G16(L.dh1)};
    const void* xs[3] = {S16(L.c2), S16(L.y1), S16(L.c1)};
    float* dws[3] = {p.dwo2, p.dwq2, p.dwo};
    float* dbs[3] = {nullptr, p.dbq2, nullptr};
    fork();
    RL_TRY(xfm_gemm_tn_batch_impl(3, dys, D, xs, D, dws, D, dbs, R, D, D, b.ws_side, b.ws_side_bytes, side));
  } else {
    RL_TRY(wgrad(G16(L.dh1), D, S16(L.c1), D, p.dwo, D, nullptr, R, D, D));
  }
  RL_TRY(xfm_gemm_nt_impl(G16(L.dh1), D, p.wo_t, p.ld_wo_t, G16(L.dc1), D, nullptr, nullptr, 0, R, D, D, EPI_BF16, 0, st));
  if (zf) (void)hipMemsetAsync(g + L.dqkv, 0, (size_t)R * 3 * D * 2, st);
  {
    AttnArgs a = rl_self_attn(p, io, L, s);
    bf16* dqkv = G16(L.dqkv);
    a.dout = G16(L.dc1); a.do_rs = D;
    a.dq = dqkv; a.dk = dqkv + D; a.dv = dqkv + 2 * D;
    a.dq_rs = a.dk_rs = a.dv_rs = 3L * D;
    a.delta = reinterpret_cast<float*>(g + L.delta1);
    a.seed_lo = a.drop_thresh ? c_att : 0u;
    RL_TRY(xfm_attn_bwd_impl(a, st));
  }
  RL_TRY(wgrad(G16(L.dqkv), 3L * D, io.x, D, p.dwqkv, D, p.dbqkv, R, 3 * D, D));
  if (b.need_dprev)
    RL_TRY(xfm_gemm_nt_impl(G16(L.dqkv), 3L * D, p.wqkv_t, p.ld_wqkv_t, G16(L.dprev), D, nullptr, nullptr, 0, R, D, 3 * D, EPI_BF16, 0, st));
  return XFM_OK;
}
