// C-ABI of libxfm_hip.so (see include/xfm_hip.h).  Single translation unit: the kernel files are included here so
// that no relocatable device code is needed.
#include "common.h"

#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void xfm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int xfm_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    xfm_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return XFM_E_LAUNCH;
  }
  return XFM_OK;
}

// compute units of the current device (persistent kernels launch one workgroup per CU)
int xfm_cu_count() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return n;
}

#include "gemm.hip"
#include "layernorm.hip"
#include "attention.hip"
#include "elementwise.hip"
#include "encoder.hip"
#include "losses.hip"
#include "dp.hip"

#define ST(s) ((hipStream_t)(s))
#define NOTNULL(p, name) XFM_REQUIRE((p) != nullptr, "%s: null argument struct", name)

extern "C" {

const char* xfm_last_error(void) { return g_err; }
int xfm_abi_version(void) { return XFM_ABI_VERSION; }

int xfm_gemm_nt(const xfm_bf16* A, long lda, const xfm_bf16* B, long ldb, void* C, long ldc, const float* bias, xfm_bf16* aux,
                long ldaux, int M, int N, int K, int epilogue, int tile_hint, void* stream) {
  XFM_REQUIRE(A && B && C, "gemm_nt: null operand");
  return xfm_gemm_nt_impl(A, lda, B, ldb, C, ldc, bias, aux, ldaux, M, N, K, epilogue, tile_hint, ST(stream));
}

int xfm_gemm_nt_plan(int M, int N, int K, int epilogue, int tile_hint, int* cfg, int* rows_a) {
  return xfm_gemm_nt_plan_impl(M, N, K, epilogue, tile_hint, cfg, rows_a);
}
long xfm_gemm_nt_ksplit_workspace(int M, int N, int K) { return xfm_gemm_nt_ksplit_workspace_impl(M, N, K); }
int xfm_gemm_nt_ksplit(const xfm_bf16* A, long lda, const xfm_bf16* B, long ldb, void* out, long ldo, int out_bf16, const float* bias, int M,
                       int N, int K, float* workspace, long workspace_bytes, void* stream) {
  XFM_REQUIRE(A && B && out, "gemm_nt_ksplit: null operand");
  return xfm_gemm_nt_ksplit_impl(A, lda, B, ldb, out, ldo, out_bf16, bias, M, N, K, workspace, workspace_bytes, ST(stream));
}
long xfm_gemm_tn_workspace(int M, int N, int K) { return xfm_gemm_tn_workspace_impl(M, N, K); }

int xfm_gemm_tn(const xfm_bf16* dY, long ldy, const xfm_bf16* X, long ldx, float* dW, long ldw, float* dbias, int M, int N,
                int K, int splits_hint, float* workspace, long workspace_bytes, void* stream) {
  XFM_REQUIRE(dY && X && dW, "gemm_tn: null operand");
  return xfm_gemm_tn_impl(dY, ldy, X, ldx, dW, ldw, dbias, M, N, K, splits_hint, workspace, workspace_bytes, ST(stream));
}
long xfm_gemm_tn_batch_workspace(int nb, int M, int N, int K) { return xfm_gemm_tn_batch_workspace_impl(nb, M, N, K); }
int xfm_gemm_tn_batch(int nb, const void* const* dY, long ldy, const void* const* X, long ldx, float* const* dW, long ldw, float* const* dbias,
                      int M, int N, int K, float* workspace, long workspace_bytes, void* stream) {
  XFM_REQUIRE(nb >= 1 && nb <= 4 && dY && X && dW, "gemm_tn_batch: 1..4 problems, non-null pointer arrays");
  for (int i = 0; i < nb; ++i) XFM_REQUIRE(dY[i] && X[i] && dW[i], "gemm_tn_batch: null operand in problem %d", i);
  return xfm_gemm_tn_batch_impl(nb, dY, ldy, X, ldx, dW, ldw, dbias, M, N, K, workspace, workspace_bytes, ST(stream));
}

long xfm_gemm_tn_group_workspace(int n, const xfm_tn_item* items, int M) { return xfm_gemm_tn_group_workspace_impl(n, items, M); }
int xfm_gemm_tn_group(int n, const xfm_tn_item* items, int M, float* workspace, long workspace_bytes, void* stream) {
  return xfm_gemm_tn_group_impl(n, items, M, workspace, workspace_bytes, ST(stream));
}

int xfm_cast_transpose_batch(const xfm_cast_item* items, int n_items, long total_tiles, void* stream) {
  return xfm_cast_transpose_batch_impl(items, n_items, total_tiles, ST(stream));
}

int xfm_cast_transpose(const float* w, int N, int K, xfm_bf16* wb, long ldb, xfm_bf16* wt, long ldt, void* stream) {
  XFM_REQUIRE(w && (wb || wt), "cast_transpose: null operand");
  return xfm_cast_transpose_impl(w, N, K, wb, ldb, wt, ldt, ST(stream));
}

long xfm_colsum_workspace(int M, int N) { return (long)256 * N * 4; }
int xfm_colsum(const xfm_bf16* Y, long ldy, int M, int N, float* out, float* workspace, long workspace_bytes, void* stream) {
  XFM_REQUIRE(Y && out, "colsum: null operand");
  return xfm_colsum_impl(Y, ldy, M, N, out, workspace, workspace_bytes, ST(stream));
}

int xfm_layernorm_fwd(const xfm_ln_fwd_args* a, int D, int mode, void* stream) {
  NOTNULL(a, "layernorm_fwd");
  XFM_REQUIRE(a->w && a->b && a->y && a->mean && a->rstd, "layernorm_fwd: null operand");
  XFM_REQUIRE(mode != XFM_LN_PLAIN || a->x32 || a->x16, "layernorm_fwd: PLAIN needs x32 or x16");
  XFM_REQUIRE(mode != XFM_LN_POST || (a->h && (a->res || a->res32) && (a->z_out || a->z32_out)),
              "layernorm_fwd: POST needs h, res or res32, z_out or z32_out");
  XFM_REQUIRE(mode != XFM_LN_LS || (a->x32 && a->h && a->ls_gamma && a->x_out && a->rows_per_sample > 0),
              "layernorm_fwd: LS needs x32, h, ls_gamma, x_out, rows_per_sample");
  return xfm_ln_fwd_impl(*a, D, mode, ST(stream));
}

int xfm_reduce_sets_batch(int n, const xfm_reduce_item* items, void* stream) { return xfm_reduce_sets_batch_impl(n, items, ST(stream)); }
long xfm_layernorm_bwd_workspace(int rows, int D, int mode) {
  const int nset = mode == XFM_LN_PLAIN ? 2 : (mode == XFM_LN_POST ? 3 : 4);
  return (long)nset * xfm_ln_bwd_grid(rows) * D * 4;
}
int xfm_layernorm_bwd(const xfm_ln_bwd_args* a, int D, int mode, float* dgamma, float* dbeta, float* dbias, float* dls,
                      float* workspace, long workspace_bytes, void* stream) {
  NOTNULL(a, "layernorm_bwd");
  XFM_REQUIRE(a->dy1 && a->mean && a->rstd && a->w && (a->x32 || a->x16), "layernorm_bwd: null operand");
  XFM_REQUIRE(mode != XFM_LN_POST || a->dh, "layernorm_bwd: POST needs dh");
  XFM_REQUIRE(mode == XFM_LN_POST || a->dres32 == nullptr, "layernorm_bwd: dres32 is a POST output");
  XFM_REQUIRE(mode != XFM_LN_LS || (a->dh && a->dstream && a->h && a->ls_gamma && a->rows_per_sample > 0),
              "layernorm_bwd: LS needs dh, dstream, h, ls_gamma, rows_per_sample");
  return xfm_ln_bwd_impl(*a, D, mode, dgamma, dbeta, dbias, dls, workspace, workspace_bytes, ST(stream));
}

int xfm_attn_fwd(const xfm_attn_args* a, void* stream) {
  NOTNULL(a, "attn_fwd");
  XFM_REQUIRE(a->q && a->k && a->v && a->o && a->lse, "attn_fwd: null operand");
  return xfm_attn_fwd_impl(*a, ST(stream));
}
int xfm_bias_tile(const float* bias, int H, int S, long ld, float scale, float* tiled, float* tiled_t, void* stream) {
  return xfm_bias_tile_impl(bias, H, S, ld, scale, tiled, tiled_t, ST(stream));
}
int xfm_attn_bwd(const xfm_attn_args* a, void* stream) {
  NOTNULL(a, "attn_bwd");
  XFM_REQUIRE(a->q && a->k && a->v && a->o && a->lse, "attn_bwd: null operand");
  return xfm_attn_bwd_impl(*a, ST(stream));
}
long xfm_attn_bwd_workspace(const xfm_attn_args* a) { return a == nullptr ? 0 : xfm_attn_bwd_workspace_impl(*a); }
int xfm_rows_index_sum(const xfm_bf16* src, const int* index, int R, int U, long len, xfm_bf16* dst, void* stream) {
  XFM_REQUIRE(src && index && dst, "rows_index_sum: null operand");
  return xfm_rows_index_sum_impl(src, index, R, U, len, dst, ST(stream));
}
int xfm_relpos_gather(const float* table, const int* index, int H, int N, long ld, float* dense, float* dense_t, void* stream) {
  XFM_REQUIRE(table && index && dense, "relpos_gather: null operand");
  return xfm_relpos_gather_impl(table, index, H, N, ld, dense, dense_t, ST(stream));
}
int xfm_relpos_grid_grad(const float* ddense, int H, int G, long ld, float* dtable, void* stream) {
  XFM_REQUIRE(ddense && dtable, "relpos_grid_grad: null operand");
  return xfm_relpos_grid_grad_impl(ddense, H, G, ld, dtable, ST(stream));
}
int xfm_relpos_scatter_sorted(const float* ddense, const int* order, const int* start, int entries, int H, int N, long ld,
                              float* dtable, void* stream) {
  XFM_REQUIRE(ddense && order && start && dtable, "relpos_scatter_sorted: null operand");
  return xfm_relpos_scatter_sorted_impl(ddense, order, start, entries, H, N, ld, dtable, ST(stream));
}

int xfm_relpos_scatter(const float* ddense, const int* index, int H, int N, long ld, float* dtable, void* stream) {
  XFM_REQUIRE(ddense && index && dtable, "relpos_scatter: null operand");
  return xfm_relpos_scatter_impl(ddense, index, H, N, ld, dtable, ST(stream));
}

int xfm_patchify(const float* image, int B, int C, int H, int W, int P, xfm_bf16* out, void* stream) {
  XFM_REQUIRE(image && out, "patchify: null operand");
  return xfm_patchify_impl(image, B, C, H, W, P, out, ST(stream));
}
int xfm_vit_tokens_fwd(const float* tok, const float* cls, const float* mask_token, const uint8_t* mask, int Bt, int Bx, int P,
                       int D, float* x0, void* stream) {
  XFM_REQUIRE(tok && cls && x0 && (mask == nullptr || mask_token != nullptr), "vit_tokens_fwd: null operand");
  return xfm_vit_tokens_fwd_impl(tok, cls, mask_token, mask, Bt, Bx, P, D, x0, ST(stream));
}
int xfm_vit_tokens_bwd(const float* dx0, const uint8_t* mask, int Bt, int Bx, int P, int D, float* dtok, float* dcls,
                       float* dmask_token, void* stream) {
  XFM_REQUIRE(dx0 && dtok && dcls && (mask == nullptr || dmask_token != nullptr), "vit_tokens_bwd: null operand");
  return xfm_vit_tokens_bwd_impl(dx0, mask, Bt, Bx, P, D, dtok, dcls, dmask_token, ST(stream));
}
int xfm_pool_rows_fwd(xfm_bf16* y, int B, int N, int D, void* stream) {
  XFM_REQUIRE(y, "pool_rows_fwd: null operand");
  return xfm_pool_rows_fwd_impl(y, B, N, D, ST(stream));
}
int xfm_pool_rows_bwd(const xfm_bf16* dy, int B, int N, int D, xfm_bf16* out, void* stream) {
  XFM_REQUIRE(dy && out, "pool_rows_bwd: null operand");
  return xfm_pool_rows_bwd_impl(dy, B, N, D, out, ST(stream));
}
int xfm_mim_loss_fwd(const xfm_bf16* x, const xfm_bf16* t, const uint8_t* mask, int B, int N, int D, float* sums, void* stream) {
  XFM_REQUIRE(x && t && mask && sums, "mim_loss_fwd: null operand");
  return xfm_mim_loss_fwd_impl(x, t, mask, B, N, D, sums, ST(stream));
}
int xfm_mim_loss_bwd(const xfm_bf16* x, const xfm_bf16* t, const uint8_t* mask, const float* sums, const float* gout, int cls_term,
                     int B, int N, int D, xfm_bf16* dx, void* stream) {
  XFM_REQUIRE(x && t && mask && sums && gout && dx, "mim_loss_bwd: null operand");
  return xfm_mim_loss_bwd_impl(x, t, mask, sums, gout, cls_term, B, N, D, dx, ST(stream));
}

int xfm_mim_masks(int B, int GH, int GW, int num, int min_num, float min_aspect, float max_aspect, uint64_t seed, uint8_t* out,
                  int* delta_hist, void* stream) {
  XFM_REQUIRE(out != nullptr, "mim_masks: null output");
  return xfm_mim_masks_impl(B, GH, GW, num, min_num, min_aspect, max_aspect, seed, out, delta_hist, ST(stream));
}
int xfm_embed_ln_fwd(const xfm_embed_args* a, int D, void* stream) {
  NOTNULL(a, "embed_ln_fwd");
  XFM_REQUIRE(a->ids && a->word && a->pos && a->type && a->w && a->b && a->y && a->mean && a->rstd && a->pos_ids,
              "embed_ln_fwd: null operand");
  return xfm_emb_fwd_impl(*a, D, ST(stream));
}
long xfm_embed_ln_bwd_workspace(int rows, int D) { return (long)3 * emb_grid(rows) * D * 4; }
int xfm_embed_ln_bwd(const xfm_embed_args* a, int D, float* dgamma, float* dbeta, float* dtype, float* workspace,
                     long workspace_bytes, void* stream) {
  NOTNULL(a, "embed_ln_bwd");
  XFM_REQUIRE(a->ids && a->word && a->pos && a->type && a->w && a->dy && a->dword && a->dpos && a->mean && a->rstd && a->pos_ids,
              "embed_ln_bwd: null operand");
  return xfm_emb_bwd_impl(*a, D, dgamma, dbeta, dtype, workspace, workspace_bytes, ST(stream));
}

int xfm_rows_gather(const xfm_bf16* src, const int* index, int R, int D, xfm_bf16* dst, void* stream) {
  XFM_REQUIRE(src && index && dst, "rows_gather: null operand");
  return xfm_rows_gather_impl(src, index, R, D, dst, ST(stream));
}
int xfm_rows_scatter_add(const xfm_bf16* src, const int* index, int R, int D, float* dst32, void* stream) {
  XFM_REQUIRE(src && index && dst32, "rows_scatter_add: null operand");
  return xfm_rows_scatter_add_impl(src, index, R, D, dst32, ST(stream));
}
int xfm_rows_segment_sum(const float* src, const int64_t* perm, const int64_t* sorted_key, long R, int D, long skip_key, float* out,
                         void* stream) {
  XFM_REQUIRE(R == 0 || (src && perm && sorted_key && out), "rows_segment_sum: null operand");
  return xfm_rows_segment_sum_impl(src, perm, sorted_key, R, D, skip_key, out, ST(stream));
}

int xfm_rlayer_layout(int R, int B, int T, int D, int H, int FF, int has_cross, int Nenc, int U, int xq_max, int flags,
                      xfm_rlayer_layout_t* out) {
  return xfm_rlayer_layout_impl(R, B, T, D, H, FF, has_cross, Nenc, U, xq_max, flags, out);
}
int xfm_rlayer_fwd(const xfm_rlayer_params* p, const xfm_rlayer_io* io, void* stream) {
  XFM_REQUIRE(p != nullptr && io != nullptr, "rlayer_fwd: null argument struct");
  return xfm_rlayer_fwd_impl(*p, *io, ST(stream));
}
int xfm_rlayer_bwd(const xfm_rlayer_params* p, const xfm_rlayer_io* io, const xfm_rlayer_bwd_args* b, void* stream) {
  XFM_REQUIRE(p != nullptr && io != nullptr && b != nullptr, "rlayer_bwd: null argument struct");
  return xfm_rlayer_bwd_impl(*p, *io, *b, ST(stream));
}

int xfm_rownorm_fwd(const float* x, int R, int E, float* y, float* inv, void* stream) {
  XFM_REQUIRE(x && y && inv, "rownorm_fwd: null operand");
  return xfm_rownorm_fwd_impl(x, R, E, y, inv, ST(stream));
}
int xfm_rownorm_bwd(const float* dy, const float* y, const float* inv, int R, int E, float* dx, void* stream) {
  XFM_REQUIRE(dy && y && inv && dx, "rownorm_bwd: null operand");
  return xfm_rownorm_bwd_impl(dy, y, inv, R, E, dx, ST(stream));
}
int xfm_itc_fwd(const float* I, const float* T, const float* temp, int N, int E, float* lse, float* loss_sum, const int64_t* idx, float* cnt,
                void* stream) {
  XFM_REQUIRE(I && T && temp && lse && loss_sum, "itc_fwd: null operand");
  return xfm_itc_fwd_impl(I, T, temp, N, E, lse, loss_sum, idx, cnt, ST(stream));
}
int xfm_itc_bwd(const float* I, const float* T, const float* temp, const float* lse, const float* g, int N, int E, float* dI, float* dT,
                float* dtemp, const int64_t* idx, const float* cnt, void* stream) {
  XFM_REQUIRE(I && T && temp && lse && g && dI && dT && dtemp, "itc_bwd: null operand");
  return xfm_itc_bwd_impl(I, T, temp, lse, g, N, E, dI, dT, dtemp, idx, cnt, ST(stream));
}
int xfm_hard_negatives(const float* I, const float* T, const float* temp, int B, int E, uint64_t seed, int64_t* image_neg,
                       int64_t* text_neg, const int64_t* idx, void* stream) {
  XFM_REQUIRE(I && T && temp && image_neg && text_neg, "hard_negatives: null operand");
  return xfm_hard_negatives_impl(I, T, temp, B, E, seed, image_neg, text_neg, idx, ST(stream));
}

int xfm_ce_fwd(const float* logits, long ld, int R, int V, const int64_t* labels, float* lse, float* loss, void* stream) {
  XFM_REQUIRE(logits && labels && lse && loss, "ce_fwd: null operand");
  return xfm_ce_fwd_impl(logits, ld, R, V, labels, lse, loss, ST(stream));
}
int xfm_ce_bwd(const float* logits, long ld, int R, int V, const int64_t* labels, const float* lse, const float* scale,
               int per_row_scale, xfm_bf16* dlogits, long ldd, void* stream) {
  XFM_REQUIRE(logits && labels && lse && scale && dlogits, "ce_bwd: null operand");
  return xfm_ce_bwd_impl(logits, ld, R, V, labels, lse, scale, per_row_scale, dlogits, ldd, ST(stream));
}

int xfm_adamw(const xfm_adamw_args* a, void* stream) {
  NOTNULL(a, "adamw");
  XFM_REQUIRE(a->p && a->g && a->m && a->v && a->group, "adamw: null operand");
  return xfm_adamw_impl(*a, ST(stream));
}
int xfm_sumsq(const float* x, long n, float* out, float* workspace, void* stream) {
  XFM_REQUIRE(x && out && workspace, "sumsq: null operand");
  return xfm_sumsq_impl(x, n, out, workspace, ST(stream));
}
int xfm_dp_unique_id(void* id) { return xfm_dp_unique_id_impl(id); }
int xfm_dp_init(const void* id, int rank, int world, void** comm) { return xfm_dp_init_impl(id, rank, world, comm); }
int xfm_dp_bucket_allreduce(void* comm, void* buf, long n, int dtype, int op, void* stream) {
  return xfm_dp_bucket_allreduce_impl(comm, buf, n, dtype, op, ST(stream));
}
int xfm_dp_allgather(void* comm, const void* send, void* recv, long n_per_rank, int dtype, void* stream) {
  return xfm_dp_allgather_impl(comm, send, recv, n_per_rank, dtype, ST(stream));
}
int xfm_dp_broadcast(void* comm, void* buf, long n, int dtype, int root, void* stream) {
  return xfm_dp_broadcast_impl(comm, buf, n, dtype, root, ST(stream));
}
int xfm_dp_finalize(void* comm) { return xfm_dp_finalize_impl(comm); }

}  // extern "C"
