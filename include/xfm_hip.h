/* libxfm_hip.so -- C ABI of the MI355X-native XFM transformer forward/backward hot path.
 *
 * Every entry point returns 0 on success or a negative XFM_E_* code; the message is available from
 * xfm_last_error() (thread local).  No entry point allocates, synchronises the device or throws: the caller owns
 * all device memory (PyTorch's caching allocator in the shipped binding), passes plain device pointers, element
 * strides and a hipStream_t (as void*), and every kernel is enqueued on that stream only, so calls are re-entrant
 * per stream and hipGraph-capturable.  There is no global mutable state.
 *
 * The reference has no FFI for this path (it bottoms out in stock ATen ops); each symbol below
 * cites the Python code whose arithmetic it replaces.  bf16 tensors are row-major uint16 payloads.
 */
#ifndef XFM_HIP_H
#define XFM_HIP_H
#include <stdint.h>

#ifdef XFM_INTERNAL_BF16
typedef __bf16 xfm_bf16;
#else
typedef uint16_t xfm_bf16;
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define XFM_OK 0
#define XFM_E_ARG (-1)
#define XFM_E_LAUNCH (-2)
#define XFM_E_UNSUPPORTED (-3)

#define XFM_ABI_VERSION 9

const char* xfm_last_error(void);
int xfm_abi_version(void);

/* ---- Linear layers (torch.nn.Linear / F.linear call sites: beit2.py:131,162,64-68,229; xroberta.py:211,224-234,
 * 301,368,382,1326,1331; xfm.py:117-120,617-620) ------------------------------------------------------------------ */
enum { XFM_EPI_BF16 = 0, XFM_EPI_F32 = 1, XFM_EPI_GELU = 2, XFM_EPI_DGELU = 3, XFM_EPI_F32_ACC = 4 };

/* C[M,N] = A[M,K] . B[N,K]^T + bias.  A, B bf16 (K contiguous).  Epilogues: bf16 out | fp32 out |
 * C = gelu(x), aux = gelu'(x) (bf16; x = the bf16-rounded pre-activation)  | C = acc * aux (the GELU dgrad, aux from the
 * forward) | fp32 C += acc.  K % 64 == 0; lda, ldb % 8 == 0.
 * tile_hint: 0 = auto, 1 = 128x128, 2 = 64x128, 3 = 64x64, 4 = 256x128 with a 3-slot LDS ring, 5 = 256x256 phase pipeline
 * (large M), 7 = 64x128 with 3 LDS stages, 8 = 64x64 with 4 (few workgroups, long K). */
int xfm_gemm_nt(const xfm_bf16* A, long lda, const xfm_bf16* B, long ldb, void* C, long ldc, const float* bias,
                xfm_bf16* aux, long ldaux, int M, int N, int K, int epilogue, int tile_hint, void* stream);

/* The launch plan xfm_gemm_nt follows for a shape (profilers / benchmarks attribute a call to its kernels with it): *cfg = tile
 * configuration (the tile_hint numbering), *rows_a > 0 = tail split: the leading rows_a rows run as whole rounds of 256x256 tiles
 * (configuration 5), the remaining rows are planned again with tile_hint -1. */
int xfm_gemm_nt_plan(int M, int N, int K, int epilogue, int tile_hint, int* cfg, int* rows_a);

/* out[M,N] = A[M,K] . B[N,K]^T (+ bias) for a very long K against few output tiles -- the activation gradient of the vocabulary
 * projection (RobertaLMHead.decoder, xroberta.py:1325-1333; BertLMPredictionHead.decoder, xbert.py:680-697: K = the padded vocabulary).
 * K is sliced over the grid; the slices' fp32 partial planes go to `workspace` (xfm_gemm_nt_ksplit_workspace bytes; 0 = the shape
 * runs unsliced and needs none) and are summed in slice order by a second kernel that rounds once to the output type
 * (out_bf16 != 0: bf16, else fp32): bit-reproducible, unlike the fp32-atomic merge of XFM_EPI_F32_ACC.  Sliced shapes need
 * N % 8 == 0 and ldo % 8 == 0. */
long xfm_gemm_nt_ksplit_workspace(int M, int N, int K);
int xfm_gemm_nt_ksplit(const xfm_bf16* A, long lda, const xfm_bf16* B, long ldb, void* out, long ldo, int out_bf16, const float* bias,
                       int M, int N, int K, float* workspace, long workspace_bytes, void* stream);

/* dW[N,K] (fp32) += dY[M,N]^T . X[M,K]   (weight gradient, split over M).
 * dbias (optional, fp32 [N]) += column sums of dY: the bias gradient rides along in the same pass over dY.
 * The split partials are written to `workspace` (xfm_gemm_tn_workspace bytes for splits_hint 0; may be 0) and summed
 * in a fixed order by a reduce kernel; a single split updates dW in place; only splits without a workspace fall back to
 * fp32 atomics.  Large edge-free shapes (M % 64 == 0; N, K % 256 == 0) run on a 256x256 pipelined kernel.
 * splits_hint: 0 = auto, > 0 = that many splits of the 128x128 kernel. */
long xfm_gemm_tn_workspace(int M, int N, int K);
int xfm_gemm_tn(const xfm_bf16* dY, long ldy, const xfm_bf16* X, long ldx, float* dW, long ldw, float* dbias, int M, int N,
                int K, int splits_hint, float* workspace, long workspace_bytes, void* stream);
/* nb (1..4) weight gradients of ONE shape (same M, N, K and leading dimensions; host arrays of device pointers; dbias / its entries may be
 * NULL) as one launch + one reduce -- the three 768 x 768 projections of a RobertaLayer with cross-attention (xroberta.py:201-304).  Falls
 * back to nb xfm_gemm_tn calls when N or K is not a multiple of 128 or the workspace is smaller than xfm_gemm_tn_batch_workspace(). */
long xfm_gemm_tn_batch_workspace(int nb, int M, int N, int K);
int xfm_gemm_tn_batch(int nb, const void* const* dY, long ldy, const void* const* X, long ldx, float* const* dW, long ldw,
                      float* const* dbias, int M, int N, int K, float* workspace, long workspace_bytes, void* stream);

/* Any number of weight gradients over the SAME M rows in persistent grouped launches (the deferred weight gradients of a whole tower:
 * every wgrad of one ViT block has 9-36 output tiles of 256 x 256 and needs 7 M-splits, partial planes and a reduce to fill the chip; the
 * tiles of all blocks together are walked whole, one owner per dW element, and only the last total % CUs tiles are cut stream-K
 * style and fixed up in workgroup order -- deterministic).  `items` is HOST memory.  Any M from 1024 rows (a short last K-step is
 * staged as zeros); problems with N or K not a multiple of 256 (or fewer rows) run through xfm_gemm_tn. */
typedef struct {
  const xfm_bf16* dY; long ldy;   /* [M, N] */
  const xfm_bf16* X; long ldx;    /* [M, K] */
  float* dW; long ldw;            /* [N, K] += */
  float* dbias;                   /* optional [N] += column sums of dY */
  int N, K;
} xfm_tn_item;
long xfm_gemm_tn_group_workspace(int n, const xfm_tn_item* items, int M);
int xfm_gemm_tn_group(int n, const xfm_tn_item* items, int M, float* workspace, long workspace_bytes, void* stream);

/* fp32 master weight [N,K] -> bf16 copy wb[N,ldb] and/or transposed bf16 copy wt[K,ldt] (zero padded). */
int xfm_cast_transpose(const float* w, int N, int K, xfm_bf16* wb, long ldb, xfm_bf16* wt, long ldt, void* stream);

/* The same for a table of weights in one launch.  `items` is DEVICE memory, sorted by tile_start; item i covers the 64x64
 * tiles [tile_start, tile_start + tiles_x * tiles_y) with tiles_x = ceil(max(K, ldb if wb) / 64), tiles_y = ceil(max(N, ldt
 * if wt) / 64) (the padded extents, so the zero padding is rewritten too); total_tiles = the sum. */
typedef struct {
  const float* w;
  xfm_bf16* wb;
  xfm_bf16* wt;
  long ldb, ldt;
  int N, K;
  int tiles_x, reserved;
  long tile_start;
} xfm_cast_item;
int xfm_cast_transpose_batch(const xfm_cast_item* items, int n_items, long total_tiles, void* stream);

/* out[n] += sum_m Y[m,n]  (bias gradients).  workspace >= xfm_colsum_workspace(M,N) bytes. */
long xfm_colsum_workspace(int M, int N);
int xfm_colsum(const xfm_bf16* Y, long ldy, int M, int N, float* out, float* workspace, long workspace_bytes,
               void* stream);

/* ---- LayerNorm family (nn.LayerNorm call sites: beit2.py:191-206,460; xroberta.py:300-304,381-385,1329;
 * xfm.py:118).  Width D in {256,512,768,1024,1536}. ---------------------------------------------------------------- */
enum { XFM_LN_PLAIN = 0, XFM_LN_POST = 1, XFM_LN_LS = 2 };

typedef struct {
  const float* x32;        /* PLAIN: fp32 input | LS: residual stream in */
  const xfm_bf16* x16;     /* PLAIN: bf16 input */
  const xfm_bf16* h;       /* POST / LS: output of the producing GEMM (bias included) */
  const xfm_bf16* res;     /* POST: residual */
  const float* ls_gamma;   /* LS: layer-scale gamma_1 / gamma_2 */
  const float* row_scale;  /* LS: per-sample drop-path scale [rows / rows_per_sample] or NULL */
  const float* w; const float* b;
  float* x_out;            /* LS: residual stream out (may alias x32) */
  xfm_bf16* z_out;         /* POST: pre-norm sum saved for backward */
  xfm_bf16* y;             /* normalised output */
  float* y32;              /* optional fp32 copy of y */
  float* mean; float* rstd;
  int rows, rows_per_sample;
  float eps;
  uint32_t drop_thresh; float drop_scale; uint32_t seed_lo, seed_hi;  /* POST: dropout on h (thresh = p * 2^32) */
  int gelu;                /* y = GELU(LN(..)): the Linear -> LayerNorm -> GELU heads (xfm.py:115-121, model_classification.py:33-48) */
  const float* res32;      /* POST: the residual in fp32 (replaces `res`): the un-rounded output of the previous LayerNorm -- the reference's
                              autocast keeps LayerNorm outputs and the residual sum in fp32 (xroberta.py:300-304, 381-385) */
  float* z32_out;          /* POST: the pre-norm sum in fp32 for the backward (z_out may then be NULL) */
} xfm_ln_fwd_args;

typedef struct {
  const xfm_bf16* dy1; const xfm_bf16* dy2; const float* dy32;  /* gradients w.r.t. y, summed on load (NULL ok) */
  const float* x32; const xfm_bf16* x16;                        /* the tensor that was normalised */
  const float* mean; const float* rstd; const float* w;
  float* dx32; xfm_bf16* dx16; int dx_accum;                     /* PLAIN: gradient w.r.t. the input */
  xfm_bf16* dh; xfm_bf16* dres;                                  /* POST / LS: to the producing GEMM / residual */
  float* dstream;                                                /* LS: in/out fp32 residual-stream gradient */
  const xfm_bf16* h; const float* ls_gamma; const float* row_scale;
  float* partial;                                                /* set by the library (workspace) */
  int rows, rows_per_sample;
  uint32_t drop_thresh; float drop_scale; uint32_t seed_lo, seed_hi;
  const float* gelu_b;     /* non-NULL: the forward applied GELU after the affine; this is the LayerNorm bias (the pre-activation
                              xhat * w + b is rebuilt and dy is multiplied by gelu'(.) first) */
  struct xfm_reduce_item_s* defer;  /* non-NULL: do NOT launch the column-sum reduce; describe it here instead.  The caller keeps `workspace`
                              untouched until it has run the item through xfm_reduce_sets_batch (one launch for a whole tower's LayerNorms
                              instead of one 7-us kernel behind each of them on the activation-gradient chain) */
  float* dres32;           /* POST: the gradient of the residual branch in fp32 (the next LayerNorm backward up the stream takes it as
                              dy32); `dres` may then be NULL */
} xfm_ln_bwd_args;

/* out[s][c] += sum over the nblocks rows of partial[s][.][c], s < nset: the deferred second half of xfm_layernorm_bwd. */
typedef struct xfm_reduce_item_s {
  const float* partial;    /* [nset][nblocks][D]; scratch the reduce may overwrite (an item split over row groups parks the groups' sums in
                              it and adds them in group order: no float atomics between the groups, bit-reproducible sums) */
  float* out[4];           /* NULL = skip the set */
  int nblocks, D, nset, reserved;
} xfm_reduce_item;
/* n items (HOST array) in launches of up to 56; the same sums, in the same order, as the per-call reduce. */
int xfm_reduce_sets_batch(int n, const xfm_reduce_item* items, void* stream);

int xfm_layernorm_fwd(const xfm_ln_fwd_args* a, int D, int mode, void* stream);
long xfm_layernorm_bwd_workspace(int rows, int D, int mode);
/* dgamma/dbeta (LN affine), dbias (column sums of dh: bias grad of the producing Linear), dls (layer-scale grad):
 * fp32 [D], accumulated (+=); NULL skips. */
int xfm_layernorm_bwd(const xfm_ln_bwd_args* a, int D, int mode, float* dgamma, float* dbeta, float* dbias, float* dls,
                      float* workspace, long workspace_bytes, void* stream);

/* ---- Attention, head_dim 64 (beit2.py:126-166; xroberta.py:201-289; causal mask xroberta.py:772-792) ------------- */
typedef struct {
  const xfm_bf16* q; long q_rs;   /* element (b,s,h,d) at ptr[(b*S + s)*rs + h*64 + d] */
  const xfm_bf16* k; long k_rs;
  const xfm_bf16* v; long v_rs;
  xfm_bf16* o; long o_rs;
  float* lse;                     /* [B,H,stat_ld] log-sum-exp of the final scores */
  const float* bias; long bias_ld;/* dense additive bias [H,Sq,bias_ld] or NULL */
  const int* key_keep;            /* [B,Sk] 1 attend / 0 padded (adds -10000) or NULL */
  int B, H, Sq, Sk;
  float scale; int causal;
  uint32_t drop_thresh; float drop_scale; uint32_t seed_lo, seed_hi;
  const xfm_bf16* dout; long do_rs;
  xfm_bf16* dq; long dq_rs; xfm_bf16* dk; long dk_rs; xfm_bf16* dv; long dv_rs;
  float* delta;                   /* [B,H,stat_ld] scratch */
  float* dbias;                   /* [H,Sq,bias_ld] fp32, += over the batch, or NULL */
  long stat_ld;                   /* row stride of lse / delta ([B,H,stat_ld]): multiple of 4, >= Sq */
  const float* bias_t; long bias_t_ld; /* optional transposed copy of bias [H,Sk,bias_t_ld] (vector loads in dK/dV) */
  const int* kv_index;            /* optional [B]: query batch row b reads keys/values (and key_keep) of source kv_index[b];
                                     dk/dv stay per query row (fold them with xfm_rows_index_sum) */
  /* optional GROUPED mode (Sq <= 64, no bias / causal / kv_index; Sk > 256 streams the keys through LDS): group g = query batch rows
     grp_rows[grp_start[g] .. grp_start[g+1]) and reads key/value source g (k, v, key_keep have n_groups batch entries).
     One workgroup per (source, head) keeps K/V LDS-resident for all of its rows; dk/dv are summed over the group and
     written per SOURCE ([n_groups*Sk] rows). */
  const int* grp_start; const int* grp_rows; int n_groups;
  /* optional PACKED (unpadded) token rows: batch entry b's queries are rows q_start[b] .. q_start[b] + q_len[b] (q_len[b] in
     [1, Sq]) of q / o / dout / dq, its keys rows k_start[b] .. + k_len[b] of k / v / dk / dv; NULL = dense b * S rows.  Sq / Sk
     stay the padded lengths (grids, statistics [B,H,stat_ld], dropout counters).  Keys past k_len get probability exactly 0,
     what the reference's additive -10000 mask yields in fp32 for a prefix-masked batch (xroberta.py:751-807), so no key_keep
     is needed.  Rows of o / dq / dk / dv outside every sequence are not written.  No bias, no kv_index in this mode; in grouped
     mode only the query side may be packed. */
  const int* q_start; const int* q_len; const int* k_start; const int* k_len;
  /* xfm_attn_bwd only: 0 = both kernels; 1 = the dQ kernel alone (writes dq and the row statistics `delta`); 2 = the dK/dV kernel
     alone (reads the `delta` a phase-1 call left) -- lets a caller put dK/dV, which in cross-attention only feed weight gradients,
     on another stream than the activation-gradient chain. */
  int bwd_phase;
  /* optional second half of the output, o_lo = bf16(O - bf16(O)) (same addressing as o): the forward writes it when non-NULL; the
     backward then takes the softmax-gradient row term delta_i = sum_j P_ij dP_ij = dO_i . O_i from dO . (o + o_lo) in its prologue
     (error ~2^-17 |dO||O| plus the forward's own bf16 rounding of P, below the bf16 noise of dS) instead of recomputing it in a
     first pass over the keys -- two of the dQ kernel's five matrix products.  NULL = the exact two-pass form. */
  xfm_bf16* o_lo;
  /* optional copies of `bias` in the MFMA accumulator layout (xfm_bias_tile), used by the batch-walking kernels of the 224-px ViT
     shape (Sq == Sk <= 224, no mask / dropout): bias_tiled for the forward, bias_t_tiled (tiles of the transposed bias) for the
     backward.  One bias tile is then one contiguous 1-KB wave load instead of 16 strided row segments.  NULL = read `bias` /
     `bias_t`. */
  const float* bias_tiled; const float* bias_t_tiled;
  /* xfm_attn_bwd with dbias on a problem too long for the in-register bias-gradient kernels (Sk > 256: the 577 / 901 tokens of the
     384 / 480 px ViT, beit2.py:126-166): optional fp32 workspace [B,H,Sq,bias_ld].  The dQ kernel then STORES every batch entry's dS
     there (coalesced 16-byte stores) and a second kernel adds the sum over the batch to dbias -- instead of one float atomic per
     score and batch entry (234 M atomics per layer at B = 24, 901 tokens: 3.1 ms; with the workspace 0.6 ms).  NULL = atomics. */
  float* dbias_ws;
} xfm_attn_args;

int xfm_attn_fwd(const xfm_attn_args* a, void* stream);
int xfm_attn_bwd(const xfm_attn_args* a, void* stream);
/* bytes of `dbias_ws` an xfm_attn_bwd call with these arguments wants (0: none).  Long dense unmasked problems with a bias gradient
 * (Sk > 256: the 577 / 901 tokens of the 384 / 480 px ViT) sum dS over the batch inside a kernel that walks the batch entries of a
 * slice with a 128 x 128 block of the gradient in registers; with more than one slice the per-slice sums are [slices, H, Sq, bias_ld]
 * planes in this workspace (a few MB) that a second kernel folds into dbias in slice order.  The general kernels (masked / dropped
 * problems) still take the [B, H, Sq, bias_ld] form described at xfm_attn_args.dbias_ws. */
long xfm_attn_bwd_workspace(const xfm_attn_args* a);
/* dense additive bias [H,S,ld] (fp32) -> two tiled copies, each [H][T][T][64 lanes][4] floats with T = ceil(S / 16), pre-divided by
 * `scale` (the kernels start the score accumulators from bias / scale):
 *   tiled  [h][a][b][lane][r] = bias[h][16a + (lane & 15)][16b + 4 (lane >> 4) + r] / scale        (query on the lane: forward)
 *   tiled_t[h][a][b][lane][r] = bias[h][16b + 4 (lane >> 4) + r][16a + (lane & 15)] / scale        (key on the lane:   backward)
 * Entries whose key is past S hold -1e30 (probability 0), entries whose query is past S hold 0.  tiled_t has T + 1 key-tile rows
 * ([H][T+1][T][64][4]): the last one is all -1e30, for a kernel wave whose second key tile lies past the sequence.  Either output may be
 * NULL. */
int xfm_bias_tile(const float* bias, int H, int S, long ld, float scale, float* tiled, float* tiled_t, void* stream);
/* dense[h,i,j] = table[index[i*N+j]*H + h] (beit2.py:139-145), rows padded to ld (dense_t: optional [h,j,i] copy);
 * and its scatter-add gradient. */
/* dst[u,:] = sum_{r: index[r]==u} src[r,:]  (bf16 rows of `len` elements, fp32 accumulation). */
int xfm_rows_index_sum(const xfm_bf16* src, const int* index, int R, int U, long len, xfm_bf16* dst, void* stream);
int xfm_relpos_gather(const float* table, const int* index, int H, int N, long ld, float* dense, float* dense_t,
                      void* stream);
int xfm_relpos_scatter(const float* ddense, const int* index, int H, int N, long ld, float* dtable, void* stream);
/* the same gradient for the standard index of a G x G patch grid + cls token (N = G*G + 1 tokens, (2G-1)^2 + 3 table rows,
 * beit2.py:92-116): coalesced reads along the grid's structure, one owner per table entry (no atomics, no sort).  dtable += . */
int xfm_relpos_grid_grad(const float* ddense, int H, int G, long ld, float* dtable, void* stream);
/* the same gradient without atomics: order = positions i*N+j sorted by index[i*N+j], start = [entries+1] offsets into order */
int xfm_relpos_scatter_sorted(const float* ddense, const int* order, const int* start, int entries, int H, int N, long ld,
                              float* dtable, void* stream);

/* ---- Patch gather for the patch-embed GEMM (beit2.py:224-230) --------------------------------------------------- */
int xfm_patchify(const float* image, int B, int C, int H, int W, int P, xfm_bf16* out, void* stream);

/* ---- ViT token assembly (beit2.py:432-446: mask-token mix `x * (1 - w) + mask_token * w`, then cat(cls, x)) ----------------
 * x0[b] = [cls | tok[b mod Bt], masked patches (mask[b, i] != 0) replaced by mask_token], fp32 [Bx, P + 1, D]; Bx a multiple
 * of Bt (several masked views of the same patch-embedded images), mask [Bx, P] bytes or NULL.
 * bwd: dtok [Bt, P, D] is written (sum over the views that kept the patch); dcls [D] and dmask_token [D] are accumulated. */
int xfm_vit_tokens_fwd(const float* tok, const float* cls, const float* mask_token, const uint8_t* mask, int Bt, int Bx, int P,
                       int D, float* x0, void* stream);
int xfm_vit_tokens_bwd(const float* dx0, const uint8_t* mask, int Bt, int Bx, int P, int D, float* dtok, float* dcls,
                       float* dmask_token, void* stream);

/* ---- BEiT pooled-cls tail (beit2.py:455-466: drop cls, mean over the patch tokens, cat([mean, patches])) --------------------------
 * fwd, in place: y[b, 0, :] = mean_i y[b, 1 + i, :] (bf16 [B, N, D], fp32 mean); bwd: out[b, 0] = 0, out[b, 1 + i] = dy[b, 1 + i] + dy[b, 0] / (N - 1). */
int xfm_pool_rows_fwd(xfm_bf16* y, int B, int N, int D, void* stream);
int xfm_pool_rows_bwd(const xfm_bf16* dy, int B, int N, int D, xfm_bf16* out, void* stream);

/* ---- MIM loss (xfm.py:624-635): x = embeddings of the masked view, t = of the clean view (detached), bf16 [B, N, D]; mask [B, N-1]
 * bytes.  fwd: sums[3] += {sum (x-t)^2 over masked patch rows, the same over the cls rows, number of masked patches} (caller zeroes);
 * loss = sums[0] / max(sums[2] * D, 1) + sums[1] / (B * D).  bwd: dx (bf16 [B, N, D], fully written) = gout[0] * d loss / d x;
 * cls_term = 0 drops the cls part (the reference's `mim_cls_only` flag returns the patch term alone). */
#define XFM_MIM_PARTIALS 512
#define XFM_MIM_SUMS_FLOATS (3 + 3 * XFM_MIM_PARTIALS)
int xfm_mim_loss_fwd(const xfm_bf16* x, const xfm_bf16* t, const uint8_t* mask, int B, int N, int D, float* sums, void* stream);
int xfm_mim_loss_bwd(const xfm_bf16* x, const xfm_bf16* t, const uint8_t* mask, const float* sums, const float* gout, int cls_term,
                     int B, int N, int D, xfm_bf16* dx, void* stream);

/* ---- Block-wise MIM mask sampler (masking_generator.py:27-105, called once per image and step on the host at beit2.py:432-439):
 * out[b, p] (bytes, [B, GH*GW]) = 1 on exactly `num` patches per image, drawn by the reference's rejection loop (rectangles of area
 * U[min_num, remaining], log-uniform aspect in [min_aspect, max_aspect], <= 10 attempts per block, uniform top-up), one wavefront per
 * image, counter-based generator keyed by (seed, image).  delta_hist (optional, int32 [GH*GW + 1], +=): how many new patches each
 * accepted block added. */
int xfm_mim_masks(int B, int GH, int GW, int num, int min_num, float min_aspect, float max_aspect, uint64_t seed, uint8_t* out,
                  int* delta_hist, void* stream);

/* ---- RoBERTa embeddings + LayerNorm + dropout (xroberta.py:104-137, 1747-1757) ----------------------------------- */
typedef struct {
  const int64_t* ids;
  const float* word; const float* pos; const float* type;
  const float* w; const float* b;
  xfm_bf16* y; float* mean; float* rstd; int* pos_ids;
  int B, T, pad_id; float eps;
  uint32_t drop_thresh; float drop_scale; uint32_t seed_lo, seed_hi;
  const xfm_bf16* dy; float* dword; float* dpos; float* partial;
  int pos_mode; /* 0: RoBERTa pad-aware cumsum positions; 1: BERT absolute positions 0..T-1 (xbert.py:198-199) */
  const int* row_map; /* optional [B*T]: token (b,t) is written to / its gradient read from row row_map[b*T+t] of y / dy (packed,
                         unpadded rows); -1 skips the token (padding).  mean / rstd / pos_ids stay [B*T]. */
  float* y32;         /* optional fp32 twin of y (same rows): the embedding output enters the encoder's fp32 residual stream un-rounded */
  const float* dy32;  /* optional fp32 gradient added to dy (the residual-branch gradient of the first layer) */
  float* dz_out;      /* optional fp32 [B*T, D] (bwd): the gradient w.r.t. the embedding sum of every token is STORED here (row b*T+t;
                         skipped tokens are not written) and dword / dpos are left alone -- the caller scatters it with
                         xfm_rows_segment_sum in a fixed order instead of one float atomic per element (XFM_DETERMINISTIC) */
} xfm_embed_args;
int xfm_embed_ln_fwd(const xfm_embed_args* a, int D, void* stream);
long xfm_embed_ln_bwd_workspace(int rows, int D);
int xfm_embed_ln_bwd(const xfm_embed_args* a, int D, float* dgamma, float* dbeta, float* dtype, float* workspace,
                     long workspace_bytes, void* stream);

/* ---- Row gather / scatter-add for packed token rows (the [CLS] / masked-position gathers of xroberta.py:1215-1216 and the
 * batch assembly of xfm.py:781-793 on unpadded rows): dst[r,:] = index[r] >= 0 ? src[index[r],:] : 0 (bf16 rows of D elements,
 * D % 8 == 0); and its adjoint dst32[index[r],:] += src[r,:] (fp32 accumulation, rows with index < 0 are skipped). ------------- */
int xfm_rows_gather(const xfm_bf16* src, const int* index, int R, int D, xfm_bf16* dst, void* stream);
int xfm_rows_scatter_add(const xfm_bf16* src, const int* index, int R, int D, float* dst32, void* stream);
/* out[key, :] += sum of src[perm[i], :] over the run of positions i with sorted_key[i] == key, rows added in position order, one owner per
 * run: the scatter-add of an embedding gradient without atomics (nn.Embedding backward, xroberta.py:80-102).  sorted_key int64 [R]
 * ascending (the caller sorts; perm int64 [R] = the matching row order, a STABLE sort keeps it deterministic); runs with key < 0 or
 * key == skip_key are dropped (padding_idx, skipped tokens).  src fp32 [*, D], out fp32 [*, D], D % 4 == 0. */
int xfm_rows_segment_sum(const float* src, const int64_t* perm, const int64_t* sorted_key, long R, int D, long skip_key, float* out,
                         void* stream);

/* ---- One whole RobertaLayer per call (xroberta.py:405-473: self-attention block, optional cross-attention block, FFN block, each
 * closed by dropout + residual + LayerNorm), forward and backward.  The kernels are the ones above; what this adds is the launch
 * SEQUENCE on the native side: a host language pays microseconds per launch (Python: 7-24 us per wrapper call against ~3.5 us for
 * the launch itself), and a text / fusion tower is ~50 launches per layer of 10-30 us kernels -- its time is the host's, not the
 * GPU's, unless the layer is one call.  All buffers are the caller's: `slab` holds the layer's activations (what the backward
 * needs), `bslab` the backward's gradient temporaries, both laid out by xfm_rlayer_layout.  The weight-gradient GEMMs and the
 * cross-attention dK/dV kernel are issued on `side_stream` behind events (NULL: everything on the main stream). ------------------ */
typedef struct {        /* static per layer: bf16 operand copies ([N,K] and transposed [K,ld]), fp32 biases / LayerNorm affine, and
                           where their gradients accumulate (fp32, +=) */
  const xfm_bf16 *wqkv, *wqkv_t, *wo, *wo_t, *wq2, *wq2_t, *wkv2_t, *wo2, *wo2_t, *wi, *wi_t, *wout, *wout_t;
  long ld_wqkv_t, ld_wo_t, ld_wq2_t, ld_wkv2_t, ld_wo2_t, ld_wi_t, ld_wout_t;
  const float *bqkv, *bo, *bq2, *bo2, *bi, *bout;
  const float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *ln3_w, *ln3_b;
  float *dwqkv, *dbqkv, *dwo, *dbo, *dwq2, *dbq2, *dwkv2, *dbkv2, *dwo2, *dbo2, *dwi, *dbi, *dwout, *dbout;
  float *dln1_w, *dln1_b, *dln2_w, *dln2_b, *dln3_w, *dln3_b;
  int D, H, FF, has_cross;
  float eps;
} xfm_rlayer_params;

typedef struct {        /* geometry + per-call inputs shared by forward and backward */
  int R, B, T;                                   /* token rows, sequences, padded sequence length */
  int R_alloc, B_alloc;                          /* geometry slab / bslab were laid out for (0: R, B); a backward over the first
                                                    sequences of a forward pass (their rows are a prefix) passes the forward's */
  int Nenc, U;                                   /* cross-attention: image tokens per image, images (0 = no cross-attention input) */
  const int* seq_start; const int* seq_len;      /* packed rows (xfm_attn_args.q_start / q_len) or NULL */
  const int* key_keep;                           /* [B,T] or NULL */
  const int* enc_keep;                           /* [U,Nenc] or NULL */
  const int* grp_start; const int* grp_rows;     /* grouped cross-attention: rows of each image listed sequence by sequence, or ... */
  const int* xq_start; const int* xq_len; int xq_max;  /* ... RANGE mode: the sequences are laid out image by image, so the queries of
                                                    image u are the contiguous rows xq_start[u] .. + xq_len[u] (<= xq_max): cross-attention
                                                    runs as one ragged problem per image on full 16-row tiles (no per-sequence tiles
                                                    that are mostly padding), dK/dV come out per image */
  int causal, zero_fill;                         /* zero_fill: attention outputs / gradients of rows outside every sequence */
  float scale;
  uint32_t att_thresh; float att_scale; uint32_t hid_thresh; float hid_scale;   /* dropout p as threshold / 1/(1-p); 0 = off */
  uint32_t seed_hi, seed_ctr;                    /* dropout stream k of the layer uses (seed_hi, seed_ctr + 1 + k) */
  const xfm_bf16* x;                             /* layer input [R,D] */
  void* slab;                                    /* forward activations, xfm_rlayer_layout(...).fwd_bytes */
  const xfm_bf16* kv; long kv_ld;                /* cross: this layer's K|V projection of the image states [U*Nenc, >= 2D] */
  void* kv_event;                                /* hipEvent_t the main stream waits for before it reads kv (NULL: none) */
  int f32_stream;                                /* 1: fp32 residual stream inside the layer (layout flag XFM_RL_F32_STREAM): every post-LN sum
                                                    z_k = dropout(h_k) + y_{k-1} uses the UN-ROUNDED fp32 output of the previous LayerNorm and is
                                                    kept in fp32 for the backward, whose residual-branch gradients are fp32 too -- what the
                                                    reference's mixed precision does (LayerNorm and the residual add run in fp32) */
  const float* x32;                              /* f32_stream: fp32 twin of x (the previous layer's y3_32) or NULL (x enters the stream as is) */
} xfm_rlayer_io;

typedef struct {        /* backward-only */
  void* bslab;                                   /* gradient temporaries, xfm_rlayer_layout(...).bwd_bytes */
  const xfm_bf16* dy_a; const xfm_bf16* dy_b;    /* gradient w.r.t. the layer output = dy_a + dy_b (dy_b may be NULL) */
  const xfm_bf16* enc;                           /* image states [U*Nenc, D] (K/V weight gradient) */
  xfm_bf16* dkv; long dkv_ld;                    /* out: dK|dV of this layer [U*Nenc, 2D] (possibly a column block) */
  float* denc32;                                 /* optional fp32 [U*Nenc, D] += dkv @ Wkv (NULL: the caller folds dkv itself) */
  int need_dprev;                                /* produce the gradient w.r.t. the layer input (dprev_a + dprev_b in bslab) */
  void* side_stream;
  float* ws_main; long ws_main_bytes; float* ws_side; long ws_side_bytes;
  float* ln_ws; long ln_ws_stride;               /* non-NULL (with ln_items): the layer's three LayerNorm backward kernels write their column-sum
                                                    partials to ln_ws + k * ln_ws_stride (floats, k = 0..2; each >= xfm_layernorm_bwd_workspace bytes)
                                                    and launch no reduce; ... */
  xfm_reduce_item* ln_items; int* ln_count;      /* ... they append their items to this HOST table instead (ln_items[*ln_count], count advanced) */
  const float* dy_b32;                           /* f32_stream: the fp32 part of the gradient w.r.t. the layer output (the next layer's dres1) or
                                                    NULL; dy_b must be NULL.  The gradient w.r.t. the layer input is dprev (bf16) + dres1 (fp32) */
  int defer_wgrad;                               /* 1: launch NO weight-gradient GEMM -- the caller queues them (the dY / X operands sit at the
                                                    layout's offsets in bslab / slab, which it keeps alive) and runs the queue of the whole
                                                    tower as grouped launches (xfm_gemm_tn_group).  Bias gradients that ride on a weight
                                                    gradient (dbi, dbq2, dbkv2, dbqkv) are deferred with it */
} xfm_rlayer_bwd_args;

typedef struct {        /* byte offsets inside slab / bslab (256-byte aligned) */
  long qkv, c1, lse1, h, z1, m1, r1, y1, q2, c2, lse2, z2, m2, r2, y2, hact, u, z3, m3, r3, y3, fwd_bytes;
  long dh3, dres3, du, d1a, dh2, dres2, dc2, dq2, delta2, d2a, dh1, dres1, dc1, dqkv, delta1, dprev, bwd_bytes;
  long ws_main_bytes, ws_side_bytes;             /* workspace the backward wants on each stream */
  long c2lo;                                     /* forward slab: low half of the cross-attention output (O = c2 + c2lo to fp32-ish precision) */
  long y1_32, y2_32, y3_32;                      /* XFM_RL_F32_STREAM: fp32 twins of y1 / y2 / y3 (-1 otherwise); z1-3 and dres1-3 are fp32 then */
} xfm_rlayer_layout_t;

enum { XFM_RL_DROPOUT = 1, XFM_RL_F32_STREAM = 2 };   /* `flags` of xfm_rlayer_layout */
int xfm_rlayer_layout(int R, int B, int T, int D, int H, int FF, int has_cross, int Nenc, int U, int xq_max, int flags,
                      xfm_rlayer_layout_t* out);
int xfm_rlayer_fwd(const xfm_rlayer_params* p, const xfm_rlayer_io* io, void* stream);
int xfm_rlayer_bwd(const xfm_rlayer_params* p, const xfm_rlayer_io* io, const xfm_rlayer_bwd_args* b, void* stream);

/* ---- Contrastive / matching glue of XFMBase (xfm.py:614-621, 683-746) as a handful of small kernels instead of a few dozen ATen
 * launches (matmul through the vendor BLAS, softmax, nll_loss, fills, multinomial):
 *   rownorm      y = x / max(|x|_2, 1e-12) per row (F.normalize, xfm.py:617-620), fp32 [R, E], E % 64 == 0, E <= 1024; inv = 1 / norm
 *   itc          logits = I . T^T / temp over N gathered rows; loss = (CE(logits, arange) + CE(logits^T, arange)) / 2 (xfm.py:699-703);
 *                fwd: lse [4N] (statistics of the rows of logits, then of logits^T; behind them the 2N rows' loss terms),
 *                loss_sum [2], zeroed by the caller: [0] += the loss, summed over the row terms in a fixed order by the block that takes
 *                the last ticket of the integer counter at [1] (bit-reproducible); bwd: dI, dT (fully written) and
 *                dtemp[0] += for the upstream gradient g[0]; `lse` is the forward's [4N] buffer: the backward parks the N row shares
 *                of dtemp in its dead upper half and adds them in a fixed order (bit-reproducible)
 *                idx != NULL (int64 [N], the gathered image ids of the retrieval fine-tuning step, xfm.py:705-713): soft labels -- the
 *                positives of row r are the rows with the same id, weight 1 / cnt_r each; fwd writes cnt [N], bwd reads it
 *   hard_neg     per row of the LOCAL batch: softmax(sim / temp) + 1e-5, own entry zeroed, ONE categorical draw (xfm.py:727-744:
 *                torch.multinomial(...).item() per row on the host there); text_neg[i] for image i, image_neg[j] for text j;
 *                idx != NULL (int64 [B]): every entry with the row's image id is zeroed (xfm.py:731-734)  -------- */
int xfm_rownorm_fwd(const float* x, int R, int E, float* y, float* inv, void* stream);
int xfm_rownorm_bwd(const float* dy, const float* y, const float* inv, int R, int E, float* dx, void* stream);
int xfm_itc_fwd(const float* I, const float* T, const float* temp, int N, int E, float* lse, float* loss_sum, const int64_t* idx,
                float* cnt, void* stream);
int xfm_itc_bwd(const float* I, const float* T, const float* temp, const float* lse, const float* g, int N, int E, float* dI, float* dT,
                float* dtemp, const int64_t* idx, const float* cnt, void* stream);
int xfm_hard_negatives(const float* I, const float* T, const float* temp, int B, int E, uint64_t seed, int64_t* image_neg,
                       int64_t* text_neg, const int64_t* idx, void* stream);

/* ---- Vocabulary cross-entropy, ignore_index -100 (xroberta.py:1296-1297, 1107-1114) ------------------------------ */
int xfm_ce_fwd(const float* logits, long ld, int R, int V, const int64_t* labels, float* lse, float* loss, void* stream);
/* dlogits = (softmax - onehot) * scale[0]  (per_row_scale = 0: mean / sum reductions) or * scale[row] (reduction 'none'). */
int xfm_ce_bwd(const float* logits, long ld, int R, int V, const int64_t* labels, const float* lse, const float* scale,
               int per_row_scale, xfm_bf16* dlogits, long ldd, void* stream);

/* ---- Flat-arena optimiser step (optim.py:4-50 + clip, apex_ddp_accelerator.py:100-110) --------------------------- */
typedef struct {
  float* p; float* g; float* m; float* v;
  const uint8_t* group;       /* group id per 256-element block */
  float lr[4]; float wd[4];
  float beta1, beta2, eps, bc1, bc2;
  const float* clip_coef;     /* device scalar or NULL */
  long n;
  int zero_grad;              /* != 0: g is zeroed in the same sweep (the step's optimizer.zero_grad(): one pass over the arena less) */
} xfm_adamw_args;
int xfm_adamw(const xfm_adamw_args* a, void* stream);
/* out[0] += sum x[i]^2 (n % 4 == 0), bit-reproducible: fixed-grid block partials through `workspace`
 * (XFM_SUMSQ_WORKSPACE_FLOATS floats, contents irrelevant) summed in a fixed order -- the clip coefficient derived from it must be
 * identical on every data-parallel rank (apex_ddp_accelerator.py:100-110 clips after the all-reduce). */
#define XFM_SUMSQ_WORKSPACE_FLOATS 1024
int xfm_sumsq(const float* x, long n, float* out, float* workspace, void* stream);

/* ---- Data-parallel exchange over RCCL / xGMI ------------------------------------------------------------------------
 * What the reference does through torch.distributed: the gradient all-reduce of DistributedDataParallel
 * (accelerators/ddp_accelerator.py:34-98, apex_ddp_accelerator.py:34-110), the feature AllGather of the contrastive loss
 * (models/xfm.py:17-50), the parameter broadcast at wrap time -- for a host without torch.distributed.  One
 * communicator per process (one process per GPU); every call is enqueued on the caller's HIP stream and returns at once; nothing is
 * allocated or synchronised.  librccl is resolved at the first call (dlopen: inside a PyTorch process that is the RCCL torch already
 * mapped); XFM_E_UNSUPPORTED when it cannot be loaded.  The Python host (accelerators/rccl_ddp_accelerator.py) keeps using
 * torch.distributed's "nccl" backend, which IS RCCL; xfm_amd/dp.py wraps these for hosts that do not.
 *   rank 0: xfm_dp_unique_id(id) -> hand the XFM_DP_ID_BYTES bytes to every rank (file, socket, environment) ->
 *   every rank, with its device current: xfm_dp_init(id, rank, world, &comm) -> ... -> xfm_dp_finalize(comm). */
#define XFM_DP_ID_BYTES 128
enum { XFM_DP_F32 = 0, XFM_DP_BF16 = 1, XFM_DP_I32 = 2 };
enum { XFM_DP_SUM = 0, XFM_DP_AVG = 1, XFM_DP_MAX = 2 };
int xfm_dp_unique_id(void* id);
int xfm_dp_init(const void* id, int rank, int world, void** comm);
/* in place over one bucket (a range of the flat gradient arena): buf[i] = op over ranks; XFM_DP_AVG is the mean DistributedDataParallel
 * takes (ReduceOp.AVG: one pass, no separate division) */
int xfm_dp_bucket_allreduce(void* comm, void* buf, long n, int dtype, int op, void* stream);
/* recv[r * n_per_rank + i] = rank r's send[i]  (models/xfm.py:17-50: the forward of AllGather; its backward is the caller's slice) */
int xfm_dp_allgather(void* comm, const void* send, void* recv, long n_per_rank, int dtype, void* stream);
int xfm_dp_broadcast(void* comm, void* buf, long n, int dtype, int root, void* stream);
int xfm_dp_finalize(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* XFM_HIP_H */
