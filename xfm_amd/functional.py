"""Raw (non-autograd) tensor-level wrappers over the C ABI.  Device tensors in, kernels enqueued on torch's current
stream.  PyTorch is used here only for memory, streams and shapes."""
import ctypes
import os

import torch

from . import _lib
from ._lib import AdamWArgs, AttnArgs, EmbedArgs, LnBwdArgs, LnFwdArgs, check

BF16, F32 = torch.bfloat16, torch.float32
EPI_BF16, EPI_F32, EPI_GELU, EPI_DGELU, EPI_F32_ACC = 0, 1, 2, 3, 4
LN_PLAIN, LN_POST, LN_LS = 0, 1, 2


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """Raw HIP handle of torch's current stream (every kernel of the library is launched on it)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _dev(t):
    if not t.is_cuda:
        raise _lib.XfmHipError("xfm_amd ops need HIP device tensors (no CPU fallback on the product path)")


_workspaces = {}


def workspace(nbytes, device):
    """Per (device, stream) scratch for the block-partial reductions; grows monotonically."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty(max(int(nbytes) // 4 + 1, 1 << 20), dtype=F32, device=device)
        _workspaces[key] = ws
    return ws


def drop_params(p, seed):
    """(thresh, scale, seed_lo, seed_hi) for dropout probability p; thresh 0 disables."""
    if p <= 0.0:
        return 0, 1.0, 0, 0
    thresh = min(int(p * 4294967296.0), 4294967295)
    return thresh, 1.0 / (1.0 - p), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF


# --------------------------------------------------------------------------------------------- GEMMs
def gemm_nt(a, b, bias=None, epi=EPI_BF16, aux=None, out=None, n=None, tile_hint=0):
    """out[M,N] = a[M,K] @ b[N(,pad),K]^T + bias.  a, b bf16 row-major (last dim contiguous)."""
    _dev(a)
    assert a.dtype == BF16 and b.dtype == BF16 and a.dim() == 2 and b.dim() == 2
    assert a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0] if n is None else n
    assert b.shape[1] == K, (a.shape, b.shape)
    if out is None:
        out = torch.empty((M, N), dtype=F32 if epi in (EPI_F32, EPI_F32_ACC) else BF16, device=a.device)
    assert out.stride(1) == 1
    if epi == EPI_GELU and aux is None:
        aux = torch.empty((M, N), dtype=BF16, device=a.device)
    check(_lib.load().xfm_gemm_nt(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0),
                                  _ptr(bias), _ptr(aux), 0 if aux is None else aux.stride(0), M, N, K, epi, tile_hint,
                                  _stream()), "gemm_nt")
    return (out, aux) if epi == EPI_GELU else out


def gemm_nt_ksplit(a, b, n=None, bias=None, out_dtype=BF16):
    """out[M,N] = a[M,K] @ b[N(,pad),K]^T (+ bias) with K sliced over the grid and the slices summed in a FIXED order (bit-reproducible;
    the LM-head activation gradient: K = the padded vocabulary).  Returns a new bf16 / fp32 tensor."""
    _dev(a)
    assert a.dtype == BF16 and b.dtype == BF16 and a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0] if n is None else n
    assert b.shape[1] == K, (a.shape, b.shape)
    lib = _lib.load()
    out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    need = lib.xfm_gemm_nt_ksplit_workspace(M, N, K)
    ws = workspace(need, a.device) if need > 0 else None
    check(lib.xfm_gemm_nt_ksplit(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), int(out_dtype == BF16),
                                 _ptr(bias), M, N, K, _ptr(ws), 0 if ws is None else ws.numel() * 4, _stream()), "gemm_nt_ksplit")
    return out


def deterministic():
    """XFM_DETERMINISTIC=1: the reductions that by default still end in float atomics because their ordered form costs a launch or a pass
    take the ordered form (read per call, like the library's own switch): the bias column sums that ride on the M-split weight-gradient
    GEMMs (here), the bias gradient of the short attention backward (per-slice planes, csrc/attention.hip), the embedding gradients
    (sorted segment sums, embed_ln_bwd).  Every other reduction of the step is ordered unconditionally (tools/bit_repro.py)."""
    return os.environ.get("XFM_DETERMINISTIC", "0") not in ("", "0")


def gemm_tn(dy, x, dw, n=None, splits=0, dbias=None):
    """dw[N,K] (fp32) += dy[M,N]^T @ x[M,K]; optionally dbias[N] (fp32) += dy.sum(0) in the same pass."""
    _dev(dy)
    assert dy.dtype == BF16 and x.dtype == BF16 and dw.dtype == F32
    assert dy.stride(1) == 1 and x.stride(1) == 1 and dw.stride(-1) == 1
    M, K = x.shape
    N = dy.shape[1] if n is None else n
    assert dy.shape[0] == M and dw.shape[0] >= N and dw.shape[1] == K, (dy.shape, x.shape, dw.shape)
    if dbias is not None and deterministic():   # the M-splits of the kernel meet in float atomics on dbias: its own ordered pass instead
        colsum(dy, dbias, N)
        dbias = None
    lib = _lib.load()
    need = lib.xfm_gemm_tn_workspace(M, N, K) if splits == 0 else (splits * ((N + 127) // 128) * ((K + 127) // 128) * 65536 if splits > 1 else 0)
    ws = workspace(need, dy.device) if need > 0 else None
    check(lib.xfm_gemm_tn(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dw.data_ptr(), dw.stride(0), _ptr(dbias), M, N, K,
                          splits, _ptr(ws), 0 if ws is None else ws.numel() * 4, _stream()), "gemm_tn")


def gemm_tn_batch(dys, xs, dws, dbiases=None):
    """dws[i][N,K] (fp32) += dys[i][M,N]^T @ xs[i][M,K] for up to four problems of one shape in one launch (+ one reduce)."""
    nb = len(dys)
    assert 1 <= nb <= 4 and len(xs) == nb and len(dws) == nb
    M, N = dys[0].shape
    K = xs[0].shape[1]
    for dy, x, dw in zip(dys, xs, dws):
        assert dy.dtype == BF16 and x.dtype == BF16 and dw.dtype == F32 and dy.shape == (M, N) and x.shape == (M, K)
        assert dy.stride(1) == 1 and x.stride(1) == 1 and dw.stride(-1) == 1 and dw.shape[0] >= N and dw.shape[1] == K
        assert dy.stride(0) == dys[0].stride(0) and x.stride(0) == xs[0].stride(0) and dw.stride(0) == dws[0].stride(0)
    if dbiases is not None and deterministic():
        for dy, db in zip(dys, dbiases):
            if db is not None:
                colsum(dy, db, N)
        dbiases = None
    lib = _lib.load()
    need = lib.xfm_gemm_tn_batch_workspace(nb, M, N, K)
    ws = workspace(need, dys[0].device) if need > 0 else None
    arr = lambda ts: (ctypes.c_void_p * nb)(*[0 if t is None else t.data_ptr() for t in ts])
    a_dy, a_x, a_dw = arr(dys), arr(xs), arr(dws)
    a_db = arr(dbiases) if dbiases is not None else None
    check(lib.xfm_gemm_tn_batch(nb, a_dy, dys[0].stride(0), a_x, xs[0].stride(0), a_dw, dws[0].stride(0), a_db, M, N, K, _ptr(ws),
                                0 if ws is None else ws.numel() * 4, _stream()), "gemm_tn_batch")


def gemm_tn_group(items):
    """items: (dy [M, N] bf16, x [M, K] bf16, dw [N, K] fp32, dbias [N] fp32 | None), all over the SAME M rows: dw += dy^T @ x (dbias +=
    column sums of dy) for all of them in persistent grouped launches -- whole 256 x 256 tiles with one owner each, no M-split planes
    (xfm_gemm_tn_group).  The caller keeps every tensor alive until the stream has run the call."""
    if not items:
        return
    M = items[0][0].shape[0]
    arr = (_lib.TnItem * len(items))()
    for a, (dy, x, dw, db) in zip(arr, items):
        assert dy.dtype == BF16 and x.dtype == BF16 and dw.dtype == F32 and dy.shape[0] == M and x.shape[0] == M
        assert dy.stride(1) == 1 and x.stride(1) == 1 and dw.stride(-1) == 1 and dw.shape[0] >= dy.shape[1] and dw.shape[1] == x.shape[1]
        assert db is None or (db.dtype == F32 and db.is_contiguous() and db.numel() >= dy.shape[1])
        a.dY, a.ldy, a.X, a.ldx, a.dW, a.ldw, a.dbias = dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dw.data_ptr(), dw.stride(0), _ptr(db)
        a.N, a.K = dy.shape[1], x.shape[1]
    lib = _lib.load()
    need = lib.xfm_gemm_tn_group_workspace(len(items), ctypes.addressof(arr), M)
    ws = workspace(need, items[0][0].device) if need > 0 else None
    check(lib.xfm_gemm_tn_group(len(items), ctypes.addressof(arr), M, _ptr(ws), 0 if ws is None else ws.numel() * 4, _stream()), "gemm_tn_group")


def cast_table(entries, device):
    """Device-resident xfm_cast_item table for cast_transpose_batch.  entries: (w fp32 [N,K], wb bf16 [N,ldb] | None,
    wt bf16 [K,ldt] | None).  Returns (table uint8 tensor, n_items, total_tiles)."""
    items = (_lib.CastItem * len(entries))()
    tiles = 0
    for i, (w, wb, wt) in enumerate(entries):
        N, K = w.shape
        assert w.dtype == F32 and w.is_contiguous()
        ldb = 0 if wb is None else wb.stride(0)
        ldt = 0 if wt is None else wt.stride(0)
        tx, ty = (max(K, ldb) + 63) // 64, (max(N, ldt) + 63) // 64
        items[i] = _lib.CastItem(w.data_ptr(), _ptr(wb), _ptr(wt), ldb, ldt, N, K, tx, 0, tiles)
        tiles += tx * ty
    table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(device)
    return table, len(entries), tiles


def cast_transpose_batch(table, n_items, total_tiles):
    """One launch refreshing every bf16 operand copy listed in `table` (cast_table)."""
    _dev(table)
    check(_lib.load().xfm_cast_transpose_batch(table.data_ptr(), n_items, total_tiles, _stream()), "cast_transpose_batch")


def cast_transpose(w, wb=None, wt=None):
    """fp32 [N,K] -> bf16 wb[N,ldb>=K] and/or wt[K,ldt>=N] (padding columns zeroed)."""
    _dev(w)
    assert w.dtype == F32 and w.is_contiguous() and w.dim() == 2
    N, K = w.shape
    check(_lib.load().xfm_cast_transpose(w.data_ptr(), N, K, _ptr(wb), 0 if wb is None else wb.stride(0), _ptr(wt),
                                         0 if wt is None else wt.stride(0), _stream()), "cast_transpose")


def colsum(y, out, n=None):
    """out[n] += sum_m y[m,n]."""
    M = y.shape[0]
    N = y.shape[1] if n is None else n
    lib = _lib.load()
    nb = lib.xfm_colsum_workspace(M, N)
    ws = workspace(nb, y.device)
    check(lib.xfm_colsum(y.data_ptr(), y.stride(0), M, N, out.data_ptr(), ws.data_ptr(), ws.numel() * 4, _stream()), "colsum")


# --------------------------------------------------------------------------------------------- LayerNorm
def _ln_out(rows, D, device):
    return (torch.empty((rows, D), dtype=BF16, device=device), torch.empty(rows, dtype=F32, device=device),
            torch.empty(rows, dtype=F32, device=device))


def ln_fwd(x, w, b, eps, y32=False, gelu=False):
    """PLAIN: x [rows, D] fp32 or bf16 -> (y bf16, mean, rstd[, y fp32]); gelu: y = GELU(LN(x))."""
    _dev(x)
    rows, D = x.shape
    assert x.is_contiguous()
    y, mean, rstd = _ln_out(rows, D, x.device)
    yf = torch.empty((rows, D), dtype=F32, device=x.device) if y32 else None
    a = LnFwdArgs(x32=_ptr(x) if x.dtype == F32 else 0, x16=_ptr(x) if x.dtype == BF16 else 0, w=w.data_ptr(), b=b.data_ptr(),
                  y=y.data_ptr(), y32=_ptr(yf), mean=mean.data_ptr(), rstd=rstd.data_ptr(), rows=rows, rows_per_sample=1, eps=eps,
                  gelu=int(gelu))
    check(_lib.load().xfm_layernorm_fwd(ctypes.byref(a), D, LN_PLAIN, _stream()), "layernorm_fwd")
    return (y, mean, rstd, yf) if y32 else (y, mean, rstd)


def ln_post_fwd(h, res, w, b, eps, drop=(0, 1.0, 0, 0), f32=False):
    """POST: z = dropout(h) + res; y = LN(z) -> (y, z, mean, rstd), all but stats bf16.
    f32 (the fp32 residual stream of the text / fusion towers): `res` may be fp32 (the previous LayerNorm's un-rounded output) or bf16
    (a tower input), z is kept in fp32 and the result is (y, z fp32, mean, rstd, y fp32)."""
    _dev(h)
    rows, D = h.shape
    assert h.is_contiguous() and res.is_contiguous() and h.dtype == BF16 and res.dtype == (F32 if f32 and res.dtype == F32 else BF16)
    y, mean, rstd = _ln_out(rows, D, h.device)
    a = LnFwdArgs(h=h.data_ptr(), w=w.data_ptr(), b=b.data_ptr(), y=y.data_ptr(),
                  mean=mean.data_ptr(), rstd=rstd.data_ptr(), rows=rows, rows_per_sample=1, eps=eps,
                  drop_thresh=drop[0], drop_scale=drop[1], seed_lo=drop[2], seed_hi=drop[3])
    if res.dtype == F32:
        a.res32 = res.data_ptr()
    else:
        a.res = res.data_ptr()
    if f32:
        z = torch.empty((rows, D), dtype=F32, device=h.device)
        y32 = torch.empty((rows, D), dtype=F32, device=h.device)
        a.z32_out, a.y32 = z.data_ptr(), y32.data_ptr()
    else:
        z = torch.empty_like(h)
        a.z_out = z.data_ptr()
    check(_lib.load().xfm_layernorm_fwd(ctypes.byref(a), D, LN_POST, _stream()), "layernorm_fwd(post)")
    return (y, z, mean, rstd, y32) if f32 else (y, z, mean, rstd)


def ln_ls_fwd(x, h, ls_gamma, row_scale, rows_per_sample, w, b, eps, x_out=None):
    """LS: x' = x + s_b * gamma * h (fp32 stream); y = LN(x') -> (x', y, mean, rstd)."""
    _dev(x)
    rows, D = x.shape
    assert x.dtype == F32 and h.dtype == BF16 and x.is_contiguous() and h.is_contiguous()
    y, mean, rstd = _ln_out(rows, D, x.device)
    if x_out is None:
        x_out = torch.empty_like(x)
    a = LnFwdArgs(x32=x.data_ptr(), h=h.data_ptr(), ls_gamma=ls_gamma.data_ptr(), row_scale=_ptr(row_scale), w=w.data_ptr(),
                  b=b.data_ptr(), x_out=x_out.data_ptr(), y=y.data_ptr(), mean=mean.data_ptr(), rstd=rstd.data_ptr(),
                  rows=rows, rows_per_sample=rows_per_sample, eps=eps)
    check(_lib.load().xfm_layernorm_fwd(ctypes.byref(a), D, LN_LS, _stream()), "layernorm_fwd(ls)")
    return x_out, y, mean, rstd


def _ln_bwd_call(a, D, mode, dgamma, dbeta, dbias, dls, device, defer=None):
    """defer = a ReduceQueue: the kernel leaves its column-sum partials in a slice of the queue's buffer and the queue folds them later, all
    LayerNorms of a tower in one launch (ReduceQueue.run / xfm_reduce_sets_batch)."""
    lib = _lib.load()
    nb = lib.xfm_layernorm_bwd_workspace(a.rows, D, mode)
    item = None
    if defer is not None:
        ws = defer.take(nb)
        item = _lib.ReduceItem()
        a.defer = ctypes.addressof(item)
    else:
        ws = workspace(nb, device)
    check(lib.xfm_layernorm_bwd(ctypes.byref(a), D, mode, _ptr(dgamma), _ptr(dbeta), _ptr(dbias), _ptr(dls), ws.data_ptr(),
                                ws.numel() * 4, _stream()), "layernorm_bwd")
    if item is not None and item.partial:   # (the wide-row form sums with atomics in the kernel: nothing was deferred)
        defer.items.append(item)


class ReduceQueue:
    """Deferred column-sum reduces of LayerNorm backward kernels (dgamma / dbeta / bias / layer-scale gradients): each kernel writes
    its per-workgroup partials to its own slice of ONE buffer, `run()` folds every slice with one xfm_reduce_sets_batch launch.  The
    folds are parameter gradients -- nothing on the activation-gradient chain waits for them -- and one 7-us kernel behind each of
    the 36 LayerNorms of a 12-layer tower is 0.3 ms of that chain."""

    persistent = True   # (a _WgradStream keeps it registered across flushes)

    def __init__(self, device, nbytes_total):
        self.buf = torch.empty(max(int(nbytes_total) // 4, 1), dtype=F32, device=device)
        self.used = 0
        self.items = []    # ReduceItems not folded yet
        self.extra = []    # overflow buffers (alive as long as the queue)

    def take(self, nbytes):
        n = (int(nbytes) // 4 + 63) // 64 * 64
        if self.used + n > self.buf.numel():   # (sized by the caller for its tower; a call beyond that gets a buffer of its own)
            self.extra.append(torch.empty(n, dtype=F32, device=self.buf.device))
            return self.extra[-1]
        out = self.buf[self.used:self.used + n]
        self.used += n
        return out

    def run(self):
        """Fold what has been queued since the last run (the buffer's slices are not reused: a fold may still be in flight)."""
        items, self.items = self.items, []
        if items:
            reduce_sets_batch(items)


def reduce_sets_batch(items):
    """items: _lib.ReduceItem structs filled by deferred LayerNorm backward calls (or by xfm_rlayer_bwd's ln_items table)."""
    n = len(items)
    arr = (_lib.ReduceItem * n)(*items)
    check(_lib.load().xfm_reduce_sets_batch(n, ctypes.addressof(arr), _stream()), "reduce_sets_batch")


def ln_bwd(dy, x, mean, rstd, w, dgamma, dbeta, dy2=None, dy32=None, dx32=None, dx16=None, dx_accum=False, gelu_b=None, defer=None):
    """PLAIN backward: writes dx32 (optionally accumulating) and/or dx16; dgamma/dbeta += .  gelu_b: the LayerNorm bias when the
    forward ran with gelu=True (dy is then the gradient of the activated output)."""
    rows, D = x.shape
    a = LnBwdArgs(dy1=dy.data_ptr(), dy2=_ptr(dy2), dy32=_ptr(dy32), x32=_ptr(x) if x.dtype == F32 else 0,
                  x16=_ptr(x) if x.dtype == BF16 else 0, mean=mean.data_ptr(), rstd=rstd.data_ptr(), w=w.data_ptr(),
                  dx32=_ptr(dx32), dx16=_ptr(dx16), dx_accum=int(dx_accum), rows=rows, rows_per_sample=1, gelu_b=_ptr(gelu_b))
    _ln_bwd_call(a, D, LN_PLAIN, dgamma, dbeta, None, None, x.device, defer)


def ln_post_bwd(dy, z, mean, rstd, w, dgamma, dbeta, dbias, dy2=None, drop=(0, 1.0, 0, 0), defer=None):
    """POST backward -> (dh, dres) bf16 (same tensor when dropout is off); dgamma/dbeta/dbias += .
    fp32 stream (z is fp32): dy2, if given, is the fp32 residual-branch gradient of the LayerNorm above; dres comes back in fp32."""
    rows, D = z.shape
    dh = torch.empty((rows, D), dtype=BF16, device=z.device)
    a = LnBwdArgs(dy1=dy.data_ptr(), mean=mean.data_ptr(), rstd=rstd.data_ptr(),
                  w=w.data_ptr(), dh=dh.data_ptr(), rows=rows, rows_per_sample=1,
                  drop_thresh=drop[0], drop_scale=drop[1], seed_lo=drop[2], seed_hi=drop[3])
    if z.dtype == F32:
        assert dy2 is None or dy2.dtype == F32
        dres = torch.empty((rows, D), dtype=F32, device=z.device)
        a.x32, a.dy32, a.dres32 = z.data_ptr(), _ptr(dy2), dres.data_ptr()
    else:
        dres = dh if drop[0] == 0 else torch.empty_like(z)
        a.x16, a.dy2, a.dres = z.data_ptr(), _ptr(dy2), dres.data_ptr()
    _ln_bwd_call(a, D, LN_POST, dgamma, dbeta, dbias, None, z.device, defer)
    return dh, dres


def ln_ls_bwd(dy, dstream, x_new, mean, rstd, w, h, ls_gamma, row_scale, rows_per_sample, dgamma, dbeta, dbias, dls,
              dy2=None, defer=None):
    """LS backward: dstream (fp32) updated in place to the gradient w.r.t. the incoming stream; returns dh (bf16)."""
    rows, D = x_new.shape
    dh = torch.empty((rows, D), dtype=BF16, device=x_new.device)
    a = LnBwdArgs(dy1=dy.data_ptr(), dy2=_ptr(dy2), x32=x_new.data_ptr(), mean=mean.data_ptr(), rstd=rstd.data_ptr(),
                  w=w.data_ptr(), dh=dh.data_ptr(), dstream=dstream.data_ptr(), h=h.data_ptr(), ls_gamma=ls_gamma.data_ptr(),
                  row_scale=_ptr(row_scale), rows=rows, rows_per_sample=rows_per_sample)
    _ln_bwd_call(a, D, LN_LS, dgamma, dbeta, dbias, dls, x_new.device, defer)
    return dh


# --------------------------------------------------------------------------------------------- attention
def _stat_ld(Sq):
    return (Sq + 3) // 4 * 4


def kv_groups(kv_index, U):
    """Group the query batch rows by key/value source (device-side, no host sync): returns (grp_start int32 [U+1],
    grp_rows int32 [B], U) for the grouped attention mode -- group u = the rows r with kv_index[r] == u."""
    idx = kv_index.long()
    order = torch.sort(idx, stable=True).indices.to(torch.int32).contiguous()
    counts = torch.zeros(U, dtype=torch.int64, device=idx.device).scatter_add_(0, idx, torch.ones_like(idx))
    start = torch.zeros(U + 1, dtype=torch.int32, device=idx.device)
    start[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return start, order, U


def attn_grouped_ok(Sq, Sk):
    """Shapes the grouped (one workgroup per key/value source) kernels cover: up to 64 queries per row; any number of keys (more than
    256 -- the 577 / 901 image tokens of the 384 / 480 px fine-tuning configurations -- stream through LDS chunk by chunk)."""
    return Sq <= 64


def _attn_args(q, k, v, o, lse, B, H, Sq, Sk, scale, bias, key_keep, causal, drop, bias_t=None, kv_index=None, groups=None,
               q_pack=None, k_pack=None, bias_tiles=None):
    for t in (q, k, v, o):
        assert t.dtype == BF16 and t.stride(-1) == 1 and t.dim() == 2
    for pk in (q_pack, k_pack):
        if pk is not None:  # (start, len) int32 [B]: packed (unpadded) token rows, see xfm_amd.packing
            assert bias is None and kv_index is None
            assert all(t.dtype == torch.int32 and t.numel() == B and t.is_contiguous() for t in pk)
    assert lse.shape[-1] == _stat_ld(Sq)
    if kv_index is not None:
        assert kv_index.dtype == torch.int32 and kv_index.numel() == B and kv_index.is_contiguous()
    g_start = g_rows = None
    n_groups = 0
    if groups is not None:
        g_start, g_rows, n_groups = groups
        assert kv_index is None and bias is None and not causal and attn_grouped_ok(Sq, Sk)
        assert g_start.dtype == torch.int32 and g_rows.dtype == torch.int32 and g_start.numel() == n_groups + 1 and g_rows.numel() == B
        assert k.shape[0] == n_groups * Sk
    tiled = tiled_t = None
    if bias_tiles is not None:  # (tiled, tiled_t) of bias_tiles(): the accumulator-layout copies the ViT-shape kernels read
        tiled, tiled_t = bias_tiles
        T = (Sq + 15) // 16
        assert bias is not None and Sq == Sk and all(t is None or (t.dtype == F32 and t.is_contiguous() and t.numel() == H * (T + e) * T * 256)
                                                     for t, e in ((tiled, 0), (tiled_t, 1)))
    return AttnArgs(stat_ld=lse.shape[-1], bias_t=_ptr(bias_t), bias_t_ld=0 if bias_t is None else bias_t.stride(1),
                    bias_tiled=_ptr(tiled), bias_t_tiled=_ptr(tiled_t),
                    kv_index=_ptr(kv_index), grp_start=_ptr(g_start), grp_rows=_ptr(g_rows), n_groups=n_groups,
                    q=q.data_ptr(), q_rs=q.stride(0), k=k.data_ptr(), k_rs=k.stride(0), v=v.data_ptr(), v_rs=v.stride(0),
                    o=o.data_ptr(), o_rs=o.stride(0), lse=lse.data_ptr(), bias=_ptr(bias),
                    bias_ld=0 if bias is None else bias.stride(1), key_keep=_ptr(key_keep), B=B, H=H, Sq=Sq, Sk=Sk,
                    scale=scale, causal=int(causal), drop_thresh=drop[0], drop_scale=drop[1], seed_lo=drop[2], seed_hi=drop[3],
                    q_start=_ptr(q_pack[0]) if q_pack else 0, q_len=_ptr(q_pack[1]) if q_pack else 0,
                    k_start=_ptr(k_pack[0]) if k_pack else 0, k_len=_ptr(k_pack[1]) if k_pack else 0)


def bias_tiles(bias, S, scale, fwd=True, bwd=True):
    """Dense additive bias fp32 [H, S, ld] -> (tiled, tiled_t): copies in the MFMA accumulator layout, divided by `scale`, that the
    batch-walking ViT-shape attention kernels read with one contiguous 1-KB load per tile (xfm_bias_tile in include/xfm_hip.h)."""
    H, T = bias.shape[0], (S + 15) // 16
    assert bias.dtype == F32 and bias.dim() == 3 and bias.stride(2) == 1 and bias.stride(0) == S * bias.stride(1)
    tiled = torch.empty(H * T * T * 256, dtype=F32, device=bias.device) if fwd else None
    tiled_t = torch.empty(H * (T + 1) * T * 256, dtype=F32, device=bias.device) if bwd else None
    check(_lib.load().xfm_bias_tile(bias.data_ptr(), H, S, bias.stride(1), float(scale), _ptr(tiled), _ptr(tiled_t), _stream()), "bias_tile")
    return tiled, tiled_t


def attn_fwd(q, k, v, B, H, Sq, Sk, scale, bias=None, key_keep=None, causal=False, drop=(0, 1.0, 0, 0), kv_index=None, groups=None,
             q_pack=None, k_pack=None, zero_fill=True, lo=False, bias_tiles=None):
    """q [B*Sq, >=H*64] / k, v [B*Sk, ...] are 2-D (possibly strided column slices of fused projection buffers).
    bias: dense fp32 [H,Sq,ld]; key_keep: int32 [B,Sk].  Returns (o [B*Sq, H*64] bf16, lse [B,H,Sq]).
    groups = kv_groups(...): grouped mode, k / v / key_keep hold one entry per SOURCE."""
    _dev(q)
    if q_pack is not None:  # one output row per packed query row; rows outside every sequence stay zero (finite for what follows;
        # zero_fill=False: the caller knows the sequences cover every row)
        o = (torch.zeros if zero_fill else torch.empty)((q.shape[0], H * 64), dtype=BF16, device=q.device)
    else:
        o = torch.empty((B * Sq, H * 64), dtype=BF16, device=q.device)
    lse = torch.empty((B, H, _stat_ld(Sq)), dtype=F32, device=q.device)
    if key_keep is not None:
        assert key_keep.dtype == torch.int32 and key_keep.is_contiguous()
    a = _attn_args(q, k, v, o, lse, B, H, Sq, Sk, scale, bias, key_keep, causal, drop, kv_index=kv_index, groups=groups,
                   q_pack=q_pack, k_pack=k_pack, bias_tiles=bias_tiles)
    o_lo = None
    if lo:  # second half of the output (what bf16 rounding of O lost): lets the backward skip its first pass over the keys
        o_lo = torch.empty_like(o)
        a.o_lo = o_lo.data_ptr()
    check(_lib.load().xfm_attn_fwd(ctypes.byref(a), _stream()), "attn_fwd")
    return (o, lse, o_lo) if lo else (o, lse)


def attn_bwd(dout, q, k, v, o, lse, dq, dk, dv, B, H, Sq, Sk, scale, bias=None, dbias=None, key_keep=None, causal=False,
             drop=(0, 1.0, 0, 0), bias_t=None, kv_index=None, groups=None, q_pack=None, k_pack=None, phase=0, delta=None, o_lo=None,
             bias_tiles=None):
    """Writes dq/dk/dv (2-D bf16 views with the same addressing convention as q/k/v); dbias (fp32 [H,Sq,ld]) += .
    Grouped mode: dk/dv are per SOURCE ([n_groups*Sk] rows), summed over each group's rows.
    Packed rows: only the rows of real tokens are written -- pass zero-initialised dq (/ dk / dv).
    phase 1 = the dQ kernel alone, phase 2 = the dK/dV kernel alone on the `delta` that the phase-1 call returned.  Returns delta."""
    a = _attn_args(q, k, v, o, lse, B, H, Sq, Sk, scale, bias, key_keep, causal, drop, bias_t, kv_index, groups, q_pack, k_pack, bias_tiles)
    if delta is None:
        assert phase != 2, "phase 2 needs the row statistics of the phase-1 call"
        delta = torch.empty((B, H, lse.shape[-1]), dtype=F32, device=q.device)  # the kernels never use its padding entries
    a.bwd_phase = phase
    if o_lo is not None:  # delta from dO . (o + o_lo): the dQ kernel runs one pass over the keys instead of two
        assert o_lo.shape == o.shape and o_lo.stride() == o.stride() and o_lo.dtype == BF16
        a.o_lo = o_lo.data_ptr()
    assert dout.dtype == BF16 and dout.stride(-1) == 1
    a.dout, a.do_rs = dout.data_ptr(), dout.stride(0)
    a.dq, a.dq_rs = dq.data_ptr(), dq.stride(0)
    a.dk, a.dk_rs = dk.data_ptr(), dk.stride(0)
    a.dv, a.dv_rs = dv.data_ptr(), dv.stride(0)
    a.delta, a.dbias = delta.data_ptr(), _ptr(dbias)
    if dbias is not None and phase != 2:
        # the library says how much scratch the bias gradient wants (xfm_attn_bwd_workspace): long sequences (the 577 / 901 tokens of
        # the 384 / 480 px ViT) a few MB of per-slice planes for the batch-walking kernel of dense unmasked problems, or the
        # [B, H, Sq, ld] per-entry dS of the general kernels; short ones nothing, or per-slice planes with XFM_DETERMINISTIC=1
        need = _lib.load().xfm_attn_bwd_workspace(ctypes.byref(a))
        if need > 0:
            a.dbias_ws = workspace(need, q.device).data_ptr()
    check(_lib.load().xfm_attn_bwd(ctypes.byref(a), _stream()), "attn_bwd")
    return delta


def rows_index_sum(src, index, U, rows_per_item):
    """src bf16 [R*rows_per_item, C] -> [U*rows_per_item, C]: item u = sum of the items r with index[r] == u."""
    R = index.numel()
    C = src.shape[1]
    assert src.is_contiguous() and src.dtype == BF16 and src.shape[0] == R * rows_per_item
    dst = torch.empty((U * rows_per_item, C), dtype=BF16, device=src.device)
    check(_lib.load().xfm_rows_index_sum(src.data_ptr(), index.data_ptr(), R, U, rows_per_item * C, dst.data_ptr(), _stream()),
          "rows_index_sum")
    return dst


def relpos_gather(table, index32, H, N, ld, transposed=False):
    """dense[h,i,j] = table[index[i,j], h] (rows padded to ld); transposed=True also returns the [h,j,i] copy."""
    dense = torch.empty((H, N, ld), dtype=F32, device=table.device)
    dense_t = torch.empty((H, N, ld), dtype=F32, device=table.device) if transposed else None
    check(_lib.load().xfm_relpos_gather(table.data_ptr(), index32.data_ptr(), H, N, ld, dense.data_ptr(), _ptr(dense_t),
                                        _stream()), "relpos_gather")
    return (dense, dense_t) if transposed else dense


def relpos_scatter(ddense, index32, H, N, ld, dtable):
    check(_lib.load().xfm_relpos_scatter(ddense.data_ptr(), index32.data_ptr(), H, N, ld, dtable.data_ptr(), _stream()),
          "relpos_scatter")


# --------------------------------------------------------------------------------------------- misc
def vit_tokens_fwd(tok, cls, mask_token, mask, Bx):
    """x0 [Bx, P+1, D] fp32 = [cls | tok[b mod Bt] with mask[b, i] patches replaced by mask_token]; tok fp32 [Bt, P, D], mask
    uint8 [Bx, P] or None."""
    _dev(tok)
    Bt, P, D = tok.shape
    assert tok.dtype == F32 and tok.is_contiguous() and cls.dtype == F32 and cls.numel() == D
    assert mask is None or (mask.dtype == torch.uint8 and mask.is_contiguous() and mask.shape == (Bx, P))
    x0 = torch.empty((Bx, P + 1, D), dtype=F32, device=tok.device)
    check(_lib.load().xfm_vit_tokens_fwd(tok.data_ptr(), cls.data_ptr(), _ptr(mask_token), _ptr(mask), Bt, Bx, P, D, x0.data_ptr(),
                                         _stream()), "vit_tokens_fwd")
    return x0


def vit_tokens_bwd(dx0, mask, Bt, dcls, dmask_token):
    """-> dtok fp32 [Bt, P, D]; dcls / dmask_token (fp32 [D]) are accumulated in place."""
    _dev(dx0)
    Bx, P1, D = dx0.shape
    assert dx0.dtype == F32 and dx0.is_contiguous()
    dtok = torch.empty((Bt, P1 - 1, D), dtype=F32, device=dx0.device)
    check(_lib.load().xfm_vit_tokens_bwd(dx0.data_ptr(), _ptr(mask), Bt, Bx, P1 - 1, D, dtok.data_ptr(), dcls.data_ptr(),
                                         _ptr(dmask_token), _stream()), "vit_tokens_bwd")
    return dtok


def pool_rows_fwd_(y, B, N):
    """In place on bf16 y [B*N, D]: row 0 of every sample <- mean of its rows 1..N-1."""
    _dev(y)
    assert y.dtype == BF16 and y.is_contiguous() and y.shape[0] == B * N
    check(_lib.load().xfm_pool_rows_fwd(y.data_ptr(), B, N, y.shape[1], _stream()), "pool_rows_fwd")
    return y


def pool_rows_bwd(dy, B, N):
    """Gradient of pool_rows_fwd_ w.r.t. its input rows: out[b, 0] = 0, out[b, 1+i] = dy[b, 1+i] + dy[b, 0] / (N - 1)."""
    assert dy.dtype == BF16 and dy.is_contiguous() and dy.shape[0] == B * N
    out = torch.empty_like(dy)
    check(_lib.load().xfm_pool_rows_bwd(dy.data_ptr(), B, N, dy.shape[1], out.data_ptr(), _stream()), "pool_rows_bwd")
    return out


def mim_loss_fwd(x, t, mask):
    """-> sums fp32 [3] = (sum (x-t)^2 over masked patch rows, over cls rows, number of masked patches); x, t bf16 [B, N, D]."""
    _dev(x)
    B, N, D = x.shape
    assert x.dtype == BF16 and t.dtype == BF16 and x.is_contiguous() and t.is_contiguous() and t.shape == x.shape
    assert mask.dtype == torch.uint8 and mask.is_contiguous() and mask.shape == (B, N - 1)
    sums = torch.zeros(_lib.MIM_SUMS_FLOATS, dtype=F32, device=x.device)   # [0:3] the sums, the rest scratch of the two-stage reduction
    check(_lib.load().xfm_mim_loss_fwd(x.data_ptr(), t.data_ptr(), mask.data_ptr(), B, N, D, sums.data_ptr(), _stream()), "mim_loss_fwd")
    return sums


def mim_loss_bwd(x, t, mask, sums, gout, cls_term):
    B, N, D = x.shape
    dx = torch.empty_like(x)
    check(_lib.load().xfm_mim_loss_bwd(x.data_ptr(), t.data_ptr(), mask.data_ptr(), sums.data_ptr(), gout.data_ptr(), int(cls_term), B, N, D,
                                       dx.data_ptr(), _stream()), "mim_loss_bwd")
    return dx


def mim_masks(B, grid, num, min_num, device, seed, min_aspect=0.3, max_aspect=None, delta_hist=None):
    """[B, grid*grid] bool block-wise MIM masks drawn on the device (masking_generator.py:27-105), exactly `num` patches each."""
    out = torch.empty((B, grid * grid), dtype=torch.uint8, device=device)
    check(_lib.load().xfm_mim_masks(B, grid, grid, num, min_num, float(min_aspect), float(max_aspect or 1.0 / min_aspect), int(seed),
                                    out.data_ptr(), _ptr(delta_hist), _stream()), "mim_masks")
    return out.view(torch.bool)


def patchify(image, patch):
    _dev(image)
    assert image.dtype == F32 and image.is_contiguous()
    B, C, H, W = image.shape
    out = torch.empty((B * (H // patch) * (W // patch), C * patch * patch), dtype=BF16, device=image.device)
    check(_lib.load().xfm_patchify(image.data_ptr(), B, C, H, W, patch, out.data_ptr(), _stream()), "patchify")
    return out


def _embed_args(ids, word, pos, typ, w, b, eps, pad_id, drop, pos_mode=0):
    B, T = ids.shape
    assert pos_mode == 0 or T <= pos.shape[0]
    return EmbedArgs(pos_mode=pos_mode, ids=ids.data_ptr(), word=word.data_ptr(), pos=pos.data_ptr(), type=typ.data_ptr(), w=w.data_ptr(),
                     b=b.data_ptr(), B=B, T=T, pad_id=pad_id, eps=eps, drop_thresh=drop[0], drop_scale=drop[1],
                     seed_lo=drop[2], seed_hi=drop[3])


def embed_ln_fwd(ids, word, pos, typ, w, b, eps, pad_id, drop=(0, 1.0, 0, 0), pos_mode=0, row_map=None, out_rows=None, y32=False):
    """row_map (int32 [B*T], -1 = skip) + out_rows: write the tokens to packed rows of a [out_rows, D] output.
    y32: also return the fp32 twin of y (the un-rounded values, for the encoder's fp32 residual stream)."""
    _dev(ids)
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    B, T = ids.shape
    D = word.shape[1]
    if row_map is not None:
        assert row_map.dtype == torch.int32 and row_map.numel() == B * T and row_map.is_contiguous()
        y = torch.zeros((out_rows, D), dtype=BF16, device=ids.device)
        yf = torch.zeros((out_rows, D), dtype=F32, device=ids.device) if y32 else None
    else:
        y = torch.empty((B * T, D), dtype=BF16, device=ids.device)
        yf = torch.empty((B * T, D), dtype=F32, device=ids.device) if y32 else None
    mean = torch.empty(B * T, dtype=F32, device=ids.device)
    rstd = torch.empty(B * T, dtype=F32, device=ids.device)
    pos_ids = torch.empty(B * T, dtype=torch.int32, device=ids.device)
    a = _embed_args(ids, word, pos, typ, w, b, eps, pad_id, drop, pos_mode)
    a.row_map = _ptr(row_map)
    a.y, a.mean, a.rstd, a.pos_ids, a.y32 = y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), pos_ids.data_ptr(), _ptr(yf)
    check(_lib.load().xfm_embed_ln_fwd(ctypes.byref(a), D, _stream()), "embed_ln_fwd")
    return (y, mean, rstd, pos_ids, yf) if y32 else (y, mean, rstd, pos_ids)


def embed_ln_bwd(dy, ids, word, pos, typ, w, b, eps, pad_id, mean, rstd, pos_ids, dword, dpos, dtype_, dgamma, dbeta,
                 drop=(0, 1.0, 0, 0), pos_mode=0, row_map=None, dy32=None):
    """dy32: optional fp32 gradient (same rows) added to dy."""
    D = word.shape[1]
    a = _embed_args(ids, word, pos, typ, w, b, eps, pad_id, drop, pos_mode)
    a.row_map = _ptr(row_map)
    a.mean, a.rstd, a.pos_ids = mean.data_ptr(), rstd.data_ptr(), pos_ids.data_ptr()
    a.dy, a.dword, a.dpos, a.dy32 = dy.data_ptr(), dword.data_ptr(), dpos.data_ptr(), _ptr(dy32)
    lib = _lib.load()
    ws = workspace(lib.xfm_embed_ln_bwd_workspace(a.B * a.T, D), dy.device)
    dz = None
    if deterministic():   # the per-token gradient is stored and scattered in sorted runs, one owner per embedding row (no float atomics)
        dz = torch.empty((a.B * a.T, D), dtype=F32, device=dy.device)
        a.dz_out = dz.data_ptr()
    check(lib.xfm_embed_ln_bwd(ctypes.byref(a), D, _ptr(dgamma), _ptr(dbeta), _ptr(dtype_), ws.data_ptr(), ws.numel() * 4,
                               _stream()), "embed_ln_bwd")
    if dz is not None:
        R = a.B * a.T
        live = None if row_map is None else (row_map.reshape(-1) >= 0)
        for keys, table, skip in ((ids.reshape(-1).to(torch.int64), dword, pad_id), (pos_ids.reshape(-1).to(torch.int64), dpos, -1 if pos_mode else pad_id)):
            if live is not None:
                keys = torch.where(live, keys, torch.full_like(keys, -1))   # tokens the kernel skipped wrote no dz row
            skey, perm = torch.sort(keys, stable=True)
            check(lib.xfm_rows_segment_sum(dz.data_ptr(), perm.data_ptr(), skey.data_ptr(), R, D, int(skip), table.data_ptr(), _stream()),
                  "rows_segment_sum")


def rows_gather(src, index, out=None):
    """out[r, :] = src[index[r], :] (zero row where index[r] < 0); src bf16 (or fp32) [*, D] contiguous, index int32 [R]."""
    _dev(src)
    if src.dtype == F32:   # an fp32 row of D values is a bf16 row of 2 D elements to a row copy
        assert out is None or out.dtype == F32
        return rows_gather(src.view(BF16), index, None if out is None else out.view(BF16)).view(F32)
    assert src.dtype == BF16 and src.dim() == 2 and src.is_contiguous() and index.dtype == torch.int32 and index.is_contiguous()
    R, D = index.numel(), src.shape[1]
    if out is None:
        out = torch.empty((R, D), dtype=BF16, device=src.device)
    check(_lib.load().xfm_rows_gather(src.data_ptr(), index.data_ptr(), R, D, out.data_ptr(), _stream()), "rows_gather")
    return out


def rows_scatter_add(src, index, dst32):
    """dst32[index[r], :] += src[r, :] (fp32 accumulation, rows with index[r] < 0 skipped): the adjoint of rows_gather."""
    _dev(src)
    assert src.dtype == BF16 and src.dim() == 2 and src.is_contiguous() and dst32.dtype == F32 and dst32.is_contiguous()
    assert index.dtype == torch.int32 and index.is_contiguous() and index.numel() == src.shape[0] and dst32.shape[1] == src.shape[1]
    check(_lib.load().xfm_rows_scatter_add(src.data_ptr(), index.data_ptr(), src.shape[0], src.shape[1], dst32.data_ptr(), _stream()),
          "rows_scatter_add")
    return dst32


def rownorm_fwd(x):
    """F.normalize(x, dim=-1) on fp32 rows -> (y, 1 / max(norm, 1e-12))."""
    _dev(x)
    assert x.dtype == F32 and x.dim() == 2 and x.is_contiguous()
    y = torch.empty_like(x)
    inv = torch.empty(x.shape[0], dtype=F32, device=x.device)
    check(_lib.load().xfm_rownorm_fwd(x.data_ptr(), x.shape[0], x.shape[1], y.data_ptr(), inv.data_ptr(), _stream()), "rownorm_fwd")
    return y, inv


def rownorm_bwd(dy, y, inv):
    assert dy.dtype == F32 and dy.is_contiguous() and dy.shape == y.shape
    dx = torch.empty_like(y)
    check(_lib.load().xfm_rownorm_bwd(dy.data_ptr(), y.data_ptr(), inv.data_ptr(), y.shape[0], y.shape[1], dx.data_ptr(), _stream()),
          "rownorm_bwd")
    return dx


def _idx64(idx, n):
    if idx is None:
        return None
    idx = idx.reshape(-1).to(torch.int64).contiguous()
    assert idx.numel() == n and idx.is_cuda
    return idx


def itc_fwd(image_feat, text_feat, temp, idx=None):
    """Contrastive loss over N gathered rows (xfm.py:683-715) -> (loss [1], lse [2N], cnt [N] | None).  idx (int64 [N], the gathered
    image ids): soft labels over the rows that share an id (xfm.py:705-713); None: in-batch labels."""
    _dev(image_feat)
    N, E = image_feat.shape
    assert image_feat.dtype == F32 and text_feat.dtype == F32 and image_feat.is_contiguous() and text_feat.is_contiguous()
    assert text_feat.shape == (N, E) and temp.dtype == F32 and temp.numel() == 1
    idx = _idx64(idx, N)
    lse = torch.empty(4 * N, dtype=F32, device=image_feat.device)   # [0, 2N): row statistics; [2N, 4N): the rows' loss terms
    cnt = torch.empty(N, dtype=F32, device=image_feat.device) if idx is not None else None
    loss = torch.zeros(2, dtype=F32, device=image_feat.device)      # [0] the loss, [1] the kernel's ticket counter
    check(_lib.load().xfm_itc_fwd(image_feat.data_ptr(), text_feat.data_ptr(), temp.data_ptr(), N, E, lse.data_ptr(), loss.data_ptr(),
                                  _ptr(idx), _ptr(cnt), _stream()), "itc_fwd")
    return loss[:1], lse[:2 * N], cnt


def itc_bwd(image_feat, text_feat, temp, lse, g, idx=None, cnt=None):
    """-> (d image_feat, d text_feat, d temp [1]) for the upstream gradient g (fp32 [1])."""
    N, E = image_feat.shape
    idx = _idx64(idx, N)
    if lse.untyped_storage().nbytes() < (lse.storage_offset() + 3 * N) * 4:   # (not itc_fwd's [4N] buffer: the backward parks N shares behind 2N)
        big = torch.empty(4 * N, dtype=F32, device=lse.device)
        big[:2 * N] = lse[:2 * N]
        lse = big
    dI, dT = torch.empty_like(image_feat), torch.empty_like(text_feat)
    dtemp = torch.zeros(1, dtype=F32, device=image_feat.device)
    check(_lib.load().xfm_itc_bwd(image_feat.data_ptr(), text_feat.data_ptr(), temp.data_ptr(), lse.data_ptr(), g.data_ptr(), N, E,
                                  dI.data_ptr(), dT.data_ptr(), dtemp.data_ptr(), _ptr(idx), _ptr(cnt), _stream()), "itc_bwd")
    return dI, dT, dtemp


def hard_negatives(image_feat, text_feat, temp, seed, idx=None):
    """One categorical draw per row from softmax(sim / temp) + 1e-5 with the own entry -- with idx (int64 [B]): every entry of the
    same image id -- zeroed (xfm.py:717-746) -> (image_neg_idx, text_neg_idx) int64 [B]."""
    _dev(image_feat)
    B, E = image_feat.shape
    assert image_feat.dtype == F32 and text_feat.dtype == F32 and image_feat.is_contiguous() and text_feat.is_contiguous()
    idx = _idx64(idx, B)
    out = torch.empty((2, B), dtype=torch.int64, device=image_feat.device)
    check(_lib.load().xfm_hard_negatives(image_feat.data_ptr(), text_feat.data_ptr(), temp.data_ptr(), B, E, int(seed),
                                         out[0].data_ptr(), out[1].data_ptr(), _ptr(idx), _stream()), "hard_negatives")
    return out[0], out[1]


def ce_fwd(logits, V, labels):
    """logits fp32 [R, ld>=V]; labels int64 [R] -> (lse [R], loss_rows [R])."""
    R = logits.shape[0]
    lse = torch.empty(R, dtype=F32, device=logits.device)
    loss = torch.empty(R, dtype=F32, device=logits.device)
    check(_lib.load().xfm_ce_fwd(logits.data_ptr(), logits.stride(0), R, V, labels.data_ptr(), lse.data_ptr(), loss.data_ptr(),
                                 _stream()), "ce_fwd")
    return lse, loss


def ce_bwd(logits, V, labels, lse, scale, ldd):
    """-> dlogits bf16 [R, ldd] = (softmax - onehot) * scale (zero in ignored rows / padding columns);
    scale: fp32 [1] (one factor) or [R] (per-row upstream gradients, reduction 'none')."""
    R = logits.shape[0]
    d = torch.empty((R, ldd), dtype=BF16, device=logits.device)
    assert scale.dtype == F32 and scale.numel() in (1, R)
    check(_lib.load().xfm_ce_bwd(logits.data_ptr(), logits.stride(0), R, V, labels.data_ptr(), lse.data_ptr(), scale.data_ptr(),
                                 int(scale.numel() == R and R > 1), d.data_ptr(), ldd, _stream()), "ce_bwd")
    return d


def sumsq(x, out):
    ws = workspace(4096, x.device)  # XFM_SUMSQ_WORKSPACE_FLOATS block partials
    check(_lib.load().xfm_sumsq(x.data_ptr(), x.numel(), out.data_ptr(), ws.data_ptr(), _stream()), "sumsq")


def adamw(p, g, m, v, group, lrs, wds, beta1, beta2, eps, step, clip_coef=None, zero_grad=False):
    """zero_grad: g is zeroed in the same sweep (saves the separate fill pass over the live gradient ranges)."""
    a = AdamWArgs(p=p.data_ptr(), g=g.data_ptr(), m=m.data_ptr(), v=v.data_ptr(), group=group.data_ptr(),
                  beta1=beta1, beta2=beta2, eps=eps, bc1=1.0 - beta1 ** step, bc2=1.0 - beta2 ** step,
                  clip_coef=_ptr(clip_coef), n=p.numel(), zero_grad=int(zero_grad))
    for i in range(4):
        a.lr[i] = lrs[i] if i < len(lrs) else 0.0
        a.wd[i] = wds[i] if i < len(wds) else 0.0
    check(_lib.load().xfm_adamw(ctypes.byref(a), _stream()), "adamw")


def relpos_sorted_index(index32, entries):
    """(order, start) for relpos_scatter_sorted: positions sorted by table entry and the per-entry offsets (built once)."""
    flat = index32.reshape(-1).long()
    order = torch.sort(flat, stable=True).indices.to(torch.int32).contiguous()
    counts = torch.zeros(entries, dtype=torch.int64, device=flat.device).scatter_add_(0, flat, torch.ones_like(flat))
    start = torch.zeros(entries + 1, dtype=torch.int32, device=flat.device)
    start[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return order, start


def relpos_grid_grad(ddense, G, H, ld, dtable):
    """dtable[(2G-1)^2 + 3, H] += the relative-position-table gradient of ddense [H, G*G+1, ld] for the STANDARD grid index
    (beit2.build_relative_position_index): coalesced, no atomics, no sorted index."""
    assert ddense.dtype == F32 and dtable.dtype == F32 and dtable.shape == ((2 * G - 1) ** 2 + 3, H) and dtable.is_contiguous()
    check(_lib.load().xfm_relpos_grid_grad(ddense.data_ptr(), H, G, ld, dtable.data_ptr(), _stream()), "relpos_grid_grad")


def relpos_scatter_sorted(ddense, order, start, H, N, ld, dtable):
    check(_lib.load().xfm_relpos_scatter_sorted(ddense.data_ptr(), order.data_ptr(), start.data_ptr(), start.numel() - 1, H, N, ld,
                                                dtable.data_ptr(), _stream()), "relpos_scatter_sorted")

