"""Autograd nodes for the stand-alone ops of the path (heads, LM head + cross-entropy).  Weight / bias gradients are
accumulated straight into the gradient arena by the wgrad kernels; autograd only carries activation gradients."""
import math

import torch

from . import functional as Fx

BF16, F32 = torch.bfloat16, torch.float32


def grad_view(p):
    from .arena import grad_of
    return grad_of(p)


class _LinearSlotFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slot, x_requires_grad, out_fp32, anchor):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.dtype != BF16:
            x2 = x2.to(BF16)
        x2 = x2.contiguous()
        y = Fx.gemm_nt(x2, slot.wb, slot.b, epi=Fx.EPI_F32 if out_fp32 else Fx.EPI_BF16)
        ctx.slot, ctx.x2, ctx.xshape, ctx.need_dx, ctx.xdtype = slot, x2, x.shape, x_requires_grad, x.dtype
        return y.view(*x.shape[:-1], slot.N)

    @staticmethod
    def backward(ctx, dy):
        s = ctx.slot
        dy2 = dy.reshape(-1, s.N)
        dy2 = (dy2 if dy2.dtype == BF16 else dy2.to(BF16)).contiguous()
        if s.N % 8 != 0:  # tiny heads (e.g. the 2-way ITM logits): pad the columns to the kernels' 16-byte granularity
            pad = (s.N + 63) // 64 * 64
            dyp = torch.zeros((dy2.shape[0], pad), dtype=BF16, device=dy2.device)
            dyp[:, :s.N] = dy2
            dy2 = dyp
        if dy2.is_cuda:   # the weight gradient leaves the chain: second stream, re-joined when the backward pass is over
            from .xroberta import _WgradStream
            wg = _WgradStream(dy2.device)
            wg.gemm_tn(dy2, ctx.x2, s.dw, n=s.N, dbias=s.db)
            wg.join_at_end()
        else:
            Fx.gemm_tn(dy2, ctx.x2, s.dw, n=s.N, dbias=s.db)
        dx = None
        if ctx.need_dx:
            if dy2.shape[1] % 64 == 0 and dy2.shape[1] <= s.wt.shape[1]:
                dx = Fx.gemm_nt(dy2, s.wt[:, :dy2.shape[1]], n=s.K)
            else:  # contraction dim must be a multiple of 64: zero-pad both operands
                pad = (dy2.shape[1] + 63) // 64 * 64
                dyp = torch.zeros((dy2.shape[0], pad), dtype=BF16, device=dy2.device)
                dyp[:, :dy2.shape[1]] = dy2
                wtp = torch.zeros((s.K, pad), dtype=BF16, device=dy2.device)
                wtp[:, :s.N] = s.wt[:, :s.N]
                dx = Fx.gemm_nt(dyp, wtp, n=s.K)
            dx = dx.view(ctx.xshape).to(ctx.xdtype)
        return dx, None, None, None, None


def linear_slot(x, slot, x_requires_grad=True, out_fp32=False):
    """y = x @ W^T + b through the bf16 MFMA GEMM.  `anchor` makes the node differentiable even when x is not."""
    anchor = slot.weights[0]
    return _LinearSlotFn.apply(x, slot, x_requires_grad and x.requires_grad, out_fp32, anchor)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, gelu):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        y, mean, rstd = Fx.ln_fwd(x2, mod.weight, mod.bias, mod.eps, gelu=gelu)
        ctx.mod, ctx.saved, ctx.shape, ctx.dtype, ctx.gelu = mod, (x2, mean, rstd), x.shape, x.dtype, gelu
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd = ctx.saved
        m = ctx.mod
        dy2 = dy.reshape(x2.shape)
        dy2 = (dy2 if dy2.dtype == BF16 else dy2.to(BF16)).contiguous()
        gb = m.bias if ctx.gelu else None
        if ctx.dtype == F32:
            dx = torch.empty_like(x2)
            Fx.ln_bwd(dy2, x2, mean, rstd, m.weight, grad_view(m.weight), grad_view(m.bias), dx32=dx, gelu_b=gb)
        else:
            dx = torch.empty_like(x2)
            Fx.ln_bwd(dy2, x2, mean, rstd, m.weight, grad_view(m.weight), grad_view(m.bias), dx16=dx, gelu_b=gb)
        return dx.view(ctx.shape), None, None


def layer_norm(x, mod, gelu=False):
    """LayerNorm over the last dim (fp32 or bf16 in, bf16 out) using `mod.weight/bias/eps`; gelu: GELU(LayerNorm(x)) in the same
    kernel (the Linear -> LayerNorm -> GELU heads, xfm.py:115-121)."""
    return _LayerNormFn.apply(x, mod, bool(gelu))


class _RowNormFn(torch.autograd.Function):
    """F.normalize(x, dim=-1) on fp32 rows (xfm.py:617-620)."""

    @staticmethod
    def forward(ctx, x):
        y, inv = Fx.rownorm_fwd(x.contiguous())
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        return Fx.rownorm_bwd(dy.float().contiguous(), y, inv)


def row_normalize(x):
    return _RowNormFn.apply(x)


class _ItcFn(torch.autograd.Function):
    """(CE(I T^T / temp, labels) + CE(T I^T / temp, labels)) / 2 (xfm.py:683-715) as one kernel each way; labels = arange, or with
    `idx` the soft labels over the rows that share an image id (xfm.py:705-713)."""

    @staticmethod
    def forward(ctx, image_feat, text_feat, temp, idx):
        I, T = image_feat.float().contiguous(), text_feat.float().contiguous()
        tv = temp.detach().float().reshape(1)
        loss, lse, cnt = Fx.itc_fwd(I, T, tv, idx)
        ctx.save_for_backward(I, T, tv, lse)
        ctx.idx, ctx.cnt = idx, cnt
        ctx.temp_shape = temp.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        I, T, tv, lse = ctx.saved_tensors
        dI, dT, dtemp = Fx.itc_bwd(I, T, tv, lse, g.float().reshape(1).contiguous(), ctx.idx, ctx.cnt)
        return dI, dT, (dtemp.reshape(ctx.temp_shape) if ctx.needs_input_grad[2] else None), None


def itc_loss(image_feat, text_feat, temp, idx=None):
    if not torch.is_tensor(temp):
        temp = torch.full((1,), float(temp), dtype=F32, device=image_feat.device)
    return _ItcFn.apply(image_feat, text_feat, temp, idx)


class _SmallCEFn(torch.autograd.Function):
    """Mean cross-entropy over a handful of classes (the 2-way ITM head, xfm.py:795-800; the classification heads,
    model_classification.py:67) through the vocabulary CE kernels.  F.cross_entropy semantics: labels of -100 are ignored and
    the mean is over the VALID rows.  Any class count: the kernels read rows of 4-column granularity, so a logits buffer
    whose width is not a multiple of 4 is copied into a padded one (ld is passed separately from C)."""

    @staticmethod
    def forward(ctx, logits, labels, n_valid):
        lg = logits.float().contiguous()
        R, C = lg.shape
        if C >= 4 and C % 4:
            padded = torch.empty((R, (C + 3) // 4 * 4), dtype=F32, device=lg.device)
            padded[:, :C] = lg
            lg = padded
        labels = labels.reshape(-1).contiguous()
        lse, rows = Fx.ce_fwd(lg, C, labels)
        if n_valid is None:
            nvalid = (labels != -100).sum().clamp(min=1).to(F32)
        else:   # the caller built the labels itself (ITM: ones | zeros) -- no counting kernels on the step's critical path
            nvalid = torch.full((), float(n_valid), dtype=F32, device=lg.device)
        ctx.save_for_backward(lg, labels, lse, nvalid)
        ctx.C = C
        return rows.sum() / nvalid

    @staticmethod
    def backward(ctx, g):
        lg, labels, lse, nvalid = ctx.saved_tensors
        C = ctx.C
        d = Fx.ce_bwd(lg, C, labels, lse, (g.float() / nvalid).reshape(1), (C + 7) // 8 * 8)
        return d[:, :C].float(), None, None


def small_ce(logits, labels, n_valid=None):
    """n_valid: the number of labels != -100 when the caller knows it (skips the count)."""
    return _SmallCEFn.apply(logits, labels, n_valid)


def gelu_grad(u):
    u = u.float()
    return 0.5 * (1.0 + torch.erf(u * (1.0 / math.sqrt(2.0)))) + u * torch.exp(-0.5 * u * u) * (1.0 / math.sqrt(2.0 * math.pi))


VOCAB_LD = 64  # logits / dlogits row stride is padded to a multiple of this (the dgrad GEMM contracts over it)


class _LMHeadCEFn(torch.autograd.Function):
    """RobertaLMHead (dense -> GELU -> LayerNorm -> decoder, xroberta.py:1325-1333) fused with the vocabulary
    cross-entropy (ignore_index=-100, xroberta.py:1296-1297).  Returns (reduced loss, fp32 logits view)."""

    @staticmethod
    def forward(ctx, x, head, labels, reduction):
        R, D = x.shape
        sd, sv = head._slot_dense, head._slot_decoder
        V = sv.N
        hact, u = Fx.gemm_nt(x, sd.wb, sd.b, epi=Fx.EPI_GELU)
        y, mean, rstd = Fx.ln_fwd(hact, head.layer_norm.weight, head.layer_norm.bias, head.layer_norm.eps)
        ldl = (V + VOCAB_LD - 1) // VOCAB_LD * VOCAB_LD
        logits = torch.empty((R, ldl), dtype=F32, device=x.device)
        # (few rows against the whole vocabulary: 4 x 197 tiles of 256 x 256 are three rounds of the persistent kernel -- 127 us against
        # 148 us for the plan's 256 x 128 ring at 960 rows; below ~512 rows the plan's choice stands)
        Fx.gemm_nt(y, sv.wb, sv.b, epi=Fx.EPI_F32, out=logits, n=V, tile_hint=5 if R >= 512 else 0)
        labels = labels.reshape(-1).contiguous()
        lse, loss_rows = Fx.ce_fwd(logits, V, labels)
        nvalid = (labels != -100).sum().clamp(min=1).to(F32)
        ctx.saved = (x, hact, u, y, mean, rstd, logits, labels, lse, nvalid)
        ctx.head, ctx.reduction = head, reduction
        ctx.mark_non_differentiable(logits)
        if reduction == "mean":
            return loss_rows.sum() / nvalid, logits
        if reduction == "sum":
            return loss_rows.sum(), logits
        return loss_rows, logits

    @staticmethod
    def backward(ctx, g, _):
        x, hact, u, y, mean, rstd, logits, labels, lse, nvalid = ctx.saved
        head = ctx.head
        sd, sv = head._slot_dense, head._slot_decoder
        V = sv.N
        if ctx.reduction == "none":
            scale = g.reshape(-1).to(F32).contiguous()  # per-row upstream gradients
        else:
            scale = (g / nvalid if ctx.reduction == "mean" else g).reshape(1).to(F32).contiguous()
        dlogits = Fx.ce_bwd(logits, V, labels, lse, scale, logits.shape[1])
        # the two weight gradients (the 50265 x 768 one takes 310 us) go to the second stream and re-join at the end of the backward
        # pass: they run under the fusion tower's latency-bound activation-gradient chain instead of in front of it
        from .xroberta import _WgradStream
        wg = _WgradStream(x.device)
        wg.gemm_tn(dlogits, y, sv.dw, n=V, dbias=sv.db)
        # dgrad of the vocabulary projection: K = 50304 against a 960 x 768 output -- K is sliced over the grid instead of 90 workgroups
        # walking 786 K-tiles each, and the slices are summed in a fixed order (xfm_gemm_nt_ksplit): this activation gradient is rounded
        # to bf16 right here, so an order-dependent sum (fp32 atomics, rounds 1-3) made the whole backward below it bimodal
        dy = Fx.gemm_nt_ksplit(dlogits, sv.wt, n=sv.K)
        dhact = torch.empty_like(hact)
        ln = head.layer_norm
        Fx.ln_bwd(dy, hact, mean, rstd, ln.weight, grad_view(ln.weight), grad_view(ln.bias), dx16=dhact)
        du = (dhact.float() * u.float()).to(BF16)  # u = gelu'(pre-activation), stored by the forward epilogue
        wg.gemm_tn(du, x, sd.dw, dbias=sd.db)
        dx = Fx.gemm_nt(du, sd.wt, n=sd.K)
        wg.join_at_end()
        return dx, None, None, None


def lm_head_ce(x, head, labels, reduction="mean"):
    return _LMHeadCEFn.apply(x.contiguous(), head, labels, reduction)


def lm_head_logits(x, head):
    """Inference-only logits (return_logits=True path, xroberta.py:1287-1288)."""
    sd, sv = head._slot_dense, head._slot_decoder
    with torch.no_grad():
        hact, _ = Fx.gemm_nt(x.contiguous(), sd.wb, sd.b, epi=Fx.EPI_GELU)
        y, _, _ = Fx.ln_fwd(hact, head.layer_norm.weight, head.layer_norm.bias, head.layer_norm.eps)
        return Fx.gemm_nt(y, sv.wb, sv.b, epi=Fx.EPI_F32)
