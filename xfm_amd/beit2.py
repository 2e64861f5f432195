"""BEiT-v2 vision tower on the HIP hot path, behind the reference's interface.

Mirrors `models/beit2.py::VisionTransformer` of the reference (constructor keywords :272-281, forward :477-481,
forward_avgpool :423-466, state_dict keys) for the configuration every shipped YAML uses: relative-position bias
per block, layer-scale gammas, q/v-only bias, mean-pooled pseudo-cls, no absolute position embedding.  The twelve
blocks run as ONE autograd node whose forward/backward are explicit sequences of C-ABI kernel launches
(xfm_amd.functional); PyTorch autograd only sees the patch/cls assembly before it and the pooling after it.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import functional as Fx
from .arena import LinearSlot, OwnsArena, ParamArena
from .ops import linear_slot


class BlockMaskGenerator:
    """Block-wise MIM mask sampler with the distribution of the reference's MaskingGenerator
    (models/masking_generator.py:27-105): rectangles of area U[min, remaining] and log-uniform aspect in [0.3, 1/0.3],
    accepted when they add between 1 and `remaining` new patches, then trimmed / topped up to exactly `num_masking`."""

    def __init__(self, input_size, num_masking_patches, min_num_patches=4, min_aspect=0.3, seed=None):
        self.h = self.w = int(input_size)
        self.num = num_masking_patches
        self.min_num = min_num_patches
        self.log_aspect = (math.log(min_aspect), math.log(1.0 / min_aspect))
        self.rng = np.random.default_rng(seed)
        self._seed0 = 0 if seed is None else int(seed)

    def _one(self):
        rng, H, W = self.rng, self.h, self.w
        mask = np.zeros((H, W), dtype=bool)
        count = 0
        while count < self.num:
            budget = self.num - count
            delta = 0
            for _ in range(10):
                area = self.min_num + (budget - self.min_num) * rng.random()  # like random.uniform: fine when budget < min
                ar = math.exp(rng.uniform(*self.log_aspect))
                h, w = int(round(math.sqrt(area * ar))), int(round(math.sqrt(area / ar)))
                if w < W and h < H:
                    top, left = int(rng.integers(0, H - h + 1)), int(rng.integers(0, W - w + 1))
                    sub = mask[top:top + h, left:left + w]
                    new = h * w - int(sub.sum())
                    if 0 < new <= budget:
                        sub[...] = True
                        delta = new
                        break
            if delta == 0:
                break
            count += delta
        flat = mask.reshape(-1)
        if count < self.num:
            free = np.flatnonzero(~flat)
            flat[rng.choice(free, self.num - count, replace=False)] = True
        elif count > self.num:
            used = np.flatnonzero(flat)
            flat[rng.choice(used, count - self.num, replace=False)] = False
        return flat

    def batch(self, n, device=None):
        """[n, h*w] bool.  On a HIP device the masks are DRAWN there (csrc/elementwise.hip mim_masks_kernel: one wavefront per image
        runs the same rejection loop with a counter-based generator; ~10 us for a batch instead of n Python rejection loops on the
        thread that also enqueues the step's kernels, beit2.py:432-439 of the reference).  On the CPU: the numpy sampler."""
        if device is not None and torch.device(device).type == "cuda":
            self._draws = getattr(self, "_draws", 0) + 1
            seed = ((torch.initial_seed() & 0xFFFFFFFF) << 32) | ((self._seed0 + self._draws) & 0xFFFFFFFF)
            return Fx.mim_masks(n, self.h, self.num, self.min_num, device, seed, min_aspect=math.exp(self.log_aspect[0]),
                                max_aspect=math.exp(self.log_aspect[1]))
        return torch.from_numpy(np.stack([self._one() for _ in range(n)], 0))


def build_relative_position_index(gh, gw):
    """[gh*gw+1]^2 int64 index into the (2gh-1)(2gw-1)+3 row bias table; last three rows are cls->tok, tok->cls,
    cls->cls (same values as beit2.py:92-116)."""
    n = gh * gw
    nrd = (2 * gh - 1) * (2 * gw - 1) + 3
    r = torch.arange(n)
    y, x = r // gw, r % gw
    rel = (y[:, None] - y[None, :] + gh - 1) * (2 * gw - 1) + (x[:, None] - x[None, :] + gw - 1)
    idx = torch.full((n + 1, n + 1), nrd - 1, dtype=torch.int64)
    idx[1:, 1:] = rel
    idx[0, 1:] = nrd - 3
    idx[1:, 0] = nrd - 2
    return idx


class _Affine(nn.Module):
    """weight/bias container for a LayerNorm (arithmetic runs in the fused HIP kernels)."""

    def __init__(self, dim, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.eps = eps


class _Dense(nn.Module):
    """weight/bias container for a Linear."""

    def __init__(self, fin, fout, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.zeros(fout)) if bias else None
        nn.init.trunc_normal_(self.weight, std=0.02, a=-2.0, b=2.0)


class Attention(nn.Module):
    def __init__(self, dim, num_heads, window_size):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = _Dense(dim, dim * 3, bias=False)
        self.q_bias = nn.Parameter(torch.zeros(dim))
        self.v_bias = nn.Parameter(torch.zeros(dim))
        self.window_size = window_size
        nrd = (2 * window_size[0] - 1) * (2 * window_size[1] - 1) + 3
        self.relative_position_bias_table = nn.Parameter(torch.zeros(nrd, num_heads))
        self.register_buffer("relative_position_index", build_relative_position_index(*window_size))
        self.proj = _Dense(dim, dim)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = _Dense(dim, hidden)
        self.fc2 = _Dense(hidden, dim)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, init_values, window_size, drop_path, eps):
        super().__init__()
        self.norm1 = _Affine(dim, eps)
        self.attn = Attention(dim, num_heads, window_size)
        self.norm2 = _Affine(dim, eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.gamma_1 = nn.Parameter(init_values * torch.ones(dim))
        self.gamma_2 = nn.Parameter(init_values * torch.ones(dim))
        self.drop_path_prob = float(drop_path)


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.patch_shape = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.patch_shape[0] * self.patch_shape[1]
        self.proj = nn.Module()
        self.proj.weight = nn.Parameter(torch.empty(embed_dim, in_chans, patch_size, patch_size))
        self.proj.bias = nn.Parameter(torch.zeros(embed_dim))
        nn.init.kaiming_uniform_(self.proj.weight, a=math.sqrt(5))


def _block_bias(vit, blk, H, N, ld, need_grad):
    """(dense, dense_t, tiles) of one block's relative-position bias, as _TrunkFn.forward builds them."""
    long_n = N > 256 and need_grad   # (long sequences: the dK/dV kernel reads the tiled transposed copy, no dense one)
    dense = Fx.relpos_gather(blk.attn.relative_position_bias_table, vit._index32, H, N, ld, transposed=not long_n)
    dense_t = tiles = None
    if not long_n:
        dense, dense_t = dense
    if 64 < N <= 224:   # the batch-walking ViT-shape kernels (csrc/attention_vit.hip) read accumulator-layout copies
        tiles = Fx.bias_tiles(dense, N, blk.attn.scale, fwd=True, bwd=_VIT_FUSED_BWD and need_grad)
    elif N > 256 and need_grad:   # 384 / 480 px: the long-sequence backward kernels (csrc/attention_long.hip) read both copies
        tiles = Fx.bias_tiles(dense, N, blk.attn.scale, fwd=True, bwd=True)
    return dense, dense_t, tiles


class _TrunkFn(torch.autograd.Function):
    """All transformer blocks + the final LayerNorm as one node.  x0: fp32 [B, N, D] -> bf16 [B, N, D]."""

    @staticmethod
    def forward(ctx, x0, vit, dp):
        B, N, D = x0.shape
        M, H = B * N, vit.num_heads
        arena = vit._arena
        x = x0.reshape(M, D).contiguous()
        ld = vit._bias_ld
        blocks = vit.blocks
        saved = []
        n0 = blocks[0].norm1
        y, mean, rstd = Fx.ln_fwd(x, n0.weight, n0.bias, n0.eps)
        ctx.first = (x, mean, rstd)
        rel_pos = getattr(vit, "_rel_pos", True)      # models/vit.py (plain ViT): no relative-position bias,
        final_norm = vit._final_norm if hasattr(vit, "_final_norm") else vit.fc_norm
        # the blocks' dense relative-position biases (table gather + accumulator-layout copies: two 5-us kernels per block) depend on the
        # weights alone: all 24 launches go to the second stream up front and run under the first block's GEMMs
        ahead = None
        if rel_pos and x.is_cuda and _RELPOS_AHEAD:
            from .xroberta import _WgradStream
            pre = _WgradStream(x.device)
            if pre.on:
                ahead = []
                with torch.cuda.stream(pre.side):
                    for blk in blocks:
                        ahead.append(_block_bias(vit, blk, H, N, ld, bool(ctx.needs_input_grad[0])))
                    ev_bias = torch.cuda.Event()
                    ev_bias.record(pre.side)
                for triple in ahead:
                    for t in triple[:2] + tuple(triple[2] or ()):
                        if t is not None:
                            t.record_stream(pre.main)
                pre.main.wait_event(ev_bias)
        for i, blk in enumerate(blocks):
            s = vit._slots[i]
            g1 = blk.gamma_1 if blk.gamma_1 is not None else vit._ones   # ... and no layer scale (gamma = 1, no gradient)
            g2 = blk.gamma_2 if blk.gamma_2 is not None else vit._ones
            dense = dense_t = tiles = None
            if ahead is not None:
                dense, dense_t, tiles = ahead[i]
            elif rel_pos:
                dense, dense_t, tiles = _block_bias(vit, blk, H, N, ld, bool(ctx.needs_input_grad[0]))
            qkv = Fx.gemm_nt(y, s["qkv"].wb, s["qkv"].b)
            # long sequences keep the low half of O: the backward then takes delta from dO . (O + O_lo) instead of a first pass over
            # the keys (two of the dQ kernel's five matrix products: 355 -> 224 us per layer at 901 tokens for 24 us of forward)
            if ctx.needs_input_grad[0] and (_FAST_DELTA or N > 256):
                ctxv, lse, ctxv_lo = Fx.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, H, N, N, blk.attn.scale, bias=dense, lo=True,
                                                 bias_tiles=tiles)
            else:
                (ctxv, lse), ctxv_lo = Fx.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, H, N, N, blk.attn.scale, bias=dense,
                                                   bias_tiles=tiles), None
            h1 = Fx.gemm_nt(ctxv, s["proj"].wb, s["proj"].b)
            dp1 = None if dp is None else dp[i, 0]
            dp2 = None if dp is None else dp[i, 1]
            x1, y2, mean2, rstd2 = Fx.ln_ls_fwd(x, h1, g1, dp1, N, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
            hact, u = Fx.gemm_nt(y2, s["fc1"].wb, s["fc1"].b, epi=Fx.EPI_GELU)
            h2 = Fx.gemm_nt(hact, s["fc2"].wb, s["fc2"].b)
            nxt = blocks[i + 1].norm1 if i + 1 < len(blocks) else final_norm
            x2, yn, meann, rstdn = Fx.ln_ls_fwd(x1, h2, g2, dp2, N, nxt.weight, nxt.bias, nxt.eps)
            saved.append((y, dense, qkv, ctxv, lse, h1, x1, mean2, rstd2, y2, u, hact, h2, x2, meann, rstdn, dp1, dp2, dense_t, ctxv_lo, tiles))
            x, y = x2, yn
        ctx.saved, ctx.vit, ctx.shape = saved, vit, (B, N, D)
        ctx.noted = bool(ctx.needs_input_grad[0])
        if ctx.noted:
            arena_note_use(vit)
        ctx.pooled = bool(getattr(vit, "_pool_tail", False))
        if ctx.pooled:  # beit2.py:455-466: the cls row leaves as the mean of the normalised patch rows (in place: y is not saved)
            Fx.pool_rows_fwd_(y, B, N)
        return y.view(B, N, D)

    @staticmethod
    def backward(ctx, dy_out):
        vit, (B, N, D) = ctx.vit, ctx.shape
        M, H = B * N, vit.num_heads
        blocks = vit.blocks
        ld = vit._bias_ld
        # two passes of one step (clean / MIM-masked view) may run on different streams; their weight gradients accumulate (+=) into the
        # same arena ranges, so a pass starts only once the previous one -- wherever it ran -- has finished
        prev = getattr(vit, "_trunk_bwd_done", None)
        if prev is not None and dy_out.is_cuda:
            torch.cuda.current_stream(dy_out.device).wait_event(prev)
        dy = dy_out.reshape(M, D).contiguous()
        if ctx.pooled:
            dy = Fx.pool_rows_bwd(dy if dy.dtype == torch.bfloat16 else dy.to(torch.bfloat16), B, N)
        dstream = torch.zeros((M, D), dtype=torch.float32, device=dy.device)
        from .xroberta import _WgradStream
        wg = _WgradStream(dy.device, "vit bwd")
        done = len(blocks)
        ddense_all = None
        # the blocks' weight gradients are QUEUED and run as one grouped launch at the end of the trunk (or before each hand-over of a
        # gradient range to the data-parallel accelerator below): 48 projections x 9-36 output tiles walk whole tiles over all of M
        queued = _DEFER_WGRAD and dy.is_cuda
        tn = wg.defer_tn if queued else wg.gemm_tn
        # ... and so are the column-sum folds of the two LayerNorm backward kernels of a block (dgamma / dbeta / bias / layer-scale gradients)
        rq = None
        if _DEFER_LN and dy.is_cuda:
            from . import _lib
            rq = Fx.ReduceQueue(dy.device, 2 * len(blocks) * ((_lib.load().xfm_layernorm_bwd_workspace(M, D, 2) + 255) // 256 * 256))   # LN_LS
            wg.defer_reduces(rq)
        for i in reversed(range(len(blocks))):
            blk, s = blocks[i], vit._slots[i]
            (y, dense, qkv, ctxv, lse, h1, x1, mean2, rstd2, y2, u, hact, h2, x2, meann, rstdn, dp1, dp2, dense_t, ctxv_lo, tiles) = ctx.saved[i]
            final_norm = vit._final_norm if hasattr(vit, "_final_norm") else vit.fc_norm
            nxt = blocks[i + 1].norm1 if i + 1 < len(blocks) else final_norm
            g = _g
            g1 = blk.gamma_1 if blk.gamma_1 is not None else vit._ones
            g2 = blk.gamma_2 if blk.gamma_2 is not None else vit._ones
            dg1 = g(blk.gamma_1) if blk.gamma_1 is not None else None
            dg2 = g(blk.gamma_2) if blk.gamma_2 is not None else None
            dh2 = Fx.ln_ls_bwd(dy, dstream, x2, meann, rstdn, nxt.weight, h2, g2, dp2, N, g(nxt.weight), g(nxt.bias),
                               s["fc2"].db, dg2, defer=rq)
            tn(dh2, hact, s["fc2"].dw)
            du = Fx.gemm_nt(dh2, s["fc2"].wt, epi=Fx.EPI_DGELU, aux=u, n=s["fc2"].K)
            tn(du, y2, s["fc1"].dw, dbias=s["fc1"].db)
            dy2 = Fx.gemm_nt(du, s["fc1"].wt, n=s["fc1"].K)
            dh1 = Fx.ln_ls_bwd(dy2, dstream, x1, mean2, rstd2, blk.norm2.weight, h1, g1, dp1, N, g(blk.norm2.weight),
                               g(blk.norm2.bias), s["proj"].db, dg1, defer=rq)
            tn(dh1, ctxv, s["proj"].dw)
            dctx = Fx.gemm_nt(dh1, s["proj"].wt, n=s["proj"].K)
            dqkv = torch.empty_like(qkv)
            ddense = None
            if dense is not None:  # one zero fill for the whole trunk's bias-gradient scratch instead of one per block
                if ddense_all is None:
                    ddense_all = torch.zeros((len(blocks),) + tuple(dense.shape), dtype=dense.dtype, device=dense.device)
                ddense = ddense_all[i]
            Fx.attn_bwd(dctx, qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], ctxv, lse, dqkv[:, :D], dqkv[:, D:2 * D],
                        dqkv[:, 2 * D:], B, H, N, N, blk.attn.scale, bias=dense, dbias=ddense, bias_t=dense_t, o_lo=ctxv_lo,
                        bias_tiles=tiles if (_VIT_FUSED_BWD or N > 256) else None)
            if dense is not None:   # the TABLE gradient from the dense bias gradient: a parameter gradient -- runs with the queued weight gradients
                G = blk.attn.window_size[0]
                dtab = g(blk.attn.relative_position_bias_table)
                if N > 256 and blk.attn.window_size[0] == blk.attn.window_size[1] and N == G * G + 1:
                    fn = lambda ddense=ddense, G=G, dtab=dtab: Fx.relpos_grid_grad(ddense, G, H, ld, dtab)   # noqa: E731  (144 -> ~20 us at 901 tokens)
                else:
                    fn = lambda ddense=ddense, dtab=dtab: Fx.relpos_scatter_sorted(ddense, vit._relpos_order, vit._relpos_start, H, N, ld, dtab)   # noqa: E731
                if queued:
                    wg.defer_call(fn, keep=(ddense_all,))
                else:
                    fn()
            tn(dqkv, y, s["qkv"].dw, dbias=s["qkv"].db)
            dy = Fx.gemm_nt(dqkv, s["qkv"].wt, n=s["qkv"].K)
            ctx.saved[i] = None
            if _WGRAD_GROUP_BLOCKS > 0 and i % _WGRAD_GROUP_BLOCKS == 0:
                wg.flush()
            # data parallel: the gradients of blocks [i + 1, done) are final now (block i's own norm1 gradient is only written while
            # block i - 1 is processed) once this pass is the tower's last pending use -- hand that arena range to the accelerator,
            # so its all-reduce runs under the remaining blocks' backward
            if ctx.noted and i > 0 and i % _GRAD_CHUNK_BLOCKS == 0 and i + 1 < done:
                hook = getattr(vit, "_block_grad_hook", None)
                # (the hook flushes the queued weight gradients itself once it knows the range really leaves now)
                if hook is not None and hook(vit, i + 1, done, wg):
                    done = i + 1
        x0, mean0, rstd0 = ctx.first
        n0 = blocks[0].norm1
        Fx.ln_bwd(dy, x0, mean0, rstd0, n0.weight, _g(n0.weight), _g(n0.bias), dx32=dstream, dx_accum=True)
        wg.join()
        if dstream.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dstream.device))
            vit._trunk_bwd_done = ev
        if ctx.noted:
            arena_note_grad(vit)
        return dstream.view(B, N, D), None, None


def region_outputs(full, idx_to_group_img, image_atts):
    """The region call form's per-sample output (beit2.py:467-475): sample i reads the normalised patch rows of image
    idx_to_group_img[i] and pools its own pseudo-cls as their image_atts-weighted mean (fp32 sums, one rounding to the tower's
    bf16).  A row gather and a [bs, P] x [bs, P, D] contraction on a handful of samples: device-side glue (differentiable ATen ops on
    the HIP tensors), no tower kernel involved."""
    idx = idx_to_group_img.to(device=full.device, dtype=torch.long).view(-1)
    x_bs = full[:, 1:, :].index_select(0, idx)
    w = image_atts.to(device=full.device)[:, 1:].to(torch.float32)
    assert w.shape == x_bs.shape[:2], "image_atts is [bs, 1 + patches]"
    cls = torch.einsum("bp,bpd->bd", w, x_bs.float()) / w.sum(dim=1, keepdim=True)
    return torch.cat([cls.to(full.dtype).unsqueeze(1), x_bs], dim=1)


class _TokensFn(torch.autograd.Function):
    """x0 = cat(cls, where(mask, mask_token, tok[b mod Bt])) in fp32; the cls / mask-token gradients are accumulated straight into
    the gradient arena."""

    @staticmethod
    def forward(ctx, tok, cls_token, mask_token, mask_u8, Bx):
        ctx.mask, ctx.Bt = mask_u8, tok.shape[0]
        ctx.cls_token, ctx.mask_token = cls_token, mask_token
        return Fx.vit_tokens_fwd(tok.contiguous(), cls_token.detach().reshape(-1), mask_token.detach().reshape(-1), mask_u8, Bx)

    @staticmethod
    def backward(ctx, dx0):
        dmask = _g(ctx.mask_token).view(-1) if ctx.mask is not None else None
        dtok = Fx.vit_tokens_bwd(dx0.contiguous(), ctx.mask, ctx.Bt, _g(ctx.cls_token).view(-1), dmask)
        return dtok, None, None, None, None


# XFM_ATTN_FAST_DELTA=1: the attention backward takes delta = dO . O from the forward's output (kept as bf16 hi + lo halves) instead of a
# first pass over the keys: dQ kernel 198 -> 165 us, forward 69 -> 80 us at B = 128 (tools/bench_attn.py), the step unchanged within
# noise (41.6 vs 41.5 ms, tools/ab.sh) -- off by default, the exact two-pass form costs nothing
_FAST_DELTA = __import__("os").environ.get("XFM_ATTN_FAST_DELTA", "0") != "0"
_VIT_FUSED_BWD = __import__("os").environ.get("XFM_ATTN_VIT_BWD", "0") != "0"   # opt-in single-pass attention backward (measured slower than the split pair: csrc/attention_vit.hip)
# the trunk's gradients leave for the all-reduce in chunks of this many blocks, from inside its backward; what is still there when the
# backward ends (blocks 0 .. chunk, the embeddings) is the exposed tail of the exchange: 2 -> 3 blocks = 85 MB of fp32 gradients behind
# five overlapped 57-MB collectives (round 2: 4 -> 5 blocks = 142 MB behind two of 114 MB)
# XFM_VIT_DEFER_WGRAD=0: one xfm_gemm_tn per projection as the backward reaches it (rounds 1-3) instead of the grouped launch
_DEFER_WGRAD = __import__("os").environ.get("XFM_VIT_DEFER_WGRAD", "1") != "0"
_RELPOS_AHEAD = __import__("os").environ.get("XFM_VIT_RELPOS_AHEAD", "1") != "0"   # A/B knob: the blocks' dense biases built up front on the second stream
_DEFER_LN = __import__("os").environ.get("XFM_VIT_DEFER_LN", "1") != "0"   # A/B knob: one batched LayerNorm column-sum reduce for the trunk
_WGRAD_GROUP_BLOCKS = int(__import__("os").environ.get("XFM_VIT_WGRAD_GROUP", "0"))   # blocks per grouped launch (0: the whole trunk at its end)
# (round 4: every hand-over also launches the chunk's queued weight gradients as one grouped call -- 0.43 ms per block in chunks of two,
# 0.37 in chunks of four, 0.33 for the whole trunk at once -- so chunks of four again: 0.6 ms less weight-gradient time than chunks of
# two for ~0.25 ms more exposed exchange (blocks 0-4 instead of 0-2 after the backward))
_GRAD_CHUNK_BLOCKS = max(1, int(__import__("os").environ.get("XFM_VIT_GRAD_CHUNK", "4")))


def _g(p):
    """gradient view of a parameter inside the arena (marks it live; re-attached if a caller dropped .grad)."""
    from .arena import grad_of
    return grad_of(p)


def arena_note_use(mod):
    hook = getattr(mod, "_use_hook", None)
    if hook is not None:
        hook(mod, +1)


def arena_note_grad(mod):
    hook = getattr(mod, "_use_hook", None)
    if hook is not None:
        hook(mod, -1)


class VisionTransformer(OwnsArena, nn.Module):
    """Drop-in for models.beit2.VisionTransformer (BEiT-v2 configuration used by XFM)."""
    _pool_tail = True  # forward_avgpool (beit2.py:455-466): the trunk node replaces the cls row by the mean of the patch rows

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.,
                 norm_layer=None, init_values=0.1, use_abs_pos_emb=False, use_rel_pos_bias=True,
                 use_shared_rel_pos_bias=False, use_mean_pooling=True, init_scale=0.001, local_attn_depth=-1,
                 num_masking_patches=75, min_num_patches=16, layer_norm_eps=1e-6):
        super().__init__()
        unsupported = []
        if use_abs_pos_emb: unsupported.append("use_abs_pos_emb")
        if not use_rel_pos_bias: unsupported.append("use_rel_pos_bias=False")
        if use_shared_rel_pos_bias: unsupported.append("use_shared_rel_pos_bias")
        if not use_mean_pooling: unsupported.append("use_mean_pooling=False")
        if local_attn_depth > 0: unsupported.append("local_attn_depth>0")
        if not qkv_bias: unsupported.append("qkv_bias=False")
        if not init_values or init_values <= 0: unsupported.append("init_values<=0")
        if drop_rate or attn_drop_rate: unsupported.append("drop_rate/attn_drop_rate")
        if (embed_dim // num_heads) != 64: unsupported.append("head_dim != 64")
        if unsupported:
            raise NotImplementedError("xfm_amd.beit2 implements the BEiT-v2 configuration XFM ships "
                                      f"(xfm.py:206-234); unsupported: {unsupported}")
        self.local_attn_depth = local_attn_depth
        self.depth, self.num_heads = depth, num_heads
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.mask_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = None
        self.generator = BlockMaskGenerator(img_size // patch_size, num_masking_patches, min_num_patches)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth, device="cpu")]
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, init_values, self.patch_embed.patch_shape,
                                           dpr[i], layer_norm_eps) for i in range(depth)])
        self.fc_norm = _Affine(embed_dim, layer_norm_eps)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        nn.init.trunc_normal_(self.mask_token, std=.02)
        for i, blk in enumerate(self.blocks):  # fix_init_weight, beit2.py:327-333
            blk.attn.proj.weight.data.div_(math.sqrt(2.0 * (i + 1)))
            blk.mlp.fc2.weight.data.div_(math.sqrt(2.0 * (i + 1)))
        self._arena = None
        self._own_arena = False

    # ---- arena plumbing -----------------------------------------------------------------------
    def linear_slots(self):
        D = self.embed_dim
        self._slots = []
        out = []
        self._slot_patch = LinearSlot("patch_embed", [self.patch_embed.proj.weight], [self.patch_embed.proj.bias])
        self._slot_patch.need_t = False
        out.append(self._slot_patch)
        for i, blk in enumerate(self.blocks):
            s = {"qkv": LinearSlot(f"blocks.{i}.qkv", [blk.attn.qkv.weight], [blk.attn.q_bias, D, blk.attn.v_bias]),
                 "proj": LinearSlot(f"blocks.{i}.proj", [blk.attn.proj.weight], [blk.attn.proj.bias]),
                 "fc1": LinearSlot(f"blocks.{i}.fc1", [blk.mlp.fc1.weight], [blk.mlp.fc1.bias]),
                 "fc2": LinearSlot(f"blocks.{i}.fc2", [blk.mlp.fc2.weight], [blk.mlp.fc2.bias])}
            self._slots.append(s)
            out.extend(s.values())
        return out

    def attach(self, arena):
        self._arena = arena
        N = self.patch_embed.num_patches + 1
        self._bias_ld = (N + 15) // 16 * 16
        self._index32 = self.blocks[0].attn.relative_position_index.to(device=arena.device, dtype=torch.int32).contiguous()
        self._relpos_order, self._relpos_start = Fx.relpos_sorted_index(
            self._index32, self.blocks[0].attn.relative_position_bias_table.shape[0])

    def finalize(self, device=None):
        """Stand-alone use (the XFM wrapper builds one arena for all towers instead)."""
        device = device or self.cls_token.device
        arena = ParamArena(self, self.linear_slots(), device)
        self.attach(arena)
        self._own_arena = True
        return self

    def _ready(self):
        if self._arena is None or not self._arena.attached():
            if self._arena is not None and not self._own_arena:
                raise RuntimeError("parameters were moved after the arena was built; call finalize() again")
            self.finalize()

    # ---- forward --------------------------------------------------------------------------------
    def forward(self, x, idx_to_group_img=None, image_atts=None, do_mask=False, ids_mask=None, drop_path_scales=None, split_stream=None):
        """`split_stream` (extension; with `ids_mask` holding k = 2 views of the B images): the views share the patch embedding and
        the token assembly, but each runs the trunk as its own pass -- view 0 on the current stream, view 1 on `split_stream` -- and
        the result is the pair (out0, out1) instead of one [2B, N, D] tensor; out1 lives on `split_stream` (the caller joins).
        Autograd replays each pass's backward on its forward stream."""
        if (idx_to_group_img is None) != (image_atts is None):
            raise ValueError("the region call form takes idx_to_group_img AND image_atts (beit2.py:467-475)")
        if idx_to_group_img is not None and (do_mask or split_stream is not None):
            raise ValueError("the region call form has no masked view (beit2.py:467-475)")
        self._ready()
        B = x.shape[0]
        D = self.embed_dim
        P = self.patch_embed.patch_size[0]
        patches = Fx.patchify(x.float().contiguous(), P)
        tok = linear_slot(patches, self._slot_patch, x_requires_grad=False, out_fp32=True).view(B, -1, D)
        mask_u8 = None
        if do_mask:
            if ids_mask is None:
                ids_mask = self.generator.batch(B, x.device)
            ids_mask = ids_mask.to(device=x.device, dtype=torch.bool).contiguous()
            # several masked views of the SAME images in one pass: ids_mask [k * B, P] against B images (the pre-training step's
            # clean + MIM-masked pair) -- patch gather, patch-embed GEMM and its weight gradient run once per image
            assert ids_mask.shape[0] % B == 0, "ids_mask rows must be a multiple of the image batch"
            mask_u8 = ids_mask.view(torch.uint8)
            B = ids_mask.shape[0]
        # mask-token mix + cls concat (beit2.py:432-446) as one kernel, forward and backward
        x0 = _TokensFn.apply(tok, self.cls_token, self.mask_token, mask_u8, B)
        dp = drop_path_scales
        if dp is None and self.training:
            keep = getattr(self, "_dp_keep", None)
            if keep is None or keep.device != x.device:  # built once: a per-forward torch.tensor(..., device=cuda) would sync
                keep = 1.0 - torch.tensor([[b.drop_path_prob] * 2 for b in self.blocks], device=x.device).view(-1, 2, 1)
                self._dp_keep = keep
            dp = (torch.rand(len(self.blocks), 2, B, device=x.device) < keep).float() / keep
        if split_stream is not None and do_mask and B == 2 * x.shape[0]:
            Bh = x.shape[0]
            main = torch.cuda.current_stream(x.device)
            out0 = _TrunkFn.apply(x0[:Bh], self, None if dp is None else dp[:, :, :Bh].contiguous())
            split_stream.wait_stream(main)   # (x0 and the drop-path draws are ready; the first view's kernels are merely queued ahead)
            with torch.cuda.stream(split_stream):
                out1 = _TrunkFn.apply(x0[Bh:], self, None if dp is None else dp[:, :, Bh:].contiguous())
            x0.record_stream(split_stream)
            return (out0, out1), ids_mask
        out = _TrunkFn.apply(x0, self, dp)           # bf16 [B, N, D]: fc_norm on every row, then row 0 <- mean of the patch rows
        if idx_to_group_img is not None:
            return region_outputs(out, idx_to_group_img, image_atts), out
        return (out, ids_mask) if do_mask else out


def interpolate_rel_pos_bias(rel_pos_bias, dst_num_pos, dst_patch_shape):
    """Relative-position table [src_num_pos, heads] -> [dst_num_pos, heads] for another patch grid (fine-tuning at 384 / 480 px),
    beit2.py:763-821: the (2s-1)^2 grid part is resampled from geometric-progression source coordinates to the integer target
    coordinates with a bicubic interpolating spline, the 3 extra (cls) entries are carried over.

    The reference calls `scipy.interpolate.interp2d(x, y, z, kind='cubic')`, removed in SciPy 1.14 (this image has 1.15); for a regular
    grid that was FITPACK's interpolating tensor-product spline (`regrid_smth` with s = 0, evaluated by `bispev`), which
    `RectBivariateSpline(kx=3, ky=3, s=0)` reaches through the same routines.  Pinned by tests/golden/relpos_interp.npz: the
    reference's own `interpolate_pos_embed` run on a formula table (224 -> 384 / 480 px) with the removed call supplied by a
    from-source restatement of it (the CPU suite holds this function to that fixture at 1e-6)."""
    import numpy as np
    from scipy.interpolate import RectBivariateSpline
    src_num_pos, num_attn_heads = rel_pos_bias.shape
    if dst_patch_shape[0] != dst_patch_shape[1]:
        raise NotImplementedError()
    num_extra_tokens = dst_num_pos - (dst_patch_shape[0] * 2 - 1) * (dst_patch_shape[1] * 2 - 1)
    src_size = int((src_num_pos - num_extra_tokens) ** 0.5)
    dst_size = int((dst_num_pos - num_extra_tokens) ** 0.5)
    if src_size == dst_size:
        return rel_pos_bias
    extra_tokens = rel_pos_bias[-num_extra_tokens:, :]
    grid = rel_pos_bias[:-num_extra_tokens, :]
    x = rel_pos_source_coordinates(src_size, dst_size)
    t = dst_size // 2.0
    dx = np.arange(-t, t + 0.1, 1.0)
    out = []
    for i in range(num_attn_heads):
        z = grid[:, i].view(src_size, src_size).float().numpy().astype(np.float64)  # z[j, i] sits at (x[i], y[j]), interp2d's convention
        f = RectBivariateSpline(x, x, z, kx=3, ky=3, s=0)                            # first axis = y: f(dy, dx)[j, i]
        out.append(torch.tensor(f(dx, dx), dtype=torch.float32).contiguous().view(-1, 1).to(rel_pos_bias.device))
    return torch.cat((torch.cat(out, dim=-1), extra_tokens), dim=0)


def rel_pos_source_coordinates(src_size, dst_size):
    """beit2.py:780-803: source relative offsets placed on a geometric progression whose half-extent matches the target's."""
    def geometric_progression(a, r, n):
        return a * (1.0 - r ** n) / (1.0 - r)

    left, right = 1.01, 1.5
    while right - left > 1e-6:
        q = (left + right) / 2.0
        gp = geometric_progression(1, q, src_size // 2)
        if gp > dst_size // 2:
            right = q
        else:
            left = q
    dis, cur = [], 1
    for i in range(src_size // 2):
        dis.append(cur)
        cur += q ** (i + 1)
    return [-d for d in reversed(dis)] + [0] + dis


def load_pretrained_beit2(model, ckpt_rpath):
    """beit2.py:572-660: unwrap `model` / `module`, drop the classification head and the `relative_position_index` buffers, expand
    a shared relative-position table to every block, resample tables of another grid size (`interpolate_rel_pos_bias`), load with
    strict=False."""
    checkpoint = torch.load(ckpt_rpath, map_location='cpu')
    checkpoint_model = None
    for model_key in ('model', 'module'):
        if model_key in checkpoint:
            checkpoint_model = checkpoint[model_key]
            break
    if checkpoint_model is None:
        checkpoint_model = checkpoint
    for k in ('head.weight', 'head.bias'):
        checkpoint_model.pop(k, None)
    if "rel_pos_bias.relative_position_bias_table" in checkpoint_model:
        shared = checkpoint_model.pop("rel_pos_bias.relative_position_bias_table")
        for i in range(len(model.blocks)):
            checkpoint_model["blocks.%d.attn.relative_position_bias_table" % i] = shared.clone()
    own = model.state_dict()
    for key in list(checkpoint_model.keys()):
        if "relative_position_index" in key:
            checkpoint_model.pop(key)
        elif "relative_position_bias_table" in key and key in own and own[key].shape != checkpoint_model[key].shape:
            checkpoint_model[key] = interpolate_rel_pos_bias(checkpoint_model[key], own[key].shape[0], model.patch_embed.patch_shape)
    msg = model.load_state_dict(checkpoint_model, strict=False)
    if getattr(model, "_arena", None) is not None:
        model._arena.bump()
    return msg


def beit_base_patch16(img_size, depth=12, **kwargs):
    return VisionTransformer(img_size=img_size, patch_size=16, embed_dim=768, depth=depth, num_heads=12, mlp_ratio=4,
                             layer_norm_eps=1e-6, **kwargs)
