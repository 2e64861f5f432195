"""RoBERTa text / fusion towers on the HIP hot path, behind the reference's interface.

Mirrors `models/xroberta.py` of the reference: RobertaModel.forward (:817-957, incl. the `encoder_embeds` bypass
:920-929 and `mode` layer ranges :504-516), RobertaForMaskedLM (:1157-1310, `.bert()` method, `masked_pos` gather,
two LM heads), RobertaForCausalLM-style decoding mask (`is_decoder`), and the state_dict key names.  The layer stack
runs as ONE autograd node: per layer a fused QKV GEMM, LDS-resident attention, output GEMM, fused
dropout+residual+LayerNorm, (cross-attention with a fused K/V GEMM over the image tokens), GELU-fused FFN.
`scale_after_qk=True` selects the xbert.py:329-330 ordering (same kernels: the scale is applied to the fp32 scores).
"""
import json
import math
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import functional as Fx
from .arena import LinearSlot, OwnsArena, ParamArena
from .beit2 import _Affine, arena_note_grad, arena_note_use
from .ops import grad_view, lm_head_ce, lm_head_logits

BF16, F32 = torch.bfloat16, torch.float32


class RobertaConfig:
    """The subset of transformers' RobertaConfig the path reads (xfm.py:273-284,482-486,528-531)."""

    def __init__(self, **kw):
        d = dict(vocab_size=50265, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                 hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, max_position_embeddings=514,
                 type_vocab_size=1, initializer_range=0.02, layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0,
                 eos_token_id=2, fusion_layer=12, encoder_width=768, add_cross_attention=False)
        d.update(kw)
        self.__dict__.update(d)

    @classmethod
    def from_json_file(cls, path):
        with open(path) as f:
            return cls(**json.load(f))


_seed_counter = [0]


def _next_seed():
    _seed_counter[0] += 1
    return ((torch.initial_seed() & 0xFFFFFFFF) << 32) | (_seed_counter[0] & 0xFFFFFFFF)


class _Lin(nn.Module):
    def __init__(self, fin, fout, std):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin).normal_(0.0, std))
        self.bias = nn.Parameter(torch.zeros(fout))


class _Emb(nn.Module):
    def __init__(self, n, dim, std, padding_idx=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, dim).normal_(0.0, std))
        self.padding_idx = padding_idx
        if padding_idx is not None:
            with torch.no_grad():
                self.weight[padding_idx].zero_()


class RobertaEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        std = config.initializer_range
        self.word_embeddings = _Emb(config.vocab_size, config.hidden_size, std, config.pad_token_id)
        self.position_embeddings = _Emb(config.max_position_embeddings, config.hidden_size, std, config.pad_token_id)
        self.token_type_embeddings = _Emb(config.type_vocab_size, config.hidden_size, std)
        self.LayerNorm = _Affine(config.hidden_size, config.layer_norm_eps)
        self.register_buffer("position_ids", torch.arange(config.max_position_embeddings).expand((1, -1)))
        self.padding_idx = config.pad_token_id
        self.p_drop = config.hidden_dropout_prob


_F32_STREAM = os.environ.get("XFM_F32_STREAM", "1") != "0"
"""The residual stream of the text / fusion towers in fp32 (A/B knob; 0 = the bf16 stream of rounds 1-4).  The reference trains in mixed
precision: its Linears produce 16-bit outputs, but LayerNorm and the residual add run in fp32 (apex O1 / autocast lists), so every
post-LN sum `LayerNorm(dropout(h) + x)` (xroberta.py:300-304, 381-385) sees the UN-ROUNDED output of the previous LayerNorm and the
gradient of the residual branch travels in fp32 too.  With the stream on, every tower hand-over is a (bf16, fp32) pair: the bf16 tensor
feeds the GEMMs, its fp32 twin the next residual add; the two are separate autograd edges (bf16 gradient from the GEMM side, fp32 from
the LayerNorm side)."""


def twin_of(t):
    """The fp32 twin of a tower output (None when there is none)."""
    return getattr(t, "_xfm_f32", None) if _F32_STREAM else None


def with_twin(t, t32):
    """Attach the fp32 twin to a tower output / to a row-wise image of one (cat / index_select / gather of both)."""
    if t32 is not None and _F32_STREAM:
        t._xfm_f32 = t32
    return t


def rowwise(fn, *tensors):
    """fn applied to tower outputs AND, when every one of them has one, to their fp32 twins: for the row-wise glue between towers
    (torch.cat / index_select / detach / row gathers), so the twin follows the rows into the next tower."""
    out = fn(*tensors)
    twins = [twin_of(t) for t in tensors]
    if all(tw is not None for tw in twins):
        with_twin(out, fn(*twins))
    return out


class _EmbedFn(torch.autograd.Function):
    """-> (y bf16, y fp32 twin or None).  The twin exists when the fp32 stream is on; its gradient (fp32) is added to y's."""

    @staticmethod
    def forward(ctx, anchor, emb, input_ids, drop, owner, pack=None):
        ids = input_ids.contiguous()
        ctx.owner = owner if ctx.needs_input_grad[0] else None
        if ctx.owner is not None:
            arena_note_use(owner)
        ctx.set_materialize_grads(False)
        ln = emb.LayerNorm
        row_map = None if pack is None else pack.row_map()  # packed (unpadded) output rows, xfm_amd.packing
        out = Fx.embed_ln_fwd(ids, emb.word_embeddings.weight, emb.position_embeddings.weight,
                              emb.token_type_embeddings.weight, ln.weight, ln.bias, ln.eps,
                              emb.padding_idx, drop, getattr(emb, "pos_mode", 0), row_map=row_map,
                              out_rows=None if pack is None else pack.cap, y32=_F32_STREAM)
        y, mean, rstd, pos_ids = out[:4]
        y32 = out[4] if _F32_STREAM else None
        ctx.emb, ctx.saved, ctx.drop, ctx.row_map = emb, (ids, mean, rstd, pos_ids), drop, row_map
        if pack is None:
            y = y.view(ids.shape[0], ids.shape[1], -1)
            y32 = None if y32 is None else y32.view(ids.shape[0], ids.shape[1], -1)
        return y, y32

    @staticmethod
    def backward(ctx, dy, dy32=None):
        emb = ctx.emb
        ids, mean, rstd, pos_ids = ctx.saved
        ln = emb.LayerNorm
        if dy is None and dy32 is None:
            return None, None, None, None, None, None
        if dy is None:
            dy = torch.zeros(dy32.shape, dtype=BF16, device=dy32.device)
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        if dy2.dtype != BF16:
            dy2 = dy2.to(BF16)
        if dy32 is not None:
            dy32 = dy32.reshape(-1, dy32.shape[-1]).contiguous()
        Fx.embed_ln_bwd(dy2, ids, emb.word_embeddings.weight, emb.position_embeddings.weight,
                        emb.token_type_embeddings.weight, ln.weight, ln.bias, ln.eps, emb.padding_idx, mean, rstd, pos_ids,
                        grad_view(emb.word_embeddings.weight), grad_view(emb.position_embeddings.weight),
                        grad_view(emb.token_type_embeddings.weight).view(-1), grad_view(ln.weight), grad_view(ln.bias), ctx.drop,
                        getattr(emb, "pos_mode", 0), row_map=ctx.row_map, dy32=dy32)
        if ctx.owner is not None:
            arena_note_grad(ctx.owner)
        return None, None, None, None, None, None


class RobertaSelfAttention(nn.Module):
    def __init__(self, config, is_cross_attention):
        super().__init__()
        std = config.initializer_range
        kv_in = config.encoder_width if is_cross_attention else config.hidden_size
        self.query = _Lin(config.hidden_size, config.hidden_size, std)
        self.key = _Lin(kv_in, config.hidden_size, std)
        self.value = _Lin(kv_in, config.hidden_size, std)


class RobertaSelfOutput(nn.Module):
    def __init__(self, config, fin=None):
        super().__init__()
        self.dense = _Lin(fin or config.hidden_size, config.hidden_size, config.initializer_range)
        self.LayerNorm = _Affine(config.hidden_size, config.layer_norm_eps)


class RobertaAttention(nn.Module):
    def __init__(self, config, is_cross_attention=False):
        super().__init__()
        self.self = RobertaSelfAttention(config, is_cross_attention)
        self.output = RobertaSelfOutput(config)


class RobertaIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = _Lin(config.hidden_size, config.intermediate_size, config.initializer_range)


class RobertaLayer(nn.Module):
    def __init__(self, config, layer_num):
        super().__init__()
        self.attention = RobertaAttention(config)
        self.has_cross_attention = layer_num >= config.fusion_layer
        if self.has_cross_attention:
            self.layer_num = layer_num
            self.crossattention = RobertaAttention(config, is_cross_attention=True)
        self.intermediate = RobertaIntermediate(config)
        self.output = RobertaSelfOutput(config, fin=config.intermediate_size)

    def linear_slots(self, prefix):
        a = self.attention
        s = {"qkv": LinearSlot(prefix + "qkv", [a.self.query.weight, a.self.key.weight, a.self.value.weight],
                               [a.self.query.bias, a.self.key.bias, a.self.value.bias]),
             "o": LinearSlot(prefix + "o", [a.output.dense.weight], [a.output.dense.bias]),
             "i": LinearSlot(prefix + "i", [self.intermediate.dense.weight], [self.intermediate.dense.bias]),
             "out": LinearSlot(prefix + "out", [self.output.dense.weight], [self.output.dense.bias])}
        if self.has_cross_attention:
            c = self.crossattention
            s["q2"] = LinearSlot(prefix + "q2", [c.self.query.weight], [c.self.query.bias])
            s["kv2"] = LinearSlot(prefix + "kv2", [c.self.key.weight, c.self.value.weight], [c.self.key.bias, c.self.value.bias])
            s["o2"] = LinearSlot(prefix + "o2", [c.output.dense.weight], [c.output.dense.bias])
        self._s = s
        self._rlp = None  # (cached native-call pointers belong to the previous arena)
        return list(s.values())


class RobertaEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.layer = nn.ModuleList([RobertaLayer(config, i) for i in range(config.num_hidden_layers)])


from . import marks as _marks  # noqa: E402


class _DeferredCall:
    """One-shot work of a _WgradStream flush."""
    persistent = False

    def __init__(self, fn, keep=()):
        self.fn, self.keep = fn, tuple(keep)

    def run(self):
        self.fn()


class _WgradStream:
    """Weight-gradient GEMMs of a tower's backward on a second HIP stream.  dW only feeds the optimizer / all-reduce, so the
    dY^T X products need not sit on the activation-gradient critical path: at the fusion / text towers' sizes (M = 7680 / 1920 rows)
    neither the dgrad nor the wgrad kernels fill the 256 CUs on their own, and the two chains overlap.  Every launch waits for an
    event recorded on the main stream after its dY was produced; the tensors it reads are kept alive until the main stream has
    re-joined (the caching allocator reuses freed blocks in main-stream order only)."""
    _streams = {}
    enabled = os.environ.get("XFM_WGRAD_STREAM", "1") != "0"
    priority = int(os.environ.get("XFM_WGRAD_PRIO", "0"))  # HIP stream priority of the second stream (lower number = served first)

    def __init__(self, device, label=None):
        self.label = label
        if label is not None:
            _marks.mark(label + " begin")
        self.main = torch.cuda.current_stream(device)
        if label is not None:   # a tower's backward starts: its gradient writes must follow a pending asynchronous zero_grad (arena.grads_ready)
            from .arena import grads_ready
            grads_ready(device)
        self.on = _WgradStream.enabled
        if self.on:
            # one side stream per LAUNCH stream: the text tower's backward runs on its own stream next to the ViT's, and its small
            # weight gradients must not queue (FIFO) behind the ViT's grouped launch on a shared side stream
            key = (device.index if device.index is not None else torch.cuda.current_device(), self.main.cuda_stream)
            if key not in _WgradStream._streams:
                _WgradStream._streams[key] = torch.cuda.Stream(device=device, priority=_WgradStream.priority)
            self.side = _WgradStream._streams[key]
            self.side.wait_stream(self.main)  # arena state (zeroed grads, earlier kernels) is visible to the side stream
        self.keep = []
        self.queue = []
        self.reduces = []

    def defer_tn(self, dy, x, dw, dbias=None):
        """Queue dw += dy^T @ x (dbias += column sums of dy) for the next flush(): everything queued -- all over the same M rows --
        runs as ONE grouped launch (Fx.gemm_tn_group: whole 256 x 256 tiles with one owner each instead of 7 M-split planes and a
        reduce per projection; the 48 weight gradients of the ViT trunk: 5.2 -> 3.9 ms).  The queue holds dy / x alive."""
        self.queue.append((dy, x, dw, dbias))

    def defer_call(self, fn, keep=()):
        """fn() runs with the next flush() (second stream, after everything the launch stream holds then): parameter-gradient kernels
        that nothing on the activation-gradient chain waits for (the relative-position table gradients of the ViT blocks, the batched
        LayerNorm folds of the layer executor).  `keep`: the tensors it touches."""
        self.reduces.append(_DeferredCall(fn, keep))

    def defer_reduces(self, rq):
        """A Fx.ReduceQueue: whatever LayerNorm backward kernels have queued on it is folded by every flush() until the join."""
        self.reduces.append(rq)

    def _run_reduces(self, reduces):
        for work in reduces:
            work.run()

    def flush(self):
        reduces = self.reduces
        self.reduces = [r for r in reduces if getattr(r, "persistent", False)]   # a ReduceQueue stays registered (more kernels will append to it)
        if reduces and not self.queue:
            if not self.on:
                return self._run_reduces(reduces)
            ev = torch.cuda.Event()
            ev.record(self.main)
            self.side.wait_event(ev)
            self.keep.append(reduces)
            with torch.cuda.stream(self.side):
                self._run_reduces(reduces)
            return
        if not self.queue:
            return
        items, self.queue = self.queue, []
        by_m = {}
        for it in items:   # one grouped launch per distinct row count (a tower's projections share M; the K/V projections of the image states
            by_m.setdefault(it[0].shape[0], []).append(it)   # and the pruned last layer have their own)
        if not self.on:
            self._run_reduces(reduces)
            for group in by_m.values():
                Fx.gemm_tn_group(group)
            return
        ev = torch.cuda.Event()
        ev.record(self.main)
        self.side.wait_event(ev)
        self.keep.append((items, reduces))
        with torch.cuda.stream(self.side):
            self._run_reduces(reduces)
            for group in by_m.values():
                Fx.gemm_tn_group(group)

    def gemm_tn(self, dy, x, dw, **kw):
        if not self.on:
            return Fx.gemm_tn(dy, x, dw, **kw)
        ev = torch.cuda.Event()
        ev.record(self.main)
        self.side.wait_event(ev)
        self.keep.append((dy, x))
        with torch.cuda.stream(self.side):
            Fx.gemm_tn(dy, x, dw, **kw)

    def project(self, x, slot):
        """x @ slot.W^T + b on the side stream; returns (tensor, event) for take()."""
        if not self.on:
            return Fx.gemm_nt(x, slot.wb, slot.b), None
        with torch.cuda.stream(self.side):
            y = Fx.gemm_nt(x, slot.wb, slot.b)
            ev = torch.cuda.Event()
            ev.record(self.side)
        y.record_stream(self.main)  # allocated on the side stream, consumed on the main one
        return y, ev

    def take(self, pair):
        y, ev = pair
        if ev is not None:
            self.main.wait_event(ev)
        return y

    def run(self, fn, keep=()):
        """fn() on the side stream, after everything enqueued on the main stream so far; `keep` = the tensors it touches."""
        if not self.on:
            return fn()
        ev = torch.cuda.Event()
        ev.record(self.main)
        self.side.wait_event(ev)
        self.keep.append(tuple(keep))
        with torch.cuda.stream(self.side):
            return fn()

    def sync_side(self):
        """The launch stream waits for what the second stream has been given so far (queued weight gradients are not launched)."""
        if self.on:
            self.main.wait_stream(self.side)

    def join_at_end(self):
        """Like join(), but the launch stream re-joins when the whole backward pass is over (an engine callback, as
        model_pretrain._JoinAfterBackward): the weight gradients given to the second stream here need not finish before the NEXT node's
        activation gradients start -- only before anyone reads the parameter gradients.  Only inside a backward pass."""
        self.flush()
        if self.on:
            main, side, keep = self.main, self.side, self.keep

            def rejoin():
                main.wait_stream(side)
                # the callback runs on the thread (and current stream) that called backward(): if this node ran on another stream (the
                # text tower's, replayed by autograd), that caller -- the optimizer, the gradient norm, the final all-reduce -- must wait
                # for these weight gradients too, whatever order the end-of-backward callbacks run in
                cur = torch.cuda.current_stream(main.device)
                if cur != main:
                    cur.wait_stream(side)
                keep.clear()
            torch.autograd.Variable._execution_engine.queue_callback(rejoin)
            self.keep = []
        else:
            self.keep.clear()

    def join(self):
        if self.label is not None:
            _marks.mark(self.label + " chain end")
        self.flush()
        if self.on:
            self.main.wait_stream(self.side)
        self.keep.clear()
        if self.label is not None:
            _marks.mark(self.label + " end")


def _grad_pair(dy, dy32, f32):
    """The two gradient edges of a tower output (bf16 tensor, fp32 twin) as the (dy_a bf16, dy_b) pair the layer loop starts from:
    dy_b is the fp32 part with the fp32 stream on (None when nobody read the twin), None otherwise."""
    if dy is None and dy32 is None:
        raise RuntimeError("encoder backward without a gradient")
    if dy is None:
        dy = torch.zeros(dy32.shape, dtype=BF16, device=dy32.device)
    dy_a = dy.contiguous()
    if dy_a.dtype != BF16:
        dy_a = dy_a.to(BF16)
    if dy32 is None:
        return dy_a, None
    if f32:
        return dy_a, dy32.contiguous()
    return (dy_a.float() + dy32).to(BF16), None


def _input_grads(dy_a, dy_b, need_dx, twin_in, rows_full):
    """Gradient w.r.t. the tower input from the last layer's (dprev bf16, residual-branch gradient): with an fp32 twin on the input the
    two leave on their own edges, un-rounded; otherwise they are summed and rounded once to the input's bf16."""
    if not need_dx:
        return None, None
    if twin_in and dy_b is not None and dy_b.dtype == F32:
        dx, dx32 = dy_a, dy_b
    else:
        dx, dx32 = (dy_a.float() + dy_b.float()).to(BF16), None
    pad = rows_full - dx.shape[0]
    if pad > 0:   # (a backward over the first sequences of the batch: the other rows took no gradient)
        dx = torch.cat([dx, torch.zeros((pad, dx.shape[1]), dtype=dx.dtype, device=dx.device)], dim=0)
        if dx32 is not None:
            dx32 = torch.cat([dx32, torch.zeros((pad, dx32.shape[1]), dtype=F32, device=dx32.device)], dim=0)
    return dx, dx32


class _EncoderFn(torch.autograd.Function):
    """Layers [lo, hi) of a RobertaEncoder.  x: bf16 [B*T, D]; x32: its fp32 twin or None; enc: bf16 [B*N, D] or None.
    -> (y bf16, y fp32 twin or None): see _F32_STREAM."""

    @staticmethod
    def forward(ctx, x, x32, enc, model, key_keep, enc_keep, lo, hi, causal, B, T, Nenc, training, enc_index, grad_batch=None, pack=None, xq=None):
        cfg = model.config
        f32 = _F32_STREAM
        ctx.set_materialize_grads(False)
        D, H = cfg.hidden_size, cfg.num_attention_heads
        scale = 1.0 / math.sqrt(D // H)
        p_att = cfg.attention_probs_dropout_prob if training else 0.0
        p_hid = cfg.hidden_dropout_prob if training else 0.0
        need_dx, need_denc = x.requires_grad or (x32 is not None and x32.requires_grad), (enc is not None and enc.requires_grad)
        saved = []
        x = x.contiguous()
        xres = x32.contiguous() if (f32 and x32 is not None) else x   # what the first residual add reads
        ctx.twin_in = f32 and x32 is not None
        qp = None if pack is None else pack.pair  # packed token rows: x is [pack.cap, D], sequences at (start, len)
        zf = pack is not None and not pack.exact  # slack rows exist: attention outputs must be zero there
        if pack is not None and (key_keep is not None or causal):
            raise ValueError("packed rows take neither a key mask (lengths say it all) nor the causal mask")
        groups = None
        if enc is not None:
            enc = enc.contiguous()
            if xq is None and enc_index is not None and Fx.attn_grouped_ok(T, Nenc):
                groups = Fx.kv_groups(enc_index, enc.shape[0] // Nenc)
        # The K/V projections of the image states depend on no text-side activation (xroberta.py:224-226 recomputes them in every
        # layer from the same encoder_hidden_states): all layers' projections are enqueued up front on the second stream and run under
        # the self-attention halves of the layers, off the dependent chain; each cross-attention waits for its own layer's event.
        kv_ready = {}
        if enc is not None:
            pre = _WgradStream(x.device)
            for li in range(lo, hi):
                layer = model.encoder.layer[li]
                if layer.has_cross_attention:
                    kv_ready[li] = pre.project(enc, layer._s["kv2"])
        for li in range(lo, hi):
            layer = model.encoder.layer[li]
            s = layer._s
            att, co = layer.attention, None
            d_att, d_h1 = Fx.drop_params(p_att, _next_seed()), Fx.drop_params(p_hid, _next_seed())
            qkv = Fx.gemm_nt(x, s["qkv"].wb, s["qkv"].b)
            c1, lse1 = Fx.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, H, T, T, scale, key_keep=key_keep,
                                          causal=causal, drop=d_att, q_pack=qp, k_pack=qp, zero_fill=zf)
            h1 = Fx.gemm_nt(c1, s["o"].wb, s["o"].b)
            ln1 = att.output.LayerNorm
            y1, z1, m1, r1, *tw = Fx.ln_post_fwd(h1, xres, ln1.weight, ln1.bias, ln1.eps, d_h1, f32=f32)
            yres = tw[0] if f32 else y1
            rec = {"x": x, "qkv": qkv, "c1": c1, "lse1": lse1, "z1": z1, "m1": m1, "r1": r1, "y1": y1, "d_att": d_att, "d_h1": d_h1}
            y2 = y1
            if layer.has_cross_attention and enc is not None:
                co = layer.crossattention
                d_att2, d_h2 = Fx.drop_params(p_att, _next_seed()), Fx.drop_params(p_hid, _next_seed())
                q2 = Fx.gemm_nt(y1, s["q2"].wb, s["q2"].b)
                kv = pre.take(kv_ready.pop(li))
                if xq is not None:  # RANGE mode: sequences laid out image by image -> one ragged problem per image, full query tiles
                    c2, lse2, c2lo = Fx.attn_fwd(q2, kv[:, :D], kv[:, D:], enc.shape[0] // Nenc, H, xq[2], Nenc, scale, key_keep=enc_keep,
                                                 drop=d_att2, q_pack=(xq[0], xq[1]), zero_fill=zf, lo=True)
                elif groups is not None:  # one workgroup per (image, head): K/V staged once for every row that reads it
                    # (lo: the low half of the output -- the backward's delta = dO . (O + Olo) then needs no sweep over the keys)
                    c2, lse2, c2lo = Fx.attn_fwd(q2, kv[:, :D], kv[:, D:], B, H, T, Nenc, scale, key_keep=enc_keep, drop=d_att2, groups=groups,
                                                 q_pack=qp, zero_fill=zf, lo=True)
                elif pack is not None:
                    raise NotImplementedError("packed rows with cross-attention need the grouped kernels (T <= 64) and an encoder_batch_index")
                else:
                    c2, lse2 = Fx.attn_fwd(q2, kv[:, :D], kv[:, D:], B, H, T, Nenc, scale, key_keep=enc_keep, drop=d_att2, kv_index=enc_index)
                    c2lo = None
                h2 = Fx.gemm_nt(c2, s["o2"].wb, s["o2"].b)
                ln2 = co.output.LayerNorm
                y2, z2, m2, r2, *tw = Fx.ln_post_fwd(h2, yres, ln2.weight, ln2.bias, ln2.eps, d_h2, f32=f32)
                yres = tw[0] if f32 else y2
                rec.update(q2=q2, kv=kv, c2=c2, c2lo=c2lo, lse2=lse2, z2=z2, m2=m2, r2=r2, y2=y2, d_att2=d_att2, d_h2=d_h2)
            d_h3 = Fx.drop_params(p_hid, _next_seed())
            hact, u = Fx.gemm_nt(y2, s["i"].wb, s["i"].b, epi=Fx.EPI_GELU)
            h3 = Fx.gemm_nt(hact, s["out"].wb, s["out"].b)
            ln3 = layer.output.LayerNorm
            y3, z3, m3, r3, *tw = Fx.ln_post_fwd(h3, yres, ln3.weight, ln3.bias, ln3.eps, d_h3, f32=f32)
            rec.update(hact=hact, u=u, z3=z3, m3=m3, r3=r3, d_h3=d_h3, cross=co is not None)
            saved.append(rec)
            x = y3
            xres = tw[0] if f32 else y3
        ctx.saved, ctx.model, ctx.enc = saved, model, enc
        if grad_batch is not None and (enc is not None or not 0 < grad_batch <= B):
            raise ValueError("grad_batch is for self-attention-only passes: 0 < grad_batch <= batch")
        ctx.meta = (lo, hi, causal, B, T, Nenc, key_keep, enc_keep, need_dx, need_denc, scale, enc_index, groups, grad_batch)
        ctx.pack, ctx.xq = pack, xq
        ctx.noted = need_dx or need_denc
        if ctx.noted:
            arena_note_use(model)
        return x, (xres if f32 else None)

    @staticmethod
    def backward(ctx, dy, dy32=None):
        model, enc = ctx.model, ctx.enc
        lo, hi, causal, B, T, Nenc, key_keep, enc_keep, need_dx, need_denc, scale, enc_index, groups, grad_batch = ctx.meta
        cfg = model.config
        D, H = cfg.hidden_size, cfg.num_attention_heads
        g = grad_view
        f32 = _F32_STREAM
        dy_a, dy_b = _grad_pair(dy, dy32, f32)
        dy = dy_a
        wg = _WgradStream(dy.device, "fusion bwd" if enc is not None else "text bwd")
        B_full = B
        pack = ctx.pack
        rows_full = dy_a.shape[0]
        if grad_batch is not None and grad_batch < B:
            # only the first `grad_batch` sequences carry gradient (the rest of the pass was a detached forward that shared the
            # GEMMs): every op is per token / per sequence, so the backward runs on the row prefix of the saved activations
            B = grad_batch
            G = B * T if pack is None else pack.rows_of_head(B)
            if pack is not None:
                pack = pack.head(B)
            dy_a = dy_a[:G]
            dy_b = None if dy_b is None else dy_b[:G]
            key_keep = None if key_keep is None else key_keep[:B]
            for rec in ctx.saved:
                for k, v in list(rec.items()):
                    if torch.is_tensor(v):
                        rec[k] = v[:B] if k.startswith("lse") else v[:G]
        qp = None if pack is None else pack.pair
        # packed rows with slack: the attention kernels only write real-token rows, the rest must be zero gradient
        new_grad = torch.zeros_like if (pack is not None and not pack.exact) else torch.empty_like
        denc32 = None
        # gradient w.r.t. the shared image states = sum over the cross-attention layers of dKV_l @ Wkv_l: with the grouped
        # kernels every layer's dKV is [images*Nenc, 2D], so the layers write column blocks of ONE buffer and a single GEMM
        # with K = layers*2D (against the column-concatenated transposed weights) replaces one fp32 read-modify-write GEMM per layer
        xq = ctx.xq
        per_image = groups is not None or xq is not None   # dK / dV come out per image (not per query sequence)
        cross_layers = [li for li in range(lo, hi) if model.encoder.layer[li].has_cross_attention] if enc is not None else []
        concat_k = need_denc and per_image and len(cross_layers) > 1
        dkv_all = torch.empty((enc.shape[0], len(cross_layers) * 2 * D), dtype=BF16, device=dy.device) if concat_k else None
        if need_denc and not concat_k:
            denc32 = torch.zeros((enc.shape[0], D), dtype=F32, device=dy.device)
        for li in reversed(range(lo, hi)):
            layer = model.encoder.layer[li]
            s, r = layer._s, ctx.saved[li - lo]
            ln3 = layer.output.LayerNorm
            dh3, dres3 = Fx.ln_post_bwd(dy_a, r["z3"], r["m3"], r["r3"], ln3.weight, g(ln3.weight), g(ln3.bias), s["out"].db,
                                        dy2=dy_b, drop=r["d_h3"])
            wg.gemm_tn(dh3, r["hact"], s["out"].dw)
            du = Fx.gemm_nt(dh3, s["out"].wt, epi=Fx.EPI_DGELU, aux=r["u"], n=s["out"].K)
            y2 = r["y2"] if r["cross"] else r["y1"]
            wg.gemm_tn(du, y2, s["i"].dw, dbias=s["i"].db)
            d1a, d1b = Fx.gemm_nt(du, s["i"].wt, n=s["i"].K), dres3
            if r["cross"]:
                ln2 = layer.crossattention.output.LayerNorm
                dh2, dres2 = Fx.ln_post_bwd(d1a, r["z2"], r["m2"], r["r2"], ln2.weight, g(ln2.weight), g(ln2.bias), s["o2"].db,
                                            dy2=d1b, drop=r["d_h2"])
                wg.gemm_tn(dh2, r["c2"], s["o2"].dw)
                dc2 = Fx.gemm_nt(dh2, s["o2"].wt, n=s["o2"].K)
                kv = r["kv"]
                dq2 = new_grad(r["q2"])
                if per_image:  # dK/dV accumulated over each image's rows in registers, written once per image
                    if concat_k:
                        j = cross_layers.index(li)
                        dkv = dkv_all[:, j * 2 * D:(j + 1) * 2 * D]
                    else:
                        dkv = torch.empty((enc.shape[0], 2 * D), dtype=BF16, device=dq2.device)
                    # dQ stays on the activation-gradient chain; dK/dV (they only feed the K/V weight gradient and, after the last
                    # layer, the gradient of the image states) run next to the weight-gradient GEMMs on the second stream
                    if xq is not None:
                        args = (dc2, r["q2"], kv[:, :D], kv[:, D:], r["c2"], r["lse2"], dq2, dkv[:, :D], dkv[:, D:], enc.shape[0] // Nenc, H,
                                xq[2], Nenc, scale)
                        kw = dict(key_keep=enc_keep, drop=r["d_att2"], q_pack=(xq[0], xq[1]), o_lo=r["c2lo"])
                    else:
                        args = (dc2, r["q2"], kv[:, :D], kv[:, D:], r["c2"], r["lse2"], dq2, dkv[:, :D], dkv[:, D:], B, H, T, Nenc, scale)
                        kw = dict(key_keep=enc_keep, drop=r["d_att2"], groups=groups, q_pack=qp, o_lo=r["c2lo"])
                    delta = Fx.attn_bwd(*args, phase=1, **kw)
                    wg.run(lambda: Fx.attn_bwd(*args, phase=2, delta=delta, **kw), keep=args[:9] + (delta,))
                else:
                    dkv = torch.empty((B * Nenc, 2 * D), dtype=BF16, device=dq2.device)  # per query row
                    Fx.attn_bwd(dc2, r["q2"], kv[:, :D], kv[:, D:], r["c2"], r["lse2"], dq2, dkv[:, :D], dkv[:, D:], B, H, T, Nenc,
                                scale, key_keep=enc_keep, drop=r["d_att2"], kv_index=enc_index)
                    if enc_index is not None:  # fold onto the unique key/value sources
                        dkv = Fx.rows_index_sum(dkv, enc_index, enc.shape[0] // Nenc, Nenc)
                wg.gemm_tn(dq2, r["y1"], s["q2"].dw, dbias=s["q2"].db)
                wg.gemm_tn(dkv, enc, s["kv2"].dw, dbias=s["kv2"].db)
                if need_denc and not concat_k:
                    if per_image:  # dkv is produced on the second stream: its consumer follows it there
                        wg.run(lambda dkv=dkv, s=s: Fx.gemm_nt(dkv, s["kv2"].wt, epi=Fx.EPI_F32_ACC, out=denc32, n=s["kv2"].K), keep=(dkv,))
                    else:
                        Fx.gemm_nt(dkv, s["kv2"].wt, epi=Fx.EPI_F32_ACC, out=denc32, n=s["kv2"].K)
                d1a, d1b = Fx.gemm_nt(dq2, s["q2"].wt, n=s["q2"].K), dres2
            ln1 = layer.attention.output.LayerNorm
            dh1, dres1 = Fx.ln_post_bwd(d1a, r["z1"], r["m1"], r["r1"], ln1.weight, g(ln1.weight), g(ln1.bias), s["o"].db,
                                        dy2=d1b, drop=r["d_h1"])
            wg.gemm_tn(dh1, r["c1"], s["o"].dw)
            dc1 = Fx.gemm_nt(dh1, s["o"].wt, n=s["o"].K)
            qkv = r["qkv"]
            dqkv = new_grad(qkv)
            Fx.attn_bwd(dc1, qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], r["c1"], r["lse1"], dqkv[:, :D], dqkv[:, D:2 * D],
                        dqkv[:, 2 * D:], B, H, T, T, scale, key_keep=key_keep, causal=causal, drop=r["d_att"], q_pack=qp, k_pack=qp)
            wg.gemm_tn(dqkv, r["x"], s["qkv"].dw, dbias=s["qkv"].db)
            if li > lo or need_dx:
                dy_a, dy_b = Fx.gemm_nt(dqkv, s["qkv"].wt, n=s["qkv"].K), dres1
            ctx.saved[li - lo] = None
        dx, dx32 = _input_grads(dy_a, dy_b, need_dx, ctx.twin_in, rows_full)
        denc = None
        if concat_k:
            wt_cat = torch.cat([model.encoder.layer[li]._s["kv2"].wt[:, :2 * D] for li in cross_layers], dim=1)  # [D_enc, layers*2D]
            wg.join()  # the dK/dV blocks were written on the second stream
            denc = Fx.gemm_nt(dkv_all, wt_cat)
        elif need_denc:
            wg.join()
            denc = denc32.to(BF16)
        wg.join()  # the weight gradients are complete in main-stream order before the tower's all-reduce / the optimizer
        if ctx.noted:
            arena_note_grad(model)
        return (dx, dx32, denc) + (None,) * 14



class _LastLayerRowsFn(torch.autograd.Function):
    """ONE RobertaLayer computed only on the token rows somebody reads afterwards.

    The consumers of the fusion tower's last layer are the [CLS] rows of the 3B matching sequences (xfm.py:795-797) and the masked
    positions of the B MLM sequences (xroberta.py:1215-1216): ~1/5 of the token rows.  No op of a layer couples token rows except
    attention, and there only through the keys: so the last layer projects K / V for every row of a sequence but runs its queries,
    both attention outputs, the three LayerNorms and the FFN on the selected rows alone (same arithmetic per selected row as the
    full layer; the other rows' outputs were dead values).  x: [R, D] packed rows (all), `rows`: int32 [S] indices into them
    (duplicates allowed), `sel` = (start, len) int32 [B] of each sequence's block inside the S-row layout, `Tq` = max block length.
    Output: [S, D].  Backward: the selected rows' gradients are scatter-added into the full-row gradient, next to dK/dV . W_kv."""

    @staticmethod
    def forward(ctx, x, x32, enc, model, li, B, T, Nenc, training, enc_index, pack, rows, sel, Tq):
        cfg = model.config
        f32 = _F32_STREAM
        ctx.set_materialize_grads(False)
        ctx.twin_in = f32 and x32 is not None
        D, H = cfg.hidden_size, cfg.num_attention_heads
        scale = 1.0 / math.sqrt(D // H)
        p_att = cfg.attention_probs_dropout_prob if training else 0.0
        p_hid = cfg.hidden_dropout_prob if training else 0.0
        layer = model.encoder.layer[li]
        s = layer._s
        cross = layer.has_cross_attention and enc is not None
        x = x.contiguous()
        kp = pack.pair
        pre = _WgradStream(x.device)
        kv_pair = pre.project(enc, s["kv2"]) if cross else None
        groups = Fx.kv_groups(enc_index, enc.shape[0] // Nenc) if cross else None
        d_att, d_h1 = Fx.drop_params(p_att, _next_seed()), Fx.drop_params(p_hid, _next_seed())
        wqkv, bqkv = s["qkv"].wb, s["qkv"].b
        kvs = Fx.gemm_nt(x, wqkv[D:3 * D], bqkv[D:3 * D])                      # K | V of every row
        xs = Fx.rows_gather(x, rows)
        xres = Fx.rows_gather(x32.contiguous(), rows) if ctx.twin_in else xs   # the selected rows' residual, un-rounded when the twin exists
        qs = Fx.gemm_nt(xs, wqkv[:D], bqkv[:D])                                # Q of the selected rows
        c1, lse1 = Fx.attn_fwd(qs, kvs[:, :D], kvs[:, D:], B, H, Tq, T, scale, drop=d_att, q_pack=sel, k_pack=kp, zero_fill=False)
        h1 = Fx.gemm_nt(c1, s["o"].wb, s["o"].b)
        ln1 = layer.attention.output.LayerNorm
        y1, z1, m1, r1, *tw = Fx.ln_post_fwd(h1, xres, ln1.weight, ln1.bias, ln1.eps, d_h1, f32=f32)
        yres = tw[0] if f32 else y1
        rec = dict(x=x, xs=xs, kvs=kvs, qs=qs, c1=c1, lse1=lse1, z1=z1, m1=m1, r1=r1, y1=y1, d_att=d_att, d_h1=d_h1, cross=cross)
        y2 = y1
        if cross:
            d_att2, d_h2 = Fx.drop_params(p_att, _next_seed()), Fx.drop_params(p_hid, _next_seed())
            q2 = Fx.gemm_nt(y1, s["q2"].wb, s["q2"].b)
            kv = pre.take(kv_pair)
            c2, lse2, c2lo = Fx.attn_fwd(q2, kv[:, :D], kv[:, D:], B, H, Tq, Nenc, scale, drop=d_att2, groups=groups, q_pack=sel, zero_fill=False,
                                         lo=True)
            h2 = Fx.gemm_nt(c2, s["o2"].wb, s["o2"].b)
            ln2 = layer.crossattention.output.LayerNorm
            y2, z2, m2, r2, *tw = Fx.ln_post_fwd(h2, yres, ln2.weight, ln2.bias, ln2.eps, d_h2, f32=f32)
            yres = tw[0] if f32 else y2
            rec.update(q2=q2, kv=kv, c2=c2, c2lo=c2lo, lse2=lse2, z2=z2, m2=m2, r2=r2, y2=y2, d_att2=d_att2, d_h2=d_h2)
        d_h3 = Fx.drop_params(p_hid, _next_seed())
        hact, u = Fx.gemm_nt(y2, s["i"].wb, s["i"].b, epi=Fx.EPI_GELU)
        h3 = Fx.gemm_nt(hact, s["out"].wb, s["out"].b)
        ln3 = layer.output.LayerNorm
        y3, z3, m3, r3, *tw = Fx.ln_post_fwd(h3, yres, ln3.weight, ln3.bias, ln3.eps, d_h3, f32=f32)
        rec.update(hact=hact, u=u, z3=z3, m3=m3, r3=r3, d_h3=d_h3)
        ctx.rec, ctx.model, ctx.enc, ctx.layer = rec, model, enc, layer
        need_dx = x.requires_grad or (x32 is not None and x32.requires_grad)
        ctx.meta = (B, T, Nenc, scale, groups, kp, rows, sel, Tq, need_dx, enc is not None and enc.requires_grad)
        ctx.noted = need_dx or (enc is not None and enc.requires_grad)
        if ctx.noted:
            arena_note_use(model)
        return y3, (tw[0] if f32 else None)

    @staticmethod
    def backward(ctx, dy, dy32=None):
        model, enc, layer, r = ctx.model, ctx.enc, ctx.layer, ctx.rec
        B, T, Nenc, scale, groups, kp, rows, sel, Tq, need_dx, need_denc = ctx.meta
        cfg = model.config
        D, H = cfg.hidden_size, cfg.num_attention_heads
        g, s = grad_view, layer._s
        dy_a, dy_b = _grad_pair(dy, dy32, _F32_STREAM)
        dy = dy_a
        wg = _WgradStream(dy.device, "fusion bwd" if enc is not None else "text bwd")
        ln3 = layer.output.LayerNorm
        dh3, dres3 = Fx.ln_post_bwd(dy_a, r["z3"], r["m3"], r["r3"], ln3.weight, g(ln3.weight), g(ln3.bias), s["out"].db, dy2=dy_b,
                                    drop=r["d_h3"])
        wg.gemm_tn(dh3, r["hact"], s["out"].dw)
        du = Fx.gemm_nt(dh3, s["out"].wt, epi=Fx.EPI_DGELU, aux=r["u"], n=s["out"].K)
        y2 = r["y2"] if r["cross"] else r["y1"]
        wg.gemm_tn(du, y2, s["i"].dw, dbias=s["i"].db)
        d1a, d1b = Fx.gemm_nt(du, s["i"].wt, n=s["i"].K), dres3
        denc = None
        if r["cross"]:
            ln2 = layer.crossattention.output.LayerNorm
            dh2, dres2 = Fx.ln_post_bwd(d1a, r["z2"], r["m2"], r["r2"], ln2.weight, g(ln2.weight), g(ln2.bias), s["o2"].db, dy2=d1b,
                                        drop=r["d_h2"])
            wg.gemm_tn(dh2, r["c2"], s["o2"].dw)
            dc2 = Fx.gemm_nt(dh2, s["o2"].wt, n=s["o2"].K)
            kv = r["kv"]
            dq2 = torch.empty_like(r["q2"])
            dkv = torch.empty((enc.shape[0], 2 * D), dtype=BF16, device=dq2.device)
            args = (dc2, r["q2"], kv[:, :D], kv[:, D:], r["c2"], r["lse2"], dq2, dkv[:, :D], dkv[:, D:], B, H, Tq, Nenc, scale)
            kw = dict(drop=r["d_att2"], groups=groups, q_pack=sel, o_lo=r["c2lo"])
            delta = Fx.attn_bwd(*args, phase=1, **kw)
            wg.run(lambda: Fx.attn_bwd(*args, phase=2, delta=delta, **kw), keep=args[:9] + (delta,))
            wg.gemm_tn(dq2, r["y1"], s["q2"].dw, dbias=s["q2"].db)
            wg.gemm_tn(dkv, enc, s["kv2"].dw, dbias=s["kv2"].db)
            if need_denc:
                denc = wg.run(lambda: Fx.gemm_nt(dkv, s["kv2"].wt, n=s["kv2"].K), keep=(dkv,))
            d1a, d1b = Fx.gemm_nt(dq2, s["q2"].wt, n=s["q2"].K), dres2
        ln1 = layer.attention.output.LayerNorm
        dh1, dres1 = Fx.ln_post_bwd(d1a, r["z1"], r["m1"], r["r1"], ln1.weight, g(ln1.weight), g(ln1.bias), s["o"].db, dy2=d1b,
                                    drop=r["d_h1"])
        wg.gemm_tn(dh1, r["c1"], s["o"].dw)
        dc1 = Fx.gemm_nt(dh1, s["o"].wt, n=s["o"].K)
        kvs, qs = r["kvs"], r["qs"]
        dqs, dkvs = torch.empty_like(qs), torch.empty_like(kvs)
        Fx.attn_bwd(dc1, qs, kvs[:, :D], kvs[:, D:], r["c1"], r["lse1"], dqs, dkvs[:, :D], dkvs[:, D:], B, H, Tq, T, scale, drop=r["d_att"],
                    q_pack=sel, k_pack=kp)
        dw, db = s["qkv"].dw, s["qkv"].db
        wg.gemm_tn(dqs, r["xs"], dw[:D], dbias=db[:D])
        wg.gemm_tn(dkvs, r["x"], dw[D:3 * D], dbias=db[D:3 * D])
        dx, dxt = None, None
        if need_dx:
            wt = s["qkv"].wt
            dx32 = Fx.gemm_nt(dkvs, wt[:, D:3 * D], epi=Fx.EPI_F32, n=s["qkv"].K)           # through K / V: every row
            Fx.rows_scatter_add(Fx.gemm_nt(dqs, wt[:, :D], n=s["qkv"].K), rows, dx32)     # through Q and the residual: selected rows
            if dres1.dtype == F32:   # fp32 stream: the residual-branch gradient of the selected rows stays fp32 ...
                if ctx.twin_in:      # ... and leaves on the twin's edge
                    dxt = torch.zeros_like(dx32).index_add_(0, rows.long(), dres1)
                else:
                    dx32.index_add_(0, rows.long(), dres1)
            else:
                Fx.rows_scatter_add(dres1, rows, dx32)
            dx = dx32.to(BF16)
        wg.join()
        if ctx.noted:
            arena_note_grad(model)
        ctx.rec = None
        return (dx, dxt, denc) + (None,) * 11


_NATIVE_LAYERS = os.environ.get("XFM_NATIVE_LAYERS", "1") != "0"  # A/B knob: one C-ABI call per RobertaLayer (csrc/encoder.hip)


def _rlayer_params(layer, cfg):
    """The layer's static pointers for xfm_rlayer_fwd / _bwd (cached: operand copies and arena views are allocated once)."""
    P = getattr(layer, "_rlp", None)
    if P is not None:
        return P
    from ._lib import RLayerParams
    s = layer._s
    for slot in s.values():
        slot._alloc()
    P = RLayerParams()

    def put(name, slot, fwd=True):
        if fwd:
            setattr(P, "w" + name, slot._wb.data_ptr())
            setattr(P, "b" + name, slot.b.data_ptr())
        setattr(P, "w" + name + "_t", slot._wt.data_ptr())
        setattr(P, "ld_w" + name + "_t", slot._wt.stride(0))
        setattr(P, "dw" + name, slot._dw.data_ptr())
        setattr(P, "db" + name, slot._db.data_ptr())

    put("qkv", s["qkv"]); put("o", s["o"]); put("i", s["i"]); put("out", s["out"])
    lns = [layer.attention.output.LayerNorm, None, layer.output.LayerNorm]
    if layer.has_cross_attention:
        put("q2", s["q2"]); put("kv2", s["kv2"], fwd=False); put("o2", s["o2"])
        lns[1] = layer.crossattention.output.LayerNorm
    for k, ln in enumerate(lns):
        if ln is not None:
            setattr(P, f"ln{k + 1}_w", ln.weight.data_ptr()); setattr(P, f"ln{k + 1}_b", ln.bias.data_ptr())
            setattr(P, f"dln{k + 1}_w", ln.weight._xfm_grad.data_ptr()); setattr(P, f"dln{k + 1}_b", ln.bias._xfm_grad.data_ptr())
    P.D, P.H, P.FF, P.has_cross, P.eps = cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size, int(layer.has_cross_attention), \
        layer.output.LayerNorm.eps
    layer._rlp = P
    arena = s["qkv"]._arena
    own = [p for k in ("qkv", "o", "i", "out") for plist in (s[k].weights, s[k].biases) for p in plist if not isinstance(p, int)]
    own += [lns[0].weight, lns[0].bias, lns[2].weight, lns[2].bias]
    cross = []
    if layer.has_cross_attention:
        cross = [p for k in ("q2", "kv2", "o2") for plist in (s[k].weights, s[k].biases) for p in plist if not isinstance(p, int)]
        cross += [lns[1].weight, lns[1].bias]
    layer._rl_params = (own, cross)
    layer._rl_units = (sorted({arena._unit_of[id(p)] for p in own}), sorted({arena._unit_of[id(p)] for p in cross}))
    return P


def _rl_touch(layer, arena, cross):
    """Mark the layer's parameters live (their gradients are written by the native backward) -- a unit-flag check per layer once
    they all are."""
    live = arena.live
    for units, params in zip(layer._rl_units[:2 if cross else 1], layer._rl_params):
        for u in units:
            if not live[u]:
                arena.touch_all(params)
                break


def _view(slab, off, rows, cols, dtype=BF16):
    n = rows * cols * (2 if dtype == BF16 else 4)
    return slab[off:off + n].view(dtype).view(rows, cols)


_KV_AHEAD = os.environ.get("XFM_KV_AHEAD", "1") != "0"   # A/B knob of RobertaModel.prefetch_cross_kv
_RL_DEFER_LN = os.environ.get("XFM_RL_DEFER_LN", "1") != "0"   # A/B knob: one batched LayerNorm column-sum reduce per tower
_RL_DEFER_WGRAD = os.environ.get("XFM_RL_DEFER_WGRAD", "1") != "0"
_TOWER_JOIN_END = os.environ.get("XFM_TOWER_JOIN_END", "0") != "0"


class _EncoderFnNative(torch.autograd.Function):
    """Layers [lo, hi) of a RobertaEncoder with ONE C-ABI call per layer and direction (csrc/encoder.hip sequences the kernels of
    the layer on the native side).  Same kernels, same order, same dropout streams as _EncoderFn -- results are bit-identical --
    but the host spends ~15 us per layer instead of ~200 us of per-kernel wrapper work, so a tower of 10-30 us kernels stops
    being bound by the Python interpreter.  Cross-attention goes through the grouped kernels (encoder_batch_index required)."""

    @staticmethod
    def forward(ctx, x, x32, enc, model, key_keep, enc_keep, lo, hi, causal, B, T, Nenc, training, enc_index, grad_batch=None, pack=None, xq=None):
        from . import _lib
        from ._lib import RLayerIO, RLayerLayout, check
        import ctypes
        lib = _lib.load()
        cfg = model.config
        f32 = _F32_STREAM
        ctx.set_materialize_grads(False)
        ctx.twin_in = f32 and x32 is not None
        xt = x32.contiguous() if ctx.twin_in else None
        D, H, FF = cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size
        p_att = cfg.attention_probs_dropout_prob if training else 0.0
        p_hid = cfg.hidden_dropout_prob if training else 0.0
        need_dx, need_denc = x.requires_grad or (x32 is not None and x32.requires_grad), (enc is not None and enc.requires_grad)
        x = x.contiguous()
        R = x.shape[0]
        if pack is not None and (key_keep is not None or causal):
            raise ValueError("packed rows take neither a key mask (lengths say it all) nor the causal mask")
        if grad_batch is not None and (enc is not None or not 0 < grad_batch <= B):
            raise ValueError("grad_batch is for self-attention-only passes: 0 < grad_batch <= batch")
        groups, U = None, 0
        if enc is not None:
            enc = enc.contiguous()
            U = enc.shape[0] // Nenc
            if xq is None:
                groups = Fx.kv_groups(enc_index, U)
        layers = [model.encoder.layer[li] for li in range(lo, hi)]
        arena = layers[0]._s["qkv"]._arena
        ver = arena._manual_ver
        for layer in layers:           # bf16 operand copies up to date (the arena's batched refresh when its version moved) and visible here
            for slot in layer._s.values():
                slot._ensure()
        d_att, d_hid = Fx.drop_params(p_att, 1), Fx.drop_params(p_hid, 1)
        io = RLayerIO()
        io.R, io.B, io.T, io.Nenc, io.U = R, B, T, Nenc if enc is not None else 0, U
        if pack is not None:
            io.seq_start, io.seq_len = pack.start.data_ptr(), pack.lens.data_ptr()
            io.zero_fill = int(not pack.exact)
        io.key_keep, io.enc_keep = Fx._ptr(key_keep), Fx._ptr(enc_keep)
        if groups is not None:
            io.grp_start, io.grp_rows = groups[0].data_ptr(), groups[1].data_ptr()
        if xq is not None:
            io.xq_start, io.xq_len, io.xq_max = xq[0].data_ptr(), xq[1].data_ptr(), int(xq[2])
        io.causal, io.scale = int(causal), 1.0 / math.sqrt(D // H)
        io.att_thresh, io.att_scale, io.hid_thresh, io.hid_scale = d_att[0], d_att[1], d_hid[0], d_hid[1]
        io.seed_hi = torch.initial_seed() & 0xFFFFFFFF
        io.f32_stream = int(f32)
        layouts = {}

        def layout(cross):
            if cross not in layouts:
                L = RLayerLayout()
                check(lib.xfm_rlayer_layout(R, B, T, D, H, FF, int(cross), io.Nenc, U, io.xq_max if xq is not None else 0,
                                            int(p_hid > 0) | (2 if f32 else 0), ctypes.byref(L)), "rlayer_layout")
                layouts[cross] = L
            return layouts[cross]

        kv_ready = {}
        pre = _WgradStream(x.device)
        if enc is not None:
            ahead = getattr(model, "_kv_ahead", None)   # prefetch_cross_kv(): the projections of THESE image states are already queued
            model._kv_ahead = None
            if ahead is not None and ahead[0] == (enc.data_ptr(), tuple(enc.shape), enc._version):
                kv_ready = {k: v for k, v in ahead[1].items() if any(k == id(layer) for layer in layers)}
            for layer in layers:
                if layer.has_cross_attention and id(layer) not in kv_ready:
                    kv_ready[id(layer)] = pre.project(enc, layer._s["kv2"])
        st = Fx._stream()
        saved = []
        for layer in layers:
            cross = layer.has_cross_attention and enc is not None
            L = layout(cross)
            slab = torch.empty(L.fwd_bytes, dtype=torch.uint8, device=x.device)
            ctr = _seed_counter[0]
            _seed_counter[0] += 5 if cross else 3
            io.x, io.slab, io.seed_ctr = x.data_ptr(), slab.data_ptr(), ctr & 0xFFFFFFFF
            io.x32 = Fx._ptr(xt)
            kv = None
            if cross:
                kv, ev = kv_ready.pop(id(layer))
                io.kv, io.kv_ld, io.kv_event = kv.data_ptr(), kv.stride(0), (ev.cuda_event if ev is not None else 0)
            else:
                io.kv, io.kv_event = 0, 0
            check(lib.xfm_rlayer_fwd(ctypes.byref(_rlayer_params(layer, cfg)), ctypes.byref(io), st), "rlayer_fwd")
            saved.append((slab, x, kv, ctr, cross, xt))
            x = _view(slab, L.y3, R, D)
            xt = _view(slab, L.y3_32, R, D, F32) if f32 else None
        ctx.saved, ctx.model, ctx.enc, ctx.io, ctx.layouts, ctx.pack = saved, model, enc, io, layouts, pack
        ctx.keep = (groups, xq, key_keep, enc_keep)   # device arrays the io struct points at
        ctx.meta = (lo, hi, B, T, Nenc, U, key_keep, need_dx, need_denc, groups, grad_batch, p_hid > 0)
        ctx.noted = need_dx or need_denc
        if ctx.noted:
            arena_note_use(model)
        return x, xt

    @staticmethod
    def backward(ctx, dy, dy32=None):
        from . import _lib
        from ._lib import RLayerBwd, check
        import ctypes
        lib = _lib.load()
        model, enc, io, pack = ctx.model, ctx.enc, ctx.io, ctx.pack
        lo, hi, B, T, Nenc, U, key_keep, need_dx, need_denc, groups, grad_batch, dropout = ctx.meta
        cfg = model.config
        D = cfg.hidden_size
        layers = [model.encoder.layer[li] for li in range(lo, hi)]
        arena = layers[0]._s["qkv"]._arena
        f32 = bool(io.f32_stream)
        dy_a, dy_b = _grad_pair(dy, dy32, f32)
        dy = dy_a
        rows_full, R = dy_a.shape[0], dy_a.shape[0]
        if grad_batch is not None and grad_batch < B:
            # only the first `grad_batch` sequences carry gradient: the backward walks the row prefix of the saved activations
            R = grad_batch * T if pack is None else pack.rows_of_head(grad_batch)
            io.R_alloc, io.B_alloc, io.R, io.B = rows_full, B, R, grad_batch
            dy_a = dy_a[:R]
            dy_b = None if dy_b is None else dy_b[:R]
            if pack is not None:
                io.zero_fill = int(not pack.head(grad_batch).exact)
        wg = _WgradStream(dy.device, "fusion bwd" if enc is not None else "text bwd")
        cross_layers = [layer for layer in layers if layer.has_cross_attention] if enc is not None else []
        concat_k = need_denc and len(cross_layers) > 1
        dkv_all = torch.empty((enc.shape[0], len(cross_layers) * 2 * D), dtype=BF16, device=dy.device) if concat_k else None
        denc32 = torch.zeros((enc.shape[0], D), dtype=F32, device=dy.device) if (need_denc and not concat_k) else None
        Lmax = max(ctx.layouts.values(), key=lambda L: L.ws_side_bytes)
        bw = RLayerBwd()
        ws_main = Fx.workspace(max(L.ws_main_bytes for L in ctx.layouts.values()), dy.device)
        bw.ws_main, bw.ws_main_bytes = ws_main.data_ptr(), ws_main.numel() * 4
        if wg.on:
            with torch.cuda.stream(wg.side):
                ws_side = Fx.workspace(Lmax.ws_side_bytes, dy.device)
            bw.side_stream = wg.side.cuda_stream
        else:
            # (single stream: the per-stream workspace is already ws_main -- take a buffer of its own)
            ws_side = torch.empty(max(Lmax.ws_side_bytes // 4 + 1, 1), dtype=F32, device=dy.device)
        bw.ws_side, bw.ws_side_bytes = ws_side.data_ptr(), ws_side.numel() * 4
        if enc is not None:
            bw.enc = enc.data_ptr()
        bw.denc32 = Fx._ptr(denc32)
        st = Fx._stream()
        keep = [ws_main, ws_side]
        # XFM_RL_DEFER_WGRAD (default on): the executor launches no weight gradient; the tower's queue runs as grouped launches at its
        # end (whole 256 x 256 tiles over all of M instead of 10 M-splits + a reduce per projection: the fusion + text towers' weight
        # gradients cost 2.8 ms of the step one by one under the activation-gradient chain, ~1.6 ms grouped)
        defer = _RL_DEFER_WGRAD and R >= 1024
        # the LayerNorm backward kernels' column-sum folds (dgamma / dbeta / output-projection bias gradients): partials stay in per-call
        # slices of one buffer and ONE batched reduce runs with the weight gradients, instead of a 7-us kernel behind each of the 3 per layer
        ln_items = ln_count = ln_ws = None
        if _RL_DEFER_LN:
            from ._lib import ReduceItem
            ln_stride = (lib.xfm_layernorm_bwd_workspace(io.R_alloc if io.R_alloc > 0 else R, D, 1) // 4 + 63) // 64 * 64   # LN_POST
            ln_ws = torch.empty(max(3 * len(layers) * ln_stride, 1), dtype=F32, device=dy.device)
            ln_items, ln_count = (ReduceItem * (3 * len(layers)))(), ctypes.c_int(0)
            bw.ln_items, bw.ln_count, bw.ln_ws_stride = ctypes.addressof(ln_items), ctypes.addressof(ln_count), ln_stride
        for k in reversed(range(len(layers))):
            layer = layers[k]
            slab, x_in, kv, ctr, cross, xt_in = ctx.saved[k]
            L = ctx.layouts[cross]
            bslab = torch.empty(L.bwd_bytes, dtype=torch.uint8, device=dy.device)
            keep.append((bslab, slab, x_in, kv, dy_a, dy_b, xt_in))
            io.x, io.slab, io.seed_ctr = x_in.data_ptr(), slab.data_ptr(), ctr & 0xFFFFFFFF
            io.x32 = Fx._ptr(xt_in)
            if cross:
                io.kv, io.kv_ld, io.kv_event = kv.data_ptr(), kv.stride(0), 0
                if concat_k:
                    j = cross_layers.index(layer)
                    bw.dkv, bw.dkv_ld = dkv_all.data_ptr() + j * 2 * D * 2, dkv_all.stride(0)
                else:
                    dkv = torch.empty((enc.shape[0], 2 * D), dtype=BF16, device=dy.device)
                    keep.append(dkv)
                    bw.dkv, bw.dkv_ld = dkv.data_ptr(), dkv.stride(0)
            else:
                io.kv = 0
            bw.bslab, bw.dy_a = bslab.data_ptr(), dy_a.data_ptr()
            bw.dy_b, bw.dy_b32 = (0, Fx._ptr(dy_b)) if f32 else (Fx._ptr(dy_b), 0)
            bw.need_dprev = int(k > 0 or need_dx)
            bw.defer_wgrad = int(defer)
            if ln_ws is not None:
                bw.ln_ws = ln_ws.data_ptr() + k * 3 * ln_stride * 4
            _rl_touch(layer, arena, cross)
            check(lib.xfm_rlayer_bwd(ctypes.byref(_rlayer_params(layer, cfg)), ctypes.byref(io), ctypes.byref(bw), st), "rlayer_bwd")
            if defer:   # the layer's weight gradients join the tower's queue: operands at the layout's offsets of the two slabs (kept alive)
                sl, FF = layer._s, cfg.intermediate_size
                Sv = lambda off, cols, slab=slab: _view(slab, off, R, cols)       # noqa: E731
                Gv = lambda off, cols, bslab=bslab: _view(bslab, off, R, cols)    # noqa: E731
                wg.defer_tn(Gv(L.dh3, D), Sv(L.hact, FF), sl["out"]._dw)
                wg.defer_tn(Gv(L.du, FF), Sv(L.y2 if cross else L.y1, D), sl["i"]._dw, sl["i"]._db)
                if cross:
                    wg.defer_tn(Gv(L.dh2, D), Sv(L.c2, D), sl["o2"]._dw)
                    wg.defer_tn(Gv(L.dq2, D), Sv(L.y1, D), sl["q2"]._dw, sl["q2"]._db)
                    dkv_v = dkv_all[:, j * 2 * D:(j + 1) * 2 * D] if concat_k else dkv
                    wg.defer_tn(dkv_v, enc, sl["kv2"]._dw, sl["kv2"]._db)   # (dK|dV comes from the side stream: the flush runs there too)
                wg.defer_tn(Gv(L.dh1, D), Sv(L.c1, D), sl["o"]._dw)
                wg.defer_tn(Gv(L.dqkv, 3 * D), x_in[:R], sl["qkv"]._dw, sl["qkv"]._db)
            dy_a, dy_b = _view(bslab, L.dprev, R, D), _view(bslab, L.dres1, R, D, F32 if f32 else BF16)
            ctx.saved[k] = None
        if ln_ws is not None and ln_count.value > 0:
            folds = [ln_items[i] for i in range(ln_count.value)]
            wg.defer_call(lambda: Fx.reduce_sets_batch(folds), keep=(ln_ws,))
        dx, dx32 = _input_grads(dy_a, dy_b, need_dx, ctx.twin_in, rows_full)
        denc = None
        if concat_k:
            wt_cat = torch.cat([layer._s["kv2"].wt[:, :2 * D] for layer in cross_layers], dim=1)
            wg.sync_side()   # dK|dV of every layer (second stream) -- the queued weight gradients are launched by the join below
            denc = Fx.gemm_nt(dkv_all, wt_cat)
        elif need_denc:
            wg.sync_side()
            denc = denc32.to(BF16)
        if _TOWER_JOIN_END:   # (A/B knob: the tower's queued weight gradients re-join at the END of the backward pass instead of here)
            wg.join_at_end()
        else:
            wg.join()
        del keep
        if ctx.noted:
            arena_note_grad(model)
        return (dx, dx32, denc) + (None,) * 14


class RobertaModel(nn.Module):
    embeddings_class = RobertaEmbeddings

    def __init__(self, config, add_pooling_layer=False):
        super().__init__()
        if add_pooling_layer:
            raise NotImplementedError("pooler is never built on the XFM path (xroberta.py:1167)")
        if config.hidden_size // config.num_attention_heads != 64:
            raise NotImplementedError("head_dim must be 64")
        self.config = config
        self.embeddings = self.embeddings_class(config)
        self.encoder = RobertaEncoder(config)
        self.pooler = None
        self._arena = None

    def linear_slots(self, prefix=""):
        out = []
        for i, layer in enumerate(self.encoder.layer):
            out.extend(layer.linear_slots(f"{prefix}layer.{i}."))
        return out

    def prefetch_cross_kv(self, encoder_hidden_states):
        """Queue the K|V projections of the image states for every cross-attention layer NOW, on the second stream (extension; the next
        forward with these very states picks them up).  In the pre-training step the launch stream idles ~0.5 ms between the ViT's last
        kernel and the fusion tower's first one -- the ITC logits, the draw of the negatives, their read-back and the host building the
        packed layout -- and the 12 projections (64 x 197 rows, 0.4 ms of GEMM) depend on none of that."""
        enc = encoder_hidden_states
        if not (_KV_AHEAD and enc.is_cuda and enc.dtype == BF16 and enc.is_contiguous() and self._arena is not None):
            return
        enc = enc.reshape(-1, enc.shape[-1])
        layers = [layer for layer in self.encoder.layer if layer.has_cross_attention]
        if not layers:
            return
        arena = layers[0]._s["kv2"]._arena
        for layer in layers:   # bf16 operand copies up to date and visible on this stream
            layer._s["kv2"]._ensure()
        pre = _WgradStream(enc.device)
        # (the entry holds `enc`: its memory cannot be handed to another tensor while the projections wait to be picked up)
        self._kv_ahead = ((enc.data_ptr(), tuple(enc.shape), enc._version), {id(layer): pre.project(enc, layer._s["kv2"]) for layer in layers}, enc)

    def attach(self, arena):
        self._arena = arena

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None,
                inputs_embeds=None, encoder_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None,
                past_key_values=None, use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None,
                is_decoder=False, mode='multi_modal', encoder_batch_index=None, grad_batch=None, pack=None, encoder_row_ranges=None,
                output_rows=None):
        """`encoder_batch_index` (extension, default None = reference behaviour): int tensor [B] mapping every text row to the
        row of `encoder_hidden_states` it attends to, so duplicated images are projected to K/V once per layer.
        `grad_batch` (extension): only the first grad_batch sequences of the batch propagate gradient through the layer stack
        (a detached pass batched behind a differentiable one, e.g. the masked-text pass of get_fuse_mlm_loss).
        `pack` (extension, xfm_amd.packing.Pack): run on unpadded token rows -- `attention_mask` is then implied by the pack's lengths,
        `encoder_embeds` is a 2-D [pack.cap, D] row buffer and `last_hidden_state` comes back in the same packed layout.
        `encoder_row_ranges` (extension, with `pack`): (start int32 [U], count int32 [U], max count) -- the packed sequences are laid
        out image by image, so the queries of image u are the contiguous rows start[u] .. + count[u]: cross-attention then runs as one
        ragged problem per image on full 16-row query tiles instead of per-sequence tiles that are mostly padding.
        `output_rows` (extension, with `pack`): (rows int32 [S], start int32 [B], len int32 [B], max len) -- the caller reads only these
        token rows of the last layer (sequence b's are the block start[b] .. + len[b] of the S-row result): the last layer then runs
        its queries / FFN / LayerNorms on them alone (_LastLayerRowsFn) and `last_hidden_state` is the [S, D] result."""
        if any(v is not None for v in (token_type_ids, position_ids, head_mask, inputs_embeds, past_key_values)):
            raise NotImplementedError("token_type_ids/position_ids/head_mask/inputs_embeds/past_key_values are not used on the XFM path")
        if isinstance(encoder_hidden_states, (list, tuple)):
            raise NotImplementedError("per-layer encoder_hidden_states lists (xroberta.py:435-444) are outside the hot-path scope")
        if self._arena is None:
            raise RuntimeError("RobertaModel is not attached to a parameter arena; build it through XFMBase or call finalize()")
        cfg = self.config
        if pack is not None and is_decoder:
            raise NotImplementedError("packed rows are for the bidirectional towers")
        if encoder_embeds is None:
            if input_ids is None:
                raise ValueError("You have to specify either input_ids or inputs_embeds")
            B, T = input_ids.shape
            drop = Fx.drop_params(cfg.hidden_dropout_prob if self.training else 0.0, _next_seed())
            x, x32 = _EmbedFn.apply(self.embeddings.word_embeddings.weight, self.embeddings, input_ids, drop, self, pack)
        else:
            if pack is not None:
                assert encoder_embeds.dim() == 2 and encoder_embeds.shape[0] == pack.cap, "packed encoder_embeds are [pack.cap, D] rows"
                B, T = pack.B, pack.T
            else:
                B, T = encoder_embeds.shape[:2]
            x32 = twin_of(encoder_embeds)   # the producing tower's fp32 twin, if this is its output (or a row-wise image of it: with_twin)
            if encoder_embeds.dtype == BF16:
                x = encoder_embeds
            else:   # fp32 states from outside: they ARE the un-rounded stream
                x, x32 = encoder_embeds.to(BF16), (encoder_embeds if encoder_embeds.dtype == F32 and _F32_STREAM else None)
        if pack is not None:
            assert (B, T) == (pack.B, pack.T)
            attention_mask = None  # implied by the lengths: keys past a sequence's end do not exist
        dev = x.device
        key_keep = None if attention_mask is None else attention_mask.to(device=dev, dtype=torch.int32).contiguous()
        enc, enc_keep, Nenc = None, None, 0
        if encoder_hidden_states is not None:
            enc = encoder_hidden_states if encoder_hidden_states.dtype == BF16 else encoder_hidden_states.to(BF16)
            Nenc = enc.shape[1]
            enc = enc.reshape(-1, enc.shape[-1])
            if encoder_attention_mask is not None and not getattr(encoder_attention_mask, "_xfm_all_ones", False):
                enc_keep = encoder_attention_mask.to(device=dev, dtype=torch.int32).contiguous()
            if encoder_batch_index is not None:
                encoder_batch_index = encoder_batch_index.to(device=dev, dtype=torch.int32).contiguous()
                assert encoder_batch_index.numel() == B
        if mode == 'text':
            lo, hi = 0, cfg.fusion_layer
        elif mode == 'fusion':
            lo, hi = cfg.fusion_layer, cfg.num_hidden_layers
        elif mode == 'multi_modal':
            lo, hi = 0, cfg.num_hidden_layers
        else:
            raise ValueError(f"mode {mode} is not supported")
        y = x.reshape(B * T, -1) if pack is None else x
        y32 = None if x32 is None else (x32.reshape(B * T, -1) if pack is None else x32)
        if hi > lo:
            # one native call per layer whenever cross-attention (if any) can take the grouped kernels; else kernel by kernel
            xq = None
            if encoder_row_ranges is not None and enc is not None:
                assert pack is not None and Nenc <= 256, "per-image row ranges need packed rows and <= 256 image tokens"
                xq = (encoder_row_ranges[0].to(device=dev, dtype=torch.int32).contiguous(),
                      encoder_row_ranges[1].to(device=dev, dtype=torch.int32).contiguous(), int(encoder_row_ranges[2]))
            native = _NATIVE_LAYERS and y.is_cuda and (enc is None or xq is not None or
                                                       (encoder_batch_index is not None and Fx.attn_grouped_ok(T, Nenc)))
            fn = _EncoderFnNative if native else _EncoderFn
            if output_rows is not None:
                assert pack is not None and xq is None and grad_batch is None and not is_decoder and key_keep is None
                assert enc is None or (encoder_batch_index is not None and Fx.attn_grouped_ok(int(output_rows[3]), Nenc))
                hi -= 1
            if hi > lo:
                y, y32 = fn.apply(y, y32, enc, self, key_keep, enc_keep, lo, hi, bool(is_decoder), B, T, Nenc, self.training,
                                  encoder_batch_index if enc is not None else None, grad_batch, pack, xq)
            if output_rows is not None:
                rows, sel_start, sel_len, tq = output_rows
                y, y32 = _LastLayerRowsFn.apply(y, y32, enc, self, hi, B, T, Nenc, self.training,
                                                encoder_batch_index if enc is not None else None,
                                                pack, rows.to(torch.int32).contiguous(),
                                                (sel_start.to(torch.int32).contiguous(), sel_len.to(torch.int32).contiguous()), int(tq))
        if pack is None:
            y, y32 = y.view(B, T, -1), (None if y32 is None else y32.view(B, T, -1))
        return SimpleNamespace(last_hidden_state=with_twin(y, y32), pooler_output=None, past_key_values=None,
                               hidden_states=None, attentions=None, cross_attentions=None)


class RobertaLMHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        std = config.initializer_range
        self.dense = _Lin(config.hidden_size, config.hidden_size, std)
        self.layer_norm = _Affine(config.hidden_size, config.layer_norm_eps)
        self.decoder = _Lin(config.hidden_size, config.vocab_size, std)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))
        self.decoder.bias = self.bias  # tied, xroberta.py:1322-1323

    def linear_slots(self, prefix):
        self._slot_dense = LinearSlot(prefix + "dense", [self.dense.weight], [self.dense.bias])
        self._slot_decoder = LinearSlot(prefix + "decoder", [self.decoder.weight], [self.bias], pad_k_to=64)
        return [self._slot_dense, self._slot_decoder]


class RobertaForMaskedLM(OwnsArena, nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.roberta = RobertaModel(config, add_pooling_layer=False)
        self.lm_head = RobertaLMHead(config)
        self.lm_cap_head = RobertaLMHead(config)
        self._arena = None

    def linear_slots(self, prefix=""):
        return self.roberta.linear_slots(prefix + "roberta.") + self.lm_head.linear_slots(prefix + "lm_head.") + \
            self.lm_cap_head.linear_slots(prefix + "lm_cap_head.")

    def attach(self, arena):
        self._arena = arena
        self.roberta.attach(arena)

    def finalize(self, device=None):
        device = device or self.lm_head.bias.device
        self.attach(ParamArena(self, self.linear_slots(), device))
        self._own_arena = True
        return self

    def bert(self, input_ids=None, **kw):  # accepts the reference's keywords plus `encoder_batch_index`
        return self.roberta(input_ids, **kw)

    def gather_seq_out_by_pos(self, seq, pos):
        return torch.gather(seq, 1, pos.unsqueeze(2).expand(-1, -1, seq.size(-1)))

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None,
                inputs_embeds=None, encoder_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None,
                labels=None, output_attentions=None, output_hidden_states=None, return_dict=None, is_decoder=False,
                reduction='mean', mode='multi_modal', return_logits=False, masked_pos=None):
        outputs = self.roberta(input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids,
                               position_ids=position_ids, head_mask=head_mask, inputs_embeds=inputs_embeds,
                               encoder_embeds=encoder_embeds, encoder_hidden_states=encoder_hidden_states,
                               encoder_attention_mask=encoder_attention_mask, is_decoder=is_decoder, mode=mode)
        seq = outputs.last_hidden_state
        if masked_pos is not None:
            seq = self.gather_seq_out_by_pos(seq, masked_pos)
        head = self.lm_cap_head if is_decoder else self.lm_head
        V = self.config.vocab_size
        if return_logits or labels is None:
            logits = lm_head_logits(seq.reshape(-1, seq.shape[-1]), head).view(*seq.shape[:2], V)
            if return_logits:
                return logits
            return SimpleNamespace(loss=None, logits=logits, hidden_states=None, attentions=None)
        if is_decoder:
            seq, labels = seq[:, :-1, :], labels[:, 1:]
        Bq, Tq = seq.shape[:2]
        loss, logits = lm_head_ce(seq.reshape(-1, seq.shape[-1]), head, labels.reshape(-1), reduction)
        return SimpleNamespace(loss=loss, logits=logits[:, :V].view(Bq, Tq, V), hidden_states=None, attentions=None)


class RobertaForCausalLM(OwnsArena, nn.Module):
    """Causal decoder with cross-attention to encoder states (the VQA answer decoder): mirrors models/xroberta.py:963-1153 --
    `roberta` + `lm_head`, causal self-attention mask, next-token shift, `reduction='none'` CE summed per sequence
    (:1107-1114)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.roberta = RobertaModel(config, add_pooling_layer=False)
        self.lm_head = RobertaLMHead(config)
        self._arena = None

    def linear_slots(self, prefix=""):
        return self.roberta.linear_slots(prefix + "roberta.") + self.lm_head.linear_slots(prefix + "lm_head.")

    def attach(self, arena):
        self._arena = arena
        self.roberta.attach(arena)

    def finalize(self, device=None):
        device = device or self.lm_head.bias.device
        self.attach(ParamArena(self, self.linear_slots(), device))
        self._own_arena = True
        return self

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None,
                inputs_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None, labels=None, past_key_values=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None, is_decoder=True,
                reduction='mean', mode='multi_modal', return_logits=False):
        if past_key_values is not None or use_cache:
            raise NotImplementedError("incremental decoding caches (generation) are outside the hot-path scope")
        outputs = self.roberta(input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids, position_ids=position_ids,
                               head_mask=head_mask, inputs_embeds=inputs_embeds, encoder_hidden_states=encoder_hidden_states,
                               encoder_attention_mask=encoder_attention_mask, is_decoder=is_decoder, mode=mode)
        seq = outputs.last_hidden_state
        V = self.config.vocab_size
        B, T = seq.shape[:2]
        if return_logits or labels is None:
            logits = lm_head_logits(seq.reshape(-1, seq.shape[-1]), self.lm_head).view(B, T, V)
            if return_logits:
                return logits[:, :-1, :].contiguous()
            return SimpleNamespace(loss=None, logits=logits, hidden_states=seq, past_key_values=None, attentions=None,
                                   cross_attentions=None)
        shifted, lab = seq[:, :-1, :], labels[:, 1:]
        loss, logits = lm_head_ce(shifted.reshape(-1, seq.shape[-1]), self.lm_head, lab.reshape(-1), reduction)
        if reduction == 'none':
            loss = loss.view(B, -1).sum(1)
        return SimpleNamespace(loss=loss, logits=logits[:, :V].view(B, T - 1, V), hidden_states=seq, past_key_values=None,
                               attentions=None, cross_attentions=None)
