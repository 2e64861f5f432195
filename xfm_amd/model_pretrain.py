"""Pre-training model: which losses run and how they are weighted (mirrors models/model_pretrain.py:13-116)."""
import os

import torch

from .xfm import XFMBase

_TEXT_STREAM_ON = os.environ.get("XFM_TEXT_STREAM", "1") != "0"
_PACK_ROWS = os.environ.get("XFM_PACK_ROWS", "1") != "0"  # A/B knob: `text_lens` given -> unpadded token rows in the text / fusion towers
# A/B (profiles/round3_mim_stream.md): the MIM-masked view of the images as its own ViT pass on a second stream, under the latency-bound
# fusion tower, instead of batched with the clean view as one 2B-row pass
_MIM_STREAM = os.environ.get("XFM_MIM_STREAM", "0") != "0"
_TEXT_AFTER_VIT = os.environ.get("XFM_TEXT_AFTER_VIT", "0") != "0"   # A/B knob, see forward_multimodal
_SIDE_STREAMS = {}
_MIM_STREAMS = {}


def _mim_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _MIM_STREAMS:
        _MIM_STREAMS[key] = torch.cuda.Stream(device=device)
    return _MIM_STREAMS[key]


def _side_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]



from . import marks as _marks  # noqa: E402


class _JoinAfterBackward(torch.autograd.Function):
    """Identity on the forward; in the backward it queues an end-of-backward callback that makes `main` wait for `side`.  The text
    tower's backward runs on the side stream (autograd replays nodes on their forward stream) and writes its weight gradients
    straight into the arena -- no AccumulateGrad node, so the engine's own end-of-backward stream sync does not cover it."""

    @staticmethod
    def forward(ctx, x, main, side):
        ctx.main, ctx.side = main, side
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        main, side = ctx.main, ctx.side
        torch.autograd.Variable._execution_engine.queue_callback(lambda: main.wait_stream(side))
        return g, None, None


def _rejoin(t, main, side):
    """A tower output computed on `side`, consumed on `main` from here on (main already waits for side): the allocator learns about
    the second stream, the backward re-joins at its end.  The fp32 twin (xroberta.twin_of) follows."""
    from .xroberta import rowwise, twin_of
    if t is None:
        return None
    t.record_stream(main)
    if twin_of(t) is not None:
        twin_of(t).record_stream(main)
    if t.requires_grad:
        t = rowwise(lambda u: _JoinAfterBackward.apply(u, main, side), t)
    return t


def towers_side_by_side(model, image, text_ids, text_atts):
    """(image_embeds, image_atts, text_embeds) of a fine-tuning model with the text tower on a second HIP stream, as the pre-training
    step does it: queued first, it runs under the ViT's chip-filling kernels, and autograd replays its backward on the same stream --
    next to the ViT's backward instead of in front of it (retrieval 384 px: ~2 ms of text backward + ~1 ms of text forward per step).
    XFM_TEXT_STREAM=0 or a CPU model: one after the other."""
    if not (image.is_cuda and _TEXT_STREAM_ON and torch.is_grad_enabled()):
        image_embeds, image_atts = model.get_vision_embeds(image)
        return image_embeds, image_atts, model.get_text_embeds(text_ids, text_atts)
    main = torch.cuda.current_stream(image.device)
    side = _side_stream(image.device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        text_embeds = model.get_text_embeds(text_ids, text_atts)
    image_embeds, image_atts = model.get_vision_embeds(image)
    main.wait_stream(side)
    return image_embeds, image_atts, _rejoin(text_embeds, main, side)


class XFM(XFMBase):
    accepts_text_lens = True  # forward_multimodal(text_lens=...): see there

    def __init__(self, config, load_vision_params=False, load_text_params=False):
        super().__init__(config, load_vision_params=load_vision_params, load_text_params=load_text_params,
                         use_contrastive_loss=True, use_matching_loss=True, use_mlm_loss=True, use_bbox_loss=True,
                         config_text=None)
        self.weights_map = {k: config.get('w' + k, 1.0) for k in ('region', 'web', 'imagenet', 'image', 'aux')}
        self.do_image_mask = config.get('do_image_mask', True)
        self.use_mm_mim_loss = config.get('use_mm_mim_loss', True)
        self.min_temp = config.get('min_temp', 0.001)
        self.max_temp = config.get('max_temp', 0.5)
        # MI355X-first batching (identical arithmetic per sample): the clean and the MIM-masked image go through the ViT as
        # one 2B batch, the ITM (3B) and MLM (B) fusion passes as one 4B batch.  Set False to run the reference's call order.
        self.batch_passes = config.get('batch_passes', True)

    def forward_multimodal(self, image, text_ids, text_atts, text_ids_masked=None, masked_pos=None, masked_ids=None,
                           text_ids_2=None, text_atts_2=None, text_ids_masked_2=None, masked_pos_2=None, masked_ids_2=None,
                           image_atts=None, idx_to_group_img=None, target_bbox=None, is_image=None, ret_mim_loss=False,
                           ret_bbox_loss=False, ret_match_loss=True, ret_mlm_loss=True, ret_bbox_giou=False,
                           ret_itc_loss=True, data_source=None, ids_mask=None, neg_idx=None, text_lens=None):
        """`text_lens` (extension; host-side lengths of the prefix masks `text_atts`, e.g. `text_atts.sum(1)` taken BEFORE the batch
        was uploaded): run the text and fusion towers on unpadded token rows (xfm_amd.packing).  Same values on every token that
        any loss reads; None = the reference's padded computation."""
        if ret_bbox_loss or ret_bbox_giou:
            raise NotImplementedError("bbox / region losses (model_pretrain.py:39-41,81-85) are outside the hot-path scope")
        if self.learnable_temp:
            self.temp.data.clamp_(self.min_temp, self.max_temp)  # via .data: leaves the arena's weight version untouched
        w = self.weights_map.get(data_source, None)
        zero = torch.zeros((), device=image.device)
        do_mim = ret_mim_loss and (data_source == 'imagenet' or self.use_mm_mim_loss)
        image_embeds_masked = None
        mim_stream = None
        # The text tower (small, latency-bound kernels) is independent of the vision tower until the ITC loss: with batch_passes
        # its forward is enqueued on a second HIP stream first and runs under the ViT's full-chip GEMMs; autograd replays each
        # node's backward on the stream of its forward, so the two backward chains overlap the same way.
        text_stream = None
        mlm_embeds = text_embeds = None
        both_passes = ret_match_loss and ret_mlm_loss and self.detach_text_forMLM and text_ids_masked is not None
        pack = None
        if text_lens is not None and self.batch_passes and both_passes and data_source != 'imagenet' and _PACK_ROWS:
            from .packing import Pack
            lens = [int(x) for x in (text_lens.tolist() if torch.is_tensor(text_lens) else text_lens)]
            pack = Pack.from_lens(lens + lens, text_ids.shape[1], image.device)   # 2B sequences: clean | masked
        def text_pass():
            nonlocal text_embeds, mlm_embeds
            with torch.cuda.stream(text_stream):
                _marks.mark("text fwd begin")
                if both_passes:
                    text_embeds, mlm_embeds = self.get_text_embeds_with_masked(text_ids, text_atts, text_ids_masked, pack=pack)
                else:
                    text_embeds = self.get_text_embeds(text_ids, text_atts)
                _marks.mark("text fwd end")

        text_late = False
        if data_source != 'imagenet' and self.batch_passes and image.is_cuda and _TEXT_STREAM_ON:
            main = torch.cuda.current_stream(image.device)
            text_stream = _side_stream(image.device)
            text_stream.wait_stream(main)   # (what the launch stream holds NOW: the previous step; not the ViT pass queued below)
            # _TEXT_AFTER_VIT: the text pass is QUEUED after the ViT pass (it still runs beside it, on its own stream): autograd replays
            # backward nodes in reverse order of creation, so the text tower's backward is then queued BEFORE the ViT's and runs beside
            # the fusion tower's latency-bound backward chain instead of beneath the ViT's chip-filling GEMMs
            text_late = _TEXT_AFTER_VIT and torch.is_grad_enabled()
            if not text_late:
                text_pass()
        _marks.mark("vit fwd begin")
        if self.batch_passes and do_mim and self.do_image_mask:
            B = image.shape[0]
            if ids_mask is None:
                ids_mask = self.vision_encoder.generator.batch(B, image.device)
            ids_mask = ids_mask.to(device=image.device, dtype=torch.bool)
            # one 2B-row ViT pass over the B images: rows [0, B) see them clean, rows [B, 2B) MIM-masked (the token assembly reads
            # every image's patch embedding for both views)
            mim_stream = _mim_stream(image.device) if (_MIM_STREAM and image.is_cuda) else None
            both, _, _ = self.get_vision_embeds(image, do_mask=True, split_stream=mim_stream,
                                                ids_mask=torch.cat([torch.zeros_like(ids_mask), ids_mask], dim=0))
            if isinstance(both, tuple):
                image_embeds, image_embeds_masked = both
            else:
                mim_stream = None
                image_embeds, image_embeds_masked = both[:B], both[B:]
            from .xfm import _ones_mask
            image_atts = _ones_mask(image_embeds)
        else:
            image_embeds, image_atts = self.get_vision_embeds(image)
        _marks.mark("vit fwd end")
        if text_late:
            text_pass()
        if data_source != 'imagenet':
            if text_stream is not None:  # re-join: the text features are consumed on the main stream from here on
                main.wait_stream(text_stream)
                text_embeds, mlm_embeds = _rejoin(text_embeds, main, text_stream), _rejoin(mlm_embeds, main, text_stream)
            elif self.batch_passes and both_passes:
                text_embeds, mlm_embeds = self.get_text_embeds_with_masked(text_ids, text_atts, text_ids_masked, pack=pack)
            else:
                text_embeds = self.get_text_embeds(text_ids, text_atts)
            if pack is not None:  # text_embeds holds packed rows: the [CLS] row of sequence b is row pack.start[b]
                from .packing import rows_gather
                text_cls = rows_gather(text_embeds, pack.start[:image.shape[0]]).unsqueeze(1)
                image_feat, text_feat = self.get_features(image_embeds, text_cls)
            else:
                image_feat, text_feat = self.get_features(image_embeds, text_embeds)
        loss_itc = loss_itm = loss_mlm = loss_mim = zero
        if ret_itc_loss and data_source != 'imagenet':
            loss_itc = self.get_contrastive_loss(image_feat, text_feat)
            if w is not None:
                loss_itc = loss_itc * w
        if self.batch_passes and ret_match_loss and ret_mlm_loss and data_source != 'imagenet':
            loss_itm, loss_mlm = self.get_matching_and_fuse_mlm_loss(image_embeds, image_atts, image_feat, text_ids, text_atts,
                                                                     text_feat, text_embeds, text_ids_masked, masked_pos,
                                                                     masked_ids, neg_idx=neg_idx, mlm_embeds=mlm_embeds, pack=pack)
            if w is not None:
                loss_itm, loss_mlm = loss_itm * w, loss_mlm * w
        else:
            if ret_match_loss and data_source != 'imagenet':
                loss_itm = self.get_matching_loss(image_embeds, image_atts, image_feat, text_ids, text_atts, text_feat,
                                                  text_embeds=text_embeds, neg_idx=neg_idx)
                if w is not None:
                    loss_itm = loss_itm * w
            if ret_mlm_loss and data_source != 'imagenet':
                loss_mlm = self.get_fuse_mlm_loss(text_ids_masked, text_atts, image_embeds, image_atts, masked_pos, masked_ids)
                if w is not None:
                    loss_mlm = loss_mlm * w
        if ret_mim_loss:
            if image_embeds_masked is None:
                image_embeds_masked, _, ids_mask = self.get_vision_embeds(image, do_mask=self.do_image_mask, ids_mask=ids_mask)
            if do_mim:
                ms = mim_stream
                if ms is not None:   # the masked view lives on its own stream: its loss too; the main stream joins once, here
                    cur = torch.cuda.current_stream(image.device)
                    ms.wait_stream(cur)   # (the clean view's embeddings are the target)
                    with torch.cuda.stream(ms):
                        loss_mim = self.get_mim_loss(image_embeds_masked, image_embeds, ids_mask)
                    cur.wait_stream(ms)
                    loss_mim.record_stream(cur)
                else:
                    loss_mim = self.get_mim_loss(image_embeds_masked, image_embeds, ids_mask)
            if w is not None:
                loss_mim = loss_mim * w
        return {'loss_itc': loss_itc, 'loss_itm': loss_itm, 'loss_mlm': loss_mlm, 'loss_mim': loss_mim,
                'loss_bbox': zero, 'loss_giou': zero}

    def forward_text(self, text_ids=None, text_atts=None, text_ids_masked=None, masked_pos=None, masked_ids=None):
        return {'loss_mlm': self.get_mlm_loss(text_ids_masked, text_atts, None, None, masked_pos, masked_ids)}

    def forward(self, image=None, text_ids=None, text_atts=None, text_ids_masked=None, masked_pos=None, masked_ids=None,
                **kw):
        if image is None:
            return self.forward_text(text_ids, text_atts, text_ids_masked, masked_pos, masked_ids)
        return self.forward_multimodal(image, text_ids, text_atts, text_ids_masked, masked_pos, masked_ids, **kw)
