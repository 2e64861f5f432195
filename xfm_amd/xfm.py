"""XFMBase on the HIP hot path: three towers + heads + ITC / ITM / MLM / MIM losses, behind the reference's interface.

Mirrors `models/xfm.py::XFMBase` (:471-853): same constructor keywords, `get_*` method names and signatures, and
state_dict key names, so the reference's task models / train loops drive it unchanged.  Towers are the HIP-backed
`xfm_amd.beit2.VisionTransformer` and `xfm_amd.xroberta.RobertaForMaskedLM`; all their parameters live in one flat
arena (xfm_amd.arena).  Deliberate MI355X-first departures, none of which changes a value the reference computes:
  * hard negatives are drawn on the device, one categorical draw per row in ONE kernel (xfm_hard_negatives; a batched
    torch.multinomial when `idx` groups positives) instead of 2B host-synchronous `.item()` draws (xfm.py:736-744) -- same
    per-row categorical distribution, no pipeline stall;
  * ITC (similarity + both cross-entropies), F.normalize, the heads' LayerNorm+GELU and the ITM cross-entropy are single HIP
    kernels (xfm_itc_*, xfm_rownorm_*, xfm_layernorm_* with gelu, xfm_ce_*): no vendor-BLAS / ATen softmax launches in the step;
  * the ITM positive (B) and negative (2B) fusion passes (xfm.py:788-793) run as one 3B-row pass;
  * the MIM loss masks with a weight tensor instead of boolean indexing (no device->host sync).
"""
import json
import os

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import functional as Fx
from . import marks as _marks
from .arena import LinearSlot, ParamArena
from .beit2 import _Affine, beit_base_patch16
from .ops import itc_loss, layer_norm, linear_slot, row_normalize, small_ce
from .xroberta import RobertaConfig, RobertaForMaskedLM, _Lin, rowwise

BF16 = torch.bfloat16
# packed fusion rows: exact layout of the hard-negative text block through ONE host read-back per step (default; measured 42.8 ->
# 41.8 ms per step against worst-case room for that block, XFM_PACK_SYNC=0, which needs no sync at all)
_PACK_SYNC = os.environ.get("XFM_PACK_SYNC", "1") != "0"
_HOST_LAYOUT = os.environ.get("XFM_HOST_LAYOUT", "1") != "0"   # A/B knob: 0 = device-side layout ops + two pageable uploads (rounds 2-3)
# cross-attention per image on contiguous query rows (the image-major layout makes them so) instead of the grouped kernels: measured
# the same (9.84 vs 9.67 ms for the fusion encoder's fwd+bwd: the generic kernels size their grid for the fullest image), so off
_LAST_ROWS = os.environ.get("XFM_LAST_LAYER_ROWS", "0") != "0"   # fusion tower: last layer only on the rows ITM / MLM read (measured neutral: off)
_XATTN_RANGES = os.environ.get("XFM_XATTN_RANGES", "0") != "0"
# ITM outside the pre-training step (retrieval fine-tuning, get_matching_loss): project every image's K / V once instead of once per
# row of the 3B (positive | negative image | negative text) stack; XFM_DEDUP_IMAGES=0 replays the reference's stacked copies
_DEDUP_IMAGES = os.environ.get("XFM_DEDUP_IMAGES", "1") != "0"


class AllGather(torch.autograd.Function):
    """all_gather with a slice-only backward (xfm.py:81-101).  backend 'nccl' is RCCL over xGMI on ROCm."""

    @staticmethod
    def forward(ctx, tensor, rank, world_size):
        output = [torch.empty_like(tensor) for _ in range(world_size)]
        dist.all_gather(output, tensor.contiguous())
        ctx.rank, ctx.batch_size = rank, tensor.shape[0]
        return torch.cat(output, 0)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output[ctx.batch_size * ctx.rank: ctx.batch_size * (ctx.rank + 1)], None, None


FORCE_COLLECTIVES = False   # set by RCCLDDPAccelerator(FORCE_COLLECTIVES=True): gather even in a group of one rank (hardware bring-up test)


def allgather(t):
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES):
        return AllGather.apply(t, dist.get_rank(), dist.get_world_size())
    return t


class _Mlp(nn.Module):
    """build_mlp (xfm.py:115-121) with reference key names 0 / 1 / 3."""

    def __init__(self, input_dim, output_dim):
        super().__init__()
        setattr(self, "0", _Lin(input_dim, input_dim * 2, 0.02))
        setattr(self, "1", _Affine(input_dim * 2, 1e-5))
        setattr(self, "3", _Lin(input_dim * 2, output_dim, 0.02))

    def linear_slots(self, prefix):
        l0, l3 = getattr(self, "0"), getattr(self, "3")
        self._s0 = LinearSlot(prefix + "0", [l0.weight], [l0.bias])
        self._s3 = LinearSlot(prefix + "3", [l3.weight], [l3.bias])
        return [self._s0, self._s3]

    def forward(self, x):
        h = layer_norm(linear_slot(x, self._s0), getattr(self, "1"), gelu=True)
        return linear_slot(h, self._s3, out_fp32=True)


def build_mlp(input_dim, output_dim):
    return _Mlp(input_dim, output_dim)


class _DeepMlp(nn.Module):
    """XFMForClassification.build_mlp (model_classification.py:33-48): four Linear-LayerNorm-GELU stages (widths 2x, 4x, 2x, 1x the
    input) and a final Linear, with the reference's nn.Sequential key names 0,1 / 3,4 / 6,7 / 9,10 / 12."""

    def __init__(self, input_dim, output_dim):
        super().__init__()
        widths = [input_dim, input_dim * 2, input_dim * 4, input_dim * 2, input_dim]
        for j in range(4):
            setattr(self, str(3 * j), _Lin(widths[j], widths[j + 1], 0.02))
            setattr(self, str(3 * j + 1), _Affine(widths[j + 1], 1e-5))
        setattr(self, "12", _Lin(input_dim, output_dim, 0.02))

    def linear_slots(self, prefix):
        self._slots = []
        for name in ("0", "3", "6", "9", "12"):
            lin = getattr(self, name)
            self._slots.append(LinearSlot(prefix + name, [lin.weight], [lin.bias]))
        return list(self._slots)

    def forward(self, x):
        for j in range(4):
            x = layer_norm(linear_slot(x, self._slots[j]), getattr(self, str(3 * j + 1)), gelu=True)
        return linear_slot(x, self._slots[4], out_fp32=True)


def _ones_mask(embeds):
    """image_atts of xfm.py:566-571: all ones.  The tensor is tagged so that the fusion tower can skip the key mask of its
    cross-attention altogether (16 mask loads per lane and key chunk, and the masked-score selects) without reading it back."""
    m = torch.ones(embeds.size()[:-1], dtype=torch.long, device=embeds.device)
    m._xfm_all_ones = True
    return m


def _read_json(path, default):
    if path and os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    return dict(default)


def build_vision_encoder(config, load_params=False):
    """xfm.py:124-255, BEiT-v2 branch (the only one a shipped config selects)."""
    if not config.get('use_beit_v2', False):
        raise ValueError("only use_beit_v2 vision encoders are implemented (xfm.py:206-234)")
    vision_config = _read_json(config.get('vision_config'), {"vision_width": 768, "patch_size": 16})
    assert config['patch_size'] == vision_config['patch_size']
    enc = beit_base_patch16(img_size=config['image_res'], drop_rate=0.0, drop_path_rate=0.1, attn_drop_rate=0.0,
                            use_mean_pooling=True, init_scale=0.001, use_rel_pos_bias=True, use_abs_pos_emb=False,
                            init_values=0.1, qkv_bias=True, local_attn_depth=config.get('local_attn_depth', -1),
                            num_masking_patches=config.get('num_masking_patches', 75),
                            min_num_patches=config.get('min_num_patches', 16), depth=config.get('vision_depth', 12))
    if load_params:  # xfm.py:230-232
        from .beit2 import load_pretrained_beit2
        load_pretrained_beit2(enc, vision_config['ckpt'])
    return enc, vision_config['vision_width']


def _text_config(config, cls=RobertaConfig):
    if 'text_config' in config:
        return cls(**config['text_config'])
    path = os.path.join(config.get('text_encoder', ''), 'config.json')
    if os.path.exists(path):
        return cls.from_json_file(path)
    return cls()  # public roberta-base / bert-base-uncased hyper-parameters


def build_text_encoder(config, vision_width, load_text_params=False, use_mlm_loss=False, config_text=None):
    """xfm.py:258-405: 'roberta' in config['text_encoder'] -> xroberta towers, 'bert' -> xbert towers."""
    name = config.get('text_encoder', 'roberta-base')
    # with the MLM loss the text tower carries its LM heads (`.bert` reaches the encoder); fine-tuning models build the bare
    # encoder (xfm.py:345-352, :398-403)
    if 'roberta' in name:
        from .xroberta import RobertaModel
        cfg_cls, model_cls = RobertaConfig, (RobertaForMaskedLM if use_mlm_loss else RobertaModel)
    elif 'bert' in name:
        from .xbert import BertConfig, BertForMaskedLM, BertModel
        cfg_cls, model_cls = BertConfig, (BertForMaskedLM if use_mlm_loss else BertModel)
    else:
        raise ValueError(name)
    if load_text_params:
        raise NotImplementedError("checkpoint loading is outside the hot-path scope; load a state_dict instead")
    if config_text is None:
        config_text = _text_config(config, cfg_cls)
        config_text.num_hidden_layers = config.get('text_num_hidden_layers', 12)
        config_text.fusion_layer = config.get('text_fusion_start_at', config_text.num_hidden_layers // 2)
    config_text.encoder_width = vision_width
    return model_cls(config_text), []


def load_pretrained(model, ckpt_rpath, config, is_eval=False, load_text=False):
    """Checkpoint key surgery of the reference (xfm.py:408-468) for the BEiT-v2 configuration: unwrap `{'model': state_dict}`,
    drop the `relative_position_index` buffers (rebuilt by the model), and -- with `load_text` -- strip the `roberta.` / `bert.`
    level from `text_encoder.*` keys so that a pre-training checkpoint (text tower with LM heads) loads into a fine-tuning model
    (bare encoder).  Returns the state_dict to pass to `load_state_dict(strict=False)`.
    A change of image resolution resamples the relative-position tables (`beit2.interpolate_rel_pos_bias`, beit2.py:753-821; pinned by
    tests/golden/relpos_interp.npz)."""
    checkpoint = torch.load(ckpt_rpath, map_location='cpu')
    state_dict = checkpoint['model'] if 'model' in checkpoint.keys() else checkpoint
    if is_eval:
        return state_dict
    if not config.get('use_beit_v2', False):
        raise NotImplementedError("only the use_beit_v2 branch of load_pretrained (xfm.py:438-449) is built")
    own = model.vision_encoder.state_dict()
    for k in list(state_dict.keys()):
        if k.startswith('vision_encoder.'):
            if 'relative_position_index' in k:
                del state_dict[k]
            elif 'relative_position_bias_table' in k and k[15:] in own and own[k[15:]].shape != state_dict[k].shape:
                from .beit2 import interpolate_rel_pos_bias
                state_dict[k] = interpolate_rel_pos_bias(state_dict[k], own[k[15:]].shape[0], model.vision_encoder.patch_embed.patch_shape)
    if load_text:
        name_to_replace = 'roberta.' if 'roberta' in config['text_encoder'] else 'bert.'
        for key in list(state_dict.keys()):
            if key.startswith('text_encoder.') and name_to_replace in key:
                state_dict[key.replace(name_to_replace, '')] = state_dict[key]
                del state_dict[key]
    return state_dict


class _MimLossFn(torch.autograd.Function):
    """loss = sum_masked (x - t)^2 / max(count * D, 1) [+ mean over the cls rows]; t carries no gradient."""

    @staticmethod
    def forward(ctx, x, t, mask, cls_term):
        x, t = x.to(torch.bfloat16).contiguous(), t.to(torch.bfloat16).contiguous()
        mask_u8 = mask.to(torch.bool).contiguous().view(torch.uint8)
        sums = Fx.mim_loss_fwd(x, t, mask_u8)
        ctx.save_for_backward(x, t, mask_u8, sums)
        ctx.cls_term = cls_term
        D = x.shape[-1]
        loss = sums[0] / (sums[2] * D).clamp(min=1.0)
        if cls_term:
            loss = loss + sums[1] / float(x.shape[0] * D)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, t, mask_u8, sums = ctx.saved_tensors
        dx = Fx.mim_loss_bwd(x, t, mask_u8, sums, g.reshape(1).float().contiguous(), ctx.cls_term)
        return dx, None, None, None


class XFMBase(nn.Module):
    def __init__(self, config=None, load_vision_params=False, load_text_params=False, use_contrastive_loss=False,
                 use_matching_loss=False, use_mlm_loss=False, use_bbox_loss=False, config_text=None):
        super().__init__()
        self.init_params = []
        self.vision_encoder, vision_width = build_vision_encoder(config, load_params=load_vision_params)
        self.text_encoder, init_params = build_text_encoder(config, vision_width=vision_width,
                                                            load_text_params=load_text_params, use_mlm_loss=use_mlm_loss,
                                                            config_text=config_text)
        self.init_params.extend(init_params)
        self.num_text_layers = self.text_encoder.config.fusion_layer
        self.num_cross_layers = self.text_encoder.config.num_hidden_layers - self.num_text_layers
        self.vision_width = vision_width
        self.text_width = self.text_encoder.config.hidden_size
        self.use_vision_tokenizer = config.get('use_vision_tokenizer', False)
        if self.use_vision_tokenizer:
            raise NotImplementedError("VQ-KD visual tokenizer (model_vqkd.py) is outside the hot-path scope")
        if use_contrastive_loss:
            self.embed_dim = config['embed_dim']
            self.vision_proj = _Lin(self.vision_width, self.embed_dim, 0.02)
            self.text_proj = _Lin(self.text_width, self.embed_dim, 0.02)
            self.init_params.extend(['vision_proj.' + n for n, _ in self.vision_proj.named_parameters()])
            self.init_params.extend(['text_proj.' + n for n, _ in self.text_proj.named_parameters()])
            self.learnable_temp = config.get('learnable_temp', True)
            if not self.learnable_temp:
                self.temp = config.get('temp', 0.07)
            else:
                self.temp = nn.Parameter(torch.ones([]) * config['temp'])
            self.init_params.extend(['temp'])
        if use_matching_loss:
            self.itm_head = build_mlp(input_dim=self.text_width, output_dim=2)
            self.init_params.extend(['itm_head.' + n for n, _ in self.itm_head.named_parameters()])
        if use_bbox_loss:
            self.bbox_head = build_mlp(input_dim=self.text_width, output_dim=4)
            self.init_params.extend(['bbox_head.' + n for n, _ in self.bbox_head.named_parameters()])
        config_fusion = _text_config(config)
        config_fusion.num_hidden_layers = config['fusion_num_hidden_layers']
        config_fusion.fusion_layer = config['fusion_fusion_start_at']
        config_fusion.encoder_width = self.vision_width
        self.fusion_layers = config_fusion.num_hidden_layers
        self.text_layers = config['text_num_hidden_layers']
        self.fusion_encoder = RobertaForMaskedLM(config=config_fusion)
        self.detach_text_forMLM = config.get('detach_text_forMLM', True)
        self.mim_cls_only = config.get('mim_cls_only', False)
        if self.vision_width != self.text_width:
            raise NotImplementedError("fusion_proj (vision_width != text_width) is not on the base-model path")
        self._arena = None

    def load_pretrained(self, ckpt_rpath, config, is_eval=False, is_domain_pretrain=False):
        """xfm.py:540-557."""
        if is_domain_pretrain:
            checkpoint = torch.load(ckpt_rpath, map_location='cpu')
            state_dict = checkpoint['model'] if 'model' in checkpoint.keys() else checkpoint
            for key in list(state_dict.keys()):
                if 'visual_encoder' in key:
                    state_dict[key.replace('visual_encoder', 'vision_encoder')] = state_dict[key]
                    del state_dict[key]
        else:
            state_dict = load_pretrained(self, ckpt_rpath, config, is_eval=is_eval, load_text=True)
        msg = self.load_state_dict(state_dict, strict=False)
        if self._arena is not None:
            self._arena.bump()
        return msg

    # ---- arena ----------------------------------------------------------------------------------
    def finalize(self, device=None):
        """Build the flat parameter / gradient arenas on `device` (call after .cuda() / .to(device))."""
        device = device or next(self.parameters()).device
        slots = self.vision_encoder.linear_slots()
        slots += self.text_encoder.linear_slots("text_encoder.")
        slots += self.fusion_encoder.linear_slots("fusion_encoder.")
        if hasattr(self, "vision_proj"):
            self._s_vproj = LinearSlot("vision_proj", [self.vision_proj.weight], [self.vision_proj.bias])
            self._s_tproj = LinearSlot("text_proj", [self.text_proj.weight], [self.text_proj.bias])
            slots += [self._s_vproj, self._s_tproj]
        if hasattr(self, "itm_head"):
            slots += self.itm_head.linear_slots("itm_head.")
        if hasattr(self, "bbox_head"):
            slots += self.bbox_head.linear_slots("bbox_head.")
        if hasattr(self, "cls_head"):  # task models (model_classification.py, model_nlvr.py)
            slots += self.cls_head.linear_slots("cls_head.")
        if hasattr(self, "text_decoder"):  # the answer decoder of model_generation.py
            slots += self.text_decoder.linear_slots("text_decoder.")
        self._arena = ParamArena(self, slots, device)
        self.vision_encoder.attach(self._arena)
        self.text_encoder.attach(self._arena)
        self.fusion_encoder.attach(self._arena)
        if hasattr(self, "text_decoder"):
            self.text_decoder.attach(self._arena)
        return self

    def _ready(self):
        if self._arena is None or not self._arena.attached():
            self.finalize()

    def zero_grad(self, set_to_none=False):
        if self._arena is not None:
            self._arena.zero_grad()
        else:
            super().zero_grad(set_to_none=set_to_none)

    # ---- towers ---------------------------------------------------------------------------------
    def get_vision_embeds(self, image, image_atts=None, idx_to_group_img=None, do_mask=False, ids_mask=None, split_stream=None):
        self._ready()
        if idx_to_group_img is not None:   # fewer images than samples (xfm.py:574-597)
            idx = idx_to_group_img.to(image.device).view(-1)
            if image_atts is None:     # every sample sees its whole image: the tower's output, one copy per sample
                image_embeds_fullatts = self.vision_encoder(image).index_select(0, idx)
                return image_embeds_fullatts, _ones_mask(image_embeds_fullatts)
            assert image_atts.size(0) == idx.size(0)
            image_embeds, image_embeds_fullatts = self.vision_encoder(image, idx_to_group_img=idx, image_atts=image_atts)
            return image_embeds, image_atts, image_embeds_fullatts.index_select(0, idx)
        if do_mask:
            image_embeds, id_masked = self.vision_encoder(image, do_mask=True, ids_mask=ids_mask, split_stream=split_stream)
            if isinstance(image_embeds, tuple):   # two views as two passes (the second one on `split_stream`)
                return image_embeds, _ones_mask(image_embeds[0]), id_masked
            return image_embeds, _ones_mask(image_embeds), id_masked
        image_embeds = self.vision_encoder(image)
        return image_embeds, _ones_mask(image_embeds)

    def get_text_embeds(self, text_ids, text_atts):
        assert text_atts is not None
        self._ready()
        encoder = self.text_encoder.bert if hasattr(self.text_encoder, 'bert') else self.text_encoder  # xfm.py:605
        return encoder(text_ids, attention_mask=text_atts, encoder_hidden_states=None, encoder_attention_mask=None,
                       return_dict=True).last_hidden_state

    def get_text_embeds_with_masked(self, text_ids, text_atts, text_ids_masked, pack=None):
        """get_text_embeds(text_ids) and the DETACHED get_text_embeds(text_ids_masked) of get_fuse_mlm_loss (xfm.py:648-649) as one
        2B-row pass through the text tower: same arithmetic per row, GEMMs twice as tall; the backward only walks the first B
        sequences (`grad_batch`).  Returns (text_embeds, mlm_embeds.detach()).
        `pack` (xfm_amd.packing.Pack over the 2B sequences: clean | masked): the tower runs on unpadded token rows and the result is
        (rows [pack.cap, D], None) -- sequences 0..B-1 are the text embeddings (with gradient), B..2B-1 the masked-text ones."""
        assert self.detach_text_forMLM
        self._ready()
        bs = text_ids.shape[0]
        if pack is not None:
            rows = self.text_encoder.bert(torch.cat([text_ids, text_ids_masked], dim=0), attention_mask=None, encoder_hidden_states=None,
                                          encoder_attention_mask=None, return_dict=True, grad_batch=bs, pack=pack).last_hidden_state
            return rows, None
        both = self.text_encoder.bert(torch.cat([text_ids, text_ids_masked], dim=0), attention_mask=torch.cat([text_atts, text_atts], dim=0),
                                      encoder_hidden_states=None, encoder_attention_mask=None, return_dict=True,
                                      grad_batch=bs).last_hidden_state
        return rowwise(lambda t: t[:bs], both), rowwise(lambda t: t[bs:].detach(), both)

    def get_features(self, image_embeds=None, text_embeds=None):
        out = []
        if image_embeds is not None:
            out.append(row_normalize(linear_slot(image_embeds[:, 0, :], self._s_vproj, out_fp32=True)))
        if text_embeds is not None:
            out.append(row_normalize(linear_slot(text_embeds[:, 0, :], self._s_tproj, out_fp32=True)))
        return out[0] if len(out) == 1 else tuple(out)

    def get_cross_embeds(self, image_embeds, image_atts, text_ids=None, text_embeds=None, text_atts=None, is_pretrain=True,
                         image_index=None):
        """`image_index` (extension, default None = the reference's call): int tensor [rows] -- text row r attends image
        image_embeds[image_index[r]].  The image states are then NOT copied once per row (xfm.py:781-793 stacks 3B of them): every
        layer projects its K / V once per image and the grouped cross-attention kernels serve all the rows of an image from one
        staging; the image gradient comes back summed over those rows.  Same values on every row."""
        self._ready()
        enc = self.fusion_encoder.bert
        kw = {} if image_index is None else {"encoder_batch_index": image_index}
        if text_embeds is None:
            return enc(text_ids, attention_mask=text_atts, encoder_hidden_states=image_embeds,
                       encoder_attention_mask=image_atts, return_dict=True, **kw).last_hidden_state
        encoder_embeds = rowwise(lambda t: t.detach(), text_embeds) if is_pretrain else text_embeds
        return enc(encoder_embeds=encoder_embeds, attention_mask=text_atts, encoder_hidden_states=image_embeds,
                   encoder_attention_mask=image_atts, return_dict=True, **kw).last_hidden_state

    # ---- grounding head ----------------------------------------------------------------------------
    def predict_bbox(self, image_embeds, text_ids, text_atts, text_embeds, is_pretrain=True):
        """xfm.py:843-854: fused [CLS] -> bbox_head -> sigmoid, (cx, cy, w, h) in [0, 1]."""
        assert image_embeds.size(0) == text_ids.size(0) == text_atts.size(0)
        image_atts = torch.ones(image_embeds.shape[:2], dtype=torch.long, device=image_embeds.device)
        output_cls = self.get_cross_embeds(image_embeds, image_atts, text_ids=text_ids, text_atts=text_atts, text_embeds=text_embeds,
                                           is_pretrain=is_pretrain)[:, 0, :]
        return self.bbox_head(output_cls).float().sigmoid()

    def get_bbox_loss(self, output_coord, target_bbox, is_image=None):
        """L1 + GIoU (xfm.py:815-840).  The reference checks for degenerate boxes with `.any()` on the host and then zeroes the GIoU
        term of the whole batch; the same rule here is a device-side select (no sync), evaluated on stand-in boxes when it fires so
        that no NaN reaches the backward."""
        from . import box_ops
        output_coord, target_bbox = output_coord.float(), target_bbox.float()
        loss_bbox = F.l1_loss(output_coord, target_bbox, reduction='none')
        boxes1 = box_ops.box_cxcywh_to_xyxy(output_coord)
        boxes2 = box_ops.box_cxcywh_to_xyxy(target_bbox)
        degenerate = (boxes1[:, 2:] < boxes1[:, :2]).any() | (boxes2[:, 2:] < boxes2[:, :2]).any()
        unit = torch.tensor([0.0, 0.0, 1.0, 1.0], device=boxes1.device).expand_as(boxes1)
        giou = box_ops.paired_generalized_box_iou(torch.where(degenerate, unit, boxes1), torch.where(degenerate, unit, boxes2))
        loss_giou = torch.where(degenerate, torch.zeros_like(giou), 1 - giou)
        if is_image is None:
            num_boxes = target_bbox.size(0)
        else:
            num_boxes = torch.sum(1 - is_image)
            loss_bbox = loss_bbox * (1 - is_image.view(-1, 1))
            loss_giou = loss_giou * (1 - is_image)
        return loss_bbox.sum() / num_boxes, loss_giou.sum() / num_boxes

    # ---- losses ---------------------------------------------------------------------------------
    def get_contrastive_loss(self, image_feat, text_feat, idx=None):
        assert image_feat.size(-1) == self.embed_dim and text_feat.size(-1) == self.embed_dim
        image_feat_all, text_feat_all = allgather(image_feat), allgather(text_feat)
        if idx is None:   # in-batch labels: similarity, both cross-entropies and their backward as one kernel each way
            return itc_loss(image_feat_all, text_feat_all, self.temp)
        # retrieval fine-tuning (xfm.py:705-713): soft labels over the gathered rows that share an image id, in the same kernels
        idx = idx.view(-1, 1)
        assert idx.size(0) == image_feat.size(0)
        idx_all = allgather(idx)
        return itc_loss(image_feat_all, text_feat_all, self.temp, idx=idx_all.reshape(-1))

    def get_hard_negatives(self, image_feat, text_feat, idx=None):
        """Returns device index tensors (image_neg_idx, text_neg_idx), each [B] int64: one kernel -- similarity row, softmax + 1e-5, own
        entry (with `idx`: every entry of the same image id, xfm.py:731-734) zeroed, one inverse-CDF draw per row."""
        from .xroberta import _next_seed
        temp = self.temp.detach().float().reshape(1) if torch.is_tensor(self.temp) else \
            torch.full((1,), float(self.temp), dtype=torch.float32, device=image_feat.device)
        return Fx.hard_negatives(image_feat.detach().float().contiguous(), text_feat.detach().float().contiguous(), temp, _next_seed(),
                                 idx=None if idx is None else idx.reshape(-1))

    def get_matching_loss(self, image_embeds, image_atts, image_feat, text_ids, text_atts, text_feat, idx=None,
                          return_cross_embeds=False, text_embeds=None, is_pretrain=True, neg_idx=None):
        assert text_ids.dim() == 2, "X-Brain uses text_ids for matching."
        if text_embeds is None:
            raise NotImplementedError("matching on text_ids through the fusion embeddings is not used by any XFM task model")
        if neg_idx is None:
            image_neg_idx, text_neg_idx = self.get_hard_negatives(image_feat, text_feat, idx=idx)
        else:
            image_neg_idx = torch.as_tensor(neg_idx[0], dtype=torch.long, device=image_embeds.device)
            text_neg_idx = torch.as_tensor(neg_idx[1], dtype=torch.long, device=image_embeds.device)
        bs = image_feat.size(0)
        # rows [0,B): positives ; [B,2B): (negative image, text) ; [2B,3B): (image, negative text)   xfm.py:781-793
        text_all = rowwise(lambda t: torch.cat([t, t, t.index_select(0, text_neg_idx)], dim=0), text_embeds)
        text_atts_all = torch.cat([text_atts, text_atts, text_atts.index_select(0, text_neg_idx)], dim=0)
        from .functional import attn_grouped_ok
        if _DEDUP_IMAGES and image_embeds.is_cuda and attn_grouped_ok(text_all.shape[1], image_embeds.shape[1]):
            # the 3B rows attend B distinct images: hand the fusion tower the images once and a row -> image index
            ar = torch.arange(bs, device=image_embeds.device)
            cross = self.get_cross_embeds(image_embeds, image_atts, text_embeds=text_all, text_atts=text_atts_all, is_pretrain=is_pretrain,
                                          image_index=torch.cat([ar, image_neg_idx, ar]))[:, 0, :]
        else:
            image_all = torch.cat([image_embeds, image_embeds.index_select(0, image_neg_idx), image_embeds], dim=0)
            image_atts_all = torch.cat([image_atts, image_atts.index_select(0, image_neg_idx), image_atts], dim=0)
            cross = self.get_cross_embeds(image_all, image_atts_all, text_embeds=text_all, text_atts=text_atts_all,
                                          is_pretrain=is_pretrain)[:, 0, :]
        output = self.itm_head(cross)
        dev = image_embeds.device  # built on the device: a host tensor + .to(device) is a blocking pageable copy
        itm_labels = torch.cat([torch.ones(bs, dtype=torch.long, device=dev), torch.zeros(2 * bs, dtype=torch.long, device=dev)], dim=0)
        loss = small_ce(output, itm_labels, n_valid=itm_labels.numel())
        if return_cross_embeds:
            return loss, cross[:bs]
        return loss

    def _matching_and_fuse_mlm_packed(self, image_embeds, image_atts, image_feat, text_feat, text_rows, pack, masked_pos, masked_ids,
                                      idx=None, neg_idx=None):
        """The 4B-row fusion pass on UNPADDED token rows (xfm_amd.packing).  `text_rows` / `pack`: the text tower's packed output over
        2B sequences (clean | masked).  The 4B fusion sequences are positives, (negative image, text), (image, negative text), MLM
        inputs.  Default: the drawn negatives are read back once, the sequences are packed exactly and laid out image by image
        (cross-attention per image on contiguous rows).  XFM_PACK_SYNC=0: no host sync at all -- four blocks in the reference's
        order, worst-case room and device-computed offsets for the block whose lengths follow the device-side draw."""
        from .ops import lm_head_ce
        from .packing import Pack, image_major_fusion_layout, image_major_layout, rows_gather
        if neg_idx is None:
            image_neg_idx, text_neg_idx = self.get_hard_negatives(image_feat, text_feat, idx=idx)
        else:
            image_neg_idx = torch.as_tensor(neg_idx[0], dtype=torch.long, device=image_embeds.device)
            text_neg_idx = torch.as_tensor(neg_idx[1], dtype=torch.long, device=image_embeds.device)
        bs = image_feat.size(0)
        dev = image_embeds.device
        lens, lh = pack.lens[:bs], pack.lens_host[:bs]
        n_rows, t_max = sum(lh), max(lh)
        ar = torch.arange(bs, device=dev)
        self._ready()
        if _PACK_SYNC or neg_idx is not None:
            # Read the drawn negatives back (the one host sync of the step, at the end of the ViT forward).  Every block is then
            # packed exactly (10-15 % fewer fusion rows than worst-case room for the negative-text block), and the 4B sequences
            # are laid out IMAGE BY IMAGE: the queries of one image are contiguous rows, so cross-attention runs as one ragged
            # problem per image on full 16-row tiles (a 17-token sequence fills 1.06 tiles of the 2 it is given on its own).
            if neg_idx is not None:   # given by the caller: host data already
                im, tn = [int(j) for j in neg_idx[0]], [int(j) for j in neg_idx[1]]
            else:
                self.fusion_encoder.roberta.prefetch_cross_kv(image_embeds)   # (second stream: runs while the host waits below)
                _marks.mark("negatives: read-back queued")
                im, tn = torch.stack([image_neg_idx, text_neg_idx]).cpu().tolist()
                _marks.mark("negatives: host has them")
            seq_len = lh + lh + [lh[j] for j in tn] + lh                       # the reference's order: pos | neg img | neg txt | mlm
            seq_img = list(range(bs)) + im + list(range(bs)) + list(range(bs))
            seq_txt = list(range(bs)) + list(range(bs)) + tn + [bs + j for j in range(bs)]   # sequence of the text tower's pack
            # rows of the last layer that are read afterwards: [CLS] of the 3B matching sequences, the M masked positions of the B MLM
            # sequences (padding slots point at position 0).  In the S-row result they sit in the reference's order.
            M = masked_pos.shape[1]
            prune = _LAST_ROWS and not _XATTN_RANGES and M <= 64
            sel_off = list(range(3 * bs)) + [3 * bs + j * M for j in range(bs)]
            sel_len = [1] * (3 * bs) + [M] * bs
            if pack.lens_host is not None and _HOST_LAYOUT:
                # the whole layout (offsets, cross-attention row ranges, the row gather from the text tower's pack) on the host, ONE upload
                src_start = [0]
                for n_tok in pack.lens_host[:-1]:
                    src_start.append(src_start[-1] + n_tok)
                fpack, meta, ranges, gidx, start_of = image_major_fusion_layout(seq_len, seq_img, bs, pack.T, dev, src_start, seq_txt,
                                                                                extra=(seq_txt, seq_img, sel_off, sel_len))
                enc_index = meta[2]
                text_all = rowwise(lambda t: rows_gather(t.detach(), gidx), text_rows)
            else:
                fpack, _, _, meta, ranges = image_major_layout(seq_len, seq_img, bs, pack.T, dev, extra=(seq_txt, seq_img, sel_off, sel_len))
                pos_dev, seq_src, enc_index = meta[0].long(), meta[1].long(), meta[2].contiguous()
                gidx = fpack.gather_index(pack, seq_src)
                text_all = rowwise(lambda t: rows_gather(t.detach(), gidx), text_rows)
                start_of = fpack.start.index_select(0, pos_dev)                     # start row of every sequence, reference order
            out_rows = None
            if prune:
                sel_rows = torch.cat([start_of[:3 * bs], (start_of[3 * bs:, None] + masked_pos.to(torch.int32)).reshape(-1)])
                out_rows = (sel_rows, meta[3].contiguous(), meta[4].contiguous(), M)
            _marks.mark("fusion fwd begin")
            seq = self.fusion_encoder.bert(encoder_embeds=text_all, attention_mask=None, encoder_hidden_states=image_embeds,
                                           encoder_attention_mask=image_atts, return_dict=True, encoder_batch_index=enc_index,
                                           pack=fpack, encoder_row_ranges=ranges if _XATTN_RANGES else None,
                                           output_rows=out_rows).last_hidden_state
            _marks.mark("fusion fwd end")
            if prune:   # the result IS the gathered rows: 3B [CLS] rows, then B x M masked positions
                output = self.itm_head(seq[:3 * bs])
                itm_labels = torch.cat([torch.ones(bs, dtype=torch.long, device=dev), torch.zeros(2 * bs, dtype=torch.long, device=dev)], dim=0)
                loss_itm = small_ce(output, itm_labels, n_valid=itm_labels.numel())
                loss_mlm, _ = lm_head_ce(seq[3 * bs:], self.fusion_encoder.lm_head, masked_ids.reshape(-1), "mean")
                return loss_itm, loss_mlm
        else:
            fpack = Pack.concat([(lens, n_rows, lh), (lens, n_rows, lh), (lens.index_select(0, text_neg_idx), bs * t_max, None),
                                 (lens, n_rows, lh)], pack.T)
            seq_src = torch.cat([ar, ar, text_neg_idx, ar + bs])             # sequence of the text tower's pack each fusion row copies
            gidx = fpack.gather_index(pack, seq_src)
            text_all = rowwise(lambda t: rows_gather(t.detach(), gidx), text_rows)   # is_pretrain: the text states are detached
            enc_index = torch.cat([ar, image_neg_idx, ar, ar]).to(torch.int32)
            seq = self.fusion_encoder.bert(encoder_embeds=text_all, attention_mask=None, encoder_hidden_states=image_embeds,
                                           encoder_attention_mask=image_atts, return_dict=True, encoder_batch_index=enc_index,
                                           pack=fpack).last_hidden_state
            start_of = fpack.start
        output = self.itm_head(rows_gather(seq, start_of[:3 * bs]))
        itm_labels = torch.cat([torch.ones(bs, dtype=torch.long, device=dev), torch.zeros(2 * bs, dtype=torch.long, device=dev)], dim=0)
        loss_itm = small_ce(output, itm_labels, n_valid=itm_labels.numel())
        mlm_index = (start_of[3 * bs:, None] + masked_pos.to(torch.int32)).reshape(-1)   # gather_seq_out_by_pos (xroberta.py:1215-1216)
        mlm_seq = rows_gather(seq, mlm_index)
        loss_mlm, _ = lm_head_ce(mlm_seq, self.fusion_encoder.lm_head, masked_ids.reshape(-1), "mean")
        return loss_itm, loss_mlm

    def get_matching_and_fuse_mlm_loss(self, image_embeds, image_atts, image_feat, text_ids, text_atts, text_feat, text_embeds,
                                       text_ids_masked, masked_pos, masked_ids, idx=None, is_pretrain=True, neg_idx=None,
                                       mlm_embeds=None, pack=None):
        """get_matching_loss (xfm.py:749-802) and get_fuse_mlm_loss (xfm.py:638-656) through ONE 4B-row fusion pass:
        rows [0,3B) are the ITM positives / negatives, rows [3B,4B) the masked-text MLM inputs.  Same arithmetic per
        row as the two separate calls (no op couples rows); larger GEMMs and a third fewer launches."""
        if pack is not None:
            assert is_pretrain, "the packed 4B pass feeds detached text states (pre-training)"
            return self._matching_and_fuse_mlm_packed(image_embeds, image_atts, image_feat, text_feat, text_embeds, pack, masked_pos,
                                                      masked_ids, idx=idx, neg_idx=neg_idx)
        if neg_idx is None:
            image_neg_idx, text_neg_idx = self.get_hard_negatives(image_feat, text_feat, idx=idx)
        else:
            image_neg_idx = torch.as_tensor(neg_idx[0], dtype=torch.long, device=image_embeds.device)
            text_neg_idx = torch.as_tensor(neg_idx[1], dtype=torch.long, device=image_embeds.device)
        bs = image_feat.size(0)
        if mlm_embeds is None:  # (else: already computed, detached, by get_text_embeds_with_masked)
            with torch.set_grad_enabled(torch.is_grad_enabled() and not self.detach_text_forMLM):
                mlm_embeds = self.get_text_embeds(text_ids_masked, text_atts)
            if self.detach_text_forMLM:
                mlm_embeds = mlm_embeds.detach()
        itm_text = rowwise(lambda t: t.detach(), text_embeds) if is_pretrain else text_embeds
        # every fusion row attends to one of the B unique images: project K/V once per image and layer and let the
        # attention kernels gather them by index (the reference re-projects the duplicated image rows 4x, xfm.py:781-793)
        ar = torch.arange(bs, device=image_embeds.device)
        enc_index = torch.cat([ar, image_neg_idx, ar, ar]).to(torch.int32)
        text_all = rowwise(lambda a, b: torch.cat([a, a, a.index_select(0, text_neg_idx), b], dim=0), itm_text, mlm_embeds)
        text_atts_all = torch.cat([text_atts, text_atts, text_atts.index_select(0, text_neg_idx), text_atts], dim=0)
        self._ready()
        seq = self.fusion_encoder.bert(encoder_embeds=text_all, attention_mask=text_atts_all, encoder_hidden_states=image_embeds,
                                       encoder_attention_mask=image_atts, return_dict=True,
                                       encoder_batch_index=enc_index).last_hidden_state
        output = self.itm_head(seq[:3 * bs, 0, :])
        dev = image_embeds.device  # built on the device: a host tensor + .to(device) is a blocking pageable copy
        itm_labels = torch.cat([torch.ones(bs, dtype=torch.long, device=dev), torch.zeros(2 * bs, dtype=torch.long, device=dev)], dim=0)
        loss_itm = small_ce(output, itm_labels, n_valid=itm_labels.numel())
        mlm_seq = self.fusion_encoder.gather_seq_out_by_pos(seq[3 * bs:], masked_pos)
        from .ops import lm_head_ce
        loss_mlm, _ = lm_head_ce(mlm_seq.reshape(-1, mlm_seq.shape[-1]), self.fusion_encoder.lm_head, masked_ids.reshape(-1), "mean")
        return loss_itm, loss_mlm

    def get_mlm_loss(self, text_ids_masked, text_atts, image_embeds, image_atts, masked_pos, masked_ids):
        self._ready()
        return self.text_encoder(text_ids_masked, attention_mask=text_atts, encoder_hidden_states=image_embeds,
                                 encoder_attention_mask=image_atts, return_dict=True, labels=masked_ids,
                                 masked_pos=masked_pos).loss

    def get_fuse_mlm_loss(self, text_ids_masked, text_atts, image_embeds, image_atts, masked_pos, masked_ids):
        # the detached text pass needs no autograd graph at all: run it without saving activations
        with torch.set_grad_enabled(torch.is_grad_enabled() and not self.detach_text_forMLM):
            encoder_embeds = self.get_text_embeds(text_ids_masked, text_atts)
        if self.detach_text_forMLM:
            encoder_embeds = encoder_embeds.detach()
        return self.fusion_encoder(encoder_embeds=encoder_embeds, attention_mask=text_atts, encoder_hidden_states=image_embeds,
                                   encoder_attention_mask=image_atts, return_dict=True, labels=masked_ids,
                                   masked_pos=masked_pos).loss

    def get_mim_loss(self, image_embeds_masked, targets, mask_tokens):
        """MSE(masked patches) + MSE(pooled cls), xfm.py:624-635: one fused pass each way, sync-free masked mean."""
        return _MimLossFn.apply(image_embeds_masked, targets.detach(), mask_tokens, not self.mim_cls_only)
