"""Plain pre-LN ViT-B/16 on the HIP hot path, behind the reference's `models/vit.py` interface.

Mirrors vit.py: `Attention.forward` :60-83 (fused qkv Linear WITH bias, `(q @ k^T) * scale`, softmax, proj),
`Block.forward` :100-103 (`x + drop_path(attn(norm1(x)))`, `x + drop_path(mlp(norm2(x)))`, no layer scale),
`VisionTransformer.forward` :177-219 (cls token, learned absolute position embedding, final `norm` over every token).
The block stack runs through the same fused node as the BEiT tower (`beit2._TrunkFn`) with a constant-ones layer scale
and no relative-position bias; the scale is applied to the fp32 scores (vit.py's order).  state_dict keys equal the
reference's: cls_token, pos_embed, patch_embed.proj.*, blocks.i.{norm1,norm2}.*, blocks.i.attn.{qkv,proj}.*,
blocks.i.mlp.{fc1,fc2}.*, norm.*.
"""
import math

import torch
import torch.nn as nn

from . import functional as Fx
from .arena import LinearSlot, OwnsArena, ParamArena
from .beit2 import PatchEmbed, _Affine, _Dense, _TrunkFn
from .ops import linear_slot


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = _Dense(dim, hidden)
        self.fc2 = _Dense(hidden, dim)


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qk_scale=None):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = _Dense(dim, dim * 3, bias=True)
        self.proj = _Dense(dim, dim)


class Block(nn.Module):
    gamma_1 = None  # no layer scale: the fused trunk node substitutes a constant vector of ones
    gamma_2 = None

    def __init__(self, dim, num_heads, mlp_ratio, qk_scale, drop_path, eps):
        super().__init__()
        self.norm1 = _Affine(dim, eps)
        self.attn = Attention(dim, num_heads, qk_scale)
        self.norm2 = _Affine(dim, eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.drop_path_prob = float(drop_path)


class VisionTransformer(OwnsArena, nn.Module):
    """Drop-in for models.vit.VisionTransformer (vit.py:106-219)."""

    _rel_pos = False

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4., qkv_bias=True, qk_scale=None, representation_size=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., norm_layer=None, local_attn_depth=0, layer_norm_eps=1e-6):
        super().__init__()
        unsupported = []
        if not qkv_bias: unsupported.append("qkv_bias=False")
        if drop_rate or attn_drop_rate: unsupported.append("drop_rate/attn_drop_rate")
        if local_attn_depth > 0: unsupported.append("local_attn_depth>0")
        if norm_layer is not None: unsupported.append("norm_layer (LayerNorm eps=1e-6 is built in; pass layer_norm_eps)")
        if (embed_dim // num_heads) != 64: unsupported.append("head_dim != 64")
        if unsupported:
            raise NotImplementedError(f"xfm_amd.vit: unsupported {unsupported} (vit.py:111-162)")
        self.num_features = self.embed_dim = embed_dim
        self.depth, self.num_heads, self.local_attn_depth = depth, num_heads, local_attn_depth
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.num_patch_embed = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.num_pos_embed = self.num_patch_embed + 1
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_pos_embed, embed_dim))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth, device="cpu")]
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qk_scale, dpr[i], layer_norm_eps) for i in range(depth)])
        self.norm = _Affine(embed_dim, layer_norm_eps)
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        self._arena = None
        self._own_arena = False

    @property
    def _final_norm(self):
        return self.norm

    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    def linear_slots(self):
        self._slots = []
        self._slot_patch = LinearSlot("patch_embed", [self.patch_embed.proj.weight], [self.patch_embed.proj.bias])
        self._slot_patch.need_t = False
        out = [self._slot_patch]
        for i, blk in enumerate(self.blocks):
            s = {"qkv": LinearSlot(f"blocks.{i}.qkv", [blk.attn.qkv.weight], [blk.attn.qkv.bias]),
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
        self._ones = torch.ones(self.embed_dim, dtype=torch.float32, device=arena.device)

    def finalize(self, device=None):
        device = device or self.cls_token.device
        self.attach(ParamArena(self, self.linear_slots(), device))
        self._own_arena = True
        return self

    def _ready(self):
        if self._arena is None or not self._arena.attached():
            if self._arena is not None and not self._own_arena:
                raise RuntimeError("parameters were moved after the arena was built; call finalize() again")
            self.finalize()

    def forward(self, x, register_blk=-1, idx_to_group_img=None, image_atts=None, drop_path_scales=None):
        if idx_to_group_img is not None or image_atts is not None or register_blk != -1:
            raise NotImplementedError("grouped-image / local-attention / attention-hook paths (vit.py:188-213) are outside the hot-path scope")
        self._ready()
        B, D = x.shape[0], self.embed_dim
        patches = Fx.patchify(x.float().contiguous(), self.patch_embed.patch_size[0])
        tok = linear_slot(patches, self._slot_patch, x_requires_grad=False, out_fp32=True).view(B, -1, D)
        x0 = torch.cat([self.cls_token.expand(B, -1, -1), tok], dim=1)
        x0 = x0 + self.pos_embed[:, :x0.size(1), :]
        dp = drop_path_scales
        if dp is None and self.training and any(b.drop_path_prob > 0 for b in self.blocks):
            keep = getattr(self, "_dp_keep", None)
            if keep is None or keep.device != x.device:
                keep = 1.0 - torch.tensor([[b.drop_path_prob] * 2 for b in self.blocks], device=x.device).view(-1, 2, 1)
                self._dp_keep = keep
            dp = (torch.rand(len(self.blocks), 2, B, device=x.device) < keep).float() / keep
        return _TrunkFn.apply(x0, self, dp)  # bf16 [B, N, D] = norm(x) on every token
