"""Test-only import shim for the Python reference under /root/reference.

This file is TEST INFRASTRUCTURE.  It runs only in the build container (where
/root/reference exists) and only from tools/oracle/gen_golden.py, to (a) check
the CPU restatement in oracle/ against the real reference and (b) emit the golden
vectors committed under tests/golden/.  It never travels to the product path and
nothing in xfm_amd/ imports it.

Why a shim is needed at all: the reference pins
transformers==4.12.5 and imports timm / torchvision, none of which match this
image (transformers 5.x, no timm, no torchvision).  Every cure below is an import
stub or an API alias; no reference source is copied or modified.
"""
import importlib.machinery
import json
import os
import sys
import tempfile
import types

REF_ROOT = os.environ.get("XFM_REFERENCE_ROOT", "/root/reference")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    m.__path__ = []  # behave like a package so sub-imports resolve through sys.modules
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    """Make `import models` (the reference package) work.  Idempotent."""
    if getattr(install, "_done", False):
        return
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference tree not found at {REF_ROOT}; the shim only runs in the build container")

    import torch
    import torch.nn as nn
    import transformers  # must be imported BEFORE the torchvision stub (its availability probe walks specs)
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu

    # --- transformers 4.12 names that moved / vanished in 5.x -------------------------------------
    for name in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(mu, name):
            setattr(mu, name, getattr(pu, name))
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        def find_pruneable_heads_and_indices(heads, n_heads, head_size, already_pruned_heads):
            mask = torch.ones(n_heads, head_size)
            heads = set(heads) - already_pruned_heads
            for head in heads:
                head = head - sum(1 if h < head else 0 for h in already_pruned_heads)
                mask[head] = 0
            mask = mask.view(-1).contiguous().eq(1)
            index = torch.arange(len(mask))[mask].long()
            return heads, index
        mu.find_pruneable_heads_and_indices = find_pruneable_heads_and_indices

    PreTrainedModel = mu.PreTrainedModel
    if not getattr(PreTrainedModel, "_xfm_shimmed", False):
        _orig_init_weights = PreTrainedModel.init_weights

        def init_weights(self, *a, **kw):
            # 4.12-style ctor calls self.init_weights(); 5.x needs post_init() bookkeeping first.
            if not hasattr(self, "all_tied_weights_keys"):
                return self.post_init()
            return _orig_init_weights(self, *a, **kw)

        PreTrainedModel.init_weights = init_weights
        if not hasattr(PreTrainedModel, "get_head_mask"):
            def get_head_mask(self, head_mask, num_hidden_layers, is_attention_chunked=False):
                assert head_mask is None, "shim supports head_mask=None only"
                return [None] * num_hidden_layers
            PreTrainedModel.get_head_mask = get_head_mask
        PreTrainedModel._xfm_shimmed = True

    # file_utils decorators used by the reference at import time
    import transformers.file_utils as fu
    for name in ("add_code_sample_docstrings", "add_start_docstrings", "add_start_docstrings_to_model_forward",
                 "replace_return_docstrings"):
        if not hasattr(fu, name):
            def _deco_factory(*a, **kw):
                def deco(fn):
                    return fn
                return deco
            setattr(fu, name, _deco_factory)

    # --- timm (absent) ----------------------------------------------------------------------------
    def drop_path(x, drop_prob: float = 0., training: bool = False):
        if drop_prob == 0. or not training:
            return x
        keep_prob = 1 - drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        random_tensor = keep_prob + torch.rand(shape, dtype=x.dtype, device=x.device)
        random_tensor.floor_()
        return x.div(keep_prob) * random_tensor

    class DropPath(nn.Module):
        def __init__(self, drop_prob=None):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            return drop_path(x, self.drop_prob, self.training)

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
        return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)

    def register_model(fn):
        return fn

    def _cfg(url='', **kwargs):
        return {'url': url, **kwargs}

    _stub("timm")
    _stub("timm.models")
    _stub("timm.models.layers", drop_path=drop_path, to_2tuple=to_2tuple, trunc_normal_=trunc_normal_,
          DropPath=DropPath)
    _stub("timm.models.registry", register_model=register_model)
    class PatchEmbed(nn.Module):
        """timm.models.vision_transformer.PatchEmbed (third-party, absent here; reference pins timm in requirements.txt):
        Conv2d(in_chans, embed_dim, kernel=stride=patch) -> flatten(2).transpose(1, 2); used by models/vit.py:8,147."""

        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
            super().__init__()
            self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
            self.num_patches = (img_size // patch_size) ** 2
            self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

        def forward(self, x):
            return self.proj(x).flatten(2).transpose(1, 2)

    _stub("timm.models.vision_transformer", _cfg=_cfg, PatchEmbed=PatchEmbed)
    _stub("timm.models.helpers", load_pretrained=None)
    _stub("timm.data")
    _stub("timm.data.constants", IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406), IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225),
          IMAGENET_INCEPTION_MEAN=(0.5, 0.5, 0.5), IMAGENET_INCEPTION_STD=(0.5, 0.5, 0.5))

    # --- torchvision (absent): only box_area is touched at import time ---------------------------------
    def box_area(boxes):
        return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])

    _stub("torchvision")
    _stub("torchvision.ops")
    _stub("torchvision.ops.boxes", box_area=box_area)

    # --- reference package path -------------------------------------------------------------------
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # models.model_vqkd drags in CLIP/torchvision.transforms/ftfy; the feature flag is never set.
    import importlib
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF_ROOT, "models")]
    pkg.__spec__ = importlib.machinery.ModuleSpec("models", loader=None, is_package=True)
    sys.modules["models"] = pkg
    _stub("models.model_vqkd", vqkd_encoder_base_decoder_3x768x12_clip=None)
    # now run the real models/__init__.py body (3 imports) inside the prepared package
    xfm_mod = importlib.import_module("models.xfm")
    pkg.XFMBase = xfm_mod.XFMBase
    pkg.build_mlp = xfm_mod.build_mlp
    pkg.load_pretrained = xfm_mod.load_pretrained
    install._done = True


ROBERTA_BASE_CONFIG = {
    # public hyper-parameters of roberta-base (the reference reads them from <text_encoder>/config.json, xfm.py:273,528)
    "architectures": ["RobertaForMaskedLM"], "attention_probs_dropout_prob": 0.1, "bos_token_id": 0, "eos_token_id": 2,
    "hidden_act": "gelu", "hidden_dropout_prob": 0.1, "hidden_size": 768, "initializer_range": 0.02,
    "intermediate_size": 3072, "layer_norm_eps": 1e-05, "max_position_embeddings": 514, "model_type": "roberta",
    "num_attention_heads": 12, "num_hidden_layers": 12, "pad_token_id": 1, "type_vocab_size": 1, "vocab_size": 50265,
}


def make_text_encoder_dir(name="roberta-base", overrides=None):
    """A temp dir whose basename contains 'roberta' holding config.json (xfm.py:260,273)."""
    d = os.path.join(tempfile.mkdtemp(prefix="xfm_ref_"), name)
    os.makedirs(d, exist_ok=True)
    cfg = dict(ROBERTA_BASE_CONFIG)
    if overrides:
        cfg.update(overrides)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(cfg, f)
    return d


def pretrain_config(text_layers=12, fusion_layers=12, overrides=None, roberta_overrides=None):
    """The shipped configs/xfm-pt/Pretrain_XBrain_base_4m.yaml keys that the model constructor reads."""
    cfg = {
        "use_beit_v2": True,
        "vision_config": os.path.join(REF_ROOT, "configs/model/config_beit2_base.json"),
        "image_res": 224, "patch_size": 16, "local_attn_depth": -1,
        "text_encoder": make_text_encoder_dir(overrides=roberta_overrides),
        "text_num_hidden_layers": text_layers, "text_fusion_start_at": text_layers,
        "fusion_num_hidden_layers": fusion_layers, "fusion_fusion_start_at": 0,
        "num_masking_patches": 75, "min_num_patches": 16,
        "embed_dim": 256, "temp": 0.07, "learnable_temp": True, "max_temp": 0.5, "min_temp": 0.001,
        "max_words": 30, "max_tokens": 30, "mask_prob": 0.5, "max_masks": 15,
    }
    if overrides:
        cfg.update(overrides)
    return cfg


def init_single_process_group():
    import torch.distributed as dist
    if not dist.is_initialized():
        f = tempfile.NamedTemporaryFile(delete=False)
        f.close()
        dist.init_process_group("gloo", init_method=f"file://{f.name}", rank=0, world_size=1)
