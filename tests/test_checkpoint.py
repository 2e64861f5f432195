"""Checkpoint key surgery (xfm_amd.xfm.load_pretrained) against what the reference's own load_pretrained (xfm.py:408-468) did to
the same synthetic pre-training checkpoint (tools/oracle/gen_golden.py::gen_checkpoint).  Host logic only: runs without a GPU."""
import os

import pytest
import torch

from golden_util import load

_DT = {"float32": torch.float32, "int64": torch.int64, "int32": torch.int32, "bool": torch.bool}


def _shape_only_state(spec):
    """A checkpoint with the spec's keys / shapes / dtypes but one element of storage per tensor (stride-0 views): the key surgery
    only renames and drops entries, so values do not matter and the 2 GB of formula weights need not be generated or written."""
    return {k: (torch.zeros(1, dtype=_DT[dt]).expand(shape) if len(shape) else torch.zeros((), dtype=_DT[dt])) for k, (shape, dt) in spec.items()}


def _cfg():
    return {"use_beit_v2": True, "image_res": 224, "patch_size": 16, "local_attn_depth": -1, "text_encoder": "roberta-base",
            "text_num_hidden_layers": 2, "text_fusion_start_at": 2, "fusion_num_hidden_layers": 2, "fusion_fusion_start_at": 0,
            "embed_dim": 256, "temp": 0.07, "learnable_temp": True, "max_temp": 0.5, "min_temp": 0.001}


def test_pretrain_checkpoint_into_retrieval_model(tmp_path):
    from xfm_amd.model_retrieval import XFMForRetrieval
    from xfm_amd.xfm import load_pretrained
    _, meta = load("checkpoint_surgery")
    sd = _shape_only_state(meta["pretrain_spec"])
    path = os.path.join(tmp_path, "ckpt.th")
    torch.save({"model": sd, "epoch": 3}, path)
    with torch.device("meta"):  # shapes and key names only: no 500 M-parameter random init on the CPU
        m = XFMForRetrieval(_cfg())
    out = load_pretrained(m, path, _cfg(), is_eval=False, load_text=True)
    assert sorted(out.keys()) == meta["keys"]
    src = {k.replace("roberta.", "") if k.startswith("text_encoder.") else k: tuple(shape) for k, (shape, _) in meta["pretrain_spec"].items()}
    for k, v in out.items():  # every entry is the checkpoint tensor of its (renamed) source key
        assert tuple(v.shape) == src[k], k
    msg = m.load_state_dict(out, strict=False, assign=True)
    assert sorted(msg.missing_keys) == meta["missing"] and sorted(msg.unexpected_keys) == meta["unexpected"]
    # is_eval returns the raw state_dict
    raw = load_pretrained(m, path, _cfg(), is_eval=True)
    assert set(raw.keys()) == set(meta["pretrain_spec"].keys())


def test_resolution_change_is_refused(tmp_path):
    from xfm_amd.model_retrieval import XFMForRetrieval
    from xfm_amd.xfm import load_pretrained
    _, meta = load("checkpoint_surgery")
    sd = _shape_only_state(meta["pretrain_spec"])
    path = os.path.join(tmp_path, "ckpt.th")
    torch.save({"model": sd}, path)
    cfg = dict(_cfg(), image_res=384)
    with torch.device("meta"):
        m = XFMForRetrieval(cfg)
    with pytest.raises(NotImplementedError, match="interp2d"):
        load_pretrained(m, path, cfg, load_text=True)
