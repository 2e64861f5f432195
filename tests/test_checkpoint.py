"""Checkpoint key surgery (xfm_amd.xfm.load_pretrained) against what the reference's own load_pretrained (xfm.py:408-468) did to
the same synthetic pre-training checkpoint (tools/oracle/gen_golden.py::gen_checkpoint).  Host logic only: runs without a GPU."""
import os

import pytest
import torch

from golden_util import load, state_from_spec


def _cfg():
    return {"use_beit_v2": True, "image_res": 224, "patch_size": 16, "local_attn_depth": -1, "text_encoder": "roberta-base",
            "text_num_hidden_layers": 2, "text_fusion_start_at": 2, "fusion_num_hidden_layers": 2, "fusion_fusion_start_at": 0,
            "embed_dim": 256, "temp": 0.07, "learnable_temp": True, "max_temp": 0.5, "min_temp": 0.001}


def test_pretrain_checkpoint_into_retrieval_model(tmp_path):
    from xfm_amd.model_retrieval import XFMForRetrieval
    from xfm_amd.xfm import load_pretrained
    _, meta = load("checkpoint_surgery")
    sd = state_from_spec(meta["pretrain_spec"])
    path = os.path.join(tmp_path, "ckpt.th")
    torch.save({"model": sd, "epoch": 3}, path)
    m = XFMForRetrieval(_cfg())
    out = load_pretrained(m, path, _cfg(), is_eval=False, load_text=True)
    assert sorted(out.keys()) == meta["keys"]
    for k, v in out.items():  # every tensor is the checkpoint tensor of its source key
        assert abs(float(v.double().sum()) - meta["sums"][k]) <= 1e-6 * max(1.0, abs(meta["sums"][k])), k
    msg = m.load_state_dict(out, strict=False)
    assert sorted(msg.missing_keys) == meta["missing"] and sorted(msg.unexpected_keys) == meta["unexpected"]
    # is_eval returns the raw state_dict
    raw = load_pretrained(m, path, _cfg(), is_eval=True)
    assert set(raw.keys()) == set(meta["pretrain_spec"].keys())


def test_resolution_change_is_refused(tmp_path):
    from xfm_amd.model_retrieval import XFMForRetrieval
    from xfm_amd.xfm import load_pretrained
    _, meta = load("checkpoint_surgery")
    sd = state_from_spec(meta["pretrain_spec"])
    path = os.path.join(tmp_path, "ckpt.th")
    torch.save({"model": sd}, path)
    cfg = dict(_cfg(), image_res=384)
    m = XFMForRetrieval(cfg)
    with pytest.raises(NotImplementedError, match="interp2d"):
        load_pretrained(m, path, cfg, load_text=True)
