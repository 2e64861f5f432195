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


def test_resolution_change_resamples_relative_position_tables(tmp_path):
    """224 -> 384 px: every block's table goes from 27 x 27 + 3 to 47 x 47 + 3 entries (beit2.py:763-821).  The values are pinned by
    tests/test_oracle_relpos_interp.py against the reference's own function; here: the plumbing through load_pretrained and the
    properties of the resampling (coordinates, exactness on cubics, untouched cls entries)."""
    import numpy as np
    from xfm_amd.beit2 import interpolate_rel_pos_bias, rel_pos_source_coordinates
    from xfm_amd.model_retrieval import XFMForRetrieval
    from xfm_amd.xfm import load_pretrained
    _, meta = load("checkpoint_surgery")
    sd = _shape_only_state(meta["pretrain_spec"])
    path = os.path.join(tmp_path, "ckpt.th")
    torch.save({"model": sd}, path)
    cfg = dict(_cfg(), image_res=384)
    with torch.device("meta"):
        m = XFMForRetrieval(cfg)
    out = load_pretrained(m, path, cfg, load_text=True)
    tabs = [k for k in out if "relative_position_bias_table" in k]
    assert len(tabs) == 12 and all(tuple(out[k].shape) == (47 * 47 + 3, 12) for k in tabs)
    msg = m.load_state_dict(out, strict=False, assign=True)
    assert not [k for k in msg.missing_keys if "relative_position_bias_table" in k]
    # source coordinates: symmetric geometric progression through 0 and +-1 whose extent is the target's half-width
    x = np.asarray(rel_pos_source_coordinates(27, 47))
    assert len(x) == 27 and x[13] == 0 and x[14] == 1 and x[12] == -1 and np.allclose(x, -x[::-1]) and abs(x[-1] - 23.0) < 1e-3
    assert np.all(np.diff(np.diff(x[13:])) > 0)  # spacing grows away from the centre
    # a bicubic interpolating spline reproduces cubic polynomials of the coordinates exactly -> pins orientation and coordinates
    yy, xx = np.meshgrid(x, x, indexing="ij")                      # table entry [j, i] sits at (x[i], y[j])
    poly = lambda X, Y: 0.3 * X - 0.2 * Y + 0.01 * X * X * Y - 0.002 * Y ** 3 + 1.5
    heads = 12
    src = torch.tensor(poly(xx, yy), dtype=torch.float32).reshape(-1, 1).repeat(1, heads) * torch.arange(1, heads + 1)
    extra = torch.arange(3 * heads, dtype=torch.float32).view(3, heads)
    got = interpolate_rel_pos_bias(torch.cat([src, extra]), 47 * 47 + 3, (24, 24))
    assert torch.equal(got[-3:], extra)
    t = np.arange(-23.0, 23.1, 1.0)
    ty, tx = np.meshgrid(t, t, indexing="ij")
    want = torch.tensor(poly(tx, ty), dtype=torch.float32).reshape(-1, 1) * torch.arange(1, heads + 1)
    assert torch.allclose(got[:-3], want, rtol=1e-4, atol=1e-3), float((got[:-3] - want).abs().max())
    # an arbitrary table is reproduced wherever a target offset coincides with a source offset (0, +-1)
    g = torch.Generator().manual_seed(0)
    rnd = torch.randn(27 * 27 + 3, heads, generator=g)
    got = interpolate_rel_pos_bias(rnd, 47 * 47 + 3, (24, 24))
    assert torch.allclose(got[:-3].view(47, 47, heads)[22:25, 22:25], rnd[:-3].view(27, 27, heads)[12:15, 12:15], atol=1e-5)
    assert interpolate_rel_pos_bias(rnd, 27 * 27 + 3, (14, 14)) is rnd  # same grid: untouched


def test_pretrain_checkpoint_into_vqa_model(tmp_path):
    """XFMForVQA.load_pretrained (model_generation.py:61-91) against what the reference's method did with the same checkpoint: which
    parameters got loaded, and from which checkpoint key each came (text tower without its `roberta.` level; the answer decoder = a
    copy of the fusion tower)."""
    from xfm_amd.model_generation import XFMForVQA
    _, meta = load("checkpoint_vqa")
    spec = meta["pretrain_spec"]
    # a checkpoint whose every tensor is filled with its own index: origins are readable from the values
    names = sorted(spec.keys())
    sd = {k: torch.full(spec[k][0], float(i + 1), dtype=_DT[spec[k][1]]) if len(spec[k][0]) else torch.tensor(float(i + 1)).to(_DT[spec[k][1]])
          for i, k in enumerate(names)}
    path = os.path.join(tmp_path, "ckpt.th")
    torch.save({"model": sd, "epoch": 3}, path)
    cfg = dict(_cfg(), pad_token_id=1, decoder_fusion_start_at=0, num_dec_layers=2)
    with torch.device("meta"):
        m = XFMForVQA(cfg)
    ours = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}
    assert ours == meta["vqa_spec"]
    m = m.to_empty(device="cpu")  # real storage without the 300 M-parameter random init; zeroed below
    with torch.no_grad():
        for p in m.parameters():
            p.zero_()
        for b in m.buffers():
            if b.dtype.is_floating_point:
                b.zero_()
    m.load_pretrained(path, cfg, is_eval=False)
    after = m.state_dict()
    loaded = sorted(k for k, v in after.items() if v.dtype.is_floating_point and float(v.double().abs().sum()) != 0.0)
    want = [k for k in meta["loaded"] if after[k].dtype.is_floating_point]
    assert loaded == sorted(want), (set(loaded) ^ set(want))
    for k in loaded:
        src = names[int(round(float(after[k].flatten()[0]))) - 1]
        assert src in meta["origin"][k], (k, src, meta["origin"][k])
