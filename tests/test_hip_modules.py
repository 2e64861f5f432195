"""Module-level parity on a real MI355X: the HIP-backed towers / pre-training step against
  (a) the committed golden vectors of the real reference (tests/golden/*.npz) and
  (b) the CPU oracle run live on the same formula weights and synthetic batches.
Stated tolerances (bf16 MFMA compute, fp32 accumulate/statistics, vs an fp32 reference):
  tower outputs  rel-L2 <= 2e-2 ;
  parameter / input gradients  rel-L2 <= 8e-2 and cosine >= 0.996 per tensor, everywhere.  The full-depth fixtures (12-block ViT,
                               12+12+12-layer step, VQA) also carry the REFERENCE's own mixed-precision floor per tensor (its
                               bf16-autocast gradient against its fp32 gradient -- the reference trains under apex O1); a tensor
                               may exceed 8e-2 only up to 1.3 x that floor (cosine: 1.7 x its 1 - cos).  Worst measured: 12.5 %
                               on fusion layers 0-3, where the reference's autocast gradients are 11-12.5 % from its fp32 ones;
  losses         total loss within 1e-3 rel (north-star bound); ITC/MLM/MIM within 3e-3 each, ITM (12-row 2-way CE) 3e-2.
"""
import json
import os

import pytest
import numpy as np
import torch

pytestmark = pytest.mark.gpu

from golden_util import load, rel_l2, state_from_spec  # noqa: E402
from xfm_amd import synthetic as syn  # noqa: E402

OUT_TOL, GRAD_TOL, COS_TOL = 2e-2, 8e-2, 0.996


def _load_into(module, spec):
    sd = state_from_spec(spec)
    missing, unexpected = module.load_state_dict(sd, strict=True), None
    return sd


def _check_out(z, prefix, t, tol=OUT_TOL):
    err, cos = rel_l2(z, prefix, t.float())
    assert err <= tol and cos >= COS_TOL, f"{prefix}: rel-L2 {err:.3e} cos {cos:.5f}"


FLOOR_ERR, FLOOR_COS = 1.3, 1.7


def _check_grads(z, prefix, module, name_prefix="", skip=("self.key.bias",), min_rms=1e-7, tol=None, cos_tol=None, abs_ok=None, floor=None,
                 floor_err=FLOOR_ERR, floor_cos=FLOOR_COS):
    """`self.key.bias` is skipped: its gradient is analytically zero (softmax is invariant to the per-query constant
    q.b_k), so the reference holds ~1e-9 rounding noise there and a relative comparison is meaningless; it is bounded
    in absolute terms against the query-bias gradient instead.
    floor: key prefix of the fixture's per-tensor mixed-precision floor (tools/oracle/gen_golden.py amp_floor: the reference's own
    bf16-autocast gradient against its fp32 gradient, [rel-L2, cosine]).  A tensor may then exceed the global tolerance only as far
    as the REFERENCE's 16-bit training arithmetic does on that same tensor: rel-L2 <= max(tol, 1.3 x floor), 1 - cos <= max(1 -
    cos_tol, 1.7 x (1 - floor cos)).  Fixtures generated since round 5 carry the floor ON THE PROBE'S OWN ENTRIES (floor[2:4]; the
    reference's autocast noise reads 1.2-1.3 x its whole-tensor figure on the strided probe of the FFN output weights), older ones the
    whole-tensor figure only (floor[0:2]); the margins cover the sampling noise of a 256-entry estimate."""
    bad, n, worst = [], 0, (0.0, 1.0)
    tol, cos_tol = tol or GRAD_TOL, cos_tol or COS_TOL
    params = dict(module.named_parameters())
    # "negligible gradient" floor: tensors whose reference gradient is < 1% of the fixture's median gradient rms (e.g. the
    # Q/K weights of saturated text self-attention layers, which shrink ~5x per layer down to 1e-10) sit at the bf16
    # noise floor in ABSOLUTE terms; a relative comparison there says nothing about the kernels.
    all_rms = sorted((float(z[k[:-6] + "/sq"]) / int(z[k[:-6] + "/n"])) ** 0.5
                     for k in z.files if k.startswith(prefix + "/") and k.endswith("/probe"))
    min_rms = max(min_rms, 1e-2 * all_rms[len(all_rms) // 2])
    for key in [k for k in z.files if k.startswith(prefix + "/") and k.endswith("/probe")]:
        name = key[len(prefix) + 1: -len("/probe")]
        if any(s in name for s in skip):
            continue
        p = params[name_prefix + name]
        g = p.grad
        assert g is not None, name
        rms = (float(z[f"{prefix}/{name}/sq"]) / int(z[f"{prefix}/{name}/n"])) ** 0.5
        if rms < min_rms:
            continue
        err, cos = rel_l2(z, f"{prefix}/{name}", g)
        n += 1
        if int((z[f"{prefix}/{name}/probe"] != 0).sum()) < 8:
            # sparse gradient (embedding rows): the strided probe sees almost nothing -> compare the global L2 norm
            err = abs(float(g.float().norm()) - float(z[f"{prefix}/{name}/sq"]) ** 0.5) / (float(z[f"{prefix}/{name}/sq"]) ** 0.5)
            cos = 1.0
        if abs_ok and name in abs_ok and float((g.float().cpu().reshape(-1)[:1] - float(z[f"{prefix}/{name}/probe"][0])).abs()) <= abs_ok[name]:
            continue  # an ill-conditioned scalar (see the caller): held in absolute terms
        worst = (max(worst[0], err), min(worst[1], cos))
        tol_t, cos_t = tol, cos_tol
        if floor is not None and f"{floor}/{name}" in z.files:
            fl = [float(v) for v in z[f"{floor}/{name}"]]
            fe, fc = fl[2:4] if len(fl) >= 4 else fl[:2]   # the reference's autocast error on the probe's own entries when the fixture has it
            tol_t, cos_t = max(tol, floor_err * fe), min(cos_tol, 1.0 - floor_cos * (1.0 - fc))
        if err > tol_t or cos < cos_t:
            bad.append((name, round(err, 4), round(cos, 5), round(tol_t, 4)))
    print(f"[{prefix}] {n} gradient tensors, worst rel-L2 {worst[0]:.4f}, worst cosine {worst[1]:.5f}")
    assert n > 0
    assert not bad, f"{len(bad)}/{n} gradients out of tolerance: {bad[:12]}"
    for name, p in params.items():
        if name.endswith("self.key.bias") and p.grad is not None:
            q = params[name.replace("self.key.bias", "self.query.bias")].grad
            # bf16 rounding of the per-key dK rows leaves O(2^-9 * |dK| * sqrt(rows)) of noise in their (zero) sum
            assert float(p.grad.abs().max()) <= float(q.abs().max()) + 1e-6, f"{name}: key-bias gradient is not ~0"


def test_state_dict_keys_match_reference():
    from xfm_amd.model_pretrain import XFM
    z, meta = load("pretrain_small")
    cfg = _pretrain_cfg(meta)
    m = XFM(cfg)
    ours = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}
    assert ours == meta["spec"]


def test_beit_tower_vs_golden():
    from xfm_amd.beit2 import VisionTransformer
    z, meta = load("beit_2blk")
    B, depth = meta["B"], meta["depth"]
    m = VisionTransformer(img_size=224, depth=depth, drop_path_rate=0.1)
    _load_into(m, meta["spec"])
    m.cuda().finalize().eval()
    image = syn.gaussian("beit.image", (B, 3, 224, 224)).cuda()
    cot = syn.symmetric("beit.cot", (B, 197, 768), 1.0).cuda()
    y = m(image)
    _check_out(z, "out", y)
    (y.float() * cot).sum().backward()
    _check_grads(z, "grad", m)
    m._arena.zero_grad()
    masks = syn.mim_block_mask(B, 14, 75, seed=7)
    ym, ids = m(image, do_mask=True, ids_mask=masks)
    assert torch.equal(ids.cpu(), masks)
    _check_out(z, "out_masked", ym)
    (ym.float() * cot).sum().backward()
    _check_grads(z, "grad_masked", m)
    # the region call form (beit2.py:467-475; through XFMBase.get_vision_embeds: xfm.py:574-597)
    m._arena.zero_grad()
    idx, atts = syn.region_case(B)
    yr, yfull = m(image, idx_to_group_img=idx.cuda(), image_atts=atts.cuda())
    _check_out(z, "out_region", yr)
    _check_out(z, "out_region_full", yfull)
    cot_r = syn.symmetric("beit.cot_region", tuple(yr.shape), 1.0).cuda()
    ((yr.float() * cot_r).sum() + 0.5 * (yfull.float() * cot).sum()).backward()
    _check_grads(z, "grad_region", m)


def test_beit_tower_at_384px_vs_oracle():
    """The fine-tuning resolution of BASELINE configs[2] (retrieval, 384 px: 24 x 24 patches, 577 tokens, 47 x 47 + 3 relative
    positions): streamed attention keys, the larger relative-position table and its gradient.  No reference fixture at this size; the
    CPU oracle (itself pinned at 224 px) is the checker."""
    from oracle import xfm_oracle as O
    from xfm_amd.beit2 import VisionTransformer
    m = VisionTransformer(img_size=384, depth=1, drop_path_rate=0.0)
    spec = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}
    sd = _load_into(m, spec)
    assert tuple(sd["blocks.0.attn.relative_position_bias_table"].shape) == (47 * 47 + 3, 12)
    m.cuda().finalize().eval()
    image = syn.gaussian("beit384.image", (1, 3, 384, 384))
    cot = syn.symmetric("beit384.cot", (1, 577, 768), 1.0)
    y = m(image.cuda())
    (y.float() * cot.cuda()).sum().backward()
    P = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    ref = O.beit_forward({"v." + k: v for k, v in P.items()}, "v.", image, depth=1)
    (ref * cot).sum().backward()
    err = float((y.float().cpu() - ref).norm() / ref.norm())
    assert err <= OUT_TOL, err
    worst = 0.0
    for name, p in m.named_parameters():
        g = P[name].grad
        if g is None or float(g.norm()) < 1e-6:
            continue
        e = float((p.grad.float().cpu() - g).norm() / g.norm())
        worst = max(worst, e)
        assert e <= GRAD_TOL, (name, e)
    print(f"384 px: output rel-L2 {err:.4f}, worst gradient rel-L2 {worst:.4f}")


def _roberta(layers, fusion_layer):
    from xfm_amd.xroberta import RobertaConfig, RobertaForMaskedLM
    cfg = RobertaConfig(num_hidden_layers=layers, fusion_layer=fusion_layer, encoder_width=768)
    return RobertaForMaskedLM(cfg)


def test_roberta_text_tower_vs_golden():
    z, meta = load("roberta_text_2L")
    B, L = meta["B"], meta["layers"]
    m = _roberta(L, L)
    _load_into(m, meta["spec"])
    m.cuda().finalize().eval()
    b = {k: v.cuda() for k, v in syn.pretrain_batch(B, seed=11, with_image=False).items()}
    h = m.bert(b["text_ids"], attention_mask=b["text_atts"], return_dict=True).last_hidden_state
    _check_out(z, "hidden", h)
    cot = syn.symmetric("roberta.cot", tuple(h.shape), 1.0).cuda()
    (h.float() * cot).sum().backward()
    _check_grads(z, "grad_hidden", m)
    m._arena.zero_grad()
    res = m(b["text_ids_masked"], attention_mask=b["text_atts"], return_dict=True, labels=b["masked_ids"], masked_pos=b["masked_pos"])
    ref = float(z["mlm_loss"])
    assert abs(float(res.loss) - ref) <= 2e-3 * abs(ref), (float(res.loss), ref)
    _check_out(z, "mlm_logits", res.logits)
    res.loss.backward()
    _check_grads(z, "grad_mlm", m)


def test_fusion_tower_vs_golden():
    z, meta = load("fusion_2L")
    B, L = meta["B"], meta["layers"]
    m = _roberta(L, 0)
    _load_into(m, meta["spec"])
    m.cuda().finalize().eval()
    b = {k: v.cuda() for k, v in syn.pretrain_batch(B, seed=12, with_image=False).items()}
    T = b["text_ids"].shape[1]
    emb = syn.gaussian("fusion.encoder_embeds", (B, T, 768), 0.7).cuda().requires_grad_(True)
    img = syn.gaussian("fusion.image_embeds", (B, 197, 768), 0.7).cuda().requires_grad_(True)
    img_atts = torch.ones(B, 197, dtype=torch.long)
    img_atts[1, 150:] = 0
    img_atts[3, 100:] = 0
    img_atts = img_atts.cuda()
    res = m(encoder_embeds=emb, attention_mask=b["text_atts"], encoder_hidden_states=img, encoder_attention_mask=img_atts,
            return_dict=True, labels=b["masked_ids"], masked_pos=b["masked_pos"])
    ref = float(z["mlm_loss"])
    assert abs(float(res.loss) - ref) <= 2e-3 * abs(ref), (float(res.loss), ref)
    res.loss.backward()
    _check_grads(z, "grad_mlm", m)
    for name, t in (("encoder_embeds", emb), ("image_embeds", img)):
        err, cos = rel_l2(z, f"grad_mlm_in/{name}", t.grad)
        assert err <= GRAD_TOL and cos >= COS_TOL, (name, err, cos)
    m._arena.zero_grad()
    emb.grad = img.grad = None
    h = m.bert(encoder_embeds=emb, attention_mask=b["text_atts"], encoder_hidden_states=img, encoder_attention_mask=img_atts,
               return_dict=True).last_hidden_state
    _check_out(z, "hidden", h)
    cot = syn.symmetric("fusion.cot", tuple(h.shape), 1.0).cuda()
    (h.float() * cot).sum().backward()
    _check_grads(z, "grad_hidden", m)
    for name, t in (("encoder_embeds", emb), ("image_embeds", img)):
        err, cos = rel_l2(z, f"grad_hidden_in/{name}", t.grad)
        assert err <= GRAD_TOL and cos >= COS_TOL, (name, err, cos)
    m._arena.zero_grad()
    labels = b["text_ids"].masked_fill(b["text_atts"] == 0, -100)
    res = m(b["text_ids"], attention_mask=b["text_atts"], encoder_hidden_states=img.detach(), encoder_attention_mask=img_atts,
            return_dict=True, labels=labels, is_decoder=True)
    ref = float(z["causal_loss"])
    assert abs(float(res.loss) - ref) <= 2e-3 * abs(ref), (float(res.loss), ref)
    res.loss.backward()
    _check_grads(z, "grad_causal", m)


def test_causal_lm_answer_decoder_vs_golden():
    """RobertaForCausalLM as the VQA answer decoder (model_generation.py:119-128): ragged answers, cross-attention to
    masked question states, reduction='none' and a weighted per-sequence sum -> exercises the per-row CE gradient."""
    from xfm_amd.xroberta import RobertaConfig, RobertaForCausalLM
    z, meta = load("causal_lm_2L")
    B, L, S = meta["B"], meta["L"], meta["S"]
    m = RobertaForCausalLM(RobertaConfig(num_hidden_layers=meta["layers"], fusion_layer=0, encoder_width=768))
    ours = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}
    assert ours == meta["spec"]
    _load_into(m, meta["spec"])
    m.cuda().finalize().eval()
    ids, atts, enc_atts = (torch.tensor(meta[k]).cuda() for k in ("ids", "atts", "enc_atts"))
    enc = syn.gaussian("causal.question_states", (B, S, 768), 0.7).cuda().requires_grad_(True)
    weights = (syn.gaussian("causal.weights", (B,), 1.0).abs() + 0.1).cuda()
    with torch.no_grad():
        full = m(ids, attention_mask=atts, encoder_hidden_states=enc, encoder_attention_mask=enc_atts)
    _check_out(z, "logits", full.logits)
    res = m(ids, attention_mask=atts, encoder_hidden_states=enc, encoder_attention_mask=enc_atts,
            labels=ids.masked_fill(ids == 1, -100), return_dict=True, reduction="none")
    ref_rows = torch.from_numpy(z["loss_rows"])
    assert torch.allclose(res.loss.float().cpu(), ref_rows, rtol=3e-3, atol=3e-3), (res.loss, ref_rows)
    loss = (weights * res.loss).sum() / B
    ref = float(z["loss"])
    assert abs(float(loss) - ref) <= 2e-3 * abs(ref), (float(loss), ref)
    loss.backward()
    _check_grads(z, "grad", m)
    err, cos = rel_l2(z, "grad_in/question_states", enc.grad)
    assert err <= GRAD_TOL and cos >= COS_TOL, (err, cos)


def test_bert_causal_lm_answer_decoder_vs_golden():
    """xbert.BertLMHeadModel: the answer decoder XFMForVQA builds for a bert-named text encoder (model_generation.py:52-54,
    xbert.py:1235-1347) -- the causal_lm_2L case on the BERT stack, fixture from the reference's own class."""
    from xfm_amd.xbert import BertConfig, BertLMHeadModel
    z, meta = load("bert_causal_lm_2L")
    B, L, S = meta["B"], meta["L"], meta["S"]
    m = BertLMHeadModel(BertConfig(num_hidden_layers=meta["layers"], fusion_layer=0, encoder_width=768))
    ours = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}
    assert ours == meta["spec"]
    _load_into(m, meta["spec"])
    m.cuda().finalize().eval()
    ids, atts, enc_atts = (torch.tensor(meta[k]).cuda() for k in ("ids", "atts", "enc_atts"))
    enc = syn.gaussian("causal.question_states", (B, S, 768), 0.7).cuda().requires_grad_(True)
    weights = (syn.gaussian("causal.weights", (B,), 1.0).abs() + 0.1).cuda()
    with torch.no_grad():
        full = m(ids, attention_mask=atts, encoder_hidden_states=enc, encoder_attention_mask=enc_atts)
        assert m(ids, attention_mask=atts, encoder_hidden_states=enc, encoder_attention_mask=enc_atts, return_logits=True).shape == (B, L - 1, 30522)
    _check_out(z, "logits", full.logits)
    res = m(ids, attention_mask=atts, encoder_hidden_states=enc, encoder_attention_mask=enc_atts,
            labels=ids.masked_fill(ids == 0, -100), return_dict=True, reduction="none")
    ref_rows = torch.from_numpy(z["loss_rows"])
    assert torch.allclose(res.loss.float().cpu(), ref_rows, rtol=3e-3, atol=3e-3), (res.loss, ref_rows)
    loss = (weights * res.loss).sum() / B
    ref = float(z["loss"])
    assert abs(float(loss) - ref) <= 2e-3 * abs(ref), (float(loss), ref)
    loss.backward()
    _check_grads(z, "grad", m)
    err, cos = rel_l2(z, "grad_in/question_states", enc.grad)
    assert err <= GRAD_TOL and cos >= COS_TOL, (err, cos)
    # and XFMForVQA picks it for a bert-named text encoder
    from xfm_amd.model_generation import XFMForVQA
    vqa = XFMForVQA({"use_beit_v2": True, "image_res": 224, "patch_size": 16, "local_attn_depth": -1, "text_encoder": "bert-base-uncased",
                     "text_num_hidden_layers": 1, "text_fusion_start_at": 1, "fusion_num_hidden_layers": 1, "fusion_fusion_start_at": 0,
                     "embed_dim": 256, "temp": 0.07, "vision_depth": 1, "pad_token_id": 0, "decoder_fusion_start_at": 0, "num_dec_layers": 1})
    assert type(vqa.text_decoder).__name__ == "BertLMHeadModel" and "text_decoder.cls.predictions.decoder.weight" in vqa.state_dict()


def test_xbert_variant_vs_golden():
    """xbert.BertForMaskedLM (B1): absolute positions / token type 0 / pad 0 embeddings, scores scaled after QK^T in the
    reference, BERT LM head; layer 0 self-only, layer 1 with cross-attention to ragged image tokens."""
    from xfm_amd.xbert import BertConfig, BertForMaskedLM
    z, meta = load("xbert_2L")
    B, L = meta["B"], meta["layers"]
    m = BertForMaskedLM(BertConfig(num_hidden_layers=L, fusion_layer=meta["fusion_layer"], encoder_width=768))
    ours = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}
    assert ours == meta["spec"]
    _load_into(m, meta["spec"])
    m.cuda().finalize().eval()
    b = {k: v.cuda() for k, v in syn.pretrain_batch(B, seed=14, with_image=False, vocab=30522).items()}
    ids, ids_masked = b["text_ids"].clone(), b["text_ids_masked"].clone()
    ids[b["text_atts"] == 0] = 0
    ids_masked[b["text_atts"] == 0] = 0
    img = syn.gaussian("xbert.image_embeds", (B, 197, 768), 0.7).cuda().requires_grad_(True)
    img_atts = torch.ones(B, 197, dtype=torch.long)
    img_atts[2, 120:] = 0
    img_atts = img_atts.cuda()
    with torch.no_grad():
        h = m.bert(ids, attention_mask=b["text_atts"], return_dict=True, mode="text").last_hidden_state
    _check_out(z, "hidden_text", h)
    res = m(ids_masked, attention_mask=b["text_atts"], encoder_hidden_states=img, encoder_attention_mask=img_atts,
            return_dict=True, labels=b["masked_ids"], masked_pos=b["masked_pos"])
    ref = float(z["mlm_loss"])
    assert abs(float(res.loss) - ref) <= 2e-3 * abs(ref), (float(res.loss), ref)
    _check_out(z, "mlm_logits", res.logits)
    res.loss.backward()
    _check_grads(z, "grad_mlm", m)
    err, cos = rel_l2(z, "grad_mlm_in/image_embeds", img.grad)
    assert err <= GRAD_TOL and cos >= COS_TOL, (err, cos)


def test_plain_vit_tower_vs_golden():
    """models/vit.py (row V0): the fused trunk with a constant-ones layer scale, no relative-position bias, qkv bias as one
    parameter, absolute position embedding and the final LayerNorm over every token."""
    from xfm_amd.vit import VisionTransformer
    z, meta = load("vit_2blk")
    m = VisionTransformer(img_size=224, patch_size=16, embed_dim=768, depth=meta["depth"], num_heads=12, mlp_ratio=4, qkv_bias=True,
                          drop_path_rate=0.1)
    ours = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}
    assert ours == meta["spec"]
    _load_into(m, meta["spec"])
    m.cuda().finalize().eval()
    image = syn.gaussian("vit.image", (meta["B"], 3, 224, 224)).cuda()
    y = m(image)
    _check_out(z, "out", y)
    cot = syn.symmetric("vit.cot", tuple(y.shape), 1.0).cuda()
    (y.float() * cot).sum().backward()
    _check_grads(z, "grad", m)


@pytest.mark.parametrize("fixture", ["retrieval_small", "retrieval_384"])
def test_retrieval_model_vs_golden(fixture):
    """model_retrieval.XFMForRetrieval: bare RobertaModel text tower (no LM heads), ITC with duplicated idx (soft labels), ITM with
    is_pretrain=False (the text tower also gets the matching gradient).  retrieval_384: BASELINE configs[2]'s real shape
    (configs/xfm-ft/Retrieval_coco.yaml:18 -- 384 px = 577 image tokens, 40 text tokens) at B = 8, from the real reference."""
    from xfm_amd.model_retrieval import XFMForRetrieval
    z, meta = load(fixture)
    m = XFMForRetrieval(_pretrain_cfg(meta))
    ours = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}
    assert ours == meta["spec"]
    _load_into(m, meta["spec"])
    m.cuda().finalize().eval()
    b = {k: v.cuda() for k, v in syn.pretrain_batch(meta["B"], seed=77, image_res=meta.get("image_res", 224),
                                                    max_tokens=meta.get("max_tokens", 30)).items()}
    assert b["image"].shape[-1] == meta.get("image_res", 224) and b["text_ids"].shape[1] == meta.get("max_tokens", 30)
    idx = torch.tensor(meta["idx"]).cuda()
    itc, itm = m(b["image"], b["text_ids"], b["text_atts"], idx=idx, neg_idx=(meta["image_neg_idx"], meta["text_neg_idx"]))
    ri, rm = float(z["loss_itc"]), float(z["loss_itm"])
    assert abs(float(itc) - ri) <= 3e-3 * max(abs(ri), 1.0) and abs(float(itm) - rm) <= 3e-2 * max(abs(rm), 1.0), (float(itc), ri, float(itm), rm)
    # the sum: 3e-3.  The ITM term is a 2-way CE over 3B = 12 rows fed by bf16 towers; two forward-attention kernels that are equally
    # close to fp64 attention (rel-L2 2.0e-3 both, tools/attn_err_probe.py) move it by 1.4e-3 (0.81007 / 0.81116 against 0.80753), i.e.
    # the sum by 1.9e-3 / 2.35e-3 -- the previous 2e-3 bound sat inside that rounding noise
    assert abs(float(itc + itm) - (ri + rm)) <= 3e-3 * (ri + rm)
    (itc + itm).backward()
    # d loss / d temp = -(1 / temp^2) * sum_ij (p_ij - y_ij) sim_ij: with temp = 0.07 a residual of ~1e-3 between cancelling terms is
    # amplified 204 x, so the bf16 features' 1e-3 similarity error moves this ONE scalar by ~0.02 of 0.21 -- held to 0.05 absolute
    # d loss / d itm bias = mean_i (p_i - y_i): 3B residuals of O(0.3) cancelling to 7e-3 at 384 px; the bf16 towers move each p_i
    # by ~1e-2 (the ITM loss itself is held to 3e-2 above), so this 2-vector is held to 2e-3 absolute
    _check_grads(z, "grad", m, min_rms=1e-6, abs_ok={"temp": 0.05, "itm_head.3.bias": 2e-3})
    # the sampler itself: device-side draws must respect the same-idx exclusion (xfm.py:731-734)
    with torch.no_grad():
        img, _ = m.get_vision_embeds(b["image"])
        txt = m.get_text_embeds(b["text_ids"], b["text_atts"])
        fi, ft = m.get_features(img, txt)
        ini, tni = m.get_hard_negatives(fi, ft, idx=idx)
    for r in range(meta["B"]):
        assert meta["idx"][int(ini[r])] != meta["idx"][r] and meta["idx"][int(tni[r])] != meta["idx"][r]


def _cls_cfg(meta, **kw):
    cfg = dict(_pretrain_cfg(meta))
    cfg.update(kw)
    return cfg


def _beit_checkpoint_config(spec, tmp_path):
    """A BEiT-v2 checkpoint file holding the fixture's vision tower + the vision_config JSON that names it (xfm.py:206-234)."""
    vis = {k[len("vision_encoder."):]: v for k, v in state_from_spec(spec).items() if k.startswith("vision_encoder.")}
    vis["head.weight"], vis["head.bias"] = torch.zeros(1000, 768), torch.zeros(1000)
    ckpt = os.path.join(tmp_path, "beit.pth")
    torch.save({"model": vis}, ckpt)
    vcfg = os.path.join(tmp_path, "config_beit2_base.json")
    with open(vcfg, "w") as f:
        json.dump({"ckpt": ckpt, "vision_width": 768, "patch_size": 16}, f)
    return vcfg


def test_classification_imagenet_branch_vs_golden(tmp_path):
    """BASELINE configs[1] (ImageNet fine-tune, ViT-only path): XFMForClassification built the reference's way -- vision tower loaded
    by load_pretrained_beit2 from a checkpoint file -- then cls + mean-patch features through the deep MLP head."""
    from xfm_amd.model_classification import XFMForClassification
    z, meta = load("classification_imagenet")
    spec = meta["spec"]
    vcfg = _beit_checkpoint_config(spec, tmp_path)
    m = XFMForClassification(_cls_cfg(meta, vision_config=vcfg, task_name="imagenet", num_labels=1000))
    ours = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}
    assert ours == spec
    ref_vis = state_from_spec(spec)
    for k, v in m.vision_encoder.state_dict().items():  # the checkpoint file reached the tower (load_vision_params=True)
        if v.dtype.is_floating_point:
            assert torch.equal(v, ref_vis["vision_encoder." + k]), k
    _load_into(m, spec)
    m.cuda().finalize().eval()
    b = {k: v.cuda() for k, v in syn.pretrain_batch(meta["B"], seed=55).items()}
    targets = torch.tensor(meta["targets"]).cuda()
    pred = m(b["image"], None, None, targets, train=False)
    _check_out(z, "pred_imagenet", pred)
    loss = m(b["image"], None, None, targets, train=True)
    ref = float(z["loss_imagenet"])
    assert abs(float(loss) - ref) <= 2e-3 * abs(ref), (float(loss), ref)
    loss.backward()
    _check_grads(z, "grad_imagenet", m, min_rms=1e-6, floor="floor_imagenet")


def test_classification_imagenet_at_batch_128_vs_oracle(tmp_path):
    """BASELINE configs[1]'s real batch (Imagenet.py:437-492, batch 128 per GPU at 224 px, 12-block tower), forward AND backward:
    predictions, loss and every parameter gradient against `imagenet_cfg.npz` -- the REFERENCE run at this very shape
    (tools/oracle/gen_golden.py --only imagenet_cfg: models.model_classification.XFMForClassification, fp32, vision tower in chunks of
    8 images) -- with the reference's own bf16-autocast floor per tensor; the live CPU oracle is held to the fixture's predictions too.
    M = 25216 token rows through the 256 x 256 GEMMs, the 197-token attention forward / backward pair and the grouped weight gradients:
    the model-level gradient check at the ViT's headline row count."""
    from oracle import xfm_oracle as O
    from xfm_amd.model_classification import XFMForClassification
    z, meta = load("imagenet_cfg")
    spec = meta["spec"]
    m = XFMForClassification(_cls_cfg(meta, vision_config=_beit_checkpoint_config(spec, tmp_path), task_name="imagenet", num_labels=1000))
    sd = _load_into(m, spec)
    m.cuda().finalize().eval()
    B = meta["B"]
    assert B == 128
    image = syn.gaussian("imagenet128.image", (B, 3, 224, 224))
    targets = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(11))
    assert targets.tolist() == meta["targets"]
    with torch.no_grad():
        pred = m(image.cuda(), None, None, targets.cuda(), train=False).float().cpu()
        ref = O.classification_forward(sd, O.default_cfg(vit_depth=12), image, None, None, deep_head=True).float()
    assert pred.shape == ref.shape == (B, 1000)
    from golden_util import check
    check(z, "pred_imagenet", ref, atol=2e-5, rtol=2e-4, what="oracle vs reference at B = 128: ")
    _check_out(z, "pred_imagenet", pred)
    err = float((pred - ref).norm() / ref.norm())
    per_row = ((pred - ref).norm(dim=1) / ref.norm(dim=1)).max()
    assert err <= OUT_TOL and float(per_row) <= 2 * OUT_TOL, (err, float(per_row))
    loss = m(image.cuda(), None, None, targets.cuda(), train=True)
    ref_loss, amp_loss = float(z["loss_imagenet"]), float(z["amp_loss_imagenet"])
    assert abs(float(loss) - ref_loss) <= 1e-3 * abs(ref_loss), (float(loss), ref_loss)
    loss.backward()
    torch.cuda.synchronize()
    print(f"ImageNet B=128: logits rel-L2 {err:.4f}, worst row {float(per_row):.4f}, loss {float(loss):.5f} vs reference {ref_loss:.5f} "
          f"(its bf16 autocast: {amp_loss:.5f})")
    _check_grads(z, "grad", m, min_rms=1e-6, floor="floor")


def test_classification_multimodal_and_text_branches_vs_golden(tmp_path):
    from xfm_amd.model_classification import XFMForClassification
    z, meta = load("classification_mm")
    spec = meta["spec"]
    vis = {k[len("vision_encoder."):]: v for k, v in state_from_spec(spec).items() if k.startswith("vision_encoder.")}
    ckpt = os.path.join(tmp_path, "beit.pth")
    torch.save({"module": vis}, ckpt)
    vcfg = os.path.join(tmp_path, "config_beit2_base.json")
    with open(vcfg, "w") as f:
        json.dump({"ckpt": ckpt, "vision_width": 768, "patch_size": 16}, f)
    m = XFMForClassification(_cls_cfg(meta, vision_config=vcfg, task_name="ve", num_labels=3))
    ours = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}
    assert ours == spec
    _load_into(m, spec)
    m.cuda().finalize().eval()
    b = {k: v.cuda() for k, v in syn.pretrain_batch(meta["B"], seed=55).items()}
    t = torch.tensor(meta["targets"]).cuda()
    loss = m(b["image"], b["text_ids"], b["text_atts"], t, train=True)
    ref = float(z["loss_mm"])
    assert abs(float(loss) - ref) <= 5e-3 * abs(ref), (float(loss), ref)  # 3-way CE over 4 rows fed by bf16 towers
    loss.backward()
    _check_grads(z, "grad_mm", m, min_rms=1e-6, floor="floor_mm")
    m._arena.zero_grad()
    loss = m(None, b["text_ids"], b["text_atts"], t, train=True)
    ref = float(z["loss_text"])
    assert abs(float(loss) - ref) <= 5e-3 * abs(ref), (float(loss), ref)
    loss.backward()
    _check_grads(z, "grad_text", m, min_rms=1e-6)


@pytest.mark.parametrize("fixture", ["vqa_small", "vqa_480"])
def test_vqa_model_vs_golden(fixture):
    """BASELINE configs[3] (VQA fine-tune): XFMForVQA -- question through text + fusion towers, answers through the causal decoder that
    cross-attends to the fused question states; weighted per-answer loss, gradients, and the inference-time answer ranking.
    vqa_480: the configuration's real resolution (480 px = 901 image tokens, 59 x 59 + 3 relative positions), from the real reference."""
    from types import SimpleNamespace as NS
    from xfm_amd.model_generation import XFMForVQA
    z, meta = load(fixture)
    cfg = dict(_pretrain_cfg(meta), pad_token_id=meta["pad_token_id"], decoder_fusion_start_at=meta["dec_fusion_start"],
               num_dec_layers=meta["dec_layers"])
    m = XFMForVQA(cfg)
    ours = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}
    assert ours == meta["spec"]
    _load_into(m, meta["spec"])
    m.cuda().finalize().eval()
    x = syn.vqa_inputs(image_res=meta.get("image_res", 224))
    q = NS(input_ids=x.q_ids.cuda(), attention_mask=x.q_atts.cuda())
    a = NS(input_ids=x.a_ids.cuda(), attention_mask=x.a_atts.cuda())
    c = NS(input_ids=x.c_ids.cuda(), attention_mask=x.c_atts.cuda())
    loss = m(x.image.cuda(), q, a, k=x.k, weights=x.weights.cuda(), train=True)
    ref = float(z["loss_vqa"])
    assert abs(float(loss) - ref) <= 1e-2 * abs(ref), (float(loss), ref)  # sum of 6 sequence NLLs over a 50k vocabulary, bf16 towers
    loss.backward()
    # 16 bf16 layers deep (12 ViT + 2 text + 2 fusion) before the decoder: the softmax-sensitive Q/K gradients of the decoder's last
    # cross-attention sit at rel-L2 0.08-0.10 (cos 0.9956) where the shallower tower fixtures hold 0.08 / 0.996
    _check_grads(z, "grad", m, min_rms=1e-6, floor="floor")
    with torch.no_grad():
        ids, probs = m(x.image.cuda(), q, c, k=x.topk, train=False)
    # question 0: the reference's first-token shortlist is decided by a 25 % probability margin -> the bf16 path must pick the same three
    assert sorted(ids[0].tolist()) == sorted(z["topk_ids"][0].tolist())
    # (questions 1 and 2: the third and fourth candidates are 4-9 % apart, below what bf16 logits resolve, so either shortlist is fine.)
    # Whenever the reference's winner made the shortlist, the sequence-likelihood re-rank must put it first with the same probability:
    # it carries > 0.999 of the mass in the reference.
    won = 0
    for r in range(ids.shape[0]):
        winner = int(z["topk_ids"][r, 0])
        if winner in ids[r].tolist():
            assert int(ids[r, 0]) == winner, (r, ids.tolist(), z["topk_ids"].tolist())
            assert abs(float(probs[r, 0]) - float(z["topk_probs"][r, 0])) < 2e-3
            won += 1
    assert won >= 1 and int(ids[0, 0]) == int(z["topk_ids"][0, 0])


def test_retrieval_evaluation_vs_golden():
    """Retrieval.py:76-184 on the HIP towers: similarities, then the re-ranked ITM scores wherever the reference's top-k choice is
    decided by a margin the bf16 features cannot flip; batching rows per fusion pass and slicing rows over ranks change nothing."""
    from xfm_amd.model_retrieval import XFMForRetrieval
    from xfm_amd.retrieval_eval import encode, evaluation
    z, meta = load("retrieval_eval")
    m = XFMForRetrieval(_pretrain_cfg(meta))
    _load_into(m, meta["spec"])
    m.cuda().finalize().eval()
    x = syn.retrieval_eval_inputs()
    img, ids, atts = x.image.cuda(), x.text_ids.cuda(), x.text_atts.cuda()
    _, fi, _, ft = encode(m, img, ids, atts)
    sims = (fi.float() @ ft.float().t()).cpu().numpy()
    tol = 1e-2
    assert np.abs(sims - z["sims"]).max() < tol, np.abs(sims - z["sims"]).max()
    # (a) the ITM scores of exactly the pairs the reference re-ranked.  (The fixture's similarities lie within 0.09 of each other
    # -- random-formula features are nearly orthogonal -- so WHICH k pairs get re-ranked is decided below bf16 resolution; the
    # shortlist is therefore checked for self-consistency in (b), the scores here on the reference's own shortlist.)
    from xfm_amd.retrieval_eval import _itm_scores
    ie, _, te, _ = encode(m, img, ids, atts)
    for want, transpose in ((z["score_i2t"], False), (z["score_t2i"], True)):
        r, c = np.nonzero(want != -100.0)
        ii, tt = (c, r) if transpose else (r, c)
        ii, tt = torch.from_numpy(ii).cuda(), torch.from_numpy(tt).cuda()
        with torch.no_grad():
            got = _itm_scores(m, ie[ii], te[tt], atts[tt]).cpu().numpy()
        assert np.abs(got - want[r, c]).max() < 5e-2 * max(1.0, float(np.abs(want[r, c]).max())), (got, want[r, c])  # ITM logits of O(1)
    # (b) evaluation(): every row re-ranks the top-k of ITS OWN similarities and stores those pairs' ITM scores
    i2t, t2i = evaluation(m, img, ids, atts, x.k_test, rows_per_pass=4)
    for got, own in ((i2t, sims), (t2i, sims.T)):
        for r in range(got.shape[0]):
            sel = got[r] != -100.0
            assert sel.sum() == x.k_test and own[r][sel].min() >= np.sort(own[r])[::-1][x.k_test - 1] - 1e-6, (r, got[r], own[r])
    r, c = np.nonzero(i2t != -100.0)
    with torch.no_grad():
        direct = _itm_scores(m, ie[torch.from_numpy(r).cuda()], te[torch.from_numpy(c).cuda()], atts[torch.from_numpy(c).cuda()]).cpu().numpy()
    assert np.abs(direct - i2t[r, c]).max() < 2e-2
    one_i2t, one_t2i = evaluation(m, img, ids, atts, x.k_test, rows_per_pass=1)
    assert np.allclose(one_i2t, i2t, atol=2e-2) and np.allclose(one_t2i, t2i, atol=2e-2)
    # two ranks' row slices tile the matrices (Retrieval.py:133-136)
    parts = [evaluation(m, img, ids, atts, x.k_test, rank=r, world=2) for r in range(2)]
    for full, a, b in ((i2t, parts[0][0], parts[1][0]), (t2i, parts[0][1], parts[1][1])):
        owned_a, owned_b = (a != -100.0).any(1), (b != -100.0).any(1)
        assert not (owned_a & owned_b).any() and (owned_a | owned_b).all()
        assert np.allclose(np.where(owned_a[:, None], a, b), full, atol=2e-2)


def test_grounding_model_vs_golden():
    """model_grounding.XFMForGrounding: box regression from the fused [CLS], L1 + GIoU losses and gradients."""
    from xfm_amd.model_grounding import XFMForGrounding
    z, meta = load("grounding_small")
    m = XFMForGrounding(_pretrain_cfg(meta))
    ours = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}
    assert ours == meta["spec"]
    _load_into(m, meta["spec"])
    m.cuda().finalize().eval()
    b = {k: v.cuda() for k, v in syn.pretrain_batch(meta["B"], seed=99).items()}
    target = torch.tensor(meta["target"]).cuda()
    with torch.no_grad():
        coord = m(b["image"], b["text_ids"], b["text_atts"])
    assert np.abs(coord.float().cpu().numpy() - z["coord"]).max() < 1e-2
    coord, l1, giou = m(b["image"], b["text_ids"], b["text_atts"], target_bbox=target)
    assert abs(float(l1) - float(z["loss_bbox"])) < 2e-2 and abs(float(giou) - float(z["loss_giou"])) < 2e-2
    (l1 + giou).backward()
    _check_grads(z, "grad", m, min_rms=1e-6)


def test_grounding_domain_pretrain_model_vs_golden():
    """model_grounding.XFMForGroundingDomainPretrain (model_grounding.py:12-33): five (expression, box) samples over three images through
    one vision pass (`idx_to_group_img`, the region call form's whole-image branch), `is_image`-weighted L1 + GIoU."""
    from xfm_amd.model_grounding import XFMForGroundingDomainPretrain
    z, meta = load("grounding_domain")
    m = XFMForGroundingDomainPretrain(_pretrain_cfg(meta))
    ours = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}
    assert ours == meta["spec"]
    _load_into(m, meta["spec"])
    m.cuda().finalize().eval()
    b = {k: v.cuda() for k, v in syn.pretrain_batch(meta["bs"], seed=98).items()}
    idx, target, is_image = (torch.tensor(meta[k]).cuda() for k in ("idx", "target", "is_image"))
    l1, giou = m(b["image"][:meta["n_images"]], b["text_ids"], b["text_atts"], idx, target, is_image=is_image)
    assert abs(float(l1) - float(z["loss_bbox"])) < 2e-2 and abs(float(giou) - float(z["loss_giou"])) < 2e-2
    (l1 + giou).backward()
    _check_grads(z, "grad", m, min_rms=1e-6)


def test_nlvr_model_vs_golden():
    from xfm_amd.model_nlvr import XFMForNLVR
    z, meta = load("nlvr_small")
    m = XFMForNLVR(_pretrain_cfg(meta))
    ours = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}
    assert ours == meta["spec"]
    _load_into(m, meta["spec"])
    m.cuda().finalize().eval()
    B = meta["B"]
    b = {k: v.cuda() for k, v in syn.pretrain_batch(2 * B, seed=95).items()}
    t = torch.tensor(meta["targets"]).cuda()
    with torch.no_grad():
        pred = m(b["image"], b["text_ids"][:B], b["text_atts"][:B], t, train=False)
    _check_out(z, "pred_nlvr", pred.float())
    loss = m(b["image"], b["text_ids"][:B], b["text_atts"][:B], t, train=True)
    assert abs(float(loss) - float(z["loss_nlvr"])) <= 2e-2 * max(abs(float(z["loss_nlvr"])), 1.0)
    loss.backward()
    _check_grads(z, "grad", m, min_rms=1e-6)


def _pretrain_cfg(meta):
    return {"use_beit_v2": True, "image_res": meta.get("image_res", 224), "patch_size": 16, "local_attn_depth": -1, "text_encoder": "roberta-base",
            "text_num_hidden_layers": meta["text_layers"], "text_fusion_start_at": meta["text_layers"],
            "fusion_num_hidden_layers": meta["fusion_layers"], "fusion_fusion_start_at": 0, "embed_dim": 256, "temp": 0.07,
            "learnable_temp": True, "max_temp": 0.5, "min_temp": 0.001, "vision_depth": meta.get("vit_depth", 12)}


def _pretrain(name, tol=None, cos_tol=None, batch_passes=True, packed=False, floor=None, returns=False):
    from xfm_amd.model_pretrain import XFM
    z, meta = load(name)
    B = meta["B"]
    m = XFM(dict(_pretrain_cfg(meta), batch_passes=batch_passes))
    _load_into(m, meta["spec"])
    m.cuda().finalize().eval()
    seed = meta.get("seed", 1234)
    hb = syn.pretrain_batch(B, seed=seed)
    b = {k: v.cuda() for k, v in hb.items()}
    masks = syn.mim_block_mask(B, 14, 75, seed=seed)
    losses = m(b["image"], b["text_ids"], b["text_atts"], text_ids_masked=b["text_ids_masked"], masked_pos=b["masked_pos"],
               masked_ids=b["masked_ids"], ret_mim_loss=True, data_source="image", ids_mask=masks,
               neg_idx=(meta["image_neg_idx"], meta["text_neg_idx"]), text_lens=hb["text_atts"].sum(1) if packed else None)
    total, ref_total = 0, 0.0
    report = {}
    for k in ("loss_itc", "loss_itm", "loss_mlm", "loss_mim"):
        ref = float(z[k])
        report[k] = (float(losses[k]), ref)
        total = total + losses[k]
        ref_total += ref
    print(json.dumps(report))
    if "floor_loss_itm" in z.files:
        print("reference under bf16 autocast:", {k: float(z["floor_" + k]) for k in report})
    # ITM is a 2-way CE over only 3B = 12 rows fed by bf16 tower outputs: its own tolerance is looser; the north-star
    # bound (total loss within 1e-3 rel) is asserted on the sum.
    ltol = {"loss_itc": 3e-3, "loss_itm": 3e-2, "loss_mlm": 3e-3, "loss_mim": 3e-3}
    for k, (got, ref) in report.items():
        assert abs(got - ref) <= ltol[k] * max(abs(ref), 1.0), report
    assert abs(float(total) - ref_total) <= 1e-3 * ref_total, (float(total), ref_total, report)
    total.backward()
    # position 1 of the position table only sees padded tokens; bias-table rows with tiny grads are skipped by min_rms
    _check_grads(z, "grad", m, min_rms=1e-6, tol=tol, cos_tol=cos_tol, floor=floor)
    unused = set(meta["unused"])
    for n, p in m.named_parameters():
        if n in unused:
            assert float(p._xfm_grad.abs().max()) == 0.0, f"{n} must receive no gradient"
    if returns:
        return m, z, meta, report


def test_pretrain_step_small_vs_golden():
    _pretrain("pretrain_small")


def test_pretrain_step_small_reference_call_order_vs_golden():
    """batch_passes=False: the reference's own sequence of tower calls (2 ViT passes, ITM and MLM fusion passes apart)."""
    _pretrain("pretrain_small", batch_passes=False)


def test_pretrain_step_small_packed_rows_vs_golden():
    """Unpadded token rows in the text / fusion towers (xfm_amd.packing) against the REAL reference's padded step."""
    _pretrain("pretrain_small", packed=True)


def test_pretrain_step_small_mim_view_on_second_stream_vs_golden(monkeypatch):
    """XFM_MIM_STREAM=1: the MIM-masked view as its own ViT pass on a second stream (forward, loss and backward), ordered against the
    clean view's backward by an event -- same losses and gradients as the batched 2B-row pass."""
    import xfm_amd.model_pretrain as mp
    monkeypatch.setattr(mp, "_MIM_STREAM", True)
    _pretrain("pretrain_small")
    _pretrain("pretrain_small", packed=True)


def test_pretrain_step_full_depth_vs_golden():
    """12 + 12 + 12 layers.  Default tolerance (8e-2 / 0.996) per tensor, relaxed only where -- and only as far as -- the reference's
    own bf16-autocast gradients leave its fp32 gradients on that tensor (fixture floor/<name>; DESIGN.md has the per-tower table:
    the lower fusion layers sit at 10-12 % in the reference itself)."""
    _pretrain("pretrain_full", floor="floor")


def test_pretrain_step_full_depth_packed_rows_vs_golden():
    _pretrain("pretrain_full", floor="floor", packed=True)


def _chi2_two_sample(a, b, min_count=40):
    """Two-sample chi-square statistic per degree of freedom over the bins where both histograms together hold >= min_count."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    keep = (a + b) >= min_count
    a, b = a[keep], b[keep]
    ka, kb = np.sqrt(b.sum() / a.sum()), np.sqrt(a.sum() / b.sum())
    return float((((ka * a - kb * b) ** 2) / (a + b)).sum() / max(keep.sum() - 1, 1))


def test_device_mim_mask_sampler_matches_the_reference_generators_distribution():
    """The on-device block-wise mask sampler against statistics of the REAL reference generator (50 000 draws of
    models/masking_generator.py, tools/oracle/gen_mask_stats.py): per-patch mask frequency, the sizes of the accepted blocks, the
    number of grid rows a mask touches, neighbour co-occurrence -- not just `75 of 196 set`."""
    from xfm_amd import functional as Fx
    z, meta = load("mim_mask_stats")
    n_ref, grid, num = meta["n"], meta["grid"], meta["num"]
    n = 20000
    hist = torch.zeros(grid * grid + 1, dtype=torch.int32, device="cuda")
    m = Fx.mim_masks(n, grid, num, meta["min_num"], "cuda", seed=987654321, delta_hist=hist)
    assert m.shape == (n, grid * grid) and bool((m.sum(1) == num).all())
    m2 = Fx.mim_masks(n, grid, num, meta["min_num"], "cuda", seed=987654321)
    assert torch.equal(m, m2), "a launch is a pure function of its seed"
    assert not torch.equal(m, Fx.mim_masks(n, grid, num, meta["min_num"], "cuda", seed=987654322))
    mf = m.float().view(n, grid, grid)
    f_dev, f_ref = mf.mean(0).reshape(-1).cpu().numpy(), z["freq"] / n_ref
    sigma = np.sqrt(f_ref * (1 - f_ref) * (1.0 / n + 1.0 / n_ref))
    zmax = float(np.abs(f_dev - f_ref).max() / sigma.max()), float((np.abs(f_dev - f_ref) / sigma).max())
    print("per-patch frequency: worst |z|", zmax)
    assert zmax[1] < 5.0, zmax
    c_delta = _chi2_two_sample(hist.cpu().numpy(), z["deltas"])
    rows_dev = torch.bincount((mf.sum(2) > 0).sum(1).long(), minlength=grid + 1).cpu().numpy()
    c_rows = _chi2_two_sample(rows_dev, z["rows"])
    print("chi2/dof: accepted-block sizes", c_delta, " rows touched", c_rows)
    assert c_delta < 2.0 and c_rows < 2.5, (c_delta, c_rows)
    pair = np.array([float((mf[:, :, :-1] * mf[:, :, 1:]).mean()), float((mf[:, :-1, :] * mf[:, 1:, :]).mean()),
                     float((mf[:, :-1, :-1] * mf[:, 1:, 1:]).mean())])
    assert np.abs(pair - z["pair"]).max() < 3e-3, (pair, z["pair"])
    # the model's generator object draws on the device and never repeats a batch
    from xfm_amd.beit2 import BlockMaskGenerator
    g = BlockMaskGenerator(14, 75, 16, seed=3)
    a, b = g.batch(8, "cuda"), g.batch(8, "cuda")
    assert a.is_cuda and a.dtype == torch.bool and bool((a.sum(1) == 75).all()) and not torch.equal(a, b)


def test_hard_negative_sampler_and_mask_generator_are_valid_draws():
    from xfm_amd.beit2 import BlockMaskGenerator
    g = BlockMaskGenerator(14, 75, 16, seed=0)
    m = g.batch(32)
    assert m.shape == (32, 196) and bool((m.sum(1) == 75).all())
    from xfm_amd.model_pretrain import XFM
    z, meta = load("pretrain_small")
    model = XFM(_pretrain_cfg(meta)).cuda().finalize()
    f = torch.nn.functional.normalize(torch.randn(16, 256, device="cuda"), dim=-1)
    t = torch.nn.functional.normalize(torch.randn(16, 256, device="cuda"), dim=-1)
    i_neg, t_neg = model.get_hard_negatives(f, t)
    ar = torch.arange(16, device="cuda")
    assert bool((i_neg != ar).all()) and bool((t_neg != ar).all())
