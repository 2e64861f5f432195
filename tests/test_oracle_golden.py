"""The CPU oracle (oracle/xfm_oracle.py) against golden vectors produced by the real reference
(tools/oracle/gen_golden.py).  fp32, same op order => 1e-5-level agreement.  Runs without a GPU."""
import pytest
import numpy as np
import torch

from oracle import xfm_oracle as O
from xfm_amd import synthetic as syn

from golden_util import check, load, state_from_spec

ATOL = 2e-5
RTOL = 2e-4  # relative to the tensor's rms (gradients of sums over many rows carry larger absolute values)


def _params(spec, requires_grad=True):
    P = state_from_spec(spec)
    for k, v in P.items():
        if v.dtype.is_floating_point and requires_grad:
            v.requires_grad_(True)
    # tied decoder bias (xroberta.py:1322-1323): one tensor under two names
    for k in list(P):
        if k.endswith("decoder.bias"):
            P[k] = P[k[: -len("decoder.bias")] + "bias"]
    return P


def _check_grads(z, prefix, P, name_prefix=""):
    n = 0
    for key in [k for k in z.files if k.startswith(prefix + "/") and k.endswith("/probe")]:
        name = key[len(prefix) + 1: -len("/probe")]
        g = P[name_prefix + name].grad
        assert g is not None, f"oracle produced no grad for {name}"
        check(z, f"{prefix}/{name}", g, ATOL, RTOL, what="grad ")
        n += 1
    assert n > 0


def test_beit_blocks_and_mask_path():
    z, meta = load("beit_2blk")
    B, depth = meta["B"], meta["depth"]
    P = _params(meta["spec"])
    image = syn.gaussian("beit.image", (B, 3, 224, 224))
    cot = syn.symmetric("beit.cot", (B, 197, 768), 1.0)
    y = O.beit_forward(P, "", image, depth=depth)
    check(z, "out", y, ATOL, RTOL)
    (y * cot).sum().backward()
    _check_grads(z, "grad", P)
    for v in P.values():
        v.grad = None
    masks = syn.mim_block_mask(B, 14, 75, seed=7)
    ym = O.beit_forward(P, "", image, depth=depth, ids_mask=masks)
    check(z, "out_masked", ym, ATOL, RTOL)
    (ym * cot).sum().backward()
    _check_grads(z, "grad_masked", P)
    # the region call form (beit2.py:467-475): per-sample pooled cls over a region mask, samples sharing images
    for v in P.values():
        v.grad = None
    idx, atts = syn.region_case(B)
    yr, yfull = O.beit_region_outputs(O.beit_forward(P, "", image, depth=depth), idx, atts)
    check(z, "out_region", yr, ATOL, RTOL)
    check(z, "out_region_full", yfull, ATOL, RTOL)
    cot_r = syn.symmetric("beit.cot_region", tuple(yr.shape), 1.0)
    ((yr * cot_r).sum() + 0.5 * (yfull * cot).sum()).backward()
    _check_grads(z, "grad_region", P)


def test_roberta_text_tower_and_mlm_head():
    z, meta = load("roberta_text_2L")
    B, L = meta["B"], meta["layers"]
    P = _params(meta["spec"])
    b = syn.pretrain_batch(B, seed=11, with_image=False)
    emb = O.roberta_embeddings(P, "roberta.embeddings.", b["text_ids"])
    check(z, "embeddings", emb, ATOL, RTOL)
    h = O.roberta_model(P, "roberta.", input_ids=b["text_ids"], att=b["text_atts"], num_layers=L, fusion_layer=L)
    check(z, "hidden", h, ATOL, RTOL)
    cot = syn.symmetric("roberta.cot", tuple(h.shape), 1.0)
    (h * cot).sum().backward()
    _check_grads(z, "grad_hidden", P)
    for v in P.values():
        v.grad = None
    seq = O.roberta_model(P, "roberta.", input_ids=b["text_ids_masked"], att=b["text_atts"], num_layers=L, fusion_layer=L)
    loss, logits = O.masked_lm_loss(P, "", seq, b["masked_pos"], b["masked_ids"])
    assert abs(float(loss) - float(z["mlm_loss"])) < 1e-4
    check(z, "mlm_logits", logits, ATOL, RTOL)
    loss.backward()
    _check_grads(z, "grad_mlm", P)


def test_fusion_tower_cross_attention_and_causal_decoder():
    z, meta = load("fusion_2L")
    B, L = meta["B"], meta["layers"]
    P = _params(meta["spec"])
    b = syn.pretrain_batch(B, seed=12, with_image=False)
    T = b["text_ids"].shape[1]
    emb = syn.gaussian("fusion.encoder_embeds", (B, T, 768), 0.7).requires_grad_(True)
    img = syn.gaussian("fusion.image_embeds", (B, 197, 768), 0.7).requires_grad_(True)
    img_atts = torch.ones(B, 197, dtype=torch.long)
    img_atts[1, 150:] = 0
    img_atts[3, 100:] = 0
    seq = O.roberta_model(P, "roberta.", att=b["text_atts"], encoder_embeds=emb, enc=img, enc_att=img_atts, num_layers=L,
                          fusion_layer=0)
    loss, _ = O.masked_lm_loss(P, "", seq, b["masked_pos"], b["masked_ids"])
    assert abs(float(loss) - float(z["mlm_loss"])) < 1e-4
    loss.backward()
    _check_grads(z, "grad_mlm", P)
    check(z, "grad_mlm_in/encoder_embeds", emb.grad, ATOL, RTOL)
    check(z, "grad_mlm_in/image_embeds", img.grad, ATOL, RTOL)
    for v in list(P.values()) + [emb, img]:
        v.grad = None
    h = O.roberta_model(P, "roberta.", att=b["text_atts"], encoder_embeds=emb, enc=img, enc_att=img_atts, num_layers=L,
                        fusion_layer=0)
    check(z, "hidden", h, ATOL, RTOL)
    cot = syn.symmetric("fusion.cot", tuple(h.shape), 1.0)
    (h * cot).sum().backward()
    _check_grads(z, "grad_hidden", P)
    check(z, "grad_hidden_in/encoder_embeds", emb.grad, ATOL, RTOL)
    check(z, "grad_hidden_in/image_embeds", img.grad, ATOL, RTOL)
    for v in P.values():
        v.grad = None
    seq = O.roberta_model(P, "roberta.", input_ids=b["text_ids"], att=b["text_atts"], enc=img.detach(), enc_att=img_atts,
                          num_layers=L, fusion_layer=0, causal=True)
    labels = b["text_ids"].masked_fill(b["text_atts"] == 0, -100)
    loss, _ = O.masked_lm_loss(P, "", seq, None, labels, head="lm_cap_head", causal_shift=True)
    assert abs(float(loss) - float(z["causal_loss"])) < 1e-4
    loss.backward()
    _check_grads(z, "grad_causal", P)


def test_causal_lm_answer_decoder():
    z, meta = load("causal_lm_2L")
    B, L, S = meta["B"], meta["L"], meta["S"]
    P = _params(meta["spec"])
    ids, atts, enc_atts = (torch.tensor(meta[k]) for k in ("ids", "atts", "enc_atts"))
    enc = syn.gaussian("causal.question_states", (B, S, 768), 0.7).requires_grad_(True)
    weights = syn.gaussian("causal.weights", (B,), 1.0).abs() + 0.1
    rows, logits = O.causal_lm_loss(P, ids, atts, enc, enc_atts, ids.masked_fill(ids == 1, -100), meta["layers"])
    assert np.allclose(rows.detach().numpy(), z["loss_rows"], rtol=1e-5, atol=1e-4)
    # the reference returns the UNSHIFTED logits in .logits
    loss = (weights * rows).sum() / B
    assert abs(float(loss) - float(z["loss"])) < 1e-4
    loss.backward()
    _check_grads(z, "grad", P)
    check(z, "grad_in/question_states", enc.grad, ATOL, RTOL)


def test_bert_causal_lm_answer_decoder():
    """xbert.BertLMHeadModel (model_generation.py:52-54), fixture from the reference's own class."""
    z, meta = load("bert_causal_lm_2L")
    B, L, S = meta["B"], meta["L"], meta["S"]
    P = _params(meta["spec"])
    ids, atts, enc_atts = (torch.tensor(meta[k]) for k in ("ids", "atts", "enc_atts"))
    enc = syn.gaussian("causal.question_states", (B, S, 768), 0.7).requires_grad_(True)
    weights = syn.gaussian("causal.weights", (B,), 1.0).abs() + 0.1
    rows, logits = O.bert_causal_lm_loss(P, ids, atts, enc, enc_atts, ids.masked_fill(ids == 0, -100), meta["layers"])
    assert np.allclose(rows.detach().numpy(), z["loss_rows"], rtol=1e-5, atol=1e-4)
    check(z, "logits", logits, ATOL, RTOL)
    loss = (weights * rows).sum() / B
    assert abs(float(loss) - float(z["loss"])) < 1e-4
    loss.backward()
    _check_grads(z, "grad", P)
    check(z, "grad_in/question_states", enc.grad, ATOL, RTOL)


def test_xbert_variant():
    z, meta = load("xbert_2L")
    B, L = meta["B"], meta["layers"]
    P = _params(meta["spec"])
    b = syn.pretrain_batch(B, seed=14, with_image=False, vocab=30522)
    ids, ids_masked = b["text_ids"].clone(), b["text_ids_masked"].clone()
    ids[b["text_atts"] == 0] = 0
    ids_masked[b["text_atts"] == 0] = 0
    img = syn.gaussian("xbert.image_embeds", (B, 197, 768), 0.7).requires_grad_(True)
    img_atts = torch.ones(B, 197, dtype=torch.long)
    img_atts[2, 120:] = 0
    check(z, "embeddings", O.bert_embeddings(P, "bert.embeddings.", ids), ATOL, RTOL)
    h = O.bert_model(P, "bert.", ids, b["text_atts"], num_layers=meta["fusion_layer"], fusion_layer=meta["fusion_layer"])
    check(z, "hidden_text", h, ATOL, RTOL)
    seq = O.bert_model(P, "bert.", ids_masked, b["text_atts"], img, img_atts, num_layers=L, fusion_layer=meta["fusion_layer"])
    loss, logits = O.bert_mlm_loss(P, seq, b["masked_pos"], b["masked_ids"])
    assert abs(float(loss) - float(z["mlm_loss"])) < 1e-4
    check(z, "mlm_logits", logits, 1e-4, RTOL)
    loss.backward()
    _check_grads(z, "grad_mlm", P)
    check(z, "grad_mlm_in/image_embeds", img.grad, ATOL, RTOL)


def test_plain_vit_tower():
    z, meta = load("vit_2blk")
    P = _params(meta["spec"])
    image = syn.gaussian("vit.image", (meta["B"], 3, 224, 224))
    y = O.vit_forward(P, "", image, depth=meta["depth"])
    check(z, "out", y, ATOL, RTOL)
    cot = syn.symmetric("vit.cot", tuple(y.shape), 1.0)
    (y * cot).sum().backward()
    _check_grads(z, "grad", P)


@pytest.mark.parametrize("fixture", ["retrieval_small", "retrieval_384"])
def test_retrieval_model(fixture):
    """retrieval_384: the fine-tuning shape of BASELINE configs[2] (384 px, 40 tokens, B = 8) from the real reference."""
    z, meta = load(fixture)
    P = _params(meta["spec"])
    cfg = O.default_cfg(text_layers=meta["text_layers"], fusion_layers=meta["fusion_layers"], vit_depth=meta.get("vit_depth", 12))
    b = syn.pretrain_batch(meta["B"], seed=77, image_res=meta.get("image_res", 224), max_tokens=meta.get("max_tokens", 30))
    idx = torch.tensor(meta["idx"])
    itc, itm = O.retrieval_forward(P, cfg, b, idx, meta["image_neg_idx"], meta["text_neg_idx"], text_prefix="text_encoder.")
    assert abs(float(itc) - float(z["loss_itc"])) < 2e-4 and abs(float(itm) - float(z["loss_itm"])) < 2e-4
    # the captured negatives respect the idx rule: never a pair with the same idx (xfm.py:731-734)
    for r, (i, t) in enumerate(zip(meta["image_neg_idx"], meta["text_neg_idx"])):
        assert meta["idx"][i] != meta["idx"][r] and meta["idx"][t] != meta["idx"][r]
    (itc + itm).backward()
    _check_grads(z, "grad", P)


def test_classification_models():
    z, meta = load("classification_imagenet")
    P = _params(meta["spec"])
    cfg = O.default_cfg(text_layers=meta["text_layers"], fusion_layers=meta["fusion_layers"], vit_depth=meta.get("vit_depth", 12))
    b = syn.pretrain_batch(meta["B"], seed=55)
    pred = O.classification_forward(P, cfg, b["image"], None, None, deep_head=True)
    check(z, "pred_imagenet", pred, 1e-4, RTOL)
    loss = torch.nn.functional.cross_entropy(pred, torch.tensor(meta["targets"]))
    assert abs(float(loss) - float(z["loss_imagenet"])) < 2e-4
    loss.backward()
    _check_grads(z, "grad_imagenet", P)
    unused = set(meta["unused"])
    assert all(P[k].grad is None or float(P[k].grad.abs().max()) == 0.0 for k in unused if k in P and P[k].dtype.is_floating_point)
    z, meta = load("classification_mm")
    P = _params(meta["spec"])
    t = torch.tensor(meta["targets"])
    loss = torch.nn.functional.cross_entropy(O.classification_forward(P, cfg, b["image"], b["text_ids"], b["text_atts"], False), t)
    assert abs(float(loss) - float(z["loss_mm"])) < 2e-4
    loss.backward()
    _check_grads(z, "grad_mm", P)
    for v in P.values():
        v.grad = None
    loss = torch.nn.functional.cross_entropy(O.classification_forward(P, cfg, None, b["text_ids"], b["text_atts"], False), t)
    assert abs(float(loss) - float(z["loss_text"])) < 2e-4
    loss.backward()
    _check_grads(z, "grad_text", P)


@pytest.mark.parametrize("fixture", ["vqa_small", "vqa_480"])
def test_vqa_model_loss_and_answer_ranking(fixture):
    """BASELINE configs[3]: XFMForVQA's weighted answer loss + gradients, and rank_answer's re-ranked shortlist (vqa_480: at the
    configuration's 480 px)."""
    z, meta = load(fixture)
    P = _params(meta["spec"])
    cfg = O.default_cfg(text_layers=meta["text_layers"], fusion_layers=meta["fusion_layers"], vit_depth=meta.get("vit_depth", 12))
    cfg.update(dec_layers=meta["dec_layers"], dec_fusion_start=meta["dec_fusion_start"])
    x = syn.vqa_inputs(image_res=meta.get("image_res", 224))
    loss = O.vqa_train_loss(P, cfg, x.image, x.q_ids, x.q_atts, x.a_ids, x.a_atts, x.k, x.weights, meta["pad_token_id"])
    assert abs(float(loss) - float(z["loss_vqa"])) < 2e-4 * abs(float(z["loss_vqa"])), (float(loss), float(z["loss_vqa"]))
    loss.backward()
    _check_grads(z, "grad", P)
    unused = set(meta["unused"])
    assert all(P[k].grad is None or float(P[k].grad.abs().max()) == 0.0 for k in unused if k in P and P[k].dtype.is_floating_point)
    with torch.no_grad():
        ids, probs, _ = O.vqa_rank_answer(P, cfg, x.image, x.q_ids, x.q_atts, x.c_ids, x.c_atts, x.topk, meta["pad_token_id"])
    assert ids.tolist() == z["topk_ids"].tolist()
    assert np.allclose(probs.numpy(), z["topk_probs"], rtol=1e-3, atol=1e-6)


def test_nlvr_model():
    z, meta = load("nlvr_small")
    P = _params(meta["spec"])
    cfg = O.default_cfg(text_layers=meta["text_layers"], fusion_layers=meta["fusion_layers"], vit_depth=meta.get("vit_depth", 12))
    B = meta["B"]
    b = syn.pretrain_batch(2 * B, seed=95)
    pred = O.nlvr_forward(P, cfg, b["image"], b["text_ids"][:B], b["text_atts"][:B])
    check(z, "pred_nlvr", pred, 1e-4, RTOL)
    loss = torch.nn.functional.cross_entropy(pred, torch.tensor(meta["targets"]))
    assert abs(float(loss) - float(z["loss_nlvr"])) < 2e-4
    loss.backward()
    _check_grads(z, "grad", P)


def test_retrieval_evaluation_rerank_and_recall():
    """Retrieval.py:76-240: the k-test re-rank score matrices against the reference model's, and the recall metrics against a
    hand-worked case (product-side numpy restatement and the oracle's must agree on it too)."""
    z, meta = load("retrieval_eval")
    P = _params(meta["spec"])
    cfg = O.default_cfg(text_layers=meta["text_layers"], fusion_layers=meta["fusion_layers"], vit_depth=meta.get("vit_depth", 12))
    x = syn.retrieval_eval_inputs()
    with torch.no_grad():
        i2t, t2i, sims = O.retrieval_score_matrices(P, cfg, x.image, x.text_ids, x.text_atts, x.k_test)
    assert np.allclose(sims.numpy(), z["sims"], atol=2e-5)
    assert np.allclose(i2t.numpy(), z["score_i2t"], atol=2e-4) and np.allclose(t2i.numpy(), z["score_t2i"], atol=2e-4)
    # hand-worked: 3 images, 4 captions (caption 3 is a second caption of image 0)
    s_i2t = np.array([[0.1, 0.9, 0.2, 0.8],    # image 0: order 1,3,2,0 -> best ground truth (0 or 3) at rank 1
                      [0.7, 0.6, 0.1, 0.0],    # image 1: truth 1 at rank 1
                      [0.0, 0.1, 0.9, 0.3]])   # image 2: truth 2 at rank 0
    s_t2i = np.array([[0.9, 0.1, 0.0], [0.2, 0.1, 0.3], [0.1, 0.2, 0.3], [0.3, 0.2, 0.1]])  # ranks 0, 2, 0, 0
    txt2img, img2txt = [0, 1, 2, 0], [[0, 3], [1], [2]]
    want = {'txt_r1': 100.0 / 3, 'txt_r5': 100.0, 'txt_r10': 100.0, 'img_r1': 75.0, 'img_r5': 100.0, 'img_r10': 100.0}
    from xfm_amd.retrieval_eval import itm_eval
    for fn in (O.itm_eval, itm_eval):
        got = fn(s_i2t, s_t2i, txt2img, img2txt)
        for k, v in want.items():
            assert abs(got[k] - v) < 1e-9, (fn.__module__, k, got[k], v)
        assert abs(got['r_mean'] - ((100.0 / 3 + 200.0) / 3 + (75.0 + 200.0) / 3) / 2) < 1e-9


def test_grounding_domain_pretrain_model():
    """XFMForGroundingDomainPretrain (model_grounding.py:12-33): several samples per image through one vision pass."""
    z, meta = load("grounding_domain")
    P = _params(meta["spec"])
    cfg = O.default_cfg(text_layers=meta["text_layers"], fusion_layers=meta["fusion_layers"], vit_depth=meta.get("vit_depth", 12))
    b = syn.pretrain_batch(meta["bs"], seed=98)
    idx, target, is_image = torch.tensor(meta["idx"]), torch.tensor(meta["target"]), torch.tensor(meta["is_image"])
    coord = O.grounding_forward(P, cfg, b["image"][:meta["n_images"]], b["text_ids"], b["text_atts"], idx_to_group_img=idx)
    l1, giou = O.bbox_loss(coord, target, is_image)
    assert abs(float(l1) - float(z["loss_bbox"])) < 2e-5 and abs(float(giou) - float(z["loss_giou"])) < 2e-5
    (l1 + giou).backward()
    _check_grads(z, "grad", P)


def test_grounding_model_and_box_losses():
    z, meta = load("grounding_small")
    P = _params(meta["spec"])
    cfg = O.default_cfg(text_layers=meta["text_layers"], fusion_layers=meta["fusion_layers"], vit_depth=meta.get("vit_depth", 12))
    b = syn.pretrain_batch(meta["B"], seed=99)
    target = torch.tensor(meta["target"])
    coord = O.grounding_forward(P, cfg, b["image"], b["text_ids"], b["text_atts"])
    assert np.allclose(coord.detach().numpy(), z["coord"], atol=2e-5)
    l1, giou = O.bbox_loss(coord, target)
    assert abs(float(l1) - float(z["loss_bbox"])) < 2e-5 and abs(float(giou) - float(z["loss_giou"])) < 2e-5
    (l1 + giou).backward()
    _check_grads(z, "grad", P)
    # the product path's paired (diagonal-only) GIoU and its sync-free degenerate-box rule agree with the reference formulation
    from xfm_amd import box_ops
    g = torch.Generator().manual_seed(3)
    c1, c2 = torch.rand(16, 4, generator=g) * 0.5 + 0.1, torch.rand(16, 4, generator=g) * 0.5 + 0.1
    b1, b2 = box_ops.box_cxcywh_to_xyxy(c1), box_ops.box_cxcywh_to_xyxy(c2)
    assert torch.allclose(box_ops.paired_generalized_box_iou(b1, b2), torch.diag(O.generalized_box_iou(b1, b2)), atol=1e-6)
    assert torch.allclose(box_ops.box_xyxy_to_cxcywh(b1), c1, atol=1e-6)
    from xfm_amd.xfm import XFMBase
    is_image = torch.tensor([0, 1] * 8)
    for co, tg, im in ((c1, c2, None), (c1, c2, is_image), (c1, torch.cat([c2[:15], torch.tensor([[0.5, 0.5, -0.2, 0.1]])]), None)):
        want = O.bbox_loss(co, tg, im)
        got = XFMBase.get_bbox_loss(None, co, tg, im)
        assert abs(float(got[0]) - float(want[0])) < 1e-6 and abs(float(got[1]) - float(want[1])) < 1e-6


def _pretrain(name):
    z, meta = load(name)
    B = meta["B"]
    P = _params(meta["spec"])
    cfg = O.default_cfg(text_layers=meta["text_layers"], fusion_layers=meta["fusion_layers"], vit_depth=meta.get("vit_depth", 12))
    b = syn.pretrain_batch(B, seed=1234)
    masks = syn.mim_block_mask(B, 14, 75, seed=1234)
    out = O.pretrain_forward(P, cfg, b, meta["image_neg_idx"], meta["text_neg_idx"], masks)
    total = 0
    for k in ("loss_itc", "loss_itm", "loss_mlm", "loss_mim"):
        assert abs(float(out[k]) - float(z[k])) < 2e-4 * max(1.0, abs(float(z[k]))), (k, float(out[k]), float(z[k]))
        total = total + out[k]
    total.backward()
    _check_grads(z, "grad", P)
    unused = set(meta["unused"])
    for k, v in P.items():
        if v.dtype.is_floating_point and k in unused:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, f"{k} should receive no gradient"


def test_pretrain_step_small():
    _pretrain("pretrain_small")


def test_pretrain_step_full_depth():
    import os
    from golden_util import GOLDEN
    if not os.path.exists(os.path.join(GOLDEN, "pretrain_full.npz")):
        pytest.skip("pretrain_full.npz not generated")
    _pretrain("pretrain_full")


def test_host_mim_mask_sampler_matches_the_reference_generators_distribution():
    """xfm_amd.beit2.BlockMaskGenerator (the numpy sampler used off the GPU) against statistics of the REAL reference generator
    (tests/golden/mim_mask_stats.npz, 50 000 draws of models/masking_generator.py): per-patch frequency and rows touched."""
    from xfm_amd.beit2 import BlockMaskGenerator
    z, meta = load("mim_mask_stats")
    n, n_ref, grid = 4000, meta["n"], meta["grid"]
    g = BlockMaskGenerator(grid, meta["num"], meta["min_num"], seed=11)
    m = g.batch(n).numpy().astype(np.float64).reshape(n, grid, grid)
    assert (m.sum((1, 2)) == meta["num"]).all()
    f_dev, f_ref = m.mean(0).reshape(-1), z["freq"] / n_ref
    sigma = np.sqrt(f_ref * (1 - f_ref) * (1.0 / n + 1.0 / n_ref))
    assert float((np.abs(f_dev - f_ref) / sigma).max()) < 5.0
    rows = np.bincount((m.sum(2) > 0).sum(1).astype(np.int64), minlength=grid + 1).astype(np.float64)
    ref = z["rows"].astype(np.float64)
    keep = (rows + ref) >= 40
    ka, kb = np.sqrt(ref[keep].sum() / rows[keep].sum()), np.sqrt(rows[keep].sum() / ref[keep].sum())
    chi = float((((ka * rows[keep] - kb * ref[keep]) ** 2) / (rows[keep] + ref[keep])).sum() / max(keep.sum() - 1, 1))
    assert chi < 3.0, chi
