"""Generate tests/golden/*.npz from the REAL reference (imported from /root/reference behind ref_shim).

Run in the build container only:   python tools/oracle/gen_golden.py [--only NAME] [--full]
Every fixture stores: the state_dict spec (key -> shape, dtype) of the reference module, strided probes + moments of
each output / gradient, and the captured random draws.  Weights and inputs are NOT stored: both sides regenerate them
from xfm_amd.synthetic (formula weights loaded into the reference with load_state_dict(strict=True)).
"""
import argparse
import json
import os
import sys
from functools import partial

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402
from xfm_amd import synthetic as syn  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
NPROBE = 2048


def probe_index(n, k):
    k = min(k, n)
    if k <= 1:
        return torch.zeros(1, dtype=torch.long)
    return (torch.arange(k, dtype=torch.long) * (n - 1)) // (k - 1)


def probe(t, k=NPROBE):
    t = t.detach().double().reshape(-1)
    n = t.numel()
    idx = probe_index(n, k)
    return {"probe": t[idx].float().numpy(), "sum": float(t.sum()), "abs": float(t.abs().sum()), "sq": float((t * t).sum()),
            "n": n}


def pack(prefix, t, out, nprobe=NPROBE):
    for k, v in probe(t, nprobe).items():
        out[f"{prefix}/{k}"] = np.asarray(v)


def spec_of(module):
    return {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in module.state_dict().items()}


def load_formula(module):
    sd = syn.formula_state_dict(module.state_dict())
    module.load_state_dict(sd, strict=True)
    return sd


CFG_PROBE = 1024   # gradient-probe entries per tensor in the config-shape fixtures (256 elsewhere): the tests compare a probe's error with
                    # the reference's own autocast error on the same entries, and a 256-entry estimate of a relative error carries ~6 %
                    # of sampling noise on each side -- over 600+ tensors the largest ratio then reads 1.25-1.3 for arithmetic of equal quality


def grads_of(module, out, prefix="grad", nprobe=256):
    for name, p in module.named_parameters():
        if p.grad is not None:
            pack(f"{prefix}/{name}", p.grad, out, nprobe)


def amp_floor(module, loss_fn, out, prefix="floor"):
    """The reference trains in mixed precision (accelerators/apex_ddp_accelerator.py:41-60, apex O1).  How far are the reference's
    OWN gradients under 16-bit autocast from its fp32 gradients?  Call after the fp32 backward (module.*.grad hold the fp32
    gradients); loss_fn() re-runs the same forward -- same weights, inputs, masks, negatives -- and returns the scalar loss.
    Stores <prefix>/<name> = [rel-L2, cosine] of (bf16-autocast gradient) against (fp32 gradient) per tensor, over the whole tensor.
    The GPU parity tests bound the HIP path's error per tensor by this floor instead of one loose global tolerance."""
    fp32 = {n: p.grad.detach().double().clone() for n, p in module.named_parameters() if p.grad is not None}
    module.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        loss = loss_fn()
    loss.float().backward()
    floor_of(module, fp32, out, prefix)
    module.zero_grad()
    return float(loss)


class FixedMasks:
    """Stands in for MaskingGenerator(): returns the rows of a pre-drawn mask tensor one by one (beit2.py:432-435)."""

    def __init__(self, masks, grid):
        self.masks, self.grid, self.i = masks.numpy().astype(np.int32), grid, 0

    def __call__(self):
        m = self.masks[self.i % len(self.masks)].reshape(self.grid, self.grid)
        self.i += 1
        return m


def save(name, out, meta):
    os.makedirs(OUT, exist_ok=True)
    out["meta"] = np.asarray(json.dumps(meta))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KiB", flush=True)


def gen_beit(depth=2, B=4):
    import torch.nn as nn
    from models.beit2 import VisionTransformer
    torch.manual_seed(0)
    m = VisionTransformer(img_size=224, patch_size=16, embed_dim=768, depth=depth, num_heads=12, mlp_ratio=4,
                          norm_layer=partial(nn.LayerNorm, eps=1e-6), drop_rate=0.0, drop_path_rate=0.1, attn_drop_rate=0.0,
                          use_mean_pooling=True, init_scale=0.001, use_rel_pos_bias=True, use_abs_pos_emb=False,
                          init_values=0.1, qkv_bias=True, local_attn_depth=-1, num_masking_patches=75, min_num_patches=16)
    load_formula(m)
    m.eval()
    image = syn.gaussian("beit.image", (B, 3, 224, 224))
    cot = syn.symmetric("beit.cot", (B, 197, 768), 1.0)
    out = {}
    y = m(image)
    pack("out", y, out)
    (y * cot).sum().backward()
    grads_of(m, out)
    m.zero_grad()
    masks = syn.mim_block_mask(B, 14, 75, seed=7)
    m.generator = FixedMasks(masks, 14)
    ym, ids = m(image, do_mask=True)
    assert torch.equal(ids, masks)
    pack("out_masked", ym, out)
    (ym * cot).sum().backward()
    grads_of(m, out, "grad_masked")
    # the region call form (beit2.py:467-475): 6 samples over the 4 images, ragged region masks
    m.zero_grad()
    idx, atts = syn.region_case(B)
    yr, yfull = m(image, idx_to_group_img=idx, image_atts=atts)
    pack("out_region", yr, out)
    pack("out_region_full", yfull, out)
    cot_r = syn.symmetric("beit.cot_region", tuple(yr.shape), 1.0)
    ((yr * cot_r).sum() + 0.5 * (yfull * cot).sum()).backward()
    grads_of(m, out, "grad_region")
    save(f"beit_{depth}blk", out, {"spec": spec_of(m), "B": B, "depth": depth})


def gen_vit(depth=2, B=3):
    """models/vit.py (unused by the shipped configs): plain pre-LN ViT-B/16, absolute position embedding."""
    from models.vit import VisionTransformer
    torch.manual_seed(0)
    m = VisionTransformer(img_size=224, patch_size=16, embed_dim=768, depth=depth, num_heads=12, mlp_ratio=4, qkv_bias=True,
                          drop_path_rate=0.1)
    load_formula(m)
    m.eval()
    image = syn.gaussian("vit.image", (B, 3, 224, 224))
    cot = syn.symmetric("vit.cot", (B, 197, 768), 1.0)
    out = {}
    y = m(image)
    pack("out", y, out)
    (y * cot).sum().backward()
    grads_of(m, out)
    save(f"vit_{depth}blk", out, {"spec": spec_of(m), "B": B, "depth": depth})


def roberta_cfg(layers, fusion_layer):
    from models.xroberta import RobertaConfig
    cfg = RobertaConfig(**ref_shim.ROBERTA_BASE_CONFIG)
    cfg.num_hidden_layers, cfg.fusion_layer, cfg.encoder_width = layers, fusion_layer, 768
    return cfg


def gen_roberta_text(layers=2, B=4):
    from models.xroberta import RobertaForMaskedLM
    torch.manual_seed(0)
    m = RobertaForMaskedLM(roberta_cfg(layers, layers))
    load_formula(m)
    m.eval()
    b = syn.pretrain_batch(B, seed=11, with_image=False)
    out = {}
    emb = m.roberta.embeddings(input_ids=b["text_ids"])
    pack("embeddings", emb, out)
    h = m.bert(b["text_ids"], attention_mask=b["text_atts"], return_dict=True).last_hidden_state
    pack("hidden", h, out)
    cot = syn.symmetric("roberta.cot", tuple(h.shape), 1.0)
    (h * cot).sum().backward()
    grads_of(m, out, "grad_hidden")
    m.zero_grad()
    res = m(b["text_ids_masked"], attention_mask=b["text_atts"], return_dict=True, labels=b["masked_ids"],
            masked_pos=b["masked_pos"])
    out["mlm_loss"] = np.asarray(float(res.loss))
    pack("mlm_logits", res.logits, out)
    res.loss.backward()
    grads_of(m, out, "grad_mlm")
    save(f"roberta_text_{layers}L", out, {"spec": spec_of(m), "B": B, "layers": layers})


def gen_fusion(layers=2, B=4):
    from models.xroberta import RobertaForMaskedLM
    torch.manual_seed(0)
    m = RobertaForMaskedLM(roberta_cfg(layers, 0))
    load_formula(m)
    m.eval()
    b = syn.pretrain_batch(B, seed=12, with_image=False)
    T = b["text_ids"].shape[1]
    emb = syn.gaussian("fusion.encoder_embeds", (B, T, 768), 0.7).requires_grad_(True)
    img = syn.gaussian("fusion.image_embeds", (B, 197, 768), 0.7).requires_grad_(True)
    img_atts = torch.ones(B, 197, dtype=torch.long)
    img_atts[1, 150:] = 0
    img_atts[3, 100:] = 0
    out = {}
    res = m(encoder_embeds=emb, attention_mask=b["text_atts"], encoder_hidden_states=img, encoder_attention_mask=img_atts,
            return_dict=True, labels=b["masked_ids"], masked_pos=b["masked_pos"])
    out["mlm_loss"] = np.asarray(float(res.loss))
    res.loss.backward()
    grads_of(m, out, "grad_mlm")
    pack("grad_mlm_in/encoder_embeds", emb.grad, out)
    pack("grad_mlm_in/image_embeds", img.grad, out)
    m.zero_grad()
    emb.grad = img.grad = None
    h = m.bert(encoder_embeds=emb, attention_mask=b["text_atts"], encoder_hidden_states=img,
               encoder_attention_mask=img_atts, return_dict=True).last_hidden_state
    pack("hidden", h, out)
    cot = syn.symmetric("fusion.cot", tuple(h.shape), 1.0)
    (h * cot).sum().backward()
    grads_of(m, out, "grad_hidden")
    pack("grad_hidden_in/encoder_embeds", emb.grad, out)
    pack("grad_hidden_in/image_embeds", img.grad, out)
    # causal decoder variant (is_decoder=True: causal self mask, lm_cap_head, shifted CE; xroberta.py:1283-1295)
    m.zero_grad()
    res = m(b["text_ids"], attention_mask=b["text_atts"], encoder_hidden_states=img.detach(), encoder_attention_mask=img_atts,
            return_dict=True, labels=b["text_ids"].masked_fill(b["text_atts"] == 0, -100), is_decoder=True)
    out["causal_loss"] = np.asarray(float(res.loss))
    res.loss.backward()
    grads_of(m, out, "grad_causal")
    save(f"fusion_{layers}L", out, {"spec": spec_of(m), "B": B, "layers": layers})


def gen_causal_lm(layers=2, B=6, L=9, S=30):
    """VQA-style answer decoder (model_generation.py:101-128): RobertaForCausalLM with cross-attention in every layer,
    causal self mask, shifted CE with reduction='none', per-sequence weighted sum."""
    from models.xroberta import RobertaForCausalLM
    torch.manual_seed(0)
    m = RobertaForCausalLM(roberta_cfg(layers, 0))
    load_formula(m)
    m.eval()
    b = syn.pretrain_batch(B, seed=13, with_image=False)
    ids = b["text_ids"][:, :L].clone()
    atts = torch.ones(B, L, dtype=torch.long)
    for r, n in enumerate([9, 4, 7, 3, 9, 5][:B]):  # ragged answers, padded with <pad>=1
        atts[r, n:] = 0
        ids[r, n:] = 1
    enc = syn.gaussian("causal.question_states", (B, S, 768), 0.7).requires_grad_(True)
    enc_atts = torch.ones(B, S, dtype=torch.long)
    enc_atts[1, 20:] = 0
    enc_atts[4, 11:] = 0
    weights = syn.gaussian("causal.weights", (B,), 1.0).abs() + 0.1
    out = {}
    res = m(ids, attention_mask=atts, encoder_hidden_states=enc, encoder_attention_mask=enc_atts,
            labels=ids.masked_fill(ids == 1, -100), return_dict=True, reduction="none")
    out["loss_rows"] = res.loss.detach().numpy().astype(np.float32)
    loss = (weights * res.loss).sum() / B
    out["loss"] = np.asarray(float(loss))
    pack("logits", res.logits, out)
    loss.backward()
    grads_of(m, out, "grad")
    pack("grad_in/question_states", enc.grad, out)
    save(f"causal_lm_{layers}L", out, {"spec": spec_of(m), "B": B, "L": L, "S": S, "layers": layers,
                                       "ids": ids.tolist(), "atts": atts.tolist(), "enc_atts": enc_atts.tolist()})


BERT_BASE_CONFIG = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                        hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                        max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-12,
                        pad_token_id=0)


def gen_bert_causal_lm(layers=2, B=6, L=9, S=30):
    """The answer decoder of a bert-named VQA model (model_generation.py:52-54 -> xbert.BertLMHeadModel, xbert.py:1235-1347): BERT
    embeddings, causal self mask, cross-attention in every layer, BertOnlyMLMHead, shifted CE with reduction='none', per-sequence
    weighted sum -- the same case as causal_lm_2L on the xbert stack ([PAD] = 0)."""
    from models.xbert import BertConfig, BertLMHeadModel
    torch.manual_seed(0)
    cfg = BertConfig(**BERT_BASE_CONFIG)
    cfg.num_hidden_layers, cfg.fusion_layer, cfg.encoder_width = layers, 0, 768
    m = BertLMHeadModel(cfg)
    load_formula(m)
    m.eval()
    b = syn.pretrain_batch(B, seed=13, with_image=False, vocab=30522)
    ids = b["text_ids"][:, :L].clone()
    ids[ids == 0] = 5   # (the RoBERTa-style generator's <s> = 0 is [PAD] here)
    atts = torch.ones(B, L, dtype=torch.long)
    for r, n in enumerate([9, 4, 7, 3, 9, 5][:B]):  # ragged answers, padded with [PAD] = 0
        atts[r, n:] = 0
        ids[r, n:] = 0
    enc = syn.gaussian("causal.question_states", (B, S, 768), 0.7).requires_grad_(True)
    enc_atts = torch.ones(B, S, dtype=torch.long)
    enc_atts[1, 20:] = 0
    enc_atts[4, 11:] = 0
    weights = syn.gaussian("causal.weights", (B,), 1.0).abs() + 0.1
    out = {}
    res = m(ids, attention_mask=atts, encoder_hidden_states=enc, encoder_attention_mask=enc_atts,
            labels=ids.masked_fill(ids == 0, -100), return_dict=True, reduction="none")
    out["loss_rows"] = res.loss.detach().numpy().astype(np.float32)
    loss = (weights * res.loss).sum() / B
    out["loss"] = np.asarray(float(loss))
    pack("logits", res.logits, out)
    loss.backward()
    grads_of(m, out, "grad")
    pack("grad_in/question_states", enc.grad, out)
    save(f"bert_causal_lm_{layers}L", out, {"spec": spec_of(m), "B": B, "L": L, "S": S, "layers": layers,
                                            "ids": ids.tolist(), "atts": atts.tolist(), "enc_atts": enc_atts.tolist()})


def gen_xbert(layers=2, B=4):
    """xbert.BertForMaskedLM (xfm.py:263-284 `use_roberta: False` branch): layer 0 self-only, layer 1 with cross-attention
    (fusion_layer=1); absolute position slice, token type 0, scores scaled AFTER QK^T (xbert.py:329-330), BERT LM head."""
    from models.xbert import BertConfig, BertForMaskedLM
    torch.manual_seed(0)
    cfg = BertConfig(**BERT_BASE_CONFIG)
    cfg.num_hidden_layers, cfg.fusion_layer, cfg.encoder_width = layers, 1, 768
    m = BertForMaskedLM(cfg)
    load_formula(m)
    m.eval()
    b = syn.pretrain_batch(B, seed=14, with_image=False, vocab=30522)
    ids, ids_masked = b["text_ids"].clone(), b["text_ids_masked"].clone()
    ids[b["text_atts"] == 0] = 0  # BERT pads with id 0
    ids_masked[b["text_atts"] == 0] = 0
    img = syn.gaussian("xbert.image_embeds", (B, 197, 768), 0.7).requires_grad_(True)
    img_atts = torch.ones(B, 197, dtype=torch.long)
    img_atts[2, 120:] = 0
    out = {}
    emb = m.bert.embeddings(input_ids=ids)
    pack("embeddings", emb, out)
    h = m.bert(ids, attention_mask=b["text_atts"], return_dict=True, mode="text").last_hidden_state
    pack("hidden_text", h, out)
    res = m(ids_masked, attention_mask=b["text_atts"], encoder_hidden_states=img, encoder_attention_mask=img_atts,
            return_dict=True, labels=b["masked_ids"], masked_pos=b["masked_pos"])
    out["mlm_loss"] = np.asarray(float(res.loss.detach()))
    pack("mlm_logits", res.logits, out)
    res.loss.backward()
    grads_of(m, out, "grad_mlm")
    pack("grad_mlm_in/image_embeds", img.grad, out)
    save(f"xbert_{layers}L", out, {"spec": spec_of(m), "B": B, "layers": layers, "fusion_layer": 1})


def gen_pretrain(name, text_layers, fusion_layers, B=4, with_floor=False):
    from models.model_pretrain import XFM
    ref_shim.init_single_process_group()
    torch.manual_seed(0)
    cfg = ref_shim.pretrain_config(text_layers=text_layers, fusion_layers=fusion_layers)
    m = XFM(cfg, load_vision_params=False, load_text_params=False)
    load_formula(m)
    m.eval()
    b = syn.pretrain_batch(B, seed=1234)
    masks = syn.mim_block_mask(B, 14, 75, seed=1234)
    m.vision_encoder.generator = FixedMasks(masks, 14)
    captured = {}
    orig = m.get_hard_negatives

    def capture(*a, **kw):
        torch.manual_seed(4321)
        r = orig(*a, **kw)
        captured["image_neg_idx"], captured["text_neg_idx"] = list(r[0]), list(r[1])
        return r

    m.get_hard_negatives = capture
    losses = m(b["image"], b["text_ids"], b["text_atts"], text_ids_masked=b["text_ids_masked"], masked_pos=b["masked_pos"],
               masked_ids=b["masked_ids"], ret_mim_loss=True, ret_bbox_loss=False, ret_bbox_giou=False, data_source="image")
    out = {}
    total = 0
    for k in ("loss_itc", "loss_itm", "loss_mlm", "loss_mim"):
        out[k] = np.asarray(float(losses[k]))
        total = total + losses[k]
        print(k, float(losses[k]), flush=True)
    total.backward()
    grads_of(m, out)
    unused = [n for n, p in m.named_parameters() if p.grad is None]
    if with_floor:
        fixed = (torch.stack([torch.as_tensor(i) for i in captured["image_neg_idx"]]),
                 torch.stack([torch.as_tensor(i) for i in captured["text_neg_idx"]]))
        m.get_hard_negatives = lambda *a, **kw: fixed
        last = {}

        def again():
            m.vision_encoder.generator = FixedMasks(masks, 14)
            l2 = m(b["image"], b["text_ids"], b["text_atts"], text_ids_masked=b["text_ids_masked"], masked_pos=b["masked_pos"],
                   masked_ids=b["masked_ids"], ret_mim_loss=True, ret_bbox_loss=False, ret_bbox_giou=False, data_source="image")
            last.update(l2)
            return sum(l2[k].float() for k in ("loss_itc", "loss_itm", "loss_mlm", "loss_mim"))

        amp_floor(m, again, out)
        for k in ("loss_itc", "loss_itm", "loss_mlm", "loss_mim"):
            out["floor_" + k] = np.asarray(float(last[k]))
            print("autocast", k, float(last[k]), flush=True)
    save(name, out, {"spec": spec_of(m), "B": B, "text_layers": text_layers, "fusion_layers": fusion_layers,
                     "image_neg_idx": [int(i) for i in captured["image_neg_idx"]],
                     "text_neg_idx": [int(i) for i in captured["text_neg_idx"]], "unused": unused})


import contextlib

SHALLOW = 2  # ViT blocks in the task-model fixtures (the towers themselves are pinned at full depth by pretrain_full / beit_2blk)


@contextlib.contextmanager
def shallow_vit(depth=SHALLOW):
    """The reference's `beit_base_patch16` hard-codes depth 12 (beit2.py:540-545); the task-model fixtures only exercise the glue
    around the towers, so they are generated with a `depth`-block vision tower to keep the CPU test suite short.  The factory is
    re-bound for the duration of the model construction; no reference source is touched."""
    from functools import partial
    import models.beit2 as rb
    orig = rb.beit_base_patch16

    def factory(img_size, **kwargs):
        model = rb.VisionTransformer(img_size=img_size, patch_size=16, embed_dim=768, depth=depth, num_heads=12, mlp_ratio=4,
                                     norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), **kwargs)
        model.default_cfg = rb._cfg()
        return model

    rb.beit_base_patch16 = factory
    try:
        yield
    finally:
        rb.beit_base_patch16 = orig


def gen_retrieval(B=4, res=224, T=30, name="retrieval_small"):
    """models/model_retrieval.py XFMForRetrieval: ITC with duplicated `idx` (soft labels, xfm.py:705-713), hard negatives that
    avoid same-idx pairs (:731-734), ITM with the text gradient kept (is_pretrain=False).  res / T: the fine-tuning shape of
    configs/xfm-ft/Retrieval_coco.yaml (384 px, 40 tokens) for the `retrieval_384` fixture."""
    from models.model_retrieval import XFMForRetrieval
    ref_shim.init_single_process_group()
    torch.manual_seed(0)
    cfg = ref_shim.pretrain_config(text_layers=2, fusion_layers=2, overrides={"image_res": res, "max_tokens": T, "max_words": T})
    with shallow_vit():
        m = XFMForRetrieval(cfg)
    load_formula(m)
    m.eval()
    b = syn.pretrain_batch(B, seed=77, image_res=res, max_tokens=T)
    idx = torch.tensor([5, 9, 5, 2, 7, 1, 9, 3][:B])
    captured = {}
    orig = m.get_hard_negatives

    def capture(*a, **kw):
        torch.manual_seed(4321)
        r = orig(*a, **kw)
        captured["image_neg_idx"], captured["text_neg_idx"] = list(r[0]), list(r[1])
        return r

    m.get_hard_negatives = capture
    loss_itc, loss_itm = m(b["image"], b["text_ids"], b["text_atts"], idx=idx)
    out = {"loss_itc": np.asarray(float(loss_itc)), "loss_itm": np.asarray(float(loss_itm))}
    (loss_itc + loss_itm).backward()
    grads_of(m, out)
    unused = [n for n, p in m.named_parameters() if p.grad is None]
    save(name, out, {"spec": spec_of(m), "B": B, "text_layers": 2, "fusion_layers": 2, "vit_depth": SHALLOW, "idx": idx.tolist(),
                                  "image_res": res, "max_tokens": T,
                                  "image_neg_idx": [int(i) for i in captured["image_neg_idx"]],
                                  "text_neg_idx": [int(i) for i in captured["text_neg_idx"]], "unused": unused})


class ChunkedVision:
    """Config-shape fixtures (B = 32 at 384 px, B = 24 at 480 px, 12-block tower): the reference's vision tower alone would hold
    2-4 GB of fp32 activations per block for the whole batch.  Same arithmetic, bounded memory: the tower's forward is re-bound (instance
    attribute, no reference source touched) to run `chunk` images at a time without a graph and hand the model a LEAF tensor; after the
    model's backward, finish() re-runs each chunk with a graph and back-propagates that chunk's slice of the leaf's gradient into the
    tower's parameters.  Gradients of a sum over images, summed in chunks."""

    def __init__(self, tower, chunk=8, masks=None, grid=14):
        self.tower, self.chunk, self.orig, self.calls = tower, chunk, tower.forward, []
        self.masks, self.grid = masks, grid       # the pre-drawn MIM masks of a `do_mask=True` call (pretrain_cfg), row i = image i
        tower.forward = self.forward

    def _chunk(self, image, i, do_mask):
        if not do_mask:
            return self.orig(image[i:i + self.chunk])
        self.tower.generator = FixedMasks(self.masks[i:i + self.chunk], self.grid)   # beit2.py:432-435 draws one mask per image, in order
        return self.orig(image[i:i + self.chunk], do_mask=True)

    def forward(self, image, *a, do_mask=False, **kw):
        assert not a and not kw, "config-shape fixtures call the tower on the image alone (or with do_mask=True)"
        with torch.no_grad():
            outs = [self._chunk(image, i, do_mask) for i in range(0, image.shape[0], self.chunk)]
        # under torch.autocast the bf16 copies of the tower's weights were just CACHED by a graph-free pass: a later pass with a graph would
        # reuse them and the weights would get no gradient (round-4 fixtures lacked the autocast floor of every ViT Linear weight for this)
        torch.clear_autocast_cache()
        ids_mask = None
        if do_mask:
            ids_mask = torch.cat([o[1] for o in outs], 0)
            outs = [o[0] for o in outs]
        leaf = torch.cat(outs, 0).float().requires_grad_(True)
        self.calls.append((image, leaf, do_mask))
        return (leaf, ids_mask) if do_mask else leaf

    def finish(self):
        for image, leaf, do_mask in self.calls:
            if leaf.grad is None:
                continue
            for i in range(0, image.shape[0], self.chunk):
                torch.clear_autocast_cache()
                out = self._chunk(image, i, do_mask)
                out = out[0] if do_mask else out
                out.backward(leaf.grad[i:i + self.chunk].to(out.dtype))
        self.calls = []

    def restore(self):
        self.tower.forward = self.orig


def floor_of(module, fp32, out, prefix="floor", nprobe=256):
    """<prefix>/<name> = [rel-L2, cosine] of the reference's bf16-autocast gradient against its fp32 gradient over the WHOLE tensor,
    then the same two figures over the entries the fixture's gradient probe holds (grads_of: 256 strided entries).  The GPU tests see
    only the probe, and a strided probe of a structured weight gradient does not read like the whole tensor (the reference's own
    autocast noise measures 1.2-1.3 x its whole-tensor figure on the probe of the FFN output weights, 0.9 x on others): the tests bound
    the HIP path's probe error by the reference's error ON THE SAME ENTRIES."""
    for n, p in module.named_parameters():
        if p.grad is None or n not in fp32:
            continue
        g, r = p.grad.detach().double().reshape(-1), fp32[n].reshape(-1)

        def pair(a, b):
            bn = float(b.norm())
            return [float((a - b).norm()) / max(bn, 1e-30), float((a @ b) / max(float(a.norm()) * bn, 1e-30))]

        ix = probe_index(r.numel(), nprobe)
        out[f"{prefix}/{n}"] = np.asarray(pair(g, r) + pair(g[ix], r[ix]), dtype=np.float32)


def gen_retrieval_cfg(B=32, res=384, T=40, name="retrieval_cfg"):
    """BASELINE configs[2] at its REAL shape (configs/xfm-ft/Retrieval_coco.yaml: batch 32, 384 px, 40 tokens, 12-block tower, 12 + 12
    layers): losses, every parameter gradient (probes + moments) and the reference's own bf16-autocast floor per tensor, on the inputs
    of tests/test_hip_configs.py::test_retrieval_step_at_config_shape_vs_oracle (same batch seed, idx pattern and given negatives)."""
    from models.model_retrieval import XFMForRetrieval
    ref_shim.init_single_process_group()
    torch.manual_seed(0)
    cfg = ref_shim.pretrain_config(text_layers=12, fusion_layers=12, overrides={"image_res": res, "max_tokens": T, "max_words": T})
    m = XFMForRetrieval(cfg)
    load_formula(m)
    m.eval()
    b = syn.pretrain_batch(B, seed=384, image_res=res, max_tokens=T)
    idx = torch.tensor([i if i % 8 else max(i - 1, 0) for i in range(B)])
    neg_i = [(i + 5) % B if idx[(i + 5) % B] != idx[i] else (i + 7) % B for i in range(B)]
    neg_t = [(i + 11) % B if idx[(i + 11) % B] != idx[i] else (i + 13) % B for i in range(B)]
    m.get_hard_negatives = lambda *a, **kw: (torch.tensor(neg_i), torch.tensor(neg_t))
    cv = ChunkedVision(m.vision_encoder)

    def run():
        loss_itc, loss_itm = m(b["image"], b["text_ids"], b["text_atts"], idx=idx)
        (loss_itc + loss_itm).float().backward()
        cv.finish()
        return float(loss_itc), float(loss_itm)

    li, lm = run()
    print("retrieval_cfg fp32", li, lm, flush=True)
    out = {"loss_itc": np.asarray(li), "loss_itm": np.asarray(lm)}
    grads_of(m, out, nprobe=CFG_PROBE)
    unused = [n for n, p in m.named_parameters() if p.grad is None]
    fp32 = {n: p.grad.detach().double().clone() for n, p in m.named_parameters() if p.grad is not None}
    m.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ai, am = run()
    print("retrieval_cfg bf16 autocast", ai, am, flush=True)
    out["amp_loss_itc"], out["amp_loss_itm"] = np.asarray(ai), np.asarray(am)
    floor_of(m, fp32, out, nprobe=CFG_PROBE)
    cv.restore()
    save(name, out, {"spec": spec_of(m), "B": B, "text_layers": 12, "fusion_layers": 12, "vit_depth": 12, "idx": idx.tolist(),
                     "image_res": res, "max_tokens": T, "image_neg_idx": neg_i, "text_neg_idx": neg_t, "unused": unused})


def gen_pretrain_cfg(B=64, name="pretrain_cfg"):
    """The HEADLINE shape (BASELINE configs[4] per GPU = bench.py's default workload): the reference's full pre-training step
    (model_pretrain.py:30-91: ITC + ITM + MLM + MIM, 12-block tower run twice, 12 + 12 layers) at B = 64, 224 px, 30 tokens, eval mode,
    with the hard negatives the reference drew (captured) and pre-drawn MIM masks: the four losses, every parameter gradient (probes +
    moments) and the reference's own bf16-autocast floor per tensor, on the inputs of
    tests/test_hip_configs.py::test_pretrain_step_at_headline_shape_vs_reference (syn.pretrain_batch(64, seed=64))."""
    from models.model_pretrain import XFM
    ref_shim.init_single_process_group()
    torch.manual_seed(0)
    cfg = ref_shim.pretrain_config(text_layers=12, fusion_layers=12)
    m = XFM(cfg, load_vision_params=False, load_text_params=False)
    load_formula(m)
    m.eval()
    b = syn.pretrain_batch(B, seed=64)
    masks = syn.mim_block_mask(B, 14, 75, seed=64)
    captured = {}
    orig = m.get_hard_negatives

    def capture(*a, **kw):
        if "neg" not in captured:
            torch.manual_seed(4321)
            r = orig(*a, **kw)
            captured["neg"] = (torch.stack([torch.as_tensor(i) for i in r[0]]), torch.stack([torch.as_tensor(i) for i in r[1]]))
        return captured["neg"]

    m.get_hard_negatives = capture
    cv = ChunkedVision(m.vision_encoder, chunk=8, masks=masks, grid=14)
    names = ("loss_itc", "loss_itm", "loss_mlm", "loss_mim")

    def run():
        losses = m(b["image"], b["text_ids"], b["text_atts"], text_ids_masked=b["text_ids_masked"], masked_pos=b["masked_pos"],
                   masked_ids=b["masked_ids"], ret_mim_loss=True, ret_bbox_loss=False, ret_bbox_giou=False, data_source="image")
        sum(losses[k].float() for k in names).backward()
        cv.finish()
        return {k: float(losses[k]) for k in names}

    l32 = run()
    print("pretrain_cfg fp32", l32, flush=True)
    out = {k: np.asarray(v) for k, v in l32.items()}
    grads_of(m, out, nprobe=CFG_PROBE)
    unused = [n for n, p in m.named_parameters() if p.grad is None]
    fp32 = {n: p.grad.detach().double().clone() for n, p in m.named_parameters() if p.grad is not None}
    m.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        l16 = run()
    print("pretrain_cfg bf16 autocast", l16, flush=True)
    for k, v in l16.items():
        out["floor_" + k] = np.asarray(v)
    floor_of(m, fp32, out, nprobe=CFG_PROBE)
    cv.restore()
    save(name, out, {"spec": spec_of(m), "B": B, "text_layers": 12, "fusion_layers": 12, "vit_depth": 12, "seed": 64,
                     "image_neg_idx": [int(i) for i in captured["neg"][0]], "text_neg_idx": [int(i) for i in captured["neg"][1]],
                     "unused": unused})


def gen_imagenet_cfg(B=128, name="imagenet_cfg"):
    """BASELINE configs[1] at its REAL shape (Imagenet.py:437-492: batch 128 per GPU, 224 px, the 12-block tower, cls + mean-patch
    features, the deep 5-Linear head, CE): loss, predictions, every parameter gradient and the reference's bf16-autocast floor, on the
    inputs of tests/test_hip_modules.py::test_classification_imagenet_at_batch_128_vs_oracle."""
    from models.beit2 import beit_base_patch16
    from models.model_classification import XFMForClassification
    ref_shim.init_single_process_group()
    torch.manual_seed(0)
    probe_m = beit_base_patch16(img_size=224, drop_rate=0.0, drop_path_rate=0.1, attn_drop_rate=0.0, use_mean_pooling=True,
                                init_scale=0.001, use_rel_pos_bias=True, use_abs_pos_emb=False, init_values=0.1, qkv_bias=True,
                                local_attn_depth=-1)
    vcfg = _beit_ckpt_config(probe_m)
    cfg = ref_shim.pretrain_config(text_layers=2, fusion_layers=2, overrides={"vision_config": vcfg, "task_name": "imagenet",
                                                                             "num_labels": 1000})
    m = XFMForClassification(cfg)
    load_formula(m)
    m.eval()
    image = syn.gaussian("imagenet128.image", (B, 3, 224, 224))
    targets = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(11))
    cv = ChunkedVision(m.vision_encoder, chunk=8)

    def run():
        loss = m(image, None, None, targets, train=True)
        loss.float().backward()
        cv.finish()
        return float(loss)

    l32 = run()
    print("imagenet_cfg fp32", l32, flush=True)
    out = {"loss_imagenet": np.asarray(l32)}
    grads_of(m, out, nprobe=CFG_PROBE)
    unused = [n for n, p in m.named_parameters() if p.grad is None]
    fp32 = {n: p.grad.detach().double().clone() for n, p in m.named_parameters() if p.grad is not None}
    m.zero_grad()
    with torch.no_grad():
        pack("pred_imagenet", m(image, None, None, targets, train=False), out)
    cv.calls = []
    with torch.autocast("cpu", dtype=torch.bfloat16):
        l16 = run()
    print("imagenet_cfg bf16 autocast", l16, flush=True)
    out["amp_loss_imagenet"] = np.asarray(l16)
    floor_of(m, fp32, out, nprobe=CFG_PROBE)
    cv.restore()
    save(name, out, {"spec": spec_of(m), "B": B, "targets": targets.tolist(), "unused": unused, "text_layers": 2, "fusion_layers": 2})


def gen_vqa_cfg(B=24, res=480, name="vqa_cfg"):
    """BASELINE configs[3] at its REAL shape (configs/xfm-ft/VQA.yaml: batch 24, 480 px, 12 + 12 layers + 12-layer answer decoder):
    the weighted answer loss, every parameter gradient and the reference's bf16-autocast floor, on the inputs of
    tests/test_hip_configs.py::test_vqa_step_at_config_shape_vs_oracle (syn.vqa_batch(24, seed=480))."""
    from types import SimpleNamespace as NS

    def build_tokenizer(*a, **kw):
        raise RuntimeError("dataset.build_tokenizer is stubbed: the reference's dataset package needs torchvision / PIL")

    ref_shim._stub("dataset", build_tokenizer=build_tokenizer)
    from models.model_generation import XFMForVQA
    ref_shim.init_single_process_group()
    torch.manual_seed(0)
    cfg = ref_shim.pretrain_config(text_layers=12, fusion_layers=12,
                                   overrides={"pad_token_id": 1, "decoder_fusion_start_at": 0, "num_dec_layers": 12, "image_res": res})
    m = XFMForVQA(cfg)
    load_formula(m)
    m.eval()
    x = syn.vqa_batch(B, seed=480, image_res=res)
    q, a = NS(input_ids=x.q_ids, attention_mask=x.q_atts), NS(input_ids=x.a_ids, attention_mask=x.a_atts)
    cv = ChunkedVision(m.vision_encoder, chunk=6)

    def run():
        loss = m(x.image, q, a, k=x.k, weights=x.weights, train=True)
        loss.float().backward()
        cv.finish()
        return float(loss)

    l32 = run()
    print("vqa_cfg fp32", l32, flush=True)
    out = {"loss_vqa": np.asarray(l32)}
    grads_of(m, out, nprobe=CFG_PROBE)
    unused = [n for n, p in m.named_parameters() if p.grad is None]
    fp32 = {n: p.grad.detach().double().clone() for n, p in m.named_parameters() if p.grad is not None}
    m.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        l16 = run()
    print("vqa_cfg bf16 autocast", l16, flush=True)
    out["amp_loss_vqa"] = np.asarray(l16)
    floor_of(m, fp32, out, nprobe=CFG_PROBE)
    cv.restore()
    save(name, out, {"spec": spec_of(m), "B": B, "image_res": res, "text_layers": 12, "fusion_layers": 12, "vit_depth": 12, "dec_layers": 12,
                     "dec_fusion_start": 0, "pad_token_id": 1, "answers": int(sum(x.k)), "unused": unused})


def gen_checkpoint():
    """Checkpoint key surgery (xfm.py:408-468) at equal resolution: a pre-training checkpoint ({'model': state_dict}, text tower with
    LM heads) loaded into the fine-tuning XFMForRetrieval (bare text encoder) by the reference's own load_pretrained."""
    import tempfile
    from models import load_pretrained
    from models.model_pretrain import XFM
    from models.model_retrieval import XFMForRetrieval
    ref_shim.init_single_process_group()
    torch.manual_seed(0)
    cfg = ref_shim.pretrain_config(text_layers=2, fusion_layers=2)
    pre = XFM(cfg, load_vision_params=False, load_text_params=False)
    sd = syn.formula_state_dict(pre.state_dict())
    path = os.path.join(tempfile.mkdtemp(), "ckpt.th")
    torch.save({"model": sd, "epoch": 3}, path)
    m = XFMForRetrieval(cfg)
    state_dict = load_pretrained(m, path, cfg, is_eval=False, load_text=True)
    msg = m.load_state_dict(state_dict, strict=False)
    # every surviving tensor is the checkpoint tensor of its (renamed) source key: record a checksum per key
    sums = {k: float(v.double().sum()) for k, v in state_dict.items()}
    save("checkpoint_surgery", {}, {"pretrain_spec": spec_of(pre), "keys": sorted(state_dict.keys()), "missing": sorted(msg.missing_keys),
                                    "unexpected": sorted(msg.unexpected_keys), "sums": sums})


def gen_checkpoint_vqa():
    """XFMForVQA.load_pretrained (model_generation.py:61-91): the same pre-training checkpoint into the VQA model -- text tower keys
    lose their `roberta.` level, the answer decoder starts as a copy of the fusion tower."""
    import tempfile
    from models.model_pretrain import XFM

    def build_tokenizer(*a, **kw):
        raise RuntimeError("stub")

    ref_shim._stub("dataset", build_tokenizer=build_tokenizer)
    from models.model_generation import XFMForVQA
    ref_shim.init_single_process_group()
    torch.manual_seed(0)
    cfg = ref_shim.pretrain_config(text_layers=2, fusion_layers=2,
                                   overrides={"pad_token_id": 1, "decoder_fusion_start_at": 0, "num_dec_layers": 2})
    pre = XFM(cfg, load_vision_params=False, load_text_params=False)
    sd = syn.formula_state_dict(pre.state_dict())
    path = os.path.join(tempfile.mkdtemp(), "ckpt.th")
    torch.save({"model": sd, "epoch": 3}, path)
    m = XFMForVQA(cfg)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_pretrained(path, cfg, is_eval=False)
    after = m.state_dict()
    loaded = sorted(k for k in after if not torch.equal(after[k], before[k]))
    # where did every loaded tensor come from?  (checksum match against the checkpoint)
    by_sum = {}
    for k, v in sd.items():
        by_sum.setdefault(round(float(v.double().sum()), 6), []).append(k)
    origin = {k: sorted(by_sum.get(round(float(after[k].double().sum()), 6), [])) for k in loaded}
    save("checkpoint_vqa", {}, {"pretrain_spec": spec_of(pre), "vqa_spec": spec_of(m), "loaded": loaded, "origin": origin})


def _beit_ckpt_config(depth_spec_model):
    """A synthetic BEiT-v2 checkpoint + vision_config json for models that are built with load_vision_params=True."""
    import tempfile
    d = tempfile.mkdtemp()
    sd = syn.formula_state_dict(depth_spec_model.state_dict())
    sd["head.weight"], sd["head.bias"] = torch.zeros(1000, 768), torch.zeros(1000)  # dropped by load_pretrained_beit2
    torch.save({"model": sd}, os.path.join(d, "beit.pth"))
    with open(os.path.join(d, "config_beit2_base.json"), "w") as f:
        json.dump({"ckpt": os.path.join(d, "beit.pth"), "vision_width": 768, "patch_size": 16}, f)
    return os.path.join(d, "config_beit2_base.json")


def gen_classification(B=4):
    """models/model_classification.py XFMForClassification, ImageNet branch (BASELINE configs[1]: ViT-only path): vision tower loaded
    through load_pretrained_beit2 from a synthetic checkpoint, cls + mean-patch features, deep MLP head, CE; and the multimodal
    branch (fusion tower, plain head) on the same weights."""
    from models.beit2 import beit_base_patch16
    from models.model_classification import XFMForClassification
    ref_shim.init_single_process_group()
    torch.manual_seed(0)
    probe = beit_base_patch16(img_size=224, drop_rate=0.0, drop_path_rate=0.1, attn_drop_rate=0.0, use_mean_pooling=True,
                              init_scale=0.001, use_rel_pos_bias=True, use_abs_pos_emb=False, init_values=0.1, qkv_bias=True,
                              local_attn_depth=-1)
    vcfg = _beit_ckpt_config(probe)
    cfg = ref_shim.pretrain_config(text_layers=2, fusion_layers=2, overrides={"vision_config": vcfg, "task_name": "imagenet",
                                                                             "num_labels": 1000})
    m = XFMForClassification(cfg)
    load_formula(m)
    m.eval()
    b = syn.pretrain_batch(B, seed=55)
    targets = torch.tensor([3, 999, 0, 512][:B])
    out = {}
    loss = m(b["image"], None, None, targets, train=True)
    out["loss_imagenet"] = np.asarray(float(loss.detach()))
    pack("pred_imagenet", m(b["image"], None, None, targets, train=False), out)
    loss.backward()
    grads_of(m, out, "grad_imagenet")
    unused = [n for n, p in m.named_parameters() if p.grad is None]
    amp_floor(m, lambda: m(b["image"], None, None, targets, train=True), out, "floor_imagenet")
    save("classification_imagenet", out, {"spec": spec_of(m), "B": B, "targets": targets.tolist(), "unused": unused,
                                          "text_layers": 2, "fusion_layers": 2})
    # multimodal branch (e.g. NLVR / VE style heads): plain 2-layer head on the fused [CLS]
    cfg2 = ref_shim.pretrain_config(text_layers=2, fusion_layers=2, overrides={"vision_config": vcfg, "task_name": "ve", "num_labels": 3})
    m2 = XFMForClassification(cfg2)
    load_formula(m2)
    m2.eval()
    t2 = torch.tensor([2, 0, 1, 1][:B])
    out2 = {}
    loss2 = m2(b["image"], b["text_ids"], b["text_atts"], t2, train=True)
    out2["loss_mm"] = np.asarray(float(loss2.detach()))
    loss2.backward()
    grads_of(m2, out2, "grad_mm")
    amp_floor(m2, lambda: m2(b["image"], b["text_ids"], b["text_atts"], t2, train=True), out2, "floor_mm")
    # text-only branch on the same model
    m2.zero_grad()
    loss3 = m2(None, b["text_ids"], b["text_atts"], t2, train=True)
    out2["loss_text"] = np.asarray(float(loss3.detach()))
    loss3.backward()
    grads_of(m2, out2, "grad_text")
    save("classification_mm", out2, {"spec": spec_of(m2), "B": B, "targets": t2.tolist(), "text_layers": 2, "fusion_layers": 2})


def gen_vqa(res=224, name="vqa_small"):
    """models/model_generation.py XFMForVQA (BASELINE configs[3]): weighted answer-decoder loss + its gradients, and the
    inference-time answer ranking (first-token shortlist, sequence log-likelihood re-rank).  res = 480 (configs/xfm-ft/VQA_480.yaml)
    for the `vqa_480` fixture."""
    from types import SimpleNamespace as NS

    def build_tokenizer(*a, **kw):  # only the captioning classes of model_generation.py call it
        raise RuntimeError("dataset.build_tokenizer is stubbed: the reference's dataset package needs torchvision / PIL")

    ref_shim._stub("dataset", build_tokenizer=build_tokenizer)  # import stub, as in ref_shim.install()
    from models.model_generation import XFMForVQA
    ref_shim.init_single_process_group()
    torch.manual_seed(0)
    cfg = ref_shim.pretrain_config(text_layers=2, fusion_layers=2,
                                   overrides={"pad_token_id": 1, "decoder_fusion_start_at": 0, "num_dec_layers": 2, "image_res": res})
    with shallow_vit():
        m = XFMForVQA(cfg)
    load_formula(m)
    m.eval()
    x = syn.vqa_inputs(image_res=res)
    q, a, c = NS(input_ids=x.q_ids, attention_mask=x.q_atts), NS(input_ids=x.a_ids, attention_mask=x.a_atts), \
        NS(input_ids=x.c_ids, attention_mask=x.c_atts)
    out = {}
    loss = m(x.image, q, a, k=x.k, weights=x.weights, train=True)
    out["loss_vqa"] = np.asarray(float(loss.detach()))
    print("loss_vqa", float(loss), flush=True)
    loss.backward()
    grads_of(m, out)
    unused = [n for n, p in m.named_parameters() if p.grad is None]
    amp_floor(m, lambda: m(x.image, q, a, k=x.k, weights=x.weights, train=True), out)
    with torch.no_grad():
        topk_ids, topk_probs = m(x.image, q, c, k=x.topk, train=False)
    out["topk_ids"] = topk_ids.numpy()
    out["topk_probs"] = topk_probs.numpy()
    print("topk", topk_ids.tolist(), topk_probs.tolist(), flush=True)
    save(name, out, {"spec": spec_of(m), "B": 3, "image_res": res, "text_layers": 2, "fusion_layers": 2, "vit_depth": SHALLOW, "dec_layers": 2, "dec_fusion_start": 0,
                            "pad_token_id": 1, "unused": unused})


def gen_nlvr(B=2):
    """models/model_nlvr.py XFMForNLVR: 2B images (first images, then second images) against B statements."""
    from models.model_nlvr import XFMForNLVR
    ref_shim.init_single_process_group()
    torch.manual_seed(0)
    cfg = ref_shim.pretrain_config(text_layers=2, fusion_layers=2)
    with shallow_vit():
        m = XFMForNLVR(cfg)
    load_formula(m)
    m.eval()
    b = syn.pretrain_batch(2 * B, seed=95)
    ids, atts = b["text_ids"][:B], b["text_atts"][:B]
    targets = torch.tensor([1, 0][:B])
    out = {}
    loss = m(b["image"], ids, atts, targets, train=True)
    out["loss_nlvr"] = np.asarray(float(loss.detach()))
    with torch.no_grad():
        pack("pred_nlvr", m(b["image"], ids, atts, targets, train=False), out)
    loss.backward()
    grads_of(m, out)
    unused = [n for n, p in m.named_parameters() if p.grad is None]
    save("nlvr_small", out, {"spec": spec_of(m), "B": B, "targets": targets.tolist(), "text_layers": 2, "fusion_layers": 2, "vit_depth": SHALLOW,
                             "unused": unused})


def gen_retrieval_eval():
    """The k-test re-rank of Retrieval.py:76-184 at world_size 1.  Retrieval.py itself cannot be imported here (its dataset / utils
    imports need torchvision, PIL and ruamel), so the LOOP below restates :128-166 call for call, while every tensor comes from the
    reference model's own methods (get_text_embeds / get_vision_embeds / get_features / get_cross_embeds / itm_head)."""
    from models.model_retrieval import XFMForRetrieval
    ref_shim.init_single_process_group()
    torch.manual_seed(0)
    cfg = ref_shim.pretrain_config(text_layers=2, fusion_layers=2)
    with shallow_vit():
        m = XFMForRetrieval(cfg)
    load_formula(m)
    m.eval()
    x = syn.retrieval_eval_inputs()
    k = x.k_test
    with torch.no_grad():
        text_feats = m.get_text_embeds(x.text_ids, x.text_atts)
        text_embeds = m.get_features(text_embeds=text_feats)
        image_feats, _ = m.get_vision_embeds(x.image)
        image_embeds = m.get_features(image_embeds=image_feats)
        sims_matrix = image_embeds @ text_embeds.t()
        i2t = torch.full(sims_matrix.shape, -100.0)
        for i, sims in enumerate(sims_matrix):
            topk_sim, topk_idx = sims.topk(k=k, dim=0)
            encoder_output = image_feats[i].repeat(k, 1, 1)
            encoder_att = torch.ones(encoder_output.size()[:-1], dtype=torch.long)
            output = m.get_cross_embeds(image_embeds=encoder_output, image_atts=encoder_att, text_ids=x.text_ids[topk_idx],
                                        text_atts=x.text_atts[topk_idx], text_embeds=text_feats[topk_idx])
            i2t[i, topk_idx] = m.itm_head(output[:, 0, :])[:, 1]
        sims_matrix = sims_matrix.t()
        t2i = torch.full(sims_matrix.shape, -100.0)
        for i, sims in enumerate(sims_matrix):
            topk_sim, topk_idx = sims.topk(k=k, dim=0)
            encoder_output = image_feats[topk_idx]
            encoder_att = torch.ones(encoder_output.size()[:-1], dtype=torch.long)
            output = m.get_cross_embeds(image_embeds=encoder_output, image_atts=encoder_att, text_ids=x.text_ids[i].repeat(k, 1),
                                        text_atts=x.text_atts[i].repeat(k, 1), text_embeds=text_feats[i].repeat(k, 1, 1))
            t2i[i, topk_idx] = m.itm_head(output[:, 0, :])[:, 1]
    out = {"sims": sims_matrix.t().numpy(), "score_i2t": i2t.numpy(), "score_t2i": t2i.numpy()}
    print(out["score_i2t"], flush=True)
    save("retrieval_eval", out, {"spec": spec_of(m), "text_layers": 2, "fusion_layers": 2, "vit_depth": SHALLOW})


def grounding_targets(B):
    # chosen so that, against the boxes the formula weights predict, every min / max / clamp of the GIoU and every sign of the L1 term
    # is decided by a margin >= 0.05: the bf16 path moves a coordinate by up to 0.01 and must not land on the other side of a kink
    return torch.tensor([[0.5, 0.3, 0.4, 0.4], [0.45, 0.3, 0.2, 0.3], [0.55, 0.4, 0.5, 0.5], [0.45, 0.55, 0.9, 0.8]][:B])


def gen_grounding(B=3):
    """models/model_grounding.py XFMForGrounding: predicted box, L1 + GIoU losses and their gradients."""
    from models.model_grounding import XFMForGrounding
    ref_shim.init_single_process_group()
    torch.manual_seed(0)
    cfg = ref_shim.pretrain_config(text_layers=2, fusion_layers=2)
    with shallow_vit():
        m = XFMForGrounding(cfg)
    load_formula(m)
    m.eval()
    b = syn.pretrain_batch(B, seed=99)
    target = grounding_targets(B)
    out = {}
    coord, loss_bbox, loss_giou = m(b["image"], b["text_ids"], b["text_atts"], target_bbox=target)
    out["coord"] = coord.detach().numpy()
    out["loss_bbox"], out["loss_giou"] = np.asarray(float(loss_bbox.detach())), np.asarray(float(loss_giou.detach()))
    print("coord", coord.tolist(), float(loss_bbox), float(loss_giou), flush=True)
    (loss_bbox + loss_giou).backward()
    grads_of(m, out)
    unused = [n for n, p in m.named_parameters() if p.grad is None]
    save("grounding_small", out, {"spec": spec_of(m), "B": B, "target": target.tolist(), "text_layers": 2, "fusion_layers": 2, "vit_depth": SHALLOW,
                                  "unused": unused})


def gen_grounding_domain(n_images=3):
    """models/model_grounding.py XFMForGroundingDomainPretrain: bs = n_images + 2 (expression, box) samples over n_images images
    (`idx_to_group_img`, every sample on its whole image), `is_image` weighting of the box losses."""
    from models.model_grounding import XFMForGroundingDomainPretrain
    ref_shim.init_single_process_group()
    torch.manual_seed(0)
    cfg = ref_shim.pretrain_config(text_layers=2, fusion_layers=2)
    with shallow_vit():
        m = XFMForGroundingDomainPretrain(cfg)
    load_formula(m)
    m.eval()
    bs = n_images + 2
    b = syn.pretrain_batch(bs, seed=98)
    idx, _ = syn.region_case(n_images)
    target = torch.tensor([[0.5, 0.3, 0.4, 0.4], [0.45, 0.3, 0.2, 0.3], [0.55, 0.4, 0.5, 0.5], [0.45, 0.55, 0.9, 0.8], [0.4, 0.5, 0.3, 0.6]][:bs])
    is_image = torch.tensor([0, 1, 0, 0, 1][:bs])
    out = {}
    loss_bbox, loss_giou = m(b["image"][:n_images], b["text_ids"], b["text_atts"], idx, target, is_image=is_image)
    out["loss_bbox"], out["loss_giou"] = np.asarray(float(loss_bbox.detach())), np.asarray(float(loss_giou.detach()))
    print("grounding_domain", float(loss_bbox), float(loss_giou), flush=True)
    (loss_bbox + loss_giou).backward()
    grads_of(m, out)
    unused = [n for n, p in m.named_parameters() if p.grad is None]
    save("grounding_domain", out, {"spec": spec_of(m), "n_images": n_images, "bs": bs, "idx": idx.tolist(), "target": target.tolist(),
                                   "is_image": is_image.tolist(), "text_layers": 2, "fusion_layers": 2, "vit_depth": SHALLOW, "unused": unused})


def gen_harness():
    """optim.py create_optimizer's four parameter groups on the reference pre-training model, and scheduler.py's linear schedule.
    (transformers 5.x dropped `transformers.optimization.AdamW`; it is aliased to torch.optim.AdamW -- an API alias, the grouping
    code under test is the reference's.)"""
    import transformers.optimization as topt
    if not hasattr(topt, "AdamW"):
        topt.AdamW = torch.optim.AdamW
    import optim as ref_optim
    import scheduler as ref_sched
    from models.model_pretrain import XFM
    ref_shim.init_single_process_group()

    class AD(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    m = XFM(ref_shim.pretrain_config(text_layers=2, fusion_layers=2), load_vision_params=False, load_text_params=False)
    opt = ref_optim.create_optimizer(AD(lr=1e-4, weight_decay=0.01, lr_mult=2), m)
    names = {id(p): n for n, p in m.named_parameters()}
    groups = [[names[id(p)] for p in g["params"]] for g in opt.param_groups]
    hyper = [[g["lr"], g["weight_decay"], list(g["betas"]), g["eps"]] for g in opt.param_groups]
    sch = ref_sched.create_scheduler(AD(sched="linear", num_warmup_steps=0.1, epochs=2, step_per_epoch=10), opt)
    lrs = []
    for _ in range(23):
        lrs.append([opt.param_groups[0]["lr"], opt.param_groups[2]["lr"]])
        opt.step()
        sch.step()
    save("harness", {"lrs": np.asarray(lrs)}, {"groups": groups, "hyper": hyper, "init_params": list(m.init_params),
                                               "text_layers": 2, "fusion_layers": 2})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--full", action="store_true", help="also emit the 12/12/12 end-to-end step (slow)")
    a = ap.parse_args()
    torch.set_num_threads(int(os.environ.get("GEN_THREADS", "8")))
    ref_shim.install()
    jobs = {"beit": lambda: gen_beit(2), "roberta_text": lambda: gen_roberta_text(2), "fusion": lambda: gen_fusion(2),
            "pretrain_small": lambda: gen_pretrain("pretrain_small", 2, 2), "causal_lm": lambda: gen_causal_lm(2), "bert_causal_lm": lambda: gen_bert_causal_lm(2), "xbert": lambda: gen_xbert(2), "vit": lambda: gen_vit(2), "retrieval": lambda: gen_retrieval(), "checkpoint": lambda: gen_checkpoint(), "classification": lambda: gen_classification(), "vqa": gen_vqa, "nlvr": gen_nlvr, "retrieval_eval": gen_retrieval_eval, "harness": gen_harness, "checkpoint_vqa": gen_checkpoint_vqa, "grounding": gen_grounding, "grounding_domain": gen_grounding_domain,
            "retrieval_384": lambda: gen_retrieval(B=8, res=384, T=40, name="retrieval_384"),
            "vqa_480": lambda: gen_vqa(res=480, name="vqa_480")}
    cfg_jobs = {"retrieval_cfg": gen_retrieval_cfg, "vqa_cfg": gen_vqa_cfg, "pretrain_cfg": gen_pretrain_cfg, "imagenet_cfg": gen_imagenet_cfg}   # config-shape fixtures: minutes of CPU each, only on request
    if a.only in cfg_jobs:
        jobs[a.only] = cfg_jobs[a.only]
    if a.full:
        jobs["pretrain_full"] = lambda: gen_pretrain("pretrain_full", 12, 12, with_floor=True)
    for k, fn in jobs.items():
        if a.only is None or a.only == k:
            print("==", k, flush=True)
            fn()


if __name__ == "__main__":
    main()
