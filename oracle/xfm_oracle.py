"""CPU oracle for the XFM transformer forward/backward hot path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
bench.py's `cpu_baseline` leg may import this module; xfm_amd/ never does.

This is a plain-PyTorch fp32 *restatement* (functional style, weights passed as a
`state_dict`-shaped mapping `P`) of the arithmetic the reference performs on its hot
path.  Each function cites the reference file:line it follows.  Stochastic pieces of
the reference (dropout, drop-path, hard-negative sampling, MIM block masks) are NOT
drawn here: the oracle runs the eval-mode arithmetic and takes the sampled indices /
masks / per-sample drop-path scales as explicit inputs, so that both sides of a parity
test consume identical draws.

Parity pin: tests/golden/*.npz hold outputs of the real reference (imported in the build
container by tools/oracle/gen_golden.py behind tools/oracle/ref_shim.py) on
formula-generated weights and inputs (xfm_amd/synthetic.py); tests/test_oracle_golden.py
checks every function here against them (<= 2e-5 abs + 2e-4 rms, fp32).  Third-party arithmetic not
under /root/reference: torch.nn.functional (torch 2.10) and the exact-erf GELU that
transformers' ACT2FN["gelu"] names (reference pins transformers==4.12.5; call sites
xroberta.py:363,1327) -- pinned only through those fixtures.
"""
import math

import torch
import torch.nn.functional as F

MASK_NEG = -10000.0  # xroberta.py:805-806 (and transformers 4.12.5 invert_attention_mask)


# --------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------
def _lin(P, key, x):
    return F.linear(x, P[key + ".weight"], P.get(key + ".bias"))


def _ln(P, key, x, eps):
    return F.layer_norm(x, (x.shape[-1],), P[key + ".weight"], P[key + ".bias"], eps)


def gelu_erf(x):
    """transformers.activations.gelu == x * 0.5 * (1 + erf(x / sqrt(2)))  (xroberta.py:363,1327)."""
    return F.gelu(x)


# --------------------------------------------------------------------------------------
# BEiT-v2 vision tower  (models/beit2.py)
# --------------------------------------------------------------------------------------
def beit_relative_position_index(gh, gw):
    """beit2.py:92-116: [gh*gw+1, gh*gw+1] int64 index into the [(2gh-1)(2gw-1)+3, H] bias table."""
    nrd = (2 * gh - 1) * (2 * gw - 1) + 3
    ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    dy = ys[:, None] - ys[None, :] + (gh - 1)
    dx = xs[:, None] - xs[None, :] + (gw - 1)
    n = gh * gw
    idx = torch.zeros(n + 1, n + 1, dtype=torch.int64)
    idx[1:, 1:] = dy * (2 * gw - 1) + dx
    idx[0, :] = nrd - 3
    idx[:, 0] = nrd - 2
    idx[0, 0] = nrd - 1
    return idx


def beit_patch_embed(P, pre, image, patch=16):
    """beit2.py:224-230: Conv2d(3,D,k=16,s=16) == per-patch GEMM over (c,ky,kx); -> [B, gh*gw, D]."""
    B, C, H, W = image.shape
    gh, gw = H // patch, W // patch
    w = P[pre + "patch_embed.proj.weight"]  # [D, C, p, p]
    cols = image.reshape(B, C, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * patch * patch)
    return cols @ w.reshape(w.shape[0], -1).t() + P[pre + "patch_embed.proj.bias"]


def beit_attention(P, pre, x, heads, image_atts=None):
    """beit2.py:126-166.  q is scaled BEFORE q@k^T (:136); bias = cat(q_bias, 0, v_bias) (:128-132)."""
    B, N, C = x.shape
    d = C // heads
    qb, vb = P[pre + "q_bias"], P[pre + "v_bias"]
    bias = torch.cat([qb, torch.zeros_like(vb), vb])
    qkv = F.linear(x, P[pre + "qkv.weight"], bias).reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (d ** -0.5), qkv[1], qkv[2]
    s = q @ k.transpose(-2, -1)
    table = P[pre + "relative_position_bias_table"]  # [(2g-1)^2+3, heads]
    index = P[pre + "relative_position_index"]       # [N, N] int64
    s = s + table[index.reshape(-1)].reshape(N, N, heads).permute(2, 0, 1).unsqueeze(0)
    if image_atts is not None:
        s = s + image_atts
    p = s.softmax(dim=-1)
    ctx = (p @ v).transpose(1, 2).reshape(B, N, C)
    return _lin(P, pre + "proj", ctx)


def beit_block(P, pre, x, heads, eps=1e-6, dp1=None, dp2=None):
    """beit2.py:191-206 (gamma branch) + Mlp :62-69.  dp1/dp2: optional per-sample drop-path scales [B]
    (= bernoulli(keep)/keep, beit2.py:38-49 / timm drop_path); None == eval mode."""
    y = P[pre + "gamma_1"] * beit_attention(P, pre + "attn.", _ln(P, pre + "norm1", x, eps), heads)
    if dp1 is not None:
        y = y * dp1.view(-1, 1, 1)
    x = x + y
    h = _lin(P, pre + "mlp.fc2", F.gelu(_lin(P, pre + "mlp.fc1", _ln(P, pre + "norm2", x, eps))))
    y = P[pre + "gamma_2"] * h
    if dp2 is not None:
        y = y * dp2.view(-1, 1, 1)
    return x + y


def beit_pool_tail(P, pre, x, eps=1e-6):
    """beit2.py:456-466: drop cls, fc_norm over patch tokens, mean over patches as pseudo-cls, concat."""
    patches = _ln(P, pre + "fc_norm", x[:, 1:], eps)
    return torch.cat([patches.mean(dim=1, keepdim=True), patches], dim=1)


def beit_forward(P, pre, image, depth=12, heads=12, ids_mask=None, drop_path=None, eps=1e-6):
    """beit2.py:423-466 forward_avgpool (use_abs_pos_emb=False, per-block rel-pos bias).
    ids_mask: optional bool [B, gh*gw] (the MaskingGenerator draws, :430-441);
    drop_path: optional list of (dp1, dp2) per block."""
    x = beit_patch_embed(P, pre, image)
    B = x.shape[0]
    if ids_mask is not None:
        w = ids_mask.unsqueeze(-1).to(x.dtype)
        x = x * (1 - w) + P[pre + "mask_token"].expand(B, x.shape[1], -1) * w
    x = torch.cat([P[pre + "cls_token"].expand(B, -1, -1), x], dim=1)
    for i in range(depth):
        dp1, dp2 = (None, None) if drop_path is None else drop_path[i]
        x = beit_block(P, f"{pre}blocks.{i}.", x, heads, eps, dp1, dp2)
    return beit_pool_tail(P, pre, x, eps)


def beit_region_outputs(full, idx_to_group_img, image_atts):
    """The region call form, beit2.py:467-475 (reached through xfm.py:574-597 get_vision_embeds(image, image_atts, idx_to_group_img)):
    `full` = the tower's ordinary output for the n distinct images [n, 1 + P, D]; every one of the bs samples picks its image
    (idx_to_group_img [bs]) and pools ITS OWN pseudo-cls as the image_atts-weighted mean of the normalised patch rows
    (image_atts [bs, 1 + P]; column 0 belongs to the cls slot and is not used) -> ([bs, 1 + P, D], full)."""
    x = full[:, 1:, :]
    x_bs = torch.gather(x, 0, idx_to_group_img.view(-1, 1, 1).expand(-1, x.shape[1], x.shape[2]))
    w = image_atts[:, 1:].unsqueeze(2).to(x.dtype)
    cls = (w * x_bs).sum(dim=1, keepdim=True) / w.sum(dim=1, keepdim=True)
    return torch.cat([cls, x_bs], dim=1), full


# --------------------------------------------------------------------------------------
# Plain ViT  (models/vit.py; SURVEY row V0, unused by the shipped configs)
# --------------------------------------------------------------------------------------
def vit_attention(P, pre, x, heads):
    """vit.py:60-83: qkv Linear WITH bias, (q @ k^T) * scale AFTER the product, softmax, @ v, proj."""
    B, N, C = x.shape
    d = C // heads
    qkv = _lin(P, pre + "qkv", x).reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    p = ((q @ k.transpose(-2, -1)) * (d ** -0.5)).softmax(dim=-1)
    return _lin(P, pre + "proj", (p @ v).transpose(1, 2).reshape(B, N, C))


def vit_forward(P, pre, image, depth=12, heads=12, eps=1e-6):
    """vit.py:177-219 (eval mode: drop-path off): patch-embed, cls token, + pos_embed, pre-LN blocks :100-103, final norm."""
    x = beit_patch_embed(P, pre, image)
    B = x.shape[0]
    x = torch.cat([P[pre + "cls_token"].expand(B, -1, -1), x], dim=1)
    x = x + P[pre + "pos_embed"][:, :x.shape[1], :]
    for i in range(depth):
        b = f"{pre}blocks.{i}."
        x = x + vit_attention(P, b + "attn.", _ln(P, b + "norm1", x, eps), heads)
        x = x + _lin(P, b + "mlp.fc2", F.gelu(_lin(P, b + "mlp.fc1", _ln(P, b + "norm2", x, eps))))
    return _ln(P, pre + "norm", x, eps)


# --------------------------------------------------------------------------------------
# RoBERTa text / fusion tower  (models/xroberta.py; xbert.py differs where flagged)
# --------------------------------------------------------------------------------------
def roberta_position_ids(input_ids, padding_idx=1):
    """xroberta.py:1747-1757."""
    m = input_ids.ne(padding_idx).int()
    return (torch.cumsum(m, dim=1).type_as(m) * m).long() + padding_idx


def roberta_embeddings(P, pre, input_ids, eps=1e-5, padding_idx=1):
    """xroberta.py:104-137 (token_type_ids all zero; dropout is eval-mode identity here)."""
    pos = roberta_position_ids(input_ids, padding_idx)
    # nn.Embedding(padding_idx=pad): the pad row takes no gradient (xroberta.py:80,100-102)
    e = F.embedding(input_ids, P[pre + "word_embeddings.weight"], padding_idx=padding_idx) \
        + P[pre + "token_type_embeddings.weight"][0] \
        + F.embedding(pos, P[pre + "position_embeddings.weight"], padding_idx=padding_idx)
    return _ln(P, pre + "LayerNorm", e, eps)


def extended_attention_mask(att, causal=False):
    """xroberta.py:751-807.  att [B,S] (1 keep / 0 pad) -> additive [B,1,1,S] (or [B,1,S,S] causal)."""
    att = att.to(torch.float32)
    if causal:
        S = att.shape[1]
        ids = torch.arange(S, device=att.device)
        cm = (ids[None, None, :] <= ids[None, :, None]).to(att.dtype)  # [1,S,S]
        ext = cm[:, None, :, :] * att[:, None, None, :]
    else:
        ext = att[:, None, None, :]
    return (1.0 - ext) * MASK_NEG


def roberta_attention(P, pre, x, add_mask, kv_src=None, heads=12, eps=1e-5, scale_after=False):
    """RobertaSelfAttention.forward xroberta.py:201-289 + RobertaSelfOutput :300-304.
    kv_src: encoder_hidden_states for cross-attention (K,V projected from it, :224-226).
    scale_after=True gives xbert.py:329-330 (scores / sqrt(d) after QK^T, non-fp16 branch)."""
    B, T, C = x.shape
    d = C // heads
    src = x if kv_src is None else kv_src
    S = src.shape[1]
    q = _lin(P, pre + "self.query", x).view(B, T, heads, d).permute(0, 2, 1, 3)
    k = _lin(P, pre + "self.key", src).view(B, S, heads, d).permute(0, 2, 1, 3)
    v = _lin(P, pre + "self.value", src).view(B, S, heads, d).permute(0, 2, 1, 3)
    if scale_after:
        s = (q @ k.transpose(-1, -2)) / math.sqrt(d)
    else:
        s = (q / math.sqrt(d)) @ k.transpose(-1, -2)
    if add_mask is not None:
        s = s + add_mask
    p = s.softmax(dim=-1)
    ctx = (p @ v).permute(0, 2, 1, 3).reshape(B, T, C)
    return _ln(P, pre + "output.LayerNorm", _lin(P, pre + "output.dense", ctx) + x, eps)


def roberta_layer(P, pre, x, add_mask, enc=None, enc_mask=None, has_cross=False, heads=12, eps=1e-5,
                  scale_after=False):
    """RobertaLayer.forward xroberta.py:405-473."""
    y = roberta_attention(P, pre + "attention.", x, add_mask, None, heads, eps, scale_after)
    if has_cross and enc is not None:
        y = roberta_attention(P, pre + "crossattention.", y, enc_mask, enc, heads, eps, scale_after)
    h = gelu_erf(_lin(P, pre + "intermediate.dense", y))
    return _ln(P, pre + "output.LayerNorm", _lin(P, pre + "output.dense", h) + y, eps)


def roberta_encoder(P, pre, x, att, enc=None, enc_att=None, num_layers=12, fusion_layer=12, heads=12,
                    eps=1e-5, causal=False, scale_after=False):
    """RobertaModel.forward xroberta.py:817-957 after the embedding step, mode='multi_modal' (:504-516).
    `pre` is the '...roberta.' prefix; x is the embedding output or `encoder_embeds` (:920-929)."""
    add_mask = extended_attention_mask(att, causal)
    enc_mask = None
    if enc is not None:
        if enc_att is None:
            enc_att = torch.ones(enc.shape[:2], device=enc.device)
        enc_mask = (1.0 - enc_att.to(torch.float32))[:, None, None, :] * MASK_NEG
    for i in range(num_layers):
        x = roberta_layer(P, f"{pre}encoder.layer.{i}.", x, add_mask, enc, enc_mask, i >= fusion_layer, heads, eps,
                          scale_after)
    return x


def roberta_model(P, pre, input_ids=None, att=None, encoder_embeds=None, enc=None, enc_att=None, **kw):
    """RobertaForMaskedLM.bert / RobertaModel.forward: embeddings (unless encoder_embeds) + encoder."""
    x = encoder_embeds if encoder_embeds is not None else roberta_embeddings(P, pre + "embeddings.", input_ids)
    return roberta_encoder(P, pre, x, att, enc, enc_att, **kw)


def roberta_lm_head(P, pre, x, eps=1e-5):
    """RobertaLMHead.forward xroberta.py:1325-1333 (decoder.bias is tied to `bias`)."""
    h = _ln(P, pre + "layer_norm", gelu_erf(_lin(P, pre + "dense", x)), eps)
    return F.linear(h, P[pre + "decoder.weight"], P[pre + "bias"])


def gather_by_pos(seq, pos):
    """xroberta.py:1215-1216."""
    return torch.gather(seq, 1, pos.unsqueeze(2).expand(-1, -1, seq.size(-1)))


def masked_lm_loss(P, model_pre, seq_out, masked_pos, labels, head="lm_head", causal_shift=False, reduction="mean"):
    """RobertaForMaskedLM.forward tail xroberta.py:1273-1299."""
    if masked_pos is not None:
        seq_out = gather_by_pos(seq_out, masked_pos)
    logits = roberta_lm_head(P, f"{model_pre}{head}.", seq_out)
    if causal_shift:
        logits, labels = logits[:, :-1, :], labels[:, 1:]
    return F.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1), reduction=reduction), logits


def causal_lm_loss(P, input_ids, att, enc, enc_att, labels, num_layers, reduction="none", pre="", fusion_layer=0):
    """RobertaForCausalLM.forward xroberta.py:1025-1113 (the VQA answer decoder, model_generation.py:119-128): causal
    self mask, cross-attention from layer `fusion_layer` on, lm_head, logits[:, :-1] vs labels[:, 1:], then view(B, -1).sum(1)."""
    seq = roberta_model(P, pre + "roberta.", input_ids=input_ids, att=att, enc=enc, enc_att=enc_att, num_layers=num_layers,
                        fusion_layer=fusion_layer, causal=True)
    loss, logits = masked_lm_loss(P, pre, seq, None, labels, head="lm_head", causal_shift=True, reduction=reduction)
    return loss.view(input_ids.shape[0], -1).sum(1), logits


def causal_lm_logits(P, input_ids, att, enc, enc_att, num_layers, pre="", fusion_layer=0):
    """The same decoder without labels: full-length logits (rank_answer's first step, model_generation.py:149-155)."""
    if att is None:
        att = torch.ones_like(input_ids)
    seq = roberta_model(P, pre + "roberta.", input_ids=input_ids, att=att, enc=enc, enc_att=enc_att, num_layers=num_layers,
                        fusion_layer=fusion_layer, causal=True)
    return roberta_lm_head(P, pre + "lm_head.", seq)


# --------------------------------------------------------------------------------------
# xbert variant (models/xbert.py): same layer stack with scores scaled AFTER QK^T, absolute positions, BERT LM head
# --------------------------------------------------------------------------------------
def bert_embeddings(P, pre, input_ids, eps=1e-12):
    """BertEmbeddings.forward xbert.py:188-215: word + token_type[0] + position_ids[:, :T], LayerNorm (dropout off)."""
    T = input_ids.shape[1]
    x = F.embedding(input_ids, P[pre + "word_embeddings.weight"], padding_idx=0)
    x = x + P[pre + "token_type_embeddings.weight"][0] + P[pre + "position_embeddings.weight"][:T].unsqueeze(0)
    return _ln(P, pre + "LayerNorm", x, eps)


def bert_model(P, pre, input_ids, att, enc=None, enc_att=None, eps=1e-12, **kw):
    """BertModel.forward xbert.py:1023-1130 (non-fp16 branch => scale_after)."""
    x = bert_embeddings(P, pre + "embeddings.", input_ids, eps)
    return roberta_encoder(P, pre, x, att, enc, enc_att, eps=eps, scale_after=True, **kw)


def bert_lm_head(P, pre, x, eps=1e-12):
    """BertLMPredictionHead xbert.py:663-697: transform (dense -> GELU -> LayerNorm) then decoder + shared bias."""
    h = _ln(P, pre + "transform.LayerNorm", gelu_erf(_lin(P, pre + "transform.dense", x)), eps)
    return F.linear(h, P[pre + "decoder.weight"], P[pre + "bias"])


def bert_causal_lm_loss(P, input_ids, att, enc, enc_att, labels, num_layers, reduction="none", pre="", fusion_layer=0):
    """BertLMHeadModel.forward xbert.py:1262-1347 (the answer decoder of a bert-named VQA model, model_generation.py:52-54): causal
    self mask, cross-attention from layer `fusion_layer` on, cls.predictions head, logits[:, :-1] vs labels[:, 1:], summed per
    sequence for reduction='none'.  Returns (loss rows, unshifted logits)."""
    seq = bert_model(P, pre + "bert.", input_ids, att, enc, enc_att, num_layers=num_layers, fusion_layer=fusion_layer, causal=True)
    logits = bert_lm_head(P, pre + "cls.predictions.", seq)
    loss = F.cross_entropy(logits[:, :-1, :].reshape(-1, logits.shape[-1]), labels[:, 1:].reshape(-1), reduction=reduction)
    return (loss.view(input_ids.shape[0], -1).sum(1) if reduction == "none" else loss), logits


def bert_mlm_loss(P, seq_out, masked_pos, labels):
    """BertForMaskedLM.forward tail xbert.py:1592-1608."""
    if masked_pos is not None:
        seq_out = gather_by_pos(seq_out, masked_pos)
    logits = bert_lm_head(P, "cls.predictions.", seq_out)
    return F.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1)), logits


# --------------------------------------------------------------------------------------
# XFMBase glue and losses  (models/xfm.py, models/model_pretrain.py)
# --------------------------------------------------------------------------------------
def build_mlp_forward(P, pre, x, eps=1e-5):
    """xfm.py:115-121: Linear(D,2D) -> LayerNorm -> GELU -> Linear(2D,out)."""
    return _lin(P, pre + "3", F.gelu(_ln(P, pre + "1", _lin(P, pre + "0", x), eps)))


def get_features(P, image_embeds=None, text_embeds=None):
    """xfm.py:614-621."""
    out = []
    if image_embeds is not None:
        out.append(F.normalize(_lin(P, "vision_proj", image_embeds[:, 0, :]), dim=-1))
    if text_embeds is not None:
        out.append(F.normalize(_lin(P, "text_proj", text_embeds[:, 0, :]), dim=-1))
    return out[0] if len(out) == 1 else tuple(out)


def contrastive_loss(image_feat_all, text_feat_all, temp, idx_all=None):
    """xfm.py:683-715 after the all-gather (feats are the gathered [B*W, E] matrices)."""
    logits = image_feat_all @ text_feat_all.t() / temp
    n = logits.shape[0]
    if idx_all is None:
        labels = torch.arange(n, device=logits.device)
        return (F.cross_entropy(logits, labels) + F.cross_entropy(logits.t(), labels)) / 2
    idx_all = idx_all.view(-1, 1)
    pos = torch.eq(idx_all, idx_all.t()).float()
    labels = pos / pos.sum(1, keepdim=True)
    l_i2t = -torch.sum(F.log_softmax(logits, dim=1) * labels, dim=1).mean()
    l_t2i = -torch.sum(F.log_softmax(logits.t(), dim=1) * labels, dim=1).mean()
    return (l_i2t + l_t2i) / 2


def hard_negative_weights(image_feat, text_feat, temp, idx=None):
    """xfm.py:717-734: the multinomial sampling weights (the draw itself is an input to the oracle)."""
    with torch.no_grad():
        w_i2t = F.softmax(image_feat @ text_feat.t() / temp, dim=1) + 1e-5
        w_t2i = F.softmax(text_feat @ image_feat.t() / temp, dim=1) + 1e-5
        if idx is None:
            w_i2t.fill_diagonal_(0)
            w_t2i.fill_diagonal_(0)
        else:
            m = torch.eq(idx.view(-1, 1), idx.view(1, -1))
            w_i2t.masked_fill_(m, 0)
            w_t2i.masked_fill_(m, 0)
    return w_i2t, w_t2i


def matching_loss(P, cfg, image_embeds, image_atts, text_embeds, text_atts, image_neg_idx, text_neg_idx,
                  is_pretrain=True):
    """xfm.py:749-802 given the sampled negatives (image_neg_idx / text_neg_idx: sequences of B ints)."""
    bs = image_embeds.shape[0]
    ini = torch.as_tensor(image_neg_idx, dtype=torch.long)
    tni = torch.as_tensor(text_neg_idx, dtype=torch.long)
    text_embeds_all = torch.cat([text_embeds, text_embeds[tni]], 0)
    text_atts_all = torch.cat([text_atts, text_atts[tni]], 0)
    image_embeds_all = torch.cat([image_embeds[ini], image_embeds], 0)
    image_atts_all = torch.cat([image_atts[ini], image_atts], 0)

    def cross(img, iatt, temb, tatt):  # xfm.py:659-680, text_embeds branch
        temb = temb.detach() if is_pretrain else temb
        return roberta_model(P, "fusion_encoder.roberta.", att=tatt, encoder_embeds=temb, enc=img, enc_att=iatt,
                             num_layers=cfg["fusion_layers"], fusion_layer=cfg["fusion_start"])[:, 0, :]

    pos = cross(image_embeds, image_atts, text_embeds, text_atts)
    neg = cross(image_embeds_all, image_atts_all, text_embeds_all, text_atts_all)
    out = build_mlp_forward(P, "itm_head.", torch.cat([pos, neg], 0))
    labels = torch.cat([torch.ones(bs, dtype=torch.long), torch.zeros(2 * bs, dtype=torch.long)])
    return F.cross_entropy(out, labels)


def fuse_mlm_loss(P, cfg, text_ids_masked, text_atts, image_embeds, image_atts, masked_pos, masked_ids):
    """xfm.py:638-656 (detach_text_forMLM=True default)."""
    emb = roberta_model(P, "text_encoder.roberta.", input_ids=text_ids_masked, att=text_atts,
                        num_layers=cfg["text_layers"], fusion_layer=cfg["text_layers"]).detach()
    seq = roberta_model(P, "fusion_encoder.roberta.", att=text_atts, encoder_embeds=emb, enc=image_embeds,
                        enc_att=image_atts, num_layers=cfg["fusion_layers"], fusion_layer=cfg["fusion_start"])
    return masked_lm_loss(P, "fusion_encoder.", seq, masked_pos, masked_ids)[0]


def mim_loss(image_embeds_masked, targets, ids_mask):
    """xfm.py:624-635 (mim_cls_only=False): MSE on masked patches + MSE on the pooled cls."""
    t = targets.detach()
    return F.mse_loss(image_embeds_masked[:, 1:, :][ids_mask], t[:, 1:, :][ids_mask]) \
        + F.mse_loss(image_embeds_masked[:, 0, :], t[:, 0, :])


def default_cfg(text_layers=12, fusion_layers=12, vit_depth=12):
    return {"text_layers": text_layers, "fusion_layers": fusion_layers, "fusion_start": 0, "vit_depth": vit_depth,
            "min_temp": 0.001, "max_temp": 0.5}


def pretrain_forward(P, cfg, batch, image_neg_idx=None, text_neg_idx=None, ids_mask=None, sampler=None):
    """model_pretrain.py:30-91 forward_multimodal, data_source='image', all four losses, world_size 1.
    If the negative indices are not given they are drawn with `sampler(weights_row)->int`
    (default torch.multinomial, xfm.py:736-744)."""
    with torch.no_grad():
        P["temp"].clamp_(cfg["min_temp"], cfg["max_temp"])
    image, text_ids, text_atts = batch["image"], batch["text_ids"], batch["text_atts"]
    image_embeds = beit_forward(P, "vision_encoder.", image, depth=cfg["vit_depth"])
    image_atts = torch.ones(image_embeds.shape[:2], dtype=torch.long)
    text_embeds = roberta_model(P, "text_encoder.roberta.", input_ids=text_ids, att=text_atts,
                                num_layers=cfg["text_layers"], fusion_layer=cfg["text_layers"])
    image_feat, text_feat = get_features(P, image_embeds, text_embeds)
    loss_itc = contrastive_loss(image_feat, text_feat, P["temp"])
    if image_neg_idx is None:
        w_i2t, w_t2i = hard_negative_weights(image_feat, text_feat, P["temp"])
        draw = sampler or (lambda w: torch.multinomial(w, 1).item())
        image_neg_idx = [draw(w_t2i[b]) for b in range(image.shape[0])]
        text_neg_idx = [draw(w_i2t[b]) for b in range(image.shape[0])]
    loss_itm = matching_loss(P, cfg, image_embeds, image_atts, text_embeds, text_atts, image_neg_idx, text_neg_idx)
    loss_mlm = fuse_mlm_loss(P, cfg, batch["text_ids_masked"], text_atts, image_embeds, image_atts,
                             batch["masked_pos"], batch["masked_ids"])
    loss_mim = torch.zeros(())
    if ids_mask is not None:
        masked = beit_forward(P, "vision_encoder.", image, depth=cfg["vit_depth"], ids_mask=ids_mask)
        loss_mim = mim_loss(masked, image_embeds, ids_mask)
    return {"loss_itc": loss_itc, "loss_itm": loss_itm, "loss_mlm": loss_mlm, "loss_mim": loss_mim,
            "image_embeds": image_embeds, "text_embeds": text_embeds,
            "image_neg_idx": list(image_neg_idx), "text_neg_idx": list(text_neg_idx)}


def retrieval_forward(P, cfg, batch, idx, image_neg_idx, text_neg_idx, text_prefix="text_encoder."):
    """model_retrieval.py:25-36 XFMForRetrieval.forward given the sampled negatives: ITC with idx soft labels, ITM with the
    text tower attached (is_pretrain=False)."""
    image_embeds = beit_forward(P, "vision_encoder.", batch["image"], depth=cfg["vit_depth"])
    image_atts = torch.ones(image_embeds.shape[:2], dtype=torch.long)
    # without the MLM loss the text tower is a bare RobertaModel: keys `text_encoder.embeddings...` (xfm.py:398-403)
    text_embeds = roberta_model(P, text_prefix, input_ids=batch["text_ids"], att=batch["text_atts"],
                                num_layers=cfg["text_layers"], fusion_layer=cfg["text_layers"])
    image_feat, text_feat = get_features(P, image_embeds, text_embeds)
    loss_itc = contrastive_loss(image_feat, text_feat, P["temp"], idx_all=idx)
    loss_itm = matching_loss(P, cfg, image_embeds, image_atts, text_embeds, batch["text_atts"], image_neg_idx, text_neg_idx,
                             is_pretrain=False)
    return loss_itc, loss_itm


def deep_mlp_forward(P, pre, x, eps=1e-5):
    """XFMForClassification.build_mlp model_classification.py:33-48: (Linear, LayerNorm, GELU) x 4 then Linear; keys 0,1 / 3,4 / 6,7 /
    9,10 / 12."""
    for j in range(4):
        x = gelu_erf(_ln(P, f"{pre}{3 * j + 1}", _lin(P, f"{pre}{3 * j}", x), eps))
    return _lin(P, pre + "12", x)


def classification_forward(P, cfg, image, text_ids, text_atts, deep_head):
    """XFMForClassification.forward model_classification.py:50-93 -> prediction logits.  Branches: image only (cls + mean of the
    patch tokens, :67-70), text only (:52-55), multimodal (fusion [CLS], is_pretrain=False, :72-78)."""
    if image is None:
        feat = roberta_model(P, "text_encoder.", input_ids=text_ids, att=text_atts, num_layers=cfg["text_layers"],
                             fusion_layer=cfg["text_layers"])[:, 0, :]
    elif text_ids is None:
        emb = beit_forward(P, "vision_encoder.", image, depth=cfg["vit_depth"])
        feat = torch.cat([emb[:, 0, :], emb[:, 1:, :].mean(dim=1)], dim=-1)
    else:
        emb = beit_forward(P, "vision_encoder.", image, depth=cfg["vit_depth"])
        atts = torch.ones(emb.shape[:2], dtype=torch.long)
        txt = roberta_model(P, "text_encoder.", input_ids=text_ids, att=text_atts, num_layers=cfg["text_layers"],
                            fusion_layer=cfg["text_layers"])
        feat = roberta_model(P, "fusion_encoder.roberta.", att=text_atts, encoder_embeds=txt, enc=emb, enc_att=atts,
                             num_layers=cfg["fusion_layers"], fusion_layer=cfg["fusion_start"])[:, 0, :]
    return deep_mlp_forward(P, "cls_head.", feat) if deep_head else build_mlp_forward(P, "cls_head.", feat)



# --------------------------------------------------------------------------------------
# XFMForVQA (models/model_generation.py) and XFMForNLVR (models/model_nlvr.py)
# --------------------------------------------------------------------------------------
def fused_question_states(P, cfg, image, q_ids, q_atts, idx_to_group_img=None):
    """model_generation.py:94,103-110: image through the vision tower, question through the bare text tower, then the fusion
    tower's cross-attention to the image (is_pretrain=False: nothing detached).  idx_to_group_img: fewer images than samples, every
    sample reads the whole output of its image (xfm.py:577-588)."""
    emb = beit_forward(P, "vision_encoder.", image, depth=cfg["vit_depth"])
    if idx_to_group_img is not None:
        emb = torch.gather(emb, 0, idx_to_group_img.view(-1, 1, 1).expand(-1, emb.shape[1], emb.shape[2]))
    atts = torch.ones(emb.shape[:2], dtype=torch.long)
    txt = roberta_model(P, "text_encoder.", input_ids=q_ids, att=q_atts, num_layers=cfg["text_layers"], fusion_layer=cfg["text_layers"])
    return roberta_model(P, "fusion_encoder.roberta.", att=q_atts, encoder_embeds=txt, enc=emb, enc_att=atts,
                         num_layers=cfg["fusion_layers"], fusion_layer=cfg["fusion_start"])


def vqa_train_loss(P, cfg, image, q_ids, q_atts, a_ids, a_atts, k, weights, pad_token_id):
    """XFMForVQA.forward(train=True) model_generation.py:96-133: question states repeated once per answer, per-answer sequence loss
    (pad -> -100) weighted and summed, divided by the number of images."""
    qs = fused_question_states(P, cfg, image, q_ids, q_atts)
    rep = torch.tensor([b for b, n in enumerate(k) for _ in range(n)])
    targets = a_ids.masked_fill(a_ids == pad_token_id, -100)
    loss, _ = causal_lm_loss(P, a_ids, a_atts, qs[rep], q_atts[rep], targets, cfg["dec_layers"], pre="text_decoder.",
                             fusion_layer=cfg["dec_fusion_start"])
    return (weights * loss).sum() / image.shape[0]


def vqa_rank_answer(P, cfg, image, q_ids, q_atts, answer_ids, answer_atts, k, pad_token_id):
    """XFMForVQA.forward(train=False) -> rank_answer model_generation.py:135-202.  Returns (topk_ids, topk_probs, first-token
    probabilities of every candidate) -- the last for tolerance-aware comparisons of the discrete ranking."""
    qs = fused_question_states(P, cfg, image, q_ids, q_atts)
    q_ones = torch.ones(qs.shape[:2], dtype=torch.long)
    nq = qs.shape[0]
    start_ids = answer_ids[0, 0].repeat(nq, 1)
    logits = causal_lm_logits(P, start_ids, None, qs, q_ones, cfg["dec_layers"], pre="text_decoder.",
                              fusion_layer=cfg["dec_fusion_start"])[:, 0, :]
    prob_first = F.softmax(logits, dim=1).index_select(1, answer_ids[:, 1])
    topk_probs, topk_ids = prob_first.topk(k, dim=1)
    flat = topk_ids.reshape(-1)
    ids, atts = answer_ids[flat], answer_atts[flat]
    targets = ids.masked_fill(ids == pad_token_id, -100)
    rep = torch.arange(nq).repeat_interleave(k)
    loss, _ = causal_lm_loss(P, ids, atts, qs[rep], q_ones[rep], targets, cfg["dec_layers"], pre="text_decoder.",
                             fusion_layer=cfg["dec_fusion_start"])
    log_probs_sum = (topk_probs.view(-1).log() - loss).view(nq, k)
    probs = F.softmax(log_probs_sum, dim=-1)
    probs, rerank = probs.topk(k, dim=1)
    return torch.gather(topk_ids, 1, rerank), probs, prob_first


def nlvr_forward(P, cfg, image, text_ids, text_atts):
    """XFMForNLVR.forward model_nlvr.py:27-44 -> prediction logits [B, 2]; image = B first images then B second images."""
    emb = beit_forward(P, "vision_encoder.", image, depth=cfg["vit_depth"])
    atts = torch.ones(emb.shape[:2], dtype=torch.long)
    txt = roberta_model(P, "text_encoder.", input_ids=text_ids, att=text_atts, num_layers=cfg["text_layers"],
                        fusion_layer=cfg["text_layers"])
    n = text_ids.shape[0]
    cls = []
    for half in (slice(0, n), slice(n, 2 * n)):
        cls.append(roberta_model(P, "fusion_encoder.roberta.", att=text_atts, encoder_embeds=txt, enc=emb[half], enc_att=atts[half],
                                 num_layers=cfg["fusion_layers"], fusion_layer=cfg["fusion_start"])[:, 0, :])
    return build_mlp_forward(P, "cls_head.", torch.cat(cls, dim=-1))


# --------------------------------------------------------------------------------------
# Retrieval evaluation (Retrieval.py:76-240): k-test re-rank and recall metrics
# --------------------------------------------------------------------------------------
def retrieval_score_matrices(P, cfg, images, text_ids, text_atts, k_test):
    """Retrieval.py:92-166 at world_size 1, one row at a time like the reference: dual-encoder similarities, the k_test best
    candidates of every row re-scored by fusion tower + ITM head (logit of class 1), -100 elsewhere."""
    txt = roberta_model(P, "text_encoder.", input_ids=text_ids, att=text_atts, num_layers=cfg["text_layers"], fusion_layer=cfg["text_layers"])
    img = beit_forward(P, "vision_encoder.", images, depth=cfg["vit_depth"])
    image_feat, text_feat = get_features(P, img, txt)
    sims = image_feat @ text_feat.t()

    def itm(image_rows, text_rows, att_rows):
        atts = torch.ones(image_rows.shape[:2], dtype=torch.long)
        out = roberta_model(P, "fusion_encoder.roberta.", att=att_rows, encoder_embeds=text_rows, enc=image_rows, enc_att=atts,
                            num_layers=cfg["fusion_layers"], fusion_layer=cfg["fusion_start"])
        return build_mlp_forward(P, "itm_head.", out[:, 0, :])[:, 1]

    i2t = torch.full(sims.shape, -100.0)
    for i, row in enumerate(sims):
        idx = row.topk(k_test).indices
        i2t[i, idx] = itm(img[i].repeat(k_test, 1, 1), txt[idx], text_atts[idx])
    t2i = torch.full(sims.t().shape, -100.0)
    for i, row in enumerate(sims.t()):
        idx = row.topk(k_test).indices
        t2i[i, idx] = itm(img[idx], txt[i].repeat(k_test, 1, 1), text_atts[i].repeat(k_test, 1))
    return i2t, t2i, sims


def itm_eval(scores_i2t, scores_t2i, txt2img, img2txt):
    """Retrieval.py:187-240, numpy as in the reference."""
    import numpy as np
    ranks = np.zeros(scores_i2t.shape[0])
    for index, score in enumerate(scores_i2t):
        inds = np.argsort(score)[::-1]
        rank = 1e20
        for i in img2txt[index]:
            tmp = np.where(inds == i)[0][0]
            if tmp < rank:
                rank = tmp
        ranks[index] = rank
    tr1 = 100.0 * len(np.where(ranks < 1)[0]) / len(ranks)
    tr5 = 100.0 * len(np.where(ranks < 5)[0]) / len(ranks)
    tr10 = 100.0 * len(np.where(ranks < 10)[0]) / len(ranks)
    ranks = np.zeros(scores_t2i.shape[0])
    for index, score in enumerate(scores_t2i):
        inds = np.argsort(score)[::-1]
        ranks[index] = np.where(inds == txt2img[index])[0][0]
    ir1 = 100.0 * len(np.where(ranks < 1)[0]) / len(ranks)
    ir5 = 100.0 * len(np.where(ranks < 5)[0]) / len(ranks)
    ir10 = 100.0 * len(np.where(ranks < 10)[0]) / len(ranks)
    tr_mean, ir_mean = (tr1 + tr5 + tr10) / 3, (ir1 + ir5 + ir10) / 3
    return {'txt_r1': tr1, 'txt_r5': tr5, 'txt_r10': tr10, 'txt_r_mean': tr_mean, 'img_r1': ir1, 'img_r5': ir5, 'img_r10': ir10,
            'img_r_mean': ir_mean, 'r_mean': (tr_mean + ir_mean) / 2}


# --------------------------------------------------------------------------------------
# XFMForGrounding (models/model_grounding.py) and the box loss (xfm.py:815-854, models/box_ops.py)
# --------------------------------------------------------------------------------------
def box_cxcywh_to_xyxy(x):
    """box_ops.py:9-13."""
    x_c, y_c, w, h = x.unbind(-1)
    return torch.stack([x_c - 0.5 * w, y_c - 0.5 * h, x_c + 0.5 * w, y_c + 0.5 * h], dim=-1)


def generalized_box_iou(boxes1, boxes2):
    """box_ops.py:24-59, the full N x M matrix as the reference builds it (torchvision's box_area = (x1 - x0)(y1 - y0))."""
    area1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    area2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = area1[:, None] + area2 - inter
    iou = inter / union
    lt = torch.min(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.max(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    area = wh[:, :, 0] * wh[:, :, 1]
    return iou - (area - union) / area


def bbox_loss(output_coord, target_bbox, is_image=None):
    """XFMBase.get_bbox_loss xfm.py:815-840."""
    loss_bbox = F.l1_loss(output_coord, target_bbox, reduction="none")
    boxes1, boxes2 = box_cxcywh_to_xyxy(output_coord), box_cxcywh_to_xyxy(target_bbox)
    if (boxes1[:, 2:] < boxes1[:, :2]).any() or (boxes2[:, 2:] < boxes2[:, :2]).any():
        loss_giou = torch.zeros(output_coord.size(0))
    else:
        loss_giou = 1 - torch.diag(generalized_box_iou(boxes1, boxes2))
    if is_image is None:
        num_boxes = target_bbox.size(0)
    else:
        num_boxes = torch.sum(1 - is_image)
        loss_bbox = loss_bbox * (1 - is_image.view(-1, 1))
        loss_giou = loss_giou * (1 - is_image)
    return loss_bbox.sum() / num_boxes, loss_giou.sum() / num_boxes


def grounding_forward(P, cfg, image, text_ids, text_atts, idx_to_group_img=None):
    """XFMForGrounding.forward model_grounding.py:50-62 -> output_coord [B, 4] (predict_bbox xfm.py:843-854, is_pretrain=False).
    idx_to_group_img (XFMForGroundingDomainPretrain, model_grounding.py:26-33): sample i reads image idx_to_group_img[i] (xfm.py:577-588)."""
    states = fused_question_states(P, cfg, image, text_ids, text_atts, idx_to_group_img=idx_to_group_img)
    return build_mlp_forward(P, "bbox_head.", states[:, 0, :]).sigmoid()
