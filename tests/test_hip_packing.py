"""Unpadded (packed) token rows against the reference's padded computation, on a real MI355X.

The padded path is itself pinned to the real reference by tests/golden (test_hip_modules.py); here the packed path must give the
same values on every real token, forward and backward: the attention kernels with (start, length) arrays vs. padded rows + key
mask, the embedding kernel through a row map, the text tower, the fusion tower with the device-computed slack block, and the
whole pre-training step (also against the golden fixture of the reference itself)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load, state_from_spec  # noqa: E402
from xfm_amd import functional as Fx  # noqa: E402
from xfm_amd import synthetic as syn  # noqa: E402
from xfm_amd.packing import Pack, pack_rows, rows_gather, unpack  # noqa: E402

BF16 = torch.bfloat16


def _lens(B, T, seed, lo=3):
    g = torch.Generator().manual_seed(seed)
    ln = torch.randint(lo, T + 1, (B,), generator=g)
    ln[0], ln[-1] = T, lo   # the extremes
    return ln


def _rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_row_gather_scatter_and_pack_layout():
    B, T, D = 7, 12, 64
    ln = _lens(B, T, 1)
    p = Pack.from_lens(ln.tolist(), T, "cuda")
    assert p.cap == int(ln.sum()) and p.start.tolist() == np.concatenate([[0], np.cumsum(ln.numpy())[:-1]]).tolist()
    x = torch.randn(B, T, D, device="cuda").to(BF16)
    rows = pack_rows(x, p)
    back = unpack(rows, p)
    keep = (torch.arange(T)[None, :] < ln[:, None]).cuda()
    assert torch.equal(back[keep], x[keep]) and float(back[~keep].float().abs().max()) == 0.0
    # adjoint: <gather(x), y> == <x, scatter(y)>, duplicates included
    idx = torch.tensor([0, 3, 3, -1, p.cap - 1, 0], dtype=torch.int32, device="cuda")
    src = torch.randn(p.cap, D, device="cuda").to(BF16).requires_grad_(True)
    out = rows_gather(src, idx)
    w = torch.randn_like(out.float())
    (out.float() * w).sum().backward()
    want = torch.zeros(p.cap, D, device="cuda")
    for r, s in enumerate(idx.tolist()):
        if s >= 0:
            want[s] += w[r].to(BF16).float()
    assert torch.allclose(src.grad.float(), want.to(BF16).float(), atol=1e-2) and float(out[3].float().abs().max()) == 0.0
    # device-side layout with a slack block: offsets follow the device lengths, nothing overlaps
    sel = torch.tensor([2, 2, 0, 5, 1, 6, 3], device="cuda")
    ld = p.lens
    f = Pack.concat([(ld, p.cap, ln.tolist()), (ld.index_select(0, sel), B * T, None)], T)
    st, le = f.start.cpu().numpy(), f.lens.cpu().numpy()
    assert f.cap == p.cap + B * T and st[B] == p.cap and (st[B:] + le[B:] <= f.cap).all()
    assert (st[1:B] == st[:B - 1] + le[:B - 1]).all() and (st[B + 1:] == st[B:-1] + le[B:-1]).all()
    gi = f.gather_index(p, torch.cat([torch.arange(B, device="cuda"), sel])).cpu().numpy()
    for j in range(2 * B):
        srcseq = j if j < B else int(sel[j - B])
        assert (gi[st[j]:st[j] + le[j]] == p.start[srcseq].item() + np.arange(le[j])).all()
    used = np.zeros(f.cap, bool)
    for j in range(2 * B):
        used[st[j]:st[j] + le[j]] = True
    assert (gi[~used] == -1).all()


@pytest.mark.parametrize("B,T", [(9, 30), (5, 64), (3, 77)])
def test_self_attention_packed_equals_padded_with_key_mask(B, T):
    H, D = 12, 768
    ln = _lens(B, T, 2)
    p = Pack.from_lens(ln.tolist(), T, "cuda")
    keep = (torch.arange(T)[None, :] < ln[:, None])
    g = torch.Generator().manual_seed(3)
    qkv = (torch.randn(B * T, 3 * D, generator=g) * 0.6).to(BF16).cuda()
    dout = (torch.randn(B * T, D, generator=g) * 0.3).to(BF16).cuda()
    km = keep.to(torch.int32).cuda().contiguous()
    scale = 1 / math.sqrt(64)
    o, lse = Fx.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, H, T, T, scale, key_keep=km)
    dq = torch.empty_like(qkv)
    Fx.attn_bwd(dout, qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, lse, dq[:, :D], dq[:, D:2 * D], dq[:, 2 * D:], B, H, T, T, scale,
                key_keep=km)
    # packed: the same tokens, rows compacted
    qkv_p = pack_rows(qkv.view(B, T, 3 * D), p).contiguous()
    dout_p = pack_rows((dout.view(B, T, D) * keep.cuda()[..., None]).to(BF16), p).contiguous()
    op, lsep = Fx.attn_fwd(qkv_p[:, :D], qkv_p[:, D:2 * D], qkv_p[:, 2 * D:], B, H, T, T, scale, q_pack=p.pair, k_pack=p.pair)
    dqp = torch.zeros_like(qkv_p)
    Fx.attn_bwd(dout_p, qkv_p[:, :D], qkv_p[:, D:2 * D], qkv_p[:, 2 * D:], op, lsep, dqp[:, :D], dqp[:, D:2 * D], dqp[:, 2 * D:], B, H, T, T,
                scale, q_pack=p.pair, k_pack=p.pair)
    kc = keep.cuda()
    assert _rel(unpack(op, p)[kc], o.view(B, T, D)[kc]) <= 2e-3
    # the padded run back-propagates dout of padded QUERY rows too (into real keys); zero them for the comparison
    dq2 = torch.empty_like(qkv)
    dz = (dout.view(B, T, D) * kc[..., None]).to(BF16).view(B * T, D)
    Fx.attn_bwd(dz, qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, lse, dq2[:, :D], dq2[:, D:2 * D], dq2[:, 2 * D:], B, H, T, T, scale,
                key_keep=km)
    got = unpack(dqp, p)[kc]
    want = dq2.view(B, T, 3 * D)[kc]
    assert _rel(got, want) <= 5e-3, _rel(got, want)
    for b in range(B):
        assert torch.allclose(lsep[b, :, :ln[b]], lse[b, :, :ln[b]], atol=1e-4)


def test_grouped_cross_attention_packed_queries_equal_padded():
    B, U, T, N, H, D = 24, 6, 30, 197, 12, 768
    ln = _lens(B, T, 4)
    p = Pack.from_lens(ln.tolist(), T, "cuda")
    keep = (torch.arange(T)[None, :] < ln[:, None]).cuda()
    g = torch.Generator().manual_seed(5)
    q = (torch.randn(B * T, D, generator=g) * 0.6).to(BF16).cuda()
    kv = (torch.randn(U * N, 2 * D, generator=g) * 0.6).to(BF16).cuda()
    dout = ((torch.randn(B, T, D, generator=g) * 0.3).cuda() * keep[..., None]).to(BF16).view(B * T, D)
    index = torch.randint(0, U, (B,), generator=g).to(torch.int32).cuda()
    groups = Fx.kv_groups(index, U)
    ek = torch.ones(U, N, dtype=torch.int32, device="cuda")
    scale = 1 / math.sqrt(64)
    o, lse = Fx.attn_fwd(q, kv[:, :D], kv[:, D:], B, H, T, N, scale, key_keep=ek, groups=groups)
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    Fx.attn_bwd(dout, q, kv[:, :D], kv[:, D:], o, lse, dq, dkv[:, :D], dkv[:, D:], B, H, T, N, scale, key_keep=ek, groups=groups)
    qp = pack_rows(q.view(B, T, D), p).contiguous()
    dp = pack_rows(dout.view(B, T, D), p).contiguous()
    op, lsep = Fx.attn_fwd(qp, kv[:, :D], kv[:, D:], B, H, T, N, scale, key_keep=ek, groups=groups, q_pack=p.pair)
    dqp, dkvp = torch.zeros_like(qp), torch.empty_like(kv)
    Fx.attn_bwd(dp, qp, kv[:, :D], kv[:, D:], op, lsep, dqp, dkvp[:, :D], dkvp[:, D:], B, H, T, N, scale, key_keep=ek, groups=groups,
                q_pack=p.pair)
    assert _rel(unpack(op, p)[keep], o.view(B, T, D)[keep]) <= 2e-3
    assert _rel(unpack(dqp, p)[keep], dq.view(B, T, D)[keep]) <= 5e-3
    assert _rel(dkvp, dkv) <= 5e-3   # (padded query rows carried zero dout above, so they add nothing to dK / dV either way)


def _roberta(layers, fusion_layer):
    from xfm_amd.xroberta import RobertaConfig, RobertaForMaskedLM
    return RobertaForMaskedLM(RobertaConfig(num_hidden_layers=layers, fusion_layer=fusion_layer, vocab_size=4096))


# key-bias gradients are analytically zero (softmax is invariant to a per-query constant): what the tensors hold is bf16
# rounding noise of the dK rows, different on every path -- they are left out of the relative comparisons below
def _grads(m):
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None and float(p.grad.abs().max()) > 0}


def test_text_tower_packed_equals_padded_forward_and_backward():
    torch.manual_seed(0)
    m = _roberta(2, 2).cuda().finalize().eval()
    B, T = 10, 30
    b = syn.pretrain_batch(B, seed=31, vocab=4096)
    ids, atts = b["text_ids"].cuda(), b["text_atts"].cuda()
    ln = b["text_atts"].sum(1)
    keep = atts.bool()
    w = torch.randn(B, T, 768, device="cuda") * keep[..., None]
    out = m.bert(ids, attention_mask=atts).last_hidden_state
    (out.float() * w).sum().backward()
    g_pad = _grads(m)
    m.zero_grad()
    p = Pack.from_lens(ln.tolist(), T, "cuda")
    rows = m.bert(ids, attention_mask=None, pack=p).last_hidden_state
    assert rows.shape == (p.cap, 768)
    got = unpack(rows, p)
    assert _rel(got[keep], out[keep]) <= 2e-3, _rel(got[keep], out[keep])
    (got.float() * w).sum().backward()
    g_pack = _grads(m)
    assert set(g_pack) == set(g_pad)
    worst = max((_rel(g_pack[n], g_pad[n]), n) for n in g_pad if float(g_pad[n].float().norm()) > 1e-6 and "key.bias" not in n)
    print("text tower, packed vs padded: worst gradient rel-L2", worst)
    assert worst[0] <= 1e-2, worst
    # grad_batch on packed rows: only the first sequences back-propagate
    m.zero_grad()
    rows = m.bert(ids, attention_mask=None, pack=p, grad_batch=4).last_hidden_state
    (unpack(rows, p).float() * w).sum().backward()
    g_head = _grads(m)
    m.zero_grad()
    p4 = Pack.from_lens(ln[:4].tolist(), T, "cuda")
    rows4 = m.bert(ids[:4], attention_mask=None, pack=p4).last_hidden_state
    (unpack(rows4, p4).float() * w[:4]).sum().backward()
    g4 = _grads(m)
    worst = max((_rel(g_head[n], g4[n]), n) for n in g4 if float(g4[n].float().norm()) > 1e-6 and "key.bias" not in n)
    assert worst[0] <= 1e-2, worst


def test_fusion_tower_packed_with_slack_block_equals_padded():
    torch.manual_seed(0)
    m = _roberta(2, 0).cuda().finalize().eval()
    B, T, N = 8, 30, 197
    b = syn.pretrain_batch(B, seed=37, vocab=4096)
    ln = b["text_atts"].sum(1)
    g = torch.Generator().manual_seed(9)
    sel = torch.randint(0, B, (B,), generator=g)
    lens2 = torch.cat([ln, ln[sel]])
    x = (torch.randn(2 * B, T, 768, generator=g) * 0.7).to(BF16).cuda()
    keep = (torch.arange(T)[None, :] < lens2[:, None]).cuda()
    x = (x * keep[..., None]).to(BF16)
    img = (torch.randn(B, N, 768, generator=g) * 0.7).to(BF16).cuda().requires_grad_(True)
    iatts = torch.ones(B, N, dtype=torch.long, device="cuda")
    index = torch.cat([torch.arange(B), torch.randint(0, B, (B,), generator=g)]).to(torch.int32).cuda()
    w = torch.randn(2 * B, T, 768, device="cuda") * keep[..., None]
    xa = x.clone().requires_grad_(True)
    out = m.bert(encoder_embeds=xa, attention_mask=keep.long(), encoder_hidden_states=img, encoder_attention_mask=iatts,
                 encoder_batch_index=index).last_hidden_state
    (out.float() * w).sum().backward()
    g_pad, dimg_pad, dx_pad = _grads(m), img.grad.clone(), xa.grad.clone()
    m.zero_grad()
    img.grad = None
    ld = ln.to(torch.int32).cuda()
    p = Pack.concat([(ld, int(ln.sum()), ln.tolist()), (ld.index_select(0, sel.cuda()), B * int(ln.max()), None)], T)   # 2nd block: slack
    assert p.cap > int(lens2.sum())
    xr = pack_rows(x, p).detach().requires_grad_(True)
    rows = m.bert(encoder_embeds=xr, attention_mask=None, encoder_hidden_states=img, encoder_attention_mask=iatts,
                  encoder_batch_index=index, pack=p).last_hidden_state
    got = unpack(rows, p)
    assert bool(torch.isfinite(rows.float()).all())
    assert _rel(got[keep], out[keep]) <= 2e-3, _rel(got[keep], out[keep])
    (got.float() * w).sum().backward()
    g_pack = _grads(m)
    assert set(g_pack) == set(g_pad)
    worst = max((_rel(g_pack[n], g_pad[n]), n) for n in g_pad if float(g_pad[n].float().norm()) > 1e-6 and "key.bias" not in n)
    print("fusion tower, packed vs padded: worst gradient rel-L2", worst)
    assert worst[0] <= 1e-2, worst
    assert _rel(img.grad, dimg_pad) <= 1e-2
    assert _rel(unpack(xr.grad, p)[keep], dx_pad[keep]) <= 1e-2
    used = p.gather_index(p) >= 0
    assert float(xr.grad[~used].float().abs().max()) == 0.0, "slack rows must carry exactly zero gradient"


@pytest.mark.parametrize("native", [True, False])
def test_fusion_tower_image_major_layout_with_row_ranges_equals_padded(native):
    """Sequences laid out image by image + cross-attention per image on contiguous query rows (encoder_row_ranges) against the padded
    tower in the caller's sequence order; both the per-layer native executor and the kernel-by-kernel path."""
    import xfm_amd.xroberta as XR
    from xfm_amd.packing import image_major_layout
    torch.manual_seed(0)
    m = _roberta(2, 0).cuda().finalize().eval()
    B, U, T, N = 13, 5, 30, 197
    ln = _lens(B, T, 6)
    g = torch.Generator().manual_seed(11)
    seq_img = torch.randint(0, U, (B,), generator=g)
    seq_img[0], seq_img[1] = U - 1, U - 1
    seq_img[seq_img == 2] = 3                       # image 2 has no sequence at all
    keep = (torch.arange(T)[None, :] < ln[:, None]).cuda()
    x = ((torch.randn(B, T, 768, generator=g) * 0.7).cuda() * keep[..., None]).to(BF16)
    img = (torch.randn(U, N, 768, generator=g) * 0.7).to(BF16).cuda()
    iatts = torch.ones(U, N, dtype=torch.long, device="cuda")
    w = torch.randn(B, T, 768, device="cuda") * keep[..., None]
    xa, ia = x.clone().requires_grad_(True), img.clone().requires_grad_(True)
    out = m.bert(encoder_embeds=xa, attention_mask=keep.long(), encoder_hidden_states=ia, encoder_attention_mask=iatts,
                 encoder_batch_index=seq_img.to(torch.int32).cuda()).last_hidden_state
    (out.float() * w).sum().backward()
    g_pad, dimg_pad, dx_pad = _grads(m), ia.grad.clone(), xa.grad.clone()
    m.zero_grad()
    pack, order, pos_of, meta, ranges = image_major_layout(ln.tolist(), seq_img.tolist(), U, T, "cuda", extra=(seq_img.tolist(),))
    assert int(ranges[1][2]) == 0 and ranges[2] == int(ranges[1].max())
    xs = pack_rows(x[torch.tensor(order)], pack).detach().requires_grad_(True)
    ib = img.clone().requires_grad_(True)
    old = XR._NATIVE_LAYERS
    XR._NATIVE_LAYERS = native
    try:
        rows = m.bert(encoder_embeds=xs, attention_mask=None, encoder_hidden_states=ib, encoder_attention_mask=iatts,
                      encoder_batch_index=meta[1].contiguous(), pack=pack, encoder_row_ranges=ranges).last_hidden_state
        got = unpack(rows, pack)[torch.tensor(pos_of)]            # back to the caller's sequence order
        assert _rel(got[keep], out[keep]) <= 2e-3, _rel(got[keep], out[keep])
        (got.float() * w).sum().backward()
    finally:
        XR._NATIVE_LAYERS = old
    g_pack = _grads(m)
    assert set(g_pack) == set(g_pad)
    worst = max((_rel(g_pack[n], g_pad[n]), n) for n in g_pad if float(g_pad[n].float().norm()) > 1e-6 and "key.bias" not in n)
    print("fusion tower, image-major ranges vs padded: worst gradient rel-L2", worst)
    assert worst[0] <= 1e-2, worst
    assert _rel(ib.grad, dimg_pad) <= 1e-2
    assert _rel(unpack(xs.grad, pack)[torch.tensor(pos_of)][keep], dx_pad[keep]) <= 1e-2


def test_last_layer_on_selected_rows_equals_full_tower():
    """output_rows: the last layer computed on the rows somebody reads ([CLS] of some sequences, M arbitrary positions -- duplicates
    included -- of the others) against the full packed tower read at those rows: outputs, every parameter gradient, the gradient of
    the image states and of the input rows (the unselected rows' outputs get no cotangent in either run)."""
    from xfm_amd.packing import image_major_layout
    torch.manual_seed(0)
    m = _roberta(3, 0).cuda().finalize().eval()
    B, U, T, N, M = 12, 4, 30, 197, 5
    ln = _lens(B, T, 8, lo=6)
    g = torch.Generator().manual_seed(5)
    seq_img = torch.randint(0, U, (B,), generator=g)
    n_cls = 7                                                   # sequences 0..6 are read at [CLS] only, 7..11 at M positions
    pos = torch.stack([torch.randint(0, int(ln[n_cls + j]), (M,), generator=g) for j in range(B - n_cls)])
    pos[0, 1] = pos[0, 0]                                       # a duplicate slot
    pos[1, :] = 0                                               # all padding slots -> [CLS]
    sel_off = list(range(n_cls)) + [n_cls + j * M for j in range(B - n_cls)]
    sel_len = [1] * n_cls + [M] * (B - n_cls)
    pack, order, pos_of, meta, _ = image_major_layout(ln.tolist(), seq_img.tolist(), U, T, "cuda",
                                                      extra=(seq_img.tolist(), sel_off, sel_len))
    keep = torch.arange(T)[None, :] < ln[:, None]
    x = ((torch.randn(B, T, 768, generator=g) * 0.7) * keep[..., None]).to(BF16).cuda()
    img = (torch.randn(U, N, 768, generator=g) * 0.7).to(BF16).cuda()
    iatts = torch.ones(U, N, dtype=torch.long, device="cuda")
    start_of = pack.start.index_select(0, torch.tensor(pos_of, device="cuda"))
    rows = torch.cat([start_of[:n_cls], (start_of[n_cls:, None] + pos.cuda().to(torch.int32)).reshape(-1)]).to(torch.int32)
    S = rows.numel()
    w = torch.randn(S, 768, device="cuda")
    xr = pack_rows(x[torch.tensor(order)], pack)
    res = {}
    for mode in ("full", "rows"):
        m.zero_grad()
        xa, ia = xr.detach().clone().requires_grad_(True), img.clone().requires_grad_(True)
        kw = dict(encoder_embeds=xa, attention_mask=None, encoder_hidden_states=ia, encoder_attention_mask=iatts,
                  encoder_batch_index=meta[1].contiguous(), pack=pack)
        if mode == "full":
            out = m.bert(**kw).last_hidden_state.index_select(0, rows.long())
        else:
            out = m.bert(output_rows=(rows, meta[2].contiguous(), meta[3].contiguous(), M), **kw).last_hidden_state
        assert out.shape == (S, 768)
        (out.float() * w).sum().backward()
        res[mode] = (out.detach().float(), _grads(m), ia.grad.clone(), xa.grad.clone())
    assert _rel(res["rows"][0], res["full"][0]) <= 2e-3, _rel(res["rows"][0], res["full"][0])
    gf, gr = res["full"][1], res["rows"][1]
    assert set(gf) == set(gr)
    worst = max((_rel(gr[n], gf[n]), n) for n in gf if float(gf[n].float().norm()) > 1e-6 and "key.bias" not in n)
    print("last layer on selected rows vs full tower: worst gradient rel-L2", worst)
    assert worst[0] <= 1e-2, worst
    assert _rel(res["rows"][2], res["full"][2]) <= 1e-2
    assert _rel(res["rows"][3], res["full"][3]) <= 1e-2


def test_pretrain_step_packed_rows_equals_padded_step():
    from xfm_amd.model_pretrain import XFM
    z, meta = load("pretrain_small")
    cfg = {"use_beit_v2": True, "image_res": 224, "patch_size": 16, "local_attn_depth": -1, "text_encoder": "roberta-base",
           "text_num_hidden_layers": meta["text_layers"], "text_fusion_start_at": meta["text_layers"],
           "fusion_num_hidden_layers": meta["fusion_layers"], "fusion_fusion_start_at": 0, "embed_dim": 256, "temp": 0.07,
           "learnable_temp": True, "max_temp": 0.5, "min_temp": 0.001, "vision_depth": meta.get("vit_depth", 12)}
    B = 8
    m = XFM(cfg)
    m.load_state_dict(state_from_spec(meta["spec"]), strict=True)
    m.cuda().finalize().eval()
    hb = syn.pretrain_batch(B, seed=77)
    b = {k: v.cuda() for k, v in hb.items()}
    masks = syn.mim_block_mask(B, 14, 75, seed=77)
    neg = ([3, 0, 1, 2, 7, 4, 5, 6], [1, 2, 3, 0, 5, 6, 7, 4])

    def run(text_lens):
        m.zero_grad()
        out = m(b["image"], b["text_ids"], b["text_atts"], text_ids_masked=b["text_ids_masked"], masked_pos=b["masked_pos"],
                masked_ids=b["masked_ids"], ret_mim_loss=True, data_source="image", ids_mask=masks, neg_idx=neg, text_lens=text_lens)
        sum(out[k] for k in ("loss_itc", "loss_itm", "loss_mlm", "loss_mim")).backward()
        torch.cuda.synchronize()
        return {k: float(out[k]) for k in ("loss_itc", "loss_itm", "loss_mlm", "loss_mim")}, _grads(m)

    l_pad, g_pad = run(None)
    l_pack, g_pack = run(hb["text_atts"].sum(1))
    print(l_pad, l_pack)
    for k in l_pad:
        assert abs(l_pad[k] - l_pack[k]) <= 2e-3 * max(abs(l_pad[k]), 1.0), (k, l_pad, l_pack)
    assert set(g_pad) == set(g_pack)
    # (key biases: analytically zero gradients -- softmax ignores a per-query constant -- hold rounding noise only)
    bad = [(n, round(_rel(g_pack[n], g_pad[n]), 4)) for n in g_pad
           if float(g_pad[n].float().norm()) > 1e-5 and "key.bias" not in n and _rel(g_pack[n], g_pad[n]) > 2e-2]
    assert not bad, bad[:10]
