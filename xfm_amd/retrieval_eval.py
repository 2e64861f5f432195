"""Image-text retrieval evaluation on the HIP towers: the k-test re-rank of Retrieval.py:76-184 and its recall metrics (:187-240).

Dual-encoder features give an [images x texts] similarity matrix; for every image the k_test best texts (and for every text the
k_test best images) are re-scored by the fusion tower + ITM head, everything else keeps -100.  The reference walks one row at a time
(one fusion pass of k_test pairs per row); here `rows_per_pass` rows share a fusion pass -- the same pairs, the same scores, far
fewer launches -- and a rank only scores its own slice of rows, to be summed across ranks exactly like the reference
(score matrices are -100-filled, so the reference's all-reduce SUM is reproduced by filling non-owned rows with 0 before the sum;
see `evaluation(..., reduce=True)`)."""
import numpy as np
import torch
import torch.distributed as dist


def _rank_slice(n, rank, world):
    """Retrieval.py:133-136: step = n // world + 1."""
    step = n // world + 1
    start = rank * step
    return start, min(n, start + step)


@torch.no_grad()
def encode(model, images, text_ids, text_atts, image_bs=64, text_bs=256):
    """Retrieval.py:92-126: token-level embeddings of both towers and their normalised ITC features."""
    text_embeds, text_feats = [], []
    for i in range(0, text_ids.shape[0], text_bs):
        e = model.get_text_embeds(text_ids[i:i + text_bs], text_atts[i:i + text_bs])
        text_embeds.append(e)
        text_feats.append(model.get_features(text_embeds=e))
    image_embeds, image_feats = [], []
    for i in range(0, images.shape[0], image_bs):
        e, _ = model.get_vision_embeds(images[i:i + image_bs])
        image_embeds.append(e)
        image_feats.append(model.get_features(image_embeds=e))
    return torch.cat(image_embeds), torch.cat(image_feats), torch.cat(text_embeds), torch.cat(text_feats)


def _itm_scores(model, image_embeds, text_embeds, text_atts):
    """ITM logit of 'match' for aligned (image, text) pairs (Retrieval.py:144-149)."""
    image_atts = torch.ones(image_embeds.shape[:2], dtype=torch.long, device=image_embeds.device)
    out = model.get_cross_embeds(image_embeds, image_atts, text_embeds=text_embeds, text_atts=text_atts)
    return model.itm_head(out[:, 0, :])[:, 1].float()


@torch.no_grad()
def evaluation(model, images, text_ids, text_atts, k_test, image_bs=64, text_bs=256, rows_per_pass=8, rank=0, world=1,
               reduce=False):
    """-> (score_matrix_i2t [images, texts], score_matrix_t2i [texts, images]) as numpy, -100 where a pair was not re-ranked.
    With world > 1 each rank fills its own row slice; `reduce=True` sums the slices over the default process group."""
    model.eval()
    image_embeds, image_feats, text_embeds, text_feats = encode(model, images, text_ids, text_atts, image_bs, text_bs)
    sims = image_feats.float() @ text_feats.float().t()
    n_img, n_txt = sims.shape
    k = k_test
    dev = sims.device
    fill = -100.0 if not (reduce and world > 1) else 0.0  # see module docstring
    i2t = torch.full((n_img, n_txt), fill, device=dev)
    start, end = _rank_slice(n_img, rank, world)
    for r0 in range(start, end, rows_per_pass):
        rows = torch.arange(r0, min(end, r0 + rows_per_pass), device=dev)
        topk_idx = sims[rows].topk(k, dim=1).indices                       # [R, k] text indices
        flat = topk_idx.reshape(-1)
        score = _itm_scores(model, image_embeds[rows].repeat_interleave(k, dim=0), text_embeds[flat], text_atts[flat])
        if fill == 0.0:
            i2t[rows] = -100.0
        i2t[rows.repeat_interleave(k), flat] = score
    t2i = torch.full((n_txt, n_img), fill, device=dev)
    sims_t = sims.t()
    start, end = _rank_slice(n_txt, rank, world)
    for r0 in range(start, end, rows_per_pass):
        rows = torch.arange(r0, min(end, r0 + rows_per_pass), device=dev)
        topk_idx = sims_t[rows].topk(k, dim=1).indices                     # [R, k] image indices
        flat = topk_idx.reshape(-1)
        rep = rows.repeat_interleave(k)
        score = _itm_scores(model, image_embeds[flat], text_embeds[rep], text_atts[rep])
        if fill == 0.0:
            t2i[rows] = -100.0
        t2i[rep, flat] = score
    if reduce and world > 1:
        dist.all_reduce(i2t)
        dist.all_reduce(t2i)
    return i2t.cpu().numpy(), t2i.cpu().numpy()


def itm_eval(scores_i2t, scores_t2i, txt2img, img2txt):
    """Recall@{1,5,10} both ways (Retrieval.py:187-240): an image's rank is that of its BEST ground-truth caption."""
    ranks = np.zeros(scores_i2t.shape[0])
    for index, score in enumerate(scores_i2t):
        inds = np.argsort(score)[::-1]
        ranks[index] = min(int(np.where(inds == i)[0][0]) for i in img2txt[index])
    tr1, tr5, tr10 = (100.0 * float(np.mean(ranks < n)) for n in (1, 5, 10))
    ranks = np.zeros(scores_t2i.shape[0])
    for index, score in enumerate(scores_t2i):
        inds = np.argsort(score)[::-1]
        ranks[index] = np.where(inds == txt2img[index])[0][0]
    ir1, ir5, ir10 = (100.0 * float(np.mean(ranks < n)) for n in (1, 5, 10))
    tr_mean, ir_mean = (tr1 + tr5 + tr10) / 3, (ir1 + ir5 + ir10) / 3
    return {'txt_r1': tr1, 'txt_r5': tr5, 'txt_r10': tr10, 'txt_r_mean': tr_mean, 'img_r1': ir1, 'img_r5': ir5, 'img_r10': ir10,
            'img_r_mean': ir_mean, 'r_mean': (tr_mean + ir_mean) / 2}
