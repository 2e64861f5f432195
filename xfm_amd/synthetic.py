"""Build-owned deterministic data: formula weights and synthetic image-text batches.

A counter-based generator (splitmix64 over `crc32(name) , element index`) written against numpy
integer arithmetic only, so that the build container, the GPU box and any later round regenerate
bit-identical tensors without shipping them.  Used by the golden-fixture generator (weights are
loaded into the reference with load_state_dict(strict=True)), the parity tests, smoke() and bench.py.

Batch layout mirrors what the reference's collate emits for the image-text pre-training source
(dataset/pretrain_dataset.py:264-312).
"""
import zlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform01(name, n, seed=0):
    """n doubles in [0,1), a pure function of (name, seed, index)."""
    with np.errstate(over="ignore"):
        key = np.uint64(zlib.crc32(name.encode()) + (int(seed) << 32))
        base = _splitmix64(np.array([key], dtype=np.uint64))[0]
        i = np.arange(n, dtype=np.uint64)
        z = _splitmix64(i * np.uint64(0x2545F4914F6CDD1D) + base)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def symmetric(name, shape, scale, seed=0):
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(name, n, seed)
    return torch.from_numpy(((u * 2.0 - 1.0) * scale).astype(np.float32).reshape(shape))


def gaussian(name, shape, std=1.0, seed=0):
    """Box-Muller over two formula streams."""
    n = int(np.prod(shape))
    u1 = np.maximum(uniform01(name + "#a", n, seed), 1e-12)
    u2 = uniform01(name + "#b", n, seed)
    g = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy((g * std).astype(np.float32).reshape(shape))


def formula_tensor(name, ref, seed=0):
    """A deterministic value for state_dict entry `name` shaped/typed like `ref` (an existing tensor).
    Scales are chosen to keep 1-2 layer stacks numerically lively (non-trivial attention, biases and
    layer-scale all matter), not to imitate any trained checkpoint."""
    shape = tuple(ref.shape)
    leaf = name.split(".")[-1]
    if not ref.dtype.is_floating_point:
        return ref.clone()  # integer buffers (relative_position_index, position_ids) keep their built values
    if name == "temp":
        return torch.tensor(0.07, dtype=ref.dtype)
    if "gamma_" in leaf:
        return 0.1 + symmetric(name, shape, 0.03, seed)
    if "relative_position_bias_table" in leaf:
        return symmetric(name, shape, 0.5, seed)
    if leaf in ("cls_token", "mask_token"):
        return symmetric(name, shape, 0.5, seed)
    is_norm = any(t in name for t in ("LayerNorm", "layer_norm", "norm1", "norm2", "fc_norm")) or \
        (name.startswith(("itm_head.1", "bbox_head.1")))
    if is_norm and leaf == "weight":
        return 1.0 + symmetric(name, shape, 0.2, seed)
    if leaf in ("bias", "q_bias", "v_bias"):
        return symmetric(name, shape, 0.1, seed)
    if "embeddings" in name:
        return symmetric(name, shape, 0.1, seed)
    # dense / conv weights: uniform with std ~ 0.05 (xavier-ish for fan_in 768)
    return symmetric(name, shape, 0.05 * 1.7320508, seed)


def formula_state_dict(reference_state, seed=0):
    """Map every entry of an existing state_dict (shapes/dtypes only are used) to its formula value."""
    out = {}
    for k, v in reference_state.items():
        t = formula_tensor(k, v, seed)
        if k.endswith(("lm_head.decoder.bias", "lm_cap_head.decoder.bias", "predictions.decoder.bias")):
            # tied to `<head>.bias` (xroberta.py:1322-1323, xbert.py:690-691)
            t = formula_tensor(k[: -len("decoder.bias")] + "bias", v, seed)
        out[k] = t.to(v.dtype)
    return out


# ----------------------------------------------------------------------------------------------
# synthetic pre-training batch
# ----------------------------------------------------------------------------------------------
BOS, PAD, EOS, MASK_ID, VOCAB = 0, 1, 2, 50264, 50265


def pretrain_batch(batch_size, seed=1234, image_res=224, max_tokens=30, max_masks=15, min_len=8, vocab=VOCAB,
                   with_image=True):
    """image fp32 [B,3,R,R] ~ N(0,1); text_ids int64 [B,T] (<s> ... </s> pad); text_atts prefix ones;
    masked_pos int64 [B,M] distinct in [1,len) zero padded; masked_ids int64 [B,M] (-100 padded);
    text_ids_masked = ids with <mask> at masked_pos."""
    B, T, M = batch_size, max_tokens, max_masks
    tag = f"batch{seed}"
    out = {}
    if with_image:
        out["image"] = gaussian(tag + ".image", (B, 3, image_res, image_res), 1.0)
    lens = (min_len + np.floor(uniform01(tag + ".len", B) * (T - min_len + 1))).astype(np.int64)
    lens = np.clip(lens, min_len, T)
    tok = (3 + np.floor(uniform01(tag + ".tok", B * T) * (vocab - 1 - 3))).astype(np.int64).reshape(B, T)
    ids = np.full((B, T), PAD, dtype=np.int64)
    atts = np.zeros((B, T), dtype=np.int64)
    mpos = np.zeros((B, M), dtype=np.int64)
    mids = np.full((B, M), -100, dtype=np.int64)
    ids_masked = None
    order_u = uniform01(tag + ".perm", B * T).reshape(B, T)
    nm_u = uniform01(tag + ".nmask", B)
    for b in range(B):
        L = int(lens[b])
        ids[b, :L] = tok[b, :L]
        ids[b, 0] = BOS
        ids[b, L - 1] = EOS
        atts[b, :L] = 1
        cand = np.arange(1, L)  # positions [1, len)
        n = int(min(M, max(1, np.floor(nm_u[b] * min(M, L - 1)) + 1)))
        pick = cand[np.argsort(order_u[b, 1:L], kind="stable")[:n]]
        pick.sort()
        mpos[b, :n] = pick
        mids[b, :n] = ids[b, pick]
    ids_masked = ids.copy()
    for b in range(B):
        n = int((mids[b] != -100).sum())
        ids_masked[b, mpos[b, :n]] = MASK_ID if vocab == VOCAB else vocab - 1
    out.update(text_ids=torch.from_numpy(ids), text_atts=torch.from_numpy(atts),
               text_ids_masked=torch.from_numpy(ids_masked), masked_pos=torch.from_numpy(mpos),
               masked_ids=torch.from_numpy(mids))
    return out


def mim_block_mask(batch_size, grid=14, num_masking=75, seed=1234):
    """A deterministic stand-in for the MaskingGenerator draws (masking_generator.py:27-105): exactly
    `num_masking` of grid*grid patches per sample, as a few rectangular blocks topped up with singles.
    Used for parity fixtures and the synthetic benchmark; the product path draws its own masks."""
    B = batch_size
    u = uniform01(f"mim{seed}", B * 64).reshape(B, 64)
    out = np.zeros((B, grid, grid), dtype=bool)
    for b in range(B):
        k = 0
        while out[b].sum() < num_masking and k < 60:
            h = 2 + int(u[b, k] * 6); w = 2 + int(u[b, k + 1] * 6)
            y = int(u[b, k + 2] * (grid - h + 1)); x = int(u[b, k + 3] * (grid - w + 1))
            k += 4
            blk = np.zeros((grid, grid), dtype=bool)
            blk[y:y + h, x:x + w] = True
            if (out[b] | blk).sum() <= num_masking:
                out[b] |= blk
        flat = out[b].reshape(-1)
        need = num_masking - int(flat.sum())
        if need > 0:
            free = np.flatnonzero(~flat)
            flat[free[:need]] = True
    return torch.from_numpy(out.reshape(B, grid * grid))


def vqa_inputs(B=3, image_res=224):
    """VQA fixture inputs shared by the golden generator and the CPU / GPU parity tests: B images + questions, k[b] answers per
    question with annotator weights, and a candidate answer list for the inference-time ranking (top `topk`)."""
    from types import SimpleNamespace as NS
    b = pretrain_batch(B, seed=91, image_res=image_res)
    k = [2, 1, 3][:B]
    ans = pretrain_batch(sum(k), seed=92, max_tokens=7, min_len=3, with_image=False)
    cand = pretrain_batch(5, seed=93, max_tokens=6, min_len=3, with_image=False)
    weights = torch.tensor([0.6, 0.4, 1.0, 0.5, 0.3, 0.2][:sum(k)])
    return NS(image=b["image"], q_ids=b["text_ids"], q_atts=b["text_atts"], k=k, a_ids=ans["text_ids"], a_atts=ans["text_atts"],
              weights=weights, c_ids=cand["text_ids"], c_atts=cand["text_atts"], topk=3)


def retrieval_eval_inputs(n_img=5, n_txt=8):
    """Retrieval-evaluation fixture inputs: n_img images, n_txt captions (caption j describes image txt2img[j]), k_test = 3."""
    from types import SimpleNamespace as NS
    b = pretrain_batch(n_txt, seed=97)
    txt2img = [j % n_img for j in range(n_txt)]
    img2txt = [[j for j in range(n_txt) if txt2img[j] == i] for i in range(n_img)]
    return NS(image=b["image"][:n_img], text_ids=b["text_ids"], text_atts=b["text_atts"], k_test=3, txt2img=txt2img, img2txt=img2txt)


def vqa_batch(batch_size, seed=1234, image_res=480, max_tokens=40, max_answers=10, answer_len=8):
    """BASELINE configs[3] synthetic batch (the collate of dataset/vqa_dataset.py as VQA.py:45-52 consumes it): B images +
    questions of up to `max_tokens` tokens, k[b] ~ U[1, max_answers] answers per question (<s> ... </s>, up to `answer_len` tokens,
    pad = 1) and one annotator weight per answer (the weights of a question sum to 1)."""
    from types import SimpleNamespace as NS
    b = pretrain_batch(batch_size, seed=seed, image_res=image_res, max_tokens=max_tokens)
    k = [int(v) for v in (1 + np.floor(uniform01(f"vqa{seed}.k", batch_size) * max_answers)).clip(1, max_answers)]
    ans = pretrain_batch(sum(k), seed=seed + 1, max_tokens=answer_len, min_len=3, with_image=False)
    w = uniform01(f"vqa{seed}.w", sum(k)) + 0.1
    o = 0
    for n in k:
        w[o:o + n] /= w[o:o + n].sum()
        o += n
    return NS(image=b["image"], q_ids=b["text_ids"], q_atts=b["text_atts"], k=k, a_ids=ans["text_ids"], a_atts=ans["text_atts"],
              weights=torch.from_numpy(w.astype(np.float32)))


def region_case(n_images, grid=14):
    """Inputs of the vision tower's region call form (beit2.py:467-475): bs = n_images + 2 samples over n_images images
    (`idx_to_group_img` [bs], some images used twice) and a ragged region mask per sample (`image_atts` [bs, 1 + grid^2] of 0 / 1: a
    rectangle of patches, column 0 = the cls slot, always 1)."""
    bs = n_images + 2
    idx = torch.tensor([(3 * i + 1) % n_images for i in range(bs)], dtype=torch.long)
    atts = torch.zeros(bs, 1 + grid * grid, dtype=torch.long)
    atts[:, 0] = 1
    for i in range(bs):
        y0, x0 = (2 * i) % (grid - 4), (3 * i + 1) % (grid - 5)
        h, w = 3 + i % 4, 4 + (2 * i) % 5
        m = torch.zeros(grid, grid, dtype=torch.long)
        m[y0:y0 + h, x0:x0 + w] = 1
        atts[i, 1:] = m.reshape(-1)
    return idx, atts
