"""Which gradient tensors of a config-shape step sit furthest above the reference's own mixed-precision floor?  Runs the retrieval
(configs[2]) or headline pre-training step exactly as tests/test_hip_configs.py does and lists err / floor per tensor, worst first.
Run on the GPU box (env knobs apply):  python tools/grad_ratio.py retrieval_cfg|pretrain_cfg [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from golden_util import load, rel_l2  # noqa: E402
from xfm_amd import synthetic as syn  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "retrieval_cfg"
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    z, meta = load(name)
    if name == "retrieval_cfg":
        from test_hip_configs import _cfg, _formula
        from xfm_amd.model_retrieval import XFMForRetrieval
        B, R, T = 32, 384, 40
        m = XFMForRetrieval(_cfg(R, 12, 12, 12))
        _formula(m)
        m.cuda().finalize().eval()
        b = syn.pretrain_batch(B, seed=384, image_res=R, max_tokens=T)
        idx = torch.tensor(meta["idx"])
        itc, itm = m(b["image"].cuda(), b["text_ids"].cuda(), b["text_atts"].cuda(), idx=idx.cuda(),
                     neg_idx=(meta["image_neg_idx"], meta["text_neg_idx"]))
        (itc + itm).backward()
        print(f"itc {float(itc):.5f} / {float(z['loss_itc']):.5f}   itm {float(itm):.5f} / {float(z['loss_itm']):.5f}")
    else:
        from test_hip_modules import _pretrain_cfg, _load_into
        from xfm_amd.model_pretrain import XFM
        B = meta["B"]
        m = XFM(_pretrain_cfg(meta))
        _load_into(m, meta["spec"])
        m.cuda().finalize().eval()
        seed = meta.get("seed", 1234)
        hb = syn.pretrain_batch(B, seed=seed)
        b = {k: v.cuda() for k, v in hb.items()}
        losses = m(b["image"], b["text_ids"], b["text_atts"], text_ids_masked=b["text_ids_masked"], masked_pos=b["masked_pos"],
                   masked_ids=b["masked_ids"], ret_mim_loss=True, data_source="image", ids_mask=syn.mim_block_mask(B, 14, 75, seed=seed),
                   neg_idx=(meta["image_neg_idx"], meta["text_neg_idx"]), text_lens=hb["text_atts"].sum(1))
        sum(losses[k] for k in ("loss_itc", "loss_itm", "loss_mlm", "loss_mim")).backward()
        print({k: (round(float(losses[k]), 5), round(float(z[k]), 5)) for k in ("loss_itc", "loss_itm", "loss_mlm", "loss_mim")})
    torch.cuda.synchronize()
    params = dict(m.named_parameters())
    rows = []
    all_rms = sorted((float(z[k[:-6] + "/sq"]) / int(z[k[:-6] + "/n"])) ** 0.5 for k in z.files if k.startswith("grad/") and k.endswith("/probe"))
    min_rms = max(1e-6, 1e-2 * all_rms[len(all_rms) // 2])
    for key in z.files:
        if not (key.startswith("grad/") and key.endswith("/probe")):
            continue
        n = key[5:-6]
        if f"floor/{n}" not in z.files or "self.key.bias" in n:
            continue
        rms = (float(z[f"grad/{n}/sq"]) / int(z[f"grad/{n}/n"])) ** 0.5
        if rms < min_rms or int((z[key] != 0).sum()) < 8:
            continue
        err, cos = rel_l2(z, f"grad/{n}", params[n].grad)
        fl = [float(v) for v in z[f"floor/{n}"]]
        fe, fc = fl[2:4] if len(fl) >= 4 else fl[:2]
        rows.append((err / max(fe, 1e-9), err, fe, (1 - cos) / max(1 - fc, 1e-12), n))
    rows.sort(reverse=True)
    over = [r for r in rows if r[1] > 8e-2]
    print(f"{len(rows)} tensors; {len(over)} above 8e-2; worst err/floor among those: {max([r[0] for r in over], default=0):.3f}")
    for ratio, err, fe, cr, n in rows[:top]:
        print(f"  {ratio:5.2f} x floor  err {err:.4f} floor {fe:.4f}  (1-cos) ratio {cr:5.2f}  {n}")
    import statistics
    print("median err/floor:", round(statistics.median(r[0] for r in rows), 3))


if __name__ == "__main__":
    main()
