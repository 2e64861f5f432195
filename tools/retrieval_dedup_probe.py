"""ITC / ITM of the retrieval fixtures with and without image-state dedup in get_matching_loss (XFM_DEDUP_IMAGES)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from golden_util import load, state_from_spec
from xfm_amd import synthetic as syn
import xfm_amd.xfm as X
from xfm_amd.model_retrieval import XFMForRetrieval

for name in ("retrieval_small", "retrieval_384"):
    z, meta = load(name)
    cfg = {"use_beit_v2": True, "image_res": meta.get("image_res", 224), "patch_size": 16, "local_attn_depth": -1, "text_encoder": "roberta-base",
           "text_num_hidden_layers": meta["text_layers"], "text_fusion_start_at": meta["text_layers"],
           "fusion_num_hidden_layers": meta["fusion_layers"], "fusion_fusion_start_at": 0, "embed_dim": 256, "temp": 0.07,
           "learnable_temp": True, "max_temp": 0.5, "min_temp": 0.001, "vision_depth": meta.get("vit_depth", 12)}
    for dedup in (True, False):
        X._DEDUP_IMAGES = dedup
        m = XFMForRetrieval(cfg)
        m.load_state_dict(state_from_spec(meta["spec"]), strict=True)
        m.cuda().finalize().eval()
        b = {k: v.cuda() for k, v in syn.pretrain_batch(meta["B"], seed=77, image_res=meta.get("image_res", 224), max_tokens=meta.get("max_tokens", 30)).items()}
        idx = torch.tensor(meta["idx"]).cuda()
        itc, itm = m(b["image"], b["text_ids"], b["text_atts"], idx=idx, neg_idx=(meta["image_neg_idx"], meta["text_neg_idx"]))
        print(f"{name} dedup={dedup}: itc {float(itc):.5f} (ref {float(z['loss_itc']):.5f})  itm {float(itm):.5f} (ref {float(z['loss_itm']):.5f})")
