"""Host enqueue time vs GPU time of the stand-alone fusion encoder fwd+bwd (is the probe launch-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from xfm_amd import synthetic as syn
from xfm_amd.packing import Pack
from xfm_amd.xfm import _ones_mask

dev = torch.device("cuda", 0)
model = bench.build_model(dev)
model.finalize()
model.train(True)
B = 64
hb = syn.pretrain_batch(B, seed=1234)
T = 30
lens_h = hb["text_atts"].sum(1)
g = torch.Generator(device="cpu").manual_seed(7)
perm = torch.randperm(B, generator=g)
img = (torch.randn(B, 197, 768, generator=g) * 0.7).to(dev, torch.bfloat16).requires_grad_(True)
iatts = _ones_mask(img)
ar = torch.arange(B, device=dev)
index = torch.cat([ar, torch.randperm(B, generator=g).to(dev), ar, ar]).to(torch.int32)
lh = lens_h.tolist()
from xfm_amd.packing import image_major_layout
seq_img = index.tolist()
pack, _, _, meta, ranges = image_major_layout(lh + lh + [lh[int(j)] for j in perm] + lh, seq_img, B, T, dev, extra=(seq_img,))
index = meta[1].contiguous()
if os.environ.get("NO_RANGES"):
    ranges = None
text = (torch.randn(pack.cap, 768, generator=g) * 0.7).to(dev, torch.bfloat16).requires_grad_(True)


def once():
    seq = model.fusion_encoder.bert(encoder_hidden_states=img, encoder_attention_mask=iatts, return_dict=True, encoder_batch_index=index,
                                    encoder_embeds=text, attention_mask=None, pack=pack, encoder_row_ranges=ranges).last_hidden_state
    seq.float().square().mean().backward()


for _ in range(3):
    once()
torch.cuda.synchronize()
for iters in (1, 4, 8, 64):
    t0 = time.perf_counter()
    for _ in range(iters):
        once()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"iters {iters}: host enqueue {1e3 * (t1 - t0) / iters:.2f} ms/iter, until GPU done {1e3 * (t2 - t0) / iters:.2f} ms/iter (rows {pack.cap})", flush=True)
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(4):
    once()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
