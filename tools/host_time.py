"""Host-side enqueue time per step vs the GPU's step time: is the Python / launch path close to becoming the bottleneck?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from xfm_amd import synthetic as syn
from xfm_amd.accelerators import RCCLDDPAccelerator

device = torch.device("cuda", 0)
model = bench.build_model(device)
optimizer = bench.make_optimizer(model)
acc = RCCLDDPAccelerator({"RNG_SEED": 42, "CLIP_GRAD_NORM": 1.0, "GRAD_ACCUMULATE_STEPS": 1})
wrapped, optimizer, _ = acc.set_up(model, optimizer, None, 0, 1, 0)
model.train(True)
hostb = syn.pretrain_batch(64, seed=1234)
batch = {k: v.to(device) for k, v in hostb.items()}
lens = hostb["text_atts"].sum(1)   # packed token rows, as bench.py runs the step (XFM_PACK_SYNC=0: no host sync inside the step)


def step():
    losses = wrapped(batch["image"], batch["text_ids"], batch["text_atts"], text_ids_masked=batch["text_ids_masked"],
                     masked_pos=batch["masked_pos"], masked_ids=batch["masked_ids"], ret_mim_loss=True, data_source="image", text_lens=lens)
    total = losses["loss_itc"] + losses["loss_itm"] + losses["loss_mlm"] + losses["loss_mim"]
    acc.backward_step(total, optimizer)
    acc.optimizer_step(optimizer, model)


for _ in range(5):
    step()
torch.cuda.synchronize()
# (a) host alone: synchronise before every step, so the host never waits for a queue slot
host = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
# (b) free running
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 20
print(f"host enqueue per step: min {min(host) * 1e3:.1f} ms, median {sorted(host)[5] * 1e3:.1f} ms; free-running step {wall * 1e3:.1f} ms")
