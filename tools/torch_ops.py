"""Which torch-side (non-library) ops run in one training step of the bench model: torch.profiler CPU-op table."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from xfm_amd import synthetic as syn
from xfm_amd.accelerators import RCCLDDPAccelerator

device = torch.device("cuda", 0)
model = bench.build_model(device)
opt = bench.make_optimizer(model)
acc = RCCLDDPAccelerator({"RNG_SEED": 42, "CLIP_GRAD_NORM": 1.0, "GRAD_ACCUMULATE_STEPS": 1})
wrapped, opt, _ = acc.set_up(model, opt, None, 0, 1, 0)
model.train(True)
batch = {k: v.to(device) for k, v in syn.pretrain_batch(64, seed=1234).items()}


def step():
    losses = wrapped(batch["image"], batch["text_ids"], batch["text_atts"], text_ids_masked=batch["text_ids_masked"],
                     masked_pos=batch["masked_pos"], masked_ids=batch["masked_ids"], ret_mim_loss=True, data_source="image")
    total = losses["loss_itc"] + losses["loss_itm"] + losses["loss_mlm"] + losses["loss_mim"]
    acc.backward_step(total, opt)
    acc.optimizer_step(opt, model)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=False).table(sort_by="cuda_time_total", row_limit=400, max_name_column_width=50, max_shapes_column_width=70))
