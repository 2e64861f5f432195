"""Which torch (aten) ops still run in one pre-training step, by GPU time and input shapes: the glue around the HIP library
(copies, fills, cats, index_selects).  python tools/torch_ops.py > gpurun_out/torch_ops.log"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("XFM_WGRAD_STREAM", "0")
os.environ.setdefault("XFM_TEXT_STREAM", "0")
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from xfm_amd import synthetic as syn
from xfm_amd.accelerators import RCCLDDPAccelerator

device = torch.device("cuda", 0)
model = bench.build_model(device)
optimizer = bench.make_optimizer(model)
acc = RCCLDDPAccelerator({"RNG_SEED": 42, "CLIP_GRAD_NORM": 1.0, "GRAD_ACCUMULATE_STEPS": 1})
wrapped, optimizer, _ = acc.set_up(model, optimizer, None, 0, 1, 0)
model.train(True)
host = syn.pretrain_batch(64, seed=1234)
batch = {k: v.to(device) for k, v in host.items()}
lens = host["text_atts"].sum(1)   # packed token rows, as bench.py runs the step


def step():
    losses = wrapped(batch["image"], batch["text_ids"], batch["text_atts"], text_ids_masked=batch["text_ids_masked"],
                     masked_pos=batch["masked_pos"], masked_ids=batch["masked_ids"], ret_mim_loss=True, data_source="image", text_lens=lens)
    total = losses["loss_itc"] + losses["loss_itm"] + losses["loss_mlm"] + losses["loss_mim"]
    acc.backward_step(total, optimizer)
    acc.optimizer_step(optimizer, model)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
def dev_us(e):
    return getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0.0))


rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and dev_us(e) > 0]
rows.sort(key=lambda e: -dev_us(e))
print("== aten ops with GPU time, by input shape")
for e in rows[:40]:
    print(f"{e.key:28s} calls={e.count:4d} gpu={dev_us(e):9.1f}us  {str(e.input_shapes)[:110]}")
rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.key.startswith("aten::") and dev_us(e) > 0]
rows.sort(key=lambda e: -dev_us(e))
print("== the same by call site")
for e in rows[:70]:
    stack = [f for f in e.stack if "xfm_amd" in f or "bench.py" in f][:2]
    print(f"{e.key:22s} calls={e.count:4d} gpu={dev_us(e):9.1f}us  {' <- '.join(x.strip()[-70:] for x in stack)}")
print("== device activities named like copies / fills")
for e in prof.key_averages():
    n = e.key.lower()
    if ("memcpy" in n or "memset" in n or "rocclr" in n or "fill" in n) and dev_us(e) > 0:
        print(f"{e.key[:90]:90s} calls={e.count:4d} gpu={dev_us(e):9.1f}us")
print("== host <-> device traffic and scalar materialisations (every one is a blit kernel or a sync on the stream)")
for e in sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.count):
    if e.key in ("aten::_to_copy", "aten::scalar_tensor", "aten::_local_scalar_dense", "aten::item", "aten::lift_fresh", "aten::tensor",
                 "aten::full", "aten::arange", "aten::zeros", "aten::ones", "aten::empty_strided") or "Memcpy" in e.key:
        print(f"{e.key:32s} calls={e.count:4d} cpu={e.cpu_time_total:9.1f}us gpu={dev_us(e):8.1f}us {str(e.input_shapes)[:90]}")
