"""Soak: N optimisation steps of the bench workload; prints the losses, the allocator's peak / current memory and the step time every 50
steps -- catches slow leaks (kept-alive tensors), NaNs and throughput drift that a 20-step bench cannot."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from xfm_amd import synthetic as syn
from xfm_amd.accelerators import RCCLDDPAccelerator
from xfm_amd.pretrain_loop import AttrDict, create_scheduler

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
device = torch.device("cuda", 0)
model = bench.build_model(device)
optimizer = bench.make_optimizer(model)
sched = create_scheduler(AttrDict(sched="linear", num_warmup_steps=0.1, num_training_steps=steps), optimizer)
acc = RCCLDDPAccelerator({"RNG_SEED": 42, "CLIP_GRAD_NORM": 1.0, "GRAD_ACCUMULATE_STEPS": 1})
wrapped, optimizer, _ = acc.set_up(model, optimizer, None, 0, 1, 0)
model.train(True)
host = [syn.pretrain_batch(64, seed=100 + i) for i in range(4)]
batches = [{k: v.to(device) for k, v in hb.items()} for hb in host]
lens = [hb["text_atts"].sum(1) for hb in host]   # host-side caption lengths: the text / fusion towers run on unpadded rows
t0 = time.perf_counter()
for s in range(steps):
    b = batches[s % 4]
    losses = wrapped(b["image"], b["text_ids"], b["text_atts"], text_ids_masked=b["text_ids_masked"], masked_pos=b["masked_pos"],
                     masked_ids=b["masked_ids"], ret_mim_loss=True, data_source="image", text_lens=lens[s % 4])
    total = losses["loss_itc"] + losses["loss_itm"] + losses["loss_mlm"] + losses["loss_mim"]
    acc.backward_step(total, optimizer)
    acc.optimizer_step(optimizer, model)
    sched.step()
    if (s + 1) % 50 == 0:
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
        vals = [round(float(losses[k]), 4) for k in ("loss_itc", "loss_itm", "loss_mlm", "loss_mim")]
        assert all(v == v and abs(v) < 1e4 for v in vals), vals
        print(f"step {s + 1}: {dt * 1e3:.1f} ms/step  losses {vals}  grad-norm {float(acc.last_grad_norm):.3f}  "
              f"mem {torch.cuda.memory_allocated() / 2 ** 30:.2f} GiB (peak {torch.cuda.max_memory_allocated() / 2 ** 30:.2f})", flush=True)
        t0 = time.perf_counter()
print("SOAK OK")
