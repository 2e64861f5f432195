"""N > 1 rehearsal with real kernels on a ONE-GPU box: every rank on cuda:0, gloo instead of RCCL (RCCL refuses two ranks on one
device).  Exercises what the world-size-2 CPU tests cannot: the GPU branch of RCCLDDPAccelerator (tower hooks that launch the
arena all-reduce from inside backward on the side stream, live-range discovery, fused clip + AdamW on the flat arena, live-range
zeroing), the ITC all-gather with slice-only backward, the second-stream weight gradients and the text-tower stream -- all with
per-rank batches.  Checks after every step that all ranks hold bit-identical gradients and parameters.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 tools/dp2_rehearsal.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import bench
from xfm_amd import synthetic as syn
from xfm_amd.accelerators import RCCLDDPAccelerator
from xfm_amd.model_pretrain import XFM

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo", world_size=world, rank=rank)
cfg = dict(bench.FULL_CFG, text_num_hidden_layers=2, text_fusion_start_at=2, fusion_num_hidden_layers=2)
torch.manual_seed(7 + rank)  # different initial weights per rank: set_up must broadcast rank 0's
model = XFM(cfg).to(dev)
opt = bench.make_optimizer(model)
acc = RCCLDDPAccelerator({"RNG_SEED": 42 + rank, "CLIP_GRAD_NORM": 1.0, "GRAD_ACCUMULATE_STEPS": 1})
wrapped, opt, _ = acc.set_up(model, opt, None, 0, world, rank)
model.train()
B = 8
batch = {k: v.to(dev) for k, v in syn.pretrain_batch(B, seed=1234 + rank).items()}


def same_everywhere(t, what):
    ref = t.detach().clone()
    dist.broadcast(ref, 0)
    assert torch.equal(ref, t), f"rank {rank}: {what} differs from rank 0 (max |d| = {float((ref - t).abs().max()):.3e})"


same_everywhere(model._arena.data, "parameters after set_up")
for step in range(3):
    losses = wrapped(batch["image"], batch["text_ids"], batch["text_atts"], text_ids_masked=batch["text_ids_masked"],
                     masked_pos=batch["masked_pos"], masked_ids=batch["masked_ids"], ret_mim_loss=True, data_source="image")
    total = losses["loss_itc"] + losses["loss_itm"] + losses["loss_mlm"] + losses["loss_mim"]
    acc.backward_step(total, opt)
    torch.cuda.synchronize()
    same_everywhere(model._arena.grad, f"step {step}: exchanged gradients")
    assert bool(torch.isfinite(model._arena.grad).all()) and float(model._arena.grad.abs().max()) > 0
    acc.optimizer_step(opt, model)
    torch.cuda.synchronize()
    same_everywhere(model._arena.data, f"step {step}: parameters after the optimizer step")
    live = sum(b - a for a, b in acc._live)
    print(f"rank {rank} step {step}: losses {[round(float(v), 4) for v in losses.values()]} live {live}/{model._arena.numel} "
          f"grad-norm {float(acc.last_grad_norm):.4f}", flush=True)
dist.barrier()
if rank == 0:
    print("DP2 REHEARSAL OK: gradients and parameters bit-identical across ranks for 3 steps", flush=True)
dist.destroy_process_group()
