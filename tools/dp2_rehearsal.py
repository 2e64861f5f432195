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


fixed_mask = syn.mim_block_mask(B, 14, 75, seed=77 + rank)
fixed_mask = (fixed_mask if torch.is_tensor(fixed_mask) else torch.from_numpy(fixed_mask)).to(dev)
fixed_neg = ([(i + 1) % B for i in range(B)], [(i + 3) % B for i in range(B)])


def fwd_bwd(fixed):
    """fixed: eval mode (no dropout / drop-path) with given MIM masks and hard negatives -> two calls compute the same gradient."""
    model.train(not fixed)
    kw = dict(ids_mask=fixed_mask, neg_idx=fixed_neg) if fixed else {}
    losses = wrapped(batch["image"], batch["text_ids"], batch["text_atts"], text_ids_masked=batch["text_ids_masked"],
                     masked_pos=batch["masked_pos"], masked_ids=batch["masked_ids"], ret_mim_loss=True, data_source="image", **kw)
    total = losses["loss_itc"] + losses["loss_itm"] + losses["loss_mlm"] + losses["loss_mim"]
    acc.backward_step(total, opt)
    torch.cuda.synchronize()
    return losses


for step in range(4):
    err = None   # (checked on steps 0 and 1 only: the later steps run in training mode, where the hand-averaged reference would draw other masks)
    if step < 2:  # step 0: plain sweep after backward (live ranges unknown yet); step 1: the overlapped, chunked exchange
        # (a) what the exchange must produce: every rank's LOCAL gradient (accelerator told it is alone), averaged by hand
        acc.world_size, acc._dist = 1, False
        fwd_bwd(True)
        expect = model._arena.grad.clone()
        acc.world_size, acc._dist = world, True
        dist.all_reduce(expect)
        expect /= world
        model._arena.zero_grad()
        # (b) the same step, data parallel (tower hooks, chunked ViT exchange, final sweep)
        losses = fwd_bwd(True)
        err = float((model._arena.grad - expect).norm() / expect.norm())
        assert err < 1e-3, f"rank {rank}: exchanged gradient is off the mean of the local gradients by {err:.3e} (rel. L2)"
    else:
        losses = fwd_bwd(False)  # training mode: dropout, drop-path, sampled masks and negatives
    same_everywhere(model._arena.grad, f"step {step}: exchanged gradients")
    assert bool(torch.isfinite(model._arena.grad).all()) and float(model._arena.grad.abs().max()) > 0
    acc.optimizer_step(opt, model)
    torch.cuda.synchronize()
    same_everywhere(model._arena.data, f"step {step}: parameters after the optimizer step")
    live = sum(b - a for a, b in acc._ranges)
    print(f"rank {rank} step {step}: losses {[round(float(v), 4) for v in losses.values()]} live {live}/{model._arena.numel} "
          f"grad-norm {float(acc.last_grad_norm):.4f} exchange-vs-mean-of-locals {'n/a (training-mode step: replica identity only)' if err is None else format(err, '.2e')} "
          f"ranges sent from inside backward {[(b - a) // 2 ** 20 for a, b in acc.overlapped_ranges]} Mi-elements", flush=True)
dist.barrier()
if rank == 0:
    print("DP2 REHEARSAL OK: exchanged gradients = mean of the ranks' local gradients; gradients and parameters bit-identical "
          "across ranks for 4 steps", flush=True)
dist.destroy_process_group()
