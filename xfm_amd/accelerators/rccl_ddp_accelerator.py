"""RCCL-over-xGMI data parallelism on the flat gradient arena.

Replaces torch-DDP / apex-DDP of the reference (accelerators/ddp_accelerator.py:23-98, apex_ddp_accelerator.py:30-110)
behind the same contract: set_up -> (wrapped_model, optimizer, lr_scheduler) with `.module` the bare model;
backward_step(loss, optimizer); optimizer_step(optimizer, model) -> total grad norm (clip -> step -> zero_grad).

MI355X-first design: gradients already live in one contiguous fp32 arena laid out tower by tower, so the exchange is a
handful of LARGE in-place all-reduces over arena ranges (xGMI is point-to-point: few big collectives beat many
25 MiB buckets), issued on a side HIP stream the moment a tower's last backward node has run (use-count hooks of the
tower nodes) so they overlap the rest of backward -- in the pre-training step the fusion and text towers finish
while the two ViT backward passes (56 % of the FLOPs) are still running.  Ranges that never receive a gradient
(LM heads of the text tower, caption heads, fusion embeddings, bbox head: 158 M of 522 M elements) are
found on the first step and skipped afterwards, identically on every rank.
"""
import math
import random

import numpy as np
import torch
import torch.distributed as dist

from .. import functional as Fx
from .accelerator import Accelerator


class _Wrapped(torch.nn.Module):
    """`.module` holder so callers written against DDP (Pretrain.py:261-263) keep working."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **kw):
        return self.module(*a, **kw)


class RCCLDDPAccelerator(Accelerator):
    def __init__(self, cfg, logger=None):
        super().__init__(cfg, logger)
        g = cfg.get if isinstance(cfg, dict) else lambda k, d=None: getattr(cfg, k, d)
        self.seed = g("RNG_SEED", 42)
        self.clip = g("CLIP_GRAD_NORM", 0.0) or 0.0
        self.accum = max(int(g("GRAD_ACCUMULATE_STEPS", 1) or 1), 1)
        self.fused_optimizer = g("FUSED_ADAMW", True)
        self.world_size, self.rank = 1, 0
        self.model = None
        self._pending = []
        self._comm_stream = None
        self._live = None
        self._step = 0
        self._opt_state = None
        self._sync_now = False
        self._done_ranges = []
        self._use = {}
        self._op = dist.ReduceOp.SUM

    # ------------------------------------------------------------------------------------------ set-up
    def set_seed(self):
        random.seed(self.seed)
        np.random.seed(self.seed)
        torch.manual_seed(self.seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(self.seed)

    def set_up(self, model, optimizer, lr_scheduler, local_rank, world_size, rank):
        self.set_seed()
        self.world_size, self.rank = world_size, rank
        use_cuda = torch.cuda.is_available()
        if use_cuda:
            torch.cuda.set_device(local_rank)
            model = model.cuda()
        if world_size > 1 and not dist.is_initialized():
            dist.init_process_group(backend="nccl" if use_cuda else "gloo", world_size=world_size, rank=rank)
        # RCCL averages in the collective itself (no extra pass over the 1.4 GB of live gradients)
        self._op = dist.ReduceOp.AVG if (world_size > 1 and dist.get_backend() == "nccl") else dist.ReduceOp.SUM
        if hasattr(model, "finalize"):
            model.finalize()
        self.model = model
        arena = getattr(model, "_arena", None)
        self.arena = arena
        if world_size > 1:
            self.broadcast()
        if arena is not None and use_cuda:
            self._comm_stream = torch.cuda.Stream()
            self._install_tower_hooks(model)
        return _Wrapped(model), optimizer, lr_scheduler

    def broadcast(self):
        """rank 0's weights to everyone: one collective over the parameter arena instead of 771 (ddp_accelerator.py:73-74)."""
        if self.arena is not None:
            dist.broadcast(self.arena.data, 0)
            self.arena.bump()
        else:
            for v in self.model.state_dict().values():
                dist.broadcast(v, 0)

    # ------------------------------------------------------------------------------------------ overlap
    def _tower_ranges(self, model):
        towers = []
        for name in ("fusion_encoder", "text_encoder", "vision_encoder"):
            mod = getattr(model, name, None)
            if mod is not None:
                ps = list(mod.parameters())
                towers.append((name, mod, self.arena.range_of(ps)))
        return towers

    def _install_tower_hooks(self, model):
        self._towers = self._tower_ranges(model)
        self._use = {}
        for name, mod, rng in self._towers:
            if not hasattr(mod, "roberta"):
                # the vision tower finishes last and its patch-embed / cls gradients trail the trunk (swept after backward), but
                # the trunk's blocks can leave in chunks from inside its backward (beit2._TrunkFn.backward)
                if hasattr(mod, "blocks"):
                    self._use[id(mod)] = [0, name, None]
                    mod._use_hook = self._on_use
                    mod._block_grad_hook = self._on_blocks_done
                continue
            nodes = [mod.roberta]
            for node in nodes:
                self._use[id(node)] = [0, name, rng]
                node._use_hook = self._on_use

    def _on_use(self, node, delta):
        rec = self._use[id(node)]
        rec[0] += delta
        # overlap starts on step 2: step 1 first has to find out which arena ranges ever receive a gradient
        if delta < 0 and rec[0] == 0 and rec[2] is not None and self.world_size > 1 and self._sync_now and self._live is not None:
            self._launch(rec[2])

    def _on_blocks_done(self, vit, lo, hi, wgrad_stream):
        """Blocks [lo, hi) of the vision trunk have their final gradients (called from the trunk's backward).  Only when this
        backward is the tower's last pending use of the step (two ViT passes accumulate into the same range).  Returns whether the
        range was handed to the all-reduce."""
        rec = self._use[id(vit)]
        if not (rec[0] == 1 and self.world_size > 1 and self._sync_now and self._live is not None):
            return False
        ps = [p for b in vit.blocks[lo:hi] for p in b.parameters()]
        self._launch(self.arena.range_of(ps), extra_stream=wgrad_stream)
        return True

    def _live_chunks(self, lo, hi):
        """Sub-ranges of [lo, hi) that received a gradient on the first step (static afterwards)."""
        if self._live is None:
            return [(lo, hi)]
        return [(max(a, lo), min(b, hi)) for a, b in self._live if max(a, lo) < min(b, hi)]

    def _launch(self, rng, extra_stream=None):
        cur = torch.cuda.current_stream()
        self._comm_stream.wait_stream(cur)
        if extra_stream is not None:  # weight-gradient GEMMs of the range still in flight on their own stream
            self._comm_stream.wait_stream(extra_stream)
        with torch.cuda.stream(self._comm_stream):
            for a, b in self._live_chunks(*rng):
                self._pending.append(dist.all_reduce(self.arena.grad[a:b], op=self._op, async_op=True))
        self._done_ranges.append(rng)

    # ------------------------------------------------------------------------------------------ step
    def backward_step(self, loss, optimizer):
        self._step += 1
        self._sync_now = (self._step % self.accum) == 0
        self._done_ranges = []
        loss.backward()
        if self.world_size > 1 and self._sync_now:
            self._finish_allreduce()

    def _finish_allreduce(self):
        arena = self.arena
        if arena is None:
            for p in self.model.parameters():
                if p.grad is not None:
                    dist.all_reduce(p.grad)
                    p.grad.div_(self.world_size)
            return
        # (which ranges ever receive a gradient is only decided at the first optimizer step, after EVERY source of a multi-source
        # step has run its backward -- Pretrain.py:211-239; until then the whole arena is exchanged)
        done = sorted(self._done_ranges)
        self.overlapped_ranges = list(done)  # (for logs / tests) what left for the all-reduce from inside backward this step
        todo, pos = [], 0
        for a, b in done + [(arena.numel, arena.numel)]:
            assert a >= pos or a == b, f"arena range ({a}, {b}) was handed to the all-reduce twice in one step"
            if a > pos:
                todo.append((pos, a))
            pos = max(pos, b)
        if self._comm_stream is not None:
            self._comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm_stream):
                for lo, hi in todo:
                    for a, b in self._live_chunks(lo, hi):
                        self._pending.append(dist.all_reduce(arena.grad[a:b], op=self._op, async_op=True))
            for w in self._pending:
                w.wait()
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        else:
            for lo, hi in todo:
                for a, b in self._live_chunks(lo, hi):
                    dist.all_reduce(arena.grad[a:b], op=self._op)
        self._pending = []
        for rec in self._use.values():
            rec[0] = 0
        if self._op == dist.ReduceOp.SUM:  # gloo has no AVG: scale the exchanged chunks
            for a, b in (self._live if self._live is not None else [(0, arena.numel)]):
                arena.grad[a:b].mul_(1.0 / self.world_size)

    def _discover_live(self):
        """1 KiB-granular map of arena blocks that got a gradient; agreed across ranks with one MAX all-reduce."""
        g = self.arena.grad.view(-1, 256)
        live = (g != 0).any(dim=1).to(torch.int32)
        if self.world_size > 1:
            dist.all_reduce(live, op=dist.ReduceOp.MAX)
        live = live.cpu().numpy().astype(bool)
        # merge into chunks, bridging dead gaps shorter than 1 MiB so the collectives stay few and large
        idx = np.flatnonzero(live)
        chunks = []
        if idx.size:
            start = prev = int(idx[0])
            for i in idx[1:]:
                i = int(i)
                if i - prev > 1024:
                    chunks.append((start * 256, (prev + 1) * 256))
                    start = i
                prev = i
            chunks.append((start * 256, (prev + 1) * 256))
        # never let a chunk straddle a tower boundary (they are launched per tower)
        cuts = sorted({r[0] for _, _, r in getattr(self, "_towers", [])} | {r[1] for _, _, r in getattr(self, "_towers", [])})
        out = []
        for a, b in chunks:
            pts = [a] + [c for c in cuts if a < c < b] + [b]
            out.extend(zip(pts[:-1], pts[1:]))
        self._live = out

    def reset_live(self):
        """Forget which arena ranges receive gradients (call when the loss mix GROWS mid-run, e.g. a head that was unused so far
        starts to train: the map is decided once, at the first optimizer step, and the reference's configs only ever turn losses
        off).  The whole gradient arena is zeroed and exchanged until the next optimizer step re-decides."""
        self._live = None
        if self.arena is not None:
            self.arena.zero_grad()

    def _grad_norm_sq(self):
        """Over the live ranges only: blocks that never receive a gradient are zero (and stay zero), no need to read them."""
        out = torch.zeros(1, dtype=torch.float32, device=self.arena.grad.device)
        for a, b in (self._live if self._live is not None else [(0, self.arena.numel)]):
            Fx.sumsq(self.arena.grad[a:b], out)
        return out

    def optimizer_step(self, optimizer, model, grad_norm: float = 0.0):
        """clip -> step -> zero_grad, returns the total gradient norm (apex_ddp_accelerator.py:100-110)."""
        if self._step % self.accum != 0:
            return 0.0
        arena = self.arena
        if arena is None or not arena.grad.is_cuda:
            if arena is not None and self._live is None:
                self._discover_live()  # later exchanges skip the ranges that never get a gradient
            total = torch.nn.utils.clip_grad_norm_(model.parameters(), self.clip if self.clip > 0 else float("inf"))
            optimizer.step()
            if arena is not None:
                arena.bump()
                arena.zero_grad()  # keeps .grad attached to the arena (optimizer.zero_grad() would drop the views)
            else:
                optimizer.zero_grad()
            return float(total)
        if self._live is None:
            self._discover_live()  # world_size 1: the first step still has to learn which ranges ever get a gradient
        norm = self._grad_norm_sq().sqrt()
        clip_coef = None
        if self.clip > 0:
            clip_coef = (self.clip / (norm + 1e-6)).clamp(max=1.0)
        if self.fused_optimizer and _is_adamw(optimizer):
            self._fused_adamw(optimizer, clip_coef)
        else:
            if clip_coef is not None:
                arena.grad.mul_(clip_coef)
            optimizer.step()
        arena.bump()
        arena.zero_grad(self._live)
        self.last_grad_norm = norm  # device tensor: no host sync on the step path
        return norm

    # ------------------------------------------------------------------------------------------ fused AdamW
    def _fused_adamw(self, optimizer, clip_coef):
        arena = self.arena
        st = self._opt_state
        if st is None:
            groups = optimizer.param_groups
            assert len(groups) <= 4, "fused AdamW supports up to 4 parameter groups (optim.py:4-50 builds 4)"
            gid = torch.zeros(arena.numel // 256, dtype=torch.uint8)
            for gi, grp in enumerate(groups):
                for p in grp["params"]:
                    o, n = arena.offsets[id(p)]
                    gid[o // 256:(o + n + 255) // 256] = gi
            st = self._opt_state = {"m": torch.zeros_like(arena.data), "v": torch.zeros_like(arena.data),
                                    "group": gid.to(arena.data.device), "t": 0}
        st["t"] += 1
        groups = optimizer.param_groups
        b1, b2 = groups[0]["betas"]
        # parameters that never receive a gradient are skipped entirely, like `if p.grad is None: continue` in the reference's
        # AdamW (no weight decay on the unused LM / caption / bbox heads), and it saves 30 % of the arena traffic
        for a, b in (self._live if self._live is not None else [(0, arena.numel)]):
            Fx.adamw(arena.data[a:b], arena.grad[a:b], st["m"][a:b], st["v"][a:b], st["group"][a // 256:(b + 255) // 256],
                     [g["lr"] for g in groups], [g.get("weight_decay", 0.0) for g in groups], b1, b2, groups[0]["eps"], st["t"],
                     clip_coef)

    def state_dict(self):
        st = self._opt_state
        return {} if st is None else {"m": st["m"], "v": st["v"], "t": st["t"]}

    def load_state_dict(self, sd):
        if sd and self._opt_state is not None:
            self._opt_state["m"].copy_(sd["m"])
            self._opt_state["v"].copy_(sd["v"])
            self._opt_state["t"] = sd["t"]


def _is_adamw(opt):
    return type(opt).__name__ in ("AdamW",)
