"""RCCL-over-xGMI data parallelism on the flat gradient arena.

Replaces torch-DDP / apex-DDP of the reference (accelerators/ddp_accelerator.py:23-98, apex_ddp_accelerator.py:30-110)
behind the same contract: set_up -> (wrapped_model, optimizer, lr_scheduler) with `.module` the bare model;
backward_step(loss, optimizer); optimizer_step(optimizer, model) -> total grad norm (clip -> step -> zero_grad).

MI355X-first design: gradients already live in one contiguous fp32 arena laid out tower by tower, so the exchange is a
handful of LARGE in-place all-reduces over arena ranges (xGMI is point-to-point: few big collectives beat many
25 MiB buckets), issued on a side HIP stream the moment a tower's last backward node has run (use-count hooks of the
tower nodes) so they overlap the rest of backward -- in the pre-training step the fusion and text towers finish
while the two ViT backward passes (56 % of the FLOPs) are still running.

Which parameters take part is decided per WHOLE PARAMETER and structurally: a parameter is live from the first backward
whose launch sites wrote a gradient for it (ParamArena.touch, called by grad_of() / slot.dw / slot.db and by an autograd hook)
and stays live for the rest of the run -- exactly the parameters whose `.grad` is not None in the reference, where
`optimizer.zero_grad()` zeroes in place (torch 1.x) and AdamW skips `grad is None` (optim.py:4-50).  Live parameters are
exchanged, clipped, weight-decayed, stepped and zeroed every step, row-sparse embedding gradients included; parameters that
never get a gradient (LM heads of the text tower, caption heads, fusion embeddings, bbox head: 158 M of 522 M elements,
SURVEY 2.2) are skipped entirely.  The set only grows (a text-only first step followed by an image step simply adds the
other towers), and ranks agree on it with one MAX all-reduce of the flag vector whenever it grew.
"""
import os
import random

import numpy as np
import torch
import torch.distributed as dist

from .. import functional as Fx
from .accelerator import Accelerator


_DRY = os.environ.get("XFM_DDP_DRY", "0") == "1"
# first-use order of the step's streams at set_up when collectives are on (see _touch_streams; measured in an RCCL group of one,
# profiles/round5_forced_collectives.md: the text tower's backward starts at 19.6 ms instead of 30.3 ms, step -0.4 ms)
_STREAM_TOUCH_DIST = "nccl,text,wmain,wtext,cast,zero,comm"
_HOST_TIMES = [] if os.environ.get("XFM_DDP_HOST_TIMES", "0") == "1" else None   # diagnostic: host milliseconds per all_reduce call
# The step's optimizer.zero_grad() (1.46 GB of fills over the live gradient ranges, ~0.3 ms at HBM speed) on a stream of its own,
# behind the AdamW kernels that read the gradients: nothing reads or writes a gradient again before the NEXT backward pass, so the
# fills run under the next step's forward instead of at the end of the serial optimizer tail; backward_step (and every accelerator
# entry that touches the gradient arena) waits for them.  A/B knob.
# XFM_DP_NATIVE=1 (opt-in): the gradient ranges leave through the library's own communicator (include/xfm_hip.h xfm_dp_bucket_allreduce,
# xfm_amd/dp.py) instead of ProcessGroupNCCL -- the collective then runs ON the communication stream it is launched from (torch's
# runs on ProcessGroupNCCL's internal stream and hands back an event).  torch.distributed still carries the bootstrap (the unique id), the
# live-set agreement, the set-up broadcast and the ITC all_gather.
_NATIVE = os.environ.get("XFM_DP_NATIVE", "0") != "0"
_ASYNC_ZERO = os.environ.get("XFM_ASYNC_ZERO", "1") != "0"
# The squared gradient norm of a tower's arena ranges as soon as the tower's backward (and, at N > 1, its all-reduce) is over: the text
# and fusion towers (70 % of the live gradient bytes) are final while the ViT backward still runs, so their share of the clip norm's
# read pass over the gradients leaves the serial optimizer tail.  Same kernel, same ranges, same order of the final sum: the norm is
# bit-identical to the one computed after backward.  A/B knob.
_EARLY_NORM = os.environ.get("XFM_EARLY_NORM", "1") != "0"


class _Wrapped(torch.nn.Module):
    """`.module` holder so callers written against DDP (Pretrain.py:261-263) keep working."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **kw):
        return self.module(*a, **kw)


# ---- interval helpers (sorted, disjoint (start, end) lists) ----------------------------------------------------------
def _clip(ranges, lo, hi):
    return [(max(a, lo), min(b, hi)) for a, b in ranges if max(a, lo) < min(b, hi)]


def _subtract(ranges, holes):
    """ranges minus holes."""
    out = []
    for a, b in ranges:
        pos = a
        for c, d in holes:
            if d <= pos or c >= b:
                continue
            if c > pos:
                out.append((pos, c))
            pos = max(pos, d)
        if pos < b:
            out.append((pos, b))
    return out


class RCCLDDPAccelerator(Accelerator):
    takes_sync_hint = True  # backward_step(loss, optimizer, sync=...) -- see there

    def __init__(self, cfg, logger=None):
        super().__init__(cfg, logger)
        g = cfg.get if isinstance(cfg, dict) else lambda k, d=None: getattr(cfg, k, d)
        self.seed = g("RNG_SEED", 42)
        self.clip = g("CLIP_GRAD_NORM", 0.0) or 0.0
        self.accum = max(int(g("GRAD_ACCUMULATE_STEPS", 1) or 1), 1)
        self.fused_optimizer = g("FUSED_ADAMW", True)
        # wire format of the gradient exchange: 'fp32' (default; the reference's DDP reduces fp32 gradients) or 'bf16' (half the
        # xGMI bytes: chunks are packed to bf16, averaged by RCCL and unpacked on the communication stream)
        self.exchange_dtype = str(g("GRAD_EXCHANGE_DTYPE", os.environ.get("XFM_GRAD_EXCHANGE", "fp32"))).lower()
        assert self.exchange_dtype in ("fp32", "bf16"), self.exchange_dtype
        # FORCE_COLLECTIVES (test / bring-up knob, also XFM_DDP_FORCE=1): run every collective of the N > 1 path -- the agreement
        # all-reduce, the overlapped arena all-reduces on the communication stream, the bf16 pack / unpack, the chunked ViT hand-over
        # -- in a process group of ONE rank, so that they execute through ProcessGroupNCCL (RCCL) on a single-GPU box; the result
        # must equal the no-collective run (mean over one rank)
        self.force = bool(g("FORCE_COLLECTIVES", os.environ.get("XFM_DDP_FORCE", "0") == "1"))
        # the reference's fp16 / apex knobs (apex_ddp_accelerator.py:26-33, ddp_accelerator.py:20-27) have no effect on the bf16 path
        ignored = [k for k in ("FP16_OPT_LEVEL", "FP16_LOSS_SCALE", "AUTO_CAST", "SYNCBN") if g(k, None) not in (None, False)]
        if ignored:
            msg = "RCCLDDPAccelerator: bf16 compute with fp32 master weights; ignoring " + ", ".join(ignored)
            (logger.info if logger is not None and hasattr(logger, "info") else print)(msg)
        self.world_size, self.rank = 1, 0
        self._dist = False         # collectives on: world_size > 1, or forced in an initialised group
        self.stats = {"exchange_bytes": 0, "exchange_calls": 0, "overlapped_bytes": 0}   # of the last synchronising backward
        self.model = None
        self.arena = None
        self._pending = []
        self._comm_stream = None
        self._ranges = []          # live arena ranges, split at tower boundaries (same on every rank)
        self._agreed_ver = -1      # arena.live_ver the ranges were derived from
        self._cuts = ()
        self._towers = []
        self._overlap_ok = False
        self._step = 0
        self._m = self._v = self._group = None
        self._t = None             # optimizer steps taken per arena unit (bias correction is per parameter in torch's AdamW)
        self._hooked = set()
        self._sync_now = False
        self._done_ranges = []
        self._use = {}
        self._op = dist.ReduceOp.SUM
        self._native = None        # xfm_amd.dp.NativeComm when XFM_DP_NATIVE=1 (set_up)
        self.timing = None         # (start, end) torch.cuda.Event pair a caller installs to time the exposed part of the exchange
        self._zero_stream = None   # the optimizer step's zero_grad runs here (_zero_async)
        self._norm_early = {}      # (a, b) live range -> (sum of squares [1], event): computed from inside backward (_early_sumsq)
        self._early_norm_ok = False

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
        self._dist = world_size > 1 or (self.force and dist.is_available() and dist.is_initialized())
        # (module-level switch of xfm.allgather, set to THIS accelerator's choice at every set_up: a forced run must not leave
        # later models of the process gathering in a group of one)
        from .. import xfm as _xfm
        _xfm.FORCE_COLLECTIVES = bool(self.force and self._dist)
        # RCCL averages in the collective itself (no extra pass over the 1.4 GB of live gradients)
        self._op = dist.ReduceOp.AVG if (self._dist and dist.get_backend() == "nccl") else dist.ReduceOp.SUM
        self._native = None
        if _NATIVE and self._dist and use_cuda and dist.get_backend() == "nccl":
            from ..dp import NativeComm
            box = [NativeComm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            self._native = NativeComm(box[0], rank, dist.get_world_size())
        if hasattr(model, "finalize"):
            model.finalize()
        self.model = model
        arena = getattr(model, "_arena", None)
        self.arena = arena
        # (data parallel: first-use order of the step's streams chosen so that RCCL's stream does not share a hardware queue with the
        # text tower's stream or with the ViT's weight-gradient stream -- _touch_streams; N = 1: the step's natural order is fine)
        touch = os.environ.get("XFM_STREAM_TOUCH", _STREAM_TOUCH_DIST if self._dist else "") if (use_cuda and arena is not None) else ""
        if self._dist and "nccl" not in touch:
            self.broadcast()
        if arena is not None:
            self._towers = self._tower_ranges(model)
            self._cuts = sorted({r[0] for _, _, r in self._towers} | {r[1] for _, _, r in self._towers})
            self._t = [0] * len(arena._units)
            if use_cuda:
                # XFM_COMM_PRIO (A/B knob): HIP priority of the communication stream
                self._comm_stream = torch.cuda.Stream(priority=int(os.environ.get("XFM_COMM_PRIO", "0")))
                self._install_tower_hooks(model)
            if touch:
                self._touch_streams(touch, local_rank)
            if optimizer is not None:
                self._hook_optimizer(optimizer)
                self._adopt_optimizer_state(optimizer)  # a resumed optimizer.load_state_dict() precedes set_up (Pretrain.py:437-447)
        return _Wrapped(model), optimizer, lr_scheduler

    def _touch_streams(self, order, local_rank):
        """HIP deals a process's streams onto its few hardware queues (GPU_MAX_HW_QUEUES = 4) in the order of their FIRST USE, and
        streams that share a queue run in submission order: a kernel behind another stream's pending event wait is stuck even though
        its own stream is free (rocprofv3 Queue_Id, tools/queue_roles.py: with the natural first-use order RCCL's stream and the text
        tower's share a queue, and the text backward starts 11 ms late behind a collective that waits for a hand-over event).  Here
        every stream of the step is created and used once, in an order chosen for the sharing it produces; `nccl` stands for the
        set-up broadcast (ProcessGroupNCCL's first collective = its stream's first use)."""
        from ..model_pretrain import _side_stream
        from ..xroberta import _WgradStream
        dev = torch.device("cuda", local_rank)
        main = torch.cuda.current_stream(dev)

        def wside(of):
            key = (dev.index, of.cuda_stream)
            if key not in _WgradStream._streams:
                _WgradStream._streams[key] = torch.cuda.Stream(device=dev, priority=_WgradStream.priority)
            return _WgradStream._streams[key]

        for name in [t.strip() for t in order.split(",") if t.strip()]:
            if name == "nccl":
                if self._dist:
                    self.broadcast()
                continue
            if name == "text":
                st = _side_stream(dev)
            elif name == "wmain":
                st = wside(main)
            elif name == "wtext":
                st = wside(_side_stream(dev))
            elif name == "cast":
                if self.arena._cast_stream is None:
                    self.arena._cast_stream = torch.cuda.Stream(device=dev)
                st = self.arena._cast_stream
            elif name == "zero":
                if self._zero_stream is None:
                    self._zero_stream = torch.cuda.Stream()
                st = self._zero_stream
            elif name == "comm":
                st = self._comm_stream
            elif name == "pad":
                st = torch.cuda.Stream()
                self._pads = getattr(self, "_pads", []) + [st]
            else:
                raise ValueError(f"XFM_STREAM_TOUCH: unknown stream '{name}'")
            with torch.cuda.stream(st):
                torch.zeros(64, device=dev).add_(1.0)   # (a first kernel: the stream gets its hardware queue now)
        torch.cuda.synchronize(dev)

    def broadcast(self):
        """rank 0's weights to everyone: one collective over the parameter arena instead of 771 (ddp_accelerator.py:73-74)."""
        if self.arena is not None:
            dist.broadcast(self.arena.data, 0)
            self.arena.bump()
        else:
            for v in self.model.state_dict().values():
                dist.broadcast(v, 0)

    # ------------------------------------------------------------------------------------------ liveness
    def _agree(self):
        """Bring the live-parameter set up to date (and identical on every rank)."""
        arena = self.arena
        if self._dist:
            flags = torch.tensor(arena.live, dtype=torch.uint8).to(arena.grad.device).to(torch.int32)
            dist.all_reduce(flags, op=dist.ReduceOp.MAX)
            merged = flags.cpu().numpy().astype(bool).tolist()
            if merged != arena.live:
                arena.live = merged
                arena.live_ver += 1
        self._ranges = arena.live_ranges(self._cuts)
        self._agreed_ver = arena.live_ver

    def live_ranges(self):
        if self.arena.live_ver != self._agreed_ver:
            self._agree()
        return self._ranges

    # ------------------------------------------------------------------------------------------ overlap
    def _tower_ranges(self, model):
        towers = []
        for name in ("fusion_encoder", "text_encoder", "vision_encoder"):
            mod = getattr(model, name, None)
            if mod is not None:
                ps = list(mod.parameters())
                if ps:
                    towers.append((name, mod, self.arena.range_of(ps)))
        return towers

    def _install_tower_hooks(self, model):
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
        # overlap needs a live set every rank already agrees on (i.e. from the second step with a given loss mix on)
        if delta < 0 and rec[0] == 0 and rec[2] is not None:
            if self._overlap_ok:
                self._launch(rec[2])
            elif self._early_norm_ok and not self._dist:
                self._early_sumsq(rec[2])

    def _on_blocks_done(self, vit, lo, hi, wgrad):
        """Blocks [lo, hi) of the vision trunk have passed their backward (called from the trunk's backward; `wgrad` is its
        _WgradStream).  Only when this backward is the tower's last pending use of the step (two ViT passes accumulate into the same
        range).  The blocks' queued weight gradients are launched first (one grouped call for the chunk); the exchange waits for the
        stream they run on.  Returns whether the range was handed to the all-reduce."""
        rec = self._use[id(vit)]
        if not (rec[0] == 1 and self._overlap_ok):
            return False
        wgrad.flush()
        ps = [p for b in vit.blocks[lo:hi] for p in b.parameters()]
        self._launch(self.arena.range_of(ps), extra_stream=wgrad.side if wgrad.on else None)
        return True

    def _exchange(self, a, b, async_ok=True):
        """All-reduce (mean) of arena.grad[a:b] on the current stream."""
        g = self.arena.grad[a:b]
        if _DRY:   # measurement knob: every hook, flush and stream wait of the N > 1 path, no collective (what does the MACHINERY cost?)
            return
        self.stats["exchange_calls"] += 1
        self.stats["exchange_bytes"] += (b - a) * (2 if self.exchange_dtype == "bf16" else 4)
        if self._native is not None:   # on the current (= communication) stream; nothing to wait for afterwards
            if self.exchange_dtype == "bf16":
                buf = g.to(torch.bfloat16)
                self._native.all_reduce(buf, "avg")
                g.copy_(buf)
            else:
                self._native.all_reduce(g, "avg")
        elif self.exchange_dtype == "bf16":
            buf = g.to(torch.bfloat16)
            w = dist.all_reduce(buf, op=self._op, async_op=True)
            w.wait()  # stream-ordered for RCCL (the host does not block); the unpack follows on this stream
            g.copy_(buf)
        elif async_ok and self._comm_stream is not None:
            if _HOST_TIMES is not None:
                import time
                t0 = time.perf_counter()
                self._pending.append(dist.all_reduce(g, op=self._op, async_op=True))
                _HOST_TIMES.append((time.perf_counter() - t0) * 1e3)
            else:
                self._pending.append(dist.all_reduce(g, op=self._op, async_op=True))
        else:
            dist.all_reduce(g, op=self._op)

    def _launch(self, rng, extra_stream=None):
        cur = torch.cuda.current_stream()
        self._comm_stream.wait_stream(cur)
        if extra_stream is not None:  # weight-gradient GEMMs of the range still in flight on their own stream
            self._comm_stream.wait_stream(extra_stream)
        # Parameter gradients are also written on the weight-gradient side streams (xroberta._WgradStream: one per launch stream) by
        # launch sites that re-join only at the END of the backward pass (join_at_end: the LM head's two weight gradients, whose
        # parameters lie inside the fusion tower's range; ops._LinearSlotFn).  A range leaves only with its gradients final, so the
        # exchange waits for everything those streams hold at this point, whichever stream the writer used -- not just the one whose
        # FIFO happens to be shared with the tower (advisor finding, round 4).
        from ..xroberta import _WgradStream
        dev = self.arena.grad.device.index
        for (d, _), side in list(_WgradStream._streams.items()):
            if d == dev and side is not extra_stream:
                self._comm_stream.wait_stream(side)
        with torch.cuda.stream(self._comm_stream):
            n_before = len(self._pending)
            for a, b in _clip(self._ranges, *rng):
                self._exchange(a, b)
                self.stats["overlapped_bytes"] += (b - a) * (2 if self.exchange_dtype == "bf16" else 4)
            if self._early_norm_ok:   # the exchanged gradients of this range are final once its collectives are: their share of the norm now
                for w in self._pending[n_before:]:
                    w.wait()          # (stream-ordered: the communication stream, not the host, waits)
                self._sumsq_ranges(_clip(self._ranges, *rng))
        self._done_ranges.append(rng)

    def _sumsq_ranges(self, ranges):
        """Sum of squares of each WHOLE live range in `ranges`, on the current stream, parked with an event for _grad_norm_sq."""
        whole = set(self._ranges)
        for a, b in ranges:
            if (a, b) not in whole:   # (a tower boundary that cuts a live range: left to the pass after backward)
                continue
            acc = torch.zeros(1, dtype=torch.float32, device=self.arena.grad.device)
            Fx.sumsq(self.arena.grad[a:b], acc)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._norm_early[(a, b)] = (acc, ev)

    def _early_sumsq(self, rng):
        """N = 1: a tower's backward is over (its weight gradients have been joined into the current stream): its ranges' share of the
        gradient norm runs on the tower's weight-gradient side stream (behind whatever that stream still holds for the tower, e.g. the LM
        head's weight gradients that re-join at the end of backward) while the launch stream goes on with the next tower's backward."""
        from ..xroberta import _WgradStream
        cur = torch.cuda.current_stream()
        side = _WgradStream._streams.get((self.arena.grad.device.index, cur.cuda_stream))
        if side is None:
            return self._sumsq_ranges(_clip(self._ranges, *rng))
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self._sumsq_ranges(_clip(self._ranges, *rng))

    # ------------------------------------------------------------------------------------------ step
    def backward_step(self, loss, optimizer, sync=None):
        """`sync=False` (extension; the reference's signature is (loss, optimizer)) keeps this backward's gradients local: a
        multi-source step (Pretrain.py:211-239: web / imagenet / image batches before ONE optimizer step) then exchanges the summed
        gradient once, on its last backward, instead of re-exchanging the arena after every source (the mean is linear)."""
        self._step += 1
        self._sync_now = (self._step % self.accum) == 0 and sync is not False
        self._done_ranges = []
        arena = self.arena
        self._grads_ready()
        if arena is not None:
            arena.reattach()  # a caller's optimizer.zero_grad(set_to_none=True) must not detach .grad from the arena
        self._overlap_ok = (self._dist and self._sync_now and arena is not None and self._comm_stream is not None
                            and arena.live_ver == self._agreed_ver and self._agreed_ver >= 0)
        # (the norm's early shares need the same settled live set; they are taken on the backward that the optimizer step follows)
        self._norm_early = {}
        self._early_norm_ok = (_EARLY_NORM and self._sync_now and arena is not None and arena.grad.is_cuda and self.clip > 0
                               and self.fused_optimizer and arena.live_ver == self._agreed_ver and self._agreed_ver >= 0
                               and (self._overlap_ok or not self._dist))
        if self._sync_now:
            self.stats = {"exchange_bytes": 0, "exchange_calls": 0, "overlapped_bytes": 0}
        loss.backward()
        self._overlap_ok = False
        self._early_norm_ok = False
        if self._dist and self._sync_now:
            if self.timing is not None and self._comm_stream is not None:   # bench.py: end of backward -> all-reduce done
                self.timing[0].record()
            self._finish_allreduce()
            if self.timing is not None and self._comm_stream is not None:
                self.timing[1].record()

    def _finish_allreduce(self):
        arena = self.arena
        if arena is None:
            for p in self.model.parameters():
                if p.grad is not None:
                    dist.all_reduce(p.grad)
                    p.grad.div_(self.world_size)
            return
        prev = list(self._ranges)
        grew = arena.live_ver != self._agreed_ver
        if grew:  # one MAX all-reduce + host read; ranks run the same program, so every rank's set grows on the same step
            self._agree()
        done = sorted(self._done_ranges)
        self.overlapped_ranges = list(done)  # (for logs / tests) what left for the all-reduce from inside backward this step
        pos = 0
        for a, b in done:
            assert a >= pos or a == b, f"arena range ({a}, {b}) was handed to the all-reduce twice in one step"
            pos = max(pos, b)
        todo = _subtract(self._ranges, done)
        if grew and done:  # parameters that became live during this backward inside a range that already left with the old map
            late = _subtract(self._ranges, prev)
            for lo, hi in done:
                todo.extend(_clip(late, lo, hi))
            todo.sort()
        if self._comm_stream is not None:
            self._comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm_stream):
                for a, b in todo:
                    self._exchange(a, b)
            for w in self._pending:
                w.wait()
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        else:
            for a, b in todo:
                self._exchange(a, b, async_ok=False)
        self._pending = []
        for rec in self._use.values():
            rec[0] = 0
        if self._op == dist.ReduceOp.SUM and self.world_size > 1:  # gloo has no AVG: scale the exchanged ranges
            for a, b in self._ranges:
                arena.grad[a:b].mul_(1.0 / self.world_size)

    def grads_ready(self):
        """The current stream may touch the gradient arena: a pending asynchronous zero_grad (optimizer_step) has to land first.
        Called by every accelerator entry; callers that read or write `.grad` themselves right after optimizer_step call it too."""
        from ..arena import grads_ready
        if self.arena is not None and self.arena.grad.is_cuda:
            grads_ready(self.arena.grad.device)

    _grads_ready = grads_ready

    def _zero_async(self):
        arena = self.arena
        if not (_ASYNC_ZERO and arena.grad.is_cuda):
            return arena.zero_grad(self._ranges)
        from ..arena import _PENDING_ZERO
        if self._zero_stream is None:
            self._zero_stream = torch.cuda.Stream()
        self._zero_stream.wait_stream(torch.cuda.current_stream())   # the optimizer kernels have read the gradients
        with torch.cuda.stream(self._zero_stream):
            arena.zero_grad(self._ranges)
            ev = torch.cuda.Event()
            ev.record(self._zero_stream)
        _PENDING_ZERO[arena.grad.device.index] = [ev, {self._zero_stream.cuda_stream}]

    def zero_grad(self):
        if self.arena is not None:
            self._grads_ready()
            self.arena.zero_grad(self.live_ranges())

    def _grad_norm_sq(self):
        """Over the live ranges only: parameters that never receive a gradient are zero (and stay zero), no need to read them."""
        out = torch.zeros(1, dtype=torch.float32, device=self.arena.grad.device)
        early, self._norm_early = self._norm_early, {}
        cur = torch.cuda.current_stream()
        for a, b in self._ranges:   # (range order either way: out = ((0 + s_0) + s_1) + ... whether s_i was taken early or is taken here)
            hit = early.get((a, b))
            if hit is None:
                Fx.sumsq(self.arena.grad[a:b], out)
            else:
                cur.wait_event(hit[1])
                out += hit[0]
        return out

    def optimizer_step(self, optimizer, model, grad_norm: float = 0.0):
        """clip -> step -> zero_grad, returns the total gradient norm (apex_ddp_accelerator.py:100-110)."""
        if self._step % self.accum != 0:
            return 0.0
        arena = self.arena
        if arena is None:
            total = torch.nn.utils.clip_grad_norm_(model.parameters(), self.clip if self.clip > 0 else float("inf"))
            optimizer.step()
            optimizer.zero_grad()
            return float(total)
        arena.reattach()
        self._grads_ready()
        self.live_ranges()
        fused = arena.grad.is_cuda and self.fused_optimizer and _is_adamw(optimizer)
        if not fused:
            # a torch optimizer on the arena views: dead parameters look like `grad is None` to it, as in the reference
            dead = [p for p in arena.params if not arena.is_live(p)]
            for p in dead:
                p.grad = None
            total = torch.nn.utils.clip_grad_norm_([p for p in arena.params if p.grad is not None],
                                                   self.clip if self.clip > 0 else float("inf"))
            optimizer.step()
            arena.bump()
            arena.zero_grad(self._ranges)  # (re-attaches the dead parameters' views as well)
            self.last_grad_norm = total
            return total if arena.grad.is_cuda else float(total)
        norm = self._grad_norm_sq().sqrt()
        clip_coef = None
        if self.clip > 0:
            clip_coef = (self.clip / (norm + 1e-6)).clamp(max=1.0)
        self._fused_adamw(optimizer, clip_coef)
        arena.bump()
        # (xfm_adamw can zero the gradients in its own sweep -- `zero_grad` -- but the fourth store stream costs the kernel more than
        # the separate fills: 837 vs 625 + 170 us per step, measured round 4)
        self._zero_async()
        self.last_grad_norm = norm  # device tensor: no host sync on the step path
        return norm

    # ------------------------------------------------------------------------------------------ fused AdamW
    def _alloc_moments(self):
        if self._m is None:
            self._m = torch.zeros_like(self.arena.data)
            self._v = torch.zeros_like(self.arena.data)

    def _fused_adamw(self, optimizer, clip_coef):
        arena = self.arena
        self._alloc_moments()
        groups = optimizer.param_groups
        if self._group is None:
            assert len(groups) <= 4, "fused AdamW supports up to 4 parameter groups (optim.py:4-50 builds 4)"
            gid = torch.zeros(arena.numel // 256, dtype=torch.uint8)
            for gi, grp in enumerate(groups):
                for p in grp["params"]:
                    o, n = arena.offsets[id(p)]
                    gid[o // 256:(o + n + 255) // 256] = gi
            self._group = gid.to(arena.data.device)
        b1, b2 = groups[0]["betas"]
        lrs = [g["lr"] for g in groups]
        wds = [g.get("weight_decay", 0.0) for g in groups]
        # one launch per run of live parameters that have taken the same number of steps (torch's AdamW counts steps, and so
        # bias-corrects, per parameter: a tower that joined a step later stays one step behind)
        t, units, live = self._t, arena._units, arena.live
        runs = []
        for u, ((a, b), on) in enumerate(zip(units, live)):
            if not on:
                continue
            t[u] += 1
            if runs and runs[-1][1] == a and runs[-1][2] == t[u]:
                runs[-1][1] = b
            else:
                runs.append([a, b, t[u]])
        for a, b, step in runs:
            Fx.adamw(arena.data[a:b], arena.grad[a:b], self._m[a:b], self._v[a:b], self._group[a // 256:(b + 255) // 256],
                     lrs, wds, b1, b2, groups[0]["eps"], step, clip_coef)

    # ---- the moments live in flat arenas; torch's optimizer.state_dict() / load_state_dict() see them as ordinary AdamW state ----
    def _hook_optimizer(self, optimizer):
        if id(optimizer) in self._hooked or not hasattr(optimizer, "register_state_dict_pre_hook"):
            return
        self._hooked.add(id(optimizer))
        # the reference's loop calls optimizer.zero_grad() itself (Pretrain.py:76,127,133); on an arena model that must zero the
        # arena and keep every .grad attached to it (torch 2's set_to_none default would drop the views and leave the arena dirty)
        optimizer.zero_grad = lambda set_to_none=True: self.zero_grad()
        optimizer.register_state_dict_pre_hook(lambda opt: self._publish_optimizer_state(opt))
        optimizer.register_load_state_dict_post_hook(lambda opt: self._adopt_optimizer_state(opt))

    def _views(self, p):
        o, n = self.arena.offsets[id(p)]
        return self._m[o:o + n].view(p.shape), self._v[o:o + n].view(p.shape)

    def _publish_optimizer_state(self, optimizer):
        """Before optimizer.state_dict(): expose step / exp_avg / exp_avg_sq of every stepped parameter in torch.optim.AdamW's
        own format (views into the moment arenas), so the reference's checkpoint code (Pretrain.py:264-274, utils/checkpointer.py)
        saves -- and a later torch or fused run resumes -- the real optimizer state."""
        if self._m is None or not _is_adamw(optimizer):
            return
        arena = self.arena
        for grp in optimizer.param_groups:
            for p in grp["params"]:
                u = arena._unit_of.get(id(p))
                if u is None or self._t[u] == 0:
                    continue
                m, v = self._views(p)
                optimizer.state[p] = {"step": torch.tensor(float(self._t[u])), "exp_avg": m, "exp_avg_sq": v}

    def _adopt_optimizer_state(self, optimizer):
        """After optimizer.load_state_dict() (or at set_up for a state loaded earlier): copy per-parameter AdamW state into the moment
        arenas, take over the step counts and mark those parameters live (they had gradients in the run that wrote the state)."""
        arena = self.arena
        if arena is None or not _is_adamw(optimizer) or not arena.grad.is_cuda or not self.fused_optimizer:
            return
        for grp in optimizer.param_groups:
            for p in grp["params"]:
                st = optimizer.state.get(p)
                u = arena._unit_of.get(id(p))
                if not st or u is None or "exp_avg" not in st:
                    continue
                self._alloc_moments()
                m, v = self._views(p)
                if st["exp_avg"].data_ptr() != m.data_ptr():
                    m.copy_(st["exp_avg"])
                    v.copy_(st["exp_avg_sq"])
                self._t[u] = int(float(st["step"]))
                arena.touch(p)
                optimizer.state[p] = {"step": torch.tensor(float(self._t[u])), "exp_avg": m, "exp_avg_sq": v}

    def state_dict(self):
        """Small metadata only: the moments travel in optimizer.state_dict() (see _publish_optimizer_state)."""
        if self.arena is None:
            return {}
        return {"unit_steps": list(self._t), "live": list(self.arena.live), "step": self._step}

    def load_state_dict(self, sd):
        if not sd:
            return
        if self.arena is None:
            raise RuntimeError("RCCLDDPAccelerator.load_state_dict before set_up (no parameter arena yet)")
        if "m" in sd:  # round-1 format: whole moment arenas + one global step count
            self._alloc_moments()
            if sd["m"].numel() != self._m.numel():
                raise ValueError("optimizer-state arena of another model layout")
            self._m.copy_(sd["m"])
            self._v.copy_(sd["v"])
            for u, on in enumerate(self.arena.live):
                if on:
                    self._t[u] = int(sd["t"])
            return
        if len(sd["unit_steps"]) != len(self._t):
            raise ValueError("accelerator state of another model layout")
        self._t = [int(x) for x in sd["unit_steps"]]
        for u, on in enumerate(sd["live"]):
            if on and not self.arena.live[u]:
                self.arena.live[u] = True
                self.arena.live_ver += 1
        self._step = int(sd.get("step", self._step))


def _is_adamw(opt):
    return type(opt).__name__ in ("AdamW",)
