"""Flat parameter / gradient arenas and bf16 weight caches.

MI355X-first memory plan (288 GB HBM3E per GPU): all fp32 master parameters of a model live in ONE flat buffer and
all gradients in another of the same layout, so that
  * weight gradients are accumulated in place by the wgrad kernels (no per-parameter autograd buffers or add kernels),
  * the data-parallel exchange is an RCCL all-reduce over contiguous arena ranges (no flatten / unflatten copies),
  * the optimiser is one elementwise kernel over the arenas,
  * projections that the reference keeps as separate nn.Linear modules (query/key/value, xroberta.py:170-177) sit
    back to back and are consumed as one fused [3*768, 768] GEMM operand.
Every 2-D weight also keeps a bf16 copy and a transposed bf16 copy resident (forward / dgrad operands of the NT GEMM).

`nn.Parameter` objects keep their identity and names (state_dict compatibility with the reference); only their
storage is re-pointed into the arena.
"""
import os
import weakref

import torch

from . import functional as Fx

_ARENAS = weakref.WeakSet()
_hook_installed = [False]


def _install_optimizer_hook():
    """Any torch optimizer step may rewrite the fp32 masters in place: invalidate every arena's bf16 caches."""
    if _hook_installed[0]:
        return
    try:
        from torch.optim.optimizer import register_optimizer_step_post_hook

        def _bump_all(optimizer, args, kwargs):
            for a in list(_ARENAS):
                a.bump()

        register_optimizer_step_post_hook(_bump_all)
    except ImportError:  # very old torch: callers must call arena.bump() after stepping
        pass
    _hook_installed[0] = True

_CAST_BATCH = os.environ.get("XFM_CAST_BATCH", "1") != "0"  # A/B knob: one batched refresh of the bf16 operand copies per step
_CAST_CHUNKS = max(1, int(os.environ.get("XFM_CAST_CHUNKS", "6")))  # ... as this many launches in first-use order (1: one launch on the using stream)
ALIGN = 256  # elements; arena segments are 1 KiB aligned (vector loads, per-block optimiser groups)


def _round(n, m=ALIGN):
    return (n + m - 1) // m * m


class LinearSlot:
    """One GEMM operand: a weight that is the row-concatenation of `weights` ([n_i, K] each, or conv [n, ...]) and a
    bias that is the concatenation of `biases` (Parameters, or an int = that many structural zeros)."""

    def __init__(self, name, weights, biases=None, pad_k_to=1):
        self.name = name
        self.weights = list(weights)
        self.biases = list(biases) if biases is not None else None
        self.N = sum(w.shape[0] for w in self.weights)
        self.K = self.weights[0].numel() // self.weights[0].shape[0]
        self.ldt = _round(self.N, pad_k_to) if pad_k_to > 1 else _round(self.N, 8)
        self.w = self.b = None
        self._dw = self._db = None
        self._wb = self._wt = None
        self._ver = None
        self._arena = None
        self._listed = False
        self._chunk = 0      # which launch of the arena's batched refresh holds this slot (ParamArena._cast)
        self.need_t = True

    # bf16 operands are (re)built lazily, on first use after the arena version moved: slots a step never touches (the
    # text tower's LM head, both caption heads, the bbox head) are never cast at all
    def _alloc(self):
        if self._wb is None:
            dev = self.w.device
            self._wb = torch.empty((self.N, self.K), dtype=torch.bfloat16, device=dev)
            self._wt = torch.zeros((self.K, self.ldt), dtype=torch.bfloat16, device=dev) if self.need_t else None

    def _ensure(self):
        a = self._arena
        if self._ver != a._manual_ver:
            a._cast(self)
        if a._batch_event is not None:
            a._batch_wait(self._chunk)

    # Gradient views are handed out through properties: every launch site that accumulates into the gradient arena asks for
    # `slot.dw` / `slot.db`, which is what marks the slot's parameters as LIVE (ParamArena.touch) -- the data-parallel exchange,
    # the optimizer and zero_grad work on whole live parameters, never on `grad != 0`.
    @property
    def dw(self):
        self._arena.touch_all(self.weights)
        return self._dw

    @property
    def db(self):
        if self._db is not None:
            self._arena.touch_all(self.biases)
        return self._db

    @property
    def wb(self):
        self._ensure()
        return self._wb

    @property
    def wt(self):
        self._ensure()
        return self._wt


class OwnsArena:
    """Mixin for modules that can build their own ParamArena (stand-alone towers): `zero_grad()` must zero the arena and keep every
    .grad attached to it (nn.Module.zero_grad defaults to set_to_none=True, which would drop the views and leave the arena dirty)."""
    _own_arena = False

    def zero_grad(self, set_to_none=False):
        arena = getattr(self, "_arena", None)
        if arena is not None and self._own_arena:
            arena.zero_grad()
        else:
            super().zero_grad(set_to_none=set_to_none)


_PENDING_ZERO = {}   # device index -> [event of the last asynchronous zero_grad (RCCLDDPAccelerator._zero_async), ids of the streams that wait for it]


def grads_ready(device=None):
    """Make the current stream wait for the last asynchronous zero_grad of the gradient arena(s) on `device` (default: current).  The
    accelerator calls it before every backward / optimizer step, every tower's backward calls it on ITS stream, and code that reads or
    writes parameter gradients on its own right after an accelerator step (a probe, a test) calls it first.  Every stream waits once
    per zero_grad (a wait for an event that has long fired is free); no-op when there has been none."""
    if not _PENDING_ZERO:
        return
    dev = torch.cuda.current_device() if device is None else (device.index if isinstance(device, torch.device) else int(device))
    rec = _PENDING_ZERO.get(dev)
    if rec is None:
        return
    cur = torch.cuda.current_stream(dev)
    if cur.cuda_stream not in rec[1]:
        cur.wait_event(rec[0])
        rec[1].add(cur.cuda_stream)


def grad_of(p):
    """Gradient view of a parameter inside its arena (re-attached if a caller dropped .grad); marks the parameter live -- every
    kernel launch site that accumulates a parameter gradient gets its destination through here or through slot.dw / slot.db."""
    p._xfm_arena.touch(p)
    if p.grad is not p._xfm_grad:
        p.grad = p._xfm_grad
    return p._xfm_grad


class ParamArena:
    def __init__(self, module, slots=(), device=None):
        params = []
        seen = set()
        for name, p in module.named_parameters():
            if id(p) not in seen:
                seen.add(id(p))
                params.append((name, p))
        device = device or params[0][1].device
        self.device = device
        self.slots = list(slots)
        # ---- layout: module (= tower) order; a fused group is placed, members adjacent and unpadded, where its first
        # member appears, so every tower owns ONE contiguous arena range (per-tower collectives, accelerators/)
        layout = []   # (param or None, numel, offset)
        placed = set()
        off = 0
        group_of = {}
        for s in self.slots:
            for kind, plist in (("w", s.weights), ("b", s.biases)):
                if plist is None:
                    continue
                for p in plist:
                    if not isinstance(p, int):
                        assert id(p) not in group_of, f"{s.name}: parameter used by two slots"
                        group_of[id(p)] = (s, kind, plist)
        group_ranges = {}
        for name, p in params:
            if id(p) in placed:
                continue
            if id(p) in group_of:
                s, kind, plist = group_of[id(p)]
                start = off
                for q in plist:
                    if isinstance(q, int):
                        layout.append((None, q, off))
                        off += q
                    else:
                        placed.add(id(q))
                        layout.append((q, q.numel(), off))
                        off += q.numel()
                group_ranges[(id(s), kind)] = (start, off)
                off = _round(off)
            else:
                placed.add(id(p))
                layout.append((p, p.numel(), off))
                off = _round(off + p.numel())
        self.numel = _round(off)
        # ---- liveness: a parameter is LIVE once any backward has written a gradient for it (launch sites ask for the gradient
        # view through grad_of() / slot.dw / slot.db; autograd-fed parameters are caught by a tensor hook).  Live parameters are
        # exchanged, clipped, stepped and zeroed as a whole, dead ones are skipped like `p.grad is None` in the reference's
        # AdamW (optim.py:4-50 / DDP find_unused_parameters).  A fused group (q|k|v rows of one GEMM operand) is one unit.
        self._units = []      # (start, end) 1-KiB aligned arena extents, in layout order
        self._unit_of = {}    # id(param) -> unit index
        extents = sorted((a, _round(b)) for a, b in group_ranges.values())
        gi, cur = 0, None
        for p, n, o in layout:
            if cur is None or o >= cur[1]:
                while gi < len(extents) and extents[gi][1] <= o:
                    gi += 1
                cur = extents[gi] if gi < len(extents) and extents[gi][0] <= o else (o, _round(o + n))
                self._units.append(cur)
            if p is not None:
                self._unit_of[id(p)] = len(self._units) - 1
        self.live = [False] * len(self._units)
        self.live_ver = 0
        self.data = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.offsets = {}
        for p, n, o in layout:
            if p is None:
                continue
            view = self.data[o:o + n].view(p.shape)
            view.copy_(p.data.to(device=device, dtype=torch.float32))
            p.data = view
            p._xfm_grad = self.grad[o:o + n].view(p.shape)
            p.grad = p._xfm_grad
            p._xfm_arena = self
            self.offsets[id(p)] = (o, n)
            if p.requires_grad:  # gradients that arrive through autograd's AccumulateGrad (e.g. the ITC temperature)
                p.register_hook(lambda g, _p=p: self.touch(_p))
        self.names = {id(p): name for name, p in params}
        for s in self.slots:
            s._arena = self
            wr, br = group_ranges[(id(s), "w")], group_ranges.get((id(s), "b"))
            s.w = self.data[wr[0]:wr[1]].view(s.N, s.K)
            s._dw = self.grad[wr[0]:wr[1]].view(s.N, s.K)
            if br is not None:
                s.b = self.data[br[0]:br[1]]
                s._db = self.grad[br[0]:br[1]]
                assert s.b.numel() == s.N, s.name
        self._manual_ver = 0
        # slots a step has touched: after the next version bump their bf16 copies are rebuilt by ONE batched launch, on
        # the stream of the first slot used; other streams wait for its event the first time they read a copy
        self._active = []
        self._table = None
        self._batch_ver = -1
        self._batch_event = None
        self._batch_events = []
        self._batch_synced = {}
        self._cast_stream = None
        self.params = [p for _, p in params]
        _ARENAS.add(self)
        _install_optimizer_hook()
        module.register_load_state_dict_post_hook(lambda mod, keys: self.bump())

    def _cast(self, slot):
        ver = self._manual_ver
        if _CAST_BATCH and slot._listed and self._batch_ver != ver and len(self._active) > 1:
            # The refresh (0.5 ms of HBM traffic for ~290 M weights) runs as _CAST_CHUNKS launches over the slots IN THE ORDER OF THEIR
            # FIRST USE, on a stream of its own, one event per launch: a tower starts as soon as the launch that holds its first layer
            # is done (the ViT after ~0.15 ms instead of 0.5) and the later launches run under its first GEMMs.
            if self._table is None:
                total = sum(s.w.numel() for s in self._active)
                per, acc, chunks = total / max(1, min(_CAST_CHUNKS, len(self._active))), 0, [[]]
                for s in self._active:
                    if acc >= per * len(chunks) and len(chunks) < _CAST_CHUNKS:
                        chunks.append([])
                    chunks[-1].append(s)
                    acc += s.w.numel()
                for k, group in enumerate(chunks):
                    for s in group:
                        s._chunk = k
                self._table = [Fx.cast_table([(s.w, s._wb, s._wt) for s in group], self.device) for group in chunks]
            cur = torch.cuda.current_stream()
            stream = cur
            if len(self._table) > 1:
                if self._cast_stream is None:
                    self._cast_stream = torch.cuda.Stream(device=self.device)
                stream = self._cast_stream
                stream.wait_stream(cur)   # (the optimizer's update of the fp32 weights is in `cur`'s past)
            self._batch_events = []
            with torch.cuda.stream(stream):
                for tbl in self._table:
                    Fx.cast_transpose_batch(*tbl)
                    ev = torch.cuda.Event()
                    ev.record(stream)
                    self._batch_events.append(ev)
            for s in self._active:
                s._ver = ver
            self._batch_ver = ver
            self._batch_event = self._batch_events[-1]
            self._batch_synced = {stream.cuda_stream: len(self._table) - 1}
            return
        slot._alloc()
        Fx.cast_transpose(slot.w, slot._wb, slot._wt)
        slot._ver = ver
        if not slot._listed:
            slot._listed = True
            self._active.append(slot)
            self._table = None

    def _batch_wait(self, chunk=None):
        """The current stream waits for launch `chunk` of the batched refresh (None: all of it); launches finish in order."""
        if chunk is None:
            chunk = len(self._batch_events) - 1
        sid = Fx._stream()
        if self._batch_synced.get(sid, -1) < chunk:
            torch.cuda.current_stream().wait_event(self._batch_events[chunk])
            self._batch_synced[sid] = chunk

    # bf16 caches are valid for one version; bumped by optimiser steps, load_state_dict and explicit bump()
    def version(self):
        return self._manual_ver

    def bump(self):
        self._manual_ver += 1

    # ---- liveness -------------------------------------------------------------------------------
    def touch(self, p):
        u = self._unit_of[id(p)]
        if not self.live[u]:
            self.live[u] = True
            self.live_ver += 1

    def touch_all(self, plist):
        for p in plist:
            if not isinstance(p, int):
                self.touch(p)
                if p.grad is not p._xfm_grad:  # a caller's zero_grad(set_to_none=True) dropped the view: the gradient lives here
                    p.grad = p._xfm_grad

    def is_live(self, p):
        return self.live[self._unit_of[id(p)]]

    def live_ranges(self, cuts=()):
        """Maximal contiguous (start, end) arena ranges of live parameters, additionally split at `cuts` (tower boundaries)."""
        out = []
        for (a, b), on in zip(self._units, self.live):
            if not on:
                continue
            if out and out[-1][1] == a:
                out[-1][1] = b
            else:
                out.append([a, b])
        res = []
        for a, b in out:
            pts = [a] + [c for c in sorted(cuts) if a < c < b] + [b]
            res.extend(zip(pts[:-1], pts[1:]))
        return res

    def reattach(self):
        """Point every parameter's .grad back at its arena view.  A caller that ran `optimizer.zero_grad()` (set_to_none=True is
        torch 2's default; the reference's loop calls it after every optimizer step, Pretrain.py:76,127,133) drops the views: a
        later autograd accumulation would then land in a fresh tensor the fused optimizer never sees.  A stray gradient tensor
        that was accumulated while detached is folded into the arena."""
        for p in self.params:
            if p.grad is not p._xfm_grad:
                if p.grad is not None:
                    p._xfm_grad.add_(p.grad.to(p._xfm_grad.dtype))
                    self.touch(p)
                p.grad = p._xfm_grad

    def zero_grad(self, ranges=None):
        """ranges: the (start, end) arena ranges of live parameters (dead ranges are never written and stay zero); None = everything."""
        if ranges is None:
            if self.grad.is_cuda:
                grads_ready(self.grad.device)
            self.grad.zero_()
        else:
            for a, b in ranges:
                self.grad[a:b].zero_()
        self.reattach()

    def attached(self):
        """True while the parameters still live in the arena (a .to()/.cuda() after finalize() breaks this; probing the
        first and last parameter is enough because Module._apply moves all of them)."""
        return all(p.data_ptr() == self.data.data_ptr() + 4 * self.offsets[id(p)][0] for p in (self.params[0], self.params[-1]))

    def range_of(self, params):
        lo = min(self.offsets[id(p)][0] for p in params)
        hi = max(self.offsets[id(p)][0] + self.offsets[id(p)][1] for p in params)
        return lo, _round(hi)
