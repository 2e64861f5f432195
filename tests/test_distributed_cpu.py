"""N > 1 path on CPU: world_size-2 `gloo` runs of the pieces that talk across ranks --
  * RCCLDDPAccelerator: weight broadcast at set_up, live-range discovery, arena all-reduce (mean), skip of never-used ranges,
    `.module` wrapper, clip -> step -> zero_grad contract (accelerators/ddp_accelerator.py:34-98 of the reference);
  * the ITC all-gather with slice-only backward (xfm.py:81-101) against a single-process loss on the concatenated batch.
The arena / accelerator code is device-agnostic; only the kernels need a GPU."""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F


class Tiny(nn.Module):
    """Two 'towers' + an unused head, laid out in a ParamArena like XFMBase."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.fusion_encoder = nn.Module()
        self.fusion_encoder.roberta = nn.Linear(8, 8)
        self.text_encoder = nn.Module()
        self.text_encoder.roberta = nn.Linear(8, 8)
        self.vision_encoder = nn.Linear(8, 8)
        self.unused_head = nn.Linear(8, 300)   # never receives a gradient (cf. lm_cap_head / bbox_head in XFM)
        self._arena = None

    def finalize(self, device=None):
        from xfm_amd.arena import ParamArena
        self._arena = ParamArena(self, (), device or torch.device("cpu"))
        return self

    def forward(self, x):
        return self.fusion_encoder.roberta(self.text_encoder.roberta(self.vision_encoder(x)))


def _accelerator_worker(rank, world, init_file, out):
    from xfm_amd.accelerators import ACCELERATOR_MAP
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    model = Tiny()
    if rank == 1:  # perturb rank 1: set_up must broadcast rank 0's weights
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    acc = ACCELERATOR_MAP["RCCLDDP"]({"RNG_SEED": 1, "CLIP_GRAD_NORM": 0.0, "GRAD_ACCUMULATE_STEPS": 1})
    wrapped, opt, _ = acc.set_up(model, opt, None, local_rank=rank, world_size=world, rank=rank)
    assert wrapped.module is model
    w0 = model.vision_encoder.weight.detach().clone()
    torch.manual_seed(100 + rank)
    x = torch.randn(4, 8)
    loss = wrapped(x).pow(2).mean()
    acc.backward_step(loss, opt)
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    acc.optimizer_step(opt, model)
    live = list(acc._ranges)  # whole live parameters, agreed across ranks
    torch.save({"w0": w0, "grads": grads, "live": live, "x": x, "numel": model._arena.numel,
                "unused_range": model._arena.range_of(list(model.unused_head.parameters())),
                "w1": model.vision_encoder.weight.detach().clone(),
                "grad_after": float(model._arena.grad.abs().max())}, out + f".{rank}")
    dist.destroy_process_group()


def _spawn(fn, *args):
    with tempfile.TemporaryDirectory() as d:
        init = os.path.join(d, "init")
        out = os.path.join(d, "out")
        mp.spawn(fn, args=(2, init, out) + args, nprocs=2, join=True)
        return [torch.load(out + f".{r}") for r in range(2)]


def test_accelerator_broadcast_allreduce_and_step_world2():
    r0, r1 = _spawn(_accelerator_worker)
    assert torch.equal(r0["w0"], r1["w0"]), "set_up must broadcast rank 0's weights"
    # reference: single process, mean of the two ranks' gradients
    ref = Tiny()
    gs = []
    for r in (r0, r1):
        ref.zero_grad()
        ref(r["x"]).pow(2).mean().backward()
        gs.append({n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None})
    for n in gs[0]:
        if n.startswith("unused_head"):
            continue
        want = (gs[0][n] + gs[1][n]) / 2
        assert torch.allclose(r0["grads"][n], want, atol=1e-6), n
        assert torch.allclose(r1["grads"][n], want, atol=1e-6), n
    # the never-used head is outside every live chunk -> not exchanged
    lo, hi = r0["unused_range"]
    for a, b in r0["live"]:
        assert b <= lo or a >= hi, f"live chunk {(a, b)} overlaps the unused range {(lo, hi)}"
    assert r0["live"] == r1["live"]
    # SGD step applied with the averaged gradient, identically on both ranks; gradients zeroed afterwards
    want_w1 = r0["w0"] - 0.1 * (gs[0]["vision_encoder.weight"] + gs[1]["vision_encoder.weight"]) / 2
    assert torch.allclose(r0["w1"], want_w1, atol=1e-6) and torch.equal(r0["w1"], r1["w1"])
    assert r0["grad_after"] == 0.0


def _multi_source_worker(rank, world, init_file, out):
    """Pretrain.py:219-239: several backward_steps (web, imagenet, image batches) before ONE optimizer_step."""
    from xfm_amd.accelerators import ACCELERATOR_MAP
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    model = Tiny()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    acc = ACCELERATOR_MAP["RCCLDDP"]({"RNG_SEED": 1, "CLIP_GRAD_NORM": 0.0, "GRAD_ACCUMULATE_STEPS": 1})
    wrapped, opt, _ = acc.set_up(model, opt, None, local_rank=rank, world_size=world, rank=rank)
    w0 = model.vision_encoder.weight.detach().clone()
    xs = []
    for k in range(3):  # three sources, one backward each, gradients accumulate in the arena
        torch.manual_seed(1000 * k + rank)
        x = torch.randn(4, 8)
        xs.append(x)
        # the FIRST source only reaches the vision tower (like an ImageNet MIM batch): the live-range map must not be frozen on it
        y = model.vision_encoder(x) if k == 0 else wrapped(x)
        acc.backward_step(y.pow(2).mean() * (k + 1), opt)
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    acc.optimizer_step(opt, model)
    torch.save({"w0": w0, "xs": xs, "grads": grads, "w1": model.vision_encoder.weight.detach().clone(), "live": list(acc._ranges),
                "text_range": model._arena.range_of(list(model.text_encoder.parameters()))}, out + f".{rank}")
    dist.destroy_process_group()


def test_several_backward_steps_before_one_optimizer_step_world2():
    """The reference's Apex DDP all-reduces at the end of every backward; re-averaging the already averaged part of the arena leaves
    it unchanged, so after three sources the gradient is the sum over sources of the rank-mean."""
    r0, r1 = _spawn(_multi_source_worker)
    ref = Tiny()
    want = None
    for k in range(3):
        per_rank = []
        for r in (r0, r1):
            ref.zero_grad()
            y = ref.vision_encoder(r["xs"][k]) if k == 0 else ref(r["xs"][k])
            (y.pow(2).mean() * (k + 1)).backward()
            per_rank.append({n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None})
        mean = {n: (per_rank[0][n] + per_rank[1][n]) / 2 for n in per_rank[0]}
        want = dict(mean) if want is None else {n: want.get(n, 0) + mean[n] for n in mean}
    for n, g in want.items():
        if n.startswith("unused_head"):
            continue
        assert torch.allclose(r0["grads"][n], g, atol=2e-6), n
        assert torch.equal(r0["grads"][n], r1["grads"][n]), n
    assert torch.allclose(r0["w1"], r0["w0"] - 0.1 * want["vision_encoder.weight"], atol=2e-6) and torch.equal(r0["w1"], r1["w1"])
    # every tower that got a gradient from ANY source is in the live map (and so is stepped by the fused optimizer on a GPU)
    lo, hi = r0["text_range"]
    assert any(a <= lo and hi <= b or (a < hi and lo < b) for a, b in r0["live"]), (r0["live"], r0["text_range"])


def _bf16_exchange_worker(rank, world, init_file, out, dtype):
    """GRAD_EXCHANGE_DTYPE = bf16 and one exchange per optimizer step (sync=False on all but the last source)."""
    from xfm_amd.accelerators import ACCELERATOR_MAP
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    model = Tiny()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    acc = ACCELERATOR_MAP["RCCLDDP"]({"RNG_SEED": 1, "CLIP_GRAD_NORM": 0.0, "GRAD_ACCUMULATE_STEPS": 1, "GRAD_EXCHANGE_DTYPE": dtype})
    wrapped, opt, _ = acc.set_up(model, opt, None, local_rank=rank, world_size=world, rank=rank)
    assert acc.takes_sync_hint
    xs, local = [], None
    for k in range(3):
        torch.manual_seed(77 * k + rank)
        x = torch.randn(4, 8)
        xs.append(x)
        acc.backward_step(wrapped(x).pow(2).mean(), opt, sync=(k == 2))
        if k == 1:   # nothing has been exchanged yet: the arena holds this rank's own sum
            local = model._arena.grad.detach().clone()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    acc.optimizer_step(opt, model)
    torch.save({"xs": xs, "grads": grads, "local": local, "arena": model._arena.data.detach().clone()}, out + f".{rank}")
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_one_exchange_per_step_in_bf16_or_fp32_keeps_replicas_identical_world2(dtype):
    r0, r1 = _spawn(_bf16_exchange_worker, dtype)
    ref = Tiny()
    per_rank = []
    for r in (r0, r1):
        ref.zero_grad()
        for x in r["xs"]:
            ref(x).pow(2).mean().backward()
        per_rank.append({n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None})
    # before the last source the arenas differ (no exchange yet) ...
    assert not torch.equal(r0["local"], r1["local"])
    # ... after it every rank holds the mean of the ranks' summed gradients: exactly in fp32, to bf16 resolution otherwise
    tol = 1e-6 if dtype == "fp32" else 1e-2
    for n in per_rank[0]:
        want = (per_rank[0][n] + per_rank[1][n]) / 2
        scale = float(want.abs().max()) + 1e-12
        assert float((r0["grads"][n] - want).abs().max()) <= tol * scale, n
        assert torch.equal(r0["grads"][n], r1["grads"][n]), f"{n}: replicas must hold bit-identical exchanged gradients"
    assert torch.equal(r0["arena"], r1["arena"]), "parameters must stay bit-identical across ranks"


def _itc_worker(rank, world, init_file, out):
    from xfm_amd.xfm import allgather
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    torch.manual_seed(7)
    img_all = F.normalize(torch.randn(8, 16), dim=-1)
    txt_all = F.normalize(torch.randn(8, 16), dim=-1)
    img = img_all[rank * 4:(rank + 1) * 4].clone().requires_grad_(True)
    txt = txt_all[rank * 4:(rank + 1) * 4].clone().requires_grad_(True)
    ia, ta = allgather(img), allgather(txt)
    logits = ia @ ta.t() / 0.07
    labels = torch.arange(8)
    loss = (F.cross_entropy(logits, labels) + F.cross_entropy(logits.t(), labels)) / 2
    loss.backward()
    torch.save({"loss": float(loss), "gi": img.grad, "gt": txt.grad, "img_all": img_all, "txt_all": txt_all}, out + f".{rank}")
    dist.destroy_process_group()


def test_itc_allgather_slice_backward_world2():
    from oracle import xfm_oracle as O
    r0, r1 = _spawn(_itc_worker)
    img = r0["img_all"].clone().requires_grad_(True)
    txt = r0["txt_all"].clone().requires_grad_(True)
    ref = O.contrastive_loss(img, txt, torch.tensor(0.07))
    ref.backward()
    assert abs(r0["loss"] - float(ref)) < 1e-6 and abs(r1["loss"] - float(ref)) < 1e-6
    # slice-only backward (xfm.py:93-98): each rank keeps d(loss_on_this_rank)/d(its own slice), no reduce-scatter;
    # the full-batch gradient is the SUM over ranks of those slices' contributions, i.e. world * (averaged DDP gradient)
    # for the rows a rank owns it sees exactly the single-process gradient of those rows
    assert torch.allclose(r0["gi"], img.grad[:4], atol=1e-6) and torch.allclose(r1["gi"], img.grad[4:], atol=1e-6)
    assert torch.allclose(r0["gt"], txt.grad[:4], atol=1e-6) and torch.allclose(r1["gt"], txt.grad[4:], atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------
# Liveness is per whole parameter and structural (never `grad != 0`): a text-only first step must not freeze the other
# towers, and row-sparse embedding gradients must still exchange / decay / step every row (optim.py:4-50 of the reference
# sees a dense `.grad`; torch-1.x zero_grad() zeroes in place, AdamW skips only `grad is None`).
# ---------------------------------------------------------------------------------------------------------------------
class Tiny2(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.fusion_encoder = nn.Module()
        self.fusion_encoder.roberta = nn.Linear(8, 8)
        self.text_encoder = nn.Module()
        self.text_encoder.roberta = nn.Embedding(2000, 8)   # 2000 rows: a batch of 4 tokens touches 0.2 % of them
        self.text_encoder.head = nn.Linear(8, 8)
        self.vision_encoder = nn.Linear(8, 8)
        self.unused_head = nn.Linear(8, 300)
        self._arena = None

    def finalize(self, device=None):
        from xfm_amd.arena import ParamArena
        self._arena = ParamArena(self, (), device or torch.device("cpu"))
        return self

    def forward_text(self, ids):
        return self.text_encoder.head(self.text_encoder.roberta(ids))

    def forward(self, x, ids):
        return self.fusion_encoder.roberta(self.forward_text(ids) + self.vision_encoder(x))


def _tiny2_batches(rank):
    g = torch.Generator().manual_seed(50 + rank)
    ids1 = torch.randint(0, 20, (4,), generator=g)          # step 1: text only, rows < 20
    ids2 = torch.randint(1000, 2000, (4,), generator=g)     # step 2: image + text, other rows
    ids3 = torch.randint(0, 2000, (4,), generator=g)
    xs = [torch.randn(4, 8, generator=g) for _ in range(3)]
    return [(None, ids1), (xs[1], ids2), (xs[2], ids3)]


def _tiny2_loss(model, x, ids):
    y = model.forward_text(ids) if x is None else model(x, ids)
    return y.pow(2).mean()


def _liveness_worker(rank, world, init_file, out):
    from xfm_amd.accelerators import ACCELERATOR_MAP
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    model = Tiny2()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=0.1, betas=(0.9, 0.98), eps=1e-8)
    acc = ACCELERATOR_MAP["RCCLDDP"]({"RNG_SEED": 1, "CLIP_GRAD_NORM": 1.0, "GRAD_ACCUMULATE_STEPS": 1})
    wrapped, opt, _ = acc.set_up(model, opt, None, local_rank=rank, world_size=world, rank=rank)
    ranges = []
    for x, ids in _tiny2_batches(rank):
        opt.zero_grad()  # the reference's loop does this (Pretrain.py:127); torch 2's set_to_none must not detach the arena
        acc.backward_step(_tiny2_loss(model, x, ids), opt)
        acc.optimizer_step(opt, model)
        opt.zero_grad()
        ranges.append(list(acc._ranges))
    torch.save({"params": {n: p.detach().clone() for n, p in model.named_parameters()}, "ranges": ranges,
                "unused_range": model._arena.range_of(list(model.unused_head.parameters())),
                "vision_range": model._arena.range_of(list(model.vision_encoder.parameters())),
                "grad_after": float(model._arena.grad.abs().max())}, out + f".{rank}")
    dist.destroy_process_group()


def test_text_only_first_step_and_sparse_embedding_rows_world2():
    r0, r1 = _spawn(_liveness_worker)
    # single-process reference: mean of the two ranks' gradients, torch.optim.AdamW, clip 1.0, in-place zero_grad
    ref = Tiny2()
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=0.1, betas=(0.9, 0.98), eps=1e-8)
    b0, b1 = _tiny2_batches(0), _tiny2_batches(1)
    for (x0, i0), (x1, i1) in zip(b0, b1):
        opt.zero_grad(set_to_none=False)
        ((_tiny2_loss(ref, x0, i0) + _tiny2_loss(ref, x1, i1)) / 2).backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        opt.step()
    want = dict(ref.named_parameters())
    for n, p in r0["params"].items():
        assert torch.allclose(p, want[n].detach(), atol=2e-6), n
        assert torch.equal(p, r1["params"][n]), f"replicas diverged on {n}"
    # step 1 (text only) leaves the vision / fusion towers dead, step 2 brings them in; the unused head never joins
    vlo, vhi = r0["vision_range"]
    assert not any(a < vhi and vlo < b for a, b in r0["ranges"][0]), "vision tower must be dead after a text-only step"
    assert any(a <= vlo and vhi <= b for a, b in r0["ranges"][1]), "vision tower must be live from the first image step on"
    ulo, uhi = r0["unused_range"]
    assert not any(a < uhi and ulo < b for a, b in r0["ranges"][2])
    assert r0["ranges"] == r1["ranges"] and r0["grad_after"] == 0.0
    # rows of the embedding no batch ever touched were still weight-decayed (dense-gradient semantics)
    emb0 = Tiny2().text_encoder.roberta.weight
    assert not torch.equal(r0["params"]["text_encoder.roberta.weight"][500], emb0[500].detach())


def test_interval_helpers():
    from xfm_amd.accelerators.rccl_ddp_accelerator import _clip, _subtract
    assert _clip([(0, 10), (20, 30)], 5, 25) == [(5, 10), (20, 25)]
    assert _subtract([(0, 10), (20, 30)], [(2, 4), (8, 22)]) == [(0, 2), (4, 8), (22, 30)]
    assert _subtract([(0, 10)], []) == [(0, 10)]
    assert _subtract([(0, 10)], [(0, 10)]) == []
